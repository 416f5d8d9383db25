// Links libalign3d_hip.so (built by `make -C align3d_amd/csrc`, or __graft_entry__.build()).
// ALIGN3D_HIP_LIB_DIR overrides the in-tree location.
fn main() {
    let dir = std::env::var("ALIGN3D_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../align3d_amd/csrc", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=align3d_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ALIGN3D_HIP_LIB_DIR");
}
