"""The Rust side of the drop-in (rust/align3d-hip) cannot be compiled in this image (no rustc / cargo), so it is
checked as text: the raw bindings against the header and against the ctypes table the GPU tests call through, and the
wrappers against the reference's public signatures."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "align3d-hip")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from align3d_amd import _abi  # noqa: E402


def _rust_functions():
    text = open(os.path.join(CRATE, "src", "sys.rs")).read()
    block = text[text.index('extern "C" {'):]
    fns = {}
    for m in re.finditer(r"pub fn (a3d_\w+)\((.*?)\)( -> ([^;]+))?;", block):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        fns[m.group(1)] = ([a.split(":", 1)[1].strip() for a in args], m.group(4))
    return text, fns


def test_sys_rs_is_the_generators_output_for_the_current_header():
    import gen_rust_sys

    assert open(os.path.join(CRATE, "src", "sys.rs")).read() == gen_rust_sys.generate(), \
        "include/align3d_hip.h changed: run python scripts/gen_rust_sys.py"


def test_every_header_symbol_is_bound_with_the_right_arity_and_pointer_kinds():
    header = open(os.path.join(ROOT, "include", "align3d_hip.h")).read()
    # (the entry points of the diagnostics build are not part of what a Rust host binds)
    header = re.sub(r"#ifdef A3D_DIAGNOSTICS.*?#endif /\* A3D_DIAGNOSTICS \*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(a3d_[a-z0-9_]+)\s*\(", header))
    text, fns = _rust_functions()
    assert set(fns) == declared == set(_abi.SIGNATURES)
    import ctypes as C

    for name, (restype, argtypes) in _abi.SIGNATURES.items():
        rust_args, rust_ret = fns[name]
        assert len(rust_args) == len(argtypes), name
        assert (rust_ret is None) == (restype is None), name
        for ra, ca in zip(rust_args, argtypes):  # a pointer on one side is a pointer on the other
            c_is_ptr = ca in (C.c_void_p, C.c_char_p) or hasattr(ca, "contents") or ca is _abi._PP
            assert ra.startswith("*") == c_is_ptr, (name, ra, ca)
    # the POD structs, field for field (names and order) against the header
    for struct in ("a3d_icp_params", "a3d_pose", "a3d_range_image_view", "a3d_point_cloud_view", "a3d_gn_state",
                   "a3d_builder_params"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
        c_body = re.sub(r"/\*.*?\*/", "", c_body, flags=re.S)
        c_fields = []
        for decl in c_body.split(";"):
            decl = decl.strip()
            if decl:
                c_fields += [re.sub(r"[\*\s]|\[\d+\]", "", f).split()[-1] if " " in f.strip() else re.sub(r"[\*\s]|\[\d+\]", "", f)
                             for f in re.sub(r"^(const\s+)?\w+\s*\**", "", decl, count=1).split(",")]
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, text, re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", r_body)
        assert r_fields == c_fields, (struct, r_fields, c_fields)
    # every status code
    for code, value in re.findall(r"(A3D_[A-Z_]+) = (\d+)", header):
        assert f"pub const {code}: a3d_status = {value};" in text


def test_wrappers_keep_the_reference_signatures_and_only_call_bound_symbols():
    _, fns = _rust_functions()
    want = {
        "src/icp/multiscale.rs": [  # src/icp/multiscale.rs:26,51
            "pub fn new(params: MsIcpParams, target_pyramid: &'pyramid_lt Vec<RangeImage>) -> Result<Self, A3dError>",
            "pub fn align(&self, source_pyramid: &[RangeImage]) -> Transform",
            "The number of range images pyramid levels and ICP parameters must be equal."],
        "src/icp/image_icp.rs": [  # src/icp/image_icp.rs:26,43
            "pub fn new(params: IcpParams, target: &'target_lt RangeImage) -> Self",
            "pub fn align(&self, source: &RangeImage) -> Transform", "pub initial_transform: Transform"],
        "src/icp/pcl_icp.rs": [  # src/icp/pcl_icp.rs:31,49
            "pub fn new(params: IcpParams, target: &'target PointCloud) -> Self",
            "pub fn align(&self, source: &PointCloud) -> Transform"],
        "src/kdtree.rs": [  # src/kdtree.rs:28,69
            "pub fn new(points: &ArrayView1<Vector3<f32>>) -> Self",
            "pub fn nearest(&self, point: &Vector3<f32>) -> (usize, f32)"],
        "src/range_image.rs": ["fn compute_normals_hip(&mut self) -> &mut Self"],  # structure.rs:184
        "src/bilateral.rs": ["-> Array2<u16>"],  # edge_aware_filter.rs:126
    }
    used = set()
    for rel, needles in want.items():
        text = " ".join(open(os.path.join(CRATE, rel)).read().split())
        for n in needles:
            assert n in text, (rel, n)
    for dirpath, _, files in os.walk(os.path.join(CRATE, "src")):
        for f in files:
            if f.endswith(".rs") and f != "sys.rs":
                used |= set(re.findall(r"sys::(a3d_\w+)\s*\(", open(os.path.join(dirpath, f)).read()))
    assert used and used <= set(fns), used - set(fns)
    # the entry points SURVEY §8b lists are all reachable from the wrappers
    for must in ("a3d_multiscale_new", "a3d_multiscale_align", "a3d_image_icp_align", "a3d_kdtree_new", "a3d_kdtree_nearest",
                 "a3d_pcl_icp_new", "a3d_pcl_icp_align", "a3d_compute_normals", "a3d_bilateral_filter_u16",
                 "a3d_range_image_build_pyramids", "a3d_multiscale_batch_new_multi"):
        assert must in used, must
    assert "UNVERIFIABLE" in open(os.path.join(CRATE, "src", "sys.rs")).read().upper()
