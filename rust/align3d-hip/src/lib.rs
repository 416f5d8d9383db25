//! `align3d-hip`: the hot path of otaviog/align3d on one or more MI355X GPUs, behind the reference's own API.
//!
//! Every public item here has the name and signature of the reference item it stands in for, so a caller switches by
//! changing `use align3d::icp::multiscale::MultiscaleAlign` to `use align3d_hip::icp::multiscale::MultiscaleAlign`
//! (same for `ImageIcp`, `Icp`, `R3dTree`, `RangeImageHip::compute_normals`, `BilateralFilter`, `RangeImageBuilder`):
//!
//! | reference item | file:line | here |
//! |---|---|---|
//! | `MultiscaleAlign::new / align` | src/icp/multiscale.rs:26,51 | `icp::multiscale` |
//! | `ImageIcp::new / align / initial_transform` | src/icp/image_icp.rs:26,43 | `icp::image_icp` |
//! | `Icp::new / align` | src/icp/pcl_icp.rs:31,49 | `icp::pcl_icp` |
//! | `R3dTree::new / nearest` | src/kdtree.rs:28,69 | `kdtree` |
//! | `RangeImage::compute_normals` | src/range_image/structure.rs:184 | `range_image::ComputeNormalsHip` |
//! | `RangeImageBuilder::build` | src/range_image/builder.rs:74 | `range_image::RangeImageBuilder` |
//! | `BilateralFilter::<u16>::filter` | src/bilateral/edge_aware_filter.rs:126 | `bilateral` |
//!
//! Where the reference panics (`expect`, `unwrap`) these wrappers panic with the same message: the C ABI reports
//! the condition as a status (`A3D_MISSING_FIELD`, `A3D_SOLVE_FAILED`, `A3D_NAN_IN_INPUT`, `A3D_CAST_OVERFLOW`).
//!
//! **Unverifiable in the build image** (no Rust toolchain): `sys.rs` is generated from the header and checked
//! symbol by symbol by `tests/test_rust_shim_cpu.py`; the rest is written against the reference's sources.
pub mod bilateral;
pub mod device;
pub mod icp;
pub mod kdtree;
pub mod range_image;
pub mod sys;
