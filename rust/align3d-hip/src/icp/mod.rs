//! `align3d::icp` on the GPU: same items, same signatures.
pub mod image_icp;
pub mod multiscale;
pub mod pcl_icp;
pub use align3d::icp::{IcpParams, MsIcpParams};
pub use image_icp::ImageIcp;
pub use pcl_icp::Icp;
