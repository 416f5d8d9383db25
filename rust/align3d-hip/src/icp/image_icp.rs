//! `ImageIcp` (replaces the body of src/icp/image_icp.rs:43-164).
use crate::{device, sys};
use align3d::{icp::IcpParams, range_image::RangeImage, transform::Transform};

pub struct ImageIcp<'target_lt> {
    pub params: IcpParams,
    target: &'target_lt RangeImage,
    pub initial_transform: Transform,
}

impl<'target_lt> ImageIcp<'target_lt> {
    /// src/icp/image_icp.rs:26-32
    pub fn new(params: IcpParams, target: &'target_lt RangeImage) -> Self {
        Self { params, target, initial_transform: Transform::eye() }
    }

    /// src/icp/image_icp.rs:43-164: all iterations run on the device; returns `best_transform`.
    /// Panics like the reference: missing target intensity map / normals, missing source intensities (`expect`,
    /// :44-57), `solve().unwrap()` on `None` (:152).
    pub fn align(&self, source: &RangeImage) -> Transform {
        let ctx = device::Context::current();
        let (target, source) = (device::DeviceImage::upload(ctx, self.target), device::DeviceImage::upload(ctx, source));
        let (params, init) = (device::params_of(&self.params), device::pose_of(&self.initial_transform));
        let mut pose = sys::a3d_pose::default();
        device::check(
            unsafe { sys::a3d_image_icp_align(ctx, &params, target.0, source.0, &init, &mut pose) },
            "ImageIcp::align",
        );
        device::transform_of(&pose)
    }
}
