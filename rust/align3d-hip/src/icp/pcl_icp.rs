//! `Icp` (replaces src/icp/pcl_icp.rs:31-107): kd-tree point-to-plane ICP; the tree is built on the device by `new`.
use crate::{device, sys};
use align3d::{icp::IcpParams, pointcloud::PointCloud, transform::Transform};

pub struct Icp<'target> {
    pub params: IcpParams,
    /// Kept for API compatibility: the reference's `align` ignores it too (pcl_icp.rs:59).
    pub initial_transform: Transform,
    target: &'target PointCloud,
    handle: *mut sys::a3d_pcl_icp,
}

fn view_of(pcl: &PointCloud) -> sys::a3d_point_cloud_view {
    sys::a3d_point_cloud_view {
        points: pcl.points.as_ptr() as *const f32, // Array1<Vector3<f32>>: 12-byte stride
        normals: pcl.normals.as_ref().map_or(std::ptr::null(), |n| n.as_ptr() as *const f32),
        len: pcl.len() as u64,
    }
}

impl<'target> Icp<'target> {
    /// src/icp/pcl_icp.rs:31-38: builds the kd-tree over `target.points` (panics on a NaN coordinate like
    /// `partial_cmp().unwrap()`, kdtree.rs:43).
    pub fn new(params: IcpParams, target: &'target PointCloud) -> Self {
        let ctx = device::Context::current();
        let (c_params, view) = (device::params_of(&params), view_of(target));
        let mut handle = std::ptr::null_mut();
        device::check(unsafe { sys::a3d_pcl_icp_new(ctx, &c_params, &view, &mut handle) }, "Icp::new");
        Self { params, initial_transform: Transform::eye(), target, handle }
    }

    /// src/icp/pcl_icp.rs:49-107.  Panics like the reference when either cloud has no normals (`expect`, :50-58)
    /// or `solve()` returns `None` (:96).
    pub fn align(&self, source: &PointCloud) -> Transform {
        let _ = self.target;
        let view = view_of(source);
        let mut pose = sys::a3d_pose::default();
        device::check(unsafe { sys::a3d_pcl_icp_align(self.handle, &view, &mut pose) }, "Icp::align");
        device::transform_of(&pose)
    }
}

impl Drop for Icp<'_> {
    fn drop(&mut self) {
        unsafe { sys::a3d_pcl_icp_free(self.handle) };
    }
}
