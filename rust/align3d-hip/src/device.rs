//! Context, status handling and RAII handles over the raw bindings.
use crate::sys;
use align3d::{range_image::RangeImage, transform::Transform};
use nalgebra::{Quaternion, Vector3};
use std::cell::RefCell;
use std::ffi::CStr;

/// Text of the most recent failure on this thread.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::a3d_last_error()).to_string_lossy().into_owned() }
}

/// Statuses that are panics in the reference panic here too, with the library's text (which quotes the reference's
/// `expect` messages); everything else unexpected is a panic as well: the reference's signatures have no error path.
pub fn check(status: sys::a3d_status, what: &str) {
    if status != sys::A3D_OK {
        panic!("{what}: {}", last_error());
    }
}

/// One GPU, one HIP stream (a3d_context).  The reference's objects are plain data used from any thread; here every
/// thread gets its own context on first use (`Context::current()`), which keeps `align(&self)` re-entrant.
pub struct Context(pub *mut sys::a3d_context);

impl Context {
    pub fn new(device_index: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::a3d_context_create(device_index, &mut ctx) }, "a3d_context_create");
        Context(ctx)
    }

    /// A context whose streams have the device's highest (`priority < 0`) or lowest (`> 0`) priority: the context a
    /// frame-builder thread uses next to an aligning one (a3d_context_create_with_priority).
    pub fn with_priority(device_index: i32, priority: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::a3d_context_create_with_priority(device_index, priority, &mut ctx) }, "a3d_context_create");
        Context(ctx)
    }

    /// An aligning context and the builder context that feeds it, created back to back (a3d_context_create_pair).
    pub fn pair(device_index: i32) -> (Self, Self) {
        let (mut a, mut b) = (std::ptr::null_mut(), std::ptr::null_mut());
        check(unsafe { sys::a3d_context_create_pair(device_index, &mut a, &mut b) }, "a3d_context_create_pair");
        (Context(a), Context(b))
    }

    /// `a3d_context_set_tiling`: 0 = throughput tiling; n > 0 = every (pair, level) is cut into n blocks whatever the
    /// batch, so that a pair's pose is bit-identical alone and in any batch (no counterpart in the reference, whose own
    /// sums depend on the order rayon delivers its chunks in: src/icp/image_icp.rs:96,143-148).
    pub fn set_tiling(&self, tiles_per_pair: u32) {
        check(unsafe { sys::a3d_context_set_tiling(self.0, tiles_per_pair) }, "a3d_context_set_tiling");
    }

    /// This thread's context on device `ALIGN3D_HIP_DEVICE` (default 0).
    pub fn current() -> *mut sys::a3d_context {
        thread_local! { static CTX: RefCell<Option<Context>> = RefCell::new(None); }
        CTX.with(|c| {
            let mut c = c.borrow_mut();
            if c.is_none() {
                let dev = std::env::var("ALIGN3D_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
                *c = Some(Context::new(dev));
            }
            c.as_ref().unwrap().0
        })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::a3d_context_destroy(self.0) };
    }
}

/// `Transform` <-> `a3d_pose` (Isometry3<f32> storage: translation + quaternion i, j, k, w).
pub fn pose_of(t: &Transform) -> sys::a3d_pose {
    let tr = t.0.translation.vector;
    let q = t.0.rotation.quaternion().coords; // (i, j, k, w)
    sys::a3d_pose { t: [tr[0], tr[1], tr[2]], q: [q[0], q[1], q[2], q[3]] }
}

pub fn transform_of(p: &sys::a3d_pose) -> Transform {
    // the library returns the quaternion exactly as nalgebra would hold it; `new` re-normalises a unit quaternion,
    // which changes nothing beyond the last bit
    Transform::new(&Vector3::new(p.t[0], p.t[1], p.t[2]), &Quaternion::new(p.q[3], p.q[0], p.q[1], p.q[2]))
}

/// The borrowed view of a `RangeImage`: standard-layout arrays go across as raw pointers, no host copies.
pub fn view_of(image: &RangeImage) -> sys::a3d_range_image_view {
    let k = &image.intrinsics;
    sys::a3d_range_image_view {
        points: image.points.as_ptr() as *const f32, // Array2<Vector3<f32>>: 12-byte stride
        mask: image.mask.as_ptr(),
        normals: image.normals.as_ref().map_or(std::ptr::null(), |n| n.as_ptr() as *const f32),
        intensities: image.intensities.as_ref().map_or(std::ptr::null(), |i| i.as_ptr()),
        intensity_map: image.intensity_map.as_ref().map_or(std::ptr::null(), |m| m.as_array().as_ptr()),
        fx: k.fx,
        fy: k.fy,
        cx: k.cx,
        cy: k.cy,
        width: image.width() as u64,
        height: image.height() as u64,
    }
}

/// One `RangeImage` resident in HBM (a3d_device_image); freed on drop.
pub struct DeviceImage(pub *mut sys::a3d_device_image);

impl DeviceImage {
    pub fn upload(ctx: *mut sys::a3d_context, image: &RangeImage) -> Self {
        assert!(image.points.is_standard_layout() && image.mask.is_standard_layout());
        let view = view_of(image);
        let mut out = std::ptr::null_mut();
        check(unsafe { sys::a3d_range_image_upload(ctx, &view, &mut out) }, "a3d_range_image_upload");
        DeviceImage(out)
    }
}

/// A whole `&[RangeImage]` pyramid in ONE call (a3d_range_image_upload_pyramid): the levels share one arena from the
/// context's pool, so a steady stream of `align` calls allocates nothing on the device.
pub fn upload_pyramid(ctx: *mut sys::a3d_context, pyramid: &[RangeImage]) -> Vec<DeviceImage> {
    if pyramid.is_empty() {
        return Vec::new();
    }
    for image in pyramid {
        assert!(image.points.is_standard_layout() && image.mask.is_standard_layout());
    }
    let views: Vec<sys::a3d_range_image_view> = pyramid.iter().map(view_of).collect();
    let mut out: Vec<*mut sys::a3d_device_image> = vec![std::ptr::null_mut(); pyramid.len()];
    check(
        unsafe { sys::a3d_range_image_upload_pyramid(ctx, views.as_ptr(), views.len() as u64, out.as_mut_ptr()) },
        "a3d_range_image_upload_pyramid",
    );
    out.into_iter().map(DeviceImage).collect()
}

impl Drop for DeviceImage {
    fn drop(&mut self) {
        unsafe { sys::a3d_range_image_free(self.0) };
    }
}

/// `IcpParams` -> `a3d_icp_params`, field for field (src/icp/icp_params.rs:8-23).
pub fn params_of(p: &align3d::icp::IcpParams) -> sys::a3d_icp_params {
    sys::a3d_icp_params {
        max_iterations: p.max_iterations as u64,
        weight: p.weight,
        color_weight: p.color_weight,
        max_point_to_plane_distance: p.max_point_to_plane_distance,
        max_distance: p.max_distance,
        max_normal_angle: p.max_normal_angle,
        max_color_distance: p.max_color_distance,
    }
}
