//! `MultiscaleAlign` (replaces the bodies of src/icp/multiscale.rs:26-67).
use crate::{device, sys};
use align3d::{error::A3dError, icp::MsIcpParams, range_image::RangeImage, transform::Transform};

/// Multiscale interface for ICP algorithms: the target pyramid is uploaded ONCE by `new` and stays resident in HBM
/// (a persistent device pyramid); `align` uploads the source pyramid and runs every level and iteration on the GPU.
pub struct MultiscaleAlign<'pyramid_lt> {
    params: MsIcpParams,
    target_pyramid: &'pyramid_lt Vec<RangeImage>, // borrowed like the reference: the host arrays outlive this object
    device_targets: Vec<device::DeviceImage>,
    handle: *mut sys::a3d_multiscale,
}

impl<'pyramid_lt> MultiscaleAlign<'pyramid_lt> {
    /// src/icp/multiscale.rs:26-40: `Err(InvalidParameter)` unless the level counts are equal.
    pub fn new(params: MsIcpParams, target_pyramid: &'pyramid_lt Vec<RangeImage>) -> Result<Self, A3dError> {
        if params.len() != target_pyramid.len() {
            return Err(A3dError::invalid_parameter(
                "The number of range images pyramid levels and ICP parameters must be equal.",
            ));
        }
        let ctx = device::Context::current();
        let device_targets = device::upload_pyramid(ctx, target_pyramid);
        let c_params: Vec<_> = params.iter().map(device::params_of).collect();
        let handles: Vec<*const sys::a3d_device_image> = device_targets.iter().map(|d| d.0 as *const _).collect();
        let mut handle = std::ptr::null_mut();
        let st = unsafe {
            sys::a3d_multiscale_new(ctx, c_params.as_ptr(), c_params.len() as u64, handles.as_ptr(),
                                    handles.len() as u64, &mut handle)
        };
        if st == sys::A3D_INVALID_PARAMETER {
            return Err(A3dError::invalid_parameter(device::last_error()));
        }
        device::check(st, "a3d_multiscale_new");
        Ok(Self { params, target_pyramid, device_targets, handle })
    }

    /// src/icp/multiscale.rs:51-67: coarsest level first, each level starts from the previous level's result; a
    /// shorter source pyramid truncates like `izip!`.  Panics where the reference panics (missing normals /
    /// intensity map / intensities: image_icp.rs:44-57; `solve().unwrap()` on `None`: image_icp.rs:152).
    pub fn align(&self, source_pyramid: &[RangeImage]) -> Transform {
        let _ = (&self.params, self.target_pyramid, &self.device_targets);
        for image in source_pyramid {
            assert!(image.points.is_standard_layout() && image.mask.is_standard_layout());
        }
        // one call uploads and aligns: the coarse levels iterate under the upload of the fine ones
        // (a3d_multiscale_align_host); nothing of the source stays resident, like the borrowed `&[RangeImage]`
        let views: Vec<sys::a3d_range_image_view> = source_pyramid.iter().map(device::view_of).collect();
        let mut pose = sys::a3d_pose::default();
        device::check(
            unsafe { sys::a3d_multiscale_align_host(self.handle, views.as_ptr(), views.len() as u64, &mut pose) },
            "MultiscaleAlign::align",
        );
        device::transform_of(&pose)
    }
}

impl MultiscaleAlign<'_> {
    /// `align` for a source pyramid that is ALREADY resident (`device::upload_pyramid`, or built on the device): no
    /// copies; what an odometry loop uses, where the source of one alignment is the target of the next.
    pub fn align_resident(&self, source_pyramid: &[device::DeviceImage]) -> Transform {
        let handles: Vec<*const sys::a3d_device_image> = source_pyramid.iter().map(|d| d.0 as *const _).collect();
        let mut pose = sys::a3d_pose::default();
        device::check(
            unsafe { sys::a3d_multiscale_align(self.handle, handles.as_ptr(), handles.len() as u64, &mut pose) },
            "MultiscaleAlign::align",
        );
        device::transform_of(&pose)
    }
}

impl Drop for MultiscaleAlign<'_> {
    fn drop(&mut self) {
        unsafe { sys::a3d_multiscale_free(self.handle) };
    }
}

/// P independent `MultiscaleAlign::new(params, target_p).align(source_p)` jobs over a list of GPUs
/// (a3d_multiscale_batch_new_multi; SURVEY §8e: contiguous blocks of pairs per device, one gather of the poses).
/// No counterpart in the reference: this is what "512 pairs over 8 GPUs" looks like from Rust.
pub fn align_pairs_multi_gpu(device_ids: &[i32], params: &MsIcpParams, pairs: &[(&Vec<RangeImage>, &Vec<RangeImage>)])
    -> Vec<Transform> {
    let mut mc = std::ptr::null_mut();
    device::check(unsafe { sys::a3d_multi_context_create(device_ids.as_ptr(), device_ids.len() as u64, &mut mc) },
                  "a3d_multi_context_create");
    let (n_pairs, n_levels, n_dev) = (pairs.len() as u64, params.len() as u64, device_ids.len() as u64);
    let mut images = Vec::new(); // keeps the uploads alive until the batch is done
    let (mut targets, mut sources) = (Vec::new(), Vec::new());
    for d in 0..n_dev {
        let (mut lo, mut hi) = (0u64, 0u64);
        device::check(unsafe { sys::a3d_multi_shard_range(n_pairs, n_dev, d, &mut lo, &mut hi) }, "a3d_multi_shard_range");
        let ctx = unsafe { sys::a3d_multi_context_device(mc, d) };
        for (t, s) in &pairs[lo as usize..hi as usize] {
            for level in 0..n_levels as usize {
                let (dt, ds) = (device::DeviceImage::upload(ctx, &t[level]), device::DeviceImage::upload(ctx, &s[level]));
                targets.push(dt.0 as *const sys::a3d_device_image);
                sources.push(ds.0 as *const sys::a3d_device_image);
                images.push(dt);
                images.push(ds);
            }
        }
    }
    let c_params: Vec<_> = params.iter().map(device::params_of).collect();
    let mut batch = std::ptr::null_mut();
    device::check(
        unsafe {
            sys::a3d_multiscale_batch_new_multi(mc, c_params.as_ptr(), n_levels, n_pairs, n_levels, targets.as_ptr(),
                                                sources.as_ptr(), &mut batch)
        },
        "a3d_multiscale_batch_new_multi",
    );
    let mut poses = vec![sys::a3d_pose::default(); pairs.len()];
    let mut status = vec![0i32; pairs.len()];
    device::check(
        unsafe {
            sys::a3d_multiscale_multi_batch_align(batch, poses.as_mut_ptr(), std::ptr::null_mut(), status.as_mut_ptr(),
                                                  std::ptr::null_mut())
        },
        "a3d_multiscale_multi_batch_align",
    );
    for (j, s) in status.iter().enumerate() {
        assert!(*s == sys::A3D_OK, "pair {j}: GaussNewton::solve() returned None"); // image_icp.rs:152 unwrap()
    }
    unsafe {
        sys::a3d_multiscale_multi_batch_free(batch);
    }
    drop(images);
    unsafe {
        sys::a3d_multi_context_destroy(mc);
    }
    poses.iter().map(device::transform_of).collect()
}
