//! `R3dTree` (replaces src/kdtree.rs:28-105): same leaf-only, no-backtracking search, bit-identical neighbours.
use crate::{device, sys};
use nalgebra::Vector3;
use ndarray::prelude::*;

pub struct R3dTree {
    handle: *mut sys::a3d_kdtree,
}

impl R3dTree {
    /// src/kdtree.rs:28-58 (stable sort per level, leaf <= 16, mid = len / 2) as a device build.
    /// Panics on a NaN coordinate like `partial_cmp().unwrap()` (:43).
    pub fn new(points: &ArrayView1<Vector3<f32>>) -> Self {
        let owned; // a strided view is gathered once; a standard-layout one goes across as it is
        let ptr = if points.is_standard_layout() {
            points.as_ptr()
        } else {
            owned = points.to_owned();
            owned.as_ptr()
        };
        let mut handle = std::ptr::null_mut();
        device::check(
            unsafe { sys::a3d_kdtree_new(device::Context::current(), ptr as *const f32, points.len() as u64, &mut handle) },
            "R3dTree::new",
        );
        Self { handle }
    }

    /// The same tree over points that are already resident in HBM (`d_points`: device pointer to `[n][3]` f32 on this
    /// thread's context's GPU; read during the call only): `a3d_kdtree_new_device`.
    ///
    /// # Safety
    /// `d_points` must be a valid device allocation of at least `12 * n` bytes.
    pub unsafe fn new_device(d_points: *const std::ffi::c_void, n: usize) -> Self {
        let mut handle = std::ptr::null_mut();
        device::check(sys::a3d_kdtree_new_device(device::Context::current(), d_points, n as u64, &mut handle), "R3dTree::new_device");
        Self { handle }
    }

    /// src/kdtree.rs:69-105: (index of the nearest neighbour in its leaf, squared distance).
    pub fn nearest(&self, point: &Vector3<f32>) -> (usize, f32) {
        let (idx, dist) = self.nearest_batch(std::slice::from_ref(point));
        (idx[0], dist[0])
    }

    /// The batched form the GPU is built for (one launch for all queries); `nearest` is this with one query.
    pub fn nearest_batch(&self, points: &[Vector3<f32>]) -> (Vec<usize>, Vec<f32>) {
        let mut idx = vec![0u64; points.len()];
        let mut dist = vec![0f32; points.len()];
        device::check(
            unsafe {
                sys::a3d_kdtree_nearest(self.handle, points.as_ptr() as *const f32, points.len() as u64, idx.as_mut_ptr(),
                                        dist.as_mut_ptr())
            },
            "R3dTree::nearest",
        );
        (idx.into_iter().map(|i| i as usize).collect(), dist)
    }
}

impl Drop for R3dTree {
    fn drop(&mut self) {
        unsafe { sys::a3d_kdtree_free(self.handle) };
    }
}
