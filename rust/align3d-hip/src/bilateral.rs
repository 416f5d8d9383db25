//! `BilateralFilter::<u16>::filter` (src/bilateral/edge_aware_filter.rs:126-135) on the device: bit-identical u16.
use crate::{device, sys};
use ndarray::Array2;

/// `filter_hip(&filter, &image)` = `filter.filter(&image)` of the reference for `I = u16`.
/// Panics like `num::cast().unwrap()` (grid.rs:129) when a sliced value is not representable as u16.
pub fn filter_hip(filter: &align3d::bilateral::BilateralFilter<u16>, image: &Array2<u16>) -> Array2<u16> {
    let (h, w) = image.dim();
    let input = image.as_standard_layout();
    let mut out = Array2::<u16>::zeros((h, w));
    device::check(
        unsafe {
            sys::a3d_bilateral_filter_u16(device::Context::current(), input.as_ptr(), w as u64, h as u64, filter.sigma_space,
                                          filter.sigma_color, out.as_mut_ptr(), std::ptr::null_mut())
        },
        "BilateralFilter::filter",
    );
    out
}

/// The same filter on `n_images` images that are already resident in device memory (`[n][height][width]` u16 at
/// `d_images`, the result at `d_out`): `a3d_bilateral_filter_u16_device` — the shape of benches/bench_bilateral.rs without
/// PCIe in it.
///
/// # Safety
/// `d_images` and `d_out` must be device pointers of the current context with room for `n_images * width * height` u16 each.
pub unsafe fn filter_hip_device(filter: &align3d::bilateral::BilateralFilter<u16>, d_images: *const u16, n_images: usize,
                                width: usize, height: usize, d_out: *mut u16) {
    device::check(
        sys::a3d_bilateral_filter_u16_device(device::Context::current(), d_images, n_images as u64, width as u64, height as u64,
                                             filter.sigma_space, filter.sigma_color, d_out),
        "BilateralFilter::filter (device)",
    );
}
