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
