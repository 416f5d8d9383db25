//! `RangeImage::compute_normals` (src/range_image/structure.rs:184-262) and `RangeImageBuilder::build`
//! (src/range_image/builder.rs:74-91) on the device.
use crate::{device, sys};
use align3d::{bilateral::BilateralFilter, range_image::RangeImage, RgbdFrame};
use nalgebra::Vector3;
use ndarray::Array2;

/// `image.compute_normals_hip()` = `image.compute_normals()` of the reference, bit for bit.
pub trait ComputeNormalsHip {
    fn compute_normals_hip(&mut self) -> &mut Self;
}

impl ComputeNormalsHip for RangeImage {
    fn compute_normals_hip(&mut self) -> &mut Self {
        let (h, w) = (self.height(), self.width());
        let mut normals = Array2::<Vector3<f32>>::zeros((h, w));
        device::check(
            unsafe {
                sys::a3d_compute_normals(device::Context::current(), self.points.as_ptr() as *const f32, self.mask.as_ptr(),
                                         w as u64, h as u64, normals.as_mut_ptr() as *mut f32)
            },
            "RangeImage::compute_normals",
        );
        self.normals = Some(normals);
        self
    }
}

/// The builder with the reference's surface (builder.rs:7-92); `build_device` keeps the pyramid resident,
/// which is what an odometry loop wants (the frame crosses PCIe once, as u16 depth + u8 RGB).
#[derive(Debug, Clone)]
pub struct RangeImageBuilder {
    with_normals: bool,
    with_intensity: bool,
    bilateral_filter: Option<BilateralFilter<u16>>,
    pyramid_levels: usize,
    blur_sigma: f32,
}

impl Default for RangeImageBuilder {
    fn default() -> Self {
        Self { with_normals: true, with_intensity: true, bilateral_filter: None, pyramid_levels: 3, blur_sigma: 1.0 }
    }
}

impl RangeImageBuilder {
    pub fn with_normals(mut self, value: bool) -> Self {
        self.with_normals = value;
        self
    }
    pub fn with_intensity(mut self, value: bool) -> Self {
        self.with_intensity = value;
        self
    }
    pub fn with_bilateral_filter(mut self, value: Option<BilateralFilter<u16>>) -> Self {
        self.bilateral_filter = value;
        self
    }
    pub fn pyramid_levels(mut self, levels: usize) -> Self {
        self.pyramid_levels = levels;
        self
    }
    pub fn blur_sigma(mut self, sigma: f32) -> Self {
        self.blur_sigma = sigma;
        self
    }

    fn c_params(&self) -> sys::a3d_builder_params {
        let mut p = std::mem::MaybeUninit::<sys::a3d_builder_params>::uninit();
        let mut p = unsafe {
            sys::a3d_builder_params_default(p.as_mut_ptr());
            p.assume_init()
        };
        p.with_normals = self.with_normals as u32;
        p.with_intensity = self.with_intensity as u32;
        p.pyramid_levels = self.pyramid_levels as u64;
        p.blur_sigma = self.blur_sigma;
        if let Some(f) = &self.bilateral_filter {
            p.use_bilateral = 1;
            p.sigma_space = f.sigma_space;
            p.sigma_color = f.sigma_color;
        }
        p
    }

    /// builder.rs:74-91 on the GPU for a whole slice of frames (one launch sequence per 16 frames); every level of
    /// every pyramid stays resident.  Returns `[frame][level]` handles.
    pub fn build_device(&self, frames: &[RgbdFrame]) -> Vec<Vec<device::DeviceImage>> {
        if frames.is_empty() {
            return Vec::new();
        }
        let k = &frames[0].camera;
        let (w, h) = (frames[0].image.width() as u64, frames[0].image.height() as u64);
        let depth_scale = frames[0].image.depth_scale.expect("RangeImageBuilder::build: the frames need a depth scale");
        // the C ABI receives bare pointers and ONE size, camera and depth scale for the whole slice: every frame must
        // match frame 0 and hold its arrays in standard layout, or the library would read past a buffer
        for (i, f) in frames.iter().enumerate() {
            assert!(f.image.width() as u64 == w && f.image.height() as u64 == h,
                    "InvalidParameter: frame {i} is not {w}x{h} like frame 0");
            assert!(f.image.depth.is_standard_layout() && f.image.color.is_standard_layout(),
                    "InvalidParameter: frame {i}: depth / color must be in standard layout");
            assert!(f.image.depth.len() as u64 == w * h && f.image.color.len() as u64 == w * h * 3,
                    "InvalidParameter: frame {i}: depth must hold h*w u16 and color h*w*3 u8");
            assert!(f.image.depth_scale == Some(depth_scale), "InvalidParameter: frame {i} has another depth scale");
            assert!(f.camera.fx == k.fx && f.camera.fy == k.fy && f.camera.cx == k.cx && f.camera.cy == k.cy,
                    "InvalidParameter: frame {i} has other intrinsics than frame 0");
        }
        let depth: Vec<*const u16> = frames.iter().map(|f| f.image.depth.as_ptr()).collect();
        let color: Vec<*const u8> = frames.iter().map(|f| f.image.color.as_ptr()).collect(); // [h][w][3] u8
        let mut out = vec![std::ptr::null_mut(); frames.len() * self.pyramid_levels];
        device::check(
            unsafe {
                sys::a3d_range_image_build_pyramids(device::Context::current(), &self.c_params(), frames.len() as u64,
                                                    depth.as_ptr(), color.as_ptr(), w, h, k.fx, k.fy, k.cx, k.cy,
                                                    depth_scale, out.as_mut_ptr())
            },
            "RangeImageBuilder::build",
        );
        out.chunks(self.pyramid_levels).map(|c| c.iter().map(|p| device::DeviceImage(*p)).collect()).collect()
    }
}
