"""BilateralFilter<u16> (src/bilateral/edge_aware_filter.rs:14-135)."""
import ctypes as C

import numpy as np

from . import _abi


class BilateralFilter:
    def __init__(self, sigma_space=None, sigma_color=None):
        if sigma_space is None or sigma_color is None:
            ss, sc = C.c_double(), C.c_double()
            _abi.load_library().a3d_bilateral_default_sigmas(C.byref(ss), C.byref(sc))
            sigma_space = ss.value if sigma_space is None else sigma_space
            sigma_color = sc.value if sigma_color is None else sigma_color
        self.sigma_space = float(sigma_space)
        self.sigma_color = float(sigma_color)
        self.last_grid_dims = None

    @staticmethod
    def default():
        return BilateralFilter()

    @staticmethod
    def new(sigma_space, sigma_color):
        return BilateralFilter(sigma_space, sigma_color)

    def filter(self, ctx, image):
        image = np.ascontiguousarray(image, np.uint16)
        h, w = image.shape
        out = np.empty_like(image)
        dims = (C.c_uint64 * 3)()
        _abi.check(
            ctx.lib.a3d_bilateral_filter_u16(ctx.handle, _abi.ptr(image), w, h, self.sigma_space, self.sigma_color,
                                             _abi.ptr(out), dims),
            "a3d_bilateral_filter_u16",
        )
        self.last_grid_dims = tuple(dims)
        return out

    def filter_device(self, ctx, d_images, n_images, width, height, d_out):
        """The same filter on `n_images` images that are already resident ([n][height][width] u16 at device pointer
        `d_images`, result at `d_out`: ctx.malloc / ctx.to_device): a3d_bilateral_filter_u16_device."""
        _abi.check(
            ctx.lib.a3d_bilateral_filter_u16_device(ctx.handle, d_images, n_images, width, height, self.sigma_space,
                                                    self.sigma_color, d_out),
            "a3d_bilateral_filter_u16_device",
        )
