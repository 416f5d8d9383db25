"""RangeImage (src/range_image/structure.rs:20-36) as numpy arrays on the host plus a handle to its copy in HBM,
and RangeImageBuilder (src/range_image/builder.rs), whose `build` runs entirely on the GPU
(a3d_range_image_build_pyramid: bilateral filter, back-projection, normals, pyramid, luma, intensity maps)."""
import ctypes as C

import numpy as np

from . import _abi

class CameraIntrinsics:
    """src/camera.rs:7-20."""

    def __init__(self, fx, fy, cx, cy, width, height):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.width, self.height = int(width), int(height)

    def scale(self, s):
        """CameraIntrinsics::scale (src/camera.rs:119-127): width/height stay as they are."""
        return CameraIntrinsics(self.fx * s, self.fy * s, self.cx * s, self.cy * s, self.width, self.height)


class RangeImage:
    """Host arrays in the reference's standard layout; `device(ctx)` gives the HBM-resident copy."""

    def __init__(self, points, mask, intrinsics, normals=None, colors=None, intensities=None, intensity_map=None):
        self.points = np.ascontiguousarray(points, np.float32)
        self.mask = np.ascontiguousarray(mask, np.uint8)
        assert self.points.shape == self.mask.shape + (3,)
        self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32)
        self.colors = None if colors is None else np.ascontiguousarray(colors, np.uint8)
        self.intensities = None if intensities is None else np.ascontiguousarray(intensities, np.uint8).reshape(-1)
        self.intensity_map = None if intensity_map is None else np.ascontiguousarray(intensity_map, np.float32)
        self.intrinsics = intrinsics
        self._device = None

    def width(self):
        return self.mask.shape[1]

    def height(self):
        return self.mask.shape[0]

    def len(self):
        return self.mask.size

    def valid_points_count(self):
        return int(np.count_nonzero(self.mask))

    def compute_normals(self, ctx):
        """RangeImage::compute_normals (structure.rs:184-262): HIP stencil kernel."""
        out = np.empty(self.points.shape, np.float32)
        _abi.check(
            ctx.lib.a3d_compute_normals(ctx.handle, _abi.ptr(self.points), _abi.ptr(self.mask), self.width(),
                                        self.height(), _abi.ptr(out)),
            "a3d_compute_normals",
        )
        self.normals = out
        self._device = None
        return self

    # -- device side ----------------------------------------------------------------------------
    def view(self):
        v = _abi.RangeImageViewC()
        v.points = _abi.ptr(self.points)
        v.mask = _abi.ptr(self.mask)
        v.normals = _abi.ptr(self.normals)
        v.intensities = _abi.ptr(self.intensities)
        v.intensity_map = _abi.ptr(self.intensity_map)
        k = self.intrinsics
        v.fx, v.fy, v.cx, v.cy = k.fx, k.fy, k.cx, k.cy
        v.width, v.height = self.width(), self.height()
        return v

    def device(self, ctx):
        if self._device is None or self._device.ctx is not ctx:
            self._device = DeviceRangeImage(ctx, self)
        return self._device


class DeviceRangeImage:
    """a3d_device_image: one RangeImage resident in HBM."""

    def __init__(self, ctx, host=None, handle=None):
        self.ctx = ctx
        if handle is not None:
            self.handle = handle
            w, h = C.c_uint64(), C.c_uint64()
            _abi.check(ctx.lib.a3d_range_image_size(self.handle, C.byref(w), C.byref(h)))
            self.shape = (int(h.value), int(w.value))
        else:
            self.handle = C.c_void_p()
            v = host.view()
            _abi.check(ctx.lib.a3d_range_image_upload(ctx.handle, C.byref(v), C.byref(self.handle)),
                       "a3d_range_image_upload")
            self.shape = host.mask.shape

    def compute_normals(self):
        _abi.check(self.ctx.lib.a3d_range_image_compute_normals(self.handle))
        return self

    def download_normals(self):
        out = np.empty(self.shape + (3,), np.float32)
        _abi.check(self.ctx.lib.a3d_range_image_download_normals(self.handle, _abi.ptr(out)))
        return out

    def download(self, normals=True, intensity=True, colors=True):
        """Reads the resident arrays back into a host RangeImage (whose device copy is this image)."""
        h, w = self.shape
        pts = np.empty((h, w, 3), np.float32)
        mask = np.empty((h, w), np.uint8)
        nrm = np.empty((h, w, 3), np.float32) if normals else None
        inten = np.empty(h * w, np.uint8) if intensity else None
        imap = np.empty((h + 2, w + 2), np.float32) if intensity else None
        col = np.empty((h, w, 3), np.uint8) if colors else None
        k = (C.c_double * 4)()
        _abi.check(self.ctx.lib.a3d_range_image_download(self.handle, _abi.ptr(pts), _abi.ptr(mask), _abi.ptr(nrm),
                                                         _abi.ptr(inten), _abi.ptr(imap), _abi.ptr(col), k),
                   "a3d_range_image_download")
        ri = RangeImage(pts, mask, CameraIntrinsics(k[0], k[1], k[2], k[3], w, h), normals=nrm, colors=col,
                        intensities=inten, intensity_map=imap)
        ri._device = self
        return ri

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.a3d_range_image_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx.handle:
                self.free()
        except Exception:
            pass


def compute_normals_batch(images):
    """RangeImage::compute_normals on a list of resident images of one size (a3d_range_image_compute_normals_batch: one
    launch per 64 images, enqueue-only)."""
    if not images:
        return
    arr = (C.c_void_p * len(images))(*[im.handle for im in images])
    _abi.check(images[0].ctx.lib.a3d_range_image_compute_normals_batch(arr, len(images)),
               "a3d_range_image_compute_normals_batch")


def upload_pyramid(ctx, host_images):
    """A `&[RangeImage]` as the host holds it -> resident images sharing one pooled arena
    (a3d_range_image_upload_pyramid): what MultiscaleAlign::new / align do with host pyramids.  The device copies are
    remembered by the host objects (`RangeImage.device`), so a pyramid is uploaded once."""
    n = len(host_images)
    views = (_abi.RangeImageViewC * max(1, n))(*[im.view() for im in host_images])
    out = (C.c_void_p * max(1, n))()
    _abi.check(ctx.lib.a3d_range_image_upload_pyramid(ctx.handle, views, n, out), "a3d_range_image_upload_pyramid")
    devs = []
    for im, h in zip(host_images, out):
        d = DeviceRangeImage(ctx, handle=C.c_void_p(h))
        im._device = d
        devs.append(d)
    return devs


class RangeImageBuilder:
    """RangeImageBuilder (src/range_image/builder.rs:7-92)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._with_normals = True
        self._with_intensity = True
        self._bilateral_filter = None
        self._pyramid_levels = 3
        self._blur_sigma = 1.0

    def with_bilateral_filter(self, f):
        self._bilateral_filter = f
        return self

    def with_normals(self, v):
        self._with_normals = v
        return self

    def with_intensity(self, v):
        self._with_intensity = v
        return self

    def pyramid_levels(self, n):
        self._pyramid_levels = n
        return self

    def blur_sigma(self, s):
        self._blur_sigma = s
        return self

    def on_context(self, ctx):
        """The same builder settings bound to another context (its own stream and scratch)."""
        b = RangeImageBuilder(ctx)
        b._with_normals, b._with_intensity = self._with_normals, self._with_intensity
        b._bilateral_filter, b._pyramid_levels, b._blur_sigma = self._bilateral_filter, self._pyramid_levels, self._blur_sigma
        return b

    def _params(self):
        p = _abi.BuilderParamsC()
        self.ctx.lib.a3d_builder_params_default(C.byref(p))
        p.with_normals = 1 if self._with_normals else 0
        p.with_intensity = 1 if self._with_intensity else 0
        p.pyramid_levels = self._pyramid_levels
        p.blur_sigma = self._blur_sigma
        if self._bilateral_filter is not None:
            p.use_bilateral = 1
            p.sigma_space = self._bilateral_filter.sigma_space
            p.sigma_color = self._bilateral_filter.sigma_color
        return p

    def build_many(self, camera, frames, depth_scale):
        """RangeImageBuilder::build (builder.rs:74-91) for a list of (depth_u16, rgb) frames of one stream (same size
        and camera) in one launch sequence on the GPU (a3d_range_image_build_pyramids): each frame crosses PCIe as
        u16 depth + u8 RGB and its pyramid levels stay resident.  Returns one list of DeviceRangeImage per frame,
        index 0 = full resolution (`.download()` gives the host arrays)."""
        if not frames:
            return []
        held = []  # keeps converted copies alive until the call returns
        h, w = np.asarray(frames[0][0]).shape[:2]
        for depth_u16, rgb in frames:
            depth_u16 = np.ascontiguousarray(depth_u16, np.uint16)
            rgb = np.ascontiguousarray(rgb, np.uint8)
            if depth_u16.shape != (h, w) or rgb.shape != (h, w, 3):
                # the C ABI receives bare pointers: this is the only place a size mismatch can be caught
                raise _abi.InvalidParameter(
                    f"depth must be [h][w] u16 and rgb [h][w][3] u8, every frame {h}x{w} (got {depth_u16.shape} and {rgb.shape})")
            held.append((depth_u16, rgb))
        n, L = len(held), self._pyramid_levels
        p = self._params()
        dptr = (C.c_void_p * n)(*[_abi.ptr(d) for d, _ in held])
        cptr = (C.c_void_p * n)(*[_abi.ptr(c) for _, c in held])
        out = (C.c_void_p * (n * L))()
        _abi.check(
            self.ctx.lib.a3d_range_image_build_pyramids(self.ctx.handle, C.byref(p), n, dptr, cptr, w, h, camera.fx, camera.fy,
                                                        camera.cx, camera.cy, float(depth_scale), out),
            "a3d_range_image_build_pyramids",
        )
        return [[DeviceRangeImage(self.ctx, handle=C.c_void_p(out[f * L + l])) for l in range(L)] for f in range(n)]

    def build(self, camera, depth_u16, rgb, depth_scale):
        """RangeImageBuilder::build (builder.rs:74-91) entirely on the GPU: see build_many."""
        if np.asarray(depth_u16).ndim != 2:
            raise _abi.InvalidParameter(f"depth must be [h][w] u16 (got shape {np.asarray(depth_u16).shape})")
        return self.build_many(camera, [(depth_u16, rgb)], depth_scale)[0]

    build_device = build  # round-1 name
