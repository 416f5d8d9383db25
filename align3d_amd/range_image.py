"""RangeImage (src/range_image/structure.rs:20-36) as numpy arrays on the host plus a handle to its
copy in HBM.  The per-frame preparation steps that feed the hot path but are not device kernels yet
(back-projection, luma, intensity map, pyramid down-scaling: SURVEY §8f "next" rows) are restated
here in numpy, f32 operation for f32 operation; compute_normals runs on the GPU."""
import ctypes as C
import math

import numpy as np

from . import _abi

F32_MAX = np.float32(3.4028235e38)


class CameraIntrinsics:
    """src/camera.rs:7-20."""

    def __init__(self, fx, fy, cx, cy, width, height):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.width, self.height = int(width), int(height)

    def scale(self, s):
        """CameraIntrinsics::scale (src/camera.rs:119-127): width/height stay as they are."""
        return CameraIntrinsics(self.fx * s, self.fy * s, self.cx * s, self.cy * s, self.width, self.height)


def rgb_to_luma_u8(rgb):
    """rgb_to_luma_u8 (src/image/luma.rs:81-83): (r*0.3 + g*0.59 + b*0.11) as u8, f32 arithmetic."""
    r = rgb[..., 0].astype(np.float32)
    g = rgb[..., 1].astype(np.float32)
    b = rgb[..., 2].astype(np.float32)
    l = (r * np.float32(0.3) + g * np.float32(0.59)) + b * np.float32(0.11)
    return np.clip(np.trunc(l), 0, 255).astype(np.uint8)


def intensity_map_from_luma(luma):
    """IntensityMap::from_luma_image (src/intensity_map.rs:37-92) including its incomplete border."""
    h, w = luma.shape
    m = np.zeros((h + 2, w + 2), np.float32)
    m[:h, :w] = luma.astype(np.float32) / np.float32(255.0)
    m[h, : w - 1] = m[h - 1, : w - 1]
    m[h + 1, : w - 1] = m[h - 1, : w - 1]
    m[: h - 1, w] = m[: h - 1, w - 1]
    m[: h - 1, w + 1] = m[: h - 1, w - 1]
    last = np.float32(luma[h - 1, w - 1]) / np.float32(255.0)
    m[h, w] = last
    m[h + 1, w + 1] = last
    return m


def _resize_pick(values, mask, dst_h, dst_w):
    """get_neighborhood_mean_point over every 2x2 block (src/range_image/resize.rs:4-40): among the
    valid (mask == 1) entries pick the one nearest to their mean; returns (picked, any_valid)."""
    src_h, src_w = mask.shape
    hr = np.float32(src_h) / np.float32(dst_h)
    wr = np.float32(src_w) / np.float32(dst_w)
    sv = (np.arange(dst_h, dtype=np.float32) * hr).astype(np.int64)
    su = (np.arange(dst_w, dtype=np.float32) * wr).astype(np.int64)
    cand, valid = [], []
    for i in range(2):
        for j in range(2):
            rr = np.minimum(sv + i, src_h - 1)[:, None]
            cc = np.minimum(su + j, src_w - 1)[None, :]
            cand.append(values[rr, cc])
            valid.append(mask[rr, cc] == 1)
    total = np.zeros((dst_h, dst_w, 3), np.float32)
    count = np.zeros((dst_h, dst_w), np.float32)
    for p, ok in zip(cand, valid):
        total = total + np.where(ok[..., None], p, np.float32(0))
        count = count + ok.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = total / count[..., None]
    min_dist = np.full((dst_h, dst_w), F32_MAX, np.float32)
    nearest = np.zeros((dst_h, dst_w, 3), np.float32)
    for p, ok in zip(cand, valid):
        d = p - mean
        dist = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        with np.errstate(invalid="ignore"):
            better = ok & (dist < min_dist)
        min_dist = np.where(better, dist, min_dist)
        nearest = np.where(better[..., None], p, nearest)
    any_valid = count > 0
    return np.where(any_valid[..., None], nearest, np.float32(0)).astype(np.float32), any_valid


def blur_rgb_and_halve(rgb, sigma):
    """py_scale_down2 (src/range_image/structure.rs:38-47): image-0.24.7 `imageops::blur` restated from
    its published algorithm (Gaussian, support 2 sigma, f32 intermediate, weights renormalised over the
    clamped taps, round to nearest) then every second pixel.  PARITY UNPINNED: no reference test pins
    the blurred values and the crate is not vendored."""
    if sigma <= 0:
        sigma = 1.0
    sigma = np.float32(sigma)
    support = np.float32(2.0) * sigma
    h, w, _ = rgb.shape

    def taps(o, size):
        centre = np.float32(o) + np.float32(0.5)
        left = min(max(int(math.floor(centre - support)), 0), size - 1)
        right = min(max(int(math.ceil(centre + support)), left + 1), size)
        c = centre - np.float32(0.5)
        x = np.arange(left, right, dtype=np.float32) - c
        wgt = (np.float32(1.0) / (np.sqrt(np.float32(2.0) * np.float32(math.pi)) * sigma)) * np.exp(
            -(x * x) / (np.float32(2.0) * sigma * sigma)
        ).astype(np.float32)
        return left, right, (wgt / wgt.sum(dtype=np.float32)).astype(np.float32)

    src = rgb.astype(np.float32)
    tmp = np.empty_like(src)
    for oy in range(h):
        l, r, wgt = taps(oy, h)
        acc = np.zeros((w, 3), np.float32)
        for k in range(r - l):
            acc = acc + src[l + k] * wgt[k]
        tmp[oy] = acc
    dh, dw = h // 2, w // 2
    out = np.empty((dh, dw, 3), np.uint8)
    rows = tmp[0 : 2 * dh : 2]
    for dx in range(dw):
        l, r, wgt = taps(2 * dx, w)
        acc = np.zeros((dh, 3), np.float32)
        for k in range(r - l):
            acc = acc + rows[:, l + k] * wgt[k]
        out[:, dx] = np.floor(np.clip(acc, 0, 255) + np.float32(0.5)).astype(np.uint8)
    return out


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

    # -- construction -------------------------------------------------------------------------
    @staticmethod
    def from_rgbd_image(camera, depth_u16, rgb, depth_scale):
        """RangeImage::from_rgbd_image (src/range_image/structure.rs:56-95) + backproject (camera.rs:101-107)."""
        depth_u16 = np.asarray(depth_u16, np.uint16)
        h, w = depth_u16.shape
        z = depth_u16.astype(np.float32) * np.float32(depth_scale)
        xs = np.arange(w, dtype=np.float32)[None, :] - np.float32(camera.cx)
        ys = np.arange(h, dtype=np.float32)[:, None] - np.float32(camera.cy)
        x = (xs * z) / np.float32(camera.fx)
        y = (ys * z) / np.float32(camera.fy)
        valid = depth_u16 > 0
        pts = np.stack([x, y, z], -1).astype(np.float32)
        pts[~valid] = 0
        return RangeImage(pts, valid.astype(np.uint8), camera, colors=rgb)

    def width(self):
        return self.mask.shape[1]

    def height(self):
        return self.mask.shape[0]

    def len(self):
        return self.mask.size

    def valid_points_count(self):
        return int(np.count_nonzero(self.mask))

    # -- per-frame preparation ---------------------------------------------------------------
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

    def compute_intensity(self):
        """structure.rs:266-277"""
        self.intensities = rgb_to_luma_u8(self.colors).reshape(-1)
        self._device = None
        return self

    def compute_intensity_map(self):
        """structure.rs:281-297"""
        if self.intensities is None:
            self.compute_intensity()
        self.intensity_map = intensity_map_from_luma(self.intensities.reshape(self.mask.shape))
        self._device = None
        return self

    def pyr_scale_down(self, sigma):
        """RangeImage::pyr_scale_down (structure.rs:309-340)."""
        h, w = self.height() // 2, self.width() // 2
        pts, any_valid = _resize_pick(self.points, self.mask, h, w)
        normals = None
        if self.normals is not None:
            normals, _ = _resize_pick(self.normals, self.mask, h, w)  # picked with the SOURCE mask
        colors = None if self.colors is None else blur_rgb_and_halve(self.colors, sigma)
        return RangeImage(pts, any_valid.astype(np.uint8), self.intrinsics.scale(0.5), normals=normals, colors=colors)

    def pyramid(self, levels, sigma):
        """RangeImage::pyramid (structure.rs:342-351)."""
        pyr = [self]
        for _ in range(levels - 1):
            pyr.append(pyr[-1].pyr_scale_down(sigma))
        return pyr

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

    def build_device(self, camera, depth_u16, rgb, depth_scale):
        """builder.rs:74-91 entirely on the GPU (a3d_range_image_build_pyramid): the frame crosses PCIe as
        u16 depth + u8 RGB and the pyramid levels stay resident.  Returns a list of DeviceRangeImage."""
        depth_u16 = np.ascontiguousarray(depth_u16, np.uint16)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w = depth_u16.shape
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
        out = (C.c_void_p * self._pyramid_levels)()
        _abi.check(
            self.ctx.lib.a3d_range_image_build_pyramid(self.ctx.handle, C.byref(p), _abi.ptr(depth_u16), _abi.ptr(rgb), w, h,
                                                       camera.fx, camera.fy, camera.cx, camera.cy, float(depth_scale), out),
            "a3d_range_image_build_pyramid",
        )
        return [DeviceRangeImage(self.ctx, handle=C.c_void_p(out[i])) for i in range(self._pyramid_levels)]

    def build(self, camera, depth_u16, rgb, depth_scale):
        """builder.rs:74-91 with the host (numpy) restatements of the per-level steps; bilateral filter and
        normals still run on the GPU.  Kept as the cross-check of build_device."""
        if self._bilateral_filter is not None:
            depth_u16 = self._bilateral_filter.filter(self.ctx, depth_u16)
        first = RangeImage.from_rgbd_image(camera, depth_u16, rgb, depth_scale)
        if self._with_normals:
            first.compute_normals(self.ctx)
        pyr = first.pyramid(self._pyramid_levels, self._blur_sigma)
        if self._with_intensity:
            for im in pyr:
                im.compute_intensity()
                im.compute_intensity_map()
        return pyr
