"""TEST INFRASTRUCTURE — numpy restatements (f32 operation for f32 operation) of the per-frame preparation steps of
RangeImageBuilder::build (src/range_image/builder.rs:74-91).  The product runs these steps as HIP kernels
(align3d_amd/csrc/frame.hip); this module is the independent cross-check of both the kernels and the C++ oracle
(tests/test_abi_cpu.py) and is imported by tests only."""
import math

import numpy as np

from align3d_amd.range_image import CameraIntrinsics, RangeImage

F32_MAX = np.float32(3.4028235e38)


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


def compute_intensity(im):
    """structure.rs:266-277"""
    im.intensities = rgb_to_luma_u8(im.colors).reshape(-1)
    return im


def compute_intensity_map(im):
    """structure.rs:281-297"""
    if im.intensities is None:
        compute_intensity(im)
    im.intensity_map = intensity_map_from_luma(im.intensities.reshape(im.mask.shape))
    return im


def pyr_scale_down(im, sigma):
    """RangeImage::pyr_scale_down (structure.rs:309-340)."""
    h, w = im.height() // 2, im.width() // 2
    pts, any_valid = _resize_pick(im.points, im.mask, h, w)
    normals = None
    if im.normals is not None:
        normals, _ = _resize_pick(im.normals, im.mask, h, w)  # picked with the SOURCE mask
    colors = None if im.colors is None else blur_rgb_and_halve(im.colors, sigma)
    return RangeImage(pts, any_valid.astype(np.uint8), im.intrinsics.scale(0.5), normals=normals, colors=colors)


def pyramid(im, levels, sigma):
    """RangeImage::pyramid (structure.rs:342-351)."""
    pyr = [im]
    for _ in range(levels - 1):
        pyr.append(pyr_scale_down(pyr[-1], sigma))
    return pyr
