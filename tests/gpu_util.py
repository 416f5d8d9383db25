"""Helpers shared by the GPU parity tests: identical inputs for the oracle and the HIP path."""
import ctypes as C

import numpy as np

import oracle_lib as O
from align3d_amd import CameraIntrinsics, IcpParams, RangeImage, Transform
from data_util import SlamTbSample

_cache = {}


def oracle_frame(sample, frame_id, use_bilateral=False):
    key = ("frame", sample, frame_id, use_bilateral)
    if key not in _cache:
        s = SlamTbSample(sample)
        _cache[key] = O.build_frame(*s.load(frame_id), *s.intrinsics(frame_id), s.depth_scale(frame_id),
                                    use_bilateral=use_bilateral)
    return _cache[key]


def oracle_pyramid(sample, frame_id, levels=3, use_bilateral=True):
    key = ("pyr", sample, frame_id, levels, use_bilateral)
    if key not in _cache:
        s = SlamTbSample(sample)
        _cache[key] = O.build_pyramid(*s.load(frame_id), *s.intrinsics(frame_id), s.depth_scale(frame_id),
                                      levels=levels, use_bilateral=use_bilateral)
    return _cache[key]


def to_range_image(fr):
    """The same arrays the oracle sees, wrapped as the product's RangeImage."""
    k = CameraIntrinsics(fr.fx, fr.fy, fr.cx, fr.cy, fr.w, fr.h)
    return RangeImage(fr.points, fr.mask, k, normals=fr.normals, colors=fr.colors, intensities=fr.intensities,
                      intensity_map=fr.intensity_map)


def params_c(p: IcpParams):
    return p.to_c()


def pose_c(t: Transform):
    return t.to_c()


def transform_diff(a: Transform, b):
    """(rotation angle, translation norm) of a^-1 * b, b a PoseC or Transform (TransformMetrics::new)."""
    pb = b.to_c() if isinstance(b, Transform) else b
    return O.transform_metrics(a.to_c(), pb)


def gn_rel_err(gpu, ref):
    """max |gpu - ref| / max|ref| over H, over g, and relative ssq error."""
    def rel(x, y):
        scale = float(np.max(np.abs(y))) or 1.0
        return float(np.max(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)))) / scale
    return rel(gpu["H"], ref["H"]), rel(gpu["g"], ref["g"]), rel([gpu["ssq"]], [ref["ssq"]])


def small_pose(seed=0, rot=0.004, trans=0.003):
    rng = np.random.default_rng(seed)
    u = np.concatenate([rng.normal(size=3) * trans, rng.normal(size=3) * rot]).astype(np.float32)
    return Transform.from_c(O.exp_se3(u))
