"""Loader of the CPU oracle (oracle/liba3d_oracle.so) for the tests.  TEST INFRASTRUCTURE: the
product never imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

from align3d_amd._abi import GnStateC, IcpParamsC, PointCloudViewC, PoseC, RangeImageViewC, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liba3d_oracle.so")

_P = C.c_void_p
_lib = None


def load(path=None):
    """Loads the oracle once per process.  `path`: another build of the same sources (bench.py's cpu_baseline leg
    times oracle/_native/liba3d_oracle_native.so, built -O3 -march=native on the host it runs on)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("A3D_ORACLE_SO")  # tests/test_oracle_sanitized_cpu.py: the ASan + UBSan build
    if path is None and not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR])
    lib = C.CDLL(path or ORACLE_SO)
    f = lib.orc_gn_mean_squared_residual
    f.restype = C.c_float
    lib.orc_backproject_depth.restype = C.c_uint64
    lib.orc_project.argtypes = [C.c_double] * 4 + [_P, _P]
    lib.orc_project_grad.argtypes = [C.c_double] * 2 + [_P, _P]
    lib.orc_backproject.argtypes = [C.c_double] * 4 + [C.c_float] * 3 + [_P]
    lib.orc_gn_add_weighted.argtypes = [_P, _P, C.c_float, C.c_float]
    lib.orc_gn_weight.argtypes = [_P, C.c_float]
    lib.orc_intensity_map_fill.argtypes = [_P, C.c_uint64, C.c_uint64, _P]
    lib.orc_intensity_map_bilinear_grad.argtypes = [_P, C.c_uint64, C.c_uint64, C.c_float, C.c_float, _P]
    lib.orc_image_icp_accumulate.argtypes = [_P, _P, _P, _P, C.c_int32, _P, _P]
    lib.orc_image_icp_align.argtypes = [_P, _P, _P, _P, C.c_int32, _P, _P]
    lib.orc_set_chunk_merge_order.argtypes = [C.c_uint64]
    lib.orc_multiscale_align.argtypes = [_P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint64, C.c_int32, _P]
    lib.orc_kdtree_new.argtypes = [_P, C.c_uint64, C.POINTER(_P)]
    lib.orc_kdtree_nearest.argtypes = [_P, _P, C.c_uint64, _P, _P]
    lib.orc_kdtree_stats.argtypes = [_P, _P]
    lib.orc_kdtree_free.argtypes = [_P]
    lib.orc_pcl_icp_accumulate.argtypes = [_P, _P, _P, _P, _P, C.c_int32, _P]
    lib.orc_pcl_icp_align.argtypes = [_P, _P, _P, _P, _P, _P]
    lib.orc_compute_normals.argtypes = [_P, _P, C.c_uint64, C.c_uint64, _P]
    lib.orc_compute_normals_mt.argtypes = [_P, _P, C.c_uint64, C.c_uint64, C.c_int32, _P]
    lib.orc_bilateral_filter_u16.argtypes = [_P, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _P, _P]
    lib.orc_bilateral_grid_slice_u16.argtypes = lib.orc_bilateral_filter_u16.argtypes
    lib.orc_backproject_depth.argtypes = [_P, C.c_uint64, C.c_uint64] + [C.c_double] * 5 + [_P, _P]
    lib.orc_rgb_to_luma_u8.argtypes = [_P, C.c_uint64, _P]
    lib.orc_resize_range_points.argtypes = [_P, _P] + [C.c_uint64] * 4 + [_P, _P]
    lib.orc_resize_range_normals.argtypes = [_P, _P] + [C.c_uint64] * 4 + [_P]
    lib.orc_rgb_pyr_down.argtypes = [_P, C.c_uint64, C.c_uint64, C.c_float, _P]
    lib.orc_transform_points.argtypes = [_P, _P, C.c_uint64, _P]
    lib.orc_transform_normals.argtypes = [_P, _P, C.c_uint64, _P]
    lib.orc_transform_metrics.argtypes = [_P, _P, _P, _P]
    _lib = lib
    return lib


# ---- small pythonic wrappers --------------------------------------------------------------------

def pose(t=(0, 0, 0), q=(0, 0, 0, 1)):
    p = PoseC()
    p.t[:] = [float(x) for x in t]
    p.q[:] = [float(x) for x in q]
    return p


def pose_tuple(p):
    return np.array(p.t[:], np.float32), np.array(p.q[:], np.float32)


def exp_se3(u):
    out = PoseC()
    arr = np.asarray(u, np.float32)
    load().orc_exp_se3(C.c_void_p(ptr(arr)), C.byref(out))
    return out


def compose(a, b):
    out = PoseC()
    load().orc_compose(C.byref(a), C.byref(b), C.byref(out))
    return out


def transform_points(p, pts):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.empty_like(pts)
    load().orc_transform_points(C.byref(p), ptr(pts), pts.size // 3, ptr(out))
    return out


def transform_metrics(a, b):
    ang, tr = C.c_float(), C.c_float()
    load().orc_transform_metrics(C.byref(a), C.byref(b), C.byref(ang), C.byref(tr))
    return ang.value, tr.value


def pose_from_matrix(m):
    m = np.ascontiguousarray(m, np.float32)
    out = PoseC()
    load().orc_pose_from_matrix(C.c_void_p(ptr(m)), C.byref(out))
    return out


def pose_to_matrix(p):
    m = np.empty((4, 4), np.float32)
    load().orc_pose_to_matrix(C.byref(p), C.c_void_p(ptr(m)))
    return m


class Frame:
    """A RangeImage as plain numpy arrays + its view struct (keeps the arrays alive)."""

    def __init__(self, points, mask, fx, fy, cx, cy, normals=None, intensities=None, intensity_map=None,
                 colors=None):
        self.points = np.ascontiguousarray(points, np.float32)
        self.mask = np.ascontiguousarray(mask, np.uint8)
        self.h, self.w = self.mask.shape
        self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32)
        self.intensities = None if intensities is None else np.ascontiguousarray(intensities, np.uint8)
        self.intensity_map = None if intensity_map is None else np.ascontiguousarray(intensity_map, np.float32)
        self.colors = colors
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)

    def view(self):
        v = RangeImageViewC()
        v.points = ptr(self.points)
        v.mask = ptr(self.mask)
        v.normals = ptr(self.normals)
        v.intensities = ptr(self.intensities)
        v.intensity_map = ptr(self.intensity_map)
        v.fx, v.fy, v.cx, v.cy = self.fx, self.fy, self.cx, self.cy
        v.width, v.height = self.w, self.h
        return v


def compute_normals(points, mask, threads=1):
    lib = load()
    h, w = mask.shape
    out = np.empty((h, w, 3), np.float32)
    if threads > 1:
        lib.orc_compute_normals_mt(ptr(points), ptr(mask), w, h, threads, ptr(out))
    else:
        lib.orc_compute_normals(ptr(points), ptr(mask), w, h, ptr(out))
    return out


def intensity_map(luma):
    lib = load()
    h, w = luma.shape
    out = np.empty((h + 2, w + 2), np.float32)
    lib.orc_intensity_map_fill(ptr(np.ascontiguousarray(luma)), w, h, ptr(out))
    return out


def bilateral(depth, sigma_space=4.50000000225, sigma_color=29.9999880000072, blur=True):
    lib = load()
    depth = np.ascontiguousarray(depth, np.uint16)
    h, w = depth.shape
    out = np.empty_like(depth)
    dims = (C.c_uint64 * 3)()
    fn = lib.orc_bilateral_filter_u16 if blur else lib.orc_bilateral_grid_slice_u16
    st = fn(ptr(depth), w, h, sigma_space, sigma_color, ptr(out), dims)
    return st, out, tuple(dims)


def build_frame(depth, rgb, fx, fy, cx, cy, depth_scale, use_bilateral=False):
    """RangeImage::from_rgbd_image + compute_normals + compute_intensity + compute_intensity_map
    (the fixture recipe of src/unit_test/range_images.rs:14-27 / benches/bench_image_icp.rs:12-18)."""
    lib = load()
    depth = np.ascontiguousarray(depth, np.uint16)
    if use_bilateral:
        st, depth, _ = bilateral(depth)
        assert st == 0
    h, w = depth.shape
    pts = np.empty((h, w, 3), np.float32)
    mask = np.empty((h, w), np.uint8)
    lib.orc_backproject_depth(ptr(depth), w, h, fx, fy, cx, cy, depth_scale, ptr(pts), ptr(mask))
    normals = compute_normals(pts, mask)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    luma = np.empty((h, w), np.uint8)
    lib.orc_rgb_to_luma_u8(ptr(rgb), w * h, ptr(luma))
    imap = intensity_map(luma)
    return Frame(pts, mask, fx, fy, cx, cy, normals, luma.reshape(-1), imap, colors=rgb)


def pyr_down(frame, sigma=1.0):
    """RangeImage::pyr_scale_down (src/range_image/structure.rs:309-340) + intensity + map."""
    lib = load()
    w, h = frame.w // 2, frame.h // 2
    pts = np.empty((h, w, 3), np.float32)
    mask = np.empty((h, w), np.uint8)
    lib.orc_resize_range_points(ptr(frame.points), ptr(frame.mask), frame.w, frame.h, w, h, ptr(pts), ptr(mask))
    normals = np.empty((h, w, 3), np.float32)
    lib.orc_resize_range_normals(ptr(frame.normals), ptr(frame.mask), frame.w, frame.h, w, h, ptr(normals))
    rgb = np.empty((h, w, 3), np.uint8)
    lib.orc_rgb_pyr_down(ptr(frame.colors), frame.w, frame.h, sigma, ptr(rgb))
    luma = np.empty((h, w), np.uint8)
    lib.orc_rgb_to_luma_u8(ptr(rgb), w * h, ptr(luma))
    imap = intensity_map(luma)
    return Frame(pts, mask, frame.fx * 0.5, frame.fy * 0.5, frame.cx * 0.5, frame.cy * 0.5, normals,
                 luma.reshape(-1), imap, colors=rgb)


def build_pyramid(depth, rgb, fx, fy, cx, cy, depth_scale, levels=3, use_bilateral=True, sigma=1.0):
    """RangeImageBuilder::build (src/range_image/builder.rs:74-91)."""
    pyr = [build_frame(depth, rgb, fx, fy, cx, cy, depth_scale, use_bilateral)]
    for _ in range(levels - 1):
        pyr.append(pyr_down(pyr[-1], sigma))
    return pyr


def params(**kw):
    """IcpParams::default() with overrides."""
    p = IcpParamsC(15, 1.0, 0.1, 0.1, 0.5, np.float32(np.deg2rad(np.float32(18.0))), 0.25)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ms_default_params():
    """MsIcpParams::default()."""
    arr = (IcpParamsC * 3)()
    for i, iters in enumerate((20, 20, 30)):
        arr[i] = IcpParamsC(iters, 1.0, 1.0, 0.1, 0.5, np.float32(np.float32(np.pi) / np.float32(10.0)), 2.75)
    return arr


def image_icp_accumulate(prm, target, source, T, accum_f64=False):
    g, c = GnStateC(), GnStateC()
    tv, sv = target.view(), source.view()
    st = load().orc_image_icp_accumulate(C.byref(prm), C.byref(tv), C.byref(sv), C.byref(T), int(accum_f64),
                                         C.byref(g), C.byref(c))
    return st, g, c


def image_icp_align(prm, target, source, init=None, threads=1, want_trace=False):
    out = PoseC()
    tv, sv = target.view(), source.view()
    trace = np.zeros((int(prm.max_iterations), 8), np.float32) if want_trace else None
    st = load().orc_image_icp_align(C.byref(prm), C.byref(tv), C.byref(sv),
                                    C.byref(init) if init is not None else None, threads, C.byref(out),
                                    ptr(trace))
    return st, out, trace


def set_chunk_merge_order(seed):
    """0 = chunk order; otherwise the following passes on this thread merge their chunks in seeded permutations
    (the orders rayon's par_bridge() may deliver, image_icp.rs:96,143-148)."""
    load().orc_set_chunk_merge_order(int(seed))


def multiscale_align(prm_arr, n_params, target_pyr, source_pyr, threads=1):
    tv = (RangeImageViewC * len(target_pyr))(*[f.view() for f in target_pyr])
    sv = (RangeImageViewC * len(source_pyr))(*[f.view() for f in source_pyr])
    out = PoseC()
    st = load().orc_multiscale_align(prm_arr, n_params, tv, len(target_pyr), sv, len(source_pyr), threads,
                                     C.byref(out))
    return st, out


class KdTree:
    def __init__(self, points):
        self.points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        self.h = C.c_void_p()
        self.status = load().orc_kdtree_new(ptr(self.points), len(self.points), C.byref(self.h))

    def nearest(self, queries):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
        idx = np.empty(len(q), np.uint64)
        d = np.empty(len(q), np.float32)
        load().orc_kdtree_nearest(self.h, ptr(q), len(q), ptr(idx), ptr(d))
        return idx, d

    def stats(self):
        s = (C.c_uint64 * 3)()
        load().orc_kdtree_stats(self.h, s)
        return tuple(s)

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().orc_kdtree_free(self.h)
            self.h = C.c_void_p()


def pcl_view(points, normals):
    v = PointCloudViewC()
    v.points = ptr(points)
    v.normals = ptr(normals)
    v.len = len(points)
    return v
