"""R3dTree (src/kdtree.rs:19-106)."""
import ctypes as C

import numpy as np

from . import _abi


class R3dTree:
    def __init__(self, ctx, points, device_points=None, n=None):
        """R3dTree::new(&points); `device_points` (a device pointer to [n][3] f32): a3d_kdtree_new_device."""
        self.ctx = ctx
        self.handle = C.c_void_p()
        if device_points is not None:
            self.points = None
            _abi.check(ctx.lib.a3d_kdtree_new_device(ctx.handle, device_points, int(n), C.byref(self.handle)),
                       "a3d_kdtree_new_device")
            return
        self.points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        _abi.check(ctx.lib.a3d_kdtree_new(ctx.handle, _abi.ptr(self.points), len(self.points), C.byref(self.handle)),
                   "a3d_kdtree_new")

    @staticmethod
    def new(ctx, points):
        return R3dTree(ctx, points)

    @staticmethod
    def new_device(ctx, d_points, n):
        """The tree over points already resident in HBM (d_points: device pointer, [n][3] f32)."""
        return R3dTree(ctx, None, device_points=d_points, n=n)

    def build_path(self):
        """1 = the selection build, 3 = the same with the chip-wide placement launches for oversized median buckets (a
        context adds them once one of its clouds had such a bucket); diagnostics build only: 2 sorting build, 0 host build."""
        v = C.c_int32()
        _abi.check(self.ctx.lib.a3d_kdtree_build_path(self.handle, C.byref(v)))
        return v.value

    def build_ms(self):
        """Device time (ms) of the build's launches (a3d_kdtree_build_ms)."""
        v = C.c_float()
        _abi.check(self.ctx.lib.a3d_kdtree_build_ms(self.handle, C.byref(v)))
        return v.value

    def nearest(self, queries):
        """R3dTree::nearest for a batch: (indices u64, squared distances f32)."""
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 3)
        idx = np.empty(len(q), np.uint64)
        d = np.empty(len(q), np.float32)
        _abi.check(self.ctx.lib.a3d_kdtree_nearest(self.handle, _abi.ptr(q), len(q), _abi.ptr(idx), _abi.ptr(d)),
                   "a3d_kdtree_nearest")
        return idx, d

    def nearest_device(self, d_queries, m, d_idx, d_sqr):
        _abi.check(self.ctx.lib.a3d_kdtree_nearest_device(self.handle, d_queries, m, d_idx, d_sqr))

    def stats(self):
        s = (C.c_uint64 * 3)()
        _abi.check(self.ctx.lib.a3d_kdtree_stats(self.handle, s))
        return tuple(s)

    def download(self):
        """(split values in heap order, leaf slots [slots, 4] as raw u32 bits) — the tree as laid out in HBM."""
        counts = (C.c_uint64 * 2)()
        _abi.check(self.ctx.lib.a3d_kdtree_download(self.handle, None, None, counts))
        split = np.empty(counts[0], np.float32)
        leaves = np.empty((counts[1], 4), np.float32)
        _abi.check(self.ctx.lib.a3d_kdtree_download(self.handle, _abi.ptr(split), _abi.ptr(leaves), counts))
        return split.view(np.uint32), leaves.view(np.uint32)

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.a3d_kdtree_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx.handle:
                self.free()
        except Exception:
            pass
