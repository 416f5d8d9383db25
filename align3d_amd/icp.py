"""ImageIcp, MultiscaleAlign, Icp — the reference's alignment API (src/icp/*.rs) over the C ABI."""
import ctypes as C

import numpy as np

from . import _abi
from .icp_params import IcpParams, MsIcpParams
from .range_image import DeviceRangeImage, RangeImage, upload_pyramid
from .transform import Transform


def _dev(ctx, image):
    if isinstance(image, DeviceRangeImage):
        return image
    if isinstance(image, RangeImage):
        return image.device(ctx)
    raise TypeError("expected a RangeImage or DeviceRangeImage")


def _dev_pyramid(ctx, images):
    """Device copies of a pyramid; host RangeImages that are not resident yet go up in ONE call sharing one arena."""
    images = list(images)
    missing = [im for im in images if isinstance(im, RangeImage) and (im._device is None or im._device.ctx is not ctx)]
    if len(missing) > 1:
        upload_pyramid(ctx, missing)
    return [_dev(ctx, im) for im in images]


def _handle_array(images):
    arr = (C.c_void_p * max(1, len(images)))()
    for i, im in enumerate(images):
        arr[i] = im.handle
    return arr


class ImageIcp:
    """ImageIcp (src/icp/image_icp.rs:19-165)."""

    def __init__(self, ctx, params, target):
        self.ctx = ctx
        self.params = params
        self.target = _dev(ctx, target)
        self.initial_transform = Transform.eye()

    @staticmethod
    def new(ctx, params, target):
        return ImageIcp(ctx, params, target)

    def align(self, source, trace=False):
        src = _dev(self.ctx, source)
        p = self.params.to_c()
        init = self.initial_transform.to_c()
        out = _abi.PoseC()
        if trace:
            tr = np.zeros((int(self.params.max_iterations), 8), np.float32)
            _abi.check(
                self.ctx.lib.a3d_image_icp_align_trace(self.ctx.handle, C.byref(p), self.target.handle, src.handle,
                                                       C.byref(init), C.byref(out), _abi.ptr(tr)),
                "ImageIcp::align",
            )
            return Transform.from_c(out), tr
        _abi.check(
            self.ctx.lib.a3d_image_icp_align(self.ctx.handle, C.byref(p), self.target.handle, src.handle,
                                             C.byref(init), C.byref(out)),
            "ImageIcp::align",
        )
        return Transform.from_c(out)

    def accumulate(self, source, transform):
        """One pass of the pixel loop from `transform`: (geom, colour) accumulators (test hook)."""
        src = _dev(self.ctx, source)
        p = self.params.to_c()
        t = transform.to_c()
        g, c = _abi.GnStateC(), _abi.GnStateC()
        _abi.check(
            self.ctx.lib.a3d_image_icp_accumulate(self.ctx.handle, C.byref(p), self.target.handle, src.handle,
                                                  C.byref(t), C.byref(g), C.byref(c)),
            "ImageIcp accumulate",
        )
        return g.as_dict(), c.as_dict()

    def accumulate_exact(self, source, transform):
        """The same pass through the cross-check kernel whose per-pixel arithmetic is the reference's operation for
        operation (a3d_image_icp_accumulate_exact, test hook)."""
        src = _dev(self.ctx, source)
        p, t = self.params.to_c(), transform.to_c()
        g, c = _abi.GnStateC(), _abi.GnStateC()
        _abi.check(
            self.ctx.lib.a3d_image_icp_accumulate_exact(self.ctx.handle, C.byref(p), self.target.handle, src.handle,
                                                        C.byref(t), C.byref(g), C.byref(c)),
            "ImageIcp accumulate_exact",
        )
        return g.as_dict(), c.as_dict()

    def accumulate_weighted(self, source, transform):
        """One pass of the opt-in merged accumulation (A3D_ICP_ACCUM=merged: a thread sums geom.add_weighted(color, w,
        cw) directly): the merged accumulator H, g, weighted residual sum, combined count (test hook)."""
        src = _dev(self.ctx, source)
        p, t, g = self.params.to_c(), transform.to_c(), _abi.GnStateC()
        _abi.check(
            self.ctx.lib.a3d_image_icp_accumulate_weighted(self.ctx.handle, C.byref(p), self.target.handle, src.handle,
                                                           C.byref(t), C.byref(g)),
            "ImageIcp accumulate_weighted",
        )
        return g.as_dict()


class MultiscaleAlign:
    """MultiscaleAlign (src/icp/multiscale.rs:7-68)."""

    def __init__(self, ctx, params, target_pyramid):
        self.ctx = ctx
        self.params = params
        self.targets = _dev_pyramid(ctx, target_pyramid)
        self.handle = C.c_void_p()
        parr = params.to_c_array()
        tarr = _handle_array(self.targets)
        st = ctx.lib.a3d_multiscale_new(ctx.handle, parr, len(params), tarr, len(self.targets), C.byref(self.handle))
        if st == _abi.A3D_INVALID_PARAMETER:
            # Err(A3dError::InvalidParameter(..)) (multiscale.rs:30-34)
            raise _abi.InvalidParameter(ctx.lib.a3d_last_error().decode())
        _abi.check(st, "MultiscaleAlign::new")

    @staticmethod
    def new(ctx, params, target_pyramid):
        return MultiscaleAlign(ctx, params, target_pyramid)

    def align(self, source_pyramid):
        source_pyramid = list(source_pyramid)
        out = _abi.PoseC()
        if source_pyramid and all(isinstance(im, RangeImage) and im._device is None for im in source_pyramid):
            # the reference's literal call: `&[RangeImage]` in host memory.  Uploaded and aligned in ONE call, the coarse
            # levels iterating under the upload of the fine ones (a3d_multiscale_align_host); nothing stays resident
            views = (_abi.RangeImageViewC * len(source_pyramid))(*[im.view() for im in source_pyramid])
            _abi.check(self.ctx.lib.a3d_multiscale_align_host(self.handle, views, len(source_pyramid), C.byref(out)),
                       "MultiscaleAlign::align")
            return Transform.from_c(out)
        srcs = _dev_pyramid(self.ctx, source_pyramid)
        _abi.check(self.ctx.lib.a3d_multiscale_align(self.handle, _handle_array(srcs), len(srcs), C.byref(out)),
                   "MultiscaleAlign::align")
        return Transform.from_c(out)

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.a3d_multiscale_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MultiscaleAlignBatch:
    """P independent MultiscaleAlign::new(params, target_p).align(source_p) in one launch sequence."""

    def __init__(self, ctx, params, target_pyramids, source_pyramids):
        assert len(target_pyramids) == len(source_pyramids) and len(target_pyramids) > 0
        self.ctx = ctx
        self.n_pairs = len(target_pyramids)
        self.n_levels = len(target_pyramids[0])
        self._keep = []
        t_flat, s_flat = [], []
        for tp, sp in zip(target_pyramids, source_pyramids):
            assert len(tp) == self.n_levels and len(sp) == self.n_levels
            t_flat += [_dev(ctx, t) for t in tp]
            s_flat += [_dev(ctx, s) for s in sp]
        self._keep = (t_flat, s_flat)
        self.handle = C.c_void_p()
        st = ctx.lib.a3d_multiscale_batch_new(ctx.handle, params.to_c_array(), len(params), self.n_pairs,
                                              self.n_levels, _handle_array(t_flat), _handle_array(s_flat),
                                              C.byref(self.handle))
        if st == _abi.A3D_INVALID_PARAMETER:
            raise _abi.InvalidParameter(ctx.lib.a3d_last_error().decode())
        _abi.check(st, "a3d_multiscale_batch_new")

    def rebind(self, target_pyramids, source_pyramids):
        """The same batch object on other pyramids (same pair / level counts): nothing is allocated or freed."""
        assert len(target_pyramids) == self.n_pairs and len(source_pyramids) == self.n_pairs
        t_flat, s_flat = [], []
        for tp, sp in zip(target_pyramids, source_pyramids):
            assert len(tp) == self.n_levels and len(sp) == self.n_levels
            t_flat += [_dev(self.ctx, t) for t in tp]
            s_flat += [_dev(self.ctx, s) for s in sp]
        _abi.check(self.ctx.lib.a3d_multiscale_batch_rebind(self.handle, _handle_array(t_flat), _handle_array(s_flat)),
                   "a3d_multiscale_batch_rebind")
        self._keep = (t_flat, s_flat)
        return self

    def align(self, matrices_device=None):
        """Runs all pairs; returns (list of Transform, int32 status array)."""
        poses = (_abi.PoseC * self.n_pairs)()
        status = np.zeros(self.n_pairs, np.int32)
        _abi.check(
            self.ctx.lib.a3d_multiscale_batch_align(self.handle, poses, matrices_device,
                                                    status.ctypes.data_as(C.POINTER(C.c_int32))),
            "a3d_multiscale_batch_align",
        )
        return [Transform.from_c(p) for p in poses], status

    def enqueue(self, matrices_device=None):
        """Enqueues one pass without synchronising the host."""
        _abi.check(self.ctx.lib.a3d_multiscale_batch_align(self.handle, None, matrices_device, None))

    def results(self):
        """(list of Transform, int32 status array) of the most recent pass: waits for that pass only, not for what
        was enqueued on the context since (another batch's pass)."""
        poses = (_abi.PoseC * self.n_pairs)()
        status = np.zeros(self.n_pairs, np.int32)
        _abi.check(self.ctx.lib.a3d_multiscale_batch_results(self.handle, poses,
                                                             status.ctypes.data_as(C.POINTER(C.c_int32))),
                   "a3d_multiscale_batch_results")
        return [Transform.from_c(p) for p in poses], status

    def set_profiling(self, on):
        _abi.check(self.ctx.lib.a3d_multiscale_batch_set_profiling(self.handle, 1 if on else 0))

    def last_timing(self):
        ms, n = C.c_float(), C.c_uint64()
        _abi.check(self.ctx.lib.a3d_multiscale_batch_last_timing(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def persistent_levels(self):
        """Bit mask of the levels the most recent align ran inside the persistent kernel."""
        m = C.c_uint32()
        _abi.check(self.ctx.lib.a3d_multiscale_batch_persistent_levels(self.handle, C.byref(m)))
        return int(m.value)

    def concurrency(self):
        """Number of pair groups whose launches run on separate streams at the same time."""
        n = C.c_uint32()
        _abi.check(self.ctx.lib.a3d_multiscale_batch_concurrency(self.handle, C.byref(n)))
        return n.value

    def last_kernel_ms(self):
        ms = C.c_float()
        _abi.check(self.ctx.lib.a3d_multiscale_batch_last_kernel_ms(self.handle, C.byref(ms)))
        return ms.value

    def last_level_ms(self, level):
        """(sum of the per-pixel kernel's launch durations at `level` in the last pass, number of launches); profiling on."""
        ms, n = C.c_float(), C.c_uint32()
        _abi.check(self.ctx.lib.a3d_multiscale_batch_last_level_ms(self.handle, int(level), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def free(self):
        if self.handle and self.ctx.handle:  # a handle must not outlive its context
            self.ctx.lib.a3d_multiscale_batch_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PointCloud:
    """PointCloud (src/pointcloud.rs:8-12): points [N,3], optional normals."""

    def __init__(self, points, normals=None):
        self.points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)

    @staticmethod
    def from_range_image(im):
        """From<&RangeImage> for PointCloud (src/range_image/structure.rs:375-406): mask != 0, row-major."""
        m = im.mask.reshape(-1) != 0
        normals = None if im.normals is None else im.normals.reshape(-1, 3)[m]
        return PointCloud(im.points.reshape(-1, 3)[m], normals)

    def len(self):
        return len(self.points)

    def view(self):
        v = _abi.PointCloudViewC()
        v.points = _abi.ptr(self.points)
        v.normals = _abi.ptr(self.normals)
        v.len = len(self.points)
        return v


class DevicePointCloud:
    """A PointCloud (src/pointcloud.rs:8-12) resident in HBM: points / normals [len][3] f32 on the context's GPU, for
    the device-pointer forms a3d_pcl_icp_new_device / a3d_pcl_icp_align_device (no PCIe traffic per call)."""

    def __init__(self, ctx, cloud):
        self.ctx = ctx
        self.n = cloud.len()
        self.d_points = ctx.to_device(cloud.points)
        self.d_normals = None if cloud.normals is None else ctx.to_device(cloud.normals)

    def len(self):
        return self.n

    def view(self):
        v = _abi.PointCloudViewC()
        v.points = self.d_points
        v.normals = self.d_normals
        v.len = self.n
        return v

    def free(self):
        for p in (self.d_points, self.d_normals):
            if p is not None and self.ctx.handle:
                self.ctx.free(p)
        self.d_points = self.d_normals = None


class Icp:
    """Icp (src/icp/pcl_icp.rs:15-108): point-to-plane ICP with kd-tree correspondences.  `target` / `source` may be
    PointCloud (host arrays, as in the reference) or DevicePointCloud (already resident)."""

    def __init__(self, ctx, params, target):
        self.ctx = ctx
        self.params = params
        self.target = target
        self.initial_transform = Transform.eye()  # public field the reference ignores (pcl_icp.rs:59)
        self.handle = C.c_void_p()
        p = params.to_c()
        v = target.view()
        fn = ctx.lib.a3d_pcl_icp_new_device if isinstance(target, DevicePointCloud) else ctx.lib.a3d_pcl_icp_new
        _abi.check(fn(ctx.handle, C.byref(p), C.byref(v), C.byref(self.handle)), "Icp::new")

    @staticmethod
    def new(ctx, params, target):
        return Icp(ctx, params, target)

    def align(self, source):
        v = source.view()
        out = _abi.PoseC()
        fn = self.ctx.lib.a3d_pcl_icp_align_device if isinstance(source, DevicePointCloud) else self.ctx.lib.a3d_pcl_icp_align
        _abi.check(fn(self.handle, C.byref(v), C.byref(out)), "Icp::align")
        return Transform.from_c(out)

    def last_device_ms(self):
        ms = C.c_float()
        _abi.check(self.ctx.lib.a3d_pcl_icp_last_device_ms(self.handle, C.byref(ms)))
        return ms.value

    def accumulate(self, source, transform):
        v = source.view()
        t = transform.to_c()
        g = _abi.GnStateC()
        _abi.check(self.ctx.lib.a3d_pcl_icp_accumulate(self.handle, C.byref(v), C.byref(t), C.byref(g)))
        return g.as_dict()

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.a3d_pcl_icp_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
