"""P independent MultiscaleAlign jobs over a device list from ONE host process (a3d_multi_context,
a3d_multiscale_batch_new_multi): contiguous blocks of pairs per device, one gather of the 4x4 poses onto
the first device over xGMI.  bench.py's N > 1 layout is the other form (one process per GPU + RCCL all-gather); this
one is what a Rust / C host that is not launched per GPU binds."""
import ctypes as C

import numpy as np

from . import _abi
from .context import Context
from .transform import Transform


def shard_range(n_items, n_devices, device):
    """a3d_multi_shard_range: the block [begin, end) of n_items owned by `device` (same rule as distributed.shard_range)."""
    lo, hi = C.c_uint64(), C.c_uint64()
    _abi.check(_abi.load_library().a3d_multi_shard_range(n_items, n_devices, device, C.byref(lo), C.byref(hi)),
               "a3d_multi_shard_range")
    return int(lo.value), int(hi.value)


def device_count():
    """a3d_device_count: HIP devices visible to the process (0 without a GPU)."""
    n = C.c_int32()
    _abi.check(_abi.load_library().a3d_device_count(C.byref(n)), "a3d_device_count")
    return int(n.value)


class MultiContext:
    """One Context per entry of `device_ids` (an id may repeat: several contexts on one GPU)."""

    def __init__(self, device_ids):
        self.lib = _abi.load_library()
        self.device_ids = [int(d) for d in device_ids]
        ids = (C.c_int32 * len(self.device_ids))(*self.device_ids)
        self.handle = C.c_void_p()
        _abi.check(self.lib.a3d_multi_context_create(ids, len(self.device_ids), C.byref(self.handle)),
                   "a3d_multi_context_create")
        self._views = {}

    def __len__(self):
        return int(self.lib.a3d_multi_context_size(self.handle))

    def device(self, index):
        """The Context of entry `index` (owned by this object): build the frames of the pairs it owns on it."""
        if index not in self._views:
            h = self.lib.a3d_multi_context_device(self.handle, index)
            if not h:
                raise IndexError(index)
            c = Context.__new__(Context)
            c.lib, c.handle, c.device_index, c._sibling = self.lib, C.c_void_p(h), self.device_ids[index], None
            c.close = lambda: None  # borrowed: destroyed with the MultiContext
            self._views[index] = c
        return self._views[index]

    def close(self):
        if self.handle:
            for v in self._views.values():
                if v._sibling is not None:  # a builder context created through a borrowed view is owned by nobody else
                    v._sibling.close()
                    v._sibling = None
                v.handle = C.c_void_p()
            self.lib.a3d_multi_context_destroy(self.handle)
            self.handle = C.c_void_p()


class MultiscaleAlignMultiBatch:
    """target_pyramids / source_pyramids: [n_pairs] lists of DeviceRangeImage levels in global pair order; pair j's
    images must be resident on entry shard owner of j."""

    def __init__(self, mctx, params, target_pyramids, source_pyramids):
        assert len(target_pyramids) == len(source_pyramids) and len(target_pyramids) > 0
        self.mctx = mctx
        self.n_pairs, self.n_levels = len(target_pyramids), len(target_pyramids[0])
        t_flat = [lv for p in target_pyramids for lv in p]
        s_flat = [lv for p in source_pyramids for lv in p]
        self._keep = (t_flat, s_flat)
        th = (C.c_void_p * len(t_flat))(*[lv.handle for lv in t_flat])
        sh = (C.c_void_p * len(s_flat))(*[lv.handle for lv in s_flat])
        self.handle = C.c_void_p()
        st = mctx.lib.a3d_multiscale_batch_new_multi(mctx.handle, params.to_c_array(), len(params), self.n_pairs,
                                                     self.n_levels, th, sh, C.byref(self.handle))
        if st == _abi.A3D_INVALID_PARAMETER:
            raise _abi.InvalidParameter(mctx.lib.a3d_last_error().decode())
        _abi.check(st, "a3d_multiscale_batch_new_multi")

    def align(self):
        """Runs every pair on its device; returns (list of Transform, int32 status array, [n_pairs][16] matrices as
        gathered on the first device)."""
        poses = (_abi.PoseC * self.n_pairs)()
        status = np.zeros(self.n_pairs, np.int32)
        mats = np.zeros((self.n_pairs, 16), np.float32)
        _abi.check(self.mctx.lib.a3d_multiscale_multi_batch_align(self.handle, poses, _abi.ptr(mats),
                                                                  status.ctypes.data_as(C.POINTER(C.c_int32)), None),
                   "a3d_multiscale_multi_batch_align")
        return [Transform.from_c(p) for p in poses], status, mats

    def free(self):
        if self.handle and self.mctx.handle:
            self.mctx.lib.a3d_multiscale_multi_batch_free(self.handle)
            self.handle = C.c_void_p()
