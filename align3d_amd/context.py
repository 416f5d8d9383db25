"""Device context: one GPU, one HIP stream (a3d_context)."""
import ctypes as C

from . import _abi


class Context:
    def __init__(self, device_index=0, priority=0, _handle=None, pair=True, library=None, main_slot=-1):
        """priority < 0: the device's highest stream priority (a frame-builder context next to an aligning one),
        0: default, > 0: lowest.  A default-priority context is created together with its sibling (the builder context
        `sibling()` returns): a3d_context_create_pair puts the two contexts' streams on the GPU's compute pipes in a
        fixed relation, which contexts created at unrelated moments do not have.  `pair=False` creates the aligning
        context alone (four streams instead of eight, no second pinned block); `sibling()` then creates the builder on
        first use, wherever the runtime places its streams at that moment.  `library`: path of another build of the
        library (_abi.DIAG_LIB_PATH: the diagnostics build that tests and probes use)."""
        self.lib = _abi.load_library(library)
        self._library = library
        self.handle = C.c_void_p()
        self.device_index = int(device_index)
        self._sibling = None
        if _handle is not None:
            self.handle = _handle
        elif int(priority) == 0 and pair:
            builder = C.c_void_p()
            _abi.check(self.lib.a3d_context_create_pair(int(device_index), C.byref(self.handle), C.byref(builder)),
                       "a3d_context_create")
            self._sibling = Context(device_index, priority=-1, _handle=builder, library=library)
        else:  # (main_slot: a3d_context_create_on_pipe — a further aligning context beside the first one)
            _abi.check(self.lib.a3d_context_create_on_pipe(int(device_index), int(priority), int(main_slot),
                                                           C.byref(self.handle)), "a3d_context_create")

    def sibling(self):
        """The builder context on the same GPU (its own streams and scratch, highest stream priority), created with this
        one and closed with it: frame builds run on it while this context aligns (align3d_amd.odometry, bench.py)."""
        if self._sibling is None:
            self._sibling = Context(self.device_index, priority=-1, library=self._library)
        return self._sibling

    def set_tiling(self, tiles_per_pair):
        """a3d_context_set_tiling: 0 = throughput tiling (the cut of a pair's pixels into blocks follows the batch size);
        n > 0 = pinned tiling: n blocks per (pair, level) whatever the batch, so that a pair's pose is bit-identical
        alone and in any batch."""
        _abi.check(self.lib.a3d_context_set_tiling(self.handle, int(tiles_per_pair)), "a3d_context_set_tiling")

    def device(self):
        """The HIP device the context sits on (a3d_context_device)."""
        return int(self.lib.a3d_context_device(self.handle))

    def synchronize(self):
        _abi.check(self.lib.a3d_context_synchronize(self.handle))

    def timer_start(self):
        _abi.check(self.lib.a3d_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        _abi.check(self.lib.a3d_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def malloc(self, nbytes):
        p = C.c_void_p()
        _abi.check(self.lib.a3d_malloc(self.handle, int(nbytes), C.byref(p)))
        return p

    def free(self, p):
        _abi.check(self.lib.a3d_free(self.handle, p))

    def pinned_empty(self, shape, dtype):
        """A numpy array in page-locked host memory (a3d_host_alloc); frames built from it are uploaded by DMA.
        The memory is released when the array (and every view of it) has been garbage collected."""
        import numpy as np

        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        _abi.check(self.lib.a3d_host_alloc(self.handle, n, C.byref(p)), "a3d_host_alloc")
        buf = (C.c_char * max(1, n)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        lib, handle = self.lib, self.handle

        import weakref
        weakref.finalize(buf, lambda: lib.a3d_host_free(handle, p) if handle else None)
        return arr

    def to_device(self, arr):
        p = self.malloc(arr.nbytes)
        _abi.check(self.lib.a3d_memcpy_h2d(self.handle, p, _abi.ptr(arr), arr.nbytes))
        return p

    def to_host(self, p, arr):
        _abi.check(self.lib.a3d_memcpy_d2h(self.handle, _abi.ptr(arr), p, arr.nbytes))
        return arr

    def last_build_stats(self):
        """{frames, grid_cells, marked_tiles, zero_tiles} of the most recent RangeImageBuilder call on this context."""
        v = (C.c_uint64 * 4)()
        _abi.check(self.lib.a3d_context_last_build_stats(self.handle, v))
        return {"frames": int(v[0]), "grid_cells": int(v[1]), "marked_tiles": int(v[2]), "zero_tiles": int(v[3])}

    def set_build_profiling(self, on):
        _abi.check(self.lib.a3d_context_set_build_profiling(self.handle, 1 if on else 0))

    def last_build_kernel_ms(self):
        """Device time of the builder's kernels in the most recent build (set_build_profiling(True) first)."""
        ms = C.c_float()
        _abi.check(self.lib.a3d_context_last_build_kernel_ms(self.handle, C.byref(ms)))
        return ms.value

    def release_lanes(self):
        """Closes the further aligning contexts run_odometry(in_flight > 1) keeps with this one."""
        for c in getattr(self, "_lanes", None) or []:
            c.close()
        self._lanes = []

    def close(self):
        self.release_lanes()
        if self._sibling is not None:
            self._sibling.close()
            self._sibling = None
        if self.handle:
            self.lib.a3d_context_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
