"""ctypes mirror of include/align3d_hip.h: POD structs, status codes and the loader of
libalign3d_hip.so.  The product path has no CPU fallback: if the HIP library is missing or a call
fails, an exception is raised (never a silent detour)."""
import ctypes as C
import os
import threading

import numpy as np

A3D_OK = 0
A3D_INVALID_PARAMETER = 1
A3D_MISSING_FIELD = 2
A3D_SOLVE_FAILED = 3
A3D_HIP_ERROR = 4
A3D_NAN_IN_INPUT = 5
A3D_CAST_OVERFLOW = 6

STATUS_NAMES = {
    0: "A3D_OK",
    1: "A3D_INVALID_PARAMETER",
    2: "A3D_MISSING_FIELD",
    3: "A3D_SOLVE_FAILED",
    4: "A3D_HIP_ERROR",
    5: "A3D_NAN_IN_INPUT",
    6: "A3D_CAST_OVERFLOW",
}


class A3dError(Exception):
    """Mirror of A3dError (src/error.rs:3-9) plus the status codes that stand in for the
    reference's panics."""

    def __init__(self, status, message=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class InvalidParameter(A3dError):
    def __init__(self, message):
        super().__init__(A3D_INVALID_PARAMETER, message)


class IcpParamsC(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_uint64),
        ("weight", C.c_float),
        ("color_weight", C.c_float),
        ("max_point_to_plane_distance", C.c_float),
        ("max_distance", C.c_float),
        ("max_normal_angle", C.c_float),
        ("max_color_distance", C.c_float),
    ]


class PoseC(C.Structure):
    _fields_ = [("t", C.c_float * 3), ("q", C.c_float * 4)]


class RangeImageViewC(C.Structure):
    _fields_ = [
        ("points", C.c_void_p),
        ("mask", C.c_void_p),
        ("normals", C.c_void_p),
        ("intensities", C.c_void_p),
        ("intensity_map", C.c_void_p),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("width", C.c_uint64),
        ("height", C.c_uint64),
    ]


class PointCloudViewC(C.Structure):
    _fields_ = [("points", C.c_void_p), ("normals", C.c_void_p), ("len", C.c_uint64)]


class BuilderParamsC(C.Structure):
    _fields_ = [
        ("with_normals", C.c_uint32),
        ("with_intensity", C.c_uint32),
        ("use_bilateral", C.c_uint32),
        ("pad", C.c_uint32),
        ("sigma_space", C.c_double),
        ("sigma_color", C.c_double),
        ("pyramid_levels", C.c_uint64),
        ("blur_sigma", C.c_float),
        ("pad2", C.c_uint32),
    ]


class GnStateC(C.Structure):
    _fields_ = [
        ("hessian", C.c_float * 36),
        ("gradient", C.c_float * 6),
        ("squared_residual_sum", C.c_float),
        ("count", C.c_uint64),
    ]

    def as_dict(self):
        return {
            "H": np.array(self.hessian[:], dtype=np.float32).reshape(6, 6),
            "g": np.array(self.gradient[:], dtype=np.float32),
            "ssq": np.float32(self.squared_residual_sum),
            "count": int(self.count),
        }


def ptr(arr):
    """Raw address of a C-contiguous numpy array (None -> NULL)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"], "arrays cross the ABI in standard layout"
    return arr.ctypes.data


# Every symbol include/align3d_hip.h declares, with (restype, argtypes).
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_ST = C.c_int
SIGNATURES = {
    "a3d_abi_version": (C.c_uint32, []),
    "a3d_last_error": (C.c_char_p, []),
    "a3d_status_string": (C.c_char_p, [C.c_int]),
    "a3d_device_count": (_ST, [C.POINTER(C.c_int32)]),
    "a3d_context_create": (_ST, [C.c_int32, _PP]),
    "a3d_context_create_with_priority": (_ST, [C.c_int32, C.c_int32, _PP]),
    "a3d_context_create_pair": (_ST, [C.c_int32, _PP, _PP]),
    "a3d_context_destroy": (_ST, [_P]),
    "a3d_context_synchronize": (_ST, [_P]),
    "a3d_context_stream": (_P, [_P]),
    "a3d_context_device": (C.c_int32, [_P]),
    "a3d_timer_start": (_ST, [_P]),
    "a3d_timer_stop": (_ST, [_P, C.POINTER(C.c_float)]),
    "a3d_malloc": (_ST, [_P, C.c_size_t, _PP]),
    "a3d_host_alloc": (_ST, [_P, C.c_size_t, _PP]),
    "a3d_host_free": (_ST, [_P, _P]),
    "a3d_free": (_ST, [_P, _P]),
    "a3d_memcpy_h2d": (_ST, [_P, _P, _P, C.c_size_t]),
    "a3d_memcpy_d2h": (_ST, [_P, _P, _P, C.c_size_t]),
    "a3d_memcpy_d2d": (_ST, [_P, _P, _P, C.c_size_t]),
    "a3d_icp_params_default": (None, [C.POINTER(IcpParamsC)]),
    "a3d_ms_icp_params_default": (None, [C.POINTER(IcpParamsC)]),
    "a3d_range_image_upload": (_ST, [_P, C.POINTER(RangeImageViewC), _PP]),
    "a3d_range_image_upload_pyramid": (_ST, [_P, C.POINTER(RangeImageViewC), C.c_uint64, _PP]),
    "a3d_range_image_free": (_ST, [_P]),
    "a3d_range_image_compute_normals": (_ST, [_P]),
    "a3d_range_image_download_normals": (_ST, [_P, _P]),
    "a3d_builder_params_default": (None, [C.POINTER(BuilderParamsC)]),
    "a3d_range_image_build_pyramid": (
        _ST,
        [_P, C.POINTER(BuilderParamsC), _P, _P, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double,
         C.c_double, C.c_double, _PP],
    ),
    "a3d_range_image_build_pyramids": (
        _ST,
        [_P, C.POINTER(BuilderParamsC), C.c_uint64, _PP, _PP, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double,
         C.c_double, C.c_double, _PP],
    ),
    "a3d_context_last_build_stats": (_ST, [_P, C.POINTER(C.c_uint64)]),
    "a3d_context_set_build_profiling": (_ST, [_P, C.c_int32]),
    "a3d_context_last_build_kernel_ms": (_ST, [_P, C.POINTER(C.c_float)]),
    "a3d_range_image_size": (_ST, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "a3d_range_image_download": (_ST, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_double)]),
    "a3d_compute_normals": (_ST, [_P, _P, _P, C.c_uint64, C.c_uint64, _P]),
    "a3d_range_image_compute_normals_batch": (_ST, [_PP, C.c_uint64]),
    "a3d_image_icp_align": (_ST, [_P, C.POINTER(IcpParamsC), _P, _P, C.POINTER(PoseC), C.POINTER(PoseC)]),
    "a3d_image_icp_accumulate": (
        _ST,
        [_P, C.POINTER(IcpParamsC), _P, _P, C.POINTER(PoseC), C.POINTER(GnStateC), C.POINTER(GnStateC)],
    ),
    "a3d_image_icp_align_trace": (
        _ST,
        [_P, C.POINTER(IcpParamsC), _P, _P, C.POINTER(PoseC), C.POINTER(PoseC), _P],
    ),
    "a3d_selftest_division": (_ST, [_P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "a3d_selftest_transform": (_ST, [_P, _P, _P, _P, C.c_uint64, _P, _P, _P]),
    "a3d_multiscale_new": (_ST, [_P, C.POINTER(IcpParamsC), C.c_uint64, _PP, C.c_uint64, _PP]),
    "a3d_multiscale_align": (_ST, [_P, _PP, C.c_uint64, C.POINTER(PoseC)]),
    "a3d_multiscale_align_host": (_ST, [_P, C.POINTER(RangeImageViewC), C.c_uint64, C.POINTER(PoseC)]),
    "a3d_multiscale_free": (_ST, [_P]),
    "a3d_multiscale_batch_new": (
        _ST,
        [_P, C.POINTER(IcpParamsC), C.c_uint64, C.c_uint64, C.c_uint64, _PP, _PP, _PP],
    ),
    "a3d_multiscale_batch_align": (_ST, [_P, C.POINTER(PoseC), _P, C.POINTER(C.c_int32)]),
    "a3d_multiscale_batch_free": (_ST, [_P]),
    "a3d_multiscale_batch_rebind": (_ST, [_P, _P, _P]),
    "a3d_multiscale_batch_last_level_ms": (_ST, [_P, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "a3d_multiscale_batch_results": (_ST, [_P, C.POINTER(PoseC), C.POINTER(C.c_int32)]),
    "a3d_multiscale_batch_last_timing": (_ST, [_P, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "a3d_multiscale_batch_set_profiling": (_ST, [_P, C.c_int32]),
    "a3d_multiscale_batch_last_kernel_ms": (_ST, [_P, C.POINTER(C.c_float)]),
    "a3d_multiscale_batch_concurrency": (_ST, [_P, C.POINTER(C.c_uint32)]),
    "a3d_multiscale_batch_persistent_levels": (_ST, [_P, C.POINTER(C.c_uint32)]),
    "a3d_context_set_tiling": (_ST, [_P, C.c_uint32]),
    "a3d_context_create_on_pipe": (_ST, [C.c_int32, C.c_int32, C.c_int32, _PP]),
    "a3d_multi_shard_range": (_ST, [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "a3d_multi_context_create": (_ST, [C.POINTER(C.c_int32), C.c_uint64, _PP]),
    "a3d_multi_context_destroy": (_ST, [_P]),
    "a3d_multi_context_size": (C.c_uint64, [_P]),
    "a3d_multi_context_device": (_P, [_P, C.c_uint64]),
    "a3d_multiscale_batch_new_multi": (
        _ST,
        [_P, C.POINTER(IcpParamsC), C.c_uint64, C.c_uint64, C.c_uint64, _PP, _PP, _PP],
    ),
    "a3d_multiscale_multi_batch_align": (_ST, [_P, C.POINTER(PoseC), _P, C.POINTER(C.c_int32), _PP]),
    "a3d_multiscale_multi_batch_free": (_ST, [_P]),
    "a3d_kdtree_stats": (_ST, [_P, C.POINTER(C.c_uint64)]),
    "a3d_kdtree_download": (_ST, [_P, _P, _P, C.POINTER(C.c_uint64)]),
    "a3d_kdtree_new": (_ST, [_P, _P, C.c_uint64, _PP]),
    "a3d_kdtree_new_device": (_ST, [_P, _P, C.c_uint64, _PP]),
    "a3d_kdtree_build_path": (_ST, [_P, C.POINTER(C.c_int32)]),
    "a3d_kdtree_build_ms": (_ST, [_P, C.POINTER(C.c_float)]),
    "a3d_kdtree_nearest": (_ST, [_P, _P, C.c_uint64, _P, _P]),
    "a3d_kdtree_nearest_device": (_ST, [_P, _P, C.c_uint64, _P, _P]),
    "a3d_kdtree_free": (_ST, [_P]),
    "a3d_pcl_icp_new": (_ST, [_P, C.POINTER(IcpParamsC), C.POINTER(PointCloudViewC), _PP]),
    "a3d_pcl_icp_align": (_ST, [_P, C.POINTER(PointCloudViewC), C.POINTER(PoseC)]),
    "a3d_pcl_icp_new_device": (_ST, [_P, C.POINTER(IcpParamsC), C.POINTER(PointCloudViewC), _PP]),
    "a3d_pcl_icp_align_device": (_ST, [_P, C.POINTER(PointCloudViewC), C.POINTER(PoseC)]),
    "a3d_pcl_icp_accumulate": (_ST, [_P, C.POINTER(PointCloudViewC), C.POINTER(PoseC), C.POINTER(GnStateC)]),
    "a3d_pcl_icp_last_device_ms": (_ST, [_P, C.POINTER(C.c_float)]),
    "a3d_pcl_icp_free": (_ST, [_P]),
    "a3d_bilateral_default_sigmas": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "a3d_bilateral_filter_u16": (
        _ST,
        [_P, _P, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _P, C.POINTER(C.c_uint64)],
    ),
    "a3d_bilateral_filter_u16_device": (_ST, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _P]),
}

# Exported by the diagnostics build only (csrc/Makefile `diag`: -DA3D_DIAGNOSTICS).
DIAG_SIGNATURES = {
    "a3d_image_icp_accumulate_exact": (
        _ST,
        [_P, C.POINTER(IcpParamsC), _P, _P, C.POINTER(PoseC), C.POINTER(GnStateC), C.POINTER(GnStateC)],
    ),
    "a3d_image_icp_accumulate_weighted": (
        _ST,
        [_P, C.POINTER(IcpParamsC), _P, _P, C.POINTER(PoseC), C.POINTER(GnStateC)],
    ),
}

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# (A3D_LIBRARY: a probe script's own build of the same sources, scripts/build_variant.sh; never another implementation)
LIB_PATH = os.environ.get("A3D_LIBRARY") or os.path.join(_CSRC, "libalign3d_hip.so")
# The diagnostics build of the same sources: environment knobs, cross-check kernels and the variants that were measured
# slower live only there (tests and probes load it through Context(library=DIAG_LIB_PATH); the product never does).
DIAG_LIB_PATH = os.path.join(_CSRC, "libalign3d_hip_diag.so")
_libs = {}


def load_library(path=None):
    """Loads libalign3d_hip.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if it is missing: there is
    deliberately no other implementation to fall back to.  `path`: another build of the same library (DIAG_LIB_PATH)."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C align3d_amd/csrc all)"
        )
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
        if restype is _ST:
            _note_failures(lib, fn)
    for name, (restype, argtypes) in DIAG_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = argtypes
            if restype is _ST:
                _note_failures(lib, fn)
    if lib.a3d_abi_version() != 1:
        raise RuntimeError("libalign3d_hip.so ABI version mismatch")
    _libs[path] = lib
    return lib


_failed = threading.local()  # .lib: the library whose call most recently returned a failure status on this thread


def _note_failures(lib, fn):
    """ctypes errcheck hook for a status-returning entry point: remembers on which LIBRARY a call failed, so that check()
    reads a3d_last_error from that one (the product and the diagnostics build can both be loaded; the text is the
    library's most recent failure and is not cleared on success, so the other library's would be stale)."""
    def errcheck(result, func, args):
        if result != A3D_OK:
            _failed.lib = lib
        return result
    fn.errcheck = errcheck


def check(status, what="", lib=None):
    """Raises A3dError for a failed call, with the a3d_last_error text of the library the failing call ran on."""
    if status != A3D_OK:
        lib = lib or getattr(_failed, "lib", None) or load_library()
        msg = lib.a3d_last_error().decode("utf-8", "replace")
        raise A3dError(status, f"{what}: {msg}" if what else msg)
