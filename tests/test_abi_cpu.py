"""CPU checks of the C-ABI library and the host-side mirror: the .so loads, exports every symbol the
header declares, reports "no device" as a status, and the parameter/host logic matches the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import _abi
from align3d_amd import IcpParams, MsIcpParams, RangeImage, CameraIntrinsics
from host_frame_prep import (intensity_map_from_luma, rgb_to_luma_u8, _resize_pick, blur_rgb_and_halve, from_rgbd_image)
from data_util import SlamTbSample

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()
    header = open(os.path.join(ROOT, "include", "align3d_hip.h")).read()
    # what the header declares under #ifdef A3D_DIAGNOSTICS is exported by the diagnostics build only
    diag_part = "".join(re.findall(r"#ifdef A3D_DIAGNOSTICS(.*?)#endif /\* A3D_DIAGNOSTICS \*/", header, flags=re.S))
    product_part = re.sub(r"#ifdef A3D_DIAGNOSTICS.*?#endif /\* A3D_DIAGNOSTICS \*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(a3d_[a-z0-9_]+)\s*\(", product_part))
    diag_declared = set(re.findall(r"\b(a3d_[a-z0-9_]+)\s*\(", diag_part))
    assert len(declared) >= 40 and diag_declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/align3d_hip.h but not exported"
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    assert diag_declared == set(_abi.DIAG_SIGNATURES), diag_declared ^ set(_abi.DIAG_SIGNATURES)
    assert lib.a3d_abi_version() == 1
    for name in diag_declared:
        assert not hasattr(lib, name), f"{name} is a diagnostics entry point but the product library exports it"
    diag = _abi.load_library(_abi.DIAG_LIB_PATH)
    for name in declared | diag_declared:
        assert hasattr(diag, name), f"{name} missing from the diagnostics build"


def test_product_library_carries_no_diagnostics():
    """no environment knob, no rocPRIM, none of the slower kernel variants in libalign3d_hip.so."""
    blob = open(_abi.LIB_PATH, "rb").read()
    for needle in (b"A3D_ICP_NOSOLVE", b"A3D_ICP_", b"A3D_KD", b"A3D_BILATERAL", b"A3D_BUILDER", b"rocprim", b"mfma_kernel",
                   b"image_icp_level_kernel", b"image_icp_exact_kernel"):
        assert needle not in blob, needle
    diag = open(_abi.DIAG_LIB_PATH, "rb").read()
    assert b"A3D_ICP_NOSOLVE" in diag and b"rocprim" in diag


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.IcpParamsC) == 32
    assert C.sizeof(_abi.PoseC) == 28
    assert C.sizeof(_abi.RangeImageViewC) == 88
    assert C.sizeof(_abi.PointCloudViewC) == 24
    assert C.sizeof(_abi.GnStateC) == 184


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_is_a_loud_error_not_a_fallback():
    from align3d_amd import A3dError, Context

    with pytest.raises(A3dError) as e:
        Context(0)
    assert e.value.status == _abi.A3D_HIP_ERROR


# src/icp/icp_params.rs:33-43 and :112-133
def test_param_defaults():
    p = IcpParams.default()
    assert p.max_iterations == 15 and p.weight == 1.0 and p.max_distance == 0.5
    assert np.float32(p.color_weight) == np.float32(0.1) and np.float32(p.max_color_distance) == np.float32(0.25)
    assert np.float32(p.max_normal_angle) == np.float32(18.0) * (np.float32(np.pi) / np.float32(180.0))
    assert np.float32(p.max_point_to_plane_distance) == np.float32(0.1)
    ms = MsIcpParams.default()
    assert len(ms) == 3 and [q.max_iterations for q in ms] == [20, 20, 30]
    for q in ms:
        assert q.weight == 1.0 and q.color_weight == 1.0 and np.float32(q.max_color_distance) == np.float32(2.75)
        assert np.float32(q.max_normal_angle) == np.float32(np.pi) / np.float32(10.0)
    # the oracle's Python-side defaults are the same numbers
    o = O.params()
    for f in ("max_iterations", "weight", "color_weight", "max_distance", "max_normal_angle", "max_color_distance"):
        assert getattr(o, f) == getattr(p.to_c(), f), f
    om = O.ms_default_params()
    for i in range(3):
        assert om[i].max_normal_angle == ms[i].to_c().max_normal_angle and om[i].max_iterations == ms[i].max_iterations


def test_ms_params_builder_api():
    ms = MsIcpParams.repeat(3, IcpParams.default()).customize(lambda i, p: setattr(p, "max_iterations", 5 + i))
    assert [p.max_iterations for p in ms.iter()] == [5, 6, 7] and not ms.is_empty() and ms.len() == 3
    ms[1].weight = 2.0
    assert ms[0].weight == 1.0 and ms[1].weight == 2.0  # repeat() copies


def test_host_frame_preparation_matches_oracle():
    """The numpy restatements of the 'next' rows agree with the oracle's C++ ones bit for bit."""
    s = SlamTbSample("sample1")
    depth, rgb = s.load(0)
    fx, fy, cx, cy = s.intrinsics(0)
    fr = O.build_frame(depth, rgb, fx, fy, cx, cy, s.depth_scale(0))
    ri = from_rgbd_image(CameraIntrinsics(fx, fy, cx, cy, 640, 480), depth, rgb, s.depth_scale(0))
    assert np.array_equal(ri.mask, fr.mask) and ri.valid_points_count() == 270213
    assert np.array_equal(ri.points.view(np.uint32), fr.points.view(np.uint32))
    luma = rgb_to_luma_u8(rgb)
    assert np.array_equal(luma.reshape(-1), fr.intensities)
    assert np.array_equal(intensity_map_from_luma(luma), fr.intensity_map)
    # pyramid step: nearest-to-mean picks for points and normals
    down = O.pyr_down(fr)
    pts, anyv = _resize_pick(fr.points, fr.mask, 240, 320)
    assert np.array_equal(anyv.astype(np.uint8), down.mask)
    assert np.array_equal(pts.view(np.uint32), down.points.view(np.uint32))
    nrm, _ = _resize_pick(fr.normals, fr.mask, 240, 320)
    assert np.array_equal(nrm.view(np.uint32), down.normals.view(np.uint32))
    # the (unpinned) colour blur: the two restatements agree to one grey level
    small = blur_rgb_and_halve(rgb, 1.0)
    assert small.shape == (240, 320, 3)
    assert np.max(np.abs(small.astype(int) - down.colors.astype(int))) <= 1


def test_cpp_host_mirror_cpu_checks():
    """include/align3d.hpp (C++ mirror of the reference API) compiles with g++ and its host logic holds."""
    import subprocess

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "host_mirror_test")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout or "GPU is present" in out.stdout


def test_multi_device_sharding_and_argument_validation_without_a_gpu():
    """The multi-GPU entry points below Python (a3d_multi_*): the partition rule is the one of SURVEY §8e (and of
    align3d_amd.distributed.shard_range), malformed arguments are statuses, and without a GPU the context list
    reports A3D_HIP_ERROR like a3d_context_create."""
    from align3d_amd.distributed import shard_range as py_shard
    from align3d_amd.multi import shard_range

    for n, d in [(512, 8), (64, 1), (10, 3), (7, 8), (0, 4), (513, 8)]:
        blocks = [shard_range(n, d, k) for k in range(d)]
        assert blocks == [py_shard(n, d, k) for k in range(d)]
        assert [j for lo, hi in blocks for j in range(lo, hi)] == list(range(n))
    assert shard_range(512, 8, 3) == (192, 256)
    lib = _abi.load_library()
    lo, hi = C.c_uint64(), C.c_uint64()
    assert lib.a3d_multi_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) == 1      # no devices
    assert lib.a3d_multi_shard_range(10, 2, 2, C.byref(lo), C.byref(hi)) == 1      # device index out of range
    assert lib.a3d_multi_shard_range(10, 2, 1, None, C.byref(hi)) == 1
    h = C.c_void_p()
    assert lib.a3d_multi_context_create(None, 1, C.byref(h)) == 1
    ids = (C.c_int32 * 2)(0, 0)
    assert lib.a3d_multi_context_create(ids, 0, C.byref(h)) == 1
    assert lib.a3d_multi_context_size(None) == 0 and not lib.a3d_multi_context_device(None, 0)
    assert lib.a3d_multiscale_batch_new_multi(None, None, 3, 4, 3, None, None, C.byref(h)) == 1
    assert lib.a3d_multiscale_multi_batch_align(None, None, None, None, None) == 1
    assert lib.a3d_multiscale_multi_batch_free(None) == 0 and lib.a3d_multi_context_destroy(None) == 0
    import torch
    if not torch.cuda.is_available():
        assert lib.a3d_multi_context_create(ids, 2, C.byref(h)) == 4  # A3D_HIP_ERROR: no device
        assert b"no HIP device" in lib.a3d_last_error()
