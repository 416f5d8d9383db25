"""GPU parity for Icp (kd-tree point-to-plane ICP, src/icp/pcl_icp.rs)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import A3dError, Icp, IcpParams, PointCloud, Transform
from align3d_amd._abi import GnStateC, PoseC
from gpu_util import gn_rel_err, oracle_frame, small_pose, to_range_image, transform_diff

pytestmark = pytest.mark.gpu


def _clouds(sample, a, b):
    ta = PointCloud.from_range_image(to_range_image(oracle_frame(sample, a, True)))
    sb = PointCloud.from_range_image(to_range_image(oracle_frame(sample, b, True)))
    return ta, sb


def _oracle_accumulate(prm, tree, tgt, src, T, f64=True):
    g = GnStateC()
    tv, sv = O.pcl_view(tgt.points, tgt.normals), O.pcl_view(src.points, src.normals)
    p, t = prm.to_c(), T.to_c()
    st = O.load().orc_pcl_icp_accumulate(C.byref(p), tree.h, C.byref(tv), C.byref(sv), C.byref(t), int(f64), C.byref(g))
    assert st == 0
    return g.as_dict()


def test_pcl_icp_per_iteration_and_end_to_end(ctx):
    # Icp::test_icp shape (src/icp/pcl_icp.rs:122-136): sample1 frames 0 / 1, 5 iterations
    tgt, src = _clouds("sample1", 0, 1)
    prm = IcpParams(max_iterations=5)
    tree = O.KdTree(tgt.points)
    icp = Icp.new(ctx, prm, tgt)
    for T in (Transform.eye(), small_pose(2)):
        ref = _oracle_accumulate(prm, tree, tgt, src, T)
        gpu = icp.accumulate(src, T)
        # correspondences come from bit-exact kd-tree queries on bit-exact transformed points
        assert gpu["count"] == ref["count"] and ref["count"] > 1000
        eh, eg, es = gn_rel_err(gpu, ref)
        assert eh < 1e-6 and eg < 1e-6 and es < 1e-6
    out = PoseC()
    tv, sv = O.pcl_view(tgt.points, tgt.normals), O.pcl_view(src.points, src.normals)
    p = prm.to_c()
    assert O.load().orc_pcl_icp_align(C.byref(p), tree.h, C.byref(tv), C.byref(sv), C.byref(out), None) == 0
    T_gpu = icp.align(src)
    ang, tr = transform_diff(T_gpu, out)
    print(f"[pcl icp sample1 0<-1] d_angle={ang:.3e} d_trans={tr:.3e}")
    assert ang <= 1e-4 and tr <= 1e-4


def test_pcl_icp_ignores_initial_transform_and_uses_weight(ctx):
    tgt, src = _clouds("sample1", 0, 5)
    src = PointCloud(src.points[::3], src.normals[::3])
    prm = IcpParams(max_iterations=3, weight=0.7)
    icp = Icp.new(ctx, prm, tgt)
    icp.initial_transform = small_pose(9)  # the reference starts from eye() regardless (pcl_icp.rs:59)
    tree = O.KdTree(tgt.points)
    out = PoseC()
    tv, sv = O.pcl_view(tgt.points, tgt.normals), O.pcl_view(src.points, src.normals)
    p = prm.to_c()
    assert O.load().orc_pcl_icp_align(C.byref(p), tree.h, C.byref(tv), C.byref(sv), C.byref(out), None) == 0
    ang, tr = transform_diff(icp.align(src), out)
    assert ang <= 1e-4 and tr <= 1e-4


def test_pcl_icp_missing_normals(ctx):
    tgt, src = _clouds("sample1", 0, 1)
    with pytest.raises(A3dError) as e:
        Icp.new(ctx, IcpParams.default(), PointCloud(tgt.points)).align(src)
    assert e.value.status == 2
    with pytest.raises(A3dError) as e:
        Icp.new(ctx, IcpParams.default(), tgt).align(PointCloud(src.points))
    assert e.value.status == 2
