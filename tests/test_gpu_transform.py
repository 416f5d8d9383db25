"""Row a9 on the DEVICE: Transform::exp / Mul / transform_vector / transform_normal as the ICP kernels' tail evaluates
them (a3d_selftest_transform runs the tail's own device functions), against the reference's known answers
(src/transform.rs:321-411) and against the oracle on random updates — including the branches an ICP trace only reaches
by luck: theta > pi/4 (device libm instead of the minimax kernels), theta^2 < 1e-16 (Taylor quaternion), and
theta^2 < 1e-8 (V = I + W/2)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import _abi

pytestmark = pytest.mark.gpu


def device_transform(ctx, updates, poses, points):
    updates = np.ascontiguousarray(updates, np.float32).reshape(-1, 6)
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    n = len(updates)
    assert len(points) == n
    parr = None
    if poses is not None:
        parr = (_abi.PoseC * n)()
        for i, p in enumerate(poses):
            parr[i].t[:], parr[i].q[:] = p.t[:], p.q[:]
    out = (_abi.PoseC * n)()
    pts, nrm = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    _abi.check(ctx.lib.a3d_selftest_transform(ctx.handle, _abi.ptr(updates), parr, _abi.ptr(points), n, out, _abi.ptr(pts),
                                              _abi.ptr(nrm)), "a3d_selftest_transform")
    return out, pts, nrm


def test_reference_known_answers_on_the_device(ctx):
    half, quarter = np.float32(np.pi) / np.float32(2), np.float32(np.pi) / np.float32(4)
    zero = [0, 0, 0, 0, 0, 0]
    updates = [[1.0, 2.0, 3.0, 0.4, 0.5, 0.3],   # test_exp, transform.rs:364-388 (theta = 0.707: minimax branch)
               [1.0, 2.0, 3.0, 0.4, 0.5, 0.3],
               zero,                              # test_mul_op / test_transform, :321-362: exp(0) * (rot pi about y, z+3)
               zero,                              # identity
               [0, 0, 3, 0, 0, 0]]                # test_compose, :390-411: translate(0,0,3) * (rot pi/2 about y, z+3)
    poses = [O.pose(), O.pose(), O.pose(t=(0, 0, 3), q=(0, np.sin(half), 0, np.cos(half))), O.pose(),
             O.pose(t=(0, 0, 3), q=(0, np.sin(quarter), 0, np.cos(quarter)))]
    points = [[5.5, 6.4, 7.8], [1, 2, 3], [1, 2, 3], [4, 5, 6], [1, 2, 3]]
    _, pts, _ = device_transform(ctx, updates, poses, points)
    assert np.all(np.abs(pts[0] - np.float32([8.9848175, 6.9635687, 9.880962])) < 1e-5)
    assert np.linalg.norm(pts[1] - np.float32([3.5280778, 2.8378963, 5.8994026])) < 1e-5
    assert np.all(np.abs(pts[2] - np.float32([-1, 2, 0])) < 1e-5)
    assert np.array_equal(pts[3], np.float32([4, 5, 6]))  # the identity moves nothing, exactly
    assert np.all(np.abs(pts[4] - np.float32([2.9999998, 2.0, 5.0])) < 1e-5)


def test_every_exp_branch_against_the_oracle(ctx):
    rng = np.random.default_rng(7)
    n = 4096
    scale = np.empty(n, np.float32)
    scale[:1024] = 10.0 ** rng.uniform(-3, -1.3, 1024)      # ICP-sized updates (minimax kernels)
    scale[1024:2048] = rng.uniform(0.8, 3.0, 1024)         # theta > pi/4: device libm
    scale[2048:3072] = 10.0 ** rng.uniform(-9.5, -8.2, 1024)  # theta^2 < 1e-16: Taylor quaternion, theta := 0
    scale[3072:] = 10.0 ** rng.uniform(-7.5, -4.2, 1024)    # 1e-16 <= theta^2 < 1e-8: V = I + W / 2
    omega = rng.normal(size=(n, 3))
    omega = (omega / np.linalg.norm(omega, axis=1, keepdims=True) * scale[:, None]).astype(np.float32)
    updates = np.concatenate([rng.normal(size=(n, 3)).astype(np.float32) * 0.05, omega], axis=1)
    theta = np.sqrt((omega.astype(np.float64) ** 2).sum(1))
    assert (theta > 0.7854).sum() > 900 and (theta ** 2 < 1e-16).sum() > 900 and ((theta ** 2 >= 1e-16) & (theta ** 2 < 1e-8)).sum() > 900
    poses = []
    for i in range(n):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        poses.append(O.pose(t=rng.normal(size=3) * 2, q=q.astype(np.float32)))
    points = (rng.normal(size=(n, 3)) * 3).astype(np.float32)
    got, pts, nrm = device_transform(ctx, updates, poses, points)
    worst_t = worst_q = worst_p = worst_n = 0.0
    for i in range(n):
        ref = O.compose(O.exp_se3(updates[i]), poses[i])
        rt, rq = O.pose_tuple(ref)
        gt, gq = np.float32(got[i].t[:]), np.float32(got[i].q[:])
        worst_t, worst_q = max(worst_t, float(np.abs(gt - rt).max())), max(worst_q, float(np.abs(gq - rq).max()))
        rp = O.transform_points(ref, points[i:i + 1])[0]
        rn = np.empty((1, 3), np.float32)
        O.load().orc_transform_normals(C.byref(ref), _abi.ptr(points[i:i + 1].copy()), 1, _abi.ptr(rn))
        scale_p = max(1.0, float(np.abs(rp).max()))
        worst_p = max(worst_p, float(np.abs(pts[i] - rp).max()) / scale_p)
        worst_n = max(worst_n, float(np.abs(nrm[i] - rn[0]).max()) / scale_p)
    print(f"device vs oracle: translation {worst_t:.2e}, quaternion {worst_q:.2e}, point {worst_p:.2e}, normal {worst_n:.2e}")
    # 1e-6: f32 round-off of a handful of operations (sin / cos within 1 ulp of the host libm's)
    assert worst_t <= 2e-6 and worst_q <= 1e-6 and worst_p <= 2e-6 and worst_n <= 2e-6


def test_update_of_exactly_zero_is_the_identity_on_the_device(ctx):
    got, pts, _ = device_transform(ctx, [[0, 0, 0, 0, 0, 0]], None, [[1.5, -2.5, 3.25]])
    assert tuple(got[0].t[:]) == (0.0, 0.0, 0.0) and tuple(got[0].q[:]) == (0.0, 0.0, 0.0, 1.0)
    assert np.array_equal(pts[0], np.float32([1.5, -2.5, 3.25]))
