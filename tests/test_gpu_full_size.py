"""BASELINE's full sizes: a 64-pair batch of 640x480 synthetic frame pairs, 3 levels x 15 iterations (the benchmark's
workload) through size-independent properties (its per-pair oracle comparison is tests/test_gpu_headline_parity.py), and
configs[2] (500 000 x 500 000 points, 15 iterations) through properties AND against the oracle run at that size."""
import numpy as np
import pytest

from align3d_amd import (BilateralFilter, IcpParams, MsIcpParams, MultiscaleAlign, MultiscaleAlignBatch,
                         RangeImageBuilder, synth)

pytestmark = pytest.mark.gpu


def _as_array(poses):
    return np.array([np.concatenate([t.t, t.q]) for t in poses], np.float32)


def test_full_size_batch_properties(ctx):
    P, W, H = 64, 640, 480
    frames, gt = synth.frame_stream(1000, P + 1, W, H)  # the benchmark's rank-0 stream
    cam = synth.camera(W, H)
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    pyr = [builder.build_device(cam, d, rgb, synth.DEPTH_SCALE) for d, rgb in frames]
    prm = MsIcpParams.repeat(3, IcpParams.default())
    batch = MultiscaleAlignBatch(ctx, prm, pyr[:P], pyr[1:])
    poses, status = batch.align()
    assert not status.any()
    full = _as_array(poses)
    # determinism: the same launch sequence gives the same bits
    again, _ = batch.align()
    assert np.array_equal(full.view(np.uint32), _as_array(again).view(np.uint32))
    # the rotation estimates beat the identity against the synthetic ground truth; the translation stays within
    # 2 cm (IcpParams::default() with the reference's gates is not a precise estimator on this scene: the CPU
    # oracle gives the same numbers, bench.py cpu_baseline.max_gpu_vs_cpu_*)
    errs, idents = [], []
    for p in range(P):
        rel = synth.relative_pose(gt[p], gt[p + 1])
        d = np.linalg.inv(rel) @ poses[p].matrix().astype(np.float64)
        errs.append((np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)), np.linalg.norm(d[:3, 3])))
        idents.append((np.arccos(np.clip((np.trace(rel[:3, :3]) - 1) / 2, -1, 1)), np.linalg.norm(rel[:3, 3])))
    errs, idents = np.array(errs), np.array(idents)
    print(f"[full size] mean error {errs.mean(axis=0)}, identity {idents.mean(axis=0)}")
    assert errs[:, 0].mean() < 0.5 * idents[:, 0].mean() and errs[:, 1].mean() < 0.02 and errs[:, 1].max() < 0.05
    # Throughput tiling (the default): the cut of a pair into blocks follows the batch size, so two half batches and a
    # lone pair associate their f32 sums differently; IcpParams::default() is not contractive, which can
    # move a pose by ~1e-3 — reported here, asserted loosely.
    lo, _ = MultiscaleAlignBatch(ctx, prm, pyr[:32], pyr[1:33]).align()
    hi, _ = MultiscaleAlignBatch(ctx, prm, pyr[32:64], pyr[33:65]).align()
    halves = np.concatenate([_as_array(lo), _as_array(hi)])
    print(f"[full size] throughput tiling: 64-batch vs two 32-batches max |d| = {np.abs(halves - full).max():.2e}")
    assert np.abs(halves - full).max() < 5e-3
    # Pinned tiling (a3d_context_set_tiling): a pair's pose is the same BITS in the 64-pair batch, in a 32-pair batch,
    # in a 5-pair batch and alone.  24 blocks per pair = what the throughput tiling gives level 0 of
    # the 64-pair batch.
    ctx.set_tiling(24)
    try:
        b64 = MultiscaleAlignBatch(ctx, prm, pyr[:P], pyr[1:])
        pinned, st = b64.align()
        assert not st.any()
        pinned = _as_array(pinned).view(np.uint32)
        lo, _ = MultiscaleAlignBatch(ctx, prm, pyr[:32], pyr[1:33]).align()
        hi, _ = MultiscaleAlignBatch(ctx, prm, pyr[32:64], pyr[33:65]).align()
        assert np.array_equal(np.concatenate([_as_array(lo), _as_array(hi)]).view(np.uint32), pinned)
        few, _ = MultiscaleAlignBatch(ctx, prm, pyr[3:8], pyr[4:9]).align()
        assert np.array_equal(_as_array(few).view(np.uint32), pinned[3:8])
        for p in (5, 40):
            single = MultiscaleAlign.new(ctx, prm, pyr[p]).align(pyr[p + 1])
            assert np.array_equal(np.concatenate([single.t, single.q]).view(np.uint32), pinned[p])
        # and it is the throughput tiling's level-0 cut of this very batch: the poses stay within rounding of it
        assert np.abs(pinned.view(np.float32) - full).max() < 5e-3
    finally:
        ctx.set_tiling(0)
    for p in pyr:
        for lv in p:
            lv.free()


def test_full_size_point_cloud_icp_properties(ctx):
    """configs[2] at its full size (500 000 target x 500 000 source points, 15 iterations): known-motion recovery,
    determinism, and the fixed point of aligning a cloud with itself."""
    from align3d_amd import Icp, PointCloud, RangeImageBuilder

    frames, poses = synth.frame_stream(7, 2, 880, 660)
    cam = synth.camera(880, 660)
    clouds = []
    for d, rgb in frames:
        ri = RangeImageBuilder(ctx).pyramid_levels(1).with_intensity(False).build(cam, d, rgb, synth.DEPTH_SCALE)[0].download(intensity=False)
        pc = PointCloud.from_range_image(ri)
        assert pc.len() >= 500_000
        clouds.append(PointCloud(pc.points[:500_000], pc.normals[:500_000]))
    tgt, src = clouds
    icp = Icp.new(ctx, IcpParams.default(), tgt)
    T = icp.align(src)
    T2 = icp.align(src)
    assert np.array_equal(T.t, T2.t) and np.array_equal(T.q, T2.q)
    rel = synth.relative_pose(poses[0], poses[1])
    dm = np.linalg.inv(rel) @ T.matrix().astype(np.float64)
    ang = np.arccos(np.clip((np.trace(dm[:3, :3]) - 1) / 2, -1, 1))
    ident_ang = np.arccos(np.clip((np.trace(rel[:3, :3]) - 1) / 2, -1, 1))
    assert ang < 0.2 * ident_ang + 1e-4 and np.linalg.norm(dm[:3, 3]) < 0.5 * np.linalg.norm(rel[:3, 3]) + 1e-3
    # configs[2] at its stated size against the ORACLE (src/icp/pcl_icp.rs:49-107 over
    # src/kdtree.rs:28-105; the oracle builds its tree in ~0.15 s and aligns in ~0.5 s): the end pose within the
    # north-star tolerance, and at iterations 0 / 7 / 14 — from the oracle's own transform before that iteration — the
    # inlier count exact (bit-exact neighbours of bit-exact transformed points) and H, g, sum r^2 within 1e-6 of the
    # f64-summed oracle (g, which vanishes at convergence, relative to the size of its terms)
    import ctypes as C

    import oracle_lib as O
    from align3d_amd import Transform
    from align3d_amd._abi import GnStateC, PoseC
    from gpu_util import gn_rel_err, transform_diff

    prm = IcpParams.default()
    tree = O.KdTree(tgt.points)
    assert tree.status == 0
    tv, sv = O.pcl_view(tgt.points, tgt.normals), O.pcl_view(src.points, src.normals)
    pc, out = prm.to_c(), PoseC()
    trace = np.zeros((int(prm.max_iterations), 8), np.float32)
    assert O.load().orc_pcl_icp_align(C.byref(pc), tree.h, C.byref(tv), C.byref(sv), C.byref(out), O.ptr(trace)) == 0
    ang, tr = transform_diff(T, out)
    print(f"[configs[2] 500k x 500k x 15 vs oracle] d_angle={ang:.3e} rad d_trans={tr:.3e} m")
    assert ang <= 1e-4 and tr <= 1e-4
    for it in (0, 7, 14):
        T_in = Transform.eye() if it == 0 else Transform(trace[it - 1, 1:4], trace[it - 1, 4:8])
        g = GnStateC()
        t_in = T_in.to_c()
        assert O.load().orc_pcl_icp_accumulate(C.byref(pc), tree.h, C.byref(tv), C.byref(sv), C.byref(t_in), 1, C.byref(g)) == 0
        ref, gpu = g.as_dict(), icp.accumulate(src, T_in)
        assert gpu["count"] == ref["count"] and ref["count"] > 100_000, (it, gpu["count"], ref["count"])
        eh, eg, es = gn_rel_err(gpu, ref)
        print(f"[configs[2] iteration {it}] inliers {ref['count']}, rel. err H {eh:.1e} g {eg:.1e} ssq {es:.1e}")
        # g vanishes at convergence (4e-6 of its own size is 1e-10 of its terms): its error is measured against the size
        # of what it sums, |g_i| <= sqrt(H_ii * sum r^2) (Cauchy-Schwarz over the inliers), not against max |g|
        Hd = np.sqrt(np.diag(np.asarray(ref["H"], np.float64).reshape(6, 6)) * float(ref["ssq"]))
        eg_terms = float(np.max(np.abs(np.asarray(gpu["g"], np.float64) - np.asarray(ref["g"], np.float64)) / Hd))
        assert eh < 1e-6 and es < 1e-6 and eg_terms < 1e-6 and (eg < 1e-6 or it > 0), (eh, eg, es, eg_terms)
    # a cloud aligned to itself stays where it is: almost every point finds itself in its leaf (the leaf-only
    # search misses the few that tie with a split value), so the residual is ~0 and the estimate the identity
    self_icp = Icp.new(ctx, IcpParams.default(), src)
    same = self_icp.accumulate(src, type(T).eye())
    assert 0.98 * 500_000 < same["count"] <= 500_000 and float(same["ssq"]) / same["count"] < 1e-5
    Ts = self_icp.align(src)
    print(f"[self alignment] t={Ts.t} q={Ts.q}")
    assert np.abs(Ts.t).max() < 1e-3 and np.abs(Ts.q[:3]).max() < 1e-3
    self_icp.free()
    icp.free()
