"""GPU parity for ImageIcp / MultiscaleAlign:
  1. per-iteration ("teacher-forced"): from an identical transform, inlier counts are exact and
     H, g, sum r^2 agree with the oracle's f64-summed accumulators to 1e-6 relative;
  2. end-to-end on contractive configurations (color_weight == weight): <= 1e-4 rad, <= 1e-4 m;
  3. non-contractive IcpParams::default(): the per-iteration check holds along the oracle's own
     trajectory; the end-to-end difference is reported, not asserted."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import (A3dError, IcpParams, ImageIcp, InvalidParameter, MsIcpParams, MultiscaleAlign,
                         MultiscaleAlignBatch, Transform)
from gpu_util import gn_rel_err, oracle_frame, oracle_pyramid, small_pose, to_range_image, transform_diff

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4  # rad  (BASELINE.json north_star)
TRANS_TOL = 1e-4  # m
ACC_TOL = 1e-6  # relative, per-iteration accumulators vs f64-summed oracle


def _check_accumulators(ctx, prm, ft, fs, T, diag_ctx=None):
    """`diag_ctx`: also run the exact-arithmetic cross-check kernel, which only the diagnostics build contains."""
    icp = ImageIcp.new(ctx, prm, to_range_image(ft))
    g_gpu, c_gpu = icp.accumulate(to_range_image(fs), T)
    st, g_ref, c_ref = O.image_icp_accumulate(prm.to_c(), ft, fs, T.to_c(), accum_f64=True)
    assert st == 0
    g_ref, c_ref = g_ref.as_dict(), c_ref.as_dict()
    assert g_gpu["count"] == g_ref["count"] and c_gpu["count"] == c_ref["count"]
    assert g_ref["count"] > 1000
    for gpu, ref in ((g_gpu, g_ref), (c_gpu, c_ref)):
        eh, eg, es = gn_rel_err(gpu, ref)
        assert eh < ACC_TOL and eg < ACC_TOL and es < ACC_TOL, (eh, eg, es)
    # The cross-check kernel computes every per-pixel value in the reference's own unfused operations; both kernels add
    # the same pixels in the same order, so their difference is exactly what the product kernel's fused Jacobians cost
    # — a real Jacobian error could not hide in it (measured ~1e-7; an error in one term would show at 1e-3 or more)
    if diag_ctx is not None:
        g_ex, c_ex = ImageIcp.new(diag_ctx, prm, to_range_image(ft)).accumulate_exact(to_range_image(fs), T)
        assert g_ex["count"] == g_ref["count"] and c_ex["count"] == c_ref["count"]
        for ex, ref, gpu in ((g_ex, g_ref, g_gpu), (c_ex, c_ref, c_gpu)):
            eh, eg, es = gn_rel_err(ex, ref)
            assert eh < ACC_TOL and eg < ACC_TOL and es < ACC_TOL, ("exact kernel vs oracle", eh, eg, es)
            fh, fg, fs_ = gn_rel_err(gpu, ex)
            assert fh < 5e-7 and fg < 5e-7 and fs_ < 5e-7, ("fused vs exact kernel", fh, fg, fs_)
    # and close to the f32 accumulation order the reference itself would use
    st, g32, c32 = O.image_icp_accumulate(prm.to_c(), ft, fs, T.to_c(), accum_f64=False)
    eh, eg, es = gn_rel_err(g_gpu, g32.as_dict())
    assert eh < 1e-4 and eg < 1e-3 and es < 1e-4


def _check_merged_accumulator(ctx, prm, ft, fs, T):
    """The opt-in merged accumulation (A3D_ICP_ACCUM=merged: 31 sums from the weighted Jacobians) against
    geom.add_weighted(color, w, cw) of the oracle's f64-summed accumulators (gaussnewton.rs:115-121)."""
    icp = ImageIcp.new(ctx, prm, to_range_image(ft))
    got = icp.accumulate_weighted(to_range_image(fs), T)
    st, g_ref, c_ref = O.image_icp_accumulate(prm.to_c(), ft, fs, T.to_c(), accum_f64=True)
    assert st == 0
    O.load().orc_gn_add_weighted(C.byref(g_ref), C.byref(c_ref), prm.weight, prm.color_weight)
    ref = g_ref.as_dict()
    assert got["count"] == ref["count"]
    eh, eg, es = gn_rel_err(got, ref)
    assert eh < ACC_TOL and eg < ACC_TOL and es < ACC_TOL, (eh, eg, es)


def _oracle_step(prm, ft, fs, T, accum_f64):
    """One reference iteration (image_icp.rs:145-153) from T with the oracle: (residual, new transform)."""
    st, g, c = O.image_icp_accumulate(prm.to_c(), ft, fs, T.to_c(), accum_f64=accum_f64)
    assert st == 0
    lib = O.load()
    lib.orc_gn_add_weighted(C.byref(g), C.byref(c), prm.weight, prm.color_weight)
    residual = lib.orc_gn_mean_squared_residual(C.byref(g))
    upd = (C.c_float * 6)()
    assert lib.orc_gn_solve(C.byref(g), upd) == 1
    new = O.compose(O.exp_se3(np.array(upd[:], np.float32)), T.to_c())
    return residual, Transform.from_c(new)


@pytest.mark.parametrize("sample,tgt,src,bilateral", [("sample1", 0, 5, False), ("sample2", 0, 1, True)])
@pytest.mark.parametrize("which", ["default", "ms"])
def test_per_iteration_accumulators(ctx, diag_ctx, sample, tgt, src, bilateral, which):
    ft, fs = oracle_frame(sample, tgt, bilateral), oracle_frame(sample, src, bilateral)
    prm = IcpParams.default() if which == "default" else MsIcpParams.default()[0]
    _check_accumulators(ctx, prm, ft, fs, Transform.eye(), diag_ctx)
    _check_accumulators(ctx, prm, ft, fs, small_pose(1), diag_ctx)
    _check_merged_accumulator(diag_ctx, prm, ft, fs, Transform.eye())
    _check_merged_accumulator(diag_ctx, prm, ft, fs, small_pose(1))


def test_teacher_forced_along_oracle_trajectory(ctx, diag_ctx):
    # bench10 shape (benches/bench_image_icp.rs): sample1 0 <- 5, IcpParams::default, 10 iterations.
    ft, fs = oracle_frame("sample1", 0), oracle_frame("sample1", 5)
    prm = IcpParams(max_iterations=10)
    st, T_ref, trace = O.image_icp_align(prm.to_c(), ft, fs, threads=4, want_trace=True)
    assert st == 0
    for it in (0, 3, 8):
        T = Transform(trace[it, 1:4], trace[it, 4:8])
        _check_accumulators(ctx, prm, ft, fs, T, diag_ctx)
        _check_merged_accumulator(diag_ctx, prm, ft, fs, T)
    # end-to-end on this non-contractive configuration: reported only
    T_gpu, tr_gpu = ImageIcp.new(ctx, prm, to_range_image(ft)).align(to_range_image(fs), trace=True)
    ang, tr = transform_diff(T_gpu, T_ref)
    print(f"[bench10 end-to-end, non-contractive] d_angle={ang:.3e} rad d_trans={tr:.3e} m")
    # The first update comes from identical inputs.  Against the oracle's f64-summed accumulators it
    # agrees tightly; against the reference-order f32 sums only as well as those sums are accurate
    # (H is ill-conditioned, so 1e-6 relative noise in H moves the update by ~1e-5).
    res64, T64 = _oracle_step(prm, ft, fs, Transform.eye(), accum_f64=True)
    assert abs(tr_gpu[0, 0] - res64) <= 1e-6 * abs(res64)
    assert np.allclose(tr_gpu[0, 1:], np.concatenate([T64.t, T64.q]), rtol=0, atol=2e-6)
    assert np.allclose(tr_gpu[0], trace[0], rtol=0, atol=1e-5)


@pytest.mark.parametrize("sample,tgt,src", [("sample1", 0, 5), ("sample2", 0, 1)])
def test_single_level_end_to_end_contractive(ctx, sample, tgt, src):
    ft, fs = oracle_frame(sample, tgt), oracle_frame(sample, src)
    prm = MsIcpParams.default()[0]
    prm.max_iterations = 10
    st, T_ref, trace = O.image_icp_align(prm.to_c(), ft, fs, threads=4, want_trace=True)
    assert st == 0
    T_gpu, tr_gpu = ImageIcp.new(ctx, prm, to_range_image(ft)).align(to_range_image(fs), trace=True)
    ang, tr = transform_diff(T_gpu, T_ref)
    assert ang <= ROT_TOL and tr <= TRANS_TOL, (ang, tr)
    assert np.allclose(tr_gpu[:, 0], trace[:, 0], rtol=1e-3)


def test_initial_transform_is_used(ctx):
    ft, fs = oracle_frame("sample2", 0), oracle_frame("sample2", 1)
    prm = MsIcpParams.default()[0]
    prm.max_iterations = 2
    init = small_pose(3, rot=0.002, trans=0.002)
    st, T_ref, _ = O.image_icp_align(prm.to_c(), ft, fs, init=init.to_c(), threads=4)
    icp = ImageIcp.new(ctx, prm, to_range_image(ft))
    icp.initial_transform = init
    ang, tr = transform_diff(icp.align(to_range_image(fs)), T_ref)
    assert ang <= ROT_TOL and tr <= TRANS_TOL


@pytest.mark.parametrize("sample,tgt,src", [("sample1", 0, 5), ("sample1", 0, 1), ("sample2", 0, 4)])
def test_multiscale_default_end_to_end(ctx, sample, tgt, src):
    """MsIcpParams::default() (20/20/30 iterations, 3 levels, bilateral on): the README usage."""
    tp, sp = oracle_pyramid(sample, tgt), oracle_pyramid(sample, src)
    prm = MsIcpParams.default()
    st, T_ref = O.multiscale_align(prm.to_c_array(), 3, tp, sp, threads=4)
    assert st == 0
    ms = MultiscaleAlign.new(ctx, prm, [to_range_image(f) for f in tp])
    T_gpu = ms.align([to_range_image(f) for f in sp])
    ang, tr = transform_diff(T_gpu, T_ref)
    print(f"[msdefault {sample} {tgt}<-{src}] d_angle={ang:.3e} rad d_trans={tr:.3e} m")
    assert ang <= ROT_TOL and tr <= TRANS_TOL, (ang, tr)


def test_multiscale_new_rejects_length_mismatch(ctx):
    tp = [to_range_image(f) for f in oracle_pyramid("sample1", 0)]
    with pytest.raises(InvalidParameter) as e:
        MultiscaleAlign.new(ctx, MsIcpParams.repeat(2, IcpParams.default()), tp)
    assert "must be equal" in str(e.value)


def test_multiscale_truncates_short_source_pyramid(ctx):
    # izip! stops at the shortest input before .rev() (multiscale.rs:54-59)
    tp, sp = oracle_pyramid("sample1", 0), oracle_pyramid("sample1", 1)
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 3))
    st, T_ref = O.multiscale_align(prm.to_c_array(), 3, tp, sp[:2], threads=4)
    T_gpu = MultiscaleAlign.new(ctx, prm, [to_range_image(f) for f in tp]).align([to_range_image(f) for f in sp[:2]])
    ang, tr = transform_diff(T_gpu, T_ref)
    assert st == 0 and ang <= ROT_TOL and tr <= TRANS_TOL


def test_missing_fields_are_reported(ctx):
    fr = oracle_frame("sample2", 0)
    full = to_range_image(fr)
    for drop in ("intensity_map", "normals"):
        t = to_range_image(fr)
        setattr(t, drop, None)
        with pytest.raises(A3dError) as e:
            ImageIcp.new(ctx, IcpParams.default(), t).align(full)
        assert e.value.status == 2
    s = to_range_image(fr)
    s.intensities = None
    with pytest.raises(A3dError) as e:
        ImageIcp.new(ctx, IcpParams.default(), full).align(s)
    assert e.value.status == 2


def test_solve_failure_is_a_status_not_a_crash(ctx):
    # a source with no valid pixel: count == 0 -> solve() == None -> the reference panics
    fr = oracle_frame("sample2", 0)
    empty = to_range_image(fr)
    empty.mask = np.zeros_like(empty.mask)
    st, _, _ = O.image_icp_align(IcpParams(max_iterations=2).to_c(), fr, O.Frame(
        fr.points, np.zeros_like(fr.mask), fr.fx, fr.fy, fr.cx, fr.cy, fr.normals, fr.intensities, fr.intensity_map))
    assert st == 3
    with pytest.raises(A3dError) as e:
        ImageIcp.new(ctx, IcpParams(max_iterations=2), to_range_image(fr)).align(empty)
    assert e.value.status == 3


def test_batch_matches_single_pairs(ctx):
    """P pairs in one launch sequence give what P separate MultiscaleAlign calls give."""
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 4))
    pairs = [("sample1", 0, 5), ("sample1", 1, 4), ("sample2", 0, 4), ("sample1", 5, 0)]
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in pairs]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in pairs]
    batch = MultiscaleAlignBatch(ctx, prm, tps, sps)
    d_mats = ctx.malloc(len(pairs) * 64)
    poses, status = batch.align(matrices_device=d_mats)
    assert not status.any()
    mats = ctx.to_host(d_mats, np.zeros((len(pairs), 4, 4), np.float32))
    for k in range(len(pairs)):
        single = MultiscaleAlign.new(ctx, prm, tps[k]).align(sps[k])
        # same kernels, but under the default (throughput) tiling the cut into blocks follows the batch size, so the
        # f32 sums are associated differently: equal to rounding, not bit for bit (pinned tiling: see
        # test_pinned_tiling_makes_a_pair_independent_of_its_batch)
        assert np.allclose(single.t, poses[k].t, atol=2e-6) and np.allclose(single.q, poses[k].q, atol=2e-6)
        assert np.allclose(mats[k], poses[k].matrix(), atol=1e-6)
        s, a, b = pairs[k]
        st, T_ref = O.multiscale_align(prm.to_c_array(), 3, oracle_pyramid(s, a), oracle_pyramid(s, b), threads=4)
        ang, tr = transform_diff(poses[k], T_ref)
        assert ang <= ROT_TOL and tr <= TRANS_TOL
    ctx.free(d_mats)


def test_shared_reciprocal_division_is_ieee_exact(ctx):
    """The kernels' a / z (reciprocal shared between the quotients of a pixel) equals IEEE division bit for
    bit over the operand ranges the path produces and well beyond (z and z^2 of 1 mm .. 1 km, numerators
    1e-6 .. 1e8, both signs, zero numerators)."""
    import ctypes as C
    from align3d_amd import _abi

    rng = np.random.default_rng(42)
    n = 4_000_000
    z = np.exp(rng.uniform(np.log(1e-3), np.log(1e6), n)).astype(np.float32)
    a = (np.exp(rng.uniform(np.log(1e-6), np.log(1e8), n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    a[:1000] = 0.0
    z[1000:2000] *= -1.0
    bad = C.c_uint64(123)
    _abi.check(ctx.lib.a3d_selftest_division(ctx.handle, _abi.ptr(a), _abi.ptr(z), n, C.byref(bad)))
    assert bad.value == 0


def test_cpp_host_mirror_on_gpu():
    """The C++ mirror (include/align3d.hpp) drives the same library: kd-tree KAT, InvalidParameter, ImageIcp."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "host_mirror_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "GPU checks OK" in out.stdout, out.stdout + out.stderr


def test_level_kernel_matches_per_iteration_launches(diag_ctx, monkeypatch):
    """The round-1 persistent form (diagnostics build: one last-block-form launch per pyramid level, blocks wait on a
    per-pair epoch word instead of exiting: A3D_ICP_PERSISTENT=1) computes the same alignment as the default path."""
    ctx = diag_ctx
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 6))
    pairs = [("sample1", 0, 5), ("sample2", 0, 4), ("sample1", 1, 4)]
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in pairs]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in pairs]
    ref_poses, ref_status = MultiscaleAlignBatch(ctx, prm, tps, sps).align()
    monkeypatch.setenv("A3D_ICP_PERSISTENT", "1")
    batch = MultiscaleAlignBatch(ctx, prm, tps, sps)
    monkeypatch.delenv("A3D_ICP_PERSISTENT")
    poses, status = batch.align()
    poses2, _ = batch.align()  # epochs and counters are reset per call
    assert not status.any() and not ref_status.any()
    for a, b, c in zip(poses, ref_poses, poses2):
        assert np.allclose(a.t, b.t, atol=2e-6) and np.allclose(a.q, b.q, atol=2e-6)
        assert np.array_equal(a.t, c.t) and np.array_equal(a.q, c.q)
    # hybrid: only the coarsest level as one launch (A3D_ICP_PERSISTENT_LEVELS is a bit mask over levels)
    monkeypatch.setenv("A3D_ICP_PERSISTENT_LEVELS", "4")
    hybrid = MultiscaleAlignBatch(ctx, prm, tps, sps)
    monkeypatch.delenv("A3D_ICP_PERSISTENT_LEVELS")
    hposes, hstatus = hybrid.align()
    assert not hstatus.any()
    for a, b in zip(hposes, ref_poses):
        assert np.allclose(a.t, b.t, atol=2e-6) and np.allclose(a.q, b.q, atol=2e-6)
    # a pair that fails (empty source) freezes without stalling the others
    empty = [to_range_image(f) for f in oracle_pyramid("sample1", 5)]
    for lv in empty:
        lv.mask = np.zeros_like(lv.mask)
    monkeypatch.setenv("A3D_ICP_PERSISTENT", "1")
    mixed = MultiscaleAlignBatch(ctx, prm, [tps[0], tps[1]], [empty, sps[1]])
    monkeypatch.delenv("A3D_ICP_PERSISTENT")
    p2, st2 = mixed.align()
    assert st2[0] == 3 and st2[1] == 0
    assert np.allclose(p2[1].t, ref_poses[1].t, atol=2e-6)


def test_stream_groups_do_not_change_any_pair(diag_ctx, monkeypatch):
    """A batch of >= 12 pairs runs as three pair groups on three HIP streams.  With the tiling pinned, a pair's
    arithmetic does not depend on which group it is in, so the poses must equal the one-stream run bit for bit —
    on every repeat (the groups drift apart in time, so they must not share any scratch: regression test for
    overlapping block-partial slices when two groups are at different pyramid levels)."""
    prm = MsIcpParams.repeat(3, IcpParams.default())
    base = [("sample1", 0, 1), ("sample1", 1, 4), ("sample1", 4, 5), ("sample2", 0, 1), ("sample2", 1, 4),
            ("sample1", 5, 4), ("sample2", 4, 0)]
    pairs = (base * 3)[:20]
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in pairs]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in pairs]
    ctx = diag_ctx  # (A3D_ICP_STREAMS is a knob of the diagnostics build)
    ctx.set_tiling(24)  # pinned: the cut of a pair into blocks must not follow the stream count

    def run(streams, repeats):
        monkeypatch.setenv("A3D_ICP_STREAMS", str(streams))
        batch = MultiscaleAlignBatch(ctx, prm, tps, sps)
        assert batch.concurrency() == streams
        outs = []
        for _ in range(repeats):
            poses, status = batch.align()
            assert not status.any()
            outs.append(np.array([np.concatenate([t.t, t.q]) for t in poses], np.float32).view(np.uint32))
        batch.free()
        return outs

    try:
        one = run(1, 1)[0]
        many = run(3, 6) + run(2, 3)
        monkeypatch.setenv("A3D_ICP_PERSIST", "0")  # and without the persistent kernel: the same bits again
        many += run(3, 2)
        monkeypatch.delenv("A3D_ICP_PERSIST")
    finally:
        ctx.set_tiling(0)
    for o in many:
        assert np.array_equal(o, one)
    # equal inputs give equal outputs wherever they sit in the batch
    for k in range(7, 20):
        assert np.array_equal(one[k], one[k % 7])


def test_batch_rebind_equals_a_new_batch(ctx):
    """a3d_multiscale_batch_rebind: one batch object pointed at other pyramids gives what a fresh batch gives."""
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 4))
    first = [("sample1", 0, 5), ("sample1", 1, 4), ("sample2", 0, 4)]
    second = [("sample2", 1, 0), ("sample1", 4, 0), ("sample1", 5, 1)]

    def pyr(pairs):
        return ([[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in pairs],
                [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in pairs])

    t1, s1 = pyr(first)
    t2, s2 = pyr(second)
    batch = MultiscaleAlignBatch(ctx, prm, t1, s1)
    batch.align()
    got, status = batch.rebind(t2, s2).align()
    want, wstatus = MultiscaleAlignBatch(ctx, prm, t2, s2).align()
    assert not status.any() and not wstatus.any()
    for a, b in zip(got, want):
        assert np.array_equal(a.t, b.t) and np.array_equal(a.q, b.q)
    back, _ = batch.rebind(t1, s1).align()
    ref, _ = MultiscaleAlignBatch(ctx, prm, t1, s1).align()
    for a, b in zip(back, ref):
        assert np.array_equal(a.t, b.t) and np.array_equal(a.q, b.q)


@pytest.mark.parametrize("w,h", [(37, 29), (64, 48), (257, 3), (300, 199)])
def test_ragged_small_images_per_iteration(ctx, w, h):
    """Image sizes that are not multiples of the block, wave or pipeline-step sizes (the last tile is partly
    empty, a thread's pixel count is odd): counts exact, sums to 1e-6 against the oracle, on a synthetic frame
    pair rendered at that size."""
    from align3d_amd import synth

    frames, _ = synth.frame_stream(31, 2, w, h)
    cam = synth.camera(w, h)
    ft = O.build_frame(frames[0][0], frames[0][1], cam.fx, cam.fy, cam.cx, cam.cy, synth.DEPTH_SCALE)
    fs = O.build_frame(frames[1][0], frames[1][1], cam.fx, cam.fy, cam.cx, cam.cy, synth.DEPTH_SCALE)
    prm = MsIcpParams.default()[0]
    icp = ImageIcp.new(ctx, prm, to_range_image(ft))
    for T in (Transform.eye(), small_pose(3)):
        g_gpu, c_gpu = icp.accumulate(to_range_image(fs), T)
        st, g_ref, c_ref = O.image_icp_accumulate(prm.to_c(), ft, fs, T.to_c(), accum_f64=True)
        assert st == 0
        g_ref, c_ref = g_ref.as_dict(), c_ref.as_dict()
        assert g_gpu["count"] == g_ref["count"] and c_gpu["count"] == c_ref["count"]
        if g_ref["count"] > 20:
            for gpu, ref in ((g_gpu, g_ref), (c_gpu, c_ref)):
                eh, eg, es = gn_rel_err(gpu, ref)
                assert eh < ACC_TOL and eg < ACC_TOL and es < ACC_TOL, (w, h, eh, eg, es)
    # and a few full iterations: same trajectory as the oracle (contractive parameters)
    prm3 = MsIcpParams.default()[0]
    prm3.max_iterations = 3
    if g_ref["count"] > 100 and min(w, h) >= 16:  # a 3-row strip is too ill-conditioned for an end-to-end bound
        T_gpu = ImageIcp.new(ctx, prm3, to_range_image(ft)).align(to_range_image(fs))
        st, T_ref, _ = O.image_icp_align(prm3.to_c(), ft, fs)
        ang, tr = transform_diff(T_gpu, T_ref)
        assert st == 0 and ang <= ROT_TOL and tr <= TRANS_TOL


@pytest.mark.parametrize("knobs", [
    {"A3D_ICP_ACCUM": "mfma"},
    {"A3D_ICP_ACCUM": "merged"},
    {"A3D_ICP_GROUP": "2"},
    {"A3D_ICP_GROUP_LEVELS": "1,2,4"},
    {"A3D_ICP_WAVES": "3"},
    {"A3D_ICP_WAVES": "0.1"},
    {"A3D_ICP_VARIANT": "7,1"},
    {"A3D_ICP_STREAMS": "2", "A3D_ICP_WAVES_LEVELS": "2,1,0.5"},
    {"A3D_ICP_PERSIST": "0"},
    {"A3D_ICP_PERSIST": "7"},
    {"A3D_ICP_PERSIST": "4", "A3D_ICP_STREAMS": "1"},
    {"A3D_ICP_SOLVE": "exact"},
    {"A3D_ICP_HANDOFF": "ticket"},
])
def test_opt_in_kernel_variants_keep_parity(diag_ctx, monkeypatch, knobs):
    """The tuning knobs select other kernels / tilings (MFMA accumulation, 2 or 4 pixels per pipeline step, odd
    pixel counts per thread, more or fewer blocks): none may change what is computed beyond the association of
    the f32 sums."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    ctx = diag_ctx
    ft, fs = oracle_frame("sample1", 0, False), oracle_frame("sample1", 5, False)
    _check_accumulators(ctx, MsIcpParams.default()[0], ft, fs, small_pose(1), diag_ctx)
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 5))
    pairs = [("sample1", 0, 5), ("sample2", 0, 4)] * 7  # 14 pairs: the stream groups are in play
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in pairs]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in pairs]
    poses, status = MultiscaleAlignBatch(ctx, prm, tps, sps).align()
    assert not status.any()
    for k in range(2):
        s, a, b = pairs[k]
        st, T_ref = O.multiscale_align(prm.to_c_array(), 3, oracle_pyramid(s, a), oracle_pyramid(s, b), threads=4)
        ang, tr = transform_diff(poses[k], T_ref)
        assert st == 0 and ang <= ROT_TOL and tr <= TRANS_TOL
    for k in range(2, 14):  # equal inputs, equal outputs wherever the pair sits
        assert np.array_equal(poses[k].t, poses[k % 2].t) and np.array_equal(poses[k].q, poses[k % 2].q)


def test_bench10_lies_inside_the_reference_nondeterminism_envelope(ctx):
    """bench10 (benches/bench_image_icp.rs: sample1 0 <- 5, IcpParams::default(), 10 iterations) is not contractive
    (SURVEY §10.1), and the reference itself is not reproducible on it: rayon's par_bridge() (image_icp.rs:96) hands
    the 75 chunk accumulators to `collect` in arbitrary order and they are then added in that order (:145-148).  The
    oracle replays that freedom (a seeded permutation of the chunk merge order per pass).  The GPU result has to be
    no farther from the oracle's results than those are from each other."""
    ft, fs = oracle_frame("sample1", 0), oracle_frame("sample1", 5)
    prm = IcpParams(max_iterations=10)
    runs = []
    try:
        for seed in range(13):  # seed 0 = chunk order
            O.set_chunk_merge_order(seed)
            st, T, _ = O.image_icp_align(prm.to_c(), ft, fs, threads=4)
            assert st == 0
            runs.append(T)
    finally:
        O.set_chunk_merge_order(0)
    pair = np.array([[O.transform_metrics(a, b) for b in runs] for a in runs])
    spread_ang, spread_tr = float(pair[..., 0].max()), float(pair[..., 1].max())
    T_gpu = ImageIcp.new(ctx, prm, to_range_image(ft)).align(to_range_image(fs))
    to_gpu = np.array([transform_diff(T_gpu, r) for r in runs])
    print(f"[bench10 envelope] oracle vs oracle over 13 merge orders: max {spread_ang:.3e} rad {spread_tr:.3e} m, "
          f"median {np.median(pair[..., 0]):.3e} rad;  GPU vs oracle runs: min {to_gpu[:, 0].min():.3e} rad "
          f"{to_gpu[:, 1].min():.3e} m, max {to_gpu[:, 0].max():.3e} rad {to_gpu[:, 1].max():.3e} m")
    assert spread_ang > ROT_TOL  # the reference's own envelope is wider than the north-star tolerance on this shape
    assert to_gpu[:, 0].min() <= spread_ang and to_gpu[:, 1].min() <= spread_tr
    assert to_gpu[:, 0].max() <= 2 * spread_ang and to_gpu[:, 1].max() <= 2 * spread_tr


def test_ms3x15_end_to_end_on_benchmark_pairs(ctx):
    """The headline workload itself (bench.py: seed-1000 synthetic 640x480 stream, device-built pyramids,
    MsIcpParams::repeat(3, IcpParams::default()) = ms3x15): the batched GPU poses against the oracle run on the very
    arrays the kernels read, <= 1e-4 rad / 1e-4 m (the synthetic texture keeps the default parameters contractive)."""
    import bench

    P = 6
    pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, P + 1, 640, 480)
    prm = MsIcpParams.repeat(3, IcpParams.default())
    batch = MultiscaleAlignBatch(ctx, prm, pyr[:P], pyr[1:])
    poses, status = batch.align()
    batch.free()
    assert not np.any(status)

    def frame(dev_level):
        ri = dev_level.download(colors=False)
        k = ri.intrinsics
        return O.Frame(ri.points, ri.mask, k.fx, k.fy, k.cx, k.cy, ri.normals, ri.intensities, ri.intensity_map)

    host = [[frame(lv) for lv in p] for p in pyr]
    worst = (0.0, 0.0)
    for p in range(P):
        st, T_ref = O.multiscale_align(prm.to_c_array(), 3, host[p], host[p + 1], threads=8)
        ang, tr = transform_diff(poses[p], T_ref)
        worst = (max(worst[0], ang), max(worst[1], tr))
        assert st == 0 and ang <= ROT_TOL and tr <= TRANS_TOL, (p, ang, tr)
    print(f"[ms3x15 on the benchmark's pairs] worst d_angle={worst[0]:.3e} rad d_trans={worst[1]:.3e} m")
    for lv in (lv for p in pyr for lv in p):
        lv.free()


# (the headline's own 64 distinct pairs against the oracle and its 13-order envelope: tests/test_gpu_headline_parity.py)


def test_images_may_be_freed_right_after_an_enqueue_only_align():
    """Lifetime rule of include/align3d_hip.h: a batch aligned WITHOUT host outputs only enqueues; freeing its images
    right away (they live on a builder context, the batch on another: two streams) and rebuilding other frames into
    the recycled arenas must not disturb the alignments still in flight — a3d_range_image_free waits for them."""
    from align3d_amd import BilateralFilter, Context, RangeImageBuilder, synth

    main, side = Context(0), Context(0)
    try:
        P = 12
        frames, _ = synth.frame_stream(321, P + 1, 640, 480)
        other, _ = synth.frame_stream(999, P + 1, 640, 480)
        cam = synth.camera(640, 480)
        b = RangeImageBuilder(side).with_bilateral_filter(BilateralFilter.default())
        prm = MsIcpParams.repeat(3, IcpParams.default())
        warm = b.build_many(cam, frames, synth.DEPTH_SCALE)   # fills the arena pool (slabs appear on the 4th request)
        for lv in (lv for p in warm for lv in p):
            lv.free()
        pyr = b.build_many(cam, frames, synth.DEPTH_SCALE)
        batch = MultiscaleAlignBatch(main, prm, pyr[:P], pyr[1:])
        d_mats = main.malloc(P * 64)
        _, status = batch.align(matrices_device=d_mats)      # host-synchronous reference result
        assert not np.any(status)
        want = main.to_host(d_mats, np.zeros((P, 16), np.float32)).copy()
        for trial in range(3):
            batch.enqueue(matrices_device=d_mats)             # enqueue only: ~3 ms of kernels now in flight
            for lv in (lv for p in pyr for lv in p):
                lv.free()                                     # must wait for the batch, then recycle the arenas
            junk = b.build_many(cam, other, synth.DEPTH_SCALE)  # lands in the arenas just released
            main.synchronize()
            got = main.to_host(d_mats, np.zeros((P, 16), np.float32))
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), trial
            for lv in (lv for q in junk for lv in q):
                lv.free()
            pyr = b.build_many(cam, frames, synth.DEPTH_SCALE)
            batch.rebind(pyr[:P], pyr[1:])
        main.free(d_mats)
        batch.free()
        for lv in (lv for p in pyr for lv in p):
            lv.free()
    finally:
        side.close()
        main.close()


def test_multi_device_batch_equals_the_single_device_batch():
    """a3d_multiscale_batch_new_multi on the one-GPU box: the device list [0] gives bit for bit the poses of
    a3d_multiscale_batch_new; the lists [0, 0] and [0, 0, 0] (several contexts = streams on the same GPU) exercise
    the block partition and the per-device host threads (the gather between contexts of ONE device is not the
    cross-device copy: that is the next test); a parameter list of the wrong length is rejected like MultiscaleAlign::new."""
    _multi_device_batch_check([[0], [0, 0], [0, 0, 0]], P=10)


def test_multi_device_batch_over_every_visible_device():
    """The same over the device list the box REPORTS (a3d_device_count): contexts on different GPUs, every device's block
    built and aligned on its own GPU, the 4x4 poses gathered onto device 0 by hipMemcpyPeerAsync — the copy that a
    list of zeros never makes.  Skipped, not faked, where only one device is visible (this build's pool: one GPU)."""
    from align3d_amd import device_count

    n = device_count()
    if n < 2:
        pytest.skip(f"{n} device visible: the cross-device gather cannot run here")
    ids = list(range(min(n, 8)))
    _multi_device_batch_check([ids, ids[::-1]], P=2 * len(ids) + 1)


def _multi_device_batch_check(id_lists, P):
    from align3d_amd import BilateralFilter, Context, MultiContext, MultiscaleAlignMultiBatch, RangeImageBuilder, synth
    from align3d_amd.multi import shard_range

    frames, _ = synth.frame_stream(77, P + 1, 320, 240)
    cam = synth.camera(320, 240)
    # contractive parameters: differences of f32 association between batch shapes are not amplified
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 6))
    single = Context(0)
    try:
        b = RangeImageBuilder(single).with_bilateral_filter(BilateralFilter.default())
        pyr = b.build_many(cam, frames, synth.DEPTH_SCALE)
        batch = MultiscaleAlignBatch(single, prm, pyr[:P], pyr[1:])
        want, want_status = batch.align()
        batch.free()
        want_bits = np.array([np.concatenate([t.t, t.q]) for t in want], np.float32).view(np.uint32)
        for ids in id_lists:
            mc = MultiContext(ids)
            assert len(mc) == len(ids)
            tp, sp = [], []
            for d in range(len(ids)):  # each entry builds the frames of the pairs it owns on its own context
                lo, hi = shard_range(P, len(ids), d)
                if lo == hi:
                    continue
                own = RangeImageBuilder(mc.device(d)).with_bilateral_filter(BilateralFilter.default()).build_many(
                    cam, frames[lo:hi + 1], synth.DEPTH_SCALE)
                tp += own[:-1]
                sp += own[1:]
            mb = MultiscaleAlignMultiBatch(mc, prm, tp, sp)
            got, status, mats = mb.align()
            got_bits = np.array([np.concatenate([t.t, t.q]) for t in got], np.float32).view(np.uint32)
            assert np.array_equal(status, want_status)
            # bit for bit the single-device batch of the same shape: a block's tiling (and with it the association of
            # the f32 block partials) depends on how many pairs the block holds, so every block is compared with the
            # single-device batch of exactly its pairs; across shapes the poses agree far inside the 1e-4 tolerance
            for d in range(len(ids)):
                lo, hi = shard_range(P, len(ids), d)
                if lo == hi:
                    continue
                blk = MultiscaleAlignBatch(single, prm, pyr[lo:hi], pyr[lo + 1:hi + 1])
                ref, _ = blk.align()
                blk.free()
                ref_bits = np.array([np.concatenate([t.t, t.q]) for t in ref], np.float32).view(np.uint32)
                assert np.array_equal(got_bits[lo:hi], ref_bits), (ids, d)
            for p in range(P):
                ang, tr = transform_diff(got[p], want[p])
                assert ang <= 1e-5 and tr <= 1e-5, (ids, p, ang, tr)
            for p in range(P):  # the gathered 4x4 matrices are these poses, in global pair order
                assert np.allclose(mats[p].reshape(4, 4), got[p].matrix(), atol=1e-6), (ids, p)
            again = mb.align()[2]
            assert np.array_equal(mats.view(np.uint32), again.view(np.uint32))
            with pytest.raises(InvalidParameter):
                MultiscaleAlignMultiBatch(mc, MsIcpParams.repeat(2, IcpParams.default()), tp, sp)
            mb.free()
            mc.close()
    finally:
        single.close()


CONTENTION_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["A3D_ROOT"])
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids
ctx = Context(0)
P = 24
pyr, _, _ = build_stream_pyramids(ctx, 1000, P + 1, 640, 480)
b = MultiscaleAlignBatch(ctx, MsIcpParams.repeat(3, IcpParams.default()), pyr[:P], pyr[1:])
ref, bad = None, 0
for it in range(40):
    poses, status = b.align()
    o = np.array([np.concatenate([t.t, t.q]) for t in poses], np.float32).view(np.uint32)
    if ref is None:
        ref = o
    elif not np.array_equal(ref, o):
        bad += 1
np.save(os.environ["A3D_OUT"], ref)
print(f"pid {os.getpid()} streams={b.concurrency()} differing_repeats={bad}", flush=True)
sys.exit(1 if bad else 0)
"""


def test_batch_is_repeat_identical_with_a_second_process_competing_for_the_gpu(tmp_path):
    """The cross-XCD hand-off of the last-block solve (write-through partials, drained stores, one relaxed ticket, no
    release / acquire fence: DESIGN.md §4) is only as safe as its tests: two fresh processes (started, never re-exec'ed)
    run the 3-stream-group batch 40 times each AT THE SAME TIME, so that blocks of one pair are descheduled and
    spread over the XCDs differently from run to run.  Every repeat must reproduce the first bit for bit, and both
    processes must agree (a race on the partials or the ticket shows up as a pose that differs in the last bits)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(CONTENTION_WORKER)
    procs = []
    for k in range(2):
        env = dict(os.environ, A3D_ROOT=root, A3D_OUT=str(tmp_path / f"poses{k}.npy"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
        assert "differing_repeats=0" in out and "streams=3" in out, out
    a, b = np.load(tmp_path / "poses0.npy"), np.load(tmp_path / "poses1.npy")
    assert np.array_equal(a, b)


def test_two_batches_alternating_over_a_stream_of_rounds():
    """a3d_multiscale_batch_results: round r is rebound and enqueued on one batch while round r-1 is still computing
    on the other (same context), then round r-1 is read and its frames freed.  Every round must give exactly what a
    host-synchronous align of the same frames gives; results() must not be disturbed by the pass enqueued after it."""
    from align3d_amd import BilateralFilter, Context, RangeImageBuilder, synth

    main, side = Context(0), Context(0)
    try:
        P, R = 10, 5
        cam = synth.camera(640, 480)
        bld = RangeImageBuilder(side).with_bilateral_filter(BilateralFilter.default())
        prm = MsIcpParams.repeat(3, IcpParams.default())
        streams = [synth.frame_stream(700 + r, P + 1, 640, 480)[0] for r in range(R)]
        want = []
        for fr in streams:  # the reference results, one synchronous batch per round
            pyr = bld.build_many(cam, fr, synth.DEPTH_SCALE)
            b = MultiscaleAlignBatch(main, prm, pyr[:P], pyr[1:])
            poses, status = b.align()
            assert not np.any(status)
            want.append(np.array([list(p.t) + list(p.q) for p in poses], np.float32))
            b.free()
            for lv in (lv for p in pyr for lv in p):
                lv.free()
        first = bld.build_many(cam, streams[0], synth.DEPTH_SCALE)
        batches = [MultiscaleAlignBatch(main, prm, first[:P], first[1:]) for _ in range(2)]
        with pytest.raises(A3dError):  # nothing enqueued yet: there are no results to read
            batches[0].results()
        for lv in (lv for p in first for lv in p):
            lv.free()
        prev, got = None, []
        for r in range(R):
            pyr = bld.build_many(cam, streams[r], synth.DEPTH_SCALE)
            b = batches[r % 2]
            b.rebind(pyr[:P], pyr[1:])
            b.enqueue()
            if prev is not None:
                poses, status = prev[0].results()
                assert not np.any(status)
                got.append(np.array([list(p.t) + list(p.q) for p in poses], np.float32))
                for lv in (lv for p in prev[1] for lv in p):
                    lv.free()
            prev = (b, pyr)
        poses, status = prev[0].results()
        got.append(np.array([list(p.t) + list(p.q) for p in poses], np.float32))
        for lv in (lv for p in prev[1] for lv in p):
            lv.free()
        for r in range(R):
            assert np.array_equal(got[r].view(np.uint32), want[r].view(np.uint32)), r
        for b in batches:
            b.free()
    finally:
        side.close()
        main.close()


@pytest.mark.gpu
def test_pyramid_upload_shares_one_arena_and_changes_nothing(ctx):
    """a3d_range_image_upload_pyramid (the drop-in call's upload: one pooled arena for all levels) against level-by-
    level uploads and against an alignment of the same pyramid resident on the device: bit-identical poses; arrays
    read back equal what went up; images uploaded without normals can still compute them."""
    from align3d_amd import DeviceRangeImage
    from align3d_amd.range_image import upload_pyramid

    tp, sp = oracle_pyramid("sample1", 0), oracle_pyramid("sample1", 1)
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 4))
    t_host, s_host = [to_range_image(f) for f in tp], [to_range_image(f) for f in sp]
    t_one = [DeviceRangeImage(ctx, im) for im in t_host]           # one upload call per level
    s_one = [DeviceRangeImage(ctx, im) for im in s_host]
    T_one = MultiscaleAlign.new(ctx, prm, t_one).align(s_one)
    # a host source pyramid: a3d_multiscale_align_host — uploaded on the copy stream coarsest level first, each level's
    # launches behind that level's arrays only, nothing left resident (round 5) ...
    ms_host = MultiscaleAlign.new(ctx, prm, t_host)
    T_pyr = ms_host.align(s_host)
    assert np.array_equal(T_one.matrix(), T_pyr.matrix())
    assert all(h._device is None for h in s_host)
    for _ in range(4):  # ... call after call (the arena and the events of one call are gone before the next)
        assert np.array_equal(ms_host.align(s_host).matrix(), T_one.matrix())
    short = ms_host.align(s_host[:2])  # a shorter source pyramid truncates like izip! (multiscale.rs:54-59)
    assert np.array_equal(short.matrix(), MultiscaleAlign.new(ctx, prm, t_one).align(s_one[:2]).matrix())
    no_int = to_range_image(sp[0])
    no_int.intensities = None
    with pytest.raises(A3dError) as e:  # the reference's expect() on the source (image_icp.rs:52-57), before anything is copied
        ms_host.align([no_int] + s_host[1:])
    assert e.value.status == 2
    assert np.array_equal(ms_host.align(s_host).matrix(), T_one.matrix())  # (and the context is fine afterwards)
    # ... and the two-call form (a3d_range_image_upload_pyramid, then align on the resident images)
    upload_pyramid(ctx, s_host)
    assert np.array_equal(MultiscaleAlign.new(ctx, prm, t_one).align(s_host).matrix(), T_one.matrix())
    for h, f in zip(s_host, sp):
        back = h._device.download(colors=False)
        assert np.array_equal(back.points.view(np.uint32), f.points.view(np.uint32)) and np.array_equal(back.mask, f.mask)
        assert np.array_equal(back.normals.view(np.uint32), f.normals.view(np.uint32))
        assert np.array_equal(back.intensity_map.view(np.uint32), f.intensity_map.view(np.uint32))
    # many upload / free cycles: the arenas go back to the pool and come out again with the same contents
    for _ in range(6):
        for h in s_host:
            h._device.free()
            h._device = None
        devs = upload_pyramid(ctx, s_host)
        assert np.array_equal(MultiscaleAlign.new(ctx, prm, t_one).align(devs).matrix(), T_one.matrix())
    # an image that went up without normals gets them from compute_normals (its own allocation inside an arena image)
    bare = to_range_image(tp[0])
    bare.normals = None
    dev = upload_pyramid(ctx, [bare, to_range_image(tp[1])])[0]
    dev.compute_normals()
    assert np.array_equal(dev.download_normals().view(np.uint32), tp[0].normals.view(np.uint32))
    dev.free()


def test_head_solve_and_last_block_handoff_give_the_same_bits(diag_ctx, monkeypatch):
    """The two hand-off forms (icp_engine.hpp) add the same partials in the same order and run the same solve: a batch
    of uploaded pyramids (masks read) and a batch of device-built ones (masks derived from z) must give bit-identical
    poses with the default head-solve form and with A3D_ICP_HANDOFF=ticket; a single pair and a trace likewise."""
    import bench
    from align3d_amd import synth

    ctx = diag_ctx  # (the last-block form exists in the diagnostics build only)
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 7))
    pairs = [("sample1", 0, 5), ("sample2", 0, 4), ("sample1", 1, 4), ("sample1", 4, 5)]
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in pairs]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in pairs]
    built, _, _ = bench.build_stream_pyramids(ctx, 77, 9, 640, 480)

    def run_all():
        out = []
        b = MultiscaleAlignBatch(ctx, prm, tps, sps)
        poses, status = b.align()
        b.free()
        assert not status.any()
        out.append(np.stack([p.matrix() for p in poses]))
        b = MultiscaleAlignBatch(ctx, MsIcpParams.repeat(3, IcpParams.default()), built[:8], built[1:])
        poses, status = b.align()
        b.free()
        assert not status.any()
        out.append(np.stack([p.matrix() for p in poses]))
        out.append(MultiscaleAlign.new(ctx, prm, tps[0]).align(sps[0]).matrix())
        T, tr = ImageIcp.new(ctx, prm[0], tps[1][0]).align(sps[1][0], trace=True)
        out.append(tr.copy())
        return out

    head = run_all()
    monkeypatch.setenv("A3D_ICP_HANDOFF", "ticket")
    ticket = run_all()
    monkeypatch.delenv("A3D_ICP_HANDOFF")
    monkeypatch.setenv("A3D_ICP_PERSIST", "0")  # head-solve form, one launch per iteration (no persistent kernel)
    per_iteration = run_all()
    monkeypatch.delenv("A3D_ICP_PERSIST")
    for h, t, q in zip(head, ticket, per_iteration):
        assert np.array_equal(h.view(np.uint32), t.view(np.uint32))
        assert np.array_equal(h.view(np.uint32), q.view(np.uint32))
    for lv in (lv for p in built for lv in p):
        lv.free()


def _pose_bits(poses):
    return np.array([np.concatenate([t.t, t.q]) for t in poses], np.float32).view(np.uint32)


def test_pinned_tiling_makes_a_pair_independent_of_its_batch(ctx):
    """a3d_context_set_tiling(n): every (pair, level) is cut into n blocks from the pair's OWN size, so a pair's pose is
    the same bits alone, at any position of any batch, and next to images of another size.  Uses
    the non-contractive IcpParams::default() on purpose: there a re-associated sum moves the pose visibly."""
    from align3d_amd import synth

    prm = MsIcpParams.repeat(3, IcpParams.default()).customize(lambda i, p: setattr(p, "max_iterations", 6))
    base = [("sample1", 0, 5), ("sample2", 0, 4), ("sample1", 1, 4), ("sample1", 4, 5), ("sample2", 1, 0)]
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in base]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in base]
    # a pair of another size (320x240 synthetic frames, 3 levels): mixed batches must not change anybody
    frames, _ = synth.frame_stream(5, 2, 320, 240)
    cam = synth.camera(320, 240)
    small = [[to_range_image(f) for f in O.build_pyramid(d, rgb, cam.fx, cam.fy, cam.cx, cam.cy, synth.DEPTH_SCALE)]
             for d, rgb in frames]
    # ... and 512x384: tiles = 24 cuts its level 1 / level 2 into 24 / 24 blocks where 640x480 takes 22 / 19 (pixels per
    # thread are rounded up to even), so the grid must follow the largest OWN cut, not the largest image (advisor r4)
    frames, _ = synth.frame_stream(6, 2, 512, 384)
    cam = synth.camera(512, 384)
    mid = [[to_range_image(f) for f in O.build_pyramid(d, rgb, cam.fx, cam.fy, cam.cx, cam.cy, synth.DEPTH_SCALE)]
           for d, rgb in frames]
    for tiles in (8, 24):
        ctx.set_tiling(tiles)
        try:
            alone = [MultiscaleAlign.new(ctx, prm, tps[k]).align(sps[k]) for k in range(len(base))]
            alone_small = MultiscaleAlign.new(ctx, prm, small[0]).align(small[1])
            alone_mid = MultiscaleAlign.new(ctx, prm, mid[0]).align(mid[1])
            for t3, s3, at in (([tps[0], mid[0], small[0]], [sps[0], mid[1], small[1]], 1),
                               ([mid[0], tps[1]], [mid[1], sps[1]], 0)):
                got3, st = MultiscaleAlignBatch(ctx, prm, t3, s3).align()
                assert not st.any()
                assert np.array_equal(_pose_bits(got3)[at], _pose_bits([alone_mid])[0]), (tiles, "512x384 pair in a mixed batch")
                assert np.array_equal(_pose_bits(got3)[1 - at], _pose_bits([alone[1 - at]])[0])
            ref = _pose_bits(alone)
            b5, st = MultiscaleAlignBatch(ctx, prm, tps, sps).align()
            assert not st.any() and np.array_equal(_pose_bits(b5), ref)
            # 17 pairs (three stream groups), rotated order, the small pair in the middle
            order = [(k * 3 + 1) % 5 for k in range(16)]
            t17 = [tps[k] for k in order[:8]] + [small[0]] + [tps[k] for k in order[8:]]
            s17 = [sps[k] for k in order[:8]] + [small[1]] + [sps[k] for k in order[8:]]
            b17, st = MultiscaleAlignBatch(ctx, prm, t17, s17).align()
            assert not st.any()
            got = _pose_bits(b17)
            assert np.array_equal(got[8], _pose_bits([alone_small])[0])
            assert np.array_equal(np.delete(got, 8, axis=0), ref[order])
        finally:
            ctx.set_tiling(0)
    # under the default tiling the same comparison only holds to rounding (that is what the mode is for)
    alone = MultiscaleAlign.new(ctx, prm, tps[0]).align(sps[0])
    b5, _ = MultiscaleAlignBatch(ctx, prm, tps, sps).align()
    print("[tiling] throughput tiling, pair alone vs in a batch of 5: max |d| =",
          float(np.abs(np.concatenate([alone.t, alone.q]) - np.concatenate([b5[0].t, b5[0].q])).max()))


def test_persistent_kernel_and_per_iteration_launches_give_the_same_bits(ctx, diag_ctx, monkeypatch):
    """The persistent head-solve kernel (diagnostics build, A3D_ICP_PERSIST=mask: many iterations in one launch, one-hop
    counter hand-off, every block sums the pair's partials itself; measured slower than kernel boundaries)
    adds the same partials in the same order and runs the same solve as the per-iteration launches: under one tiling both
    give the same bits — for a lone pair (all levels persistent, outputs written by the kernel), a trace, a small batch
    and a three-group batch (coarse levels persistent) — and so does the product library."""
    prm = MsIcpParams.default().customize(lambda i, p: setattr(p, "max_iterations", 7))
    base = [("sample1", 0, 5), ("sample2", 0, 4), ("sample1", 1, 4), ("sample1", 4, 5)]
    tps = [[to_range_image(f) for f in oracle_pyramid(s, a)] for s, a, b in base]
    sps = [[to_range_image(f) for f in oracle_pyramid(s, b)] for s, a, b in base]
    t14, s14 = (tps * 4)[:14], (sps * 4)[:14]

    def run_all(c, mask14):
        out = []
        out.append(MultiscaleAlign.new(c, prm, tps[0]).align(sps[0]).matrix())
        T, tr = ImageIcp.new(c, prm[0], tps[1][0]).align(sps[1][0], trace=True)
        out.append(tr.copy())
        for t, s, mask in ((tps, sps, None), (t14, s14, mask14)):
            if mask is not None:
                monkeypatch.setenv("A3D_ICP_PERSIST", mask)
            b = MultiscaleAlignBatch(c, prm, t, s)
            poses, status = b.align()
            again, _ = b.align()  # the pairs' counters keep counting across alignments of one batch
            masks = b.persistent_levels()
            b.free()
            assert not status.any() and np.array_equal(_pose_bits(poses), _pose_bits(again))
            out.append(_pose_bits(poses))
            out.append(np.array([masks]))
        return out

    for c in (ctx, diag_ctx):
        c.set_tiling(12)
    try:
        product = run_all(ctx, None)
        monkeypatch.setenv("A3D_ICP_PERSIST", "7")
        persistent = run_all(diag_ctx, "6")
        monkeypatch.setenv("A3D_ICP_PERSIST", "0")
        per_iteration = run_all(diag_ctx, "0")
        monkeypatch.delenv("A3D_ICP_PERSIST")
    finally:
        for c in (ctx, diag_ctx):
            c.set_tiling(0)
    assert persistent[3][0] == 0b111 and persistent[5][0] == 0b110
    assert product[3][0] == 0 and product[5][0] == 0 and per_iteration[3][0] == 0 and per_iteration[5][0] == 0
    for k in (0, 1, 2, 4):
        assert np.array_equal(product[k].view(np.uint32), per_iteration[k].view(np.uint32)), k
        assert np.array_equal(product[k].view(np.uint32), persistent[k].view(np.uint32)), k


def test_rsqrt_solve_and_ieee_solve_round_to_the_same_update(diag_ctx, monkeypatch):
    """The 6x6 solve multiplies by a refined 1/sqrt(pivot) instead of nalgebra's sqrt and divisions (icp_engine.hpp);
    A3D_ICP_SOLVE=exact (diagnostics build) runs the IEEE sqrt / divide / unfused form.  From identical states both must
    round to the same f32 update: one-iteration alignments from several initial poses on the golden pairs."""
    ctx = diag_ctx
    cases = [("sample1", 0, 5, IcpParams.default()), ("sample2", 0, 1, IcpParams.default()),
             ("sample1", 0, 1, MsIcpParams.default()[0]), ("sample2", 0, 4, MsIcpParams.default()[0])]
    same = total = 0
    for s, a, b, prm in cases:
        prm.max_iterations = 1
        ft, fs = to_range_image(oracle_frame(s, a)), to_range_image(oracle_frame(s, b))
        for seed in range(6):
            icp = ImageIcp.new(ctx, prm, ft)
            icp.initial_transform = small_pose(seed) if seed else Transform.eye()
            fast = icp.align(fs)
            monkeypatch.setenv("A3D_ICP_SOLVE", "exact")
            icp2 = ImageIcp.new(ctx, prm, ft)
            icp2.initial_transform = icp.initial_transform
            exact = icp2.align(fs)
            monkeypatch.delenv("A3D_ICP_SOLVE")
            x, y = np.concatenate([fast.t, fast.q]), np.concatenate([exact.t, exact.q])
            total += 1
            same += bool(np.array_equal(x.view(np.uint32), y.view(np.uint32)))
            assert np.abs(x - y).max() <= 2.4e-7, (s, a, b, seed, x, y)  # at most last-bit ties of the f32 update
    print(f"[solve] rsqrt vs IEEE solve: {same} of {total} one-iteration poses bit-identical")
    assert same >= total - 3


def test_context_destroyed_before_its_images_defers_to_the_last_image():
    """a3d_context_destroy with images alive must not free their arenas; the context goes when the
    last image is freed (C and Rust callers are not protected by Python's handle check)."""
    import ctypes as C
    from align3d_amd import Context, _abi

    c = Context(0, pair=False)
    fr = oracle_pyramid("sample1", 0)
    devs = [to_range_image(f).device(c) for f in fr]
    handles = [d.handle for d in devs]
    lib, ch = c.lib, c.handle
    assert lib.a3d_context_destroy(ch) == 0  # deferred: the images keep it alive
    c.handle = C.c_void_p()                  # (Python's own guard would skip the frees below)
    # a zombie context refuses new work that needs an arena ...
    v = to_range_image(fr[0]).view()
    h = C.c_void_p()
    assert lib.a3d_range_image_upload(ch, C.byref(v), C.byref(h)) != 0
    # ... its images are still readable and freeable; the last free tears the context down
    w, hh = C.c_uint64(), C.c_uint64()
    assert lib.a3d_range_image_size(handles[0], C.byref(w), C.byref(hh)) == 0 and (w.value, hh.value) == (640, 480)
    for d, hd in zip(devs, handles):
        assert lib.a3d_range_image_free(hd) == 0
        d.handle = C.c_void_p()
