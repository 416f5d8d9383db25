"""configs[3]-style odometry on the GPU: device-built pyramids, MultiscaleAlign between consecutive frames,
accumulated trajectory against ground truth (examples/src/bin/odometry.rs)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import (BilateralFilter, MsIcpParams, MultiscaleAlign, RangeImageBuilder, SlamTbDataset,
                         SyntheticDataset, TransformMetrics, run_odometry, run_odometry_batched)
from data_util import GOLDEN
from gpu_util import oracle_pyramid, transform_diff

pytestmark = pytest.mark.gpu


def test_real_consecutive_pairs_follow_ground_truth(ctx):
    """sample1 frames 0 -> 1 and 4 -> 5 (the consecutive pairs among the fixtures): device builder + default
    MsIcpParams; the estimate must beat the identity against the dataset's ground truth and equal the oracle."""
    ds = SlamTbDataset.load(os.path.join(GOLDEN, "rgbd", "sample1"))  # fixture frames 0..19
    gt = ds.trajectory()
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    for a, b, ids in ((0, 1, (0, 1)), (4, 5, (4, 5))):
        target = builder.build_device(*ds.get(a))
        source = builder.build_device(*ds.get(b))
        T = MultiscaleAlign.new(ctx, MsIcpParams.default(), target).align(source)
        rel_gt = gt.get_relative_transform(b, a)
        err = TransformMetrics.new(T, rel_gt)
        ident = TransformMetrics.new(type(T).eye(), rel_gt)
        assert err.total() < ident.total()
        st, T_ref = O.multiscale_align(MsIcpParams.default().to_c_array(), 3, oracle_pyramid("sample1", ids[0]),
                                       oracle_pyramid("sample1", ids[1]), threads=4)
        ang, tr = transform_diff(T, T_ref)
        assert st == 0 and ang <= 1e-4 and tr <= 1e-4


def test_sample1_20_frame_odometry_against_oracle_and_ground_truth(ctx):
    """configs[3] on real data: the 20-frame odometry loop of examples/src/bin/odometry.rs:38-56 (README.md:79-116) on
    the reference's own sample1 sequence (no TUM / IL-RGBD data exists in the image): RangeImageBuilder::default() +
    BilateralFilter::default(), MsIcpParams::default(), frame i-1 the target and frame i the source.  Every one of the
    19 alignments is held to the oracle's (<= 1e-4 rad / 1e-4 m), so is every accumulated pose, and the
    mean trajectory error against the dataset's ground truth is reported and must equal the oracle's."""
    from align3d_amd import Trajectory, Transform, TrajectoryBuilder

    ds = SlamTbDataset.load(os.path.join(GOLDEN, "rgbd", "sample1"))
    n = 20
    assert ds.len() == n
    pred, metrics = run_odometry(ctx, ds)
    assert pred.len() == n
    # the same loop without the prefetch worker, keeping every per-pair transform
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    prm = MsIcpParams.default()
    last = builder.build(*ds.get(0))
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    ref_poses = [Transform.eye()]
    ref_last = O.pose()  # oracle-side TrajectoryBuilder::accumulate (trajectory.rs:164-168): last = T * last
    worst_pair = worst_acc = (0.0, 0.0)
    for i in range(1, n):
        cur = builder.build(*ds.get(i))
        T = MultiscaleAlign.new(ctx, prm, last).align(cur)
        tb.accumulate(T, float(i))
        st, T_ref = O.multiscale_align(prm.to_c_array(), 3, oracle_pyramid("sample1", i - 1), oracle_pyramid("sample1", i),
                                       threads=8)
        assert st == 0
        ang, tr = transform_diff(T, T_ref)
        assert ang <= 1e-4 and tr <= 1e-4, (i, ang, tr)
        worst_pair = (max(worst_pair[0], ang), max(worst_pair[1], tr))
        ref_last = O.compose(T_ref, ref_last)
        ref_poses.append(Transform.from_c(ref_last))
        ang, tr = transform_diff(tb.build()[i], ref_last)
        assert ang <= 1e-4 and tr <= 1e-4, ("accumulated", i, ang, tr)
        worst_acc = (max(worst_acc[0], ang), max(worst_acc[1], tr))
        for lv in last:
            lv.free()
        last = cur
    for lv in last:
        lv.free()
    seq = tb.build()
    for a, b in zip(pred.camera_to_world, seq.camera_to_world):  # the pipelined loop = the sequential loop
        assert np.array_equal(a.t, b.t) and np.array_equal(a.q, b.q)
    gt = ds.trajectory().slice(0, n).first_frame_at_origin()
    still, ref_traj = Trajectory(), Trajectory()
    for i in range(n):
        still.push(Transform.eye(), float(i))
        ref_traj.push(ref_poses[i], float(i))
    m_still = TransformMetrics.mean_trajectory_error(still, gt)
    m_ref = TransformMetrics.mean_trajectory_error(ref_traj, gt)
    print(f"[sample1 20-frame odometry] Mean trajectory error: {metrics} (oracle: {m_ref}; camera held still: {m_still}); "
          f"worst pair vs oracle {worst_pair[0]:.2e} rad {worst_pair[1]:.2e} m; worst accumulated pose vs oracle "
          f"{worst_acc[0]:.2e} rad {worst_acc[1]:.2e} m")
    assert abs(metrics.angle - m_ref.angle) <= 1e-4 and abs(metrics.translation - m_ref.translation) <= 1e-4
    assert metrics.total() < m_still.total()  # the estimate tracks the ground truth better than no motion at all


def test_synthetic_stream_odometry(ctx):
    ds = SyntheticDataset(5, 6)
    pred, metrics = run_odometry(ctx, ds)
    assert pred.len() == 6
    print(f"[synthetic 6-frame odometry] Mean trajectory error: {metrics}")
    # the motion is ~0.36 deg and ~4 mm per frame; the estimate has to stay well inside it
    assert metrics.angle < np.deg2rad(0.25) and metrics.translation < 0.01


def test_tum_formatted_stream_odometry(ctx, tmp_path):
    """The odometry example's path for `--format tum`: the synthetic stream written as a TUM RGB-D directory
    (depth in 1/5000 m, 640x480, the reader's fixed intrinsics do not match the generator's, so the check is
    that the loop runs through reader -> SubsetDataset -> device builder -> MultiscaleAlign and produces the
    same trajectory as feeding the reader's arrays by hand)."""
    from PIL import Image

    from align3d_amd import SubsetDataset, TumRgbdDataset, TrajectoryBuilder, Transform

    src = SyntheticDataset(9, 4)
    (tmp_path / "rgb").mkdir()
    (tmp_path / "depth").mkdir()
    rgb_txt, depth_txt, gt_txt = ["# color"], ["# depth"], ["# timestamp tx ty tz qx qy qz qw"]
    gt = src.trajectory()
    for i in range(4):
        _, d, rgb, _ = src.get(i)
        Image.fromarray(rgb).save(tmp_path / "rgb" / f"{i}.png")
        Image.fromarray((d.astype(np.uint32) * 5).clip(0, 65535).astype(np.uint16)).save(tmp_path / "depth" / f"{i}.png")
        t = 10.0 + i / 30.0
        rgb_txt.append(f"{t:.4f} rgb/{i}.png")
        depth_txt.append(f"{t + 0.004:.4f} depth/{i}.png")
        p = gt[i]
        gt_txt.append(f"{t + 0.001:.4f} " + " ".join(f"{v:.9g}" for v in list(p.t) + list(p.q)))
    (tmp_path / "rgb.txt").write_text("\n".join(rgb_txt) + "\n")
    (tmp_path / "depth.txt").write_text("\n".join(depth_txt) + "\n")
    (tmp_path / "groundtruth.txt").write_text("\n".join(gt_txt) + "\n")
    ds = SubsetDataset.new(TumRgbdDataset.load(str(tmp_path)), range(3))
    assert ds.len() == 3 and ds.trajectory().len() == 3
    pred, metrics = run_odometry(ctx, ds)
    assert pred.len() == 3 and metrics is not None and np.isfinite(metrics.total())
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    pyr = [builder.build_device(*ds.get(i)) for i in range(3)]
    for i in (1, 2):
        tb.accumulate(MultiscaleAlign.new(ctx, MsIcpParams.default(), pyr[i - 1]).align(pyr[i]), float(i))
    for a, b in zip(pred.camera_to_world, tb.build().camera_to_world):
        assert np.array_equal(a.t, b.t) and np.array_equal(a.q, b.q)


def test_batched_odometry_of_a_recorded_sequence_equals_the_frame_by_frame_loop(ctx):
    """run_odometry_batched: the 19 alignments of the 20 sample1 frames as ONE batch (and as windows of 7 pairs that
    overlap by a frame) give the frame-by-frame trajectory — same arithmetic per pair; the batch kernels sum in another
    order (<= 1e-5) — and the same mean trajectory error against the ground truth."""
    ds = SlamTbDataset.load(os.path.join(GOLDEN, "rgbd", "sample1"))
    seq, m_seq = run_odometry(ctx, ds, prefetch=False)
    for window in (64, 7):
        pred, m = run_odometry_batched(ctx, ds, window=window)
        assert pred.len() == seq.len() == 20
        for i, (a, b) in enumerate(zip(pred.camera_to_world, seq.camera_to_world)):
            ang, tr = transform_diff(a, b)
            assert ang <= 1e-5 and tr <= 1e-5, (window, i, ang, tr)
        assert abs(m.angle - m_seq.angle) <= 1e-5 and abs(m.translation - m_seq.translation) <= 1e-5
    # a stream whose camera changes half-way is cut into runs at the change
    syn = SyntheticDataset(5, 6)
    a, _ = run_odometry(ctx, syn, prefetch=False)
    b, _ = run_odometry_batched(ctx, syn, window=2)
    for x, y in zip(a.camera_to_world, b.camera_to_world):
        ang, tr = transform_diff(x, y)
        assert ang <= 1e-5 and tr <= 1e-5


def test_alignments_in_flight_give_the_same_trajectory_bit_for_bit(ctx):
    """run_odometry(in_flight=2, 3): the alignment of frames i -> i + 1 starts while frames i - 1 -> i are still being
    aligned (each on its own aligning context; every alignment starts from Transform::eye(), multiscale.rs:52, so they
    do not depend on each other).  Same pairs, same tiling, same kernels: the trajectory is the sequential loop's, bit
    for bit, on the 20 real sample1 frames and on a synthetic stream."""
    for ds in (SlamTbDataset.load(os.path.join(GOLDEN, "rgbd", "sample1")), SyntheticDataset(11, 9)):
        seq, m_seq = run_odometry(ctx, ds, prefetch=False)
        # (in_flight, prefetch): without prefetch — and on the two-frame stream below — the frames used to be built on the
        # very context lane 0 aligns on, from another thread (advisor r4): they are built on the sibling context now
        for k, prefetch in ((2, True), (3, True), (2, False)):
            pred, m = run_odometry(ctx, ds, in_flight=k, prefetch=prefetch)
            assert pred.len() == seq.len()
            for a, b in zip(pred.camera_to_world, seq.camera_to_world):
                assert np.array_equal(np.asarray(a.t).view(np.uint32), np.asarray(b.t).view(np.uint32))
                assert np.array_equal(np.asarray(a.q).view(np.uint32), np.asarray(b.q).view(np.uint32))
            assert m.angle == m_seq.angle and m.translation == m_seq.translation
    short, _ = run_odometry(ctx, SyntheticDataset(11, 9), max_frames=2, in_flight=2)  # n < 3: no prefetch of its own
    ref2, _ = run_odometry(ctx, SyntheticDataset(11, 9), max_frames=2, prefetch=False)
    assert short.len() == ref2.len() == 2
    assert np.array_equal(np.asarray(short.camera_to_world[1].t).view(np.uint32), np.asarray(ref2.camera_to_world[1].t).view(np.uint32))


def test_masks_derived_from_z_change_nothing(ctx, diag_ctx):
    """Device-built pyramids carry mask == (z != 0), so the alignment kernel skips the two mask bytes per pixel
    (ZMASK, image_icp.hip).  With the bytes read (diagnostics build, A3D_ICP_ZMASK=0), with the same pyramids uploaded
    from host arrays (which never take the short cut), and on frames with large invalid regions: bit-identical poses."""
    from align3d_amd import MultiscaleAlignBatch

    ds = SlamTbDataset.load(os.path.join(GOLDEN, "rgbd", "sample1"))
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    cam, _, _, scale = ds.get(0)
    frames = []
    for i in (0, 1, 4, 5):
        _, depth, rgb, _ = ds.get(i)
        depth = depth.copy()
        depth[(i * 37) % 200:(i * 37) % 200 + 90, 100:400] = 0  # a large hole on top of the data's own 12 % invalid pixels
        frames.append((depth, rgb))
    pyr = builder.build_many(cam, frames, scale)
    prm = MsIcpParams.default()
    pairs = [(0, 1), (2, 3), (1, 2)]

    def run(tp, sp, c=ctx):
        b = MultiscaleAlignBatch(c, prm, tp, sp)
        poses, status = b.align()
        b.free()
        assert not status.any()
        return np.stack([p.matrix() for p in poses])

    tp, sp = [pyr[a] for a, _ in pairs], [pyr[b] for _, b in pairs]
    fast = run(tp, sp)
    # the same frames built by the diagnostics build of the library (its own pyramids): masks derived, then masks read
    dbuilder = RangeImageBuilder(diag_ctx).with_bilateral_filter(BilateralFilter.default())
    dpyr = dbuilder.build_many(cam, frames, scale)
    dtp, dsp = [dpyr[a] for a, _ in pairs], [dpyr[b] for _, b in pairs]
    assert np.array_equal(fast.view(np.uint32), run(dtp, dsp, diag_ctx).view(np.uint32))
    os.environ["A3D_ICP_ZMASK"] = "0"
    try:
        slow = run(dtp, dsp, diag_ctx)
    finally:
        del os.environ["A3D_ICP_ZMASK"]
    assert np.array_equal(fast.view(np.uint32), slow.view(np.uint32))
    # the same pyramids as host RangeImages, uploaded (never flagged): the masks are read
    host = [[lv.download(colors=False) for lv in p] for p in pyr]
    for p in host:
        for lv in p:
            lv._device = None
    up = run([host[a] for a, _ in pairs], [host[b] for _, b in pairs])
    assert np.array_equal(fast.view(np.uint32), up.view(np.uint32))
    # and the flag's premise, pixel by pixel: mask == (z != 0) on every level of every built frame
    for p in host:
        for lv in p:
            assert np.array_equal(lv.mask != 0, lv.points[..., 2] != 0) and set(np.unique(lv.mask)) <= {0, 1}
