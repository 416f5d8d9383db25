"""configs[3]-style odometry on the GPU: device-built pyramids, MultiscaleAlign between consecutive frames,
accumulated trajectory against ground truth (examples/src/bin/odometry.rs)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from align3d_amd import (BilateralFilter, MsIcpParams, MultiscaleAlign, RangeImageBuilder, SlamTbDataset,
                         SyntheticDataset, TransformMetrics, run_odometry)
from data_util import GOLDEN
from gpu_util import oracle_pyramid, transform_diff

pytestmark = pytest.mark.gpu


def test_real_consecutive_pairs_follow_ground_truth(ctx):
    """sample1 frames 0 -> 1 and 4 -> 5 (the consecutive pairs among the fixtures): device builder + default
    MsIcpParams; the estimate must beat the identity against the dataset's ground truth and equal the oracle."""
    ds = SlamTbDataset.load(os.path.join(GOLDEN, "rgbd", "sample1"))  # fixture frames 0, 1, 4, 5
    gt = ds.trajectory()
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    for a, b, ids in ((0, 1, (0, 1)), (2, 3, (4, 5))):
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
