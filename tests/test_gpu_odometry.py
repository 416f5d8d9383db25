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
