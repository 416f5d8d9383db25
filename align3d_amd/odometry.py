"""Frame-to-frame odometry (examples/src/bin/odometry.rs:28-69, README.md:55-111): every frame becomes a
resident pyramid once (device-side RangeImageBuilder), consecutive frames are aligned with
MultiscaleAlign, the transforms are accumulated into a trajectory and compared with the ground truth."""
from .bilateral import BilateralFilter
from .icp import MultiscaleAlign
from .icp_params import MsIcpParams
from .range_image import RangeImageBuilder
from .trajectory import TrajectoryBuilder, TransformMetrics
from .transform import Transform


def run_odometry(ctx, dataset, params=None, builder=None, max_frames=None):
    """Returns (predicted Trajectory, mean TransformMetrics against the dataset's ground truth or None)."""
    params = params or MsIcpParams.default()
    builder = builder or RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    n = dataset.len() if max_frames is None else min(max_frames, dataset.len())
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    last = builder.build_device(*dataset.get(0))
    for i in range(1, n):
        cur = builder.build_device(*dataset.get(i))
        icp = MultiscaleAlign.new(ctx, params, last)  # the previous frame's pyramid is the target
        tb.accumulate(icp.align(cur), float(i))
        icp.free()
        for lv in last:
            lv.free()
        last = cur
    for lv in last:
        lv.free()
    pred = tb.build()
    gt = dataset.trajectory()
    metrics = None
    if gt is not None:
        metrics = TransformMetrics.mean_trajectory_error(pred, gt.slice(0, n).first_frame_at_origin())
    return pred, metrics
