"""Frame-to-frame odometry (examples/src/bin/odometry.rs:28-69, README.md:55-111): every frame becomes a
resident pyramid once (device-side RangeImageBuilder), consecutive frames are aligned with
MultiscaleAlign, the transforms are accumulated into a trajectory and compared with the ground truth.

The loop is a two-stage pipeline: while frames i-1 and i are being aligned on the caller's context, frame i+1 is
decoded, uploaded and built into its pyramid by a worker thread on a second context (its own HIP stream and scratch
arena) of the same GPU, so the frame build and its PCIe copy hide under the alignment.  The results are the
same as the sequential loop's: both stages are deterministic and share nothing but finished pyramids."""
import queue
import threading

from .bilateral import BilateralFilter
from .icp import MultiscaleAlign
from .icp_params import MsIcpParams
from .range_image import RangeImageBuilder
from .trajectory import TrajectoryBuilder, TransformMetrics
from .transform import Transform


def _pyramids(dataset, builder, n, side):
    """Yields the device pyramid of frame 0, 1, ... n-1; with a `side` context, built ahead on it by a worker."""
    if side is None:
        for i in range(n):
            yield builder.build_device(*dataset.get(i))
        return
    side_builder = builder.on_context(side)
    q = queue.Queue(maxsize=2)  # at most two finished pyramids wait for the aligner
    stop = threading.Event()

    def work():
        try:
            for i in range(n):
                if stop.is_set():
                    return
                pyr = side_builder.build_device(*dataset.get(i))
                side.synchronize()  # the pyramid is complete before another stream reads it
                q.put(pyr)
            q.put(None)
        except BaseException as e:  # hand the failure to the consumer
            q.put(e)

    t = threading.Thread(target=work, name="a3d-frame-builder", daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        while t.is_alive():  # drain so that a blocked put() can finish, and free what was built ahead
            try:
                item = q.get(timeout=0.05)
                if isinstance(item, list):
                    for lv in item:
                        lv.free()
            except queue.Empty:
                pass
        t.join()


def run_odometry(ctx, dataset, params=None, builder=None, max_frames=None, prefetch=True):
    """Returns (predicted Trajectory, mean TransformMetrics against the dataset's ground truth or None)."""
    params = params or MsIcpParams.default()
    builder = builder or RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    n = dataset.len() if max_frames is None else min(max_frames, dataset.len())
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    last = None
    side = ctx.sibling() if (prefetch and n >= 3) else None
    frames = _pyramids(dataset, builder, n, side)
    try:
        for i, cur in enumerate(frames):
            if last is not None:
                icp = MultiscaleAlign.new(ctx, params, last)  # the previous frame's pyramid is the target
                tb.accumulate(icp.align(cur), float(i))
                icp.free()
                for lv in last:
                    lv.free()
            last = cur
    finally:
        frames.close()
        if last is not None:
            for lv in last:
                lv.free()
    pred = tb.build()
    gt = dataset.trajectory()
    metrics = None
    if gt is not None:
        metrics = TransformMetrics.mean_trajectory_error(pred, gt.slice(0, n).first_frame_at_origin())
    return pred, metrics
