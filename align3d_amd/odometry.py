"""Frame-to-frame odometry (examples/src/bin/odometry.rs:28-69, README.md:55-111): every frame becomes a
resident pyramid once (device-side RangeImageBuilder), consecutive frames are aligned with
MultiscaleAlign, the transforms are accumulated into a trajectory and compared with the ground truth.

The loop is a two-stage pipeline: while frames i-1 and i are being aligned on the caller's context, frame i+1 is
decoded, uploaded and built into its pyramid by a worker thread on a second context (its own HIP stream and scratch
arena) of the same GPU, so the frame build and its PCIe copy hide under the alignment.  The results are the
same as the sequential loop's: both stages are deterministic and share nothing but finished pyramids."""
import queue
import threading

from . import _abi
from .bilateral import BilateralFilter
from .icp import MultiscaleAlign, MultiscaleAlignBatch
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


def run_odometry(ctx, dataset, params=None, builder=None, max_frames=None, prefetch=True, in_flight=1):
    """Returns (predicted Trajectory, mean TransformMetrics against the dataset's ground truth or None).

    `in_flight` > 1: frame-to-frame alignments do not depend on each other — MultiscaleAlign::align starts every pair
    from Transform::eye() (multiscale.rs:52) and only the trajectory is a running product (trajectory.rs:164-168) — so
    the alignment of frames i -> i + 1 may start as soon as frame i + 1 is built, while frames i - 1 -> i are still being
    aligned.  Each alignment in flight runs on its own aligning context (its own HIP stream); poses are delivered in
    frame order, one alignment late.  A lone alignment is a chain of 70 dependent launches that leaves the GPU almost
    idle, so two in flight nearly double the frame rate; the poses are the same bits as with one in flight."""
    params = params or MsIcpParams.default()
    builder = builder or RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    n = dataset.len() if max_frames is None else min(max_frames, dataset.len())
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    last = None
    # With alignments in flight lane 0 aligns on `ctx` from a worker thread, so the frames must never be built on `ctx`
    # by this thread at the same time (calls that take one context may not run concurrently: shared stream, scratch,
    # cached engine — include/align3d_hip.h): they are then always built on the sibling context, prefetch or not.
    side = ctx.sibling() if ((prefetch and n >= 3) or in_flight > 1) else None
    frames = _pyramids(dataset, builder, n, side)
    if in_flight > 1:
        return _finish(_run_pipelined(ctx, frames, params, tb, int(in_flight)), dataset, n)
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
    return _finish(tb, dataset, n)


def _finish(tb, dataset, n):
    pred = tb.build()
    gt = dataset.trajectory()
    metrics = None
    if gt is not None:
        metrics = TransformMetrics.mean_trajectory_error(pred, gt.slice(0, n).first_frame_at_origin())
    return pred, metrics


def _run_pipelined(ctx, frames, params, tb, in_flight):
    """The loop of run_odometry with up to `in_flight` alignments running at once: one lane per alignment in flight, each
    a worker thread with its own aligning context (its own HIP stream, on a compute pipe of its own:
    a3d_context_create_on_pipe) that runs the ordinary synchronous MultiscaleAlign::align.  An alignment is ~70 dependent
    launches: ~0.25 ms of host launch time and ~0.64 ms of an almost idle GPU; lanes overlap both.  Pairs are dealt to
    the lanes round robin and collected in frame order."""
    from .context import Context

    # one more aligning context per further lane (each creates its main stream on a compute pipe of its own:
    # a3d_context_create_on_pipe).  Creating one costs milliseconds (streams, page-locked block, its single-pair engine):
    # they are kept with the main context and closed with it (Context.close / release_lanes).
    aligners = getattr(ctx, "_lanes", None) or []
    for k in range(len(aligners), in_flight - 1):
        aligners.append(Context(ctx.device_index, pair=False, library=ctx._library, main_slot=1 + k % 2))
    ctx._lanes = aligners
    ctxs = [ctx] + aligners[:in_flight - 1]
    jobs = [queue.Queue() for _ in ctxs]
    done = [queue.Queue() for _ in ctxs]

    def lane(k):
        while True:
            job = jobs[k].get()
            if job is None:
                return
            i, target, source = job
            try:
                icp = MultiscaleAlign.new(ctxs[k], params, target)  # the previous frame's pyramid is the target
                T = icp.align(source)
                icp.free()
            except BaseException as e:  # hand the failure to the collector
                T = e
            done[k].put((i, T))

    workers = [threading.Thread(target=lane, args=(k,), name=f"a3d-aligner-{k}", daemon=True) for k in range(len(ctxs))]
    for w in workers:
        w.start()
    pending = []  # (frame index, lane, target pyramid to free once the result is in)
    last = None

    def collect():
        i, k, target = pending.pop(0)
        j, T = done[k].get()
        assert j == i
        for lv in target:
            lv.free()
        if isinstance(T, BaseException):
            raise T
        tb.accumulate(T, float(i))

    try:
        for i, cur in enumerate(frames):
            if last is not None:
                if len(pending) == in_flight:
                    collect()
                k = i % in_flight
                jobs[k].put((i, last, cur))
                pending.append((i, k, last))
            last = cur
        while pending:
            collect()
    finally:
        frames.close()
        for q in jobs:
            q.put(None)
        for w in workers:
            w.join()
        for i, k, target in pending:  # an exception on the way: the lanes have finished what was in flight
            for lv in target:
                lv.free()
        if last is not None:
            for lv in last:
                lv.free()
    return tb


def _same_camera(a, b):
    return (a.fx, a.fy, a.cx, a.cy, a.width, a.height) == (b.fx, b.fy, b.cx, b.cy, b.width, b.height)


def run_odometry_batched(ctx, dataset, params=None, builder=None, max_frames=None, window=64):
    """run_odometry for a RECORDED sequence (the example's use: a dataset on disk, examples/src/bin/odometry.rs:28-69).
    Frame-to-frame odometry is a chain only in its last step: the alignment of frames i-1 and i depends on nothing
    but those two frames, the trajectory is the running product of the results (TrajectoryBuilder::accumulate,
    trajectory.rs:164-168).  So a window of up to `window` + 1 consecutive frames is built in ONE batched builder call
    and its `window` alignments run as ONE MultiscaleAlignBatch (frame i-1 the target, frame i the source), windows
    overlapping by one frame; the transforms are accumulated on the host in frame order.  Same arithmetic per pair as
    run_odometry (results agree to the batch kernels' summation order, ~1e-7); many times its frame rate, because the
    single-pair path is a chain of 70 dependent ~14 us launches per alignment and the batch is not.
    Returns (predicted Trajectory, mean TransformMetrics against the dataset's ground truth or None)."""
    params = params or MsIcpParams.default()
    builder = builder or RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    n = dataset.len() if max_frames is None else min(max_frames, dataset.len())
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    free = lambda pyr: [lv.free() for lv in pyr]
    batches = {}  # pairs in the batch -> MultiscaleAlignBatch (rebound from window to window)
    carry = None  # the last frame of the previous window: (index, pyramid)
    i = 0
    try:
        while i < n:
            # a run of frames that one builder call can take: same camera, depth scale and size
            cam, depth, rgb, scale = dataset.get(i)
            run = [(depth, rgb)]
            while i + len(run) < n and len(run) < window + (0 if carry else 1):
                cam2, d2, c2, s2 = dataset.get(i + len(run))
                if not (_same_camera(cam, cam2) and s2 == scale and d2.shape == depth.shape):
                    break
                run.append((d2, c2))
            pyrs = builder.build_many(cam, run, scale)
            seq = ([carry[1]] if carry else []) + pyrs  # consecutive frames i - 1 (if carried), i, i + 1, ...
            first = i - (1 if carry else 0)
            if len(seq) >= 2:
                P = len(seq) - 1
                if P in batches:
                    batches[P].rebind(seq[:-1], seq[1:])
                else:
                    batches[P] = MultiscaleAlignBatch(ctx, params, seq[:-1], seq[1:])
                poses, status = batches[P].align()
                for k, (T, st) in enumerate(zip(poses, status)):
                    if st != 0:
                        raise _abi.A3dError(int(st), f"alignment of frames {first + k} and {first + k + 1}: GaussNewton::solve() "
                                                     f"returned None (count == 0 or Cholesky failed)")
                    tb.accumulate(T, float(first + k + 1))
            for pyr in seq[:-1]:
                free(pyr)
            carry = (first + len(seq) - 1, seq[-1])
            i += len(run)
    finally:
        if carry is not None:
            free(carry[1])
        for b in batches.values():
            b.free()
    pred = tb.build()
    gt = dataset.trajectory()
    metrics = None
    if gt is not None:
        metrics = TransformMetrics.mean_trajectory_error(pred, gt.slice(0, n).first_frame_at_origin())
    return pred, metrics
