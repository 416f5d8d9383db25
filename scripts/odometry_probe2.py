"""Per-stage wall time of run_odometry's main loop with and without the prefetch worker."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import align3d_amd.odometry as od
from align3d_amd import (BilateralFilter, Context, MsIcpParams, MultiscaleAlign, RangeImageBuilder, SyntheticDataset,
                         TrajectoryBuilder, Transform)

ctx = Context(0)
ds = SyntheticDataset(7, 20)
params = MsIcpParams.default()
for prefetch in (False, True, False, True):
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    side = ctx.sibling() if prefetch else None
    frames = od._pyramids(ds, builder, 20, side)
    tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
    T = {"get": 0.0, "new": 0.0, "align": 0.0, "acc": 0.0, "free": 0.0}
    last = None
    t_all = time.perf_counter()
    it = iter(frames)
    i = 0
    while True:
        t0 = time.perf_counter()
        try:
            cur = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter(); T["get"] += t1 - t0
        if last is not None:
            icp = MultiscaleAlign.new(ctx, params, last); t2 = time.perf_counter(); T["new"] += t2 - t1
            tr = icp.align(cur); t3 = time.perf_counter(); T["align"] += t3 - t2
            tb.accumulate(tr, float(i)); t4 = time.perf_counter(); T["acc"] += t4 - t3
            icp.free(); [lv.free() for lv in last]; t5 = time.perf_counter(); T["free"] += t5 - t4
        last = cur
        i += 1
    total = time.perf_counter() - t_all
    [lv.free() for lv in last]
    if side: ctx.synchronize()
    print("prefetch=%s total %.2f ms/frame " % (prefetch, total / 19 * 1e3), {k: round(v / 19 * 1e3, 3) for k, v in T.items()})
