"""run_odometry's loop with the frame-builder context at the highest (the sibling), default and lowest stream priority."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import align3d_amd.odometry as od
from align3d_amd import (BilateralFilter, Context, MsIcpParams, MultiscaleAlign, RangeImageBuilder, SyntheticDataset,
                         TrajectoryBuilder, Transform)

ctx = Context(0)
ds = SyntheticDataset(7, 20)
frames_host = [ds.get(i) for i in range(20)]
class Mem:
    def len(self): return 20
    def get(self, i): return frames_host[i]
    def trajectory(self): return ds.trajectory()
mem = Mem()
params = MsIcpParams.default()
sides = {"sibling(highest)": ctx.sibling(), "default": Context(0, priority=0, pair=False), "lowest": Context(0, priority=1)}
for rep in range(2):
    for name, side in sides.items():
        builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
        per = []
        for _ in range(3):
            frames = od._pyramids(mem, builder, 20, side)
            tb = TrajectoryBuilder.with_start(Transform.eye(), 0.0)
            last = None
            t0 = time.perf_counter()
            for i, cur in enumerate(frames):
                if last is not None:
                    icp = MultiscaleAlign.new(ctx, params, last)
                    tb.accumulate(icp.align(cur), float(i))
                    icp.free()
                    [lv.free() for lv in last]
                last = cur
            per.append((time.perf_counter() - t0) / 19 * 1e3)
            [lv.free() for lv in last]
        print(f"builder priority {name:18s}: {min(per):.3f} ms per frame ({1e3 / min(per):.0f} frames/s)", flush=True)
