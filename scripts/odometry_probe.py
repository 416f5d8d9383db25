"""Where a frame of the ONLINE odometry loop goes (align3d_amd/odometry.py, MsIcpParams::default(), 640x480 synthetic
stream): alignment alone, single-frame build alone, the two at once on the aligner + builder contexts, and the loop."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import (BilateralFilter, Context, MsIcpParams, MultiscaleAlign, RangeImageBuilder, SyntheticDataset,
                         run_odometry, synth)

ctx = Context(0)
side = ctx.sibling()
n = 40
frames, _ = synth.frame_stream(7, n, 640, 480)
cam = synth.camera(640, 480)
bm = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
bs = RangeImageBuilder(side).with_bilateral_filter(BilateralFilter.default())
prm = MsIcpParams.default()
p0 = bm.build_device(cam, *frames[0], synth.DEPTH_SCALE)
p1 = bm.build_device(cam, *frames[1], synth.DEPTH_SCALE)


def t_align(k=30):
    icp = MultiscaleAlign.new(ctx, prm, p0)
    icp.align(p1)
    t = time.perf_counter()
    for _ in range(k):
        icp.align(p1)
    return (time.perf_counter() - t) / k * 1e3


def t_build(k=30):
    for lv in bs.build_device(cam, *frames[2], synth.DEPTH_SCALE):
        lv.free()
    t = time.perf_counter()
    for i in range(k):
        for lv in bs.build_device(cam, *frames[2 + i % 8], synth.DEPTH_SCALE):
            lv.free()
    side.synchronize()
    return (time.perf_counter() - t) / k * 1e3


print(f"align alone      {t_align():.3f} ms", flush=True)
print(f"build alone      {t_build():.3f} ms", flush=True)
out = {}
th = threading.Thread(target=lambda: out.setdefault("b", t_build(60)))
th.start()
a = t_align(60)
th.join()
print(f"both at once     align {a:.3f} ms, build {out['b']:.3f} ms", flush=True)
ds = SyntheticDataset(7, n, 640, 480)
for prefetch in (True, False):
    run_odometry(ctx, ds, max_frames=6, prefetch=prefetch)
    t = time.perf_counter()
    run_odometry(ctx, ds, max_frames=n, prefetch=prefetch)
    print(f"loop prefetch={prefetch}: {(time.perf_counter() - t) / (n - 1) * 1e3:.3f} ms per frame", flush=True)
for k in (2, 3):
    run_odometry(ctx, ds, max_frames=6, in_flight=k)
    t = time.perf_counter()
    run_odometry(ctx, ds, max_frames=n, in_flight=k)
    print(f"loop in_flight={k}: {(time.perf_counter() - t) / (n - 1) * 1e3:.3f} ms per frame", flush=True)
