import os, sys, threading, time
sys.path.insert(0, "/root/repo")
import numpy as np
from align3d_amd import BilateralFilter, Context, MsIcpParams, MultiscaleAlign, MultiscaleAlignBatch, RangeImageBuilder, synth
ctx = Context(0)
c2 = Context(0, pair=False, main_slot=int(os.environ.get("SLOT", "-1")))
frames, _ = synth.frame_stream(7, 4, 640, 480)
cam = synth.camera(640, 480)
bm = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
prm = MsIcpParams.default()
p = [bm.build_device(cam, *f, synth.DEPTH_SCALE) for f in frames]
def loop(c, a, b, k, out, key):
    icp = MultiscaleAlign.new(c, prm, a)
    icp.align(b)
    t = time.perf_counter()
    for _ in range(k): icp.align(b)
    out[key] = (time.perf_counter() - t) / k * 1e3
out = {}
loop(ctx, p[0], p[1], 40, out, "alone"); print("alone", out["alone"])
t1 = threading.Thread(target=loop, args=(ctx, p[0], p[1], 80, out, "a"))
t2 = threading.Thread(target=loop, args=(c2, p[2], p[3], 80, out, "b"))
t1.start(); t2.start(); t1.join(); t2.join()
print("two threads, two contexts:", out["a"], out["b"])
# enqueue-only batches of one pair on the two contexts from ONE thread
b1 = MultiscaleAlignBatch(ctx, prm, [p[0]], [p[1]]); b2 = MultiscaleAlignBatch(c2, prm, [p[2]], [p[3]])
for b in (b1, b2): b.align()
t = time.perf_counter()
for _ in range(40):
    b1.enqueue(); b2.enqueue(); b1.results(); b2.results()
print("one thread, enqueue both then read both: per alignment", (time.perf_counter() - t) / 80 * 1e3)
t = time.perf_counter()
for _ in range(40):
    b1.enqueue()
print("host time of one enqueue", (time.perf_counter() - t) / 40 * 1e3); ctx.synchronize()
