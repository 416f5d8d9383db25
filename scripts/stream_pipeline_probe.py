"""Host timeline of the pipelined streaming loop (bench.py streaming_bench): when each build ends, when each round is
rebound / enqueued / read, per round (tuning aid).  usage: stream_pipeline_probe.py [rounds] [queue_depth] [builder priority: -1 high]"""
import os, queue, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import BilateralFilter, Context, IcpParams, MsIcpParams, MultiscaleAlignBatch, RangeImageBuilder, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P, W, H = 64, 640, 480
prio = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx, bctx = Context(0), Context(0, priority=prio)
frames, _ = synth.frame_stream(4242, P + 1, W, H)
all_d, all_c = ctx.pinned_empty((P + 1, H, W), np.uint16), ctx.pinned_empty((P + 1, H, W, 3), np.uint8)
for i, (d, c) in enumerate(frames):
    all_d[i], all_c[i] = d, c
frames = [(all_d[i], all_c[i]) for i in range(P + 1)]
cam = synth.camera(W, H)
bld = RangeImageBuilder(bctx).with_bilateral_filter(BilateralFilter.default())
prm = MsIcpParams.repeat(3, IcpParams.default())
free = lambda pyr: [lv.free() for p in pyr for lv in p]
cur = bld.build_many(cam, frames, synth.DEPTH_SCALE)
batches = [MultiscaleAlignBatch(ctx, prm, cur[:P], cur[1:]) for _ in range(2)]
for b in batches:
    b.align()
t = time.perf_counter(); batches[0].align(); print(f"align alone {(time.perf_counter()-t)*1e3:.2f} ms")
free(cur)
t = time.perf_counter(); x = bld.build_many(cam, frames, synth.DEPTH_SCALE); print(f"build alone {(time.perf_counter()-t)*1e3:.2f} ms"); free(x)
built = queue.Queue(maxsize=depth)
T0 = time.perf_counter()
now = lambda: (time.perf_counter() - T0) * 1e3
log = []
def producer():
    for r in range(R):
        a = now(); pyr = bld.build_many(cam, frames, synth.DEPTH_SCALE); b = now()
        built.put(pyr); log.append((r, "build", a, b, now()))
th = threading.Thread(target=producer); th.start()
prev = None
for r in range(R):
    a = now(); pyr = built.get(); b0 = now()
    b = batches[r % 2]
    b.rebind(pyr[:P], pyr[1:]); c = now()
    b.enqueue(); d = now()
    e = f = d
    if prev is not None:
        prev[0].results(); e = now()
        free(prev[1]); f = now()
    prev = (b, pyr)
    print(f"round {r}: wait for build {b0-a:.2f} (at {b0:.2f})  rebind {c-b0:.2f}  enqueue {d-c:.2f}  results(r-1) {e-d:.2f}  free {f-e:.2f}  -> {f:.2f}")
prev[0].results(); end = now()
th.join(); free(prev[1])
for r, what, a, b, c in log:
    print(f"  build {r}: {a:.2f} -> {b:.2f} ({b-a:.2f} ms), queued at {c:.2f}")
print(f"total {end:.2f} ms for {R} rounds = {R*P/end*1e3:.0f} pairs/s")
