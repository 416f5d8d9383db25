"""Where a streaming round's time goes: build thread wall time, rebind, align, wait for the builder (tuning aid)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import BilateralFilter, Context, IcpParams, MsIcpParams, MultiscaleAlignBatch, RangeImageBuilder, synth
P, W, H = 64, 640, 480
ctx = Context(0)
frames, _ = synth.frame_stream(4242, P + 1, W, H)
all_d, all_c = ctx.pinned_empty((P + 1, H, W), np.uint16), ctx.pinned_empty((P + 1, H, W, 3), np.uint8)
for i, (d, c) in enumerate(frames):
    all_d[i], all_c[i] = d, c
frames = [(all_d[i], all_c[i]) for i in range(P + 1)]
cam = synth.camera(W, H)
bctx = [Context(0), Context(0)]
bld = [RangeImageBuilder(c).with_bilateral_filter(BilateralFilter.default()) for c in bctx]
prm = MsIcpParams.repeat(3, IcpParams.default())
cur = bld[0].build_many(cam, frames, synth.DEPTH_SCALE)
warm = bld[1].build_many(cam, frames, synth.DEPTH_SCALE)
for lv in (lv for p in warm for lv in p): lv.free()
batch = MultiscaleAlignBatch(ctx, prm, cur[:P], cur[1:])
batch.align()
# alone
t0 = time.perf_counter(); batch.align(); print(f"align alone {(time.perf_counter()-t0)*1e3:.2f} ms")
t0 = time.perf_counter(); x = bld[1].build_many(cam, frames, synth.DEPTH_SCALE); print(f"build alone {(time.perf_counter()-t0)*1e3:.2f} ms")
for lv in (lv for p in x for lv in p): lv.free()
for r in range(6):
    out, tb = {}, {}
    def work(which=(r + 1) % 2):
        t = time.perf_counter(); out["p"] = bld[which].build_many(cam, frames, synth.DEPTH_SCALE); tb["b"] = time.perf_counter() - t
    th = threading.Thread(target=work); t_round = time.perf_counter(); th.start()
    t = time.perf_counter(); batch.rebind(cur[:P], cur[1:]); t_rebind = time.perf_counter() - t
    t = time.perf_counter(); batch.align(); t_align = time.perf_counter() - t
    t = time.perf_counter(); th.join(); t_join = time.perf_counter() - t
    t = time.perf_counter()
    for lv in (lv for p in cur for lv in p): lv.free()
    t_free = time.perf_counter() - t
    cur = out["p"]
    print(f"round {r}: build {tb['b']*1e3:.2f}  rebind {t_rebind*1e3:.2f}  align {t_align*1e3:.2f}  join wait {t_join*1e3:.2f}  free {t_free*1e3:.2f}  total {(time.perf_counter()-t_round)*1e3:.2f} ms")
