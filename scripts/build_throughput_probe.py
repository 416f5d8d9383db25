"""Frame-build throughput (u16 depth + RGB -> resident 3-level pyramid): single builds, batched builds of 16 / 65
frames (pageable and page-locked host buffers), and N builder threads each with its own context."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import BilateralFilter, Context, RangeImageBuilder, synth

W, H = 640, 480
frames, _ = synth.frame_stream(4242, 65, W, H)
cam = synth.camera(W, H)
ctx = Context(0)
b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())


def free(pyrs):
    for lv in (lv for p in pyrs for lv in p):
        lv.free()


def timed(label, fn, n_frames, reps=5):
    free(fn())
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
        free(out)
    t = float(np.median(ts))
    print(f"{label}: {t / n_frames * 1e3:.3f} ms per frame ({n_frames / t:.0f} frames/s)", flush=True)


timed("single builds, pageable", lambda: [b.build(cam, d, c, synth.DEPTH_SCALE) for d, c in frames[:16]], 16)
timed("batch of 16, pageable", lambda: b.build_many(cam, frames[:16], synth.DEPTH_SCALE), 16)
timed("batch of 65, pageable", lambda: b.build_many(cam, frames, synth.DEPTH_SCALE), 65)
pinned = []
for d, c in frames:
    pd, pc = ctx.pinned_empty(d.shape, d.dtype), ctx.pinned_empty(c.shape, c.dtype)
    pd[...], pc[...] = d, c
    pinned.append((pd, pc))
timed("single builds, page-locked", lambda: [b.build(cam, d, c, synth.DEPTH_SCALE) for d, c in pinned[:16]], 16)
timed("batch of 16, page-locked", lambda: b.build_many(cam, pinned[:16], synth.DEPTH_SCALE), 16)
timed("batch of 65, page-locked", lambda: b.build_many(cam, pinned, synth.DEPTH_SCALE), 65)
cd, cc = ctx.pinned_empty((65, H, W), np.uint16), ctx.pinned_empty((65, H, W, 3), np.uint8)
for i, (d, c) in enumerate(frames):
    cd[i], cc[i] = d, c
contig = [(cd[i], cc[i]) for i in range(65)]
timed("batch of 65, page-locked, frames back to back in one buffer", lambda: b.build_many(cam, contig, synth.DEPTH_SCALE), 65)
nb = RangeImageBuilder(ctx)
timed("batch of 65, page-locked, no bilateral filter", lambda: nb.build_many(cam, pinned, synth.DEPTH_SCALE), 65)
for n_threads in (2, 4):
    ctxs = [Context(0) for _ in range(n_threads)]
    builders = [RangeImageBuilder(c).with_bilateral_filter(BilateralFilter.default()) for c in ctxs]
    def work(k, out):
        out[k] = builders[k].build_many(cam, pinned[k::n_threads], synth.DEPTH_SCALE)
    def run():
        out = [None] * n_threads
        ts = [threading.Thread(target=work, args=(k, out)) for k in range(n_threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
        return [p for o in out for p in o]
    timed(f"65 frames over {n_threads} builder threads / contexts, page-locked", run, 65)
    for c in ctxs:
        c.close()
