"""Times the per-pixel kernel on ONE pyramid level for a batch of pairs (tuning aid)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch, synth
from bench import build_stream_pyramids, level_bytes

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=64)
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--frames", type=int, default=9, help="distinct synthetic frames, reused round-robin")
args = ap.parse_args()
ctx = Context(0)
pyr, _ = build_stream_pyramids(ctx, 1000, args.frames, 640, 480)
# distinct device copies per pair so that the working set is the real one
import copy
targets, sources = [], []
for p in range(args.pairs):
    t = copy.copy(pyr[p % (args.frames - 1)][args.level]); t._device = None
    s = copy.copy(pyr[p % (args.frames - 1) + 1][args.level]); s._device = None
    targets.append([t]); sources.append([s])
prm = MsIcpParams.repeat(1, IcpParams(max_iterations=args.iters))
batch = MultiscaleAlignBatch(ctx, prm, targets, sources)
for _ in range(2):
    batch.enqueue()
ctx.synchronize()
batch.set_profiling(True)
ks, tot = [], []
for _ in range(5):
    batch.enqueue(); ctx.synchronize()
    ks.append(batch.last_kernel_ms()); tot.append(batch.last_timing()[0])
w, h = 640 >> args.level, 480 >> args.level
kus = np.median(ks) / args.iters * 1e3
gbs = args.pairs * level_bytes(w, h) / (kus * 1e-6) / 1e9
print(f"level {args.level} pairs {args.pairs}: kernel {kus:.1f} us/launch, step {np.median(tot)/args.iters*1e3:.1f} us/iter, "
      f"{gbs:.0f} GB/s algorithmic ({gbs/8000:.3f} of 8 TB/s)  env={ {k:v for k,v in os.environ.items() if k.startswith('A3D_')} }")
