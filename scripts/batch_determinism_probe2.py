"""Repeat-identity of a 64-pair batch over many runs (run two copies at once to add scheduling noise)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids
ctx = Context(0)
P = 64
pyr, _, _ = build_stream_pyramids(ctx, 1000, P + 1, 640, 480)
prm = MsIcpParams.repeat(3, IcpParams.default())
b = MultiscaleAlignBatch(ctx, prm, [pyr[p] for p in range(P)], [pyr[p + 1] for p in range(P)])
ref = None
bad = 0
for it in range(60):
    poses, status = b.align()
    o = np.array([np.concatenate([t.t, t.q]) for t in poses], np.float32).view(np.uint32)
    if ref is None:
        ref = o
    elif not np.array_equal(ref, o):
        bad += 1
        rows = np.nonzero((ref != o).any(axis=1))[0]
        print(f"run {it}: {len(rows)} pair(s) differ: {rows[:10]} max|d|={np.abs(ref.view(np.float32) - o.view(np.float32)).max():.3g}", flush=True)
print(f"pid {os.getpid()} streams={b.concurrency()}: {bad} of 59 repeats differ", flush=True)
