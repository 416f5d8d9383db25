"""Batch of 16 pairs: repeated aligns bit-identical? equal to the one-stream run? equal to the matrices output?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids
ctx = Context(0)
P = 16
pyr, _, _ = build_stream_pyramids(ctx, 1000, P + 1, 640, 480)
prm = MsIcpParams.repeat(3, IcpParams.default())
def run(streams):
    os.environ["A3D_ICP_STREAMS"] = str(streams)
    b = MultiscaleAlignBatch(ctx, prm, [pyr[p] for p in range(P)], [pyr[p + 1] for p in range(P)])
    outs = []
    for _ in range(3):
        poses, status = b.align()
        outs.append(np.array([np.concatenate([t.t, t.q]) for t in poses], np.float32))
    d_m = ctx.malloc(P * 64)
    b.enqueue(matrices_device=d_m)
    m = np.zeros((P, 16), np.float32)
    ctx.to_host(d_m, m)
    mats = np.array([t.matrix().reshape(16) for t in poses], np.float32)
    print(f"streams={b.concurrency()} repeat-identical={all(np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs)} "
          f"matrices max|diff|={np.abs(m - mats).max():.3g}")
    b.free()
    return outs[0]
a = run(1)
c = run(3)
print("1-stream vs 3-stream max |diff| =", np.abs(a - c).max())
