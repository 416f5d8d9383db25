"""Cost of short blocks: the 64-pair ms3x15 step under pinned tilings (tiles per pair and level).
  python3 scripts/tiling_probe.py 0 24 48 96 136 200"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
ctx = Context(0)
prm = MsIcpParams.repeat(3, IcpParams.default())
P = 64
pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2 * P, 640, 480)
for T in [int(a) for a in sys.argv[1:]] or [0, 24, 136]:
    ctx.set_tiling(T)
    b = MultiscaleAlignBatch(ctx, prm, [pyr[2 * p] for p in range(P)], [pyr[2 * p + 1] for p in range(P)])
    for _ in range(10):
        b.enqueue()
    ctx.synchronize()
    reps = []
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(30):
            b.enqueue()
        ctx.synchronize()
        reps.append((time.perf_counter() - t) / 30 * 1e3)
    ms = float(np.median(reps))
    print(f"tiles per pair {T or 'throughput tiling'}: {ms:.3f} ms per step = {P / ms:.1f} k pairs/s", flush=True)
    b.free()
