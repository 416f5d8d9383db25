"""How much faster is the 64-pair ms3x15 step when its frames fit the 256 MB Infinity Cache?  64 pairs over the frames of
only D distinct pairs (pair p uses the frames of pair p % D): the launch geometry of the headline, a working set of
2 D pyramids of 8.8 MB.  python3 scripts/mall_probe.py 64 16 8 4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
ctx = Context(0)
prm = MsIcpParams.repeat(3, IcpParams.default())
P = 64
pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2 * P, 640, 480)
for D in [int(a) for a in sys.argv[1:]] or [64, 8]:
    b = MultiscaleAlignBatch(ctx, prm, [pyr[2 * (p % D)] for p in range(P)], [pyr[2 * (p % D) + 1] for p in range(P)])
    for _ in range(10):
        b.enqueue()
    ctx.synchronize()
    reps = []
    for _ in range(7):
        t = time.perf_counter()
        for _ in range(30):
            b.enqueue()
        ctx.synchronize()
        reps.append((time.perf_counter() - t) / 30 * 1e3)
    ms = float(np.median(reps))
    print(f"{D} distinct pairs ({2 * D * 8.8:.0f} MB of pyramids): {ms:.3f} ms per 64-pair step = {P / ms:.1f} k pairs/s", flush=True)
    b.free()
