"""A lone ms3x15 alignment, N times (for rocprofv3 --kernel-trace: scripts/gap_table.py turns the trace into the
end -> start gap table of the 45-launch chain, level by level).  Prints the host-side latency per alignment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlign
from bench import build_stream_pyramids
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 1000, 2, 640, 480)
ms = MultiscaleAlign.new(ctx, MsIcpParams.repeat(3, IcpParams.default()), pyr[0])
for _ in range(3):
    ms.align(pyr[1])
t = []
for _ in range(n):
    t0 = time.perf_counter()
    ms.align(pyr[1])
    t.append((time.perf_counter() - t0) * 1e3)
print(f"lone ms3x15 pair: median {np.median(t):.4f} ms, min {min(t):.4f} ms over {n} alignments")
