"""Persistent head-solve kernel against per-iteration launches (diagnostics build; knobs in the environment):
   python scripts/persist_probe.py lone   -> ms per lone ms3x15 alignment
   python scripts/persist_probe.py batch  -> ms per 64-pair step
One configuration per process (the knobs are read when a batch is created)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlign, MultiscaleAlignBatch, _abi

mode = sys.argv[1]
ctx = Context(0, library=_abi.DIAG_LIB_PATH)
prm = MsIcpParams.repeat(3, IcpParams.default())
knobs = {k: v for k, v in os.environ.items() if k.startswith("A3D_ICP")}
if mode == "lone":
    pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2, 640, 480)
    ms = MultiscaleAlign.new(ctx, prm, pyr[0])
    for _ in range(5):
        ms.align(pyr[1])
    lat = []
    for _ in range(40):
        t = time.perf_counter()
        ms.align(pyr[1])
        lat.append((time.perf_counter() - t) * 1e3)
    print(f"lone {knobs}: median {np.median(lat):.4f} ms  min {np.min(lat):.4f}")
else:
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2 * P, 640, 480)
    b = MultiscaleAlignBatch(ctx, prm, [pyr[2 * p] for p in range(P)], [pyr[2 * p + 1] for p in range(P)])
    for _ in range(10):
        b.enqueue()
    ctx.synchronize()
    reps = []
    for _ in range(10):
        t = time.perf_counter()
        for _ in range(30):
            b.enqueue()
        ctx.synchronize()
        reps.append((time.perf_counter() - t) / 30 * 1e3)
    print(f"batch {P} {knobs}: median {np.median(reps):.4f} ms/step  min {np.min(reps):.4f}  persistent levels {b.persistent_levels():03b}")
