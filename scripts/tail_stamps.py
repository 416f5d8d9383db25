"""Reads the s_memrealtime stamps of the solve tail from the diagnostic build (scripts/stampbuild)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["A3D_LIBRARY"] = os.path.join(ROOT, "scripts", "stampbuild", "libalign3d_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids
ctx = Context(0)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pyr, _, _ = build_stream_pyramids(ctx, 1000, pairs + 1, 640, 480)
names = ["enter publish", "ticket drawn (last)", "partials summed", "H built", "Cholesky done", "substitution done", "state written"]
# [8] block 0 of pair 0 enters the kernel, [9] its pixel loop is done, [0] the pair's LAST block has reduced and enters publish
for level in (0, 1, 2):
    prm = MsIcpParams.repeat(1, IcpParams(max_iterations=1))
    batch = MultiscaleAlignBatch(ctx, prm, [[pyr[p][level]] for p in range(pairs)], [[pyr[p + 1][level]] for p in range(pairs)])
    rows = []
    for _ in range(5):
        batch.align()
        st = (C.c_ulonglong * 16)()
        assert ctx.lib.a3d_debug_tail_stamps(st) == 0
        rows.append([st[k] for k in range(10)])
    r = np.median(np.array(rows, np.float64), axis=0)
    print(f"level {level}, {pairs} pair(s): block 0 pixel loop {(r[9]-r[8])/100:.2f}us; from block 0's entry to the last block's publish "
          f"{(r[0]-r[8])/100:.2f}us; " + "; ".join(f"{names[k]} +{(r[k]-r[k-1])/100:.2f}us" for k in range(1, 7)) +
          f"; tail {(r[6]-r[0])/100:.2f}us; entry to state written {(r[6]-r[8])/100:.2f}us")
