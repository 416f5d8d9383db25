"""A lone pair, level by level: device time per iteration of a 15-iteration single-level alignment, for a list of
A3D_ICP_WAVES values (rounds of resident blocks the grid makes; fewer = fewer, fatter blocks)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the A3D_ICP_* knobs exist in the diagnostics build only
os.environ.setdefault("A3D_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "align3d_amd", "csrc", "libalign3d_hip_diag.so"))
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 1000, 2, 640, 480)
iters = 15
for waves in [float(a) for a in sys.argv[1:]] or [0.25]:
    os.environ["A3D_ICP_WAVES"] = str(waves)
    row = []
    for level in (0, 1, 2):
        prm = MsIcpParams.repeat(1, IcpParams(max_iterations=iters))
        b = MultiscaleAlignBatch(ctx, prm, [[pyr[0][level]]], [[pyr[1][level]]])
        for _ in range(3):
            b.enqueue()
        ctx.synchronize()
        t = []
        for _ in range(9):
            b.enqueue(); ctx.synchronize()
            t.append(b.last_timing()[0])
        row.append(float(np.median(t)) / iters * 1e3)
        b.free()
    print(f"waves {waves:6.4f}: us per iteration level 0 / 1 / 2 = {row[0]:.2f} / {row[1]:.2f} / {row[2]:.2f}   handoff={os.environ.get('A3D_ICP_HANDOFF', 'head')}", flush=True)
