"""Per-level cost of the ICP kernel in a 64-pair batch: device time of a 15-iteration single-level sequence
(all stream groups included) per iteration, with and without the solve tail, for 1 and 3 stream groups."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the A3D_ICP_* knobs exist in the diagnostics build only
os.environ.setdefault("A3D_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "align3d_amd", "csrc", "libalign3d_hip_diag.so"))
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids, level_bytes
ctx = Context(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pyr, _, _ = build_stream_pyramids(ctx, 1000, P + 1, 640, 480)
iters = 15
for level in (0, 1, 2):
    for streams, waves in ((1, 1.0), (3, 1.5)):
        for nosolve in (0, 1):
            os.environ["A3D_ICP_STREAMS"] = str(streams)
            os.environ["A3D_ICP_WAVES"] = str(waves)
            if nosolve:
                os.environ["A3D_ICP_NOSOLVE"] = "1"
            else:
                os.environ.pop("A3D_ICP_NOSOLVE", None)
            prm = MsIcpParams.repeat(1, IcpParams(max_iterations=iters))
            b = MultiscaleAlignBatch(ctx, prm, [[pyr[p][level]] for p in range(P)], [[pyr[p + 1][level]] for p in range(P)])
            for _ in range(3):
                b.enqueue()
            ctx.synchronize()
            t = []
            for _ in range(7):
                b.enqueue(); ctx.synchronize()
                t.append(b.last_timing()[0])
            us = float(np.median(t)) / iters * 1e3
            gbs = P * level_bytes(640 >> level, 480 >> level) / (us * 1e-6) / 1e9
            print(f"level {level} streams {streams} nosolve {nosolve}: {us:7.1f} us per iteration of {P} pairs = {gbs:6.0f} GB/s ({gbs / 8000:.3f})", flush=True)
            b.free()
