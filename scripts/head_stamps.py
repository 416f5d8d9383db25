"""Where one iteration's launch of a LONE pair spends its time: s_memrealtime stamps (100 MHz) written by the block of
the last tile of image_icp_head_kernel in the stamp build (scripts/build_stamps.sh).  Two consecutive launches are
kept (slot base alternates with the launch), so the end -> entry interval across the kernel boundary is visible too."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["A3D_LIBRARY"] = os.path.join(ROOT, "scripts", "stampbuild", "libalign3d_hip_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 1000, 2, 640, 480)
print("level | tiles | entry -> descriptor + source requests | -> head done (partials summed, solve, pose) | -> pixel pass done | -> partial stored | whole launch body | previous launch's store -> this entry (kernel boundary)")
for level in (2, 1, 0):
    prm = MsIcpParams.repeat(1, IcpParams(max_iterations=8))
    batch = MultiscaleAlignBatch(ctx, prm, [[pyr[0][level]]], [[pyr[1][level]]])
    rows = []
    for _ in range(9):
        batch.align()
        st = (C.c_ulonglong * 48)()
        assert ctx.lib.a3d_debug_head_stamps(st) == 0
        a, b = [st[k] for k in range(5)], [st[8 + k] for k in range(5)]
        late, early = (a, b) if a[0] > b[0] else (b, a)  # the later launch and the one before it
        hs = [st[32 + k] for k in range(7)]  # inside the head of the last launch that ran one (job_finish_head_kernel has no such block)
        hc = [st[40 + k] for k in range(7)]  # the same stamps in shader-clock cycles (s_memtime)
        rows.append([late[1] - late[0], late[2] - late[1], late[3] - late[2], late[4] - late[3], late[4] - late[0], late[0] - early[4]]
                    + [hs[k + 1] - hs[k] for k in range(6)] + [(hc[5] - hc[2]) / max(1, hs[5] - hs[2])])
    r = np.median(np.array(rows, np.float64), axis=0) / 100.0
    print(f"{level} | {batch.last_timing()[1]} launches | " + " | ".join(f"{v:.2f} us" for v in r[:6]))
    print("      inside the head: partials loaded + added %.2f | totals in LDS (2 barriers) %.2f | Cholesky %.2f | substitutions %.2f | exp + compose + best %.2f | state to LDS + barrier %.2f us" % tuple(r[6:12]))
    print("      shader clock during the solve (s_memtime cycles per s_memrealtime tick x 100 MHz): %.0f MHz" % (r[12] * 100.0 * 100.0))
    batch.free()
