import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch, R3dTree
from bench import build_stream_pyramids
def used():
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 2**20
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 5, 17, 640, 480)
prm = MsIcpParams.repeat(3, IcpParams.default())
b = MultiscaleAlignBatch(ctx, prm, pyr[:16], pyr[1:]); b.align(); b.free()
m0 = used()
for k in range(5):
    for rep in range(40):
        b = MultiscaleAlignBatch(ctx, prm, pyr[:16], pyr[1:]); b.align(); b.free()
    print(f"batch cycles {40 * (k + 1)}: {used() - m0:+.1f} MiB", flush=True)
pts = np.random.default_rng(0).random((200000, 3), dtype=np.float32)
R3dTree.new(ctx, pts).free()
m1 = used()
for k in range(4):
    for rep in range(40):
        R3dTree.new(ctx, pts).free()
    print(f"tree builds {40 * (k + 1)}: {used() - m1:+.1f} MiB", flush=True)
