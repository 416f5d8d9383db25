"""Leak check: device memory in use before / after many odometry runs, batch creations and tree builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from align3d_amd import (Context, Icp, IcpParams, MsIcpParams, MultiscaleAlignBatch, PointCloud, R3dTree, SyntheticDataset,
                         run_odometry)
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
ctx = Context(0)
ds = SyntheticDataset(3, 12)
run_odometry(ctx, ds)
base = used()
print(f"after warm-up: {base:.0f} MiB in use")
for rep in range(15):
    run_odometry(ctx, ds)
print(f"after 15 odometry runs (165 alignments, 180 frame builds): {used() - base:+.0f} MiB")
from bench import build_stream_pyramids
pyr, _, _ = build_stream_pyramids(ctx, 5, 17, 640, 480)
mid = used()
for rep in range(20):
    b = MultiscaleAlignBatch(ctx, MsIcpParams.repeat(3, IcpParams.default()), pyr[:16], pyr[1:])
    b.align(); b.free()
print(f"after 20 batch create/align/free cycles: {used() - mid:+.0f} MiB")
pts = np.random.default_rng(0).random((200000, 3), dtype=np.float32)
nrm = np.tile(np.array([[0, 0, 1]], np.float32), (200000, 1))
t0 = used()
for rep in range(20):
    R3dTree.new(ctx, pts).free()
    icp = Icp.new(ctx, IcpParams.default(), PointCloud(pts, nrm))
    try:
        icp.align(PointCloud(pts[:50000], nrm[:50000]))  # degenerate normals: the solve is allowed to fail
    except Exception:
        pass
    icp.free()
print(f"after 20 tree builds + 20 Icp new/align/free: {used() - t0:+.0f} MiB")
for p in pyr:
    for lv in p: lv.free()
