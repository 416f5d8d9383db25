"""Device memory in use before / after hundreds of cycles of every object that allocates: batches, trees, Icp,
batched frame builds (arena pool and slabs), the multi-device batch, odometry.  Growth after the first cycles = a leak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from align3d_amd import (BilateralFilter, Context, Icp, IcpParams, MsIcpParams, MultiContext, MultiscaleAlignBatch,
                         MultiscaleAlignMultiBatch, PointCloud, R3dTree, RangeImageBuilder, SyntheticDataset, run_odometry, synth)
from bench import build_stream_pyramids


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def cycle(label, fn, groups=4, reps=25):
    fn()
    fn()
    m = used()
    for k in range(groups):
        for _ in range(reps):
            fn()
        print(f"{label} x{reps * (k + 1)}: {used() - m:+.1f} MiB", flush=True)


ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 5, 17, 640, 480)
prm = MsIcpParams.repeat(3, IcpParams.default())


def batch_cycle():
    b = MultiscaleAlignBatch(ctx, prm, pyr[:16], pyr[1:])
    b.align()
    b.free()


cycle("batch new/align/free", batch_cycle)
pts = np.random.default_rng(0).random((200000, 3), dtype=np.float32)
cycle("R3dTree new/free", lambda: R3dTree.new(ctx, pts).free())
nrm = np.random.default_rng(1).normal(size=(200000, 3)).astype(np.float32)
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
pc = PointCloud(pts, nrm)


def icp_cycle():
    icp = Icp.new(ctx, IcpParams(max_iterations=2), pc)
    icp.align(pc)
    icp.free()


cycle("Icp new/align/free", icp_cycle)
frames, _ = synth.frame_stream(9, 20, 640, 480)
cam = synth.camera(640, 480)
bld = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())


def build_cycle():
    for p in bld.build_many(cam, frames, synth.DEPTH_SCALE):
        for lv in p:
            lv.free()


cycle("build_many(20 frames)/free", build_cycle, reps=10)


def multi_cycle():
    mc = MultiContext([0, 0])
    tp, sp = [], []
    for d in range(2):
        own = RangeImageBuilder(mc.device(d)).build_many(cam, frames[4 * d:4 * d + 5], synth.DEPTH_SCALE)
        tp += own[:-1]
        sp += own[1:]
    mb = MultiscaleAlignMultiBatch(mc, prm, tp, sp)
    mb.align()
    mb.free()
    mc.close()


cycle("MultiContext + multi batch", multi_cycle, reps=5)
ds = SyntheticDataset(7, 6)
cycle("run_odometry(6 frames)", lambda: run_odometry(ctx, ds), reps=5)
from align3d_amd import run_odometry_batched
cycle("run_odometry_batched(6 frames, windows of 2)", lambda: run_odometry_batched(ctx, ds, window=2), reps=5)
import bench
cycle("streaming_bench (3 rounds of 16 pairs, two alternating batches, images freed behind fences)",
      lambda: bench.streaming_bench(ctx, prm, 16, 640, 480, rounds=3), groups=3, reps=4)


# round 3: the drop-in call from host pyramids (pooled pyramid upload + align + free) and compute_normals on uploaded images
from align3d_amd import MultiscaleAlign
host_src = [lv.download(colors=False) for lv in pyr[1]]
ms = MultiscaleAlign.new(ctx, prm, pyr[0])


def drop_in_cycle():
    for h in host_src:
        if h._device is not None:
            h._device.free()
        h._device = None
    ms.align(host_src)


cycle("drop-in align from host pyramids", drop_in_cycle)


def normals_cycle():
    h = host_src[0]
    if h._device is not None:
        h._device.free()
    h._device = None
    h.device(ctx).compute_normals()


cycle("upload + compute_normals + free", normals_cycle)
