"""Repeat-identity of Icp::align (500k x 500k) and of a 64-pair one-stream batch; run two copies at once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import Context, Icp, IcpParams, MsIcpParams, MultiscaleAlignBatch, PointCloud, RangeImageBuilder, synth
from bench import build_stream_pyramids
ctx = Context(0)
frames, poses = synth.frame_stream(7, 2, 880, 660)
cam = synth.camera(880, 660)
clouds = []
for d, rgb in frames:
    ri = RangeImageBuilder(ctx).pyramid_levels(1).with_intensity(False).build(cam, d, rgb, synth.DEPTH_SCALE)[0].download(intensity=False)
    pc = PointCloud.from_range_image(ri)
    clouds.append(PointCloud(pc.points[:500000], pc.normals[:500000]))
icp = Icp.new(ctx, IcpParams.default(), clouds[0])
ref, bad = None, 0
for it in range(40):
    T = icp.align(clouds[1])
    o = np.concatenate([T.t, T.q]).astype(np.float32).view(np.uint32)
    if ref is None: ref = o
    elif not np.array_equal(ref, o): bad += 1
print(f"pid {os.getpid()} Icp::align: {bad} of 39 repeats differ", flush=True)
os.environ["A3D_ICP_STREAMS"] = "1"
P = 64
pyr, _, _ = build_stream_pyramids(ctx, 1000, P + 1, 640, 480)
b = MultiscaleAlignBatch(ctx, MsIcpParams.repeat(3, IcpParams.default()), [pyr[p] for p in range(P)], [pyr[p + 1] for p in range(P)])
ref, bad = None, 0
for it in range(40):
    ps, st = b.align()
    o = np.array([np.concatenate([t.t, t.q]) for t in ps], np.float32).view(np.uint32)
    if ref is None: ref = o
    elif not np.array_equal(ref, o): bad += 1
print(f"pid {os.getpid()} one-stream batch: {bad} of 39 repeats differ", flush=True)
