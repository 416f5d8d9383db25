import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from align3d_amd import BilateralFilter, Context, MsIcpParams, MultiscaleAlignBatch, RangeImageBuilder, SyntheticDataset
ctx = Context(0)
ds = SyntheticDataset(7, 20)
frames = [ds.get(i) for i in range(20)]
cam, scale = frames[0][0], frames[0][3]
run = [(f[1], f[2]) for f in frames]
b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
prm = MsIcpParams.default()
for rep in range(3):
    t0 = time.perf_counter(); pyr = b.build_many(cam, run, scale); t1 = time.perf_counter()
    batch = MultiscaleAlignBatch(ctx, prm, pyr[:-1], pyr[1:]); t2 = time.perf_counter()
    poses, st = batch.align(); t3 = time.perf_counter()
    poses, st = batch.align(); t4 = time.perf_counter()
    batch.free(); t5 = time.perf_counter()
    for p in pyr:
        for lv in p: lv.free()
    t6 = time.perf_counter()
    print(f"build {1e3*(t1-t0):.2f}  batch new {1e3*(t2-t1):.2f}  align {1e3*(t3-t2):.2f}  align again {1e3*(t4-t3):.2f}  batch free {1e3*(t5-t4):.2f}  frees {1e3*(t6-t5):.2f} ms")
