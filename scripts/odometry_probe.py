"""Where the odometry loop's time goes: worker build time vs main-thread wait / new / align / free."""
import os, sys, time, threading, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import (BilateralFilter, Context, MsIcpParams, MultiscaleAlign, RangeImageBuilder, SyntheticDataset)

ctx = Context(0)
side = Context(0)
ds = SyntheticDataset(7, 20)
b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
sb = b.on_context(side)
params = MsIcpParams.default()
# sequential reference timings
pyr = [b.build_device(*ds.get(i)) for i in range(3)]
t = {"build": [], "new": [], "align": [], "free": []}
for rep in range(10):
    t0 = time.perf_counter(); p = b.build_device(*ds.get(3)); ctx.synchronize(); t1 = time.perf_counter()
    icp = MultiscaleAlign.new(ctx, params, pyr[0]); t2 = time.perf_counter()
    icp.align(pyr[1]); t3 = time.perf_counter()
    icp.free(); [lv.free() for lv in p]; t4 = time.perf_counter()
    t["build"].append(t1 - t0); t["new"].append(t2 - t1); t["align"].append(t3 - t2); t["free"].append(t4 - t3)
print("sequential ms:", {k: round(float(np.median(v)) * 1e3, 3) for k, v in t.items()})

# concurrent: worker builds continuously on the side context while the main thread aligns
stop = False
builds = []
def work():
    while not stop:
        t0 = time.perf_counter()
        p = sb.build_device(*ds.get(4)); side.synchronize()
        builds.append(time.perf_counter() - t0)
        [lv.free() for lv in p]
th = threading.Thread(target=work); th.start()
aligns = []
icp = MultiscaleAlign.new(ctx, params, pyr[0])
for rep in range(30):
    t0 = time.perf_counter(); icp.align(pyr[1]); aligns.append(time.perf_counter() - t0)
stop = True; th.join()
print("concurrent ms: align %.3f  build %.3f (n=%d)" % (np.median(aligns) * 1e3, np.median(builds) * 1e3, len(builds)))
