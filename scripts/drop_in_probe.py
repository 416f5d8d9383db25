"""N drop-in calls MultiscaleAlign::align(&[RangeImage]) from host pyramids (upload + align + free each).  Run it under
`rocprofv3 --hip-trace --stats` with two values of N: the hipMalloc / hipFree counts must not grow with N."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlign
from bench import build_stream_pyramids
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 1000, 2, 640, 480)
host = [lv.download(colors=False) for lv in pyr[1]]
ms = MultiscaleAlign.new(ctx, MsIcpParams.repeat(3, IcpParams.default()), pyr[0])
t = []
for _ in range(n):
    for h in host:
        if h._device is not None:
            h._device.free()
        h._device = None
    t0 = time.perf_counter()
    T = ms.align(host)
    t.append((time.perf_counter() - t0) * 1e3)
print(f"{n} drop-in calls: median {np.median(t):.3f} ms, first {t[0]:.3f} ms")
