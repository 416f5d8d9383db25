"""Single-pair MultiscaleAlign latency (configs[1]) for ms3x15 and msdefault (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlign
from bench import build_stream_pyramids
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 1000, 2, 640, 480)
for name, prm in (("ms3x15", MsIcpParams.repeat(3, IcpParams.default())), ("msdefault", MsIcpParams.default())):
    ms = MultiscaleAlign.new(ctx, prm, pyr[0])
    for _ in range(3):
        ms.align(pyr[1])
    t0 = time.perf_counter()
    for _ in range(20):
        ms.align(pyr[1])
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per alignment  env={ {k: v for k, v in os.environ.items() if k.startswith('A3D_')} }")
