"""compute_normals batched: us per launch and fraction of 8 TB/s (25 B per pixel) for 1..64 resident 640x480 frames.
Diagnostics build: A3D_NORMALS_SHAPE selects the tile shape (0: 32x8x1, 1: 64x16x4, 2: 64x32x8, 3: 128x8x4, 4: 64x8x2)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from align3d_amd import Context, compute_normals_batch, _abi
ctx = Context(0, library=os.environ.get("A3D_LIBRARY") or (_abi.DIAG_LIB_PATH if os.environ.get("A3D_NORMALS_SHAPE") else None))
pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2, 640, 480)
host = pyr[0][0].download()
devs = []
for _ in range(64):
    r = copy.copy(host); r._device = None; devs.append(r.device(ctx))
for n in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64]:
    for _ in range(3): compute_normals_batch(devs[:n])
    ctx.synchronize(); per = []
    for _ in range(9):
        ctx.timer_start()
        for _ in range(10): compute_normals_batch(devs[:n])
        per.append(ctx.timer_stop() / 10)
    ms = float(np.median(per))
    print(f"shape {os.environ.get('A3D_NORMALS_SHAPE', 'default')} {n} frames: {ms * 1e3:.1f} us -> {n * 25 * 640 * 480 / (ms * 1e-3) / 1e9 / 8000:.3f} of 8 TB/s", flush=True)
