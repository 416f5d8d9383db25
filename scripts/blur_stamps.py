"""Where a tile of blur_fused_kernel's walk spends its time: s_memtime stamps (shader clock) of wave 0 of block 0 of frame 0,
the first 8 tiles of its walk (stamp build: VARIANT=blurstamps bash scripts/build_frame_variant.sh -DA3D_BLUR_STAMPS).
  python3 scripts/blur_stamps.py [frames per build]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["A3D_LIBRARY"] = os.path.join(ROOT, "scripts", "variantbuild_blurstamps", "libalign3d_hip_variant.so")
sys.path.insert(0, ROOT)
import numpy as np
from align3d_amd import BilateralFilter, Context, RangeImageBuilder, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames, _ = synth.frame_stream(4242, F, 640, 480)
cam = synth.camera(640, 480)
ctx = Context(0, priority=-1)
b = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
rows = []
for rep in range(6):
    st = (C.c_ulonglong * 80)()
    ctx.lib.a3d_debug_blur_stamps(st)  # (resets the tile counter)
    for p in b.build_many(cam, frames, synth.DEPTH_SCALE):
        for lv in p:
            lv.free()
    assert ctx.lib.a3d_debug_blur_stamps(st) == 0
    if rep == 0:
        continue
    for t in range(1, 7):  # (the first tile of a walk has no predecessor; the later ones are the steady state)
        s8 = [st[8 * t + k] for k in range(8)]
        end, prev_end = st[64 + t], st[64 + t - 1]
        rows.append([s8[0] - prev_end] + [s8[k + 1] - s8[k] for k in range(7)] + [end - s8[7], end - prev_end])
r = np.median(np.array(rows, np.float64), axis=0)
names = ["previous barrier -> tile entered (next window's 16 loads issued, tile decoded)", "row + channel passes (axis 0, axis 2)",
         "LDS writes + barrier", "LDS reads + conversions", "column passes (axis 1)", "12 quotients", "next window taken (wait)",
         "12 stores issued", "last barrier", "WHOLE TILE"]
print(f"# blur_fused_kernel, wave 0 of block 0, {F} frames per launch: shader-clock cycles per tile (median of {len(rows)} tiles)")
for n, v in zip(names, r):
    print(f"{v:9.0f}  {n}")
