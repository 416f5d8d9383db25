"""Where block 0 of sel_narrow_kernel (the kd-tree selection build's in-block levels, the product's LDS network) spends its
time on the uniform 500 k cloud: s_memtime stamps (shader clock).  Stamp build:
  bash scripts/build_stamps.sh -DA3D_NARROW_NET_STAMPS      (then python3 scripts/narrow_stamps.py)"""
import ctypes, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
os.environ["A3D_LIBRARY"] = str(ROOT / "scripts" / "stampbuild" / "libalign3d_hip_stamps.so")
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from align3d_amd import Context, R3dTree  # noqa: E402
TAGS = {30: "entry", 31: "range in LDS (points)", 32: "level set up (level)", 33: "network done (cell size)", 34: "records moved (level)",
        35: "entry level: equal-key runs ordered (level)", 36: "all levels done", 37: "leaves + slot_of_point + padding written",
        38: "entry level: neighbours compared (any tie)", 39: "entry level: short runs ranked (long runs)", 43: "entry level: tied points in place (moved)",
        40: "first phases inside the thread done (kk)", 41: "distances >= 256 through LDS done (kk)", 42: "merge phase done (kk)"}
ctx = Context(0)
fn = ctx.lib.a3d_debug_sel_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_uint32, ctypes.c_uint32]
fn.restype = ctypes.c_int
out = (ctypes.c_ulonglong * 64)()
uni = np.ascontiguousarray(bench.synth.uniform01_f32(10, 3 * 500_000).reshape(-1, 3))
if os.environ.get("NARROW_STAMPS_PCL"):  # the depth-image cloud of configs[2] instead (long runs of equal z: the 128-bit network)
    (tgt, _), _ = bench.pcl_clouds(ctx, 500_000)
    uni = np.ascontiguousarray(tgt.points)
R3dTree.new(ctx, uni).free()
runs = []
for rep in range(5):
    fn(out, 99, 0)
    R3dTree.new(ctx, uni).free()
    fn(out, 99, 0)
    runs.append(list(out))
v = runs[-1]
print("# sel_narrow_kernel<2048, REGS>, block 0 (the product: words in registers; A3D_KDTREE_SORTNET=lds: in LDS), uniform 500 k cloud: shader-clock cycles (last of 5 builds)")
prev = v[0]
for k in range(32):
    if v[2 * k] == 0:
        break
    print(f"   {TAGS.get(v[2 * k + 1] >> 32, '?'):48s} {v[2 * k + 1] & 0xffffffff:6d}  +{v[2 * k] - prev:8d} cycles   at {v[2 * k] - v[0]:8d}")
    prev = v[2 * k]
