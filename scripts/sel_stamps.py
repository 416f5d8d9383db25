"""Where one sel_resolve_kernel block's time goes on the configs[2] cloud (a depth image's points: thousands share a
quantised z).  Needs scripts/build_stamps.sh; it loads scripts/stampbuild/libalign3d_hip_stamps.so.
`sel_stamps.py narrow`: block 0 of sel_narrow_select_kernel (the in-block selection kernel, diagnostics build) on the
uniform cloud: set-up, rounds and the rest of every level.
Stamps: (s_memrealtime at 100 MHz, tag, set size) — tags: 0 entry, 1 streaming min/max, 2 streaming round done, 3 set
copied to LDS, 4 LDS min/max, 5 LDS round done, 6 before the final ranking, 7 placed, 8 children's plans written."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
os.environ["A3D_LIBRARY"] = str(ROOT / "scripts" / "stampbuild" / "libalign3d_hip_stamps.so")
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from align3d_amd import Context, R3dTree  # noqa: E402

TAGS = {20: "narrow: entry", 21: "narrow: range in LDS", 22: "narrow: level set up, small cells sorted", 23: "narrow: round done", 24: "narrow: level done", 9: "after place kernel", 0: "entry", 1: "stream min/max", 2: "stream round", 3: "to LDS", 4: "lds min/max", 5: "lds round", 6: "pre-final", 7: "placed", 8: "plans"}


def main():
    ctx = Context(0)
    (tgt, _), _ = bench.pcl_clouds(ctx, 500_000)
    if os.environ.get("SEL_STAMPS_UNIFORM"):  # the uniform cloud of scripts/kd_probe.py instead
        class _U: points = np.ascontiguousarray(bench.synth.uniform01_f32(10, 3 * 500_000).reshape(-1, 3))
        tgt = _U
    lib = ctx.lib
    fn = lib.a3d_debug_sel_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_uint32, ctypes.c_uint32]
    fn.restype = ctypes.c_int
    out = (ctypes.c_ulonglong * 64)()
    pts = np.ascontiguousarray(tgt.points)
    R3dTree.new(ctx, pts).free()
    todo = [(2, j) for j in range(4)] + [(0, 0), (1, 0)]
    if len(sys.argv) > 1 and sys.argv[1] == "narrow":  # block 0 (and 100) of the in-block kernel, on the uniform cloud too
        todo = [(99, 0)]  # (the in-block selection kernel's block 0 stamps whatever level is asked for; 99 keeps the resolve blocks quiet)
        os.environ["A3D_KDTREE_SORTNET"] = "select"  # diagnostics build: the in-block levels by selection (the stamped kernel)
        sys.argv = sys.argv[:1]
        import bench as _b
        uni = np.ascontiguousarray(_b.synth.uniform01_f32(10, 3 * 500_000).reshape(-1, 3))
        for level, node in todo:
            fn(out, level, node)
            R3dTree.new(ctx, uni).free()
            fn(out, level, node)
            v = list(out)
            print(f"uniform cloud, block {node}")
            prev = v[0]
            for k in range(32):
                if v[2 * k] == 0:
                    break
                print(f"   {TAGS.get(v[2 * k + 1] >> 32):44s} {v[2 * k + 1] & 0xffffffff:5d}  +{(v[2 * k] - prev) / 100.0:7.2f} us   at {(v[2 * k] - v[0]) / 100.0:7.2f}")
                prev = v[2 * k]
    elif len(sys.argv) > 1:  # `sel_stamps.py LEVEL`: every node of that level, the slowest one printed in full
        level = int(sys.argv[1])
        todo = [(level, j) for j in range(1 << level)]
    rows = []
    for level, node in todo:
        fn(out, level, node)
        R3dTree.new(ctx, pts).free()
        fn(out, level, node)
        rows.append((level, node, list(out)))
    if len(sys.argv) > 1:
        for level, node, v in rows:
            last = max(k for k in range(32) if v[2 * k])
            print(f"level {level} node {node}: {(v[2 * last] - v[0]) / 100.0:7.2f} us, set of {v[1] & 0xffffffff}")
        rows = [max(rows, key=lambda r: r[2][2 * max(k for k in range(32) if r[2][2 * k])] - r[2][0])]
    for level, node, v in rows:
        print(f"level {level} node {node}")
        t0 = v[0]
        if v[60] and v[61]:  # fused launch: the node's first split block started / its last block took the ticket (before the resolve step's entry)
            print(f"   split launch: first block of the node started {(t0 - v[60]) / 100.0:.2f} us, last block took the ticket {(t0 - v[61]) / 100.0:.2f} us before the resolve step's entry")
        prev = t0
        for k in range(32):
            t, w = v[2 * k], v[2 * k + 1]
            if t == 0:
                break
            print(f"   {str(TAGS.get(w >> 32, w >> 32)):18s} c={w & 0xffffffff:7d}  +{(t - prev) / 100.0:7.2f} us   at {(t - t0) / 100.0:7.2f}")
            prev = t


if __name__ == "__main__":
    main()
