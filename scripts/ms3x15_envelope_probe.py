"""ms3x15 on the headline's own distinct pairs (stream 1000, frames 2p -> 2p + 1): GPU pose against the oracle, and the
oracle against itself under other chunk-merge orders (the orders rayon's par_bridge() may deliver, image_icp.rs:96,
143-148).  Shows whether a GPU-vs-oracle difference above 1e-4 on some pair is the pair's own sensitivity
(IcpParams::default() is not contractive everywhere) or the kernels'."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
from align3d_amd import Context, IcpParams, MsIcpParams, MultiscaleAlignBatch
from bench import build_stream_pyramids

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = Context(0)
pyr, _, _ = build_stream_pyramids(ctx, 1000, 2 * P, 640, 480)
prm = MsIcpParams.repeat(3, IcpParams.default())
pairs = [(2 * p, 2 * p + 1) for p in range(P)]
batch = MultiscaleAlignBatch(ctx, prm, [pyr[a] for a, _ in pairs], [pyr[b] for _, b in pairs])
poses, status = batch.align()
assert not status.any()

def frame(dev_level):
    ri = dev_level.download(colors=False)
    k = ri.intrinsics
    return O.Frame(ri.points, ri.mask, k.fx, k.fy, k.cx, k.cy, ri.normals, ri.intensities, ri.intensity_map)

for p, (a, b) in enumerate(pairs):
    ta, tb = [frame(lv) for lv in pyr[a]], [frame(lv) for lv in pyr[b]]
    runs = []
    for seed in range(seeds):
        O.set_chunk_merge_order(seed)
        st, T = O.multiscale_align(prm.to_c_array(), 3, ta, tb, threads=8)
        assert st == 0
        runs.append(T)
    O.set_chunk_merge_order(0)
    gpu = [O.transform_metrics(poses[p].to_c(), r) for r in runs]
    orc = [O.transform_metrics(runs[0], r) for r in runs[1:]]
    print(f"pair {p} (frames {a}->{b}): GPU vs oracle runs: angle {min(g[0] for g in gpu):.2e}..{max(g[0] for g in gpu):.2e} rad, "
          f"trans {min(g[1] for g in gpu):.2e}..{max(g[1] for g in gpu):.2e} m | oracle vs oracle (other merge orders): "
          f"angle up to {max(abs(o[0]) for o in orc):.2e}, trans up to {max(o[1] for o in orc):.2e}", flush=True)
