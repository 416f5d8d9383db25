"""Frame builder alone: live kernel time per frame (hipEvent brackets around each 16-frame chunk, as bench.py's
extra.frame_build.roofline measures it) and wall time per 64-frame build from page-locked host memory.
  python3 scripts/builder_probe.py [priority]     priority -1: a builder context as made beside an aligner (no colour fork)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from align3d_amd import Context
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = Context(0, priority=prio, library=os.environ.get("A3D_LIBRARY"))
r = bench.frame_build_roofline(ctx, 640, 480)
print(f"priority {prio}: kernels {r['kernel_us_per_frame']:.2f} us/frame {r['kernel_us_per_frame_stats']}  frac {r['frac']:.3f}  "
      f"algorithmic {r['algorithmic_bytes_per_frame'] / 1e6:.1f} MB/frame")
