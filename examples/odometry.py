#!/usr/bin/env python3
"""Counterpart of the reference's `odometry` example (examples/src/bin/odometry.rs):
    python examples/odometry.py --format slamtb tests/golden/rgbd/sample1 [--max-frames N]
    python examples/odometry.py --format synthetic 7 --max-frames 20      (seed 7, 20 frames)
prints "Mean trajectory error: angle: X°, translation: Y" like the reference."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import Context, SlamTbDataset, SyntheticDataset, run_odometry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--format", choices=["slamtb", "synthetic"], required=True)
ap.add_argument("dataset", help="dataset directory (slamtb) or seed (synthetic)")
ap.add_argument("--max-frames", type=int, default=None)
args = ap.parse_args()
ctx = Context(0)
ds = SlamTbDataset.load(args.dataset) if args.format == "slamtb" else SyntheticDataset(int(args.dataset), args.max_frames or 20)
pred, metrics = run_odometry(ctx, ds, max_frames=args.max_frames)
print(f"Mean trajectory error: {metrics}")
