#!/usr/bin/env python3
"""Counterpart of the reference's `odometry` example (examples/src/bin/odometry.rs):
    python examples/odometry.py --format tum    /data/rgbd_dataset_freiburg1_xyz [--max-frames N]
    python examples/odometry.py --format ilrgbd /data/indoor_lidar/apartment     [--max-frames N]
    python examples/odometry.py --format slamtb tests/golden/rgbd/sample1
    python examples/odometry.py --format synthetic 7 --max-frames 20      (seed 7, 20 frames)
    ... --batched [--window 64]: the recorded sequence as windows of consecutive frames, each built in one batched call
        and aligned as ONE batch (run_odometry_batched: same arithmetic per pair, several times the frame rate)
prints "Mean trajectory error: angle: X°, translation: Y" like the reference."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3d_amd import (Context, SubsetDataset, SyntheticDataset, load_dataset, run_odometry,  # noqa: E402
                         run_odometry_batched)

ap = argparse.ArgumentParser()
ap.add_argument("--format", choices=["ilrgbd", "tum", "slamtb", "synthetic"], required=True)
ap.add_argument("dataset", help="dataset directory, or the seed for --format synthetic")
ap.add_argument("--max-frames", type=int, default=None)
ap.add_argument("--batched", action="store_true", help="align windows of consecutive frames as one batch each")
ap.add_argument("--window", type=int, default=64, help="pairs per batch with --batched")
ap.add_argument("--in-flight", type=int, default=1,
                help="frame-by-frame loop with this many alignments running at once (each on its own stream; same poses)")
args = ap.parse_args()
ctx = Context(0)
if args.format == "synthetic":
    ds = SyntheticDataset(int(args.dataset), args.max_frames or 20)
else:
    ds = load_dataset(args.format, args.dataset)
    if args.max_frames is not None:  # odometry.rs:32-34
        ds = SubsetDataset.new(ds, range(min(args.max_frames, ds.len())))
pred, metrics = (run_odometry_batched(ctx, ds, window=args.window) if args.batched
                 else run_odometry(ctx, ds, in_flight=args.in_flight))
print(f"Mean trajectory error: {metrics}")
