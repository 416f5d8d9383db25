#!/usr/bin/env python3
"""Benchmark of the ICP hot path on MI355X.

One "step" = one pass of MultiscaleAlign over a batch of independent synthetic 640x480 frame pairs
that are already resident in HBM (`ms3x15`: 3 pyramid levels, IcpParams::default() = 15 iterations
per level — the shape BASELINE.json's metric is quoted on).  Per GPU the batch is 64 pairs (the
per-GPU shard of configs[4]: 512 pairs over 8 GPUs).  With N > 1 ranks ONE global list of 64 N pairs is
sharded in contiguous blocks (align3d_amd.distributed.shard_range), every rank aligns its own block and one
RCCL all-gather collects the 4x4 poses in global pair order (gather_poses) — weak scaling.

    python bench.py --gpus 1 --steps 600 --warmup 20
    python bench.py --gpus N ...            (no WORLD_SIZE in the env: starts N fresh rank processes itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (WORLD_SIZE must equal N)

Prints ONE JSON line on rank 0 (contract in the round instructions) with `roofline` for the dominant
kernel (the per-pixel kernel) and `cpu_baseline` (the CPU oracle timed on a bounded sample); `extra` holds the
secondary workloads (kd-tree, Icp, frame preparation, odometry, streaming), each with its own roofline and CPU
baseline, and the min / median / max of repeated timings."""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from align3d_amd import (BilateralFilter, Context, IcpParams, MsIcpParams, MultiscaleAlign,  # noqa: E402
                         MultiscaleAlignBatch, R3dTree, RangeImageBuilder, synth)
from align3d_amd.distributed import gather_poses, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PAIRS_PER_STREAM = 64  # the global pair list is a concatenation of 65-frame streams (seed 1000 + stream)


# algorithmic bytes of one ImageIcp iteration at each level of a 640x480 pyramid:
# source 14 B/px + target 25 B/px + the (H+2)(W+2) f32 intensity map.
def level_bytes(w, h):
    return 39 * w * h + 4 * (w + 2) * (h + 2)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def stats(samples):
    a = np.asarray(samples, np.float64)
    return {"min": float(a.min()), "median": float(np.median(a)), "max": float(a.max()), "repeats": int(a.size)}


def roofline(alg_bytes, ms, traffic=None, traffic_src=None, **more):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
         "algorithmic_bytes_per_launch": alg_bytes}
    r.update(more)
    return r


def measured_traffic(name, **match):
    """HBM bytes per launch of a workload's dominant kernel from the committed PMC passes (scripts/traffic_pmc.sh;
    counters cannot be read from inside an unprofiled run).  A profile counts only when EVERY key of `match` is recorded
    in it with that value (workload shape and, where a kernel has had more than one, its `layout`); among those the
    highest round wins.  File names play no part.  Otherwise (None, None)."""
    import re

    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", f"round*_{name}_traffic*.json")):
        m = re.match(rf"round(\d+)_{name}_traffic", os.path.basename(f))
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        if m is None or any(rec.get(k) != v for k, v in match.items()):
            continue
        if best is None or int(m.group(1)) > best[0]:
            best = (int(m.group(1)), float(rec["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT))
    return (best[1], best[2]) if best else (None, None)


def build_stream_pyramids(ctx, seed, n_frames, width, height, first=0, total=None):
    """Synthetic frame stream -> resident pyramids, built on the device in batches (bilateral filter,
    back-projection, normals, pyramid, luma, intensity maps: a3d_range_image_build_pyramids).  Returns the device
    pyramids, the ground-truth camera poses and the per-frame build time (depth + RGB upload included)."""
    frames, poses = synth.frame_stream(seed, total or n_frames, width, height, first=first, count=n_frames)
    all_d, all_c = np.stack([d for d, _ in frames]), np.stack([c for _, c in frames])  # one buffer per array
    frames = [(all_d[i], all_c[i]) for i in range(n_frames)]
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    cam = synth.camera(width, height)
    for lv in builder.build(cam, *frames[0], synth.DEPTH_SCALE):  # first use of the context: scratch, tap tables
        lv.free()
    t0 = time.perf_counter()
    pyramids = builder.build_many(cam, frames, synth.DEPTH_SCALE)
    build_ms = (time.perf_counter() - t0) / n_frames * 1e3
    return pyramids, poses, build_ms


def kdtree_bench(ctx, n=500_000, reps=20, groups=7):
    db = synth.uniform01_f32(10, 3 * n).reshape(n, 3)
    q = synth.uniform01_f32(11, 3 * n).reshape(n, 3)
    # the first call pays the one-off code-object load of the build's kernels; a new context also runs its first four builds with
    # the placement launches for oversized median buckets (a3d_kdtree_build_path = 3) and drops them when none showed up: the
    # timed builds below are the settled ones of a context that keeps building this kind of cloud (path reported beside them)
    for _ in range(5):
        R3dTree.new(ctx, db).free()
    builds = []
    for _ in range(3):
        t0 = time.perf_counter()
        tree = R3dTree.new(ctx, db)
        builds.append((time.perf_counter() - t0) * 1e3)
        tree.free()
    # R3dTree::new over points that are already resident (a3d_kdtree_new_device): wall time of the call (allocation of the
    # tree's arrays + 11 launches + the flag read-back) and the device time of its launches (events on the stream)
    d_db = ctx.to_device(db)
    R3dTree.new_device(ctx, d_db, n).free()
    dev_wall, dev_kernels = [], []
    for _ in range(9):
        t0 = time.perf_counter()
        tree = R3dTree.new_device(ctx, d_db, n)
        dev_wall.append((time.perf_counter() - t0) * 1e3)
        dev_kernels.append(tree.build_ms())
        path_seen = tree.build_path()
        tree.free()
    ctx.free(d_db)
    tree = R3dTree.new(ctx, db)
    d_q = ctx.to_device(q)
    d_i, d_d = ctx.malloc(4 * n), ctx.malloc(4 * n)
    for _ in range(3):
        tree.nearest_device(d_q, n, d_i, d_d)
    per = []
    for _ in range(groups):
        ctx.timer_start()
        for _ in range(reps):
            tree.nearest_device(d_q, n, d_i, d_d)
        per.append(ctx.timer_stop() / reps)
    ms = float(np.median(per))
    for p in (d_q, d_i, d_d):
        ctx.free(p)
    leaves, internal, depth = tree.stats()
    tree.free()
    alg_bytes = 216 * n + 4 * internal  # 216 B/query + the split table once
    traffic, src = measured_traffic("kdtree", queries=n, points=n)
    return {
        "metric": "kdtree 500k queries/s (500k database, uniform [0,1)^3)",
        "value": n / (ms * 1e-3), "unit": "queries/s", "ms_per_500k_queries": ms,
        "ms_per_500k_queries_stats": stats(per),
        "build_ms_incl_pcie": float(np.median(builds)),  # R3dTree::new from host points: upload + device build
        # benches/bench_kdtree.rs:25-27 "R3dTree creation".  Algorithmic bytes: 20 B per point and level (12 B of
        # coordinates read, a 4-byte index read and written: what a level-by-level build moves at least) x `depth` levels
        "build": {"device_ms": float(np.median(dev_wall)), "device_ms_stats": stats(dev_wall),
                  "kernel_ms": float(np.median(dev_kernels)), "kernel_ms_stats": stats(dev_kernels),
                  # pack (+ the workspace's zeroing), hist0 (+ the root's plan), one split launch per wide level (+ the node's resolve step), narrow —
                  # and, on build path 3 (a context that has seen an oversized median bucket), sel_place and sel_resolve as
                  # launches of their own on each of the first place_levels = 7 wide levels
                  "launches": (lambda wide: 3 + wide + (2 * min(wide, 7) if int(path_seen) == 3 else 0))(
                      max(0, int(np.ceil(np.log2(max(1, n / 2048)))))),
                  "build_path": int(path_seen),
                  "roofline": roofline(20 * n * depth, float(np.median(dev_kernels)), kernel="the selection build's launches (kdtree_select.hip), first to last",
                                       levels=int(depth))},
        "roofline": roofline(alg_bytes, ms, traffic, src, kernel="kdtree_nearest_kernel",
                             binding_resource="L2 / Infinity Cache gather rate and latency (HBM-nominal fraction: the 8 MB leaf "
                                              "table and the 6 MB query array stay cache-resident between launches)"),
    }


def pcl_clouds(ctx, n=500_000):
    from align3d_amd import PointCloud

    frames, poses = synth.frame_stream(7, 2, 880, 660)
    cam = synth.camera(880, 660)
    clouds = []
    for d, rgb in frames:
        lv = RangeImageBuilder(ctx).pyramid_levels(1).with_intensity(False).build(cam, d, rgb, synth.DEPTH_SCALE)[0]
        pc = PointCloud.from_range_image(lv.download(intensity=False))
        lv.free()
        clouds.append(PointCloud(pc.points[:n], pc.normals[:n]))
    return clouds, poses


def pcl_icp_bench(ctx, n=500_000):
    """configs[2]: Icp (kd-tree point-to-plane) on 500k target x 500k source points, IcpParams::default()."""
    from align3d_amd import Icp

    (tgt, src), poses = pcl_clouds(ctx, n)
    Icp.new(ctx, IcpParams.default(), tgt).free()  # first use: code objects, the context's kd-tree scratch region
    news = []
    for _ in range(3):
        t0 = time.perf_counter()
        icp = Icp.new(ctx, IcpParams.default(), tgt)
        news.append((time.perf_counter() - t0) * 1e3)
        if len(news) < 3:
            icp.free()
    build_ms = float(np.median(news))
    icp.align(src)
    times, walls = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        T = icp.align(src)
        walls.append((time.perf_counter() - t0) * 1e3)
        times.append(icp.last_device_ms())
    ms = float(np.median(times))
    iters = 15
    alg = 252 * src.len()  # 252 B per source point per iteration
    # the same with both clouds resident (a3d_pcl_icp_new_device / _align_device): Icp::new + align end to end, no PCIe
    from align3d_amd import DevicePointCloud
    dt, dsrc = DevicePointCloud(ctx, tgt), DevicePointCloud(ctx, src)
    Icp.new(ctx, IcpParams.default(), dt).free()
    d_new, d_align, d_both = [], [], []
    for _ in range(7):
        t0 = time.perf_counter()
        icp_d = Icp.new(ctx, IcpParams.default(), dt)
        t1 = time.perf_counter()
        Td = icp_d.align(dsrc)
        t2 = time.perf_counter()
        d_new.append((t1 - t0) * 1e3), d_align.append((t2 - t1) * 1e3), d_both.append((t2 - t0) * 1e3)
        icp_d.free()
    same_bits = bool(np.array_equal(np.concatenate([Td.t, Td.q]).view(np.uint32), np.concatenate([T.t, T.q]).view(np.uint32)))
    # the kd-tree build inside Icp::new on THIS cloud (a depth image's points: a wall facing the camera is tens of
    # thousands of equal z, which the uniform cloud of the kd-tree benchmark never has)
    kd_ms, kd_path = [], 0
    for _ in range(7):
        tr = R3dTree.new_device(ctx, dt.d_points, tgt.len())
        kd_ms.append(tr.build_ms())
        kd_path = tr.build_path()
        tr.free()
    dt.free(), dsrc.free()
    gt = synth.relative_pose(poses[0], poses[1])
    dm = np.linalg.inv(gt) @ T.matrix().astype(np.float64)
    icp.free()
    traffic, tsrc = measured_traffic("pcl_icp", source_points=src.len(), target_points=tgt.len())
    return {
        "workload": f"Icp::align, {tgt.len()} target x {src.len()} source points, 15 iterations (configs[2])",
        "device_ms_per_align": ms, "device_ms_per_align_stats": stats(times), "aligns_per_s": 1e3 / ms,
        "us_per_iteration": ms * 1e3 / iters, "icp_new_ms_incl_pcie": build_ms,
        "icp_new_device_ms": float(np.median(d_new)), "align_device_wall_ms": float(np.median(d_align)),
        "new_plus_align_device_ms": float(np.median(d_both)), "new_plus_align_device_ms_stats": stats(d_both),
        "device_forms_give_the_same_pose_bits": same_bits,
        "gpu_pose_t_q": [float(x) for x in list(T.t) + list(T.q)],
        "kd_build_kernel_ms": float(np.median(kd_ms)), "kd_build_kernel_ms_stats": stats(kd_ms),
        "kd_build_path": int(kd_path),  # 3: with the chip-wide placement of oversized median buckets (a3d_kdtree_build_path)
        "align_wall_ms_incl_pcie": float(np.median(walls)),  # Icp::align from host clouds: 12 MB upload + 15 iterations
        "error_vs_synthetic_gt": {"angle_rad": float(np.arccos(np.clip((np.trace(dm[:3, :3]) - 1) / 2, -1, 1))),
                                  "translation_m": float(np.linalg.norm(dm[:3, 3]))},
        "roofline": roofline(alg, ms / iters, traffic, tsrc, kernel="pcl_icp_head_kernel", launches_per_align=iters,
                             binding_resource="L2 / Infinity Cache gather rate and latency (HBM-nominal fraction: fabric "
                                              "traffic is well below the algorithmic bytes, neighbouring queries share leaves)"),
    }, (tgt, src)


def frame_prep_bench(ctx, host_pyramid_level0, depth_u16):
    """compute_normals (device-resident) and the bilateral filter (host in/out API, so the figure includes the
    PCIe copies of the 0.6 MB image) on one 640x480 frame."""
    import copy

    ri = copy.copy(host_pyramid_level0)
    ri._device = None
    dev = ri.device(ctx)
    for _ in range(3):
        dev.compute_normals()
    ctx.synchronize()
    per = []
    for _ in range(7):
        ctx.timer_start()
        for _ in range(20):
            dev.compute_normals()
        per.append(ctx.timer_stop() / 20)
    n_ms = float(np.median(per))
    # the batched form: 64 resident frames per launch (a3d_range_image_compute_normals_batch)
    from align3d_amd import compute_normals_batch
    devs = [dev]
    for _ in range(63):
        r2 = copy.copy(host_pyramid_level0)
        r2._device = None
        devs.append(r2.device(ctx))
    for _ in range(3):
        compute_normals_batch(devs)
    ctx.synchronize()
    bper = []
    for _ in range(7):
        ctx.timer_start()
        for _ in range(10):
            compute_normals_batch(devs)
        bper.append(ctx.timer_stop() / 10)
    nb_ms = float(np.median(bper))
    for d in devs:
        d.free()
    f = BilateralFilter.default()
    f.filter(ctx, depth_u16)
    b = []
    for _ in range(7):
        t0 = time.perf_counter()
        f.filter(ctx, depth_u16)
        b.append((time.perf_counter() - t0) * 1e3)
    b_ms = float(np.median(b))
    n_px = depth_u16.size
    cells = int(np.prod(f.last_grid_dims))
    return {
        "compute_normals_ms": n_ms, "compute_normals_ms_stats": stats(per),
        "compute_normals_roofline": roofline(25 * n_px, n_ms, kernel="compute_normals_kernel (one frame per launch)"),
        "compute_normals_batch_of_64_ms": nb_ms, "compute_normals_batch_of_64_ms_stats": stats(bper),
        "compute_normals_batch_roofline": roofline(64 * 25 * n_px, nb_ms, *measured_traffic("normals", frames=64, pixels=int(n_px), layout="xcd_contiguous"),
                                                   kernel="compute_normals_kernel (64 frames per launch)"),
        "bilateral_filter_ms_host_to_host": b_ms, "bilateral_filter_ms_stats": stats(b),
        "bilateral_grid_dims": list(f.last_grid_dims),
        "bilateral_note": "host-in / host-out call: the time includes two PCIe copies of the 0.6 MB image, so no roofline "
                          "is quoted for it; the filter's kernels are part of extra.frame_build.roofline",
        "bilateral_grid_cells": cells,
    }


def bilateral_device_bench(ctx, W, H, n_images=32, reps=7):
    """benches/bench_bilateral.rs's shape without PCIe: BilateralFilter::default() on `n_images` synthetic
    depth images resident in HBM, u16 in -> u16 out (a3d_bilateral_filter_u16_device).  Device time of the call's launch
    sequence by hipEvents; bytes as DESIGN §4 counts the filter: per image the depth read twice (min / max, splat) and once
    more by the slice, the result written, 4 B per packed cell and 8 B per blurred cell over the marked tiles' 12^3 cells,
    and the slice's eight 8-byte gathers per pixel counted once per distinct cell (= the blurred cells again)."""
    frames, _ = synth.frame_stream(4242, n_images, W, H)
    depth = np.ascontiguousarray(np.stack([d for d, _ in frames]), np.uint16)
    f = BilateralFilter.default()
    d_in = ctx.to_device(depth)
    d_out = ctx.malloc(depth.nbytes)
    try:
        f.filter_device(ctx, d_in, n_images, W, H, d_out)  # first use: grid scratch sized
        out = np.empty_like(depth)
        ctx.to_host(d_out, out)
        same = bool(np.array_equal(out[0], f.filter(ctx, depth[0])))  # (the host-pointer form: the tests' bit-exact path)
        ctx.set_build_profiling(True)
        ms, wall = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            f.filter_device(ctx, d_in, n_images, W, H, d_out)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(ctx.last_build_kernel_ms())
        ctx.set_build_profiling(False)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    # marked tiles of the last call's images (the blur's own statistics), as frame_build_roofline counts them
    st = ctx.last_build_stats()
    us = float(np.median(ms)) * 1e3 / n_images
    px = W * H
    tiles = st["marked_tiles"] / st["frames"] if st["frames"] else None
    alg = 4 * 2 * px + ((4 + 8 + 8) * 1728 * tiles if tiles else 0)
    return {"workload": f"BilateralFilter::default() on {n_images} resident {W}x{H} u16 depth images per call (bench_bilateral.rs shape, no PCIe)",
            "kernel_us_per_image": us, "kernel_us_per_image_stats": stats([m * 1e3 / n_images for m in ms]),
            "wall_us_per_image": float(np.median(wall)) * 1e3 / n_images, "images_per_s": 1e6 / us,
            "equals_host_pointer_form_bit_for_bit": same, "marked_tiles_per_image": tiles,
            "algorithmic_bytes_per_image": alg,
            "roofline": roofline(alg, us * 1e-3, kernel="minmax + dims + splat + blur_fused + slice + unsplat (six launches)",
                                 binding_resource="VALU issue of blur_fused and the latency of six dependent launches")}


def frame_build_bench(ctx, n_frames, W, H):
    """RangeImageBuilder::build for a stream of n_frames frames in one call, from page-locked host memory with the
    frames back to back in one buffer per array (a capture ring): ms per frame including the PCIe upload."""
    frames, _ = synth.frame_stream(4242, n_frames, W, H)
    all_d, all_c = ctx.pinned_empty((n_frames, H, W), np.uint16), ctx.pinned_empty((n_frames, H, W, 3), np.uint8)
    for i, (d, c) in enumerate(frames):
        all_d[i], all_c[i] = d, c
    frames = [(all_d[i], all_c[i]) for i in range(n_frames)]
    cam = synth.camera(W, H)
    out = {}
    for label, builder in (("bilateral_on", RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())),
                           ("bilateral_off", RangeImageBuilder(ctx))):
        per = []
        for rep in range(6):
            t0 = time.perf_counter()
            pyr = builder.build_many(cam, frames, synth.DEPTH_SCALE)
            if rep:
                per.append((time.perf_counter() - t0) / n_frames * 1e3)
            for lv in (lv for p in pyr for lv in p):
                lv.free()
        out[label] = {"ms_per_frame": float(np.median(per)), "ms_per_frame_stats": stats(per),
                      "frames_per_s": 1e3 / float(np.median(per))}
    out["workload"] = f"{n_frames} frames {W}x{H} per a3d_range_image_build_pyramids call, 3 levels, normals + intensity maps"
    return out


def live_pixel_fractions(ctx, params, target_pyramid, source_pyramid, pose):
    """Per level: the share of source pixels whose geometric term is accumulated at `pose` (count_g / N from one
    a3d_image_icp_accumulate pass): what fraction of the pixels the kernel streams also reaches its Jacobian stage."""
    from align3d_amd import ImageIcp

    out = []
    for l in range(len(params)):
        g, _ = ImageIcp.new(ctx, params[l], target_pyramid[l]).accumulate(source_pyramid[l], pose)
        h, w = source_pyramid[l].shape
        out.append(g["count"] / float(h * w))
    return out


def pinned_tiling_bench(ctx, params, targets, sources, lone_ms_default, step_ms_default, tiles=24):
    """What a3d_context_set_tiling(24) costs: with the cut of every (pair, level) into blocks pinned, a
    pair's pose is the same bits alone and in any batch; 24 blocks per pair is the throughput tiling's own level-0 cut of a
    64-pair batch, so the batch pays only for the coarse levels' fixed cut, a lone pair for running on 24 blocks."""
    P = len(targets)
    ctx.set_tiling(tiles)
    try:
        batch = MultiscaleAlignBatch(ctx, params, targets, sources)
        for _ in range(5):
            batch.enqueue()
        ctx.synchronize()
        reps = []
        for _ in range(10):
            t1 = time.perf_counter()
            for _ in range(20):
                batch.enqueue()
            ctx.synchronize()
            reps.append((time.perf_counter() - t1) / 20 * 1e3)
        poses, status = batch.align()
        batch.free()
        ms1 = MultiscaleAlign.new(ctx, params, targets[5 % P])
        alone = ms1.align(sources[5 % P])
        lat = []
        for _ in range(15):
            t1 = time.perf_counter()
            ms1.align(sources[5 % P])
            lat.append((time.perf_counter() - t1) * 1e3)
    finally:
        ctx.set_tiling(0)
    a = np.concatenate([alone.t, alone.q]).view(np.uint32)
    b = np.concatenate([poses[5 % P].t, poses[5 % P].q]).view(np.uint32)
    step = float(np.median(reps))
    # (second value: the poses, for the parity leg — cpu_baseline_main checks the same 64 pairs under pinned tiling)
    return {"tiles_per_pair": tiles, "ms_per_step": step, "ms_per_step_stats": stats(reps), "pairs_per_s": P / step * 1e3,
            "cost_vs_throughput_tiling": step / step_ms_default - 1.0,
            "lone_pair_latency_ms": float(np.median(lat)), "lone_pair_latency_ms_throughput_tiling": lone_ms_default,
            "pair_alone_equals_pair_in_batch_bit_for_bit": bool(np.array_equal(a, b)), "failed_pairs": int(np.count_nonzero(status))}, poses


def drop_in_bench(ctx, params, target_pyramid, source_pyramid, reps=15):
    """The call the reference makes, literally: MultiscaleAlign::align(&[RangeImage]) (src/icp/multiscale.rs:51) with
    the SOURCE pyramid in host memory — uploaded (a3d_range_image_upload_pyramid: one pooled arena, one copy per
    array), aligned, freed, every call — and the target pyramid resident (MultiscaleAlign::new borrowed it once).
    Beside it ImageIcp::align (image_icp.rs:43) on level 0 alone from a host RangeImage (bench10 shape).  Pageable
    numpy arrays and page-locked ones (a3d_host_alloc)."""
    from align3d_amd import ImageIcp, RangeImage

    def host_copy(dev_level, pinned):
        ri = dev_level.download(colors=False)
        if pinned:  # the same arrays in page-locked memory (ascontiguousarray inside RangeImage does not copy them)
            def pin(a):
                b = ctx.pinned_empty(a.shape, a.dtype)
                b[...] = a
                return b
            ri = RangeImage(pin(ri.points), pin(ri.mask), ri.intrinsics, normals=pin(ri.normals),
                            intensities=pin(ri.intensities), intensity_map=pin(ri.intensity_map))
        ri._device = None
        return ri

    out = {}
    ms = MultiscaleAlign.new(ctx, params, target_pyramid)
    icp0 = ImageIcp.new(ctx, IcpParams(max_iterations=10), target_pyramid[0])
    resident = ms.align(source_pyramid)
    for label, pinned in (("pageable", False), ("page_locked", True)):
        host = [host_copy(lv, pinned) for lv in source_pyramid]
        nbytes = sum(a.nbytes for h in host for a in (h.points, h.mask, h.normals, h.intensities, h.intensity_map))

        def forget():
            for h in host:
                if h._device is not None:
                    h._device.free()
                h._device = None

        def timed(fn):
            for _ in range(3):
                forget()
                fn()
            per = []
            for _ in range(reps):
                forget()
                t0 = time.perf_counter()
                T = fn()
                per.append((time.perf_counter() - t0) * 1e3)
            forget()
            return T, per

        T, per = timed(lambda: ms.align(host))
        same = bool(np.array_equal(T.matrix(), resident.matrix()))
        _, per0 = timed(lambda: icp0.align(host[0]))
        out[label] = {"ms3x15_ms": float(np.median(per)), "ms3x15_ms_stats": stats(per),
                      "equals_device_resident_result": same, "uploaded_bytes_per_call": int(nbytes),
                      "image_icp_bench10_ms": float(np.median(per0)), "image_icp_bench10_ms_stats": stats(per0)}
    ms.free()
    out["workload"] = ("MultiscaleAlign::align(&[RangeImage]) with the 3-level 640x480 source pyramid in host memory "
                       "(upload + 45 iterations + free per call), target pyramid resident; ImageIcp::align likewise on "
                       "level 0 with 10 iterations")
    return out


def bench_icp_shape(ctx):
    """benches/bench_icp.rs:9-39, the reference's own Icp bench: sample1 frames 0 (target) and 5 (source) as point
    clouds (RangeImage::from_rgbd_frame + compute_normals + PointCloud::from, no bilateral filter), Icp::new(IcpParams {
    max_iterations: 10, ..default }, &pcl0).align(&pcl1).  Needs the fixture frames (tests/golden)."""
    from align3d_amd import Icp, PointCloud, SlamTbDataset

    real = os.path.join(ROOT, "tests", "golden", "rgbd", "sample1")
    if not os.path.isdir(real):
        return None, None
    ds = SlamTbDataset.load(real)
    clouds = []
    for i in (0, 5):
        cam, depth, rgb, depth_scale = ds.get(i)
        lv = RangeImageBuilder(ctx).pyramid_levels(1).with_intensity(False).build(cam, depth, rgb, depth_scale)[0]
        clouds.append(PointCloud.from_range_image(lv.download(intensity=False)))
        lv.free()
    tgt, src = clouds
    prm = IcpParams(max_iterations=10)
    Icp.new(ctx, prm, tgt).free()
    news = []
    for _ in range(3):
        t0 = time.perf_counter()
        icp = Icp.new(ctx, prm, tgt)
        news.append((time.perf_counter() - t0) * 1e3)
        if len(news) < 3:
            icp.free()
    icp.align(src)
    dev, wall = [], []
    for _ in range(9):
        t0 = time.perf_counter()
        T = icp.align(src)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(icp.last_device_ms())
    icp.free()
    from align3d_amd import DevicePointCloud
    dt, dsrc = DevicePointCloud(ctx, tgt), DevicePointCloud(ctx, src)
    Icp.new(ctx, prm, dt).free()
    d_new, d_both = [], []
    for _ in range(9):
        t0 = time.perf_counter()
        icp_d = Icp.new(ctx, prm, dt)
        t1 = time.perf_counter()
        Td = icp_d.align(dsrc)
        d_new.append((t1 - t0) * 1e3), d_both.append((time.perf_counter() - t0) * 1e3)
        icp_d.free()
    dt.free(), dsrc.free()
    same_bits = bool(np.array_equal(np.concatenate([Td.t, Td.q]).view(np.uint32), np.concatenate([T.t, T.q]).view(np.uint32)))
    out = {"workload": f"benches/bench_icp.rs: sample1 0 <- 5 as clouds ({tgt.len()} target x {src.len()} source points), "
                       "IcpParams{max_iterations: 10}",
           "device_ms_per_align": float(np.median(dev)), "device_ms_per_align_stats": stats(dev),
           "align_wall_ms_incl_pcie": float(np.median(wall)), "align_wall_ms_stats": stats(wall),
           "icp_new_ms_incl_pcie": float(np.median(news)), "gpu_pose": [float(x) for x in T.matrix().reshape(-1)],
           # clouds already resident (a3d_pcl_icp_new_device / _align_device): the bench's body without the copies
           "icp_new_device_ms": float(np.median(d_new)), "new_plus_align_device_ms": float(np.median(d_both)),
           "new_plus_align_device_ms_stats": stats(d_both), "device_forms_give_the_same_pose_bits": same_bits}
    return out, (tgt, src)


def frame_build_kernel_us(frames_per_probe=320):
    """Kernel time per frame of the batched frame builder from the committed rocprofv3 summary of
    scripts/build_trace_probe.py (10 builds of 32 frames): sum over the builder's kernels of TotalDurationNs / frames.
    (Measured live the build is PCIe-bound: a chunk of u16 depth + u8 RGB frames takes longer to upload than to build.)"""
    import csv

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_frame_build_kernel_stats.csv")), reverse=True)
    if not files:
        return None, None, None
    total_ns, per_kernel = 0.0, {}
    for r in csv.DictReader(open(files[0])):
        name = r["Name"]
        if "rocclr" in name:  # the runtime's copy / fill kernels: PCIe staging and the scalar memset
            short = "runtime copy/fill"
        else:
            import re

            m = re.search(r"(\w+_kernel)", name)
            short = m.group(1) if m else name.split("(")[0]
        per_kernel[short] = per_kernel.get(short, 0.0) + float(r["TotalDurationNs"]) / frames_per_probe / 1e3
        total_ns += float(r["TotalDurationNs"])
    return total_ns / frames_per_probe / 1e3, per_kernel, os.path.relpath(files[0], ROOT)


def frame_build_roofline(ctx, W, H, levels=3):
    """Algorithmic bytes of ONE frame through the fused builder (DESIGN.md §4 "frame builder"): u16 depth + u8 RGB in;
    every array of every pyramid level written once (colours 3, points 12, mask 1, normals 12, intensities 1 B per
    pixel + the (h+2)(w+2) f32 map); per blur tile the splat marked: its 16^3-cell window read once (4 B per cell) and its
    12^3 blurred cells written once and read once by the slice (8 B each); first-channel zero tiles written once.
    (Until round 4 the packed grid was also cleared per build, 4 B per cell of the grid = 9.6 of then 43.9 MB per frame:
    the builder no longer does that — its last kernel puts back the zeros the splat replaced — so those bytes are no
    longer counted: the same kernel time now gives a LOWER fraction.)  Kernel time: live, see below."""
    frames, _ = synth.frame_stream(4242, 64, W, H)
    builder = RangeImageBuilder(ctx).with_bilateral_filter(BilateralFilter.default())
    cam = synth.camera(W, H)
    # LIVE kernel time: every chunk of a build (64 frames: two of 32) bracketed by hipEvents on the builder's stream behind the wait for
    # its upload (a3d_context_set_build_profiling): the builder's kernels without PCIe, measured in this very run
    ctx.set_build_profiling(True)
    live = []
    st = None
    for rep in range(6):
        pyr = builder.build_many(cam, frames, synth.DEPTH_SCALE)
        if rep:
            live.append(ctx.last_build_kernel_ms() / len(frames) * 1e3)
        st = ctx.last_build_stats()
        for lv in (lv for p in pyr for lv in p):
            lv.free()
    ctx.set_build_profiling(False)
    n = st["frames"] or 1
    px = sum((W >> l) * (H >> l) for l in range(levels))
    out_bytes = 29 * px + sum(4 * ((W >> l) + 2) * ((H >> l) + 2) for l in range(levels))
    grid_bytes = (st["marked_tiles"] * (4096 * 4 + 1728 * 16) + st["zero_tiles"] * 1728 * 8) / n
    alg = 5 * W * H + out_bytes + grid_bytes
    us = float(np.median(live))
    prof_us, per_kernel, src = frame_build_kernel_us()
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "the 9 kernels of one batched build, per frame",
         "algorithmic_bytes_per_frame": alg, "input_bytes": 5 * W * H, "pyramid_bytes": out_bytes, "grid_bytes": grid_bytes,
         "grid_cells_per_frame": st["grid_cells"] / n, "marked_tiles_per_frame": st["marked_tiles"] / n,
         "kernel_us_per_frame": us, "kernel_us_per_frame_stats": stats(live),
         "kernel_us_source": "live: hipEvent brackets around each chunk's kernels (2 chunks of 32 frames per build, 5 builds)",
         "profile_kernel_us_per_frame": prof_us, "profile_kernel_us_by_kernel": per_kernel, "profile_source": src,
         "achieved": alg / (us * 1e-6) / 1e9}
    # fabric bytes per frame of the builder's kernels together, from the committed PMC passes (scripts/profile_round.sh)
    r["traffic"], r["traffic_source"] = measured_traffic("frame_build", frames_per_build=32, width=W, height=H)
    r["traffic_unit"] = "bytes/frame"
    r["frac"] = r["achieved"] / HBM_PEAK_GBS
    return r


def named_shapes_bench(ctx, targets, sources):
    """The other two ICP shapes BASELINE.md names, on the same resident pyramids: `bench10` = the reference's
    published bench (benches/bench_image_icp.rs: one 640x480 level, IcpParams::default() with 10 iterations; README:
    38.576 ms on 16 CPU threads) and `msdefault` = MsIcpParams::default() (20/20/30 iterations)."""
    out = {}
    P = len(targets)
    for name, prm, levels in (("bench10", MsIcpParams.repeat(1, IcpParams(max_iterations=10)), 1),
                              ("msdefault", MsIcpParams.default(), 3)):
        tp = [t[:levels] for t in targets]
        sp = [s[:levels] for s in sources]
        batch = MultiscaleAlignBatch(ctx, prm, tp, sp)
        for _ in range(3):
            batch.enqueue()
        ctx.synchronize()
        per = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(10):
                batch.enqueue()
            ctx.synchronize()
            per.append((time.perf_counter() - t0) / 10 * 1e3)
        batch_ms = float(np.median(per))
        _, status = batch.align()
        batch.free()
        one = MultiscaleAlign.new(ctx, prm, tp[0])
        for _ in range(2):
            one.align(sp[0])
        lat = []
        for _ in range(9):
            t0 = time.perf_counter()
            one.align(sp[0])
            lat.append((time.perf_counter() - t0) * 1e3)
        iters = [int(p.max_iterations) for p in prm]
        alg = sum(iters[l] * level_bytes(640 >> l, 480 >> l) for l in range(levels))
        out[name] = {"pairs_per_s_batch_of_%d" % P: P / (batch_ms * 1e-3), "batch_ms_stats": stats(per),
                     "single_pair_latency_ms": float(np.median(lat)), "single_pair_latency_ms_stats": stats(lat),
                     "algorithmic_bytes_per_pair": alg,
                     "frac_of_8TBs_batched": alg * P / (batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "failed_pairs": int(np.count_nonzero(status))}
    pub_ms = 38.576  # README.md:130, i7-11800H x 16 threads
    b10 = out["bench10"]
    b10["published_cpu_ms"] = pub_ms
    b10["x_vs_published_single_pair"] = pub_ms / b10["single_pair_latency_ms"]
    b10["x_vs_published_batched"] = pub_ms * 1e-3 * b10["pairs_per_s_batch_of_%d" % P]
    return out


def streaming_bench(ctx, params, P, W, H, rounds=32, builders=1, pinned=True, builder_priority=-1):
    """End to end from host frames: every round, P + 1 NEW frames (u16 depth + u8 RGB in host memory) are uploaded
    and built into resident pyramids by `builders` threads (whole rounds in turn, each thread on its own context = HIP
    stream + scratch, ONE batched build call per round) running ahead of the alignment, which alternates two batch
    objects on the main context so that round r is enqueued behind round r-1 without waiting for it.  Reports pairs/s
    with every frame crossing PCIe and being filtered / back-projected / pyramided exactly once."""
    import threading

    frames, _ = synth.frame_stream(4242, P + 1, W, H)  # the same host frames every round: they are rebuilt each time
    # the stream's frames lie back to back in ONE buffer per array (as a capture ring would hold them): the builder
    # then uploads a chunk (up to 48 frames) with one copy per array; page-locked: a DMA at the PCIe rate
    alloc = ctx.pinned_empty if pinned else (lambda shape, dtype: np.empty(shape, dtype))
    all_d, all_c = alloc((P + 1, H, W), np.uint16), alloc((P + 1, H, W, 3), np.uint8)
    for i, (d, rgb) in enumerate(frames):
        all_d[i], all_c[i] = d, rgb
    frames = [(all_d[i], all_c[i]) for i in range(P + 1)]
    cam = synth.camera(W, H)
    # the builders' streams get the device's highest priority: the build is the dependent chain of the two workloads
    # (the first builder is the main context's sibling — the odometry loop's builder context — so that the process does not
    # hold more streams than the runtime has hardware queues: streams that share a queue serialise)
    ctxs = [ctx.sibling() if (k == 0 and builder_priority < 0) else Context(ctx.device_index, priority=builder_priority)
            for k in range(builders)]
    bld = [RangeImageBuilder(c).with_bilateral_filter(BilateralFilter.default()) for c in ctxs]

    def build_round(k=0):  # one batched build call: 10 launches per chunk of up to 48 frames
        return bld[k].build_many(cam, frames, synth.DEPTH_SCALE)

    def free_round(pyr):
        for lv in (lv for p in pyr for lv in p):
            lv.free()

    for k in range(builders):  # warm-up rounds (scratch, arena pools, code objects)
        cur = build_round(k)
        if k + 1 < builders:
            free_round(cur)
    # two batch objects alternate: round r is rebound and enqueued behind round r-1 while that one still computes, so
    # the alignment stream never waits for the host; the builders run one round ahead through a bounded queue
    batches = [MultiscaleAlignBatch(ctx, params, cur[:P], cur[1:]) for _ in range(2)]
    for b in batches:
        b.align()
    free_round(cur)
    import queue

    # builder k builds rounds k, k + builders, ... (whole rounds, so that the exposed upload of one builder's first
    # chunk lies under the other builder's kernels), each one round ahead of the alignment
    built = [queue.Queue(maxsize=1) for _ in range(builders)]

    def producer(k):
        for _ in range(k, rounds, builders):
            built[k].put(build_round(k))

    t0 = time.perf_counter()
    prods = [threading.Thread(target=producer, args=(k,)) for k in range(builders)]
    for t in prods:
        t.start()
    failed, prev = 0, None
    for r in range(rounds):
        pyr = built[r % builders].get()
        b = batches[r % 2]
        b.rebind(pyr[:P], pyr[1:])                   # (nothing allocated; waits for this batch's own round r-2 only)
        b.enqueue()
        if prev is not None:                         # read round r-1 while round r computes
            _, status = prev[0].results()
            failed += int(np.count_nonzero(status))
            free_round(prev[1])
        prev = (b, pyr)
    _, status = prev[0].results()
    failed += int(np.count_nonzero(status))
    dt = time.perf_counter() - t0
    for t in prods:
        t.join()
    free_round(prev[1])
    for b in batches:
        b.free()
    for c in ctxs:
        if c is not ctx._sibling:
            c.close()
    return {"workload": f"{rounds} rounds of {P} pairs, {P + 1} new frames per round from "
                        f"{'page-locked' if pinned else 'pageable'} host memory, "
                        f"{builders} builder threads (whole rounds in turn, one batched build call per round) running "
                        f"ahead of the alignment, two batch objects alternating (round r enqueued behind round r-1)",
            "pairs_per_s": rounds * P / dt, "frames_built_per_s": rounds * (P + 1) / dt, "failed_pairs": failed}


def odometry_bench(ctx, n_frames=20):
    """configs[3]: a 20-frame odometry stream, frames enter as u16 depth + u8 RGB, persistent device pyramids,
    MsIcpParams::default() (README usage) — on the synthetic stream, and on the reference's own 20 sample1 frames
    (tests/golden, real data with ground truth) when the fixture directory travels with the repository."""
    from align3d_amd import SlamTbDataset, SyntheticDataset, run_odometry, run_odometry_batched

    def run(ds, label):
        run_odometry(ctx, ds, max_frames=3)
        t0 = time.perf_counter()
        run_odometry(ctx, ds, prefetch=False)
        dt_seq = time.perf_counter() - t0
        per = []
        for _ in range(5):
            t0 = time.perf_counter()
            pred, metrics = run_odometry(ctx, ds)  # frame i+1 is built on a second stream while i-1 -> i is aligned
            per.append(time.perf_counter() - t0)
        dt = float(np.median(per))
        n = ds.len()
        # alignments in flight: frames i -> i + 1 start aligning while i - 1 -> i still run (independent alignments, each
        # lane its own context / stream / compute pipe; poses delivered in frame order; the same bits)
        lanes = {}
        for k in (2, 3):
            run_odometry(ctx, ds, max_frames=5, in_flight=k)
            perk = []
            for _ in range(5):
                t0 = time.perf_counter()
                predk, _ = run_odometry(ctx, ds, in_flight=k)
                perk.append(time.perf_counter() - t0)
            dtk = float(np.median(perk))
            lanes[k] = {"frames_per_s": (n - 1) / dtk, "ms_per_frame": dtk / (n - 1) * 1e3,
                        "ms_per_frame_stats": stats([t / (n - 1) * 1e3 for t in perk]),
                        "same_bits_as_one_in_flight": bool(all(np.array_equal(a.t, b.t) and np.array_equal(a.q, b.q)
                                                                for a, b in zip(predk.camera_to_world, pred.camera_to_world)))}
        # the same sequence as a recorded one: one batched build + ONE batch of n - 1 alignments (run_odometry_batched)
        run_odometry_batched(ctx, ds)
        per_b = []
        for _ in range(5):
            t0 = time.perf_counter()
            pred_b, metrics_b = run_odometry_batched(ctx, ds)
            per_b.append(time.perf_counter() - t0)
        dt_b = float(np.median(per_b))
        return {"workload": f"{n}-frame {label} stream, device RangeImageBuilder + MsIcpParams::default()",
                "frames_per_s": (n - 1) / dt, "ms_per_frame": dt / (n - 1) * 1e3,
                "ms_per_frame_stats": stats([t / (n - 1) * 1e3 for t in per]),
                "frames_per_s_without_prefetch": (n - 1) / dt_seq,
                "two_alignments_in_flight": lanes[2], "three_alignments_in_flight": lanes[3],
                "recorded_sequence_batched": {
                    "frames_per_s": (n - 1) / dt_b, "ms_per_frame": dt_b / (n - 1) * 1e3,
                    "ms_per_frame_stats": stats([t / (n - 1) * 1e3 for t in per_b]),
                    "mean_trajectory_error": {"angle_deg": float(np.degrees(metrics_b.angle)),
                                              "translation_m": metrics_b.translation}},
                "mean_trajectory_error": {"angle_deg": float(np.degrees(metrics.angle)), "translation_m": metrics.translation}}

    class InMemory:
        """The dataset with its frames decoded once: the timed loop measures the alignment path, not PNG decoding."""

        def __init__(self, ds):
            self.ds, self.frames = ds, [ds.get(i) for i in range(ds.len())]

        def len(self):
            return len(self.frames)

        def get(self, i):
            return self.frames[i]

        def trajectory(self):
            return self.ds.trajectory()

    out = run(InMemory(SyntheticDataset(7, n_frames)), "synthetic")
    real = os.path.join(ROOT, "tests", "golden", "rgbd", "sample1")
    if os.path.isdir(real):
        out["sample1_real_data"] = run(InMemory(SlamTbDataset.load(real)), "sample1 (reference test data, decoded once)")
    return out


# ---- CPU baselines (the oracle: a restatement, kind "port", never the Rust reference) ---------------------------

def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the GPU box gives a 1-GPU job a 16-core share of the host; do not oversubscribe it
    return model, max(1, min(len(os.sched_getaffinity(0)), 16))


def load_cpu_oracle():
    """The oracle built for the host it is timed on (-O3 -march=native), falling back to the portable
    test build.  Test infrastructure used as the measured CPU baseline only (never by the product)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    native = os.path.join(ROOT, "oracle", "_native", "liba3d_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=300)  # -B: -march=native must mean THIS host
    except Exception as e:  # no compiler on the box: time the portable build
        log(f"native oracle build failed ({e}); timing the portable -O2 build")
    O.load(native if os.path.exists(native) else None)
    return O, ("-O3 -march=native" if os.path.exists(native) else "-O2 (portable)")


def cpu_baseline_main(ctx, O, host_pyramids, pair_frames, params, n_pairs, gpu_poses, cores, orders=13, budget_s=25.0,
                      gpu_poses_pinned=None, teacher=True):
    """ms3x15 on the batch's pairs, threaded like the reference (4096-pixel chunks over the host's cores).
    TIMED: the chunk-order run of every pair (until `budget_s` of CPU work is spent) -> `value`.
    UNTIMED (parity of the headline): every pair also under `orders` - 1 seeded chunk-merge orders
    (what rayon's par_bridge() does to the reference, image_icp.rs:96,143-148) -> per-pair GPU-vs-oracle difference,
    the oracle's own spread and the GPU's rank inside it; then the two most sensitive pairs teacher-forced at all 45
    iterations (tests/headline_parity.py)."""
    import headline_parity as HP

    parr = params.to_c_array()
    per, entries, entries_pinned = [], [], []
    keep = {}
    for p in range(n_pairs):
        fa, fb = pair_frames[p]
        ta, tb = [HP.host_frame(r) for r in host_pyramids[fa]], [HP.host_frame(r) for r in host_pyramids[fb]]
        t0 = time.perf_counter()
        st, T = O.multiscale_align(parr, len(params), ta, tb, threads=cores)
        dt = time.perf_counter() - t0
        if sum(per) <= budget_s:
            per.append(dt)
        if st != 0 or gpu_poses is None:
            continue
        runs = [T]
        try:
            for seed in range(1, orders):
                O.set_chunk_merge_order(seed)
                st2, T2 = O.multiscale_align(parr, len(params), ta, tb, threads=cores)
                if st2 == 0:
                    runs.append(T2)
        finally:
            O.set_chunk_merge_order(0)
        table = HP.oracle_distance_table(runs)
        e = HP.envelope_from_distances(gpu_poses[p], runs, table)
        e["pair"] = p
        entries.append(e)
        if gpu_poses_pinned is not None:  # the same pair under a3d_context_set_tiling(24), against the same oracle runs
            e2 = HP.envelope_from_distances(gpu_poses_pinned[p], runs, table)
            e2["pair"] = p
            entries_pinned.append(e2)
        keep[p] = (ta, tb)
        if len(keep) > 4:  # host copies of the most sensitive pairs only
            best = sorted(keep, key=lambda q: -next(x for x in entries if x["pair"] == q)["gpu_vs_cpu_translation_m"])[:2]
            keep = {q: keep[q] for q in best}
    fa, fb = pair_frames[0]
    t0 = time.perf_counter()
    O.multiscale_align(parr, len(params), [HP.host_frame(r) for r in host_pyramids[fa]],
                       [HP.host_frame(r) for r in host_pyramids[fb]], threads=1)
    single_ms = (time.perf_counter() - t0) * 1e3
    out = {
        "value": len(per) / sum(per), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
        "sample": f"{len(per)} of the batch's {n_pairs} frame pairs, same ms3x15 workload, oracle threaded over {cores} host "
                  f"threads (4096-pixel chunks like the reference's rayon loop); parity (untimed) on {len(entries)} pairs x "
                  f"{orders} chunk-merge orders",
        "ms_per_pair_stats": stats([t * 1e3 for t in per]), "single_thread_ms_per_pair": single_ms,
    }
    if entries:
        out.update(HP.summarize(entries))
        # ms3x15 = IcpParams::default() per level, which is not contractive on every pair: where the GPU
        # differs from the oracle by more than 1e-4 the oracle differs from ITSELF as much under another chunk-merge order
        out["per_pair_parity"] = entries
        if entries_pinned:
            sp = HP.summarize(entries_pinned)
            out["pinned_tiling_parity"] = dict(sp, per_pair_parity=entries_pinned)
            out["pinned_tiling_pairs_over_1e-4"] = sp["pairs_over_1e-4"]
            out["pinned_tiling_pairs_over_1e-4_and_outside_the_cpu_envelope"] = sp["pairs_over_1e-4_and_outside_the_cpu_envelope"]
            out["pinned_tiling_max_gpu_vs_cpu_translation_m"] = sp["max_gpu_vs_cpu_translation_m"]
        worst = sorted(keep, key=lambda q: -next(x for x in entries if x["pair"] == q)["gpu_vs_cpu_translation_m"])[:2]
        tfs = []
        for p in (worst if teacher else []):
            fa, fb = pair_frames[p]
            ta, tb = keep[p]
            tf = HP.teacher_forced(ctx, params, ta, tb, host_pyramids[fa], host_pyramids[fb], threads=cores)
            tf["pair"] = p
            tfs.append(tf)
        if teacher:
            out["teacher_forced_most_sensitive_pairs"] = tfs
            out["teacher_forced_count_mismatches"] = sum(t["count_mismatches"] for t in tfs)
            out["teacher_forced_max_rel_err_sums"] = max((t["max_rel_err_sums"] for t in tfs), default=0.0)
            out["teacher_forced_max_one_step_translation_m"] = max((t["max_one_step_translation_m"] for t in tfs), default=0.0)
    return out


def cpu_baselines_secondary(O, cores, level0_host, depth_u16, bench10_pair, clouds, bench_icp_clouds=None, gpu_pcl_pose=None):
    """The reference's other benches (benches/bench_{kdtree,icp,compute_normals,bilateral,image_icp}.rs,
    README.md:130-134) on the oracle, threaded as the reference threads them, each bounded to a few seconds."""
    out = {}

    def timed(fn, reps):
        fn()
        per = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            per.append((time.perf_counter() - t0) * 1e3)
        return per

    # kd-tree: 500k queries on a 500k-point tree, single thread (bench_kdtree.rs:11-43; README: 101.75 ms)
    n = 500_000
    db = synth.uniform01_f32(10, 3 * n).reshape(n, 3)
    q = synth.uniform01_f32(11, 3 * n).reshape(n, 3)
    t0 = time.perf_counter()
    tree = O.KdTree(db)
    kd_build_ms = (time.perf_counter() - t0) * 1e3
    per = timed(lambda: tree.nearest(q), 3)
    out["kdtree_500k"] = {"value": n / (np.median(per) * 1e-3), "unit": "queries/s", "cores": 1, "kind": "port",
                          "ms_stats": stats(per), "build_ms": kd_build_ms, "published_reference_ms": 101.75,
                          "sample": "500k queries, 500k-point tree, single thread like R3dTree::nearest in a loop"}
    del tree
    # Icp 500k x 500k, single thread like the reference (bench_icp.rs:9-39): 2 iterations timed, scaled to 15
    tgt, src = clouds
    t0 = time.perf_counter()
    ptree = O.KdTree(tgt.points)
    icp_new_ms = (time.perf_counter() - t0) * 1e3
    prm = O.params(max_iterations=2)
    tv, sv = O.pcl_view(tgt.points, tgt.normals), O.pcl_view(src.points, src.normals)
    pose = O.pose()

    def icp2():
        assert O.load().orc_pcl_icp_align(C.byref(prm), ptree.h, C.byref(tv), C.byref(sv), C.byref(pose), None) == 0

    per = timed(icp2, 2)
    it_ms = float(np.median(per)) / 2
    out["pcl_icp_500k"] = {"value": 1e3 / (15 * it_ms), "unit": "aligns/s (15 iterations)", "cores": 1, "kind": "port",
                           "ms_per_iteration": it_ms, "icp_new_ms": icp_new_ms,
                           "sample": "2 of the 15 iterations timed, single thread like Icp::align; x 7.5 for an align"}
    if gpu_pcl_pose is not None:
        # configs[2] at its stated size against the oracle — the full 15-iteration Icp::align on the CPU
        # (~0.5 s, untimed) and the GPU's pose beside it (north-star tolerance: 1e-4 rad / 1e-4 m)
        prm15, full = O.params(max_iterations=15), O.pose()
        assert O.load().orc_pcl_icp_align(C.byref(prm15), ptree.h, C.byref(tv), C.byref(sv), C.byref(full), None) == 0
        ang, tr = O.transform_metrics(O.pose(gpu_pcl_pose[:3], gpu_pcl_pose[3:]), full)
        out["pcl_icp_500k"]["gpu_vs_cpu_angle_rad"], out["pcl_icp_500k"]["gpu_vs_cpu_translation_m"] = abs(float(ang)), float(tr)
    del ptree
    # benches/bench_icp.rs on the oracle: sample1 0 <- 5 clouds, 10 iterations, single thread like the reference
    if bench_icp_clouds is not None:
        btgt, bsrc = bench_icp_clouds
        t0 = time.perf_counter()
        btree = O.KdTree(btgt.points)
        bnew_ms = (time.perf_counter() - t0) * 1e3
        bprm = O.params(max_iterations=10)
        btv, bsv = O.pcl_view(btgt.points, btgt.normals), O.pcl_view(bsrc.points, bsrc.normals)
        bpose = O.pose()

        def bench_icp_align():
            assert O.load().orc_pcl_icp_align(C.byref(bprm), btree.h, C.byref(btv), C.byref(bsv), C.byref(bpose), None) == 0

        per = timed(bench_icp_align, 3)
        out["bench_icp"] = {"value": float(np.median(per)), "unit": "ms per Icp::align (10 iterations)", "cores": 1,
                            "kind": "port", "ms_stats": stats(per), "icp_new_ms": bnew_ms,
                            "cpu_pose_t_q": [float(x) for x in list(bpose.t[:]) + list(bpose.q[:])],
                            "sample": "3 repetitions, single thread like Icp::align (benches/bench_icp.rs:9-39)"}
        del btree
    # compute_normals on one 640x480 frame (bench_compute_normals.rs; README: 1.1778 ms, rayon over 1024-px chunks)
    pts, msk = level0_host.points, level0_host.mask
    per_mt = timed(lambda: O.compute_normals(pts, msk, threads=cores), 20)
    per_st = timed(lambda: O.compute_normals(pts, msk), 10)
    out["compute_normals_640x480"] = {"value": float(np.median(per_mt)), "unit": "ms", "cores": cores, "kind": "port",
                                      "ms_stats": stats(per_mt), "single_thread_ms": float(np.median(per_st)),
                                      "published_reference_ms": 1.1778, "sample": "20 repetitions of one frame"}
    # bilateral filter on one 640x480 depth image, single thread like the reference (bench_bilateral.rs)
    per = timed(lambda: O.bilateral(depth_u16), 3)
    out["bilateral_640x480"] = {"value": float(np.median(per)), "unit": "ms", "cores": 1, "kind": "port",
                                "ms_stats": stats(per), "sample": "3 repetitions of one frame, single thread"}
    # bench10: one 640x480 level, IcpParams::default() with 10 iterations (bench_image_icp.rs; README: 38.576 ms)
    ft, fs = bench10_pair
    p10 = O.params(max_iterations=10)
    per = timed(lambda: O.image_icp_align(p10, ft, fs, threads=cores), 5)
    per1 = timed(lambda: O.image_icp_align(p10, ft, fs, threads=1), 1)
    out["bench10"] = {"value": float(np.median(per)), "unit": "ms per alignment", "cores": cores, "kind": "port",
                      "ms_stats": stats(per), "single_thread_ms": float(np.median(per1)),
                      "published_reference_ms": 38.576, "sample": "5 repetitions of one pair of the batch"}
    return out


LINE_BUDGET_BYTES = 8000  # the driver keeps the last 8 KiB of stdout next to its parsed copy; the line must fit into it

# extra.<short name> <- dotted path in the full result (bench_detail.json).  Scalars only; a missing path is skipped.
_EXTRA_SCALARS = [
    ("ms_per_step_repeated_min", "extra.timing.ms_per_step_repeated.min"),
    ("ms_per_step_repeated_median", "extra.timing.ms_per_step_repeated.median"),
    ("ms_per_step_repeated_max", "extra.timing.ms_per_step_repeated.max"),
    ("ms_per_step_repeats_x_steps", "extra.timing.repeats_x_steps"),
    ("mean_error_vs_synthetic_gt_m", "extra.mean_error_vs_synthetic_gt.translation_m"),
    ("lone_pair_ms3x15_ms", "extra.single_pair_ms3x15_latency_ms"),
    ("lone_pair_frac", "extra.single_pair_ms3x15_frac"),
    ("pinned_tiling_pairs_per_s", "extra.pinned_tiling.pairs_per_s"),
    ("pinned_tiling_lone_pair_ms", "extra.pinned_tiling.lone_pair_latency_ms"),
    ("pinned_tiling_bit_identical_alone_and_batched", "extra.pinned_tiling.pair_alone_equals_pair_in_batch_bit_for_bit"),
    ("drop_in_ms3x15_ms_from_host_range_images", "extra.drop_in_ms3x15_ms_from_host_range_images"),
    ("bench10_pairs_per_s", "extra.named_shapes.bench10.pairs_per_s_batch_of_64"),
    ("bench10_frac", "extra.named_shapes.bench10.frac_of_8TBs_batched"),
    ("bench10_lone_pair_ms", "extra.named_shapes.bench10.single_pair_latency_ms"),
    ("msdefault_pairs_per_s", "extra.named_shapes.msdefault.pairs_per_s_batch_of_64"),
    ("msdefault_frac", "extra.named_shapes.msdefault.frac_of_8TBs_batched"),
    ("bench_icp_align_ms", "extra.named_shapes.bench_icp.device_ms_per_align"),
    ("bench_icp_new_device_ms", "extra.named_shapes.bench_icp.icp_new_device_ms"),
    ("bench_icp_new_plus_align_device_ms", "extra.named_shapes.bench_icp.new_plus_align_device_ms"),
    ("kdtree_queries_per_s", "extra.kdtree.value"),
    ("kdtree_ms_per_500k_queries", "extra.kdtree.ms_per_500k_queries"),
    ("kdtree_nearest_frac", "extra.kdtree.roofline.frac"),
    ("kdtree_nearest_traffic", "extra.kdtree.roofline.traffic"),
    ("kdtree_build_ms", "extra.kdtree.build.device_ms"),
    ("kdtree_build_kernel_ms", "extra.kdtree.build.kernel_ms"),
    ("kdtree_build_frac", "extra.kdtree.build.roofline.frac"),
    ("kdtree_build_ms_incl_pcie", "extra.kdtree.build_ms_incl_pcie"),
    ("pcl_icp_us_per_iteration", "extra.pcl_icp.us_per_iteration"),
    ("pcl_icp_frac", "extra.pcl_icp.roofline.frac"),
    ("pcl_icp_new_device_ms", "extra.pcl_icp.icp_new_device_ms"),
    ("pcl_icp_new_plus_align_device_ms", "extra.pcl_icp.new_plus_align_device_ms"),
    ("pcl_icp_gpu_vs_cpu_translation_m", "extra.cpu_baselines.pcl_icp_500k.gpu_vs_cpu_translation_m"),
    ("pcl_icp_gpu_vs_cpu_angle_rad", "extra.cpu_baselines.pcl_icp_500k.gpu_vs_cpu_angle_rad"),
    ("bilateral_device_us_per_image", "extra.bilateral_device.kernel_us_per_image"),
    ("bilateral_device_frac", "extra.bilateral_device.roofline.frac"),
    ("odometry_frames_per_s", "extra.odometry.frames_per_s"),
    ("odometry_frames_per_s_two_in_flight", "extra.odometry.two_alignments_in_flight.frames_per_s"),
    ("odometry_frames_per_s_recorded_batched", "extra.odometry.recorded_sequence_batched.frames_per_s"),
    ("odometry_sample1_translation_error_m", "extra.odometry.sample1_real_data.mean_trajectory_error.translation_m"),
    ("odometry_sample1_angle_error_deg", "extra.odometry.sample1_real_data.mean_trajectory_error.angle_deg"),
    ("compute_normals_one_frame_us", "extra.frame_prep.compute_normals_us"),
    ("compute_normals_one_frame_frac", "extra.frame_prep.compute_normals_roofline.frac"),
    ("compute_normals_batch64_frac", "extra.frame_prep.compute_normals_batch_roofline.frac"),
    ("compute_normals_batch64_traffic", "extra.frame_prep.compute_normals_batch_roofline.traffic"),
    ("frame_build_kernel_us_per_frame", "extra.frame_build.roofline.kernel_us_per_frame"),
    ("frame_build_frac", "extra.frame_build.roofline.frac"),
    ("frame_build_traffic_per_frame", "extra.frame_build.roofline.traffic"),
    ("frame_build_ms_per_frame_incl_pcie", "extra.frame_build_page_locked.bilateral_on.ms_per_frame"),
    ("streaming_pairs_per_s", "extra.streaming_from_host_frames.pairs_per_s"),
    ("cpu_kdtree_queries_per_s", "extra.cpu_baselines.kdtree_500k.value"),
    ("cpu_kdtree_build_ms", "extra.cpu_baselines.kdtree_500k.build_ms"),
    ("cpu_pcl_icp_ms_per_iteration", "extra.cpu_baselines.pcl_icp_500k.ms_per_iteration"),
    ("cpu_bench_icp_ms", "extra.cpu_baselines.bench_icp.value"),
    ("cpu_compute_normals_ms", "extra.cpu_baselines.compute_normals_640x480.value"),
    ("cpu_bilateral_ms", "extra.cpu_baselines.bilateral_640x480.value"),
    ("cpu_bench10_ms", "extra.cpu_baselines.bench10.value"),
]
_ROOFLINE_KEYS = ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "traffic_source",
                  "algorithmic_bytes_per_launch", "avg_launch_us", "launches_per_step", "concurrent_launches",
                  "launch_sequence_ms", "kernel_share_of_step", "delivered_frac", "hbm_copy_ceiling_GBs",
                  "frac_of_copy_ceiling", "level0_frac", "level1_frac", "level2_frac", "level0_avg_launch_us",
                  "level1_avg_launch_us", "level2_avg_launch_us", "failed_pairs"]
_CPU_KEYS = ["value", "unit", "cores", "kind", "sample", "cpu_model", "compiler_flags", "single_thread_ms_per_pair",
             "pairs_compared", "merge_orders_per_pair", "pairs_over_1e-4", "pairs_over_1e-4_and_outside_the_cpu_envelope",
             "pairs_over_1e-4_and_beyond_the_median_cpu_order", "pairs_over_1e-4_and_farther_than_every_cpu_order",
             "ranks_of_pairs_over_1e-4", "pinned_tiling_pairs_over_1e-4", "max_gpu_vs_cpu_angle_rad", "max_gpu_vs_cpu_translation_m",
             "median_gpu_vs_cpu_translation_m", "max_cpu_spread_translation_m", "pairs_whose_cpu_spread_exceeds_1e-4",
             "teacher_forced_count_mismatches", "teacher_forced_max_rel_err_sums", "teacher_forced_max_one_step_translation_m",
             "pinned_tiling_pairs_over_1e-4_and_outside_the_cpu_envelope", "pinned_tiling_max_gpu_vs_cpu_translation_m",
             "msdefault_pairs_compared", "msdefault_pairs_over_1e-4", "msdefault_max_gpu_vs_cpu_translation_m",
             "msdefault_max_gpu_vs_cpu_angle_rad"]


def _dig(d, path):
    for k in path.split("."):
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _short(v):
    """Six significant digits for floats (the full precision is in the detail file); everything else unchanged."""
    if isinstance(v, float):
        return float(f"{v:.6g}")
    if isinstance(v, dict):
        return {k: _short(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_short(x) for x in v]
    return v


def compact_line(full, detail_file=None):
    """The ONE stdout line the driver parses: the contract keys, `roofline` and `cpu_baseline` as scalars, and a flat
    `extra` with one number per secondary workload.  Every list and nested table of the full result (per-pair parity,
    teacher-forced rows, per-level tables, repeated-timing statistics) stays in `detail_file`.  Round 4's line carried
    them (92 KB) and the driver could not parse it; tests/test_bench_line_cpu.py holds this below LINE_BUDGET_BYTES."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "pairs_per_gpu", "global_pairs", "levels", "iterations_per_level",
                                               "distinct_frames", "ranks_in_collective", "gathered_pairs",
                                               "gather_matches_local_poses", "collective") if cfg.get(k) is not None}
    roof = full.get("roofline") or {}
    line["roofline"] = {k: roof[k] for k in _ROOFLINE_KEYS if k in roof}
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = None if cpu is None else {k: cpu[k] for k in _CPU_KEYS if k in cpu}
    extra = {}
    for short, path in _EXTRA_SCALARS:
        v = _dig(full, path)
        if v is not None and not isinstance(v, (dict, list)):
            extra[short] = v
    errs = _dig(full, "extra.errors")
    if errs:  # which secondary legs failed (their messages are in the detail file)
        extra["failed_legs"] = ",".join(sorted(errs))
    line["extra"] = extra
    line["detail_file"] = detail_file
    text = json.dumps(_short(line), separators=(",", ":"))
    if len(text) > LINE_BUDGET_BYTES:  # never print a line the driver cannot parse: drop the secondary numbers first
        line["extra"] = {"dropped": "line over budget; see detail_file"}
        text = json.dumps(_short(line), separators=(",", ":"))
    return text


def hbm_copy_ceiling(device=0):
    """What the memory system of THIS chip delivers to a compute-free kernel: scripts/copy_ceiling (built by
    __graft_entry__.build()) as a child process on HIP device `device` -> {"read_GBs", "copy_GBs"} or None when the probe is
    not built."""
    exe = os.path.join(ROOT, "scripts", "copy_ceiling")
    if not os.path.exists(exe):
        return None
    try:
        env = dict(os.environ)
        visible = [v for v in env.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
        env["HIP_VISIBLE_DEVICES"] = visible[device] if device < len(visible) else str(device)  # (the child sees one device: its 0)
        return json.loads(subprocess.run([exe], capture_output=True, timeout=120, check=True, env=env).stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        log(f"copy ceiling probe failed: {e}")
        return None


def launch_ranks(n, argv):
    """`--gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one
    per GPU) before this process has made any HIP call, let rank 0's JSON line through on the shared stdout and exit
    non-zero if any rank fails.  Children are started, never exec'ed into (a process that has touched the GPU must
    not be replaced)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # A rank that hangs (a dead GPU, a rendezvous that never completes) must not hang the bench: A3D_RANK_TIMEOUT_S
    # (default 900 s) after the start every rank still running is stopped and named.
    deadline = time.time() + float(os.environ.get("A3D_RANK_TIMEOUT_S", "900"))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                log(f"rank {r} (HIP device {r}, pid {p.pid}) exited with {code}: stopping ranks {sorted(pending)}")
                for q in pending.values():  # exactly the processes started above
                    q.terminate()
        if pending and time.time() > deadline:
            log(f"ranks {sorted(pending)} (HIP devices {sorted(pending)}) still running after A3D_RANK_TIMEOUT_S: stopping them")
            for q in pending.values():
                q.terminate()
            for q in pending.values():
                try:
                    q.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    q.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


def run_leg(errors, name, fn, *a, **kw):
    """One secondary measurement.  A failure costs that leg only: it is logged, recorded under extra.errors and the line is
    still printed (an exception in any leg after the timed region used to lose the whole line)."""
    try:
        return fn(*a, **kw)
    except Exception as e:  # noqa: BLE001 — whatever a probe raises must not reach the contract line
        log(f"[bench] leg '{name}' failed: {e!r}")
        errors[name] = repr(e)[:300]
        return None


def repeated_step_timings(ctx, batch, rep_steps, repeats=20):
    """the same step again, `repeats` x `rep_steps` steps: per-step ms of each repetition."""
    reps = []
    for _ in range(repeats):
        ctx.synchronize()
        t1 = time.perf_counter()
        for _ in range(rep_steps):
            batch.enqueue()
        ctx.synchronize()
        reps.append((time.perf_counter() - t1) / rep_steps * 1e3)
    return reps


def headline_roofline(ctx, batch, params, P, W, H, distinct, ms_per_step):
    """Roofline of the dominant kernel (image_icp_head_kernel), HIP events on the launch streams.
    The batch runs `conc` pair groups on separate streams, so `conc` launches are in flight at once.  per_launch: what one
    launch does (its own bytes / its own duration: what rocprofv3 --kernel-trace shows per kernel).  achieved: the
    chip-level figure = the step's algorithmic bytes / the device time of the whole launch sequence (first launch start ->
    last launch end, events on the context stream), i.e. the concurrent launches summed without double counting."""
    conc = batch.concurrency()
    kernel_ms, region_ms, launches = [], [], 0
    level_ms = [[] for _ in range(len(params))]
    level_launches = [0] * len(params)
    for _ in range(20):  # launch sequence as it runs in the timed steps (no per-launch events)
        batch.enqueue()
        ctx.synchronize()
        region_ms.append(batch.last_timing()[0])
    batch.set_profiling(True)
    for _ in range(20):  # per-launch durations, events around every launch on its stream
        batch.enqueue()
        ctx.synchronize()
        kernel_ms.append(batch.last_kernel_ms())
        launches = batch.last_timing()[1]
        for l in range(len(params)):
            ms_l, level_launches[l] = batch.last_level_ms(l)
            level_ms[l].append(ms_l)
    batch.set_profiling(False)
    kms, rms = float(np.median(kernel_ms)), float(np.median(region_ms))
    iters = [int(p.max_iterations) for p in params]
    step_alg_bytes = P * sum(iters[l] * level_bytes(W >> l, H >> l) for l in range(len(params)))
    bytes_per_launch = step_alg_bytes / max(1, launches)
    avg_launch_ms = kms / max(1, launches)
    achieved = step_alg_bytes / (rms * 1e-3) / 1e9
    traffic, traffic_src = (measured_traffic("bench", pairs_per_gpu=P, concurrent_launches=conc, distinct_frames=int(distinct))
                            if (W, H) == (640, 480) else (None, None))
    batch_kernel_name = "image_icp_kernel" if os.environ.get("A3D_ICP_HANDOFF") == "ticket" else "image_icp_head_kernel"
    roof = {
        "bound": "hbm", "kernel": batch_kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes/launch",
        "traffic_source": traffic_src,
        "launches_per_step": int(launches), "concurrent_launches": conc,
        "avg_launch_us": avg_launch_ms * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
        "per_launch_GBs": bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9,
        "launch_sequence_ms": rms, "launch_sequence_ms_stats": stats(region_ms),
        "kernel_share_of_step": rms / ms_per_step,
        # The frames come from the device builder, whose images carry "mask == (z > 0)": the kernel derives both masks
        # from z and never reads the two mask bytes per pixel that SURVEY §8(d)'s 39 B per pixel credit it with, so
        # the rate it actually delivers from HBM is 37/39 of `achieved` (the PMC `traffic` above shows the same).
        "delivered_GBs": achieved * (37 * W * H + 4 * (W + 2) * (H + 2)) / level_bytes(W, H),
        "delivered_frac": achieved * (37 * W * H + 4 * (W + 2) * (H + 2)) / level_bytes(W, H) / HBM_PEAK_GBS,
        "delivered_note": "masks derived from z (builder-made frames): 37 instead of 39 B per pixel are read",
    }
    # per pyramid level: `conc` launches of a level run at once (the pair groups move in step), so the level's share
    # of the sequence is its summed launch time / conc, and its bandwidth the level's bytes over that
    per_level = []
    for l in range(len(params)):
        ms_l = float(np.median(level_ms[l])) / max(1, conc)
        b_l = P * iters[l] * level_bytes(W >> l, H >> l)
        per_level.append({"level": l, "size": f"{W >> l}x{H >> l}", "launches": int(level_launches[l]),
                          "avg_launch_us": float(np.median(level_ms[l])) / max(1, level_launches[l]) * 1e3,
                          "ms_of_sequence": ms_l, "algorithmic_bytes": b_l,
                          "achieved_GBs": (b_l / (ms_l * 1e-3) / 1e9) if ms_l > 0 else None,
                          "frac": (b_l / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_l > 0 else None})
    roof["per_level"] = per_level
    for lv in per_level:  # flat copies: the driver's record keeps scalars only
        roof[f"level{lv['level']}_frac"] = lv["frac"]
        roof[f"level{lv['level']}_avg_launch_us"] = lv["avg_launch_us"]
        roof[f"level{lv['level']}_share_of_sequence"] = lv["ms_of_sequence"] / rms if rms > 0 else None
    return roof, step_alg_bytes, iters


def add_copy_ceiling(roof, device):
    """The chip's own ceiling beside the nominal 8 TB/s: scripts/copy_ceiling as a child process on this rank's device."""
    ceil = hbm_copy_ceiling(device)
    if ceil and "read_GBs" in ceil:
        roof["hbm_copy_ceiling"] = ceil
        roof["hbm_copy_ceiling_GBs"] = max(ceil["read_GBs"], ceil["copy_GBs"])
        roof["frac_of_copy_ceiling"] = roof["achieved"] / roof["hbm_copy_ceiling_GBs"]


def lone_pair_leg(ctx, params, target, source, step_alg_bytes, P):
    """configs[1]: one pair alone on the GPU (latency-bound: 45 dependent iterations)."""
    ms1 = MultiscaleAlign.new(ctx, params, target)
    for _ in range(2):
        ms1.align(source)
    lat = []
    for _ in range(15):
        t1 = time.perf_counter()
        ms1.align(source)
        lat.append((time.perf_counter() - t1) * 1e3)
    med = float(np.median(lat))
    return {"single_pair_ms3x15_latency_ms": med,
            # configs[1] as written (one pair alone): 45 dependent launches; its algorithmic bytes over its latency
            "single_pair_ms3x15_frac": (step_alg_bytes / P) / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "single_pair_ms3x15_latency_ms_stats": stats(lat)}


def secondary_legs(ctx, errors, extra, params, targets, sources, P, W, H, build_ms, ms_per_step, step_alg_bytes):
    """Everything beside the headline (N = 1 only), each leg on its own (run_leg).  Returns what the CPU legs need."""
    keep = {"level0_host": None, "depth0": None, "clouds": None, "bench_icp_clouds": None, "poses_pinned": None}
    lone = run_leg(errors, "lone_pair", lone_pair_leg, ctx, params, targets[0], sources[0], step_alg_bytes, P)
    if lone:
        extra.update(lone)
        r = run_leg(errors, "pinned_tiling", pinned_tiling_bench, ctx, params, targets, sources,
                    lone["single_pair_ms3x15_latency_ms"], ms_per_step)
        if r:
            extra["pinned_tiling"], keep["poses_pinned"] = r
    r = run_leg(errors, "drop_in", drop_in_bench, ctx, params, targets[0], sources[0])
    if r:
        extra["drop_in_from_host_range_images"] = r
        extra["drop_in_ms3x15_ms_from_host_range_images"] = r["page_locked"]["ms3x15_ms"]
    extra["named_shapes"] = run_leg(errors, "named_shapes", named_shapes_bench, ctx, targets, sources) or {}
    r = run_leg(errors, "bench_icp", bench_icp_shape, ctx)
    if r and r[0] is not None:
        extra["named_shapes"]["bench_icp"], keep["bench_icp_clouds"] = r
    r = run_leg(errors, "kdtree", kdtree_bench, ctx)
    if r:
        r["x_vs_published_cpu_101.75ms"] = 101.75 / r["ms_per_500k_queries"]
        extra["kdtree"] = r
    r = run_leg(errors, "pcl_icp", pcl_icp_bench, ctx)
    if r:
        extra["pcl_icp"], keep["clouds"] = r
    r = run_leg(errors, "odometry", odometry_bench, ctx)
    if r:
        extra["odometry"] = r
    run_leg(errors, "release_lanes", ctx.release_lanes)  # (the lanes' contexts of the in-flight runs: no idle streams beside what follows)

    def frame_prep():
        keep["level0_host"] = targets[0][0].download()
        keep["depth0"] = synth.frame_stream(1000, 1, W, H)[0][0][0]
        return frame_prep_bench(ctx, keep["level0_host"], keep["depth0"])

    r = run_leg(errors, "frame_prep", frame_prep)
    if r:
        extra["frame_prep"] = r
    r = run_leg(errors, "bilateral_device", bilateral_device_bench, ctx, W, H)
    if r:
        extra["bilateral_device"] = r
    # what a caller with host buffers pays per new frame: u16 depth + u8 RGB over PCIe, then bilateral,
    # back-projection, normals, pyramid, luma and intensity maps on the device (batched build of 65 frames)
    extra["frame_build_ms_incl_pcie"] = build_ms  # pageable host frames, cold arena pool (the run's first batch)
    r = run_leg(errors, "frame_build_page_locked", frame_build_bench, ctx, P + 1, W, H)
    if r:
        extra["frame_build_page_locked"] = r
    r = run_leg(errors, "frame_build_roofline", frame_build_roofline, ctx, W, H)
    if r:
        extra["frame_build"] = {"roofline": r}
    extra["pairs_per_s_including_one_frame_build_per_pair"] = 1e3 / (build_ms + ms_per_step / P)

    def streaming():
        # the pipelined loop is sensitive to how its two host threads and the two launch chains interleave (10-15 k
        # pairs/s between otherwise identical runs): five repetitions, the median reported, all of them kept
        runs = [streaming_bench(ctx, params, P, W, H) for _ in range(5)]
        runs.sort(key=lambda x: x["pairs_per_s"])
        return dict(runs[len(runs) // 2], pairs_per_s_stats=stats([x["pairs_per_s"] for x in runs]),
                    failed_pairs=int(sum(x["failed_pairs"] for x in runs)))

    r = run_leg(errors, "streaming", streaming)
    if r:
        extra["streaming_from_host_frames"] = r
    return keep


def msdefault_parity_leg(ctx, O, host_pyramids, pair_frames, n_pairs, cores):
    """the headline's own pairs under the reference's contractive parameter set MsIcpParams::default()
    (icp_params.rs:112-133), one oracle run per pair (chunk order): how many differ from the oracle by more than the
    north-star tolerance 1e-4 rad / 1e-4 m — literally, no envelope."""
    import headline_parity as HP

    prm = MsIcpParams.default()
    pairs = pair_frames[:n_pairs]
    batch = MultiscaleAlignBatch(ctx, prm, [host_pyramids[a] for a, _ in pairs], [host_pyramids[b] for _, b in pairs])
    poses, status = batch.align()
    batch.free()
    worst_a = worst_t = 0.0
    over = 0
    t0 = time.perf_counter()
    for p, (fa, fb) in enumerate(pairs):
        ta, tb = [HP.host_frame(r) for r in host_pyramids[fa]], [HP.host_frame(r) for r in host_pyramids[fb]]
        run = HP.oracle_runs(prm, ta, tb, cores, 1)[0]
        ang, tr = O.transform_metrics(poses[p].to_c(), run)
        ang = abs(ang)
        worst_a, worst_t = max(worst_a, ang), max(worst_t, tr)
        over += int(ang > 1e-4 or tr > 1e-4)
    return {"msdefault_pairs_compared": len(pairs), "msdefault_pairs_over_1e-4": over,
            "msdefault_max_gpu_vs_cpu_angle_rad": float(worst_a), "msdefault_max_gpu_vs_cpu_translation_m": float(worst_t),
            "msdefault_failed_pairs": int(np.count_nonzero(status)), "msdefault_oracle_seconds": time.perf_counter() - t0}


def cpu_legs(ctx, errors, extra, args, world, host_pyramids, pair_frames, params, P, poses, targets, sources, keep):
    """cpu_baseline (rank 0).  N = 1: all pairs, 13 merge orders, the teacher-forced check, msdefault parity and the
    secondary baselines.  N > 1: a BOUNDED sample so that the scaling line carries a cpu_baseline
    too — chunk-order timing of at most 8 pairs (<= 5 s) and the envelope of those pairs under 5 merge orders."""
    r = run_leg(errors, "load_cpu_oracle", load_cpu_oracle)
    if not r:
        return None
    O, flags = r
    model, cores = cpu_info()
    if world == 1:
        cpu = run_leg(errors, "cpu_baseline", cpu_baseline_main, ctx, O, host_pyramids, pair_frames, params,
                      min(args.cpu_pairs, P), poses, cores, orders=max(1, args.cpu_orders), gpu_poses_pinned=keep["poses_pinned"])
    else:
        cpu = run_leg(errors, "cpu_baseline", cpu_baseline_main, ctx, O, host_pyramids, pair_frames, params,
                      min(args.cpu_pairs, P, 8), poses, cores, orders=min(5, max(1, args.cpu_orders)), budget_s=5.0,
                      teacher=False)
    if not cpu:
        return None
    cpu["cpu_model"], cpu["compiler_flags"] = model, flags
    if world > 1:
        return cpu
    r = run_leg(errors, "msdefault_parity", msdefault_parity_leg, ctx, O, host_pyramids, pair_frames, min(args.cpu_pairs, P), cores)
    if r:
        cpu.update(r)
    if keep["level0_host"] is not None and keep["clouds"] is not None:
        def oframe(dev_level):
            ri = dev_level.download(colors=False)
            k = ri.intrinsics
            return O.Frame(ri.points, ri.mask, k.fx, k.fy, k.cx, k.cy, ri.normals, ri.intensities, ri.intensity_map)

        sec = run_leg(errors, "cpu_baselines_secondary", cpu_baselines_secondary, O, cores, keep["level0_host"], keep["depth0"],
                      (oframe(targets[0][0]), oframe(sources[0][0])), keep["clouds"], keep["bench_icp_clouds"],
                      gpu_pcl_pose=(extra.get("pcl_icp") or {}).get("gpu_pose_t_q"))
        if sec:
            for v in sec.values():
                v["cpu_model"], v["compiler_flags"] = model, flags
            extra["cpu_baselines"] = sec

            def ratios():
                # GPU / CPU beside each other (a reported baseline, not the target: the roofline fraction is)
                g = {
                    "kdtree_500k": extra["kdtree"]["value"] / sec["kdtree_500k"]["value"],
                    "pcl_icp_iteration": sec["pcl_icp_500k"]["ms_per_iteration"] * 1e3 / extra["pcl_icp"]["us_per_iteration"],
                    "compute_normals": sec["compute_normals_640x480"]["value"] / extra["frame_prep"]["compute_normals_ms"],
                    "bilateral": sec["bilateral_640x480"]["value"] / extra["frame_prep"]["bilateral_filter_ms_host_to_host"],
                    "bench10_single_pair": sec["bench10"]["value"] / extra["named_shapes"]["bench10"]["single_pair_latency_ms"],
                }
                if "bench_icp" in sec and "bench_icp" in extra["named_shapes"]:
                    g["bench_icp_align"] = sec["bench_icp"]["value"] / extra["named_shapes"]["bench_icp"]["device_ms_per_align"]
                return g

            r = run_leg(errors, "gpu_vs_cpu", ratios)
            if r:
                extra["gpu_vs_cpu"] = r
    return cpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600, help="timed steps (600 x 3.4 ms = a 2 s timed region)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs-per-gpu", type=int, default=PAIRS_PER_STREAM)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cpu-pairs", type=int, default=64,
                    help="pairs run on the CPU oracle (rank 0, N=1 only): timed in chunk order until ~25 s of CPU work, and "
                         "(untimed) under --cpu-orders chunk-merge orders for the parity envelope")
    ap.add_argument("--cpu-orders", type=int, default=13, help="chunk-merge orders per pair of the parity envelope")
    ap.add_argument("--no-extras", action="store_true", help="skip the kd-tree / single-pair secondary numbers")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (gloo + --device 0 rehearses the multi-rank path on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device for this rank (default: LOCAL_RANK)")
    ap.add_argument("--shared-frames", action="store_true",
                    help="round-2 workload: pair p = frames (p, p + 1) of ONE 65-frame stream, so neighbouring pairs "
                         "share a frame.  Default: 2 P distinct frames, pair p = frames (2p, 2p + 1): independent pairs")
    ap.add_argument("--dump-gathered", default=None,
                    help="rank 0 writes the gathered [world * P, 16] pose matrices of the last step to this .npy file")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="run the N > 1 code path (process group, all-gather on the context stream) even with one "
                         "rank: checks the RCCL path on a one-GPU box (launch under torchrun --nproc-per-node 1)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:  # nobody launched ranks for us: do it here, before anything touches the GPU
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        log(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to report a wrong n_gpus")
        sys.exit(2)
    # Only the JSON line goes to stdout: RCCL's version banner and gloo's connection messages are printed to fd 1 by the
    # libraries themselves, so fd 1 points at stderr for the whole run and the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    device = local_rank if args.device is None else args.device
    use_nccl = args.backend == "nccl"
    multi = world > 1 or args.rehearse_collective  # the collective path
    if multi:
        import torch
        import torch.distributed as dist

        if use_nccl:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    # one rank per GPU: the node must show at least as many devices as there are local ranks, and this rank's context must
    # sit on ITS device before any work (a wrong LOCAL_RANK -> device mapping would silently stack ranks on one GPU)
    if use_nccl and args.device is None:
        from align3d_amd.multi import device_count

        n_dev = device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if n_dev < local_world:
            log(f"rank {rank}: {n_dev} HIP device(s) visible but {local_world} local ranks: refusing to stack ranks on a GPU")
            sys.exit(3)
    ctx = Context(device)
    if ctx.device() != device:
        log(f"rank {rank}: the context sits on HIP device {ctx.device()}, expected {device}")
        sys.exit(3)
    P, W, H = args.pairs_per_gpu, args.width, args.height
    params = MsIcpParams.repeat(3, IcpParams.default())  # ms3x15
    # ONE global list of world x P pairs, sharded in contiguous blocks.  Global pair j = frames
    # (j % P, j % P + 1) of stream 1000 + j // P, so a rank's block is (a slice of) one stream and needs P + 1 frames.
    lo, hi = shard_range(world * P, world, rank)
    assert hi - lo == P and lo % P == 0, "a rank's block is exactly one stream"
    stream_id = lo // P
    # Headline workload (configs[4]: INDEPENDENT pairs): 2 P distinct frames, pair p = (frame 2p -> frame 2p + 1), so no
    # array is read by two pairs.  --shared-frames: the round-2 workload, pair p = (frame p, frame p + 1) of a
    # (P + 1)-frame stream, where frame p + 1 is pair p's source and pair p + 1's target.
    distinct = not args.shared_frames
    pair_frames = [(2 * p, 2 * p + 1) for p in range(P)] if distinct else [(p, p + 1) for p in range(P)]
    n_frames = 2 * P if distinct else P + 1
    t0 = time.time()
    host_pyramids, poses_gt, build_ms = build_stream_pyramids(ctx, seed=1000 + stream_id, n_frames=n_frames, width=W,
                                                              height=H)
    if rank == 0:
        log(f"rendered and built {n_frames} synthetic frame pyramids in {time.time() - t0:.1f}s "
            f"({build_ms:.3f} ms per frame on the device, PCIe upload of depth + RGB included)")
    # pair p: target = its first frame, source = its second; every pyramid level resident in HBM
    targets = [host_pyramids[a] for a, _ in pair_frames]
    sources = [host_pyramids[b] for _, b in pair_frames]
    batch = MultiscaleAlignBatch(ctx, params, targets, sources)

    d_mats = gathered = ext_stream = host_mats = mats = None
    if multi and use_nccl:
        mats = torch.zeros((P, 16), dtype=torch.float32, device="cuda")
        d_mats = C.c_void_p(mats.data_ptr())
        ext_stream = torch.cuda.ExternalStream(ctx.lib.a3d_context_stream(ctx.handle))
    elif multi:  # gloo rehearsal: the poses go through host memory
        d_mats = ctx.malloc(P * 64)
        host_mats = np.zeros((P, 16), np.float32)

    def step():
        nonlocal gathered
        batch.enqueue(matrices_device=d_mats)
        if multi and use_nccl:
            with torch.cuda.stream(ext_stream):  # ordered after the kernels on the context stream
                gathered = gather_poses(mats)
        elif multi:
            ctx.to_host(d_mats, host_mats)
            gathered = gather_poses(torch.from_numpy(host_mats))

    def sync_all():
        if multi:
            dist.barrier()
            if use_nccl:
                torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if use_nccl else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * P * args.steps / elapsed

    out = None
    if rank == 0:
        errors = {}
        # ---- repeated short timings of the same step (>= 20 repetitions, min / median / max) ----
        rep_steps = max(1, min(30, args.steps))
        reps = [] if multi else (run_leg(errors, "repeated_timings", repeated_step_timings, ctx, batch, rep_steps) or [])
        r = run_leg(errors, "roofline", headline_roofline, ctx, batch, params, P, W, H, distinct, ms_per_step)
        iters = [int(p.max_iterations) for p in params]
        if r:
            roof, step_alg_bytes, iters = r
        else:  # (the contract's object even when the event-timed passes failed: the step's bytes over the timed step)
            step_alg_bytes = P * sum(iters[l] * level_bytes(W >> l, H >> l) for l in range(len(params)))
            ach = step_alg_bytes / (ms_per_step * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "image_icp_head_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "note": "from the timed step: the per-launch passes failed"}
        if not args.no_extras:  # the chip's own ceiling beside the nominal 8 TB/s (N > 1 too: VERDICT r5 item 2)
            run_leg(errors, "copy_ceiling", add_copy_ceiling, roof, device)
        poses, status = batch.align()
        failed_pairs = int(np.count_nonzero(status))
        roof["failed_pairs"] = failed_pairs  # a failed pair freezes and its blocks stop working: must be 0

        def live_fractions():
            # proof of work: the share of source pixels that pass every gate and reach the Jacobian stage, per level (one
            # accumulate pass per level at the pair's final pose; dead pixels skip that stage)
            for l, fr in enumerate(live_pixel_fractions(ctx, params, targets[0], sources[0], poses[0])):
                roof[f"level{l}_live_pixel_frac_pair0"] = fr

        run_leg(errors, "live_pixel_fractions", live_fractions)
        extra = {"failed_pairs": failed_pairs,
                 "timing": {"timed_region_s": elapsed, "ms_per_step_repeated": stats(reps) if reps else None,
                            "steps_per_repeat": rep_steps, "repeats_x_steps": f"{len(reps)} x {rep_steps}"}}
        if multi:  # the gathered buffer holds every rank's block in global pair order; this rank's starts at lo
            own = gathered[lo:hi].cpu().numpy().reshape(P, 4, 4)
            extra["gather_matches_local_poses"] = bool(
                all(np.allclose(own[p], poses[p].matrix(), atol=1e-6) for p in range(P)))
            extra["gathered_pairs"] = int(gathered.shape[0])
            if args.dump_gathered:
                np.save(args.dump_gathered, gathered.cpu().numpy())
        # accuracy against the synthetic ground truth (reported, not a parity claim)
        errs = []
        for p, (fa, fb) in enumerate(pair_frames):
            gt = synth.relative_pose(poses_gt[fa], poses_gt[fb])
            d = np.linalg.inv(gt) @ poses[p].matrix().astype(np.float64)
            errs.append((np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)), np.linalg.norm(d[:3, 3])))
        extra["mean_error_vs_synthetic_gt"] = {"angle_rad": float(np.mean([e[0] for e in errs])),
                                               "translation_m": float(np.mean([e[1] for e in errs]))}
        keep = {"level0_host": None, "depth0": None, "clouds": None, "bench_icp_clouds": None, "poses_pinned": None}
        if not args.no_extras and world == 1:
            keep = secondary_legs(ctx, errors, extra, params, targets, sources, P, W, H, build_ms, ms_per_step, step_alg_bytes)
        cpu = None
        if args.cpu_pairs > 0:
            cpu = cpu_legs(ctx, errors, extra, args, world, host_pyramids, pair_frames, params, P, poses, targets, sources, keep)
        if errors:
            extra["errors"] = errors
        out = {
            "metric": "ICP frame-pairs/sec (640x480, 3-lvl, 15 iters)",
            "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"ms3x15 x {P} pairs/GPU, {'distinct' if distinct else 'shared'} frames, {W}x{H}, "
                                    "3 lvl x 15 it, HBM-resident (configs[4] shard)"),
                       "workload_detail": f"{P} independent {W}x{H} frame pairs per GPU resident in HBM ("
                                          + ("2 P distinct frames: pair p = frames (2p, 2p+1), no array shared between pairs"
                                             if distinct else "P + 1 frames: pair p = frames (p, p+1), neighbours share a frame")
                                          + "), MsIcpParams::repeat(3, IcpParams::default()) = 3 levels x 15 iterations "
                                            "(configs[1] pair shape, batched as the per-GPU shard of configs[4])",
                       "distinct_frames": bool(distinct), "frames_resident_per_gpu": n_frames,
                       "pairs_per_gpu": P, "global_pairs": world * P, "levels": 3, "iterations_per_level": iters,
                       "ranks_in_collective": (dist.get_world_size() if multi else 1),
                       "gathered_pairs": (int(gathered.shape[0]) if multi else None),
                       "gather_matches_local_poses": extra.get("gather_matches_local_poses"),
                       "failed_pairs": failed_pairs,
                       "sharding": f"contiguous blocks of one {world * P}-pair list (shard_range): this rank [{lo}, {hi})",
                       "collective": (f"one all-gather of 16 f32 per pair per step ({'RCCL' if use_nccl else 'gloo rehearsal'})"
                                      if multi else "none")},
            "roofline": roof, "cpu_baseline": cpu, "extra": extra,
        }
    sync_all()
    batch.free()
    if multi:
        dist.destroy_process_group()
    if rank == 0:
        detail = None
        try:  # the full result (lists, per-level / per-pair tables, statistics): a file, not stdout
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            detail = os.path.join("gpurun_out", f"bench_detail_n{world}.json")
            with open(os.path.join(ROOT, detail), "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            log(f"could not write the detail file: {e}")
            detail = None
        sys.stdout.flush()
        os.write(json_fd, (compact_line(out, detail) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
