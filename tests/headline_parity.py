"""Parity of the headline workload (ms3x15 on the benchmark's own pairs) against the CPU oracle, pair by pair.
TEST INFRASTRUCTURE: used by tests/test_gpu_headline_parity.py and by bench.py's untimed cpu_baseline leg; the product
never imports it.

ms3x15 runs IcpParams::default() on every level, which is not contractive on every pair: the
reference's own result then depends on the order in which rayon's par_bridge() delivers the 75 chunk accumulators
(src/icp/image_icp.rs:96,143-148).  So a pair is compared three ways:
  * envelope(): GPU pose against the oracle's chunk-order run AND against the oracle's own spread over >= 13 seeded
    merge orders (every permutation is a legitimate reference result);
  * teacher_forced(): at EVERY iteration of every level, from the oracle's own transform, the GPU accumulators against
    the oracle's f64-summed ones (counts exact, sums <= 1e-6) and the GPU's one-iteration pose against the oracle's;
  * the growth of |T_gpu - T_oracle| along the free-running trajectories, iteration by iteration.
"""
import dataclasses

import numpy as np

import oracle_lib as O
from align3d_amd import ImageIcp, Transform


def host_frame(dev_level):
    """The very arrays the GPU path reads, as an oracle Frame."""
    ri = dev_level.download(colors=False)
    k = ri.intrinsics
    return O.Frame(ri.points, ri.mask, k.fx, k.fy, k.cx, k.cy, ri.normals, ri.intensities, ri.intensity_map)


def oracle_runs(params, target, source, threads, orders):
    """MultiscaleAlign on the oracle for merge orders 0 (chunk order) .. orders - 1 (seeded permutations)."""
    parr = params.to_c_array()
    runs = []
    try:
        for seed in range(orders):
            O.set_chunk_merge_order(seed)
            st, T = O.multiscale_align(parr, len(params), target, source, threads=threads)
            if st != 0:
                raise RuntimeError(f"oracle multiscale_align failed with status {st}")
            runs.append(T)
    finally:
        O.set_chunk_merge_order(0)
    return runs


def oracle_distance_table(runs):
    return np.array([[[abs(x[0]), x[1]] for x in (O.transform_metrics(a, b) for b in runs)] for a in runs])


def envelope(gpu_pose, runs):
    """GPU against the oracle's runs.  `rank`: how many of the other merge orders lie farther from the chunk-order run
    than the GPU does (orders - 1 = the GPU is closer to it than every alternative order; 0 = farther than all)."""
    return envelope_from_distances(gpu_pose, runs, oracle_distance_table(runs))


def _envelope(to_runs, among, n_runs):
    from_first = among[0, 1:]
    return {
        "gpu_vs_cpu_angle_rad": float(to_runs[0, 0]), "gpu_vs_cpu_translation_m": float(to_runs[0, 1]),
        "gpu_to_nearest_cpu_run_angle_rad": float(to_runs[:, 0].min()),
        "gpu_to_nearest_cpu_run_translation_m": float(to_runs[:, 1].min()),
        "cpu_spread_angle_rad": float(among[..., 0].max()), "cpu_spread_translation_m": float(among[..., 1].max()),
        "cpu_other_orders_vs_chunk_order_angle_rad": float(from_first[:, 0].max()) if len(from_first) else 0.0,
        "cpu_other_orders_vs_chunk_order_translation_m": float(from_first[:, 1].max()) if len(from_first) else 0.0,
        "gpu_rank_inside_cpu_spread": int(np.sum(from_first[:, 1] > to_runs[0, 1])) if len(from_first) else 0,
        # the tighter pairing: the GPU's distance to the chunk-order run against the MEDIAN distance
        # of the other merge orders to that same run (same reference point on both sides, not nearest-vs-max)
        "cpu_median_other_order_vs_chunk_order_angle_rad": float(np.median(from_first[:, 0])) if len(from_first) else 0.0,
        "cpu_median_other_order_vs_chunk_order_translation_m": float(np.median(from_first[:, 1])) if len(from_first) else 0.0,
        "merge_orders": n_runs,
    }


def within_tolerance(e, tol=1e-4):
    return e["gpu_vs_cpu_angle_rad"] <= tol and e["gpu_vs_cpu_translation_m"] <= tol


def inside_envelope(e):
    """No farther from the nearest oracle run than the oracle's runs are from each other."""
    return (e["gpu_to_nearest_cpu_run_angle_rad"] <= e["cpu_spread_angle_rad"]
            and e["gpu_to_nearest_cpu_run_translation_m"] <= e["cpu_spread_translation_m"])


def within_median_order(e):
    """The GPU is no farther from the oracle's chunk-order run than the median alternative merge order is."""
    return (e["gpu_vs_cpu_angle_rad"] <= e["cpu_median_other_order_vs_chunk_order_angle_rad"]
            and e["gpu_vs_cpu_translation_m"] <= e["cpu_median_other_order_vs_chunk_order_translation_m"])


def envelope_from_distances(gpu_pose, runs, among):
    """envelope() with the oracle-to-oracle distance table precomputed (a second GPU result against the same runs)."""
    g = gpu_pose.to_c()
    to_runs = np.array([[abs(a), t] for a, t in (O.transform_metrics(g, r) for r in runs)])
    return _envelope(to_runs, among, len(runs))


def _pose_dist(a, b):
    ang, tr = O.transform_metrics(a, b)
    return abs(ang), tr


def teacher_forced(ctx, params, target_host, source_host, target_dev, source_dev, threads=8):
    """Every iteration of every level of one pair.  The oracle runs level by level (chunk order) with a trace; at
    iteration k of a level, from the oracle's transform BEFORE that iteration:
      counts of both accumulators: GPU == oracle exactly; H, g, sum r^2: GPU vs the f64-summed oracle (relative);
      one_step: |GPU one-iteration result - oracle's transform after iteration k| (rad, m).
    And the free-running GPU trajectory (each level started from the GPU's own previous result) against the oracle's,
    iteration by iteration.  Returns a dict with the per-iteration rows and the maxima."""
    from gpu_util import gn_rel_err

    L = len(params)
    rows = []
    T_or = O.pose()   # oracle: Transform::eye() (multiscale.rs:52)
    T_gpu = Transform.eye()
    worst = {"count_mismatches": 0, "max_rel_err_sums": 0.0, "max_one_step_angle_rad": 0.0, "max_one_step_translation_m": 0.0}
    for l in reversed(range(L)):  # coarsest first (multiscale.rs:54-60)
        prm = params[l]
        n_it = int(prm.max_iterations)
        st, T_level, trace = O.image_icp_align(prm.to_c(), target_host[l], source_host[l], init=T_or, threads=threads,
                                               want_trace=True)
        assert st == 0
        icp = ImageIcp.new(ctx, prm, target_dev[l])
        icp.initial_transform = T_gpu
        T_gpu_level, gtrace = icp.align(source_dev[l], trace=True)
        icp1 = ImageIcp.new(ctx, dataclasses.replace(prm, max_iterations=1), target_dev[l])
        T_in = Transform.from_c(T_or)
        for it in range(n_it):
            g_gpu, c_gpu = icp.accumulate(source_dev[l], T_in)
            st, g_ref, c_ref = O.image_icp_accumulate(prm.to_c(), target_host[l], source_host[l], T_in.to_c(),
                                                      accum_f64=True)
            assert st == 0
            g_ref, c_ref = g_ref.as_dict(), c_ref.as_dict()
            mism = int(g_gpu["count"] != g_ref["count"]) + int(c_gpu["count"] != c_ref["count"])
            rel = max(max(gn_rel_err(g_gpu, g_ref)), max(gn_rel_err(c_gpu, c_ref)))
            icp1.initial_transform = T_in
            T_one = icp1.align(source_dev[l])
            T_after = Transform(trace[it, 1:4], trace[it, 4:8])
            os_ang, os_tr = _pose_dist(T_one.to_c(), T_after.to_c())
            G_after = Transform(gtrace[it, 1:4], gtrace[it, 4:8])
            fr_ang, fr_tr = _pose_dist(G_after.to_c(), T_after.to_c())
            rows.append({"level": l, "iteration": it, "count_geom": int(g_ref["count"]), "count_color": int(c_ref["count"]),
                         "count_mismatch": mism, "rel_err_sums": float(rel),
                         "one_step_angle_rad": float(os_ang), "one_step_translation_m": float(os_tr),
                         "free_running_angle_rad": float(fr_ang), "free_running_translation_m": float(fr_tr),
                         "oracle_residual": float(trace[it, 0]), "gpu_residual": float(gtrace[it, 0])})
            worst["count_mismatches"] += mism
            worst["max_rel_err_sums"] = max(worst["max_rel_err_sums"], float(rel))
            worst["max_one_step_angle_rad"] = max(worst["max_one_step_angle_rad"], float(os_ang))
            worst["max_one_step_translation_m"] = max(worst["max_one_step_translation_m"], float(os_tr))
            T_in = T_after
        T_or, T_gpu = T_level, T_gpu_level  # best_transform of the level starts the next one
    # growth of the free-running difference per iteration (ratio to the previous iteration's, where that is not tiny)
    growth = []
    for a, b in zip(rows[:-1], rows[1:]):
        if a["free_running_translation_m"] > 1e-7:
            growth.append(b["free_running_translation_m"] / a["free_running_translation_m"])
    end_ang, end_tr = _pose_dist(T_gpu.to_c(), T_or)
    return dict(worst, iterations=len(rows), rows=rows,
                median_growth_per_iteration=float(np.median(growth)) if growth else None,
                max_growth_per_iteration=float(np.max(growth)) if growth else None,
                level_by_level_end_angle_rad=float(end_ang), level_by_level_end_translation_m=float(end_tr),
                oracle_final=(list(T_or.t[:]), list(T_or.q[:])))


def summarize(entries):
    """Flat scalars over the per-pair envelope entries (what bench.py puts into cpu_baseline)."""
    over = [e for e in entries if not within_tolerance(e)]
    return {
        "pairs_compared": len(entries),
        "merge_orders_per_pair": entries[0]["merge_orders"] if entries else 0,
        "pairs_over_1e-4": len(over),
        "pairs_over_1e-4_and_outside_the_cpu_envelope": sum(1 for e in over if not inside_envelope(e)),
        "pairs_over_1e-4_and_beyond_the_median_cpu_order": sum(1 for e in over if not within_median_order(e)),
        "pairs_over_1e-4_and_farther_than_every_cpu_order": sum(1 for e in over if e["gpu_rank_inside_cpu_spread"] == 0),
        "ranks_of_pairs_over_1e-4": [e["gpu_rank_inside_cpu_spread"] for e in over],
        "max_gpu_vs_cpu_angle_rad": max((e["gpu_vs_cpu_angle_rad"] for e in entries), default=0.0),
        "max_gpu_vs_cpu_translation_m": max((e["gpu_vs_cpu_translation_m"] for e in entries), default=0.0),
        "median_gpu_vs_cpu_translation_m": float(np.median([e["gpu_vs_cpu_translation_m"] for e in entries])) if entries else 0.0,
        "max_cpu_spread_angle_rad": max((e["cpu_spread_angle_rad"] for e in entries), default=0.0),
        "max_cpu_spread_translation_m": max((e["cpu_spread_translation_m"] for e in entries), default=0.0),
        "pairs_whose_cpu_spread_exceeds_1e-4": sum(1 for e in entries if e["cpu_spread_translation_m"] > 1e-4
                                                   or e["cpu_spread_angle_rad"] > 1e-4),
    }
