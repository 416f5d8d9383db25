"""the headline's own 64 pairs (bench.py default: stream 1000, pair p = frames 2p -> 2p + 1, ms3x15 =
MsIcpParams::repeat(3, IcpParams::default())), every one of them against the CPU oracle.

The north-star tolerance is 1e-4 rad / 1e-4 m on the output transform.  ms3x15 is not contractive on every pair (SURVEY
§0-11, §10: the w-vs-w^2 weighting of add_weighted over-steps the colour term), and on such a pair the REFERENCE does not
reproduce itself to 1e-4: rayon's par_bridge() delivers the 75 chunk accumulators in arbitrary order and they are added in
that order (src/icp/image_icp.rs:96,143-148).  So each pair must be
  (a) within 1e-4 of the oracle's chunk-order run, or
  (b) no farther from the nearest of 13 oracle runs (13 seeded merge orders) than those runs are from each other, AND —
      for the most sensitive pairs — shown, at all 45 iterations, to be an amplification of per-step differences of the
      size of f32 round-off: teacher-forced from the oracle's own transform the GPU's inlier counts are exact, its sums
      within 1e-6 of the f64-summed oracle and its one-iteration pose within 5e-6 of the oracle's."""
import numpy as np
import pytest

import headline_parity as HP
from align3d_amd import IcpParams, MsIcpParams, MultiscaleAlignBatch

pytestmark = pytest.mark.gpu

ORDERS = 13
P = 64


@pytest.fixture(scope="module")
def headline(ctx):
    """The 128 resident frames of the headline batch and, per pair, the oracle's 13 runs (832 oracle alignments, once for
    both tilings) with their pairwise distance table."""
    import bench

    pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2 * P, 640, 480)
    prm = MsIcpParams.repeat(3, IcpParams.default())
    pairs = [(2 * p, 2 * p + 1) for p in range(P)]
    _, cores = bench.cpu_info()
    runs, tables = [], []
    for a, b in pairs:
        ta, tb = [HP.host_frame(lv) for lv in pyr[a]], [HP.host_frame(lv) for lv in pyr[b]]
        r = HP.oracle_runs(prm, ta, tb, cores, ORDERS)
        runs.append(r)
        tables.append(HP.oracle_distance_table(r))
    yield {"pyr": pyr, "prm": prm, "pairs": pairs, "cores": cores, "runs": runs, "tables": tables}
    for lv in (lv for q in pyr for lv in q):
        lv.free()


@pytest.mark.parametrize("tiles", [0, 24], ids=["throughput_tiling", "pinned_tiling_24"])
def test_all_64_headline_pairs_against_the_oracle_and_its_own_envelope(ctx, headline, tiles):
    """`tiles` = 24: the same under a3d_context_set_tiling(24), the mode a caller who wants reproducible poses uses (every
    pair cut into 24 blocks per level whatever the batch: a different association of the f32 sums than the default)."""
    pyr, prm, pairs, cores = headline["pyr"], headline["prm"], headline["pairs"], headline["cores"]
    ctx.set_tiling(tiles)
    try:
        batch = MultiscaleAlignBatch(ctx, prm, [pyr[a] for a, _ in pairs], [pyr[b] for _, b in pairs])
        poses, status = batch.align()
        batch.free()
    finally:
        ctx.set_tiling(0)
    assert not np.any(status)
    entries = []
    for p in range(P):
        e = HP.envelope_from_distances(poses[p], headline["runs"][p], headline["tables"][p])
        e["pair"] = p
        entries.append(e)
        if not HP.within_tolerance(e):
            print(f"[headline pair {p}, tiles {tiles}] GPU vs oracle {e['gpu_vs_cpu_angle_rad']:.2e} rad {e['gpu_vs_cpu_translation_m']:.2e} m; "
                  f"nearest of {ORDERS} oracle runs {e['gpu_to_nearest_cpu_run_translation_m']:.2e} m; oracle spread "
                  f"{e['cpu_spread_angle_rad']:.2e} rad {e['cpu_spread_translation_m']:.2e} m; other orders vs chunk order: median "
                  f"{e['cpu_median_other_order_vs_chunk_order_translation_m']:.2e} m, max "
                  f"{e['cpu_other_orders_vs_chunk_order_translation_m']:.2e} m; rank {e['gpu_rank_inside_cpu_spread']} of {ORDERS - 1}")
    s = HP.summarize(entries)
    print(f"[headline parity, tiles {tiles}]", s)
    # (a) or (b) for every pair
    for e in entries:
        assert HP.within_tolerance(e) or HP.inside_envelope(e), e
    for e in entries:
        if not HP.within_tolerance(e):
            # wherever the GPU is beyond 1e-4, the reference's own spread is beyond 1e-4 as well ...
            assert e["cpu_spread_translation_m"] > 1e-4 or e["cpu_spread_angle_rad"] > 1e-4, e
            # ... and, measured from the SAME reference point (the chunk-order run), the GPU is not the outlier: at least
            # one of the 12 alternative merge orders of the reference itself lies farther from it (rank 0 fails); whether it
            # is also within the median order is reported (`pairs_over_1e-4_and_beyond_the_median_cpu_order`)
            assert e["gpu_rank_inside_cpu_spread"] >= 1, e
    assert s["pairs_over_1e-4"] <= P // 8  # the sensitive pairs are the exception
    assert s["pairs_over_1e-4_and_farther_than_every_cpu_order"] == 0
    # the two most sensitive pairs, at all 45 iterations
    worst = sorted(range(P), key=lambda q: -entries[q]["gpu_vs_cpu_translation_m"])[:2]
    ctx.set_tiling(tiles)
    try:
        for p in worst:
            a, b = pairs[p]
            ta, tb = [HP.host_frame(lv) for lv in pyr[a]], [HP.host_frame(lv) for lv in pyr[b]]
            tf = HP.teacher_forced(ctx, prm, ta, tb, pyr[a], pyr[b], threads=cores)
            print(f"[headline pair {p}, tiles {tiles}, teacher-forced at {tf['iterations']} iterations] count mismatches "
                  f"{tf['count_mismatches']}, sums <= {tf['max_rel_err_sums']:.1e}, one step <= {tf['max_one_step_angle_rad']:.1e} rad "
                  f"{tf['max_one_step_translation_m']:.1e} m; free-running growth per iteration median "
                  f"{tf['median_growth_per_iteration']}, max {tf['max_growth_per_iteration']}")
            if tiles == 0:
                for r in tf["rows"]:
                    print(f"    level {r['level']} it {r['iteration']:2d}: one step {r['one_step_translation_m']:.1e} m, free-running "
                          f"{r['free_running_angle_rad']:.1e} rad {r['free_running_translation_m']:.1e} m, sums {r['rel_err_sums']:.1e}")
            assert tf["iterations"] == 45 and tf["count_mismatches"] == 0
            assert tf["max_rel_err_sums"] <= 1e-6
            assert tf["max_one_step_angle_rad"] <= 5e-6 and tf["max_one_step_translation_m"] <= 5e-6
    finally:
        ctx.set_tiling(0)


def test_all_64_headline_pairs_literally_within_tolerance_under_msdefault(ctx, headline):
    """the SAME 64 synthetic pairs whose throughput is the headline, under the reference's own
    contractive parameter set MsIcpParams::default() (src/icp/icp_params.rs:112-133: 20 / 20 / 30 iterations, weight 1,
    max_color_distance 2.75): every pair LITERALLY within 1e-4 rad / 1e-4 m of the oracle's chunk-order run — no envelope.
    This is what shows that the three ms3x15 outliers belong to that parameter set (IcpParams::default() on every level is
    not contractive) and not to the kernel."""
    pyr, pairs, cores = headline["pyr"], headline["pairs"], headline["cores"]
    prm = MsIcpParams.default()
    batch = MultiscaleAlignBatch(ctx, prm, [pyr[a] for a, _ in pairs], [pyr[b] for _, b in pairs])
    poses, status = batch.align()
    batch.free()
    assert not np.any(status)
    worst_ang = worst_tr = 0.0
    over = 0
    for p, (a, b) in enumerate(pairs):
        ta, tb = [HP.host_frame(lv) for lv in pyr[a]], [HP.host_frame(lv) for lv in pyr[b]]
        run = HP.oracle_runs(prm, ta, tb, cores, 1)[0]
        ang, tr = O_metrics(poses[p], run)
        worst_ang, worst_tr = max(worst_ang, ang), max(worst_tr, tr)
        over += int(ang > 1e-4 or tr > 1e-4)
    print(f"[headline pairs under msdefault] 64 pairs: max d_angle={worst_ang:.2e} rad, max d_trans={worst_tr:.2e} m, over 1e-4: {over}")
    assert over == 0 and worst_ang <= 1e-4 and worst_tr <= 1e-4


def O_metrics(gpu_pose, oracle_pose):
    import oracle_lib as O

    ang, tr = O.transform_metrics(gpu_pose.to_c(), oracle_pose)
    return abs(ang), tr
