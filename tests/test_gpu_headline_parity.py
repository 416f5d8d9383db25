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


def test_all_64_headline_pairs_against_the_oracle_and_its_own_envelope(ctx):
    import bench

    P = 64
    pyr, _, _ = bench.build_stream_pyramids(ctx, 1000, 2 * P, 640, 480)
    prm = MsIcpParams.repeat(3, IcpParams.default())
    pairs = [(2 * p, 2 * p + 1) for p in range(P)]
    batch = MultiscaleAlignBatch(ctx, prm, [pyr[a] for a, _ in pairs], [pyr[b] for _, b in pairs])
    poses, status = batch.align()
    batch.free()
    assert not np.any(status)
    _, cores = bench.cpu_info()
    entries, hosts = [], {}
    for p, (a, b) in enumerate(pairs):
        ta, tb = [HP.host_frame(lv) for lv in pyr[a]], [HP.host_frame(lv) for lv in pyr[b]]
        runs = HP.oracle_runs(prm, ta, tb, cores, ORDERS)
        e = HP.envelope(poses[p], runs)
        e["pair"] = p
        entries.append(e)
        hosts[p] = (ta, tb)
        if not HP.within_tolerance(e):
            print(f"[headline pair {p}] GPU vs oracle {e['gpu_vs_cpu_angle_rad']:.2e} rad {e['gpu_vs_cpu_translation_m']:.2e} m; "
                  f"nearest of {ORDERS} oracle runs {e['gpu_to_nearest_cpu_run_translation_m']:.2e} m; oracle spread "
                  f"{e['cpu_spread_angle_rad']:.2e} rad {e['cpu_spread_translation_m']:.2e} m; rank {e['gpu_rank_inside_cpu_spread']}")
        if len(hosts) > 6:  # keep the host copies of the most sensitive pairs only (12 MB per frame)
            keep = sorted(hosts, key=lambda q: -entries[q]["gpu_vs_cpu_translation_m"])[:3]
            hosts = {q: hosts[q] for q in keep}
    s = HP.summarize(entries)
    print("[headline parity]", s)
    # (a) or (b) for every pair
    for e in entries:
        assert HP.within_tolerance(e) or HP.inside_envelope(e), e
    # wherever the GPU is beyond 1e-4, the reference's own spread is beyond 1e-4 as well
    for e in entries:
        if not HP.within_tolerance(e):
            assert e["cpu_spread_translation_m"] > 1e-4 or e["cpu_spread_angle_rad"] > 1e-4, e
    assert s["pairs_over_1e-4"] <= P // 8  # the sensitive pairs are the exception
    # the two most sensitive pairs, at all 45 iterations
    worst = sorted(range(P), key=lambda q: -entries[q]["gpu_vs_cpu_translation_m"])[:2]
    for p in worst:
        a, b = pairs[p]
        ta, tb = hosts[p]
        tf = HP.teacher_forced(ctx, prm, ta, tb, pyr[a], pyr[b], threads=cores)
        print(f"[headline pair {p}, teacher-forced at {tf['iterations']} iterations] count mismatches {tf['count_mismatches']}, "
              f"sums <= {tf['max_rel_err_sums']:.1e}, one step <= {tf['max_one_step_angle_rad']:.1e} rad "
              f"{tf['max_one_step_translation_m']:.1e} m; free-running growth per iteration median "
              f"{tf['median_growth_per_iteration']}, max {tf['max_growth_per_iteration']}")
        for r in tf["rows"]:
            print(f"    level {r['level']} it {r['iteration']:2d}: one step {r['one_step_translation_m']:.1e} m, free-running "
                  f"{r['free_running_angle_rad']:.1e} rad {r['free_running_translation_m']:.1e} m, sums {r['rel_err_sums']:.1e}")
        assert tf["iterations"] == 45 and tf["count_mismatches"] == 0
        assert tf["max_rel_err_sums"] <= 1e-6
        assert tf["max_one_step_angle_rad"] <= 5e-6 and tf["max_one_step_translation_m"] <= 5e-6
    for lv in (lv for q in pyr for lv in q):
        lv.free()
