"""bench.py's stdout line must stay something the driver can parse: round 4's carried per-pair parity arrays and
teacher-forced rows (92 KB) and was recorded as `parsed: null`.  The line is rebuilt here from that very result."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _round4_full_result():
    return json.loads(open(os.path.join(ROOT, "profiles", "round4_bench.json")).read())


def test_line_from_the_round4_result_fits_the_budget_and_keeps_the_contract():
    full = _round4_full_result()
    assert len(json.dumps(full)) > 80_000  # the line that could not be parsed
    text = bench.compact_line(full, "gpurun_out/bench_detail_n1.json")
    assert len(text) < bench.LINE_BUDGET_BYTES and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert abs(line["value"] / full["value"] - 1) < 1e-5
    # the parity verdict survives as scalars, the tables do not
    assert line["cpu_baseline"]["pairs_compared"] == 64
    assert line["cpu_baseline"]["pairs_over_1e-4_and_outside_the_cpu_envelope"] == 0

    def no_tables(o, depth=0):
        if isinstance(o, list):
            assert len(o) <= 8 and all(not isinstance(x, (dict, list)) for x in o)
        elif isinstance(o, dict):
            assert depth < 2, "flat: at most one level of nesting below the top"
            for v in o.values():
                no_tables(v, depth + 1)

    no_tables(line)
    assert line["detail_file"] == "gpurun_out/bench_detail_n1.json"


def test_line_never_exceeds_the_budget_even_with_absurd_extras():
    full = _round4_full_result()
    full["config"]["workload"] = "w" * 200
    for i in range(400):
        full["extra"].setdefault("kdtree", {})["value"] = 1.0
    bench._EXTRA_SCALARS.extend((f"pad{i}", "extra.kdtree.value") for i in range(600))
    try:
        text = bench.compact_line(full, None)
    finally:
        del bench._EXTRA_SCALARS[-600:]
    assert len(text) < bench.LINE_BUDGET_BYTES
    assert json.loads(text)["roofline"]["frac"] > 0


def test_traffic_profile_is_chosen_by_recorded_keys_not_by_file_name():
    t, src = bench.measured_traffic("normals", frames=64, pixels=307200, layout="xcd_contiguous")
    assert src.endswith("_normals_traffic.json") and abs(t - 491.7e6) < 1e6  # not the 657 MB plain-order pass
    t2, src2 = bench.measured_traffic("normals", frames=64, pixels=307200, layout="plain_order")
    assert src2.endswith("plain_order.json") and t2 > t
    assert bench.measured_traffic("normals", frames=63, pixels=307200, layout="xcd_contiguous") == (None, None)


def test_line_from_the_round5_detail_matches_the_line_that_was_printed():
    """The round-5 detail file (the full result `main` wrote beside the line) rebuilds the very line the run printed."""
    detail = os.path.join(ROOT, "profiles", "round5_bench_detail.json")
    printed = os.path.join(ROOT, "profiles", "round5_bench.json")
    if not (os.path.exists(detail) and os.path.exists(printed)):
        import pytest

        pytest.skip("round-5 profiles not present")
    full = json.loads(open(detail).read())
    was = json.loads(open(printed).read())
    now = json.loads(bench.compact_line(full, was.get("detail_file")))
    assert len(json.dumps(now)) < bench.LINE_BUDGET_BYTES
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "cpu_baseline"):
        assert now[k] == was[k], k
    # the secondary workloads the round added are on the line as scalars
    for k in ("kdtree_build_kernel_ms", "kdtree_build_frac", "pcl_icp_new_plus_align_device_ms", "bench_icp_new_plus_align_device_ms",
              "frame_build_traffic_per_frame"):
        assert k in now["extra"], k
