"""CPU: the committed evidence under profiles/r3/ is internally consistent with the bench contract -- the JSON lines carry every
field the contract names, the roofline block is arithmetic on its own fields, the kernel it names is the one in the committed
rocprofv3 stats and in the PMC file, and the live HIP-event timing agrees with the profiler's average for that kernel."""
import csv
import json
import os

import pytest

from tests.conftest import ROOT

P = os.path.join(ROOT, "profiles", "r3")
LINES = ["bench.json", "bench_fp32.json", "bench_bf16x3.json"]


def _load(name):
    return json.load(open(os.path.join(P, name)))


@pytest.mark.parametrize("name", LINES)
def test_bench_line_follows_the_contract(name):
    d = _load(name)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6                       # value = 1 / time per frame at N = 1
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["commit"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] * 1e9 - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6)) < 1e-3 * r["achieved"] * 1e9
    rb = d["roofline_backbone"]
    assert rb["bound"] == "mfma" and abs(rb["frac"] - rb["achieved"] / rb["peak"]) < 1e-9
    assert d["dtype"] == {"bench.json": "bf16", "bench_fp32.json": "fp32", "bench_bf16x3.json": "bf16x3"}[name]


def test_headline_line_has_cpu_baseline_and_parity():
    d = _load("bench.json")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and 0 < c["value"] < 1.0
    p = d["parity"]
    assert p["tolerance_north_star"] == 1e-3 and p["meets_tolerance"] is False and p["logit_max_abs"] > 1e-3   # bf16: measured, not met
    x3 = _load("bench_bf16x3.json")["parity"]
    assert x3["meets_tolerance"] is True and x3["logit_max_abs"] < 1e-3
    # the SAME driver-style run also times an engine that is inside the tolerance: a first-class block
    t = d["in_tolerance"]
    assert t["dtype"] == "bf16x3" and t["meets_tolerance"] is True and t["parity"]["logit_max_abs"] < 1e-3
    assert t["unit"] == "samples/s" and abs(t["value"] * t["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert t["value"] >= 70.0, "the in-tolerance engine fell below the round's 70 samples/s bar: %.1f" % t["value"]
    rb = t["roofline_backbone"]
    assert rb["bound"] == "mfma" and abs(rb["frac"] - rb["achieved"] / rb["peak"]) < 1e-9 and abs(rb["peak"] - 2500.0 / 3) < 1e-6
    # the like-for-like ratio against the reference's sync-per-frame protocol is printed next to vs_baseline
    assert abs(d["vs_baseline_sync_per_frame"] - d["protocol"]["sync_per_frame"]["samples_per_s_mean"] / 6.4) < 1e-9


def test_roofline_kernel_matches_the_committed_profiles():
    d = _load("bench.json")
    kernel = d["roofline"]["kernel"].split(" ")[0]
    pmc = _load("aggregate_pmc.json")
    assert pmc["kernel"] == kernel and pmc["commit"] == d["config"]["commit"]
    assert d["roofline"]["traffic_measured_in_this_run"] is False
    assert d["roofline"]["traffic"] is None or abs(d["roofline"]["traffic"] - pmc["hbm_bytes_per_launch"]) / pmc["hbm_bytes_per_launch"] < 0.02
    rows = [r for r in csv.DictReader(open(os.path.join(P, "bench_kernel_stats.csv"))) if kernel in r["Name"]]
    assert len(rows) == 1, [r["Name"][:40] for r in rows]
    avg_us = float(rows[0]["AverageNs"]) * 1e-3
    live_us = d["roofline"]["avg_launch_us"]
    # the profiler's in-frame average and the live back-to-back timing of the same launches agree within 10 %
    assert abs(avg_us - live_us) / live_us < 0.10, (avg_us, live_us)
    # the fp32-row modes time the same kernel on 1-KiB rows and do not borrow the bf16 traffic figure
    for name in ("bench_fp32.json", "bench_bf16x3.json"):
        assert _load(name)["roofline"]["traffic"] is None


def test_threshold_mode_and_stage_time_evidence():
    thr = _load("bench_threshold.json")
    assert thr["config"]["proposals"] == "threshold" and thr["value"] > 30.0 and "count on the device" in thr["config"]["workload"]
    txt = open(os.path.join(P, "stage_times_bf16.txt")).read()
    st = json.loads(txt[txt.index("{"):])
    # the query-sharded decoder's measured non-result: an eighth of the queries costs as much as all of them (latency floors)
    assert st["decoder_query_share_1_of_8_ms"] > 0.85 * st["decoder_all_queries_ms"]
    assert st["camera_stages_1cam_ms"] < st["camera_stages_2cam_ms"] < st["camera_stages_4cam_ms"] < st["camera_stages_7cam_ms"]


def test_gpu_suite_and_smoke_logs_are_green():
    txt = open(os.path.join(P, "pytest_gpu.txt")).read()
    assert " passed" in txt and "failed" not in txt and "error" not in txt.lower()
    assert "smoke ok" in open(os.path.join(P, "smoke.txt")).read()
