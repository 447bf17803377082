"""CPU: the AV2 detection metric (far3d_amd.data_pipeline.av2_metric).

(a) against tests/golden/far3d_av2_metric.npz -- per-box TP flags / ATE / ASE / AOE / is_evaluated and the summary table produced by
    the REFERENCE's own accumulate / summarize_metrics (tools/gen_golden_metric.py; av2-package helpers were stand-ins there, see
    the module header: parity for those helpers is unpinned);
(b) closed-form cases: perfect detections, a known centre shift, a known yaw error (incl. the pi-fold of wrap_angles), a known size
    ratio, the 100-detection cap, the range filter, and the ROI switch that cannot be served without map data."""
import math
import os

import numpy as np
import pandas as pd
import pytest

from far3d_amd.data_pipeline import av2_metric as M
from tests.conftest import ROOT

Z = np.load(os.path.join(ROOT, "tests", "golden", "far3d_av2_metric.npz"))


def _tables():
    U = list(M.UUID_COLUMN_NAMES)
    def tab(uuid, num, last):
        df = pd.DataFrame(num, columns=list(M.ORDERED_CUBOID_COL_NAMES) + [last])
        for j, c in enumerate(U):
            df[c] = uuid[:, j]
        df["timestamp_ns"] = df["timestamp_ns"].astype(int)
        return df
    return tab(Z["dts_uuid"], Z["dts_num"], "score"), tab(Z["gts_uuid"], Z["gts_num"], "num_interior_pts")


def test_matches_the_reference_functions_on_the_fixture():
    dts, gts = _tables()
    cfg = M.DetectionCfg(categories=tuple(str(c) for c in Z["categories"]))
    d, g, metrics, recall = M.evaluate(dts, gts, cfg)
    cols = list(cfg.affinity_thresholds_m) + ["ATE", "ASE", "AOE", "is_evaluated"]
    # same uuid-sorted row order as the reference's evaluate (score identifies a detection inside its group)
    assert (d[list(M.UUID_COLUMN_NAMES)].to_numpy().astype(str) == Z["dts_sorted_uuid"]).all()
    assert np.array_equal(d["score"].to_numpy(), Z["dts_sorted_score"])
    assert np.allclose(d[cols].to_numpy(dtype=float), Z["dts_metrics"], atol=1e-9)
    assert (g[list(M.UUID_COLUMN_NAMES)].to_numpy().astype(str) == Z["gts_sorted_uuid"]).all()
    assert np.allclose(g[cols].to_numpy(dtype=float), Z["gts_metrics"], atol=1e-9)
    assert list(metrics.index) == [str(s) for s in Z["metrics_index"]] and list(metrics.columns) == [str(s) for s in Z["metrics_columns"]]
    assert np.allclose(metrics.to_numpy(dtype=float), Z["metrics"], atol=1e-9)
    assert np.allclose(recall.to_numpy(dtype=float), Z["recall"], atol=1e-9)
    assert d["is_evaluated"].sum() > 100 and d[2.0].sum() > 20          # the fixture is not vacuous


def _boxes(n, seed=0):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(-60, 60, size=(n, 3)) * np.array([1, 1, 0.02])
    dims = rng.uniform(0.5, 5.0, size=(n, 3))
    yaw = rng.uniform(-3.0, 3.0, size=n)
    return ctr, dims, yaw


def _frame(ctr, dims, yaw, last_name, last, cat="BUS", log="a", ts=5):
    q = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], axis=-1)
    df = pd.DataFrame(np.concatenate([ctr, dims, q, np.asarray(last, dtype=float)[:, None]], axis=1),
                      columns=list(M.ORDERED_CUBOID_COL_NAMES) + [last_name])
    df["log_id"], df["timestamp_ns"], df["category"] = log, ts, cat
    return df


def test_perfect_detections_score_one():
    ctr, dims, yaw = _boxes(12)
    cfg = M.DetectionCfg(categories=("BUS",))
    _, _, m, r = M.evaluate(_frame(ctr, dims, yaw, "score", np.linspace(0.9, 0.1, 12)), _frame(ctr, dims, yaw, "num_interior_pts", np.full(12, 5)), cfg)
    row = m.loc["BUS"]
    assert row["AP"] == 1.0 and row["ATE"] == 0.0 and row["ASE"] == 0.0 and row["AOE"] == 0.0 and row["CDS"] == 1.0 and row["RECALL"] == 1.0
    assert (m.loc["AVERAGE_METRICS"] == row).all() and (r.loc["BUS"] == 1.0).all()


def test_known_shift_yaw_and_size_errors():
    ctr, dims, yaw = _boxes(10, seed=1)
    cfg = M.DetectionCfg(categories=("BUS",))
    gts = _frame(ctr, dims, yaw, "num_interior_pts", np.full(10, 3))
    # 1.5 m shift in x: true positives at the 2 m and 4 m thresholds only (strict `>`), ATE 1.5
    d, _, m, _ = M.evaluate(_frame(ctr + np.array([1.5, 0, 0]), dims, yaw, "score", np.linspace(0.9, 0.1, 10)), gts, cfg)
    assert d[0.5].sum() == 0 and d[1.0].sum() == 0 and d[2.0].sum() == 10 and d[4.0].sum() == 10
    assert m.loc["BUS", "ATE"] == 1.5 and m.loc["BUS", "AP"] == 0.5
    # yaw off by 0.25 rad, and by pi + 0.25 (wrap_angles folds at pi: error pi - 0.25... of the absolute difference)
    d, _, m, _ = M.evaluate(_frame(ctr, dims, yaw + 0.25, "score", np.linspace(0.9, 0.1, 10)), gts, cfg)
    assert np.allclose(d["AOE"], 0.25, atol=1e-9) and m.loc["BUS", "AOE"] == 0.25
    got = M.wrap_angles(np.array([0.25, -0.25, math.pi + 0.25, 2 * math.pi + 0.1]))
    assert np.allclose(got, [0.25, 0.25, math.pi - 0.25, math.pi - 0.1])
    # every size 2x too long: IoU of origin-aligned boxes = 1/2, ASE 0.5
    d, _, m, _ = M.evaluate(_frame(ctr, dims * np.array([2.0, 1, 1]), yaw, "score", np.linspace(0.9, 0.1, 10)), gts, cfg)
    assert np.allclose(d["ASE"], 0.5) and m.loc["BUS", "ASE"] == 0.5
    # CDS = mAP * mean(1 - ATE/2, 1 - ASE/1, 1 - AOE/pi)
    assert m.loc["BUS", "CDS"] == round(1.0 * np.mean([1.0, 0.5, 1.0]), 3)


def test_detection_cap_range_filter_and_unobserved_ground_truth():
    ctr, dims, yaw = _boxes(150, seed=2)
    cfg = M.DetectionCfg(categories=("BUS",))
    far = ctr.copy(); far[:5, 0] = 400.0                         # beyond eval_range_m
    pts = np.full(150, 2.0); pts[5:10] = 0                       # never seen by the lidar
    d, g, _, _ = M.evaluate(_frame(far, dims, yaw, "score", np.linspace(0.99, 0.01, 150)), _frame(far, dims, yaw, "num_interior_pts", pts), cfg)
    d = d.sort_values("score", ascending=False)
    assert d["is_evaluated"].sum() == 100                        # the 100 best-scoring in-range detections
    assert not d["is_evaluated"].iloc[:5].any() and d["is_evaluated"].iloc[5:105].all() and not d["is_evaluated"].iloc[105:].any()
    assert g["is_evaluated"].sum() == 140


def test_roi_pruning_needs_map_data():
    ctr, dims, yaw = _boxes(3)
    with pytest.raises(ValueError):
        M.evaluate(_frame(ctr, dims, yaw, "score", [0.5, 0.4, 0.3]), _frame(ctr, dims, yaw, "num_interior_pts", [1, 1, 1]),
                   M.DetectionCfg(categories=("BUS",), eval_only_roi_instances=True))
    seen = []
    def roi(cuboids, log_id, ts):
        seen.append((log_id, ts))
        return np.arange(len(cuboids)) != 0                      # drop the first object of every table
    d, g, _, _ = M.evaluate(_frame(ctr, dims, yaw, "score", [0.5, 0.4, 0.3]), _frame(ctr, dims, yaw, "num_interior_pts", [1, 1, 1]),
                            M.DetectionCfg(categories=("BUS",), eval_only_roi_instances=True, roi_mask_fn=roi))
    assert seen and d["is_evaluated"].sum() == 2 and g["is_evaluated"].sum() == 2
