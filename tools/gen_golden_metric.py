"""Fixture for the AV2 detection metric -- BUILD CONTAINER ONLY (needs /root/reference).

  python tools/gen_golden_metric.py     # rewrites tests/golden/far3d_av2_metric.npz

The expected values come from the REFERENCE's own functions, loaded where they lie: `accumulate` (datasets/av2_utils.py:71-118,
with assign / distance / the evaluated-object masks under it) and `summarize_metrics` (datasets/summarize_metrics_av2.py:44-129).
Their third-party helpers from `av2==0.2.1` (not in the reference tree, not in this image) are supplied as stand-ins written from
that package's published semantics: quat_to_mat / mat_to_xyz through scipy's Rotation (what av2 itself calls), wrap_angles,
iou_3d_axis_aligned, the detection constants and enums.  The per-sweep grouping and table assembly of `evaluate`
(datasets/av2_eval_util.py:88-147) is followed here step by step instead of called: it fans out over a spawn()ed process pool
that would re-import av2.  Fixtures hold data only.
"""
import enum
import importlib
import math
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd.data_pipeline import av2_metric  # noqa: E402
from far3d_amd.data_pipeline.results import AV2_CLASSES  # noqa: E402
from oracle import refload  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CATS = ("REGULAR_VEHICLE", "PEDESTRIAN", "BUS", "BOLLARD")


def install_av2_standins():
    from scipy.spatial.transform import Rotation
    refload.install_data_stubs(AV2_CLASSES)
    c = importlib.import_module("av2.evaluation.detection.constants")
    c.MAX_SCALE_ERROR, c.MAX_YAW_RAD_ERROR, c.MIN_AP, c.MIN_CDS = 1.0, math.pi, 0.0, 0.0
    c.MAX_NORMALIZED_ASE, c.NUM_DECIMALS = 1.0, 3
    c.AffinityType = enum.Enum("AffinityType", {"CENTER": "CENTER"}, type=str)
    c.DistanceType = enum.Enum("DistanceType", {k: k for k in ("TRANSLATION", "SCALE", "ORIENTATION")}, type=str)
    c.FilterMetricType = enum.Enum("FilterMetricType", {"EUCLIDEAN": "EUCLIDEAN"}, type=str)
    c.InterpType = enum.Enum("InterpType", {"ALL": "ALL"}, type=str)
    importlib.import_module("av2.utils.constants").EPS = 1e-10
    g = importlib.import_module("av2.geometry.geometry")
    g.quat_to_mat = lambda q: Rotation.from_quat(np.asarray(q)[..., [1, 2, 3, 0]]).as_matrix()      # wxyz -> scipy's xyzw
    g.mat_to_xyz = lambda m: Rotation.from_matrix(m).as_euler("xyz", degrees=False)
    g.wrap_angles = av2_metric.wrap_angles
    importlib.import_module("av2.geometry.iou").iou_3d_axis_aligned = av2_metric.iou_3d_axis_aligned
    importlib.import_module("av2.structures.cuboid").ORDERED_CUBOID_COL_NAMES = av2_metric.ORDERED_CUBOID_COL_NAMES


def quat_z(yaw):
    return np.stack([np.cos(yaw / 2), np.zeros_like(yaw), np.zeros_like(yaw), np.sin(yaw / 2)], axis=-1)


def scenario(rng):
    """Ground truth and detections of 2 logs x 3 sweeps x 4 categories: matched boxes with centre / size / yaw noise, misses,
    false positives, objects beyond the evaluation range, unobserved ground truth, one group with > 100 detections."""
    drows, grows = [], []
    for li in range(2):
        for ti in range(3):
            for cat in CATS:
                m = int(rng.integers(0, 7))
                ctr = rng.uniform(-120, 120, size=(m, 3)) * np.array([1, 1, 0.02])
                if m and rng.random() < 0.4:
                    ctr[0, :2] = (170.0, 30.0)                                     # outside eval_range_m
                dims = rng.uniform(0.4, 6.0, size=(m, 3))
                yaw = rng.uniform(-math.pi, math.pi, size=m)
                pts = rng.integers(0, 40, size=m) * (rng.random(m) > 0.15)          # some ground truth has no lidar points
                for j in range(m):
                    grows.append(("log%d" % li, 1000 + ti, cat, *ctr[j], *dims[j], *quat_z(yaw[j:j + 1])[0], float(pts[j])))
                keep = rng.random(m) > 0.25
                nd = int(keep.sum())
                dc = ctr[keep] + rng.normal(0, 0.6, size=(nd, 3)) * np.array([1, 1, 0.1])
                dd = dims[keep] * rng.uniform(0.7, 1.3, size=(nd, 3))
                dy = yaw[keep] + rng.normal(0, 0.3, size=nd) + math.pi * (rng.random(nd) < 0.1)
                nf = int(rng.integers(0, 5)) + (120 if (li, ti, cat) == (1, 2, "PEDESTRIAN") else 0)
                dc = np.concatenate([dc, rng.uniform(-140, 140, size=(nf, 3)) * np.array([1, 1, 0.02])])
                dd = np.concatenate([dd, rng.uniform(0.4, 6.0, size=(nf, 3))])
                dy = np.concatenate([dy, rng.uniform(-math.pi, math.pi, size=nf)])
                sc = rng.permutation(nd + nf) / (nd + nf + 1.0) * 0.9 + rng.uniform(0.01, 0.05)   # distinct scores
                for j in range(nd + nf):
                    drows.append(("log%d" % li, 1000 + ti, cat, *dc[j], *dd[j], *quat_z(dy[j:j + 1])[0], float(sc[j])))
    cols = list(av2_metric.UUID_COLUMN_NAMES) + list(av2_metric.ORDERED_CUBOID_COL_NAMES)
    dts = pd.DataFrame(drows, columns=cols + ["score"]).sample(frac=1.0, random_state=1).reset_index(drop=True)   # shuffled input order
    gts = pd.DataFrame(grows, columns=cols + ["num_interior_pts"]).sample(frac=1.0, random_state=2).reset_index(drop=True)
    return dts, gts


def reference_evaluate(dts, gts, ref_utils, ref_sum):
    """av2_eval_util.py:88-147 step by step, with the reference's accumulate / summarize_metrics."""
    cfg = ref_utils.DetectionCfg(categories=CATS, eval_only_roi_instances=False)
    U = list(av2_metric.UUID_COLUMN_NAMES)
    dts, gts = dts.sort_values(U), gts.sort_values(U)
    dn, gn = dts[list(av2_metric.DTS_COLUMN_NAMES)].to_numpy(), gts[list(av2_metric.GTS_COLUMN_NAMES)].to_numpy()

    def groupby(names, values):      # av2.evaluation.detection.utils.groupby: name -> rows, names sorted
        out = {}
        for i, nme in enumerate(names):
            out.setdefault(nme, []).append(i)
        return {k: values[v] for k, v in out.items()}
    u2d = groupby([":".join(map(str, x)) for x in dts[U].to_numpy().tolist()], dn)
    u2g = groupby([":".join(map(str, x)) for x in gts[U].to_numpy().tolist()], gn)
    outs = []
    for uuid in sorted(u2d.keys() | u2g.keys()):
        sd, sg = u2d.get(uuid, np.zeros((0, 11))), u2g.get(uuid, np.zeros((0, 11)))
        outs.append(ref_utils.accumulate(sd, sg, cfg, None, None))
    # rows of one uuid are contiguous after the sort, and uuids are visited in the sorted order
    only_d = [u for u in sorted(u2d.keys() | u2g.keys())]
    dm = np.concatenate([o[0] for o, u in zip(outs, only_d)])
    gm = np.concatenate([o[1] for o, u in zip(outs, only_d)])
    cols = list(cfg.affinity_thresholds_m) + ["ATE", "ASE", "AOE", "is_evaluated"]
    dts, gts = dts.copy(), gts.copy()
    dts.loc[:, cols] = dm
    gts.loc[:, cols] = gm
    metrics, recall = ref_sum.summarize_metrics(dts, gts, cfg)
    metrics.loc["AVERAGE_METRICS"] = metrics.mean()
    recall.loc["AVERAGE_METRICS"] = recall.mean()
    return dts, gts, metrics.round(3), recall.round(3), cols


def main():
    install_av2_standins()
    ref_utils = refload.ref("datasets.av2_utils")
    ref_sum = refload.ref("datasets.summarize_metrics_av2")
    rng = np.random.default_rng(7)
    dts, gts = scenario(rng)
    rd, rg, metrics, recall, cols = reference_evaluate(dts, gts, ref_utils, ref_sum)
    # our restatement must agree before the fixture is written
    md, mg, mm, mr = av2_metric.evaluate(dts, gts, av2_metric.DetectionCfg(categories=CATS))
    assert np.allclose(md[cols].to_numpy(dtype=float), rd[cols].to_numpy(dtype=float), atol=1e-9), "accumulate differs from the reference"
    assert np.allclose(mg[cols].to_numpy(dtype=float), rg[cols].to_numpy(dtype=float), atol=1e-9)
    assert np.allclose(mm.to_numpy(dtype=float), metrics.to_numpy(dtype=float), atol=1e-9), (mm, metrics)
    assert np.allclose(mr.to_numpy(dtype=float), recall.to_numpy(dtype=float), atol=1e-9)
    U = list(av2_metric.UUID_COLUMN_NAMES)
    gold = dict(
        categories=np.array(CATS),
        dts_uuid=dts[U].to_numpy().astype(str), dts_num=dts[list(av2_metric.DTS_COLUMN_NAMES)].to_numpy(dtype=np.float64),
        gts_uuid=gts[U].to_numpy().astype(str), gts_num=gts[list(av2_metric.GTS_COLUMN_NAMES)].to_numpy(dtype=np.float64),
        # expected, in the uuid-sorted row order evaluate() returns
        dts_sorted_uuid=rd[U].to_numpy().astype(str), dts_sorted_score=rd["score"].to_numpy(dtype=np.float64),
        dts_metrics=rd[cols].to_numpy(dtype=np.float64), gts_metrics=rg[cols].to_numpy(dtype=np.float64),
        gts_sorted_uuid=rg[U].to_numpy().astype(str),
        metrics=metrics.to_numpy(dtype=np.float64), metrics_index=np.array(list(metrics.index)), metrics_columns=np.array(list(metrics.columns)),
        recall=recall.to_numpy(dtype=np.float64))
    path = os.path.join(GOLD, "far3d_av2_metric.npz")
    np.savez_compressed(path, **gold)
    print(metrics)
    print("[golden] wrote %s: %d detections, %d ground-truth boxes, %d evaluated detections, %d true positives at 2 m" %
          (path, len(dts), len(gts), int(rd["is_evaluated"].sum()), int(rd[2.0].sum())))


if __name__ == "__main__":
    if not refload.available():
        sys.exit("reference checkout not found: fixtures can only be regenerated in the build container")
    main()
