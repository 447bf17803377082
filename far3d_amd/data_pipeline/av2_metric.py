"""AV2 3D-detection metric of the reference's evaluation step: AP over centre-distance thresholds,
ATE / ASE / AOE of the true positives, CDS -- the numbers `Argoverse2Dataset.evaluate` prints (datasets/argoverse2_dataset.py:224-265).

Host-side numpy / pandas, like the reference (there is no device work here).  Restated from the reference's own files
  * datasets/av2_utils.py:35-234   DetectionCfg, accumulate, assign, distance, compute_affinity_matrix, evaluated-object masks
  * datasets/av2_eval_util.py:60-156   evaluate (grouping by (log_id, timestamp_ns, category), table assembly, rounding)
  * datasets/summarize_metrics_av2.py:44-191   summarize_metrics, compute_average_precision, interpolate_precision
and, for the helpers those files import from the third-party package `av2==0.2.1` (absent from the reference tree and from this
image), from that package's published semantics: `iou_3d_axis_aligned`, `wrap_angles`, `mat_to_xyz(quat_to_mat(q))[..., 2]`, the
constants of `av2.evaluation.detection.constants` and `av2.utils.constants.EPS`.  PARITY for those helpers is UNPINNED (they cannot
be executed here); the reference-owned logic is pinned by tests/golden/far3d_av2_metric.npz (tools/gen_golden_data.py runs the
reference's accumulate / summarize_metrics with these helpers supplied as stand-ins).

Not built: ROI pruning (`eval_only_roi_instances`): it needs the AV2 map rasters and city poses of the dataset
(av2_utils.py:236-252); `evaluate` raises if it is requested without a `roi_mask_fn`.
"""
import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Tuple

import numpy as np

from .results import AV2_CLASSES

# av2.evaluation.detection.constants / av2.utils.constants (av2 0.2.1)
MAX_SCALE_ERROR = 1.0
MAX_YAW_RAD_ERROR = math.pi
MIN_AP = 0.0
MIN_CDS = 0.0
MAX_NORMALIZED_ASE = 1.0
NUM_DECIMALS = 3
EPS = 1e-10
ORDERED_CUBOID_COL_NAMES = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz")
DTS_COLUMN_NAMES = ORDERED_CUBOID_COL_NAMES + ("score",)
GTS_COLUMN_NAMES = ORDERED_CUBOID_COL_NAMES + ("num_interior_pts",)
UUID_COLUMN_NAMES = ("log_id", "timestamp_ns", "category")
TP_ERROR_COLUMNS = ("ATE", "ASE", "AOE")
METRIC_NAMES = ("AP", "ATE", "ASE", "AOE", "CDS", "RECALL")


@dataclass(frozen=True)
class DetectionCfg:
    """av2_utils.py:35-69 (the reference's variant: four thresholds, an evaluation RANGE instead of av2's max range)."""
    affinity_thresholds_m: Tuple[float, ...] = (0.5, 1.0, 2.0, 4.0)
    categories: Tuple[str, ...] = tuple(AV2_CLASSES)
    eval_only_roi_instances: bool = False
    max_num_dts_per_category: int = 100
    eval_range_m: Tuple[float, ...] = (0.0, 150.0)
    num_recall_samples: int = 100
    tp_threshold_m: float = 2.0
    roi_mask_fn: Optional[Callable] = field(default=None, compare=False)    # (cuboids (n,>=10), log_id, timestamp_ns) -> bool (n,)

    @property
    def metrics_defaults(self):
        return (MIN_AP, self.tp_threshold_m, MAX_NORMALIZED_ASE, MAX_YAW_RAD_ERROR, MIN_CDS, MIN_AP)

    @property
    def tp_normalization_terms(self):
        return (self.tp_threshold_m, MAX_SCALE_ERROR, MAX_YAW_RAD_ERROR)


# ------------------------------------------------------------------------------------------ av2 package helpers (published semantics)
def iou_3d_axis_aligned(src_dims_m, target_dims_m):
    """av2.geometry.iou.iou_3d_axis_aligned: both cuboids centred at the origin and aligned to +x; (n,3) dims each."""
    inter = np.minimum(src_dims_m, target_dims_m).prod(axis=1)
    union = src_dims_m.prod(axis=1) + target_dims_m.prod(axis=1) - inter
    return inter / union


def quat_to_yaw(quat_wxyz):
    """mat_to_xyz(quat_to_mat(q))[..., 2] (av2.geometry.geometry): the z angle of the extrinsic-xyz Euler decomposition,
    atan2(R[1,0], R[0,0]) of the rotation matrix of the (normalised) scalar-first quaternion."""
    q = np.asarray(quat_wxyz, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r10 = 2.0 * (x * y + w * z)
    r00 = 1.0 - 2.0 * (y * y + z * z)
    return np.arctan2(r10, r00)


def wrap_angles(angles, period=math.pi):
    """av2.geometry.geometry.wrap_angles: map angles from (-inf, inf) to [0, period): |angle| folded at multiples of the period."""
    angles = np.abs(np.asarray(angles, dtype=np.float64))
    divs, mods = np.divmod(angles, period)
    out = angles.copy()
    comp = divs != 0
    out[comp] = period - mods[comp]
    return out


# ------------------------------------------------------------------------------------------ reference-owned logic
def compute_affinity_matrix(dts_xyz, gts_xyz):
    """av2_utils.py:182-191: negative Euclidean distance of the 3D centres (AffinityType.CENTER)."""
    d = dts_xyz[:, None, :].astype(np.float64) - gts_xyz[None, :, :].astype(np.float64)
    return -np.sqrt((d * d).sum(-1))


def distance(dts, gts, metric):
    """av2_utils.py:163-180.  metric: 'translation' (n,3 centres), 'scale' (n,3 dims), 'orientation' (n,4 wxyz quaternions)."""
    if metric == "translation":
        return np.linalg.norm(dts - gts, axis=1)
    if metric == "scale":
        return 1.0 - iou_3d_axis_aligned(dts, gts)
    if metric == "orientation":
        return wrap_angles(quat_to_yaw(dts) - quat_to_yaw(gts))
    raise NotImplementedError(metric)


def compute_evaluated_dts_mask(xyz_m_ego, cfg):
    """av2_utils.py:193-211: inside the evaluation range, at most max_num_dts_per_category of them (input is score-sorted)."""
    if len(xyz_m_ego) == 0:
        return np.zeros((0,), dtype=bool)
    norm = np.linalg.norm(xyz_m_ego, axis=1)
    ok = np.logical_and(norm > cfg.eval_range_m[0], norm < cfg.eval_range_m[1])
    over = np.where(np.cumsum(ok) > cfg.max_num_dts_per_category)[0]
    if len(over) > 0:
        ok[over[0]:] = False
    return ok


def compute_evaluated_gts_mask(xyz_m_ego, num_interior_pts, cfg):
    """av2_utils.py:213-228: inside the evaluation range and observed by at least one lidar point."""
    if len(xyz_m_ego) == 0:
        return np.zeros((0,), dtype=bool)
    norm = np.linalg.norm(xyz_m_ego, axis=1)
    return np.logical_and(np.logical_and(norm > cfg.eval_range_m[0], norm < cfg.eval_range_m[1]), num_interior_pts > 0)


def assign(dts, gts, cfg):
    """av2_utils.py:120-161: every detection goes to its nearest ground truth; the FIRST (= best-scoring) detection of each
    ground truth is a true positive at every threshold its distance clears; TP errors at the tp threshold."""
    aff = compute_affinity_matrix(dts[:, :3], gts[:, :3])
    idx_gts_all = aff.argmax(axis=1)
    affinities = aff[np.arange(len(dts)), idx_gts_all]
    idx_gts, idx_dts = np.unique(idx_gts_all, return_index=True)
    T, E = len(cfg.affinity_thresholds_m), 3
    dts_metrics = np.zeros((len(dts), T + E))
    dts_metrics[:, T:] = cfg.metrics_defaults[1:4]
    gts_metrics = np.zeros((len(gts), T + E))
    gts_metrics[:, T:] = cfg.metrics_defaults[1:4]
    for i, thr in enumerate(cfg.affinity_thresholds_m):
        is_tp = affinities[idx_dts] > -thr
        dts_metrics[idx_dts[is_tp], i] = True
        gts_metrics[idx_gts, i] = True
        if thr != cfg.tp_threshold_m or not np.any(is_tp):
            continue
        td, tg = dts[idx_dts[is_tp]], gts[idx_gts[is_tp]]
        dts_metrics[idx_dts[is_tp], T:] = np.stack((distance(td[:, :3], tg[:, :3], "translation"),
                                                    distance(td[:, 3:6], tg[:, 3:6], "scale"),
                                                    distance(td[:, 6:10], tg[:, 6:10], "orientation")), axis=-1)
    return dts_metrics, gts_metrics


def accumulate(dts, gts, cfg, roi_masks=None):
    """av2_utils.py:71-118.  dts (N,11) cuboid+score, gts (M,11) cuboid+num_interior_pts of ONE (sweep, category).
    roi_masks: optional (dts mask in score order applied after sorting, gts mask) from cfg.roi_mask_fn.
    Returns (N,T+4), (M,T+4): TP flags per threshold, ATE/ASE/AOE, is_evaluated -- detections in their INPUT order."""
    N, M = len(dts), len(gts)
    T, E = len(cfg.affinity_thresholds_m), 3
    perm = np.argsort(-dts[:, -1], kind="stable") if N else np.zeros((0,), dtype=np.int64)
    dts = dts[perm]
    ev_d, ev_g = np.ones(N, dtype=bool), np.ones(M, dtype=bool)
    if roi_masks is not None:
        ev_d &= roi_masks[0][perm]
        ev_g &= roi_masks[1]
    ev_d &= compute_evaluated_dts_mask(dts[:, :3], cfg)
    ev_g &= compute_evaluated_gts_mask(gts[:, :3], gts[:, -1], cfg)
    dts_aug, gts_aug = np.zeros((N, T + E + 1)), np.zeros((M, T + E + 1))
    dts_aug[ev_d, -1] = True
    gts_aug[ev_g, -1] = True
    if ev_d.sum() > 0 and ev_g.sum() > 0:
        da, ga = assign(dts[ev_d], gts[ev_g], cfg)
        dts_aug[ev_d, :-1] = da
        gts_aug[ev_g, :-1] = ga
    inv = np.empty(N, dtype=np.int64)
    inv[perm] = np.arange(N)
    return dts_aug[inv], gts_aug


def interpolate_precision(precision):
    """summarize_metrics_av2.py:163-191 (VOC 'all points'): p_interp(r) = max over r' >= r of p(r')."""
    return np.maximum.accumulate(precision[::-1])[::-1]


def compute_average_precision(tps, recall_interpolated, num_gts):
    """summarize_metrics_av2.py:131-161 -> (average precision, interpolated precision, recall of the ranked list)."""
    cum_tps = np.cumsum(tps)
    cum_fps = np.cumsum(~tps)
    cum_fns = num_gts - cum_tps
    precision = interpolate_precision(cum_tps / (cum_tps + cum_fps + EPS))
    recall = cum_tps / (cum_tps + cum_fns)
    pi = np.interp(recall_interpolated, recall, precision, right=0)
    return float(np.mean(pi)), pi, float(cum_tps[-1] / num_gts)


def summarize_metrics(dts, gts, cfg):
    """summarize_metrics_av2.py:44-129.  dts / gts: pandas tables holding `category`, `score` (dts), one column per affinity
    threshold, ATE/ASE/AOE and `is_evaluated`.  Returns (summary (C, 6): AP ATE ASE AOE CDS RECALL, recall per threshold)."""
    import pandas as pd
    recall_interpolated = np.linspace(0, 1, cfg.num_recall_samples, endpoint=True)
    summary = pd.DataFrame({s: cfg.metrics_defaults[i] for i, s in enumerate(METRIC_NAMES)}, index=list(cfg.categories))
    aps = pd.DataFrame({t: 0.0 for t in cfg.affinity_thresholds_m}, index=list(cfg.categories))
    recs = pd.DataFrame({t: 0.0 for t in cfg.affinity_thresholds_m}, index=list(cfg.categories))
    for cat in cfg.categories:
        valid = np.logical_and(dts["category"] == cat, dts["is_evaluated"].astype(bool))
        cd = dts.loc[valid].sort_values(by="score", ascending=False).reset_index(drop=True)
        num_gts = gts.loc[gts["category"] == cat, "is_evaluated"].sum()
        if num_gts == 0:
            continue
        for thr in cfg.affinity_thresholds_m:
            tps = cd[thr].astype(bool).to_numpy()
            if len(tps) == 0:
                continue
            ap, _, rec = compute_average_precision(tps, recall_interpolated, num_gts)
            aps.loc[cat, thr] = ap
            recs.loc[cat, thr] = rec
        m_ap, m_rec = aps.loc[cat].to_numpy().mean(), recs.loc[cat].to_numpy().mean()
        mid = cfg.affinity_thresholds_m[len(cfg.affinity_thresholds_m) // 2]
        is_tp = cd[mid].to_numpy().astype(bool)
        tp_errors = np.array(cfg.tp_normalization_terms)
        if np.any(is_tp):
            tp_errors = cd.loc[is_tp, list(TP_ERROR_COLUMNS)].to_numpy().mean(axis=0)
        tp_scores = 1 - np.divide(tp_errors, cfg.tp_normalization_terms)
        summary.loc[cat] = np.array([m_ap, *tp_errors, m_ap * np.mean(tp_scores), m_rec])
    return summary, recs


def evaluate(dts, gts, cfg):
    """av2_eval_util.py:60-156 (sequential; the reference fans `accumulate` out over a process pool).
    dts: table with UUID columns + the 10 cuboid columns + `score`; gts: the same + `num_interior_pts`.
    Returns (dts, gts) with the metric columns filled, the (C+1, 6) metrics table incl. the AVERAGE_METRICS row, recall table."""
    if cfg.eval_only_roi_instances and cfg.roi_mask_fn is None:
        raise ValueError("ROI pruning needs the AV2 map rasters and city poses (av2_utils.py:236-252): pass DetectionCfg(roi_mask_fn=...) "
                         "or eval_only_roi_instances=False")
    dts = dts.sort_values(list(UUID_COLUMN_NAMES)).reset_index(drop=True)
    gts = gts.sort_values(list(UUID_COLUMN_NAMES)).reset_index(drop=True)
    dn = dts[list(DTS_COLUMN_NAMES)].to_numpy(dtype=np.float64)
    gn = gts[list(GTS_COLUMN_NAMES)].to_numpy(dtype=np.float64)
    key = lambda df: [":".join(map(str, x)) for x in df[list(UUID_COLUMN_NAMES)].to_numpy().tolist()]

    def groups(keys):
        out, start = {}, 0
        for i in range(1, len(keys) + 1):
            if i == len(keys) or keys[i] != keys[start]:
                out[keys[start]] = (start, i)
                start = i
        return out
    gd, gg = groups(key(dts)), groups(key(gts))
    T = len(cfg.affinity_thresholds_m)
    dm, gm = np.zeros((len(dts), T + 4)), np.zeros((len(gts), T + 4))
    for uuid in sorted(gd.keys() | gg.keys()):
        ds, de = gd.get(uuid, (0, 0))
        gs, ge = gg.get(uuid, (0, 0))
        sd, sg = dn[ds:de], gn[gs:ge]
        roi = None
        if cfg.eval_only_roi_instances:
            log_id, ts, _ = uuid.split(":")
            roi = (np.asarray(cfg.roi_mask_fn(sd, log_id, int(ts)), dtype=bool), np.asarray(cfg.roi_mask_fn(sg, log_id, int(ts)), dtype=bool))
        a, b = accumulate(sd, sg, cfg, roi)
        dm[ds:de], gm[gs:ge] = a, b
    cols = list(cfg.affinity_thresholds_m) + list(TP_ERROR_COLUMNS) + ["is_evaluated"]
    for j, c in enumerate(cols):
        dts[c] = dm[:, j]
        gts[c] = gm[:, j]
    metrics, recall = summarize_metrics(dts, gts, cfg)
    metrics.loc["AVERAGE_METRICS"] = metrics.mean()
    recall.loc["AVERAGE_METRICS"] = recall.mean()
    return dts, gts, metrics.round(NUM_DECIMALS), recall.round(NUM_DECIMALS)
