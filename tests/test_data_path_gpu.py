"""GPU: the whole data path around the hot path, chained (ref datasets/argoverse2_dataset.py:224-341,
datasets/argoverse2_dataset_t.py:143-240, tools/test.py):

    raw uint8 camera images + calibration                     (synthetic, seeded)
      -> StreamingIndex: the order frames reach one rank, scene tokens, ego poses        (f4)
      -> ImagePreprocessor: resize / crop / normalise / pad on the device, ida -> lidar2img   (f1)
      -> Far3D detector built from the registry config, streaming memory across frames    (a1-a12, b)
      -> format_results: AV2 detection rows                                               (f3)
      -> av2_metric.evaluate against ground truth

Ground truth = what the CPU oracle detects on the SAME pre-processed frames (fed in the same order, with its own streaming
memory), written in the AV2 annotation layout.  If every link of the chain is right the metric must say so: AP ~ 1, translation /
scale / orientation errors ~ 0, CDS ~ 1 for every category that occurs.  A wrong frame order (memory reset at the wrong time), a
calibration that misses the augmentation homography, a quaternion or size-column mix-up in the result rows or a broken matching
rule each show up as lost AP or non-zero errors.  This is the harness the "mAP within 0.1" check needs the day a real checkpoint
and the AV2 slice are supplied; here it pins the plumbing."""
import numpy as np
import pytest
import torch

from far3d_amd import config, plugin, synth, weights
from far3d_amd import data_pipeline as dp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NCAM, RAW_HW, NET_HW = 2, (150, 210), (64, 96)
CFG = dict(num_cams=NCAM, num_query=60, num_propagated=16, memory_len=64, topk_proposals=16)


def _infos():
    """Two scenes (3 + 2 sweeps), with city poses that move; raw images are drawn per (sweep, camera)."""
    infos = []
    for k in range(5):
        scene = "log-a" if k < 3 else "log-b"
        th = 0.05 * k
        R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
        infos.append(dict(scene_id=scene, lidar_timestamp_ns=315969904359876000 + 100000000 * k,
                          city_SE3_ego=(R, np.array([2.0 * k, -0.5 * k, 0.02 * k]))))
    return infos


def _raw_frame(k):
    rs = np.random.RandomState(100 + k)
    imgs = [rs.randint(0, 256, size=RAW_HW + (3,), dtype=np.uint8) for _ in range(NCAM)]
    intr, extr, _ = synth.ring_cameras(NCAM, RAW_HW)          # pinhole cameras of the RAW resolution
    return dict(img=imgs, intrinsics=[intr[n].double().numpy().copy() for n in range(NCAM)],
                extrinsics=[extr[n].double().numpy().copy() for n in range(NCAM)])


def _to_batch(res, frame):
    f32 = lambda a: torch.as_tensor(np.stack(a), dtype=torch.float32)[None]
    data = dict(img=res["img"][None].float(), lidar2img=f32(res["lidar2img"]), intrinsics=f32(res["intrinsics"]),
                extrinsics=f32(res["extrinsics"]), ego_pose=torch.as_tensor(frame["ego_pose"], dtype=torch.float32)[None],
                ego_pose_inv=torch.as_tensor(frame["ego_pose_inv"], dtype=torch.float32)[None],
                timestamp=torch.tensor([float(frame["timestamp"])], dtype=torch.float64))
    metas = [dict(pad_shape=res["pad_shape"], scene_token=frame["scene_token"])]
    return data, metas


def test_raw_images_to_av2_metric_through_the_detector(hip_lib):
    import pandas as pd
    from oracle import far3d_oracle
    from far3d_amd.data_pipeline import av2_metric as M
    from far3d_amd.data_pipeline import results as R
    infos = _infos()
    index = dp.streaming.StreamingIndex(infos, interval_test=False, num_replicas=1, rank=0)
    pre = dp.preprocess.ImagePreprocessor(dict(dp.preprocess.DEFAULT_AUG, final_dim=NET_HW), device=DEV, rng=np.random.RandomState(3))
    spec = weights.detector_spec("V-99-eSE", num_query=CFG["num_query"], num_propagated=CFG["num_propagated"])
    sd = weights.init_state_dict(spec, seed=7)
    det = plugin.build_detector(config.default_model_cfg(**CFG))
    det.load_state_dict(sd)
    det.prepare(DEV, precision="fp32")
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(**CFG))

    outputs, used, gt_rows, scenes_seen = [], [], [], []
    for i, frame in index:
        res = pre(_raw_frame(i))
        assert tuple(res["img"].shape) == (NCAM, 3) + NET_HW and res["pad_shape"][0][:2] == NET_HW
        data, metas = _to_batch(res, frame)
        scenes_seen.append((frame["scene_token"], frame["prev_exists"]))
        out = det(return_loss=False, rescale=True, img_metas=metas, **{k: v.to(DEV) for k, v in data.items()})[0]
        outputs.append({"pts_bbox": {k: v.cpu() for k, v in out["pts_bbox"].items()}})
        used.append(index.infos[i])
        with torch.no_grad():
            want = orc.simple_test({k: v.cpu() for k, v in data.items()}, metas)["result"]
        # ground truth: ALL the oracle's detections of the sweep (capped like the metric caps detections: 100 per category and
        # sweep).  On the CPU, oracle-vs-oracle gives AP = 1.000 for every category; the device may turn a near-tie at the top-k
        # cut the other way (one detection in 300), which the 0.97 bar below absorbs.
        cub = R.box_to_av2(want["boxes_3d"]).numpy()
        lab = want["labels_3d"].numpy()
        for c in np.unique(lab):
            rows = cub[lab == c][:100]
            for r in rows:
                gt_rows.append(dict(zip(R.LABEL_ATTR, r), log_id=frame["scene_token"], timestamp_ns=int(index.infos[i]["lidar_timestamp_ns"]),
                                    category=R.AV2_CLASSES[int(c)], num_interior_pts=5))
    # f4: the frames of a scene arrive in order and the memory is reset exactly at the scene boundary
    assert scenes_seen == [("log-a", False), ("log-a", True), ("log-a", True), ("log-b", False), ("log-b", True)]

    dts = R.format_results(outputs, used).reset_index()
    gts = pd.DataFrame(gt_rows)
    assert len(gts) > 50 and set(dts.columns) >= set(R.LABEL_ATTR) | {"score", "log_id", "timestamp_ns", "category"}
    cfg = M.DetectionCfg(eval_only_roi_instances=False)
    dts_e, gts_e, metrics, recall = M.evaluate(dts, gts, cfg)
    present = sorted(set(gts["category"]))
    got = metrics.loc[present]
    print("\n" + got.to_string())
    # every ground-truth box inside the evaluation range is found at every centre-distance threshold, with negligible errors
    evaluated = gts_e[gts_e["is_evaluated"].astype(bool)]
    assert len(evaluated) > 30
    cats = sorted(set(evaluated["category"]))
    # (a single near-tie flipped at the device's top-k cut costs a small category several points of AP: the per-category bar is
    # therefore 0.9, the bar on the mean over categories and on the fraction of matched ground-truth boxes 0.98 / 0.99)
    assert (metrics.loc[cats, "AP"] > 0.9).all() and metrics.loc[cats, "AP"].mean() > 0.98, metrics.loc[cats]
    assert (metrics.loc[cats, "ATE"] < 2e-2).all() and (metrics.loc[cats, "ASE"] < 2e-2).all() and (metrics.loc[cats, "AOE"] < 2e-2).all()
    assert metrics.loc[cats, "CDS"].mean() > 0.97
    assert evaluated[0.5].mean() > 0.99, "ground-truth boxes matched at the 0.5 m threshold: %.4f" % evaluated[0.5].mean()
    assert (recall.loc[cats] > 0.9).all().all()
    # and the metric is not vacuous: shifting the detections by 5 m loses them at every threshold
    moved = dts.copy()
    moved["tx_m"] += 5.0
    _, _, m2, _ = M.evaluate(moved, gts, cfg)
    assert (m2.loc[cats, "AP"] < 0.2).all()
