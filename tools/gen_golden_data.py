"""Fixtures for the data contract around the hot path -- BUILD CONTAINER ONLY.

  python tools/gen_golden_data.py      # rewrites tests/golden/far3d_data_contract.npz

Every expected value is produced by the REFERENCE's own classes, loaded where they lie (oracle/refload.py):
  f1  AV2ResizeCropFlipRotImageV2 -> NormalizeMultiviewImage -> AV2PadMultiViewImage on seeded uint8 images (incl. a portrait
      camera and a flipped case); Pillow does the pixels, mmcv.imnormalize / impad are restated stand-ins (refload);
  f3  Argoverse2Dataset.box_to_av2 / format_results, av2_utils.yaw_to_quat on seeded boxes;
  f4  Argoverse2DatasetT group flags / interval_test order / get_data_info fields, samplers.DistributedSampler shards.
av2's SE3 (rotation, translation, inverse, compose = right-multiply) is a 10-line stand-in written from av2 0.2.1's published
semantics.  Fixtures hold data only."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd.data_pipeline.results import AV2_CLASSES  # noqa: E402
from oracle import refload  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class SE3:
    def __init__(self, rotation, translation):
        self.rotation, self.translation = np.asarray(rotation, dtype=np.float64), np.asarray(translation, dtype=np.float64)

    def inverse(self):
        return SE3(self.rotation.T, -self.rotation.T @ self.translation)

    def compose(self, right):
        return SE3(self.rotation @ right.rotation, self.rotation @ right.translation + self.translation)


def rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def main():
    refload.install_data_stubs(AV2_CLASSES)
    cp = refload.ref("datasets.pipelines.custom_pipeline")
    t3 = refload.ref("datasets.pipelines.transform_3d")
    ds = refload.ref("datasets.argoverse2_dataset")
    dt = refload.ref("datasets.argoverse2_dataset_t")
    sm = refload.ref("datasets.samplers.distributed_sampler")
    u = refload.ref("datasets.av2_utils")
    gold = {}

    # ------------------------------------------------------------------ f1: image pipeline
    rng = np.random.RandomState(3)
    for case, (flip, seed) in enumerate(((False, 7), (True, 11))):
        conf = dict(resize_lim=(0.47, 0.55), final_dim=(64, 96), bot_pct_lim=(0.0, 0.0), rot_lim=(0.0, 0.0), rand_flip=flip)
        shapes = [(155, 205), (155, 205), (205, 155)]          # (H, W): two landscape cameras, one portrait
        imgs = [rng.randint(0, 256, (h, w, 3)).astype(np.uint8) for h, w in shapes]
        for k in range(3):                                      # low-frequency content so that resampling matters smoothly too
            yy, xx = np.mgrid[0:shapes[k][0], 0:shapes[k][1]]
            imgs[k] = np.clip(imgs[k] * 0.3 + 120 + 90 * np.sin(xx / 9.0 + k)[..., None] * np.cos(yy / 7.0)[..., None], 0, 255).astype(np.uint8)
        intr = [np.eye(4) for _ in range(3)]
        for k in range(3):
            intr[k][0, 0] = intr[k][1, 1] = 180.0 + 5 * k
            intr[k][0, 2], intr[k][1, 2] = shapes[k][1] / 2.0, shapes[k][0] / 2.0
        extr = [np.eye(4) for _ in range(3)]
        for k in range(3):
            extr[k][:3, :3] = rot_z(0.7 * k)
            extr[k][:3, 3] = [0.1 * k, -0.2, 1.5]
        results = dict(img=[im.astype(np.float32) for im in imgs], intrinsics=copy.deepcopy(intr), extrinsics=copy.deepcopy(extr))
        np.random.seed(seed)
        results = cp.AV2ResizeCropFlipRotImageV2(data_aug_conf=conf)(results)
        results = t3.NormalizeMultiviewImage(mean=[103.530, 116.280, 123.675], std=[57.375, 57.120, 58.395], to_rgb=False)(results)
        results = cp.AV2PadMultiViewImage(size="same2max")(results)
        p = "pre%d_" % case
        for k in range(3):
            gold[p + "in%d" % k] = imgs[k]
        gold[p + "intr_in"], gold[p + "extr"] = np.stack(intr), np.stack(extr)
        gold[p + "seed"], gold[p + "flip"] = np.array(seed), np.array(flip)
        gold[p + "img"] = np.stack(results["img"]).astype(np.float32)                       # (N, H, W, 3)
        gold[p + "intr_out"] = np.stack(results["intrinsics"])
        gold[p + "lidar2img"] = np.stack(results["lidar2img"])
        gold[p + "ida_mat"] = np.stack(results["ida_mat"])
        gold[p + "pad_shape"] = np.array(results["pad_shape"])
        print("[data] f1 case %d: out %s, flip=%s" % (case, gold[p + "img"].shape, flip))

    # ------------------------------------------------------------------ f3: result path
    g = torch.Generator().manual_seed(5)
    outs, infos = [], []
    for s in range(3):
        n = 40 + 7 * s
        boxes = torch.cat([(torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([300.0, 300.0, 8.0]), torch.rand(n, 3, generator=g) * 4 + 0.3,
                           (torch.rand(n, 1, generator=g) - 0.5) * 6.2], dim=1)
        outs.append(dict(pts_bbox=dict(boxes_3d=refload.LiDARBoxes(boxes), scores_3d=torch.rand(n, generator=g),
                                       labels_3d=torch.randint(0, 26, (n,), generator=g))))
        infos.append(dict(scene_id="log-%c" % "bac"[s], lidar_timestamp_ns=315969904359876000 + 100000000 * s))
        gold["res_boxes%d" % s], gold["res_scores%d" % s], gold["res_labels%d" % s] = boxes.numpy(), outs[-1]["pts_bbox"]["scores_3d"].numpy(), \
            outs[-1]["pts_bbox"]["labels_3d"].numpy()
    dset = object.__new__(ds.Argoverse2Dataset)
    dset.data_infos = infos
    gold["res_cuboids0"] = dset.box_to_av2(outs[0]["pts_bbox"]["boxes_3d"]).numpy()
    gold["res_quat"] = u.yaw_to_quat(torch.linspace(-3.2, 3.2, 33)).numpy()
    frame = dset.format_results(outs).reset_index()
    gold["res_scene_ids"] = np.array([i["scene_id"] for i in infos])
    gold["res_ts"] = np.array([i["lidar_timestamp_ns"] for i in infos], dtype=np.int64)
    gold["res_frame_values"] = frame[list(ds.LABEL_ATTR) + ["score"]].to_numpy(dtype=np.float64)
    gold["res_frame_log_id"] = frame["log_id"].to_numpy().astype(str)
    gold["res_frame_ts"] = frame["timestamp_ns"].to_numpy().astype(np.int64)
    gold["res_frame_category"] = frame["category"].to_numpy().astype(str)
    print("[data] f3: %d detection rows" % len(frame))

    # ------------------------------------------------------------------ f4: streaming dataset / sampler semantics
    scene_ids = ["s0"] * 5 + ["s1"] * 7 + ["s2"] * 3 + ["s3"] * 6
    T = object.__new__(dt.Argoverse2DatasetT)
    T.data_infos = [dict(scene_id=s) for s in scene_ids]
    for split in (1, 2, "all"):
        T.seq_split_num = split
        T._set_sequence_group_flag()
        gold["seq_flag_%s" % split] = T.flag.copy()
    gold["seq_scene_ids"] = np.array(scene_ids)
    infos5 = list(range(23))
    gold["interval_order"] = np.array(infos5[::5] + infos5[1::5] + infos5[2::5] + infos5[3::5] + infos5[4::5])   # ref :27-31, spelled out
    shards = []
    for n, world in ((21, 4), (150, 8), (7, 8), (16, 2)):
        for rank in range(world):
            smp = sm.DistributedSampler(dataset=list(range(n)), num_replicas=world, rank=rank, shuffle=False)
            shards.append([n, world, rank] + list(iter(smp)))
    width = max(len(s) for s in shards)
    gold["sampler_shards"] = np.array([s + [-1] * (width - len(s)) for s in shards], dtype=np.int64)
    # get_data_info on synthetic infos
    rs = np.random.RandomState(1)
    infos = []
    for k in range(3):
        ego = SE3(rot_z(0.3 * k + 0.1), [10.0 * k, -3.0 + k, 0.2 * k])
        cams = {}
        for c in range(2):
            K = np.array([[1700.0 + c, 0, 1024.0], [0, 1701.0, 775.0 + c], [0, 0, 1.0]])
            cams["cam%d" % c] = dict(cam_timestamp_ns=315969904359876000 + k * 10 ** 8 + c, fpath="x/%d_%d.jpg" % (k, c),
                                     city_SE3_ego_cam_t=SE3(rot_z(0.3 * k + 0.1 + 0.001 * c), np.array([10.0 * k, -3.0 + k, 0.2 * k]) + 0.01 * c),
                                     ego_SE3_cam=SE3(rot_z(1.1 * c) @ np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]]), rs.randn(3)), intrinsics=K)
        infos.append(dict(scene_id="s%d" % (k // 2), lidar_timestamp_ns=315969904359876000 + k * 10 ** 8, city_SE3_ego_lidar_t=ego, cam_infos=cams))
    from pathlib import Path
    T.data_infos, T.split, T.data_root, T.modality, T.test_mode = infos, "val", Path("/data"), dict(use_camera=True), True
    for k in range(3):
        d = T.get_data_info(k)
        gold["info%d_ego_pose" % k], gold["info%d_ego_pose_inv" % k] = np.asarray(d["ego_pose"]), np.asarray(d["ego_pose_inv"])
        gold["info%d_timestamp" % k] = np.array(d["timestamp"])
        gold["info%d_intrinsics" % k], gold["info%d_extrinsics" % k], gold["info%d_lidar2img" % k] = np.stack(d["intrinsics"]), np.stack(d["extrinsics"]), \
            np.stack(d["lidar2img"])
        for c in range(2):
            ci = infos[k]["cam_infos"]["cam%d" % c]
            gold["info%d_cam%d_K" % (k, c)] = ci["intrinsics"]
            for nm in ("city_SE3_ego_cam_t", "ego_SE3_cam"):
                gold["info%d_cam%d_%s_R" % (k, c, nm)], gold["info%d_cam%d_%s_t" % (k, c, nm)] = ci[nm].rotation, ci[nm].translation
        gold["info%d_ego_R" % k], gold["info%d_ego_t" % k] = infos[k]["city_SE3_ego_lidar_t"].rotation, infos[k]["city_SE3_ego_lidar_t"].translation
    gold["info_scene_ids"] = np.array([i["scene_id"] for i in infos])
    path = os.path.join(GOLD, "far3d_data_contract.npz")
    np.savez_compressed(path, **gold)
    print("[data] wrote %s (%d arrays, %.0f KB)" % (path, len(gold), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if not refload.available():
        sys.exit("reference checkout not found: fixtures can only be regenerated in the build container")
    main()
