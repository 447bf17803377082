"""CPU: the data contract around the hot path (SURVEY.md §8 f1 host part, f3, f4) against fixtures produced by the reference's
own classes (tools/gen_golden_data.py -> tests/golden/far3d_data_contract.npz), and the Pillow resampling tables against Pillow."""
import os

import numpy as np
import pytest
import torch

from far3d_amd import data_pipeline as dp
from far3d_amd.data_pipeline import results, streaming
from tests.conftest import ROOT

Z = np.load(os.path.join(ROOT, "tests", "golden", "far3d_data_contract.npz"))


# ------------------------------------------------------------------------------------------------ f1 (host side)
@pytest.mark.parametrize("geom", [(155, 205, 77, 102), (100, 64, 135, 86), (60, 90, 60, 90), (31, 47, 16, 25)])
def test_resample_tables_reproduce_pillow_bit_for_bit(geom):
    """The fixed-point tables + two-pass integer arithmetic the HIP kernels execute == PIL.Image.resize (default BICUBIC)."""
    from PIL import Image
    H, W, oh, ow = geom
    img = np.random.RandomState(H + W).randint(0, 256, (H, W, 3)).astype(np.uint8)
    want = np.array(Image.fromarray(img).resize((ow, oh)))
    from oracle.resample import resample_u8_reference
    assert np.array_equal(resample_u8_reference(img, ow, oh), want)


@pytest.mark.parametrize("case", [0, 1])
def test_augmentation_draws_and_calibration_match_reference(case):
    p = "pre%d_" % case
    conf = dict(dp.preprocess.DEFAULT_AUG, final_dim=(64, 96), rand_flip=bool(Z[p + "flip"]))
    rng = np.random.RandomState(int(Z[p + "seed"]))
    pre = dp.ImagePreprocessor.__new__(dp.ImagePreprocessor)        # host logic only: no device needed
    pre.conf, pre.rng = conf, rng
    plans = pre.plan([tuple(Z[p + "in%d" % k].shape[:2]) for k in range(3)])
    res = dict(intrinsics=[m.copy() for m in Z[p + "intr_in"]], extrinsics=[m.copy() for m in Z[p + "extr"]])
    dp.ImagePreprocessor.update_calibration(res, plans)
    assert np.allclose(np.stack(res["ida_mat"]), Z[p + "ida_mat"], atol=0, rtol=0)
    assert np.allclose(np.stack(res["intrinsics"]), Z[p + "intr_out"], rtol=1e-12, atol=1e-12)
    assert np.allclose(np.stack(res["lidar2img"]), Z[p + "lidar2img"], rtol=1e-12, atol=1e-12)
    assert plans[2][0] is not None and plans[0][0] is None          # the portrait camera takes the two-stage path
    assert any(pl[3] for pl in plans) == bool(case == 1 and any(pl[3] for pl in plans))


# ------------------------------------------------------------------------------------------------ f3
def test_yaw_to_quat_and_box_to_av2():
    assert np.allclose(results.yaw_to_quat(torch.linspace(-3.2, 3.2, 33)).numpy(), Z["res_quat"], atol=1e-7)
    got = results.box_to_av2(torch.from_numpy(Z["res_boxes0"])).numpy()
    assert np.allclose(got, Z["res_cuboids0"], atol=1e-6)
    assert np.allclose(got[:, 2], Z["res_boxes0"][:, 2] + Z["res_boxes0"][:, 5] * 0.5)     # gravity centre


def test_format_results_reproduces_reference_frame(tmp_path):
    outs = [dict(pts_bbox=dict(boxes_3d=torch.from_numpy(Z["res_boxes%d" % s]), scores_3d=torch.from_numpy(Z["res_scores%d" % s]),
                               labels_3d=torch.from_numpy(Z["res_labels%d" % s]))) for s in range(3)]
    infos = [dict(scene_id=str(Z["res_scene_ids"][s]), lidar_timestamp_ns=int(Z["res_ts"][s])) for s in range(3)]
    frame = results.format_results(outs, infos, feather_path=str(tmp_path / "dets")).reset_index()
    assert list(frame["log_id"].astype(str)) == list(Z["res_frame_log_id"])
    assert np.array_equal(frame["timestamp_ns"].to_numpy().astype(np.int64), Z["res_frame_ts"])
    assert list(frame["category"].astype(str)) == list(Z["res_frame_category"])
    assert np.allclose(frame[list(results.LABEL_ATTR) + ["score"]].to_numpy(dtype=np.float64), Z["res_frame_values"], atol=1e-6)
    import pandas as pd
    sub = pd.read_feather(str(tmp_path / "dets.feather"))
    assert list(sub["score"]) == sorted(sub["score"], reverse=True) and len(sub) == len(frame)


# ------------------------------------------------------------------------------------------------ f4
def test_sequence_flags_interval_order_and_contiguous_shards():
    ids = list(Z["seq_scene_ids"])
    for split in (1, 2, "all"):
        assert np.array_equal(streaming.sequence_group_flags(ids, split), Z["seq_flag_%s" % split])
    assert streaming.interval_test_order(23) == list(Z["interval_order"])
    for row in Z["sampler_shards"]:
        n, world, rank = int(row[0]), int(row[1]), int(row[2])
        want = [int(v) for v in row[3:] if v >= 0]
        assert streaming.contiguous_shard(n, world, rank) == want, (n, world, rank)


def test_frame_fields_match_get_data_info():
    ids = [str(s) for s in Z["info_scene_ids"]]
    infos = [dict(scene_id=ids[k], lidar_timestamp_ns=k, city_SE3_ego=(Z["info%d_ego_R" % k], Z["info%d_ego_t" % k])) for k in range(3)]
    idx = streaming.StreamingIndex(infos)
    frames = list(idx)
    for k, (i, f) in enumerate(frames):
        assert i == k and f["timestamp"] == int(Z["info%d_timestamp" % k]) == k       # the memory's clock is the dataset index
        assert np.array_equal(f["ego_pose"], Z["info%d_ego_pose" % k])
        assert f["ego_pose_inv"].dtype == np.float32 and np.array_equal(f["ego_pose_inv"], Z["info%d_ego_pose_inv" % k])
        intr, extr, l2i = [], [], []
        for c in range(2):
            pose = lambda nm: (Z["info%d_cam%d_%s_R" % (k, c, nm)], Z["info%d_cam%d_%s_t" % (k, c, nm)])
            a, b, m = streaming.camera_matrices(Z["info%d_cam%d_K" % (k, c)], pose("ego_SE3_cam"), pose("city_SE3_ego_cam_t"),
                                                (Z["info%d_ego_R" % k], Z["info%d_ego_t" % k]))
            intr.append(a); extr.append(b); l2i.append(m)
        assert np.allclose(np.stack(intr), Z["info%d_intrinsics" % k], atol=1e-12)
        assert np.allclose(np.stack(extr), Z["info%d_extrinsics" % k], atol=1e-12)
        assert np.allclose(np.stack(l2i), Z["info%d_lidar2img" % k], atol=1e-9)
    assert [f["prev_exists"] for _, f in frames] == [False, True, False]       # scene s0, s0, s1


def test_two_rank_streams_keep_scenes_together_where_the_sampler_allows():
    infos = [dict(scene_id="s%d" % (k // 5), lidar_timestamp_ns=k, city_SE3_ego=(np.eye(3), np.zeros(3))) for k in range(20)]
    a, b = streaming.StreamingIndex(infos, num_replicas=2, rank=0), streaming.StreamingIndex(infos, num_replicas=2, rank=1)
    assert a.indices == list(range(10)) and b.indices == list(range(10, 20))
    assert [f["prev_exists"] for _, f in b][:6] == [False, True, True, True, True, False]
