"""Streaming dataset / sampler semantics the temporal memory relies on.

The FarHead memory is only valid when one rank sees the frames of a scene in order: the reference guarantees it with
 * Argoverse2DatasetT.get_data_info (ref datasets/argoverse2_dataset_t.py:143-240): `timestamp = index` (the dataset index, not
   the sensor time), `scene_token = scene_id`, ego_pose = city_SE3_ego (4x4) and its closed-form inverse in float32;
 * the `interval_test` re-ordering (ref :27-31) and the per-scene group flags (ref :41-80);
 * the non-shuffling, CONTIGUOUS per-rank shard of the test sampler (ref datasets/samplers/distributed_sampler.py:28-47) so that a
   scene's frames stay on one rank, and the scene-change test of the detector (ref detectors/far3d.py:252-257).
This module restates exactly that index arithmetic; tests/golden/far3d_data_contract.npz holds outputs of the reference's own
classes for it."""
import math

import numpy as np


def interval_test_order(n, k=5):
    """Index order after `interval_test` (ref argoverse2_dataset_t.py:27-31): every k-th info first, then offset 1, ..."""
    idx = list(range(n))
    out = []
    for s in range(k):
        out += idx[s::k]
    return out


def sequence_group_flags(scene_ids, seq_split_num=1):
    """ref argoverse2_dataset_t.py:41-80 (_set_sequence_group_flag): one group per run of equal scene ids, optionally split into
    `seq_split_num` sub-sequences ('all' = every frame its own group)."""
    res, cur, scene = [], -1, None
    for s in scene_ids:
        if s != scene:
            scene = s
            cur += 1
        res.append(cur)
    flag = np.array(res, dtype=np.int64)
    if seq_split_num != 1:
        if seq_split_num == "all":
            return np.array(range(len(scene_ids)), dtype=np.int64)
        bin_counts = np.bincount(flag)
        new_flags, cur_new = [], 0
        for cur_flag in range(len(bin_counts)):
            edges = np.array(list(range(0, bin_counts[cur_flag], math.ceil(bin_counts[cur_flag] / seq_split_num))) + [bin_counts[cur_flag]])
            for n_sub in (edges[1:] - edges[:-1]):
                new_flags += [cur_new] * int(n_sub)
                cur_new += 1
        assert len(new_flags) == len(flag)
        flag = np.array(new_flags, dtype=np.int64)
    return flag


def contiguous_shard(n, num_replicas, rank):
    """Indices of `rank` under the reference's test sampler (ref samplers/distributed_sampler.py:28-47): no shuffle, the index
    list padded by wrap-around to a multiple of the world size, then a CONTIGUOUS block per rank."""
    num_samples = int(math.ceil(n / num_replicas))
    total = num_samples * num_replicas
    indices = (list(range(n)) * math.ceil(total / n))[:total]
    return indices[rank * num_samples:(rank + 1) * num_samples]


def invert_ego_pose(ego_pose):
    """ref argoverse2_dataset_t.py:266-274: closed-form inverse of a rigid 4x4, float32."""
    inv = np.zeros((4, 4), dtype=np.float32)
    rot, tr = ego_pose[:3, :3], ego_pose[:3, 3]
    inv[:3, :3] = rot.T
    inv[:3, 3] = -np.dot(rot.T, tr)
    inv[3, 3] = 1.0
    return inv


def camera_matrices(intrinsic, ego_SE3_cam, city_SE3_ego_cam_t, city_SE3_ego_lidar_t):
    """ref argoverse2_dataset_t.py:189-214.  Arguments are (R (3,3), t (3,)) pairs (av2 SE3 rotation / translation); returns
    (viewpad intrinsics 4x4, extrinsics 4x4 = cam <- ego at lidar time, lidar2img)."""
    def inv(p):
        R, t = p
        return R.T, -R.T @ t

    def compose(a, b):          # av2 SE3.compose: a then-right-multiplied by b
        return a[0] @ b[0], a[0] @ b[1] + a[1]
    ego_cam_t_SE3_ego_lidar_t = compose(inv(city_SE3_ego_cam_t), city_SE3_ego_lidar_t)
    cam_SE3_ego = compose(inv(ego_SE3_cam), ego_cam_t_SE3_ego_lidar_t)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = cam_SE3_ego
    viewpad = np.eye(4)
    viewpad[:intrinsic.shape[0], :intrinsic.shape[1]] = intrinsic
    return viewpad, T, viewpad @ T


class StreamingIndex:
    """What one rank iterates at test time: (dataset index, frame dict) pairs in the reference's order, with the fields the
    detector's streaming logic consumes.  infos: list of dicts with `scene_id`, `lidar_timestamp_ns`, `city_SE3_ego` = (R, t)."""

    def __init__(self, infos, interval_test=False, num_replicas=1, rank=0):
        order = interval_test_order(len(infos)) if interval_test else list(range(len(infos)))
        self.infos = [infos[i] for i in order]
        self.flag = sequence_group_flags([i["scene_id"] for i in self.infos])
        self.indices = contiguous_shard(len(self.infos), num_replicas, rank)

    def frame(self, index):
        info = self.infos[index]
        ego = np.eye(4)
        ego[:3, :3], ego[:3, 3] = info["city_SE3_ego"]
        return dict(scene_token=info["scene_id"], timestamp=index, lidar_timestamp=info["lidar_timestamp_ns"], ego_pose=ego,
                    ego_pose_inv=invert_ego_pose(ego))

    def __iter__(self):
        prev = None
        for i in self.indices:
            f = self.frame(i)
            f["prev_exists"] = prev is not None and prev == f["scene_token"]     # what far3d.py:252-257 derives from scene_token
            prev = f["scene_token"]
            yield i, f
