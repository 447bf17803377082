"""Result path after the hot path: decoded boxes -> Argoverse 2 detection rows.

Mirrors Argoverse2Dataset.format_results / box_to_av2 (ref datasets/argoverse2_dataset.py:267-341) and yaw_to_quat / xyz_to_quat
(ref datasets/av2_utils.py:240-283).  The boxes handed over are what the detector returns (bbox3d2result of
LiDARInstance3DBoxes, ref detectors/far3d.py:262-265): (x, y, z_bottom, w, l, h, yaw[, vx, vy]) rows -- the engine's `boxes_3d`
already carries the bottom-centre z (ref farhead.py:1236-1238).  The frame produced here is the input of the AV2 metric
(av2_metric.evaluate in this package)."""
import torch

LABEL_ATTR = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz")   # ref argoverse2_dataset.py:15-17
# av2.evaluation.detection.constants.CompetitionCategories, in the order the reference config lists them (far3d.py:15-21)
AV2_CLASSES = ("ARTICULATED_BUS", "BICYCLE", "BICYCLIST", "BOLLARD", "BOX_TRUCK", "BUS", "CONSTRUCTION_BARREL", "CONSTRUCTION_CONE",
               "DOG", "LARGE_VEHICLE", "MESSAGE_BOARD_TRAILER", "MOBILE_PEDESTRIAN_CROSSING_SIGN", "MOTORCYCLE", "MOTORCYCLIST",
               "PEDESTRIAN", "REGULAR_VEHICLE", "SCHOOL_BUS", "SIGN", "STOP_SIGN", "STROLLER", "TRUCK", "TRUCK_CAB", "VEHICULAR_TRAILER",
               "WHEELCHAIR", "WHEELED_DEVICE", "WHEELED_RIDER")


def xyz_to_quat(xyz_rad):
    """Euler angles (roll, pitch, yaw) -> scalar-first quaternions (ref av2_utils.py:240-268)."""
    x, y, z = xyz_rad[..., 0], xyz_rad[..., 1], xyz_rad[..., 2]
    cy, sy = torch.cos(z * 0.5), torch.sin(z * 0.5)
    cp, sp = torch.cos(y * 0.5), torch.sin(y * 0.5)
    cr, sr = torch.cos(x * 0.5), torch.sin(x * 0.5)
    qw = cr * cp * cy + sr * sp * sy
    qx = sr * cp * cy - cr * sp * sy
    qy = cr * sp * cy + sr * cp * sy
    qz = cr * cp * sy - sr * sp * cy
    return torch.stack([qw, qx, qy, qz], dim=-1)


def yaw_to_quat(yaw_rad):
    """ref av2_utils.py:271-283."""
    xyz = torch.zeros_like(yaw_rad)[..., None].repeat_interleave(3, dim=-1)
    xyz[..., -1] = yaw_rad
    return xyz_to_quat(xyz)


def box_to_av2(boxes_3d):
    """(n, >=7) rows (x, y, z_bottom, w, l, h, yaw, ...) -> (n,10) AV2 cuboids (gravity centre, the three size columns in the
    tensor's order, quaternion) -- ref argoverse2_dataset.py:333-341; gravity centre = bottom centre + h/2 (mmdet3d
    LiDARInstance3DBoxes.gravity_center)."""
    t = boxes_3d if isinstance(boxes_3d, torch.Tensor) else torch.as_tensor(boxes_3d)
    centre = torch.cat([t[:, :2], t[:, 2:3] + t[:, 5:6] * 0.5], dim=1)
    return torch.cat([centre, t[:, [3, 4, 5]], yaw_to_quat(t[:, 6])], dim=1)


def format_results(outputs, data_infos, classes=AV2_CLASSES, feather_path=None):
    """ref argoverse2_dataset.py:267-331.  outputs: one dict per sample with boxes_3d / scores_3d / labels_3d (optionally under
    'pts_bbox'); data_infos: per sample `scene_id` and `lidar_timestamp_ns`.  Returns the detections frame indexed by
    (log_id, timestamp_ns), sorted like the reference; feather_path additionally writes the score-sorted submission file."""
    import pandas as pd
    assert len(data_infos) == len(outputs)
    frames = []
    for out_i, info in zip(outputs, data_infos):
        if "pts_bbox" in out_i:
            out_i = out_i["pts_bbox"]
        labels = torch.as_tensor(out_i["labels_3d"]).cpu().numpy().tolist()
        df = pd.DataFrame(box_to_av2(torch.as_tensor(out_i["boxes_3d"]).cpu()).numpy(), columns=list(LABEL_ATTR))
        df["score"] = torch.as_tensor(out_i["scores_3d"]).cpu().numpy()
        df["log_id"] = info["scene_id"]
        df["timestamp_ns"] = int(info["lidar_timestamp_ns"])
        df["category"] = [classes[i].upper() for i in labels]
        frames.append(df)
    dts = pd.concat(frames).set_index(["log_id", "timestamp_ns"]).sort_index()
    dts = dts.sort_values("score", ascending=False).reset_index()
    if feather_path is not None:
        if not feather_path.endswith(".feather"):
            feather_path = feather_path + ".feather"
        dts.to_feather(feather_path)
    return dts.set_index(["log_id", "timestamp_ns"]).sort_index()
