"""Deterministic synthetic inputs for the Far3D hot path: a ring of pinhole cameras,
N(0,1) images, identity ego pose.  Used by tests, bench.py and smoke(); contains no model code."""
import math

import torch

PC_RANGE = [-152.4, -152.4, -5.0, 152.4, 152.4, 5.0]   # reference projects/configs/far3d.py:10
LEVEL_STRIDES = (8, 16, 32, 64)                        # far3d.py:41


def level_shapes(pad_hw, strides=LEVEL_STRIDES):
    """FPN level (h,w) for a padded image: stride-8 map is ceil(H/8); each further level is the
    3x3/s2/p1 conv size of the previous one (mmdet FPN extra level) == ceil(prev/2)."""
    hw = []
    for s in strides:
        hw.append((-(-pad_hw[0] // s), -(-pad_hw[1] // s)))
    return hw


def level_starts(hw):
    st, acc = [], 0
    for h, w in hw:
        st.append(acc)
        acc += h * w
    return st, acc


def ring_cameras(num_cams, pad_hw, height=1.5, focal_scale=0.9, dtype=torch.float32):
    """Returns intrinsics, extrinsics (lidar->camera), lidar2img, each (num_cams,4,4).
    Lidar frame x fwd / y left / z up; camera frame x right / y down / z fwd; yaw_k = 2*pi*k/num_cams."""
    H, W = pad_hw
    f = focal_scale * W
    K = torch.eye(4, dtype=torch.float64)
    K[0, 0] = K[1, 1] = f
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    intr, extr, l2i = [], [], []
    for k in range(num_cams):
        th = 2.0 * math.pi * k / num_cams
        R = torch.tensor([[math.sin(th), -math.cos(th), 0.0],
                          [0.0, 0.0, -1.0],
                          [math.cos(th), math.sin(th), 0.0]], dtype=torch.float64)
        c = torch.tensor([0.0, 0.0, height], dtype=torch.float64)
        E = torch.eye(4, dtype=torch.float64)
        E[:3, :3] = R
        E[:3, 3] = -R @ c
        intr.append(K.clone())
        extr.append(E)
        l2i.append(K @ E)
    cvt = lambda xs: torch.stack(xs).to(dtype)
    return cvt(intr), cvt(extr), cvt(l2i)


def ego_pose_at(frame_index):
    """A gently curving drive: 1.2 m forward per frame, 2 degrees of yaw per frame, slight climb (ego -> global, 4x4 f32)."""
    th = math.radians(2.0) * frame_index
    E = torch.eye(4, dtype=torch.float64)
    E[0, 0], E[0, 1], E[1, 0], E[1, 1] = math.cos(th), -math.sin(th), math.sin(th), math.cos(th)
    E[0, 3], E[1, 3], E[2, 3] = 1.2 * frame_index * math.cos(th / 2), 1.2 * frame_index * math.sin(th / 2), 0.02 * frame_index
    return E


def make_frame(num_cams=7, pad_hw=(640, 960), seed=0, frame_index=0, device="cpu", ego_motion=False):
    """The `data` dict + img_metas the reference's forward_test hands to simple_test (batch 1).  ego_motion=False keeps the
    identity ego pose of SURVEY.md §8(d); True adds a per-frame pose so that the memory warp (farhead.py:464-477,503-507) is
    exercised with non-trivial matrices."""
    g = torch.Generator().manual_seed(seed * 1000 + frame_index)
    img = torch.randn(1, num_cams, 3, pad_hw[0], pad_hw[1], generator=g)
    intr, extr, l2i = ring_cameras(num_cams, pad_hw)
    pose = ego_pose_at(frame_index) if ego_motion else torch.eye(4, dtype=torch.float64)
    data = dict(
        img=img,
        lidar2img=l2i[None],
        intrinsics=intr[None],
        extrinsics=extr[None],
        ego_pose=pose.float()[None],
        ego_pose_inv=torch.linalg.inv(pose).float()[None],
        timestamp=torch.tensor([float(frame_index)], dtype=torch.float64),
    )
    data = {k: v.to(device) for k, v in data.items()}
    img_metas = [dict(pad_shape=[(pad_hw[0], pad_hw[1], 3)] * num_cams, scene_token="synthetic-scene-0")]
    return data, img_metas


def recipe_frame(rc, fi, device="cpu"):
    """Frame `fi` of a golden-fixture recipe (tests/golden/*.npz `recipe`): optional ego motion and a scene change (new
    scene_token, timestamps restart) at frame `scene_change_at`."""
    sc = rc.get("scene_change_at")
    scene = 1 if (sc is not None and fi >= sc) else 0
    local = fi - sc if scene else fi
    data, metas = make_frame(rc["num_cams"], tuple(rc["pad_hw"]), seed=rc["data_seed"] + 100 * scene, frame_index=local, device=device,
                             ego_motion=bool(rc.get("ego_motion")))
    metas[0]["scene_token"] = "synthetic-scene-%d" % scene
    return data, metas
