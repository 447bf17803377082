"""GPU: every registry class of the drop-in surface works as a stand-alone module: FPN, YOLOXHeadCustom
(forward + get_bboxes), FarHead (forward + get_bboxes) and MultiheadAttention are called with the reference's signatures
(ref detectors/far3d.py:79-85,122-135,244-266; detr3d_transformer.py:385-394) and compared with the oracle's statement of the
same stage, fp32, on the golden recipe's seeded weights and frames."""
import json
import os

import numpy as np
import pytest
import torch

from far3d_amd import config, plugin, synth, weights
from tests.conftest import ROOT, assert_detections_match

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rig(hip_lib):
    from oracle import far3d_oracle
    z = np.load(os.path.join(ROOT, "tests", "golden", "far3d_small_seq.npz"))
    rc = json.loads(bytes(z["recipe"]).decode())
    kw = dict(num_cams=rc["num_cams"], num_query=rc["num_query"], num_propagated=rc["num_propagated"], memory_len=rc["memory_len"],
              topk_proposals=rc["topk_proposals"])
    det = plugin.build_detector(config.default_model_cfg(**kw))
    sd = weights.init_state_dict(weights.detector_spec(rc["backbone"], num_query=rc["num_query"], num_propagated=rc["num_propagated"]),
                                 seed=rc["weight_seed"])
    det.load_state_dict(sd)
    for m in (det.img_backbone, det.img_neck, det.img_roi_head, det.pts_bbox_head):
        m.precision = "fp32"
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(**kw))
    return det, orc, rc


def test_modules_chain_like_the_reference_detector(rig):
    det, orc, rc = rig
    rel = lambda a, b: (a - b).abs().max().item() / max(1e-6, b.abs().max().item())
    for fi in range(2):
        data, metas = synth.recipe_frame(rc, fi)
        img = data["img"][0]
        with torch.no_grad():
            w_bb = orc.backbone(img)
            w_fpn = orc.fpn(w_bb)
            w_roi = orc.roi_head(w_fpn)
            w_roi.update(orc.get_bboxes(w_roi))
        # ---- detectors/far3d.py:79-85: backbone, neck
        bb = det.img_backbone(img.to(DEV))
        assert all(rel(g.cpu(), w) < 1e-4 for g, w in zip(bb, w_bb))
        fpn = det.img_neck(bb)
        assert len(fpn) == 4 and all(rel(g.cpu(), w) < 1e-4 for g, w in zip(fpn, w_fpn))
        feats = [f[None] for f in fpn]                               # (B, N, C, h, w)
        # ---- detectors/far3d.py:122-124, 247-249: 2D head
        dev_data = {k: v.to(DEV) for k, v in data.items()}
        roi = det.img_roi_head(None, img_feats=feats, **{k: v for k, v in dev_data.items() if k != "img"})
        for key in ("enc_cls_scores", "enc_bbox_preds", "objectnesses"):
            assert all((g.cpu() - w).abs().max().item() < 2e-4 for g, w in zip(roi[key], w_roi[key])), key
        assert (roi["depth_logit"].cpu() - w_roi["depth_logit"]).abs().max().item() < 2e-4
        assert len(roi["pred_centers2d_offset"]) == 4 and roi["pred_centers2d_offset"][0].shape[1] == 2
        roi.update(det.img_roi_head.get_bboxes(roi))
        assert torch.equal(roi["valid_indices"].cpu(), w_roi["valid_indices"])
        assert torch.allclose(torch.cat(roi["bbox_list"]).cpu(), torch.cat(w_roi["bbox_list"]), rtol=2e-3, atol=2e-3)
        assert (roi["bbox2d_scores"].cpu() - w_roi["bbox2d_scores"]).abs().max().item() < 1e-4
        # ---- detectors/far3d.py:252-266: scene-change flag, 3D head, decode
        prev = torch.zeros(1) if fi == 0 else torch.ones(1)
        with torch.no_grad():
            w_out = orc.head_forward(w_fpn, w_roi, data, prev, tuple(rc["pad_hw"]))
            w_res = orc.decode(w_out)
        out = det.pts_bbox_head(metas, roi, img_feats=feats, prev_exists=prev, **{k: v for k, v in dev_data.items() if k != "img"})
        assert out["dn_mask_dict"] is None
        assert (out["all_cls_scores"].cpu() - w_out["all_cls_scores"]).abs().max().item() < 1e-3
        boxes, scores, labels = det.pts_bbox_head.get_bboxes(out, metas)[0]
        assert_detections_match((labels.cpu().numpy(), boxes.cpu().numpy(), scores.cpu().numpy()),
                                tuple(w_res[k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)


def test_multihead_attention_module_matches_mmcv_semantics(rig):
    det, orc, rc = rig
    mha = det.pts_bbox_head.transformer.decoder.layers[2].attentions[0]
    mha.precision = "fp32"
    g = torch.Generator().manual_seed(4)
    x, qpos = torch.randn(1, 41, 256, generator=g), torch.randn(1, 41, 256, generator=g)
    mem, mpos = torch.randn(1, 23, 256, generator=g), torch.randn(1, 23, 256, generator=g)
    with torch.no_grad():
        want = orc._self_attn(x, qpos, mem, mpos, "pts_bbox_head.transformer.decoder.layers.2.")
    key, kpos = torch.cat([x, mem], 1), torch.cat([qpos, mpos], 1)
    got = mha.to(DEV)(x.to(DEV), key.to(DEV), key.to(DEV), None, query_pos=qpos.to(DEV), key_pos=kpos.to(DEV))
    assert got.shape == want.shape and (got.cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_backbone_module_takes_more_images_than_cameras_in_bf16(rig):
    """the registry VoVNet runs its engine with the default 7-camera configuration and must take any batch of images
    (ref models/backbones/vovnet.py:349-360 -- cameras are just the batch dimension).  In bf16 mode the eSE channel sums come from
    the concat GEMM's epilogue into a per-image scratch sized for the configured cameras: 9 images must run, and give image by image
    what 3 + 3 + 3 give."""
    det, orc, rc = rig
    bb = det.img_backbone
    old = bb.precision
    bb.precision = "bf16"
    try:
        g = torch.Generator().manual_seed(11)
        imgs = torch.randn(9, 3, rc["pad_hw"][0], rc["pad_hw"][1], generator=g).to(DEV)
        all9 = [t.float().clone() for t in bb(imgs)]
        for k in range(3):
            part = bb(imgs[3 * k:3 * k + 3])
            for a, b in zip(all9, part):      # the tile table may pick another tile for another pixel count: bf16 rounding, not bits
                d = (a[3 * k:3 * k + 3] - b.float()).abs().max().item()
                assert d <= 2e-2 * max(1.0, a.abs().max().item()), d
    finally:
        bb.precision = old
