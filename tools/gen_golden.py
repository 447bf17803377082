"""Pin the oracle to the reference and emit golden fixtures (BUILD CONTAINER ONLY: needs /root/reference).

  python tools/gen_golden.py [name ...]   # checks oracle == reference, rewrites tests/golden/*.npz|json (all, or the named fixtures)

What it does
  1. builds the reference's own Far3D detector (its files, loaded where they lie, oracle/refload.py),
  2. checks far3d_amd.weights.detector_spec() == the reference state-dict schema (names and shapes),
  3. loads the SAME seeded weights into the reference and into oracle/far3d_oracle.py, runs synthetic streaming sequences
     through both -- 2 cameras with ego motion and a scene change, the single-camera 256x256 case of BASELINE.json
     configs[0], and a 7-frame sequence that overflows the streaming memory queue -- and asserts agreement,
  4. writes the inputs' recipe (seeds, sizes) and the reference's outputs to tests/golden/.
Fixtures hold data only (seeds + expected tensors) -- never reference source.
"""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd import synth, weights  # noqa: E402
from oracle import far3d_oracle, refload  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# streaming sequence with ego motion and a scene change (frames 0-1 scene 0, frames 2-3 scene 1): memory warp, prev_exists
# masking and the memory reset are all exercised against the reference
SMALL = dict(name="far3d_small_seq", num_cams=2, pad_hw=(64, 96), num_query=60, num_propagated=16, memory_len=64, topk_proposals=16,
             backbone="V-99-eSE", weight_seed=1, data_seed=5, frames=4, ego_motion=True, scene_change_at=2)
# BASELINE.json configs[0] scale: ONE camera, 256x256, 100 fixed queries (84 learned + 16 propagated).  The reference's
# Far3D.extract_img_feat cannot take a single camera (img.squeeze_() drops the camera axis, detectors/far3d.py:71-72), so its
# backbone + neck are called directly and everything downstream runs through the reference's own methods.
C1 = dict(name="far3d_c1_seq", num_cams=1, pad_hw=(256, 256), num_query=84, num_propagated=16, memory_len=64, topk_proposals=16,
          backbone="V-99-eSE", weight_seed=2, data_seed=9, frames=2, ego_motion=True)
# A streaming sequence that OVERFLOWS the memory queue (farhead.py:453-508): 7 frames of one scene with ego motion, memory_len =
# 3 x num_propagated, 16 entries pushed per frame -- full after frame 2, so frames 3-6 drop live entries at the truncation of
# pre_update_memory (farhead.py:467-471) and carry timestamps / poses that are >= 3 frames old (configs[4]'s "4-frame memory queue"
# at toy size)
OVERFLOW = dict(name="far3d_overflow_seq", num_cams=2, pad_hw=(64, 96), num_query=60, num_propagated=16, memory_len=48, topk_proposals=16,
                backbone="V-99-eSE", weight_seed=3, data_seed=11, frames=7, ego_motion=True)


def run_reference(model, data, img_metas):
    """The body of Far3D.simple_test / simple_test_pts (detectors/far3d.py:244-277), calling the reference's methods."""
    data = dict(data)
    if data["img"].shape[1] == 1:      # single camera: see C1 above
        feats = model.img_neck(model.img_backbone(data["img"][0]))
        data["img_feats"] = [f[None] for f in feats]
    else:
        data["img_feats"] = model.extract_img_feat(data["img"])
    location = model.prepare_location(img_metas, **data)
    outs_roi = model.forward_roi_head(location, **data)
    outs_roi.update(model.img_roi_head.get_bboxes(outs_roi))
    if img_metas[0]["scene_token"] != model.prev_scene_token:
        model.prev_scene_token = img_metas[0]["scene_token"]
        data["prev_exists"] = data["img"].new_zeros(1)
        model.pts_bbox_head.reset_memory()
    else:
        data["prev_exists"] = data["img"].new_ones(1)
    outs = model.pts_bbox_head(img_metas, outs_roi, **data)
    boxes, scores, labels = model.pts_bbox_head.get_bboxes(copy.copy(outs), img_metas)[0]
    return dict(img_feats=[f[0] for f in data["img_feats"]], roi=outs_roi, outs=outs,
                result=dict(boxes_3d=boxes.tensor, scores_3d=scores, labels_3d=labels))


def main():
    only = set(sys.argv[1:])
    for c in (SMALL, C1, OVERFLOW):
        if not only or c["name"] in only:
            generate(c)


def generate(c):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg, _ = refload.reference_model_cfg(num_cams=c["num_cams"], num_query=c["num_query"], num_propagated=c["num_propagated"],
                                         memory_len=c["memory_len"], topk_proposals=c["topk_proposals"])
    model = refload.build_reference_detector(cfg)
    ref_sd = model.state_dict()

    # ---- 2. schema check (small and full-size variants share the code path; full size differs only in 2 embeddings)
    spec = weights.detector_spec(c["backbone"], num_query=c["num_query"], num_propagated=c["num_propagated"])
    canon = {}
    for k, v in ref_sd.items():
        ck = weights.canonical_key(k)
        if ck is not None:
            canon[ck] = tuple(v.shape)
    assert set(canon) == set(spec), (sorted(set(canon) - set(spec))[:5], sorted(set(spec) - set(canon))[:5])
    for k in spec:
        assert tuple(spec[k]) == canon[k], (k, spec[k], canon[k])
    print("[golden] schema: %d tensors, %.1f M parameters -- matches the reference" %
          (len(spec), sum(int(np.prod(s)) for s in spec.values()) / 1e6))
    if c is SMALL:
        full = weights.detector_spec("V-99-eSE")
        manifest = {k: list(v) for k, v in full.items()}
        json.dump(manifest, open(os.path.join(GOLD, "state_dict_manifest.json"), "w"), indent=0, sort_keys=True)

    # ---- 3. same weights into both
    sd = weights.init_state_dict(spec, seed=c["weight_seed"])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(weights.canonical_key(m) is None or weights.canonical_key(m) != m for m in missing), missing
    ocfg = far3d_oracle.default_cfg(num_cams=c["num_cams"], num_query=c["num_query"], num_propagated=c["num_propagated"],
                                    memory_len=c["memory_len"], topk_proposals=c["topk_proposals"])
    orc = far3d_oracle.Far3DOracle(sd, ocfg)

    gold = {}
    worst = 0.0
    with torch.no_grad():
        for fi in range(c["frames"]):
            data, metas = synth.recipe_frame(c, fi)
            metas[0]["box_type_3d"] = refload.LiDARBoxes
            r = run_reference(model, copy.deepcopy(data), metas)
            o = orc.simple_test(copy.deepcopy(data), metas)
            pairs = [("fpn%d" % l, r["img_feats"][l], o["feat_levels"][l]) for l in range(4)]
            pairs += [("depth_logit", r["roi"]["depth_logit"], o["roi"]["depth_logit"]),
                      ("bbox2d_scores", r["roi"]["bbox2d_scores"], o["roi"]["bbox2d_scores"]),
                      ("bbox2d", torch.cat(r["roi"]["bbox_list"]), torch.cat(o["roi"]["bbox_list"])),
                      ("all_cls_scores", r["outs"]["all_cls_scores"], o["all_cls_scores"]),
                      ("all_bbox_preds", r["outs"]["all_bbox_preds"], o["all_bbox_preds"])]
            for name, a, b in pairs:
                assert a.shape == b.shape, (fi, name, a.shape, b.shape)
                err = (a - b).abs().max().item() if a.numel() else 0.0
                worst = max(worst, err)
                if name == "all_cls_scores":
                    # how far two fp32 restatements of the same arithmetic (reference vs oracle: other summation orders) are apart on
                    # this frame's logits: the yardstick a third fp32 implementation (the HIP engine) is read against in the tests
                    gold["f%d_oracle_logit_dev" % fi] = np.float32(err)
                scale = max(1.0, a.abs().max().item()) if a.numel() else 1.0   # box coordinates are metres / pixels
                assert err < 2e-4 * scale, "frame %d %s: oracle deviates from the reference by %.3e" % (fi, name, err)
            # decoded boxes: compared as a set above the top-k bar (the rank of near-tied scores is summation-order noise)
            from tests.conftest import assert_detections_match
            assert_detections_match(tuple(o["result"][k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                    tuple(r["result"][k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)
            assert torch.equal(r["roi"]["valid_indices"], o["roi"]["valid_indices"])
            M = r["roi"]["bbox2d_scores"].shape[0]
            print("[golden] frame %d: M=%d adaptive queries, A=%d, %d boxes, max |oracle-reference| so far %.2e" %
                  (fi, M, r["outs"]["all_cls_scores"].shape[2], r["result"]["boxes_3d"].shape[0], worst))
            gold["f%d_all_cls_scores" % fi] = r["outs"]["all_cls_scores"].numpy()
            gold["f%d_all_bbox_preds" % fi] = r["outs"]["all_bbox_preds"].numpy()
            gold["f%d_boxes_3d" % fi] = r["result"]["boxes_3d"].numpy()
            gold["f%d_scores_3d" % fi] = r["result"]["scores_3d"].numpy()
            gold["f%d_labels_3d" % fi] = r["result"]["labels_3d"].numpy()
            gold["f%d_bbox2d" % fi] = torch.cat(r["roi"]["bbox_list"]).numpy()
            gold["f%d_bbox2d_scores" % fi] = r["roi"]["bbox2d_scores"].numpy()
            gold["f%d_valid_idx" % fi] = r["roi"]["valid_indices"].nonzero().numpy().astype(np.int32)
            gold["f%d_depth_argmax" % fi] = r["roi"]["pred_depth"].argmax(1).numpy().astype(np.int16)
            for l in range(4):
                gold["f%d_fpn%d_sample" % (fi, l)] = r["img_feats"][l][:, ::16, ::2, ::3].numpy()   # sparse sample of the maps
    gold["recipe"] = np.frombuffer(json.dumps(c).encode(), dtype=np.uint8)
    path = os.path.join(GOLD, c["name"] + ".npz")
    np.savez_compressed(path, **gold)
    print("[golden] wrote %s (worst oracle-vs-reference deviation %.2e)" % (path, worst))


if __name__ == "__main__":
    if not refload.available():
        sys.exit("reference checkout not found: fixtures can only be regenerated in the build container")
    main()
