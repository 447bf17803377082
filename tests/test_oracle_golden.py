"""CPU: the oracle reproduces the golden vectors the reference's own files produced (tools/gen_golden.py), and the
state-dict schema matches the reference's (manifest fixture)."""
import json
import os

import numpy as np
import torch

from far3d_amd import synth, weights
from oracle import far3d_oracle
import pytest

from tests.conftest import ROOT, assert_detections_match

GOLD = os.path.join(ROOT, "tests", "golden")


def load_small(name="far3d_small_seq"):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    recipe = json.loads(bytes(z["recipe"]).decode())
    return z, recipe


def small_oracle(recipe):
    spec = weights.detector_spec(recipe["backbone"], num_query=recipe["num_query"], num_propagated=recipe["num_propagated"])
    sd = weights.init_state_dict(spec, seed=recipe["weight_seed"])
    cfg = far3d_oracle.default_cfg(num_cams=recipe["num_cams"], num_query=recipe["num_query"],
                                   num_propagated=recipe["num_propagated"], memory_len=recipe["memory_len"],
                                   topk_proposals=recipe["topk_proposals"])
    return far3d_oracle.Far3DOracle(sd, cfg), sd


def test_schema_matches_reference_manifest():
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    spec = weights.detector_spec("V-99-eSE")
    assert {k: list(v) for k, v in spec.items()} == man
    assert len(spec) == 870
    assert weights.canonical_key("pts_bbox_head.cls_branches.4.6.weight") == "pts_bbox_head.cls_branches.0.6.weight"
    assert weights.canonical_key("img_backbone.stem.stem_1/norm.num_batches_tracked") is None
    assert "img_backbone.stem.stem_1/conv.weight" in spec   # '/'-containing names are kept


@pytest.mark.parametrize("name", ["far3d_small_seq", "far3d_c1_seq", "far3d_overflow_seq"])
def test_oracle_reproduces_reference_golden_sequence(name):
    """far3d_small_seq: 2 cameras, 4 frames with ego motion and a scene change at frame 2; far3d_c1_seq: the single-camera
    256x256 case of BASELINE.json configs[0]; far3d_overflow_seq: 7 frames of one scene into a 48-slot memory queue (16 pushed per
    frame: it overflows from frame 3 on, farhead.py:467-471).  All were produced by the reference's own files (tools/gen_golden.py)."""
    z, rc = load_small(name)
    orc, _ = small_oracle(rc)
    with torch.no_grad():
        for fi in range(rc["frames"]):
            data, metas = synth.recipe_frame(rc, fi)
            o = orc.simple_test(data, metas)
            for key, got in (("all_cls_scores", o["all_cls_scores"]), ("all_bbox_preds", o["all_bbox_preds"]),
                             ("bbox2d", torch.cat(o["roi"]["bbox_list"])), ("bbox2d_scores", o["roi"]["bbox2d_scores"])):
                want = torch.from_numpy(z["f%d_%s" % (fi, key)])
                assert got.shape == want.shape, (fi, key)
                scale = max(1.0, want.abs().max().item())   # box coordinates are metres / pixels -> relative tolerance
                assert (got - want).abs().max().item() < 2e-4 * scale, (fi, key)   # north_star logit tolerance is 1e-3
            assert_detections_match(tuple(o["result"][k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                    tuple(z["f%d_%s" % (fi, k)] for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)
            assert np.array_equal(o["roi"]["valid_indices"].nonzero().numpy(), z["f%d_valid_idx" % fi])
            assert np.array_equal(o["roi"]["pred_depth"].argmax(1).numpy(), z["f%d_depth_argmax" % fi])


def test_closed_form_known_answers():
    orc = far3d_oracle.Far3DOracle({}, far3d_oracle.default_cfg())
    # depth bin 0 -> depth_min (farhead.py:521-527); bins are monotone and end near depth_max
    d = orc._bin_to_depth(torch.arange(51, dtype=torch.float32))
    assert abs(d[0].item() - 0.1) < 1e-6 and (d[1:] > d[:-1]).all() and abs(d[50].item() - 110.0) < 1e-3
    e = far3d_oracle.pos2posemb3d(torch.zeros(1, 3))
    assert e.shape == (1, 384) and torch.equal(e[0, 0::2], torch.zeros(192)) and torch.equal(e[0, 1::2], torch.ones(192))
    x = torch.tensor([0.0, 0.25, 1.0])
    assert torch.allclose(far3d_oracle.inverse_sigmoid(x).sigmoid(), torch.tensor([1e-5, 0.25, 1 - 1e-5]), atol=1e-6)


def test_forced_depth_hook_replaces_exactly_the_argmax_it_is_given():
    """The test-rig hook for near-tie depth bins (Far3DOracle.simple_test(forced_depth=...)): handing the oracle its own argmax changes
    nothing; moving the bin of every cell moves the adaptive queries' reference points and with them the logits -- the hook reaches
    the proposal construction, it is not a no-op."""
    z, rc = load_small("far3d_small_seq")
    seen = {}

    def same(pred, idx):
        seen["shape"] = (tuple(pred.shape), tuple(idx.shape))
        return idx

    def shifted(pred, idx):
        return (idx + 7) % pred.shape[1]

    outs = []
    for hook in (None, same, shifted):
        orc, _ = small_oracle(rc)
        with torch.no_grad():
            data, metas = synth.recipe_frame(rc, 0)
            outs.append(orc.simple_test(data, metas, forced_depth=hook))
    base, ident, moved = outs
    assert seen["shape"][0][0] == seen["shape"][1][0] and seen["shape"][1][-1] == 1 and seen["shape"][0][2:] == seen["shape"][1][1:3]
    assert torch.equal(ident["all_cls_scores"], base["all_cls_scores"]) and torch.equal(ident["reference_points"], base["reference_points"])
    assert base["roi"]["valid_indices"].any(), "the recipe frame has proposals"
    assert not torch.equal(moved["reference_points"], base["reference_points"])
    assert (moved["all_cls_scores"] - base["all_cls_scores"]).abs().max().item() > 1e-4
