"""GPU: the whole frame through the HIP engine vs the golden vectors produced by the REFERENCE's own files
(tests/golden/far3d_small_seq.npz: 2 cameras, 4 streaming frames with ego motion and a scene change;
tests/golden/far3d_c1_seq.npz: BASELINE.json configs[0], one camera at 256x256) -- north_star tolerance 1e-3 on logits in
fp32 mode.  The benchmarked 7 x 640x960 configuration is covered by tests/test_engine_full_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

from far3d_amd import synth, weights
from tests.conftest import ROOT, assert_detections_match

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")


def _golden_engine(precision, name="far3d_small_seq", **over):
    from far3d_amd import engine
    z = np.load(os.path.join(GOLD, name + ".npz"))
    rc = json.loads(bytes(z["recipe"]).decode())
    spec = weights.detector_spec(rc["backbone"], num_query=rc["num_query"], num_propagated=rc["num_propagated"])
    sd = weights.init_state_dict(spec, seed=rc["weight_seed"])
    cfg = engine.default_cfg(backbone=rc["backbone"], num_cams=rc["num_cams"], num_query=rc["num_query"],
                             num_propagated=rc["num_propagated"], memory_len=rc["memory_len"], topk_proposals=rc["topk_proposals"], **over)
    return engine.Far3DEngine(sd, cfg, device=DEV, precision=precision), z, rc


# Every frame of both golden sequences is held to the north-star bar itself (1e-3 on logits).  Round 2 allowed 10x on the streaming
# frame of the single-camera case, citing a 1.2e-2 oracle-vs-reference deviation measured with the ROUND-1 weight initialisation;
# with the reference-like key-point initialisation the goldens were regenerated with tools/gen_golden.py
# measures 5.2e-4 * scale there, and the allowance is gone.  The observed errors are printed and recorded in
# gpurun_out/golden_errors.json.
LOOSE = {}


@pytest.mark.parametrize("name", ["far3d_small_seq", "far3d_c1_seq", "far3d_overflow_seq"])
def test_engine_fp32_matches_reference_golden_sequence(hip_lib, name):
    """far3d_overflow_seq: 7 frames of one scene, memory_len = 3 x num_propagated, 16 entries pushed per frame --
    the queue is full after frame 2, so frames 3-6 run the truncation of pre_update_memory (farhead.py:467-471) on LIVE entries and
    carry timestamps / poses three frames old; every frame is held to the 1e-3 bar against the REFERENCE's own outputs."""
    eng, z, rc = _golden_engine("fp32", name)
    for fi in range(rc["frames"]):
        loose = LOOSE.get((name, fi), 1.0)
        data, metas = synth.recipe_frame(rc, fi)
        if name == "far3d_overflow_seq" and fi >= 3:
            # the queue the frame starts from is full of live entries (none of the zero rows a scene start leaves): what the
            # post-update of this frame pushes drops the oldest topk_proposals of them
            assert int((eng.mem["emb"][0].abs().sum(-1) > 0).sum().item()) == rc["memory_len"], "frame %d: the memory queue is not full" % fi
        o = eng.forward_frame(data, metas)
        # 2D proposals: same peaks, same order
        want_idx = z["f%d_valid_idx" % fi]          # rows (camera, flat index, 0)
        cnt = o["sel_cnt"].cpu().numpy()
        got = [(n, int(i)) for n in range(rc["num_cams"]) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()]
        assert got == [(int(r[0]), int(r[1])) for r in want_idx], "frame %d: proposal set differs" % fi
        # w,h = exp(pred)*stride can be huge with random weights: relative tolerance (only the centre feeds the 3D head)
        assert np.allclose(o["bbox2d"].cpu().numpy(), z["f%d_bbox2d" % fi], rtol=2e-3, atol=2e-3)
        assert np.abs(o["bbox2d_scores"].cpu().numpy() - z["f%d_bbox2d_scores" % fi][:, 0]).max() < 1e-4
        for key in ("all_cls_scores", "all_bbox_preds"):
            want = z["f%d_%s" % (fi, key)]
            g = o[key].cpu().numpy()
            assert g.shape == want.shape, (fi, key, g.shape, want.shape)
            # logits: the north_star's absolute 1e-3.  Box codes: the centre is sigmoid(reg logit) * 304.8 m, so 1e-3 on the
            # reg logit is 1e-3 * 304.8 / 4 = 0.076 m; the other channels (log sizes, sin, cos) are raw outputs: 1e-3.
            err = np.abs(g - want)
            print("%s frame %d %s: max abs err %.3e (centre %.3e)" % (name, fi, key, err.max() if key == "all_cls_scores" else err[..., 3:].max(),
                                                                     err[..., :3].max()))
            if key == "all_cls_scores":
                assert err.max() < 1e-3 * loose, "frame %d logits: max abs err %.3e" % (fi, err.max())
            else:
                assert err[..., :3].max() < 0.076 * loose and err[..., 3:].max() < 1e-3 * loose, \
                    "frame %d boxes: centre err %.3e m, code err %.3e" % (fi, err[..., :3].max(), err[..., 3:].max())
        r = o["result"]
        keep = r["keep"].cpu().numpy()
        assert_detections_match(tuple(r[k].cpu().numpy()[keep] for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                tuple(z["f%d_%s" % (fi, k)] for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi,
                                score_tol=1e-3 * loose, box_tol=2e-2 * loose * loose)


def test_engine_bf16_deviation_is_bounded_and_reported(hip_lib):
    """bf16 activations through 60 convs cannot meet 1e-3 on logits; this pins the measured deviation of the maps at toy
    size (tests/test_engine_full_gpu.py measures the logits at the benchmarked size)."""
    eng, z, rc = _golden_engine("bf16")
    worst = 0.0
    for fi in range(1):   # later frames depend on data-dependent proposal sets that bf16 may flip
        data, metas = synth.recipe_frame(rc, fi)
        o = eng.forward_frame(data, metas)
        for l in range(4):
            want = z["f%d_fpn%d_sample" % (fi, l)]                       # (N, C/16, h/2, w/3) NCHW sample
            g = o["fpn"][l].float().permute(0, 3, 1, 2)[:, ::16, ::2, ::3].cpu().numpy()
            rel = np.abs(g - want).max() / max(1e-6, np.abs(want).max())
            worst = max(worst, rel)
            assert rel < 0.08, "fpn level %d relative deviation %.3f" % (l, rel)
    print("bf16 backbone+FPN max relative deviation vs fp32 reference: %.4f" % worst)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_engine_graph_mode_with_scene_change_is_bitwise_eager(hip_lib, precision):
    """Single-GPU hipGraph mode (static top-K proposals) over the golden sequence's scene change: the first frame of each
    scene runs eagerly and resets the streaming memory IN PLACE, so the graph captured in scene 0 stays valid in scene 1
    (the old engine re-allocated the memory and replayed on freed buffers)."""
    res = {}
    for mode in ("eager", "graph", "pipeline"):
        eng, z, rc = _golden_engine(precision, proposal_topk=12)
        eng.use_graph = mode != "eager"
        eng.pipeline = mode == "pipeline"
        out = []
        for fi in list(range(rc["frames"])) + [3] * 5:     # extra steady frames of scene 1: every pipeline buffer set captures AND replays
            data, metas = synth.recipe_frame(rc, fi)
            o = eng.forward_frame(data, metas)
            eng.wait_outputs()                             # pipeline mode: outputs are produced on the head stream
            out.append((o["all_cls_scores"].clone(), o["all_bbox_preds"].clone(), {k: v.clone() for k, v in eng.mem.items()}))
        if mode == "graph":
            assert eng._graph is not None
        if mode == "pipeline":
            assert sorted(eng._pipe["g_head"]) == list(range(eng.pipeline_sets))    # every buffer set captured its camera / head graphs
        res[mode] = out
    for mode in ("graph", "pipeline"):
        for fi, (a, b) in enumerate(zip(res["eager"], res[mode])):
            assert torch.isfinite(a[0]).all()
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "frame %d: %s differs from eager" % (fi, mode)
            for k in a[2]:
                assert torch.equal(a[2][k], b[2][k]), "frame %d (%s): streaming memory '%s' differs" % (fi, mode, k)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("seq", [[0, 1, 1, 1, 2, 3, 3, 3, 3], [0, 1, 2, 3, 3]])
def test_pipelined_frames_in_flight_end_in_the_same_state(hip_lib, precision, seq):
    """Pipeline mode with frames submitted back to back and NO wait in between (camera stages of frame i+1 really overlap the
    head of frame i; a scene change in the middle): the last frame's outputs and the streaming memory equal the eager engine's.
    The short sequence is ADVICE r2's case: scene A, A, B -- the scene start arrives while the FIRST steady frame (whose buffer set
    was allocated inside the pipelined call) still has its head in flight; the eager scene start must wait for it."""
    fin = {}
    for mode in ("eager", "pipeline"):
        eng, z, rc = _golden_engine(precision, proposal_topk=12)
        eng.use_graph = eng.pipeline = mode == "pipeline"
        frames = [synth.recipe_frame(rc, fi, device=DEV) for fi in range(rc["frames"])]
        for fi in seq:
            o = eng.forward_frame(*frames[fi])
        torch.cuda.synchronize()
        fin[mode] = (o["all_cls_scores"].clone(), o["all_bbox_preds"].clone(), {k: v.clone() for k, v in eng.mem.items()},
                     {k: v.clone() for k, v in o["result"].items()})
    a, b = fin["eager"], fin["pipeline"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_threshold_mode_grows_the_proposal_capacity(hip_lib):
    """Reference mode keeps EVERY peak above score_thr: a capacity smaller than a camera's peak count must not drop any."""
    eng, z, rc = _golden_engine("fp32", proposal_cap=4)
    data, metas = synth.recipe_frame(rc, 0)
    o = eng.forward_frame(data, metas)
    want_idx = z["f0_valid_idx"]
    cnt = o["sel_cnt"].cpu().numpy()
    got = [(n, int(i)) for n in range(rc["num_cams"]) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()]
    assert got == [(int(r[0]), int(r[1])) for r in want_idx]
    assert np.abs(o["all_cls_scores"].cpu().numpy() - z["f0_all_cls_scores"]).max() < 1e-3
