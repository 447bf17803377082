"""GPU: the BENCHMARKED configuration against the oracle.

BASELINE.json configs[1]: 7 cameras x 3x640x960, VoV-99, A = 644 learned + 7x92 adaptive (static top-K proposals) + 256
propagated = 1544 queries, 2312 self-attention keys, streaming memory, 3 frames with ego motion.  The oracle (CPU restatement,
pinned to the reference's files by tools/gen_golden.py) runs the same seeded weights / inputs on the host cores.

  (a) fp32 engine vs oracle: logits within the north-star 1e-3, same detections;
  (b) hipGraph replay == eager, BITWISE, in both precisions (incl. a scene change while the graph exists);
  (c) bf16 engine: max / mean logit error vs the oracle measured per stage and recorded (gpurun_out/parity_full.json).
"""
import json
import os

import numpy as np
import pytest
import torch

from far3d_amd import synth, weights
from tests.conftest import ROOT, assert_detections_match

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K = 92
FRAMES = 3


def _frames(device="cpu"):
    return [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=device, ego_motion=True) for fi in range(FRAMES)]


@pytest.fixture(scope="module")
def oracle_run():
    """3 streaming frames through the oracle (~15-25 s per frame on the box's host cores)."""
    from oracle import far3d_oracle
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    sd = weights.init_state_dict(weights.detector_spec("V-99-eSE"), seed=0)
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(proposal_topk=K))
    outs = []
    with torch.no_grad():
        for data, metas in _frames():
            o = orc.simple_test(data, metas)
            outs.append(dict(logits=o["all_cls_scores"].clone(), boxes=o["all_bbox_preds"].clone(), outs_dec=o["outs_dec"].clone(),
                             feat_flatten=o["feat_flatten"].clone(), fpn=[f.clone() for f in o["feat_levels"]],
                             valid=o["roi"]["valid_indices"].clone(), ref=o["reference_points"].clone(),
                             result={k: v.clone() for k, v in o["result"].items()}))
    return sd, outs


def _engine(sd, precision, use_graph=False):
    from far3d_amd import engine
    eng = engine.Far3DEngine(sd, engine.default_cfg(proposal_topk=K), device=DEV, precision=precision)
    eng.use_graph = use_graph
    return eng


def _stage_errors(o, want):
    """max-abs / mean-abs error per stage, relative to the stage's max magnitude (absolute for the logits)."""
    rep = {}
    for l in range(4):
        g = o["fpn"][l].float().permute(0, 3, 1, 2).cpu()
        d = (g - want["fpn"][l]).abs()
        rep["fpn%d_rel_max" % l] = d.max().item() / want["fpn"][l].abs().max().item()
    d = (o["feat_flatten"].float().cpu() - want["feat_flatten"]).abs()
    rep["value_maps_rel_max"] = d.max().item() / want["feat_flatten"].abs().max().item()
    rep["value_maps_rel_mean"] = d.mean().item() / want["feat_flatten"].abs().mean().item()
    for li in range(6):
        d = (o["outs_dec"][li].cpu() - want["outs_dec"][li, 0]).abs()
        rep["dec%d_abs_max" % li] = d.max().item()
    d = (o["all_cls_scores"].cpu() - want["logits"]).abs()
    rep["logit_max_abs"], rep["logit_mean_abs"] = d.max().item(), d.mean().item()
    rep["logit_last_layer_max_abs"] = d[-1].max().item()
    return rep


def _same_proposals(o, want):
    cnt = o["sel_cnt"].cpu().numpy()
    got = sorted((n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy())
    ref = sorted((int(n), int(i)) for n, i, _ in want["valid"].nonzero().numpy())
    return got == ref


def test_fp32_engine_matches_oracle_at_full_size(hip_lib, oracle_run):
    sd, want = oracle_run
    eng = _engine(sd, "fp32")
    for fi, (data, metas) in enumerate(_frames()):
        o = eng.forward_frame(data, metas)
        assert o["all_cls_scores"].shape == want[fi]["logits"].shape == (6, 1, 1544, 26)
        assert _same_proposals(o, want[fi]), "frame %d: static top-%d proposal set differs from the oracle's" % (fi, K)
        e = (o["all_cls_scores"].cpu() - want[fi]["logits"]).abs().max().item()
        assert e < 1e-3, "frame %d: max abs logit error %.3e (north-star tolerance 1e-3)" % (fi, e)
        eb = (o["all_bbox_preds"].cpu() - want[fi]["boxes"]).abs().max().item()
        assert eb < 1e-3 * max(1.0, want[fi]["boxes"].abs().max().item() / 10.0), "frame %d: box error %.3e" % (fi, eb)
        r = o["result"]
        keep = r["keep"].cpu().numpy()
        assert_detections_match(tuple(r[k].cpu().numpy()[keep] for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                tuple(want[fi]["result"][k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graph_replay_is_bitwise_eager(hip_lib, oracle_run, precision):
    """Frames 0..2 of scene A, then a NEW scene (memory reset while the captured graph exists), then 2 more frames: the
    hipGraph engine must reproduce the eager engine bit for bit (deterministic kernels, in-place memory reset)."""
    sd, _ = oracle_run
    seq = _frames(DEV)
    d2, m2 = synth.make_frame(7, (640, 960), seed=1, frame_index=0, device=DEV, ego_motion=True)
    m2 = [dict(m2[0], scene_token="synthetic-scene-1")]
    d3, m3 = synth.make_frame(7, (640, 960), seed=1, frame_index=1, device=DEV, ego_motion=True)
    m3 = [dict(m3[0], scene_token="synthetic-scene-1")]
    seq = seq + [(d2, m2), (d3, m3), (seq[2][0], m3)]
    res = {}
    for mode in ("eager", "graph"):
        eng = _engine(sd, precision, use_graph=(mode == "graph"))
        out = []
        for data, metas in seq:
            o = eng.forward_frame(data, metas)
            out.append((o["all_cls_scores"].clone(), o["all_bbox_preds"].clone(), o["result"]["scores_3d"].clone(),
                        {k: v.clone() for k, v in eng.mem.items()}))
        if mode == "graph":
            assert eng._graph is not None
        res[mode] = out
        del eng
        torch.cuda.empty_cache()
    for fi, (a, b) in enumerate(zip(res["eager"], res["graph"])):
        assert torch.equal(a[0], b[0]), "frame %d: logits differ between hipGraph replay and eager (%s)" % (fi, precision)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), "frame %d: boxes / scores differ" % fi
        for k in a[3]:
            assert torch.equal(a[3][k], b[3][k]), "frame %d: streaming memory '%s' differs" % (fi, k)


@pytest.mark.parametrize("precision", ["bf16", "bf16_fp32dec"])
def test_bf16_error_budget_is_measured_and_bounded(hip_lib, oracle_run, precision):
    """bf16 activations through 60 convolutions cannot meet 1e-3 on logits against an fp32 reference; this measures how
    far it is, stage by stage, and records it (DESIGN.md §4 quotes the file)."""
    sd, want = oracle_run
    eng = _engine(sd, precision)
    report = []
    for fi, (data, metas) in enumerate(_frames()):
        o = eng.forward_frame(data, metas)
        rep = _stage_errors(o, want[fi])
        rep["same_proposals"] = _same_proposals(o, want[fi])
        rep["frame"] = fi
        report.append(rep)
        if not rep["same_proposals"]:
            break          # later frames are no longer comparable query by query
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_full_%s.json" % precision), "w") as f:
        json.dump(report, f, indent=1)
    print("\n%s error budget vs the fp32 oracle (configs[1], frame 0): %s" % (precision, json.dumps(report[0])))
    r0 = report[0]
    assert r0["value_maps_rel_max"] < 0.08 and all(r0["fpn%d_rel_max" % l] < 0.08 for l in range(4))
    assert np.isfinite(r0["logit_max_abs"]) and r0["logit_max_abs"] < 0.5, r0
