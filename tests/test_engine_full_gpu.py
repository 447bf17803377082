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
import torch.nn.functional as F  # noqa: F401

from far3d_amd import synth, weights
from tests.conftest import ROOT, assert_detections_match

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K = 92
FRAMES = 3
FRAMES_QUEUE = 5      # frames of the un-adopted witness: the 1024-slot memory queue (4 x 256, far3d.py:80-82) is full after frame 3 and
                      # frame 4 drops live entries (farhead.py:467-471) -- BASELINE configs[4]'s "4-frame memory queue"
# The per-camera stages of the oracle (backbone, FPN, 2D head: ~90 % of a frame's CPU time) do not depend on the streaming state or on
# any adopted decision: computed once per (frame, dtype) for the whole module and re-used by every oracle run of these weights
# (the suite's wall time is mostly full-size oracle frames).
_CAMERA_CACHE = {}


def _frames(device="cpu", n=FRAMES):
    return [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=device, ego_motion=True) for fi in range(n)]


LEVEL_HW = [(80, 120), (40, 60), (20, 30), (10, 15)]      # 640x960 at strides 8, 16, 32, 64


def _neighbour_max(raw):
    """(N,S) -> (N,S): the largest score among the 8 neighbours of every cell (per pyramid level, borders clipped)."""
    out, st = [], 0
    for h, w in LEVEL_HW:
        seg = F.pad(raw[:, st:st + h * w].view(-1, h, w), (1, 1, 1, 1), value=-1.0)
        shifts = [seg[:, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]
        out.append(torch.stack(shifts).max(dim=0).values.reshape(-1, h * w))
        st += h * w
    return torch.cat(out, dim=1)


def _resolve_topk_ties(sel, tag, tol=2e-5, dev_scores=None):
    """The static top-K proposal mode keeps the K best 2D peaks per camera (peak: score == maxpool3x3(score)); both the peak test
    and the K-th place are decisions on near-equal numbers that rounding noise can turn either way -- and one flipped peak test
    shifts the K-th place by a whole rank.  Returns a callback for the oracle: it checks that the device's set D is one the
    oracle's own scores allow when relative differences below `tol` may go either way, and then selects exactly D so that the
    frames stay comparable row by row.  In the oracle's scores a cell is a ROBUST peak if it exceeds its 8 neighbours by more
    than tol, a NON-peak if it is below their maximum by more than tol, FRAGILE otherwise.  D is allowed iff
      * |D| = K and D contains no non-peak;
      * every robust peak outside D scores no higher than the lowest score in D (up to tol).
    dev_scores (M,), the device's 2D score of every selected cell in row order (camera-major, ascending cell): a selected cell whose
    peak test FAILED on the device is a zero-weight filler there (score 0 -> the context's log-odds term sits at its clamp, 14 below
    a real score: found in round 4 as the single 0.19-off query of the pair-stored mode).  The oracle must count the same cells as
    fillers: allowed iff such a cell is not a robust peak, and a cell the device scores is not a non-peak (checked above)."""
    def pick(own):
        raw = own["raw_weight"][..., 0]                      # (N,S) score before the 3x3 peak test
        N = raw.shape[0]
        assert raw.shape[1] == sum(h * w for h, w in LEVEL_HW)
        nb = _neighbour_max(raw)
        robust, nonpeak = raw > (1 + tol) * nb, raw < (1 - tol) * nb
        mask = torch.zeros_like(own["valid_indices"])
        for n, i in sel:
            mask[n, i, 0] = True
        D = mask[..., 0]
        for n in range(N):
            assert int(D[n].sum()) == int(own["valid_indices"][n].sum()), "%s camera %d: %d peaks" % (tag, n, int(D[n].sum()))
            bad = (D[n] & nonpeak[n]).nonzero().flatten().tolist()
            assert not bad, "%s camera %d: the device selected cells %s that are not peaks (beyond a relative %.0e)" % (tag, n, bad, tol)
            low = raw[n][D[n]].min().item()
            miss = (robust[n] & ~D[n] & (raw[n] > (1 + tol) * low)).nonzero().flatten().tolist()
            assert not miss, "%s camera %d: the device skipped robust peaks %s that outscore its lowest pick %.6f" % (tag, n, miss, low)
        if dev_scores is None:
            return mask
        peak = mask.clone()
        cells = D.nonzero()                                   # row order of the adaptive queries: camera-major, ascending cell
        assert cells.shape[0] == dev_scores.numel(), (cells.shape, dev_scores.shape)
        filler = dev_scores.flatten() <= 0
        for (n, i) in cells[filler].tolist():
            assert not bool(robust[n, i]), "%s camera %d: the device weighs cell %d zero although it is a robust peak" % (tag, n, i)
            peak[n, i, 0] = False
        return mask, peak
    return pick


def _resolve_memory_ties(idx_dev, tag, tol=1e-2):
    """post_update_memory keeps the 256 best-scoring queries (farhead.py:488-491), a discrete decision on scores that sit
    ~1e-4 apart at the cut: rounding noise decides which side a few of them fall on, and with a different memory the next frame
    is a different computation (the oracle in fp32 and in fp64 disagree by O(1) on whole rows after one such flip).  The callback
    checks that the device's selection differs from the oracle's own only by such near-ties (scores within `tol` of the cut) and
    adopts the device's selection, order included, so that the following frames stay comparable."""
    def pick(score, own):
        dev = torch.as_tensor(idx_dev, dtype=torch.long)
        cut = score[own].min()
        only = torch.tensor(sorted(set(dev.tolist()) ^ set(own.tolist())), dtype=torch.long)
        if only.numel():
            gap = (score[only] - cut).abs().max().item()
            assert gap < tol, "%s: memory top-k sets differ beyond near-ties at the cut (score gap %.2e, %d elements)" % (tag, gap, only.numel())
        # same elements in a different order: only neighbours with near-equal scores may swap
        assert (score[dev][:-1] - score[dev][1:]).min().item() > -tol, "%s: device top-k order is not descending in the oracle's scores" % tag
        return dev
    return pick


def _resolve_depth_ties(depth_logit_dev, tag, tol=1e-3):
    """The adaptive queries take their depth from the argmax over the 51 depth bins at the proposal's cell (farhead.py:736-766,
    topk = 1): a discrete choice per cell that rounding noise flips where the two best bins are near-equal -- and a flipped bin
    replaces a whole query (it bit in round 3: one query 0.19 off with every continuous quantity within 3e-5).  The callback
    makes the oracle adopt the device's per-cell argmax (first maximum of the device's own logits, what far3d_proposal_gather
    computes) after checking that every cell where the two differ IS a near-tie in the oracle's own probabilities: the device's
    bin within a relative `tol` of the oracle's best."""
    dev = depth_logit_dev.argmax(dim=-1)                          # (N, h, w) from the NHWC logits
    def pick(pred_depth, own):                                    # (BN, D, h, w) softmax, (BN, h, w, 1)
        d = dev.view(own.shape).to(own.device)
        diff = d != own
        if diff.any():
            p = pred_depth.permute(0, 2, 3, 1)
            p_own, p_dev = p.gather(-1, own)[diff], p.gather(-1, d)[diff]
            gap = ((p_own - p_dev) / p_own).max().item()
            assert gap < tol, "%s: %d depth-bin choices differ beyond near-ties (relative probability gap %.2e)" % (tag, int(diff.sum()), gap)
        return d
    return pick


def _run_engine(sd, precision, frames=FRAMES, light=False):
    """`frames` streaming frames through the engine; everything the comparisons need, on the host (light: logits and decisions only)."""
    eng = _engine(sd, precision)
    got = []
    for data, metas in _frames(n=frames):
        live = int((eng.mem["emb"][0].abs().sum(-1) > 0).sum().item()) if eng._mem_valid else 0     # live memory slots the frame starts from
        o = eng.forward_frame(data, metas)
        cnt = o["sel_cnt"].cpu().numpy()
        if light:
            got.append(dict(sel=[(n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()], memory_live=live,
                            memory_topk=o["memory_topk"].cpu().clone(), depth_logit=o["depth_logit"].float().cpu().clone(),
                            score2d=o["bbox2d_scores"].float().cpu().clone(), all_cls_scores=o["all_cls_scores"].cpu().clone()))
            continue
        got.append(dict(sel=[(n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()], memory_live=live,
                        memory_topk=o["memory_topk"].cpu().clone(), depth_logit=o["depth_logit"].float().cpu().clone(),
                        ref=o["reference_points"].float().cpu().clone(), bbox2d=o["bbox2d"].float().cpu().clone(),
                        score2d=o["bbox2d_scores"].float().cpu().clone(),
                        all_cls_scores=o["all_cls_scores"].cpu().clone(), all_bbox_preds=o["all_bbox_preds"].cpu().clone(),
                        outs_dec=o["outs_dec"].cpu().clone(), feat_flatten=o["feat_flatten"].float().cpu().clone(),
                        fpn=[eng.act_to_nchw(f).permute(0, 2, 3, 1).cpu().clone() for f in o["fpn"]],   # NHWC f32 (pair storage decoded)
                        result={k: v.cpu().clone() for k, v in o["result"].items()}))
    del eng
    torch.cuda.empty_cache()
    return got


def _run_oracle(sd, got, with_f64, tie_tol=2e-5):
    """The oracle on the same frames, adopting the device's near-tie decisions (checked to BE near-ties: 2D scores within a
    relative `tie_tol`, the rounding-noise level of the mode under test).  As many frames as `got` holds."""
    from oracle import far3d_oracle
    keep = lambda o: dict(logits=o["all_cls_scores"].clone(), boxes=o["all_bbox_preds"].clone(), outs_dec=o["outs_dec"].clone(),
                          feat_flatten=o["feat_flatten"].clone(), fpn=[f.clone() for f in o["feat_levels"]],
                          valid=o["roi"]["valid_indices"].clone(), ref=o["reference_points"].clone(),
                          result={k: v.clone() for k, v in o["result"].items()})
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(proposal_topk=K))
    o64 = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(proposal_topk=K), dtype=torch.float64) if with_f64 else None
    outs = []
    with torch.no_grad():
        for fi, (data, metas) in enumerate(_frames(n=len(got))):
            w = keep(orc.simple_test(data, metas, forced_valid=_resolve_topk_ties(got[fi]["sel"], "frame %d" % fi, tie_tol, got[fi]["score2d"]),
                                     forced_topk=_resolve_memory_ties(got[fi]["memory_topk"], "frame %d" % fi),
                                     forced_depth=_resolve_depth_ties(got[fi]["depth_logit"], "frame %d" % fi, max(1e-4, 20 * tie_tol)),
                                     camera_cache=(_CAMERA_CACHE, (fi, "f32"))))
            if with_f64:
                d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
                w["f64"] = keep(o64.simple_test(d64, metas, forced_valid=_resolve_topk_ties(got[fi]["sel"], "frame %d (fp64)" % fi, tie_tol, got[fi]["score2d"]),
                                                forced_topk=_resolve_memory_ties(got[fi]["memory_topk"], "frame %d (fp64)" % fi),
                                                forced_depth=_resolve_depth_ties(got[fi]["depth_logit"], "frame %d (fp64)" % fi, max(1e-4, 20 * tie_tol)),
                                                camera_cache=(_CAMERA_CACHE, (fi, "f64"))))
            outs.append(w)
    return outs


@pytest.fixture(scope="module")
def oracle_run(hip_lib):
    """The fp32 engine on 3 streaming frames, then the oracle in fp32 AND fp64 on the same frames (~15-25 s per fp32 frame on
    the box's host cores).  The fp64 run is the yardstick for the fp32 rounding noise of the reference arithmetic itself."""
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    sd = weights.init_state_dict(weights.detector_spec("V-99-eSE"), seed=0)
    got = _run_engine(sd, "fp32")
    return sd, _run_oracle(sd, got, with_f64=True), got


def _engine(sd, precision, use_graph=False):
    from far3d_amd import engine
    eng = engine.Far3DEngine(sd, engine.default_cfg(proposal_topk=K), device=DEV, precision=precision)
    eng.use_graph = use_graph
    return eng


def _engine_sel(o):
    cnt = o["sel_cnt"].cpu().numpy()
    return [(n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()]


def _matched_rows(sel, want):
    """Query rows comparable one to one: the 644 learned queries, the adaptive queries whose 2D peak (camera, cell) both sides
    selected, and the 256 propagated ones.  Returns (engine rows, oracle rows, number of common proposals)."""
    ws = [(int(n), int(i)) for n, i, _ in want["valid"].nonzero().numpy()]
    pos = {k: j for j, k in enumerate(ws)}
    nq, A = 644, want["logits"].shape[2]
    rg, rw = list(range(nq)), list(range(nq))
    common = 0
    for j, k in enumerate(sel):
        if k in pos:
            rg.append(nq + j); rw.append(nq + pos[k]); common += 1
    rg += list(range(nq + len(sel), A)); rw += list(range(nq + len(ws), A))
    return rg, rw, common


def _stage_errors(o, want, rows=None):
    """max-abs / mean-abs error per stage, relative to the stage's max magnitude (absolute for decoder states and logits).
    rows: optional (engine rows, oracle rows) when the two sides do not hold the same adaptive queries."""
    rep = {}
    for l in range(4):
        g = o["fpn"][l].float().permute(0, 3, 1, 2).cpu()     # NHWC (device or host) -> NCHW
        d = (g - want["fpn"][l]).abs()
        rep["fpn%d_rel_max" % l] = d.max().item() / want["fpn"][l].abs().max().item()
    d = (o["feat_flatten"].float().cpu() - want["feat_flatten"]).abs()
    rep["value_maps_rel_max"] = d.max().item() / want["feat_flatten"].abs().max().item()
    rep["value_maps_rel_mean"] = d.mean().item() / want["feat_flatten"].abs().mean().item()
    rg, rw = rows if rows is not None else (slice(None), slice(None))
    for li in range(6):
        d = (o["outs_dec"][li].cpu()[rg] - want["outs_dec"][li, 0][rw]).abs()
        rep["dec%d_abs_max" % li], rep["dec%d_abs_mean" % li] = d.max().item(), d.mean().item()
    d = (o["all_cls_scores"].cpu()[:, 0][:, rg] - want["logits"][:, 0][:, rw]).abs()
    rep["logit_max_abs"], rep["logit_mean_abs"] = d.max().item(), d.mean().item()
    rep["logit_p50_abs"], rep["logit_p99_abs"] = torch.quantile(d.flatten().double(), 0.5).item(), torch.quantile(d.flatten()[::4].double(), 0.99).item()
    rep["logit_last_layer_max_abs"] = d[-1].max().item()
    return rep


def _same_proposals(o, want):
    cnt = o["sel_cnt"].cpu().numpy()
    got = sorted((n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy())
    ref = sorted((int(n), int(i)) for n, i, _ in want["valid"].nonzero().numpy())
    return got == ref


def _pct(d):
    q = torch.quantile(d.flatten()[:: max(1, d.numel() // 2000000)].double(), torch.tensor([0.5, 0.99, 0.999], dtype=torch.float64))
    return dict(p50=q[0].item(), p99=q[1].item(), p999=q[2].item(), max=d.max().item())


def test_fp32_engine_matches_oracle_at_full_size(hip_lib, oracle_run):
    """North-star tolerance 1e-3 on logits, at the benchmarked size, 3 streaming frames with ego motion.

    With key-point offsets initialised like the reference does (weights.init_state_dict) every logit of all three frames is
    within 1e-3 (asserted below).  The test additionally reports a yardstick that explains what happens when the computation is
    conditioned worse (e.g. 3x wider key-point clouds, the round-1 weights: key points that project next to a camera plane --
    the reference has no behind-camera mask, detr3d_transformer.py:550 -- turn 1e-7 input noise into O(1) sampling changes and
    the oracle run in fp32 and in fp64 then differs by ~2e-3 itself): errors are also bounded relative to the oracle's own
    fp32-vs-fp64 deviation:
      * the discrete decisions of the path (K-th 2D peak per camera, top-256 memory selection) are near-ties at rounding-noise
        level; the fixture lets the oracle adopt the device's choice after checking that it IS a near-tie (without that, frames
        1+ of the oracle in fp32 and in fp64 already disagree by O(1) on whole rows);
      * 99.9 % of all logits within 1e-3 (observed: p99.9 ~ 3e-4, median ~ 2e-6);
      * every logit whose own fp32-vs-fp64 oracle noise is below 2.5e-5 (the well-conditioned ones, > 99 %) within 1e-3;
      * the worst logit within 10x the oracle's own worst fp32-vs-fp64 deviation;
      * FPN / value maps within 2e-5 relative; same detections;
      * frames 1-2 (streaming): every error percentile within 6x the oracle's own fp32-vs-fp64 percentile."""
    sd, want, got = oracle_run
    report = []
    for fi in range(FRAMES):
        o, w, w64 = got[fi], want[fi], want[fi]["f64"]
        assert o["all_cls_scores"].shape == w["logits"].shape == (6, 1, 1544, 26)
        rep = _stage_errors(o, w)
        rep["frame"] = fi
        err = (o["all_cls_scores"] - w["logits"]).abs()
        noise = (w["logits"].double() - w64["logits"]).abs()
        rep["logit_abs_err_vs_oracle32"] = _pct(err)
        rep["logit_abs_err_vs_oracle64"] = _pct((o["all_cls_scores"].double() - w64["logits"]).abs())
        rep["oracle32_vs_oracle64"] = _pct(noise)
        well = noise < 2.5e-5
        rep["well_conditioned_fraction"] = well.float().mean().item()
        rep["well_conditioned_max_err"] = err[well].max().item()
        for li in range(6):
            rep["oracle32_vs_oracle64_dec%d_abs_max" % li] = (w["outs_dec"][li].double() - w64["outs_dec"][li]).abs().max().item()
        report.append(rep)
        print("\nfp32 engine vs oracle, frame %d: %s" % (fi, json.dumps(rep)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_full_fp32.json"), "w") as f:
        json.dump(report, f, indent=1)
    for fi, rep in enumerate(report):
        assert rep["value_maps_rel_max"] < 2e-5 and all(rep["fpn%d_rel_max" % l] < 2e-5 for l in range(4)), rep
        e, nz = rep["logit_abs_err_vs_oracle32"], rep["oracle32_vs_oracle64"]
        # the north-star bar itself, on every frame of the streaming sequence (observed max: 4e-5 / 3e-4 / 3e-4)
        assert e["max"] < 1e-3, "frame %d: max abs logit error %.3e (north-star tolerance 1e-3); %s" % (fi, e["max"], e)
        if fi == 0:
            assert e["p999"] < 1e-3, (fi, e)
            assert rep["well_conditioned_fraction"] > 0.99 and rep["well_conditioned_max_err"] < 1e-3, (fi, rep["well_conditioned_fraction"], rep["well_conditioned_max_err"])
            assert e["max"] < 10 * max(nz["max"], 1e-4), (fi, e, nz)
        else:
            # streaming frames: the propagated queries feed the rounding noise of the previous frame back in, and the oracle's own
            # fp32-vs-fp64 deviation grows ~50x (p99 ~2e-3 at frame 1).  The engine must stay within a small multiple of that
            # yardstick at every percentile (observed: 2.5-3.5x; its fp32 GEMMs accumulate K serially, torch's CPU GEMMs blockwise)
            for q in ("p50", "p99", "p999"):
                assert e[q] < 6 * nz[q] + 1e-5, (fi, q, e, nz)
            assert e["max"] < 15 * nz["max"], (fi, e, nz)
        r = got[fi]["result"]
        keep = r["keep"].numpy()
        # detections: same set above the top-k bar; a detection riding on an ill-conditioned logit may move by the noise above
        tol = 10 * max(nz["p999"], 1e-3)
        assert_detections_match(tuple(r[k].numpy()[keep] for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                tuple(want[fi]["result"][k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi,
                                score_tol=tol, box_tol=100 * tol)


@pytest.mark.parametrize("precision", ["bf16x3"] + (["bf16x3_all", "bf16x3_2d1"] if os.environ.get("FAR3D_TEST_ALL_MODES") else []))
def test_split_bf16_modes_against_the_logit_tolerance(hip_lib, oracle_run, precision):
    """The split-bf16 modes (fp32 data; conv products -- and in bf16x3_all also the decoder GEMMs -- as hi*hi' + hi*lo' + lo*hi'
    on the bf16 MFMA with fp32 accumulation; far3d_hip.h FAR3D_DT_F32_BF16X3) against the oracle at the benchmarked size.

    What holds (asserted): on a single frame (frame 0, what the north star describes) EVERY logit is within 2e-4, five times
    inside the 1e-3 bar, and on the streaming frames -- where the propagated queries feed each frame's rounding noise into the
    next one and the reference arithmetic itself (oracle fp32 vs fp64) disagrees by 4.6e-4 -- EVERY logit is within the 1e-3 bar
    too (observed 3.0e-4 / 7.6e-4).  Rounds 2-3 allowed 4e-3 on the worst streaming logit: that slack was an artifact of the rig,
    not of the arithmetic -- one selected 2D cell whose 3x3 peak test (an equality on near-equal scores) came out differently on
    the two sides, so one side carried a zero-weight filler where the other carried a scored proposal (the context's log-odds
    term 14 apart); the oracle now adopts the device's outcome like the other near-tie decisions (_resolve_topk_ties).  Where the
    decisions (K-th 2D peak, peak test, depth-bin argmax, memory top-256 cut) coincide with the fp32 engine's, the fixture's oracle
    run is reused; otherwise the oracle is run again on this engine's decisions."""
    sd, want, got32 = oracle_run
    got = _run_engine(sd, precision)
    same = all(sorted(got[fi]["sel"]) == sorted(got32[fi]["sel"]) and torch.equal(got[fi]["memory_topk"], got32[fi]["memory_topk"]) and
               torch.equal(got[fi]["depth_logit"].argmax(-1), got32[fi]["depth_logit"].argmax(-1)) and
               torch.equal(got[fi]["score2d"] > 0, got32[fi]["score2d"] > 0) for fi in range(FRAMES))
    if not same:
        want = _run_oracle(sd, got, with_f64=False, tie_tol=5e-4)     # 2D scores carry ~2e-5 mean / 2e-4 max relative noise in this mode
    report = []
    for fi in range(FRAMES):
        o, w = got[fi], want[fi]
        rep = _stage_errors(o, w)
        rep.update(frame=fi, precision=precision, decisions_equal_fp32_engine=same,
                   logit_abs_err_vs_oracle32=_pct((o["all_cls_scores"] - w["logits"]).abs()),
                   logit_abs_diff_vs_fp32_engine=_pct((o["all_cls_scores"] - got32[fi]["all_cls_scores"]).abs()) if same else None)
        report.append(rep)
        print("\n%s engine vs oracle, frame %d: %s" % (precision, fi, json.dumps(rep)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_full_%s.json" % precision), "w") as f:
        json.dump(report, f, indent=1)
    for fi in range(FRAMES):      # where do the worst rows come from?  (a whole-query replacement is the signature of a discrete decision)
        o, w = got[fi], want[fi]
        d0 = (o["outs_dec"][0] - w["outs_dec"][0, 0]).abs().max(dim=-1).values          # (A,) first decoder layer
        worst = torch.topk(d0, 5).indices.tolist()
        dref = (o["ref"].view(-1, 3) - w["ref"].view(-1, 3)).abs().max(dim=-1).values
        print("\n%s frame %d: worst rows of decoder layer 0 %s (644..1287 adaptive, 1288.. propagated); their errors %s; their "
              "reference-point differences %s; max reference-point difference overall %.3e at row %d" %
              (precision, fi, worst, [round(d0[i].item(), 5) for i in worst], [round(dref[i].item(), 6) for i in worst], dref.max().item(), int(dref.argmax())))
    for fi, rep in enumerate(report):
        assert rep["value_maps_rel_max"] < 2e-4 and all(rep["fpn%d_rel_max" % l] < 2e-4 for l in range(4)), rep
        e = rep["logit_abs_err_vs_oracle32"]
        if fi == 0:
            assert e["max"] < 2e-4, "frame 0: max abs logit error %.3e; %s" % (e["max"], e)
        else:
            assert e["p999"] < 1e-3 and e["max"] < 1e-3, "frame %d: %s" % (fi, e)      # the north-star bar itself, streaming frames included
        r = got[fi]["result"]
        keep = r["keep"].numpy()
        assert_detections_match(tuple(r[k].numpy()[keep] for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                tuple(want[fi]["result"][k].numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi,
                                score_tol=1e-2, box_tol=1.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graph_replay_is_bitwise_eager(hip_lib, oracle_run, precision):
    """Frames 0..2 of scene A, then a NEW scene (memory reset while the captured graph exists), then 2 more frames: the
    hipGraph engine must reproduce the eager engine bit for bit (deterministic kernels, in-place memory reset)."""
    sd = oracle_run[0]
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


@pytest.mark.parametrize("precision", ["bf16", "bf16_fp32dec", "bf16_fp32val"])
def test_bf16_error_budget_is_measured_and_bounded(hip_lib, oracle_run, precision):
    """bf16 activations through 60 convolutions cannot meet 1e-3 on logits against an fp32 reference; this measures how far
    each precision assignment is, stage by stage (gpurun_out/parity_full_<precision>.json; DESIGN.md §4 quotes the numbers):
      bf16          everything the benchmark runs in bf16 (conv activations/weights, value maps, decoder GEMM operands);
      bf16_fp32dec  bf16 backbone/FPN/2D head and value maps, fp32 decoder + FarHead GEMMs and attention;
      bf16_fp32val  as above with fp32 value maps (the FPN output conv stores fp32 tokens).
    bf16 noise moves the K-th 2D peak of some cameras, so adaptive queries are compared where both sides chose the same peak."""
    sd, want, _ = oracle_run
    eng = _engine(sd, precision)
    report = []
    for fi, (data, metas) in enumerate(_frames()):
        o = eng.forward_frame(data, metas)
        rg, rw, common = _matched_rows(_engine_sel(o), want[fi])
        rep = _stage_errors(o, want[fi], (rg, rw))
        rep.update(frame=fi, proposals_in_common=common, proposals=7 * K, rows_compared=len(rg))
        report.append(rep)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_full_%s.json" % precision), "w") as f:
        json.dump(report, f, indent=1)
    for rep in report:
        print("\n%s error budget vs the fp32 oracle (configs[1]), frame %d: %s" % (precision, rep["frame"], json.dumps(rep)))
    r0 = report[0]
    assert r0["value_maps_rel_max"] < 0.08 and all(r0["fpn%d_rel_max" % l] < 0.08 for l in range(4))
    # frames 1+ are reported for completeness only: a bf16 engine fills its memory with a different top-256 than the oracle, and
    # from then on the two compute different things (see _resolve_memory_ties)
    assert r0["proposals_in_common"] > 0.7 * 7 * K
    assert np.isfinite(r0["logit_max_abs"]) and r0["logit_mean_abs"] < 0.05 and r0["logit_p99_abs"] < 0.3, r0


def _row_ids(sel, prev_ids, prev_topk, A, nq=644):
    """Identity of every query row of one frame: ("L", i) learned query i; ("A", camera, cell) the adaptive query built on that 2D
    peak; ("P", identity of the previous frame's row it was propagated from) for the num_propagated rows at the end (the newest
    memory slots = the previous frame's top-256 in rank order, farhead.py:488-508); None for a first frame's empty memory rows."""
    ids = [("L", i) for i in range(nq)] + [("A", int(n), int(i)) for n, i in sel]
    n_prop = A - len(ids)
    if prev_ids is None:
        return ids + [("P0", j) for j in range(n_prop)]          # scene start: the memory rows are the pseudo reference points, slot by slot
    return ids + [("P", prev_ids[int(r)]) for r in prev_topk[:n_prop].tolist()]


def _unadopted_oracle(sd, dtype, frames):
    """The oracle on `frames` streaming frames with NO forced_* hook: it takes every discrete decision (K-th 2D peak, 3x3 peak test,
    depth-bin argmax, memory top-256) by itself; a recording pass-through only notes its memory selection."""
    from oracle import far3d_oracle
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(proposal_topk=K), dtype=dtype)
    tag = "f64" if dtype == torch.float64 else "f32"
    out = []
    with torch.no_grad():
        for fi, (data, metas) in enumerate(_frames(n=frames)):
            if dtype == torch.float64:
                data = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
            rec = []
            o = orc.simple_test(data, metas, forced_topk=lambda score, own: (rec.append(own.clone()), own)[1],      # adopts nothing
                                camera_cache=(_CAMERA_CACHE, (fi, tag)))
            out.append(dict(sel=[(int(n), int(i)) for n, i, _ in o["roi"]["valid_indices"].nonzero().numpy()],
                            memory_topk=rec[0].flatten().clone(), all_cls_scores=o["all_cls_scores"].clone()))
    return out


def _identity_matched(a, b, tag):
    """Two runs of the same frames (each a list of dicts with sel, memory_topk, all_cls_scores) compared on the query rows that are
    the SAME query on both sides: learned queries by index, adaptive queries by (camera, cell) of their 2D peak (what bench.py's parity
    block does, yolox_head.py:429-458), propagated queries by the identity of the previous frame's row they came from
    (farhead.py:488-491, 736-766).  Returns the per-frame report."""
    report, ids_a, ids_b, topk_a, topk_b = [], None, None, None, None
    for fi, (ga, gb) in enumerate(zip(a, b)):
        A = ga["all_cls_scores"].shape[2]
        assert gb["all_cls_scores"].shape[2] == A
        ids_a = _row_ids(ga["sel"], ids_a, topk_a, A)
        ids_b = _row_ids(gb["sel"], ids_b, topk_b, A)
        topk_a, topk_b = ga["memory_topk"].flatten(), gb["memory_topk"].flatten()
        pos = {k: j for j, k in enumerate(ids_b)}
        assert len(pos) == len(ids_b), "%s: row identities are not unique" % tag
        ra = [j for j, k in enumerate(ids_a) if k in pos]
        rb = [pos[ids_a[j]] for j in ra]
        common = sum(1 for j in ra if ids_a[j][0] == "A")
        d = (ga["all_cls_scores"][:, 0][:, ra].double() - gb["all_cls_scores"][:, 0][:, rb].double()).abs()
        rep = dict(pair=tag, frame=fi, adaptive_in_common=common, adaptive=len(gb["sel"]), rows_compared=len(ra), rows_excluded=A - len(ra),
                   excluded_adaptive=len(ga["sel"]) - common, excluded_propagated=(A - len(ra)) - (len(ga["sel"]) - common),
                   memory_topk_in_common=len(set(topk_a.tolist()) & set(topk_b.tolist())),
                   logit_max_abs=d.max().item(), logit_mean_abs=d.mean().item(), **{"logit_" + k: v for k, v in _pct(d).items() if k != "max"})
        report.append(rep)
    return report


def test_streaming_parity_without_adopting_any_device_decision(hip_lib):
    """The un-adopted witness: the in-tolerance engine (bf16x3) AND the exact-fp32 engine on 5 streaming
    frames at the benchmarked size -- the 1024-slot memory queue is full after frame 3 and overflows at frame 4 (BASELINE configs[4]'s
    "4-frame memory queue", farhead.py:453-508) -- against the oracle run with NO forced_* hooks: the oracle takes every discrete
    decision by itself.  Rows are matched by IDENTITY, not by position (_identity_matched).

    Two yardsticks say what the REFERENCE ARITHMETIC itself does over the same five frames, matched the same way:
      Y1  rounding: the oracle in fp32 against the oracle in fp64, both taking their own decisions;
      Y2  one near-tie decision falling the other way: the fp32 oracle taking its own decisions against the SAME fp32 oracle made to
          adopt the engine's decisions where they differ -- each such difference checked to BE a near-tie (K-th-place 2D scores within
          a relative 5e-4, top-256 cut within 1e-2, depth-bin probabilities within 1e-2: _resolve_*_ties).  Same arithmetic on both
          sides, so Y2 is purely what a flipped near-tie costs: the differing query is excluded from the comparison but stays a
          self-attention key of every other query of its frame and feeds the memory rows of the following ones -- the two runs
          compute slightly different things from then on, whatever the arithmetic.
    Asserted per frame, for both engines: >= 640 of the 644 adaptive queries in common, 99.9 % of the matched logits inside 1e-3, the
    worst matched logit <= max(1e-3, 2 x max(Y1, Y2) of that frame) -- no hard-coded allowance -- and, against the oracle that adopts
    the engine's near-tie decisions, EVERY logit of all five frames inside 1e-3 (the in-bar statement, now with the queue overflowing)."""
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    sd = weights.init_state_dict(weights.detector_spec("V-99-eSE"), seed=0)
    n = FRAMES_QUEUE
    got = {p: _run_engine(sd, p, frames=n, light=True) for p in ("bf16x3", "fp32")}
    for p, g in got.items():
        live = [f["memory_live"] for f in g]
        assert live[4] == 1024 and live[3] == 768, "%s: the memory queue does not fill (live slots at frame start: %s)" % (p, live)
    o32 = _unadopted_oracle(sd, torch.float32, n)
    o64 = _unadopted_oracle(sd, torch.float64, n)
    yard = _identity_matched(o32, o64, "Y1: oracle fp32 vs oracle fp64, nothing adopted")
    reports = {"yardstick_rounding": yard}
    for p, tol in (("bf16x3", 5e-4), ("fp32", 2e-5)):
        reports[p] = _identity_matched(got[p], o32, "%s engine vs oracle fp32, nothing adopted" % p)
        reports[p + "_vs_oracle64"] = _identity_matched(got[p], o64, "%s engine vs oracle fp64, nothing adopted" % p)
        # the fp32 oracle adopting this engine's near-tie decisions (camera stages from the cache: heads only)
        ad = _run_oracle(sd, got[p], with_f64=False, tie_tol=tol)
        o32a = [dict(sel=[(int(a), int(b)) for a, b, _ in w["valid"].nonzero().numpy()], memory_topk=got[p][fi]["memory_topk"].flatten(),
                     all_cls_scores=w["logits"]) for fi, w in enumerate(ad)]
        reports[p + "_yardstick_flipped_near_ties"] = _identity_matched(o32a, o32, "Y2: oracle fp32 adopting the %s engine's near-tie decisions vs oracle fp32 on its own" % p)
        reports[p + "_vs_oracle_adopting"] = _identity_matched(got[p], o32a, "%s engine vs oracle fp32 adopting its near-tie decisions" % p)
    for k, rep in reports.items():
        for r in rep:
            print("\n%s: %s" % (k, json.dumps(r)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_full_unadopted.json"), "w") as f:
        json.dump(reports, f, indent=1)
    for p in ("bf16x3", "fp32"):
        for rep, y1, y2, ra in zip(reports[p], yard, reports[p + "_yardstick_flipped_near_ties"], reports[p + "_vs_oracle_adopting"]):
            assert rep["adaptive_in_common"] >= 640, rep
            assert rep["rows_excluded"] <= 16 + 8 * rep["frame"], rep       # different queries on the two sides and the rows propagated from them
            assert rep["logit_p999"] < 1e-3, rep
            y = max(y1["logit_max_abs"], y2["logit_max_abs"])
            assert rep["logit_max_abs"] <= max(1e-3, 2 * y), "%s frame %d: worst matched logit %.3e > max(1e-3, 2 x yardstick %.3e (Y1 %.3e, Y2 %.3e))" % (
                p, rep["frame"], rep["logit_max_abs"], y, y1["logit_max_abs"], y2["logit_max_abs"])
            # with the near-tie decisions adopted the two sides hold the same queries: every row, the bar itself
            assert ra["rows_excluded"] == 0 and ra["logit_max_abs"] < 1e-3, ra
