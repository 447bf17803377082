"""GPU: the reference's threshold proposal rule with STATIC shapes ("cap + count").

The reference keeps every 2D peak with score > 0.1 (ref models/dense_heads/yolox_head.py:429-458, farhead.py:576-581,635-638): the
number M of adaptive queries is data dependent.  The fixed-capacity mode reserves `proposal_capacity` rows, counts M on the device
and masks the unused rows (the "hole") wherever rows interact: as self-attention keys, in the memory top-k, in the decode and in
the aggregation.  No host sync -> the frame can be captured into hipGraphs, pipelined and camera-sharded.  Checked here:
  * the building blocks against torch (masked attention, hole-aware finalisation / ordering / aggregation, gather + compaction);
  * the engine on BOTH golden sequences of the reference (which were generated in threshold mode): eager, hipGraph and pipelined
    runs reproduce the reference's proposals, logits and detections on the rows that hold queries, bit-identical across the modes;
  * overflow is reported, not silently truncated."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from far3d_amd import synth, weights
from tests.conftest import ROOT, assert_detections_match

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")


def _i32(v):
    return torch.tensor([v], dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cnt", [0, 5, 70, 200])
def test_attention_key_hole_equals_attention_without_those_keys(hip_lib, dt, cnt):
    """Keys [start + cnt, end) are masked: the result equals attention over the remaining keys (the hole spans whole 64-key tiles,
    partial tiles and, for cnt = 200, nothing)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(cnt)
    Aq, Nk, E, start, end = 300, 700, 256, 150, 350
    q, k, v = (torch.randn(n, E, generator=g).to(dt) for n in (Aq, Nk, Nk))
    kd, vd = k.to(DEV).clone(), v.to(DEV).clone()
    kd[start + cnt:end] = 0          # the producer zero-fills hole rows; their content must not matter beyond being finite
    vd[start + cnt:end] = 1e4
    got = ops.attention_forward(q.to(DEV), kd, vd, hole=(_i32(cnt), start, end)).float().cpu()
    keep = torch.ones(Nk, dtype=torch.bool)
    keep[start + cnt:end] = False
    want = ops.attention_forward(q.to(DEV), k[keep].to(DEV), v[keep].to(DEV)).float().cpu()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert (got - want).abs().max().item() < tol
    full = ops.attention_forward(q.to(DEV), k.to(DEV), v.to(DEV)).float().cpu()
    if cnt < 200:
        assert (full - want).abs().max().item() > 10 * tol      # the mask matters


def test_hole_aware_finalize_order_and_aggregation(hip_lib):
    from far3d_amd import ops
    from tests import cases
    g = torch.Generator().manual_seed(3)
    layers, A, ncls, code = 3, 120, 26, 8
    start, end, cnt = 40, 90, 17
    reg, ref = torch.randn(layers * A, code, generator=g), torch.rand(A, 3, generator=g)
    cls = torch.randn(layers, 1, A, ncls, generator=g)
    cd = cls.to(DEV).clone()
    box, sc = ops.head_finalize(reg.to(DEV), ref.to(DEV), cd, synth.PC_RANGE, layers, ncls, hole=(_i32(cnt), start, end))
    box0, sc0 = ops.head_finalize(reg.to(DEV), ref.to(DEV), cls.to(DEV), synth.PC_RANGE, layers, ncls)
    hole = torch.zeros(A, dtype=torch.bool)
    hole[start + cnt:end] = True
    assert torch.equal(box, box0)
    assert torch.isinf(sc.cpu()[hole]).all() and (sc.cpu()[hole] < 0).all() and torch.equal(sc.cpu()[~hole], sc0.cpu()[~hole])
    assert torch.isinf(cd.cpu()[:, 0, hole]).all() and torch.equal(cd.cpu()[:, 0, ~hole], cls[:, 0, ~hole])
    idx = ops.topk(sc, 40).cpu()
    assert not hole[idx].any()
    # ordering + aggregation: hole rows come back as exact zeros, the others are untouched
    c = cases.aggregate_case(num_cams=3, pad_hw=(64, 96), A=A, seed=5)
    d = lambda t: t.to(DEV).contiguous()
    args = [d(c[k]) for k in ("ref", "offsets", "lidar2img", "U", "Vc")]
    perm = ops.aggregation_order(args[0], args[2], c["pc_range"], c["pad_hw"], hole=(_i32(cnt), start, end))
    p = perm.cpu()
    assert sorted(torch.where(p < 0, ~p, p).tolist()) == list(range(A))
    assert sorted((~p[p < 0]).tolist()) == list(range(start + cnt, end))
    for variant in (0, 3):
        for fdt in (torch.float32, torch.bfloat16):
            feat = d(c["feat"].to(fdt))
            got = ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, variant=variant).cpu()
            want = ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], variant=variant).cpu()
            assert (got[hole] == 0).all() and torch.equal(got[~hole], want[~hole])


def test_gather_fixed_capacity_and_block_compaction(hip_lib):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(4)
    N, C, strides, hw = 3, 256, (8, 16), [(8, 12), (4, 6)]
    S = sum(h * w for h, w in hw)
    cls = [torch.randn(N, h, w, 26, generator=g).to(DEV) for h, w in hw]
    reg = [torch.randn(N, h, w, 5, generator=g).to(DEV) for h, w in hw]
    depth = torch.randn(N, 8, 12, 51, generator=g).to(DEV)
    i2l = torch.linalg.inv(synth.ring_cameras(N, (64, 96))[2]).float().contiguous().to(DEV)
    feat = torch.randn(N, S, C, generator=g).to(DEV)
    dcfg = dict(num_depth_bins=50, depth_min=0.1, depth_max=110.0)
    wgt, sel_idx, sel_cnt = ops.proposal_select(cls, reg, strides, 32, thr=0.3)
    M = int(sel_cnt.sum().item())
    assert 3 < M < 60
    legacy = ops.proposal_gather(reg, strides, sel_idx, sel_cnt, wgt, depth, 8, dcfg, i2l, feat, synth.PC_RANGE)
    for rows in (M + 9, M, M - 2):
        m_out, ovf = _i32(-1), _i32(-1)
        out = tuple(torch.full((rows, w), 7.0, device=DEV) for w in (3, C + 1, 4)) + (torch.full((rows,), 7.0, device=DEV),)
        got = ops.proposal_gather(reg, strides, sel_idx, sel_cnt, wgt, depth, 8, dcfg, i2l, feat, synth.PC_RANGE, out=out, rows_total=rows,
                                  m_out=m_out, overflow_out=ovf)
        mv = min(M, rows)
        assert int(m_out.item()) == mv and int(ovf.item()) == (1 if M > rows else 0)
        for a, b in zip(got, legacy):
            assert torch.equal(a[:mv], b[:mv])
        assert (got[0][mv:] == 0).all() and (got[1][mv:] == 0).all() and (got[2][mv:] == 0).all() and (got[3][mv:] == 0).all()
    # a camera that fills its selection capacity raises the flag even when the rows suffice
    w2, si2, sc2 = ops.proposal_select(cls, reg, strides, 2, thr=0.3)
    m_out, ovf = _i32(-1), _i32(0)
    out = tuple(torch.zeros((64, w), device=DEV) for w in (3, C + 1, 4)) + (torch.zeros((64,), device=DEV),)
    ops.proposal_gather(reg, strides, si2, sc2, w2, depth, 8, dcfg, i2l, feat, synth.PC_RANGE, out=out, rows_total=64, m_out=m_out, overflow_out=ovf)
    assert int(ovf.item()) == 1 and int(m_out.item()) == int(sc2.sum().item())
    # block -> compact
    src = torch.randn(4, 6, 10, generator=g).to(DEV)
    counts = torch.tensor([2, 0, 6, 3], dtype=torch.int32, device=DEV)
    for rows in (16, 11, 9):
        dst, m_out, ovf = torch.full((rows, 10), 9.0, device=DEV), _i32(-1), _i32(0)
        ops.compact_rows(src, counts, dst, m_out, ovf)
        want = torch.cat([src[0, :2], src[2, :6], src[3, :3]])[:rows]
        assert int(m_out.item()) == min(11, rows) and int(ovf.item()) == (1 if rows < 11 else 0)
        assert torch.equal(dst[:len(want)], want) and (dst[len(want):] == 0).all()


def _golden_engine(precision, name, **over):
    from far3d_amd import engine
    z = np.load(os.path.join(GOLD, name + ".npz"))
    rc = json.loads(bytes(z["recipe"]).decode())
    spec = weights.detector_spec(rc["backbone"], num_query=rc["num_query"], num_propagated=rc["num_propagated"])
    sd = weights.init_state_dict(spec, seed=rc["weight_seed"])
    cfg = engine.default_cfg(backbone=rc["backbone"], num_cams=rc["num_cams"], num_query=rc["num_query"],
                             num_propagated=rc["num_propagated"], memory_len=rc["memory_len"], topk_proposals=rc["topk_proposals"], **over)
    return engine.Far3DEngine(sd, cfg, device=DEV, precision=precision), z, rc


def _valid_rows(t, dim, nq, M, cap):
    """Drop the hole rows [nq + M, nq + cap) along `dim`: what remains is the reference's query order."""
    idx = torch.cat([torch.arange(0, nq + M), torch.arange(nq + cap, t.shape[dim])]).to(t.device)
    return t.index_select(dim, idx)


@pytest.mark.parametrize("name", ["far3d_small_seq", "far3d_c1_seq"])
def test_fixed_capacity_threshold_mode_reproduces_the_reference_in_graph_and_pipeline_mode(hip_lib, name):
    cap = {"far3d_small_seq": 48, "far3d_c1_seq": 96}[name]      # the goldens hold 26-29 / 80-81 proposals per frame
    res = {}
    for mode in ("eager", "graph", "pipeline"):
        eng, z, rc = _golden_engine("fp32", name, proposal_topk=None, proposal_capacity=cap)
        eng.use_graph = mode != "eager"
        eng.pipeline = mode == "pipeline"
        nq = rc["num_query"]
        out = []
        for fi in list(range(rc["frames"])) + [rc["frames"] - 1] * 5:      # + steady frames so that every pipeline buffer set captures and replays
            data, metas = synth.recipe_frame(rc, fi)
            o = eng.forward_frame(data, metas)
            eng.wait_outputs()
            eng.check_proposal_overflow()
            M = int(o["num_adaptive_dev"].item())
            assert o["num_adaptive"] == cap and M <= cap
            out.append((M, o["all_cls_scores"].clone(), o["all_bbox_preds"].clone(), {k: v.clone() for k, v in o["result"].items()},
                        {k: v.clone() for k, v in eng.mem.items()}, o["sel_idx"].clone(), o["sel_cnt"].clone()))
        if mode == "graph":
            assert eng._graph is not None
        if mode == "pipeline":
            assert sorted(eng._pipe["g_head"]) == list(range(eng.pipeline_sets))
        res[mode] = out
    # the reference's own outputs (goldens are generated in threshold mode) on the rows that hold queries
    for fi in range(rc["frames"]):
        M, cls, box, r, mem, sel_idx, sel_cnt = res["eager"][fi]
        want_idx = z["f%d_valid_idx" % fi]
        assert M == len(want_idx), (fi, M, len(want_idx))
        cnt = sel_cnt.cpu().numpy()
        got = [(n, int(i)) for n in range(rc["num_cams"]) for i in sel_idx[n, :cnt[n]].cpu().numpy()]
        assert got == [(int(a[0]), int(a[1])) for a in want_idx]
        g_cls = _valid_rows(cls, 2, nq, M, cap).cpu().numpy()
        g_box = _valid_rows(box, 2, nq, M, cap).cpu().numpy()
        w_cls, w_box = z["f%d_all_cls_scores" % fi], z["f%d_all_bbox_preds" % fi]
        assert g_cls.shape == w_cls.shape, (g_cls.shape, w_cls.shape)
        assert np.abs(g_cls - w_cls).max() < 1e-3, "frame %d logits: %.3e" % (fi, np.abs(g_cls - w_cls).max())
        assert np.abs(g_box - w_box)[..., :3].max() < 0.076 and np.abs(g_box - w_box)[..., 3:].max() < 1e-3
        hole = cls[:, 0, nq + M:nq + cap]
        assert hole.numel() == 0 or bool(torch.isinf(hole).all())
        keep = r["keep"].cpu().numpy()
        assert_detections_match(tuple(r[k].cpu().numpy()[keep] for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                tuple(z["f%d_%s" % (fi, k)] for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)
    # hipGraph replay and the pipelined engine: bit-identical to eager, hole rows included
    for mode in ("graph", "pipeline"):
        for fi, (a, b) in enumerate(zip(res["eager"], res[mode])):
            assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), "frame %d: %s differs from eager" % (fi, mode)
            for k in a[4]:
                assert torch.equal(a[4][k], b[4][k]), "frame %d (%s): streaming memory '%s' differs" % (fi, mode, k)


def test_capacity_overflow_is_reported(hip_lib):
    from far3d_amd import lib
    eng, z, rc = _golden_engine("fp32", "far3d_small_seq", proposal_topk=None, proposal_capacity=2)
    data, metas = synth.recipe_frame(rc, 0)
    assert len(z["f0_valid_idx"]) > 2
    o = eng.forward_frame(data, metas)
    assert int(o["num_adaptive_dev"].item()) == 2 and torch.isfinite(o["all_cls_scores"][:, 0, :rc["num_query"] + 2]).all()
    with pytest.raises(lib.Far3dHipError):
        eng.check_proposal_overflow()


def test_detector_from_registry_config_in_capacity_mode(hip_lib):
    """The registry-level detector with `proposal_capacity` (INTEGRATION.md "Proposal modes"): graphs + pipeline on, the reference's
    boxes of the golden sequence come out, and a capacity that is too small raises instead of truncating."""
    from far3d_amd import config, lib, plugin
    z = np.load(os.path.join(GOLD, "far3d_small_seq.npz"))
    rc = json.loads(bytes(z["recipe"]).decode())
    spec = weights.detector_spec(rc["backbone"], num_query=rc["num_query"], num_propagated=rc["num_propagated"])
    for cap in (48, 4):
        det = plugin.build_detector(config.default_model_cfg(num_cams=rc["num_cams"], num_query=rc["num_query"], num_propagated=rc["num_propagated"],
                                                             memory_len=rc["memory_len"], topk_proposals=rc["topk_proposals"], proposal_capacity=cap))
        det.load_state_dict(weights.init_state_dict(spec, seed=rc["weight_seed"]))
        det.prepare(DEV, precision="fp32")
        det.engine.use_graph = det.engine.pipeline = True
        if cap == 4:
            data, metas = synth.recipe_frame(rc, 0)
            with pytest.raises(lib.Far3dHipError):
                det(return_loss=False, rescale=True, img_metas=metas, **data)
            continue
        for fi in range(rc["frames"]):
            data, metas = synth.recipe_frame(rc, fi)
            res = det(return_loss=False, rescale=True, img_metas=metas, **data)[0]["pts_bbox"]
            assert_detections_match(tuple(res[k].cpu().numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                    tuple(z["f%d_%s" % (fi, k)] for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)


@pytest.mark.parametrize("mode", ["graph", "pipeline"])
def test_capacity_overflow_on_a_replayed_frame_of_each_buffer_set(hip_lib, mode):
    """the overflow flag lives in the frame's own buffer set; a REPLAYED graph never runs the Python line that recorded it
    at capture time, so the engine binds it per frame.  Frames that fit (scene start, a capture and a first replay per buffer
    set), then one replay per buffer set whose 29 proposals exceed the 27 reserved rows: each must raise, the frames before must not."""
    from far3d_amd import lib
    eng, z, rc = _golden_engine("fp32", "far3d_small_seq", proposal_topk=None, proposal_capacity=27)
    assert len(z["f1_valid_idx"]) <= 27 < len(z["f0_valid_idx"])
    eng.use_graph = True
    eng.pipeline = mode == "pipeline"
    n = eng.pipeline_sets if mode == "pipeline" else 1
    for step, fi in enumerate([1] * (1 + 2 * n) + [0] * n):      # scene start, a capture and a replay per buffer set, then an overflowing replay on each
        data, metas = synth.recipe_frame(rc, fi)      # frames 0 and 1 belong to the same scene
        eng.forward_frame(data, metas)
        eng.wait_outputs()
        if fi == 1:
            eng.check_proposal_overflow()
        else:
            with pytest.raises(lib.Far3dHipError):
                eng.check_proposal_overflow()
    if mode == "pipeline":
        assert sorted(eng._pipe["g_head"]) == list(range(eng.pipeline_sets))
