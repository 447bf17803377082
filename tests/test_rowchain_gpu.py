"""GPU: the row-resident decoder chains (far3d_rowchain_attn_out / far3d_rowchain_ffn, csrc/rowchain.hip) against
 (a) a plain PyTorch fp32 statement of the same arithmetic (bf16-rounded operands at the same points, fp32 everywhere else),
 (b) the unfused kernel sequence they replace (far3d_conv2d_nhwc + far3d_layernorm), and
 (c) themselves on row subsets (a row's result must not depend on the rows launched with it: bit-identical),
and the engine with fused_rows=True against the default engine on the golden toy sequence.
Reference semantics: models/utils/detr3d_transformer.py:378-422,522-569 (decoder layer), as restated in oracle/far3d_oracle.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from far3d_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E, FF, NWL = 256, 1024, 455


def _layer(seed):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    pk = lambda w, b: ops.PackedConv(w, b, dtype=torch.bfloat16, device=DEV)
    ly = dict(out=pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), wl=pk(rnd(NWL, 2 * E, sc=(2 * E) ** -0.5), rnd(NWL, sc=0.1)),
              oproj=pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), ffn1=pk(rnd(FF, E, sc=E ** -0.5), rnd(FF, sc=0.1)),
              ffn2=pk(rnd(E, FF, sc=FF ** -0.5), rnd(E, sc=0.1)), qkv=pk(rnd(3 * E, 2 * E, sc=(2 * E) ** -0.5), rnd(3 * E, sc=0.1)),
              norms=[((1 + 0.1 * rnd(E)).to(DEV), (0.1 * rnd(E)).to(DEV)) for _ in range(3)])
    assert ops.RowChainLayer.supported(ly, E, torch.bfloat16)
    ly["rc"] = ops.RowChainLayer(ly)
    return ly


def _inputs(M, seed, wide=False):
    """att / agg bf16, x / qpos f32; wide: as column slices of wider buffers (row strides != E)."""
    g = torch.Generator().manual_seed(1000 + seed)
    def mk(dtype, cols=E):
        t = torch.randn(M, 3 * cols if wide else cols, generator=g).to(DEV).to(dtype)
        return t[:, cols:2 * cols] if wide else t
    return mk(torch.bfloat16), mk(torch.float32), mk(torch.float32) * 0.5


W = lambda pc: pc.w[:pc.Cout].float()
B = lambda pc: pc.bias[:pc.Cout]
bf = lambda t: t.to(torch.bfloat16).float()


def _torch_attn_out(att, x, qpos, ly):
    y = att.float() @ W(ly["out"]).T + B(ly["out"]) + x
    x1 = F.layer_norm(y, (E,), *ly["norms"][0], 1e-5)
    ul = torch.cat([bf(x1 + qpos), bf(x1)], 1) @ W(ly["wl"]).T + B(ly["wl"])
    return x1, ul


def _torch_ffn(agg, x1, qpos, ly, nxt):
    x2 = F.layer_norm(agg.float() @ W(ly["oproj"]).T + B(ly["oproj"]) + x1, (E,), *ly["norms"][1], 1e-5)
    h = bf(torch.relu(bf(x2) @ W(ly["ffn1"]).T + B(ly["ffn1"])))
    out = F.layer_norm(h @ W(ly["ffn2"]).T + B(ly["ffn2"]) + x2, (E,), *ly["norms"][2], 1e-5)
    xop = torch.cat([bf(out + qpos), bf(out)], 1)
    qkv = bf(xop @ W(nxt["qkv"]).T + B(nxt["qkv"])) if nxt is not None else None
    return out, xop, qkv


def _unfused_attn_out(att, x, qpos, ly):
    y = ops.linear(att, ly["out"], res=x)
    xw = torch.empty(att.shape[0], 2 * E, dtype=torch.bfloat16, device=DEV)
    x1 = torch.empty(att.shape[0], E, dtype=torch.float32, device=DEV)
    ops.layernorm(y, *ly["norms"][0], out=x1, add=qpos, y2=xw[:, :E], yb=xw[:, E:])
    return x1, ops.linear(xw, ly["wl"])


def _unfused_ffn(agg, x1, qpos, ly, nxt):
    M = agg.shape[0]
    x2 = torch.empty(M, E, dtype=torch.float32, device=DEV)
    x2b = torch.empty(M, E, dtype=torch.bfloat16, device=DEV)
    ops.layernorm(ops.linear(agg, ly["oproj"], res=x1), *ly["norms"][1], out=x2, yb=x2b)
    hdn = ops.linear(x2b, ly["ffn1"], act="relu", out_dtype=torch.bfloat16)
    out = torch.empty(M, E, dtype=torch.float32, device=DEV)
    xop = torch.empty(M, 2 * E, dtype=torch.bfloat16, device=DEV)
    ops.layernorm(ops.linear(hdn, ly["ffn2"], res=x2), *ly["norms"][2], out=out, add=qpos, y2=xop[:, :E], yb=xop[:, E:])
    qkv = ops.linear(xop, nxt["qkv"], out_dtype=torch.bfloat16) if nxt is not None else None
    return out, xop, qkv


def _close(got, want, tol, what):
    err = (got.float() - want.float()).abs().max().item()
    print("%s: max abs err %.3e (scale %.2f)" % (what, err, want.float().abs().max().item()))
    assert np.isfinite(err) and err < tol, "%s: max abs err %.3e >= %.1e" % (what, err, tol)


# fp32 accumulation order is the only difference up to the first bf16 rounding (x1 / out: ~1e-5); a rounding of [x1 + pos | x1]
# or of the FFN hidden row that flips on such a difference moves one operand by 2^-8 relative: a few 1e-3 after the GEMM.
@pytest.mark.parametrize("M,wide", [(1544, False), (37, True), (16, False), (1, False)])
def test_attn_out_chain_matches_torch_and_the_unfused_kernels(hip_lib, M, wide):
    ly = _layer(3)
    att, x, qpos = _inputs(M, M, wide)
    x1 = torch.full((M, E), float("nan"), device=DEV)
    ulbuf = torch.full((M, 512), float("nan"), device=DEV)
    ops.rowchain_attn_out(att, x, qpos, ly["rc"], x1, ulbuf)
    torch.cuda.synchronize()
    assert torch.isnan(ulbuf[:, NWL:]).all(), "columns past n_wl were written"
    ul = ulbuf[:, :NWL]
    tx1, tul = _torch_attn_out(att, x, qpos, ly)
    _close(x1, tx1, 5e-5, "x1 vs torch")
    _close(ul, tul, 1e-2, "ul vs torch")
    ux1, uul = _unfused_attn_out(att, x, qpos, ly)
    _close(x1, ux1, 5e-5, "x1 vs unfused kernels")
    _close(ul, uul, 1e-2, "ul vs unfused kernels")
    assert (ul - tul).abs().mean().item() < 1e-4


@pytest.mark.parametrize("M,wide,tail", [(1544, False, True), (1544, False, False), (37, True, True), (5, False, True)])
def test_ffn_chain_matches_torch_and_the_unfused_kernels(hip_lib, M, wide, tail):
    ly, nxt = _layer(5), (_layer(6) if tail else None)
    agg, x1, qpos = _inputs(M, 7 * M, wide)
    out = torch.full((M, E), float("nan"), device=DEV)
    qbuf = torch.zeros(M, 6 * 3 * E, dtype=torch.bfloat16, device=DEV)         # a layer's column block of the engine's QKV buffer
    qkv = qbuf[:, 3 * E:6 * E] if tail else None
    xop = torch.zeros(M, 2 * E, dtype=torch.bfloat16, device=DEV)
    ops.rowchain_ffn(agg, x1, qpos, ly["rc"], out, nxt=nxt["rc"] if tail else None, qkv=qkv, xop=xop)
    torch.cuda.synchronize()
    for name, (tout, txop, tqkv) in (("torch", _torch_ffn(agg, x1, qpos, ly, nxt)), ("unfused kernels", _unfused_ffn(agg, x1, qpos, ly, nxt))):
        _close(out, tout, 2e-2, "out vs " + name)
        assert (out - tout).abs().mean().item() < 2e-4
        _close(xop, txop, 6e-2, "xop vs " + name)
        if tail:
            _close(qkv, tqkv, 6e-2, "qkv vs " + name)
            assert (qkv.float() - tqkv.float()).abs().mean().item() < 2e-3
    if tail:
        assert not qbuf[:, :3 * E].any() and not qbuf[:, 6 * E:].any(), "wrote outside the layer's column block"


def _branches(seed, n_cls=26, n_reg=8):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    pk = lambda w, b: ops.PackedConv(w, b, dtype=torch.bfloat16, device=DEV)
    cls = [pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(n_cls, E, sc=E ** -0.5), rnd(n_cls, sc=0.1))]
    reg = [pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(n_reg, E, sc=E ** -0.5), rnd(n_reg, sc=0.1))]
    lns = [((1 + 0.1 * rnd(E)).to(DEV), (0.1 * rnd(E)).to(DEV)) for _ in range(2)]
    assert ops.RowChainBranches.supported(cls, reg, E, torch.bfloat16)
    return cls, lns, reg, ops.RowChainBranches(cls, lns, reg)


@pytest.mark.parametrize("M,n_cls,n_reg", [(6 * 1544, 26, 8), (45, 10, 10), (3, 32, 1)])
def test_branch_chain_matches_torch_and_the_unfused_kernels(hip_lib, M, n_cls, n_reg):
    cls, lns, reg, rb = _branches(21, n_cls, n_reg)
    h = torch.randn(M, E, generator=torch.Generator().manual_seed(M)).to(DEV).to(torch.bfloat16)
    co, ro = torch.full((M, n_cls), float("nan"), device=DEV), torch.full((M, n_reg), float("nan"), device=DEV)
    ops.rowchain_branches(h, rb, co, ro)
    torch.cuda.synchronize()
    # torch fp32, bf16 roundings where the kernels round
    t1 = bf(torch.relu(F.layer_norm(h.float() @ W(cls[0]).T + B(cls[0]), (E,), *lns[0], 1e-5)))
    t2 = bf(torch.relu(F.layer_norm(t1 @ W(cls[1]).T + B(cls[1]), (E,), *lns[1], 1e-5)))
    tc = t2 @ W(cls[2]).T + B(cls[2])
    tr = bf(torch.relu(bf(torch.relu(h.float() @ W(reg[0]).T + B(reg[0]))) @ W(reg[1]).T + B(reg[1]))) @ W(reg[2]).T + B(reg[2])
    # the unfused kernels (far3d_amd.engine.head_stage)
    lin = lambda x, pc, **kw: ops.linear(x, pc, **kw)
    r1 = ops.layernorm(lin(h, cls[0]), *lns[0], act="relu", bf16_copy=True)
    r2 = ops.layernorm(lin(r1[1], cls[1]), *lns[1], act="relu", bf16_copy=True)
    uc = lin(r2[1], cls[2])
    ur = lin(lin(lin(h, reg[0], act="relu", out_dtype=torch.bfloat16), reg[1], act="relu", out_dtype=torch.bfloat16), reg[2])
    for name, wc, wr in (("torch", tc, tr), ("unfused kernels", uc, ur)):
        _close(co, wc, 2e-2, "cls vs " + name)
        _close(ro, wr, 2e-2, "reg vs " + name)
        assert (co - wc).abs().mean().item() < 3e-4 and (ro - wr).abs().mean().item() < 3e-4


def test_chains_give_a_row_the_same_bits_in_any_launch(hip_lib):
    """The query-sharded decoder runs row subsets: rows [a0, a1) of a full launch == a launch over those rows alone."""
    M = 200
    ly, nxt = _layer(8), _layer(9)
    att, x, qpos = _inputs(M, 11)
    agg = _inputs(M, 12)[0]

    def run(a0, a1):
        n = a1 - a0
        x1 = torch.empty(n, E, device=DEV); ul = torch.zeros(n, 512, device=DEV); out = torch.empty(n, E, device=DEV)
        qkv = torch.empty(n, 3 * E, dtype=torch.bfloat16, device=DEV)
        ops.rowchain_attn_out(att[a0:a1], x[a0:a1], qpos[a0:a1], ly["rc"], x1, ul)
        ops.rowchain_ffn(agg[a0:a1], x1, qpos[a0:a1], ly["rc"], out, nxt=nxt["rc"], qkv=qkv)
        return x1, ul, out, qkv

    full = run(0, M)
    for a0, a1 in ((0, 16), (7, 60), (183, 200), (199, 200)):
        for f, p in zip(full, run(a0, a1)):
            assert torch.equal(f[a0:a1], p)


@pytest.mark.parametrize("M", [1544, 21])
def test_in_projection_alone_is_bitwise_the_ffn_chains_tail(hip_lib, M):
    """far3d_rowchain_qkv (what the query-sharded decoder runs after its exchange) == the tail of far3d_rowchain_ffn, bit for bit."""
    ly, nxt = _layer(14), _layer(15)
    agg, x1, qpos = _inputs(M, 31)
    out = torch.empty(M, E, device=DEV)
    tail = torch.empty(M, 3 * E, dtype=torch.bfloat16, device=DEV)
    ops.rowchain_ffn(agg, x1, qpos, ly["rc"], out, nxt=nxt["rc"], qkv=tail)
    out2 = torch.empty(M, E, device=DEV)
    ops.rowchain_ffn(agg, x1, qpos, ly["rc"], out2)                     # the chain without its tail: same rows
    assert torch.equal(out, out2)
    alone = torch.zeros(M, 6 * E, dtype=torch.bfloat16, device=DEV)
    ops.rowchain_qkv(out, qpos, nxt["rc"], alone[:, 3 * E:])
    assert torch.equal(alone[:, 3 * E:], tail) and not alone[:, :3 * E].any()


@pytest.mark.parametrize("M", [1544, 37])
def test_attn_out_chain_stores_the_logits_through_a_row_map(hip_lib, M):
    """ul_rows (round 6: far3d_agg_order's inv): row i of the logits / offsets lands at row ul_rows[i], bit for bit; x1 stays in place."""
    ly = _layer(4)
    att, x, qpos = _inputs(M, M + 1)
    inv = torch.randperm(M, generator=torch.Generator().manual_seed(M)).to(torch.int32).to(DEV)
    x1a, x1b = torch.empty(M, E, device=DEV), torch.empty(M, E, device=DEV)
    ula, ulb = torch.zeros(M, 512, device=DEV), torch.zeros(M, 512, device=DEV)
    ops.rowchain_attn_out(att, x, qpos, ly["rc"], x1a, ula)
    ops.rowchain_attn_out(att, x, qpos, ly["rc"], x1b, ulb, ul_rows=inv)
    assert torch.equal(x1a, x1b)
    assert torch.equal(ulb[inv.long()], ula)


def test_bad_arguments_are_refused(hip_lib):
    from far3d_amd.lib import Far3dHipError
    ly = _layer(2)
    att, x, qpos = _inputs(32, 1)
    x1 = torch.empty(32, E, device=DEV)
    with pytest.raises(ValueError):
        ops.rowchain_attn_out(att.float(), x, qpos, ly["rc"], x1, torch.empty(32, 512, device=DEV))
    with pytest.raises(ValueError):
        ops.rowchain_attn_out(att, x, qpos, ly["rc"], x1, torch.empty(32, 400, device=DEV))          # too narrow for n_wl
    with pytest.raises(ValueError):
        ops.rowchain_ffn(att, x, qpos, ly["rc"], x1, nxt=ly["rc"])                                     # tail without an output
    buf = torch.zeros(32 * E + 4, device=DEV)
    with pytest.raises(Far3dHipError):                                                                  # 4-byte aligned residual rows
        ops.rowchain_attn_out(att, buf[1:1 + 32 * E].view(32, E), qpos, ly["rc"], x1, torch.empty(32, 512, device=DEV))
    bad = ops.PackedConv(torch.randn(300, 2 * E), torch.randn(300), dtype=torch.bfloat16, device=DEV)
    assert not ops.RowChainLayer.supported(dict(ly, wl=bad), E, torch.bfloat16)
    assert not ops.RowChainLayer.supported(ly, E, torch.float32)


def test_engine_with_fused_rows_matches_the_default_engine(hip_lib):
    """Golden toy sequence, bf16: the fused decoder's logits against the default (unfused) bf16 engine's.  Op by op the chains
    agree with the unfused kernels to ~1e-7 plus an occasional flipped bf16 rounding (tests above); through six layers of
    attention those flips spread to every row and grow to a fraction of the bf16 decoder's own rounding noise (measured on the
    first GPU run: mean 2.5e-3, max 1.8e-2 on the logits; the bf16 engine is 6e-3..8e-3 / 5e-2..8e-2 from the fp32 oracle,
    profiles/r4/parity_full_bf16.json).  Bounds: half of that noise in the mean, its size in the max."""
    from far3d_amd import synth
    from tests.test_engine_gpu import _golden_engine
    ref_eng, z, rc = _golden_engine("bf16")
    eng, _, _ = _golden_engine("bf16")
    eng.fused_rows = True
    assert all(ly["rc"] is not None for ly in eng.layers) and eng.branch_rc is not None
    data, metas = synth.recipe_frame(rc, 0)
    a, b = ref_eng.forward_frame(data, metas), eng.forward_frame(data, metas)
    d = (a["all_cls_scores"] - b["all_cls_scores"]).abs()
    print("fused vs default logits: max %.3e mean %.3e" % (d.max().item(), d.mean().item()))
    assert d.max().item() < 8e-2 and d.mean().item() < 5e-3
    # box codes: the centre is sigmoid(reg logit) * 304.8 m, i.e. 76 m per unit of reg logit at the middle (measured: 0.69 m max);
    # the other channels are raw outputs like the logits
    d = (a["all_bbox_preds"] - b["all_bbox_preds"]).abs()
    print("fused vs default boxes: centre max %.3e m, codes max %.3e" % (d[..., :3].max().item(), d[..., 3:].max().item()))
    assert d[..., :3].max().item() < 8e-2 * 76 and d[..., 3:].max().item() < 8e-2
    # (no comparison with the golden fp32 logits here: in bf16 the toy frame selects 106 proposals where the reference has 105, see
    # tests/test_engine_gpu.py::test_engine_bf16_deviation_is_bounded_and_reported; the benchmark-size numbers against the oracle
    # are profiles/r4/rowchain_engine_diff.jsonl + parity_full_bf16.json)


def test_fused_rows_graph_and_pipeline_replay_are_bitwise_eager(hip_lib):
    """hipGraph / frame-pipeline replay of the fused decoder over the golden sequence's scene change (the set-up of
    tests/test_engine_gpu.py::test_engine_graph_mode_with_scene_change_is_bitwise_eager)."""
    from far3d_amd import synth
    from tests.test_engine_gpu import _golden_engine
    res = {}
    for mode in ("eager", "graph", "pipeline"):
        eng, z, rc = _golden_engine("bf16", proposal_topk=12)
        eng.fused_rows = True
        eng.use_graph = mode != "eager"
        eng.pipeline = mode == "pipeline"
        out = []
        for fi in list(range(rc["frames"])) + [3] * 5:
            data, metas = synth.recipe_frame(rc, fi)
            o = eng.forward_frame(data, metas)
            eng.wait_outputs()
            out.append((o["all_cls_scores"].clone(), o["all_bbox_preds"].clone(), {k: v.clone() for k, v in eng.mem.items()}))
        res[mode] = out
    for mode in ("graph", "pipeline"):
        for fi, (a, b) in enumerate(zip(res["eager"], res[mode])):
            assert torch.isfinite(a[0]).all()
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "frame %d: %s differs from eager" % (fi, mode)
            for k in a[2]:
                assert torch.equal(a[2][k], b[2][k]), "frame %d (%s): streaming memory '%s' differs" % (fi, mode, k)
