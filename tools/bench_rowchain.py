"""Row-resident decoder chains (far3d_rowchain_attn_out / far3d_rowchain_ffn) against the unfused kernel sequences they replace:
device time per call (hipGraph of back-to-back launches, HIP events) at the benchmark's 1544 query rows, and the whole head stage
of a benchmark frame with fused_rows off / on.  One JSON line per measurement.

  python tools/bench_rowchain.py [--rows 1544] [--no-engine]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far3d_amd import engine, ops, synth, weights  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

E, FF, NWL = 256, 1024, 455
DEV = "cuda:0"


def layer(seed):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    pk = lambda w, b: ops.PackedConv(w, b, dtype=torch.bfloat16, device=DEV)
    ly = dict(out=pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), wl=pk(rnd(NWL, 2 * E, sc=(2 * E) ** -0.5), rnd(NWL, sc=0.1)),
              oproj=pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), ffn1=pk(rnd(FF, E, sc=E ** -0.5), rnd(FF, sc=0.1)),
              ffn2=pk(rnd(E, FF, sc=FF ** -0.5), rnd(E, sc=0.1)), qkv=pk(rnd(3 * E, 2 * E, sc=(2 * E) ** -0.5), rnd(3 * E, sc=0.1)),
              norms=[((1 + 0.1 * rnd(E)).to(DEV), (0.1 * rnd(E)).to(DEV)) for _ in range(3)])
    ly["rc"] = ops.RowChainLayer(ly)
    return ly


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1544)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-engine", action="store_true")
    ap.add_argument("--engine-only", action="store_true")
    a = ap.parse_args()
    M = a.rows
    if a.engine_only:
        return head_stage()
    ly, nxt = layer(1), layer(2)
    att = torch.randn(M, E, device=DEV).to(torch.bfloat16)
    x, qpos = torch.randn(M, E, device=DEV), torch.randn(M, E, device=DEV) * 0.5
    x1, out = torch.empty(M, E, device=DEV), torch.empty(M, E, device=DEV)
    ul = torch.empty(M, 512, device=DEV)
    qkv = torch.empty(M, 3 * E, dtype=torch.bfloat16, device=DEV)
    xw = torch.empty(M, 2 * E, dtype=torch.bfloat16, device=DEV)
    x2, x2b = torch.empty(M, E, device=DEV), torch.empty(M, E, dtype=torch.bfloat16, device=DEV)
    y = torch.empty(M, E, device=DEV)
    hdn = torch.empty(M, FF, dtype=torch.bfloat16, device=DEV)

    def unfused_a():
        ops.linear(att, ly["out"], res=x, out=y)
        ops.layernorm(y, *ly["norms"][0], out=x1, add=qpos, y2=xw[:, :E], yb=xw[:, E:])
        ops.linear(xw, ly["wl"], out=ul[:, :NWL])

    def unfused_b():
        ops.linear(att, ly["oproj"], res=x1, out=y)
        ops.layernorm(y, *ly["norms"][1], out=x2, yb=x2b)
        ops.linear(x2b, ly["ffn1"], act="relu", out=hdn, out_dtype=torch.bfloat16)
        ops.linear(hdn, ly["ffn2"], res=x2, out=y)
        ops.layernorm(y, *ly["norms"][2], out=out, add=qpos, y2=xw[:, :E], yb=xw[:, E:])
        ops.linear(xw, nxt["qkv"], out=qkv, out_dtype=torch.bfloat16)

    cases = [("attn_out: out-proj + LN0 + wl", 3, unfused_a, lambda: ops.rowchain_attn_out(att, x, qpos, ly["rc"], x1, ul),
              (E * E + 2 * E * 464) * 2),
             ("ffn: oproj + LN1 + FFN + LN2 + next qkv", 6, unfused_b,
              lambda: ops.rowchain_ffn(att, x1, qpos, ly["rc"], out, nxt=nxt["rc"], qkv=qkv), (E * E + 2 * E * FF + 2 * E * 3 * E) * 2),
             ("ffn without the qkv tail (last layer)", 5, None, lambda: ops.rowchain_ffn(att, x1, qpos, ly["rc"], out), (E * E + 2 * E * FF) * 2)]
    for name, n_unf, unf, fused, wbytes in cases:
        tf = timeit(fused, a.iters)
        rec = dict(chain=name, rows=M, fused_us=round(tf * 1e6, 2), workgroups=-(-M // 16),
                   weight_stream_GBps_per_cu=round(wbytes / tf / 1e9, 1))
        if unf is not None:
            tu = timeit(unf, a.iters)
            rec.update(unfused_us=round(tu * 1e6, 2), unfused_launches=n_unf, speedup=round(tu / tf, 2))
        print(json.dumps(rec), flush=True)
    # the classification / regression branches over all layers' outputs (6 x rows)
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s_, sc=1.0: torch.randn(*s_, generator=g) * sc
    pk = lambda w, b_: ops.PackedConv(w, b_, dtype=torch.bfloat16, device=DEV)
    cls = [pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(26, E, sc=E ** -0.5), rnd(26, sc=0.1))]
    reg = [pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(E, E, sc=E ** -0.5), rnd(E, sc=0.1)), pk(rnd(8, E, sc=E ** -0.5), rnd(8, sc=0.1))]
    lns = [((1 + 0.1 * rnd(E)).to(DEV), (0.1 * rnd(E)).to(DEV)) for _ in range(2)]
    rb = ops.RowChainBranches(cls, lns, reg)
    Mb = 6 * M
    h = torch.randn(Mb, E, device=DEV).to(torch.bfloat16)
    co, ro = torch.empty(Mb, 26, device=DEV), torch.empty(Mb, 8, device=DEV)

    def unfused_br():
        r1 = ops.layernorm(ops.linear(h, cls[0]), *lns[0], act="relu", bf16_copy=True)
        r2 = ops.layernorm(ops.linear(r1[1], cls[1]), *lns[1], act="relu", bf16_copy=True)
        ops.linear(r2[1], cls[2])
        ops.linear(ops.linear(ops.linear(h, reg[0], act="relu", out_dtype=torch.bfloat16), reg[1], act="relu", out_dtype=torch.bfloat16), reg[2])

    tf, tu = timeit(lambda: ops.rowchain_branches(h, rb, co, ro), a.iters), timeit(unfused_br, a.iters)
    print(json.dumps(dict(chain="cls + reg branches", rows=Mb, fused_us=round(tf * 1e6, 2), unfused_us=round(tu * 1e6, 2), unfused_launches=8,
                          workgroups=-(-Mb // 16), speedup=round(tu / tf, 2))), flush=True)
    if not a.no_engine:
        head_stage()


def head_stage():
    """The head stage of a steady benchmark frame, unfused vs fused: device time (hipGraph replay, no launch gaps), C-ABI calls per
    stage (one call = one kernel launch, far3d_agg_order / decode aside), logits of the two engines on the same frame."""
    from far3d_amd import lib as _lib
    cfg = engine.default_cfg(proposal_topk=92)
    sd = weights.init_state_dict(weights.detector_spec(cfg["backbone"]), seed=0)
    res = {}
    for fused in (False, True):
        eng = engine.Far3DEngine(sd, cfg, device=DEV, precision="bf16", parts=("backbone", "neck", "roi", "head"))
        eng.fused_rows = fused
        frames = [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=DEV, ego_motion=True) for fi in range(2)]
        o = eng.forward_frame(*frames[0])          # frame 0 (no streaming memory yet) is what the two engines are compared on:
        logits0 = o["all_cls_scores"].clone()      # later frames also differ by whatever a flipped memory top-k entry does downstream
        eng.forward_frame(*frames[1])
        data, metas = frames[1]
        pad_hw = tuple(metas[0]["pad_shape"][0][:2])
        dd = eng._stage_inputs(data)
        st = eng._camera_part(dd, pad_hw)
        torch.cuda.synchronize()
        calls, real_check = [0], _lib.check

        def counting_check(status, what):
            calls[0] += 1
            return real_check(status, what)
        _lib.check = ops._lib.check = counting_check
        try:
            eng._head_part(st, dd, metas, pad_hw)
        finally:
            _lib.check = ops._lib.check = real_check
        t = timeit(lambda: eng._head_part(st, dd, metas, pad_hw), 3, warmup=2)
        res[fused] = (t, logits0)
        print(json.dumps(dict(stage="head", fused_rows=fused, ms=round(t * 1e3, 4), c_abi_calls=calls[0])), flush=True)
    d = (res[False][1] - res[True][1]).abs()
    print(json.dumps(dict(stage="head", frame=0, logits_max_abs_diff=round(d.max().item(), 6), logits_mean_abs_diff=round(d.mean().item(), 7),
                          speedup=round(res[False][0] / res[True][0], 3))))


if __name__ == "__main__":
    main()
