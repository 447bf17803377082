"""A/B of the aggregation kernel's SORTED mode (round 6: operands in launch order + hoisted projection, include/far3d_hip.h) against the
unsorted call, on the seeded config-2 case: HIP events around hipGraphs of 24 launches, the two forms interleaved, bf16 and fp32 value
rows.  Prints one line per (dtype, form) with the launch time and the bitwise comparison of the two results."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:      # A/B of two builds in one GPU call: --lib <path to another libfar3d_hip.so>
    from far3d_amd import lib as _flib
    _flib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
from far3d_amd import ops  # noqa: E402
from tests import cases  # noqa: E402

DEV = "cuda:0"


def graph_time(fn, iters=24, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 1544
    c = cases.aggregate_case(num_cams=7, pad_hw=(640, 960), A=A, seed=0)
    d = lambda t: t.to(DEV).contiguous()
    ref, offs, l2i, U, Vc = d(c["ref"]), d(c["offsets"]).reshape(A, -1), d(c["lidar2img"]), d(c["U"]), d(c["Vc"])
    tab = ops.agg_tables(Vc)
    perm, (inv, qbase) = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"], sorted_operands=True)
    Us, Os = torch.empty_like(U), torch.empty_like(offs)
    Us[inv.long()], Os[inv.long()] = U, offs
    for dt in (torch.bfloat16, torch.float32):
        feat = d(c["feat"].to(dt))
        out_a, out_b = torch.empty(A, 256, device=DEV, dtype=dt), torch.empty(A, 256, device=DEV, dtype=dt)
        fa = lambda: ops.aggregate_forward(feat, ref, offs, l2i, U, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab,
                                           out=out_a, variant=8)
        fb = lambda: ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab,
                                           out=out_b, qbase=qbase)
        out_s = torch.empty(A, 256, device=DEV, dtype=dt)
        lists = ops.AggLists(A, DEV)
        fs = lambda: ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab,
                                           out=out_s, qbase=qbase, variant=13, lists=lists)
        ts = [graph_time(fs) for _ in range(4)]
        fb_ = lambda: ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab,
                                            out=out_b, qbase=qbase)
        tf = [graph_time(fb_) for _ in range(4)]
        print("%-8s rows  TWO-KERNEL SPLIT (variant 13: list build -> global lists -> gather) %s us   fused, sorted %s us   min %.2f against %.2f (%+.1f %%)   "
              "bitwise equal: %s; most entries of a wave %d of 1024" %
              (str(dt).split(".")[1], " ".join("%.2f" % t for t in ts), " ".join("%.2f" % t for t in tf), min(ts), min(tf), 100.0 * (min(ts) / min(tf) - 1.0),
               torch.equal(out_s, out_b), int(lists.counts.max())))
        out_g = torch.empty(A, 256, device=DEV, dtype=dt)
        fg = lambda: ops.aggregate_forward(feat, ref, offs, l2i, U, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab,
                                           out=out_g, variant=12)
        tg, ta, tb = [], [], []
        for _ in range(4):
            tg.append(graph_time(fg))
            ta.append(graph_time(fa))
            tb.append(graph_time(fb))
        same = torch.equal(out_a, out_b)
        print("%-8s rows  greedy deal (variant 12) %s us   split deal, unsorted %s us   split deal, sorted %s us   min %.2f -> %.2f -> %.2f (%+.1f %% / %+.1f %%)   "
              "sorted bitwise == unsorted: %s; |split - greedy| max %.1e" %
              (str(dt).split(".")[1], " ".join("%.2f" % t for t in tg), " ".join("%.2f" % t for t in ta), " ".join("%.2f" % t for t in tb), min(tg), min(ta), min(tb),
               100.0 * (min(ta) / min(tg) - 1.0), 100.0 * (min(tb) / min(tg) - 1.0), same, (out_a.float() - out_g.float()).abs().max().item()))


if __name__ == "__main__":
    main()
