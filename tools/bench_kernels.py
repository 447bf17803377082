"""Per-kernel micro-benchmarks on one MI355X (HIP events on the launch stream).  Writes JSON lines.

  python tools/bench_kernels.py [--iters 200] [--out gpurun_out/kernels.jsonl]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far3d_amd import ops  # noqa: E402
from tests import cases  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters, warmup=20):
    """Device time per call: `iters` launches captured in ONE hipGraph (no host launch gaps), replayed 5x."""
    for _ in range(warmup):
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
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (5 * iters)


def timeit_eager(fn, iters, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def agg_bytes(N, S, C, A, P, G, L, ev):
    # value maps once + unreplicated points + attention weights + fused output
    return N * S * C * ev + A * P * N * 2 * 4 + N * A * G * L * P * 4 + A * C * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--out", default="")
    ap.add_argument("--A", type=int, default=1544, help="number of queries (1544 = BASELINE configs[1])")
    a = ap.parse_args()
    dev = "cuda:0"
    rows = []
    c = cases.aggregate_case(num_cams=7, pad_hw=(640, 960), A=a.A, seed=0)
    N, S, C = c["feat"].shape
    A = c["ref"].shape[0]
    for dt, ev in ((torch.bfloat16, 2), (torch.float32, 4)):
        d = lambda t: t.to(dev).contiguous()
        feat = d(c["feat"].to(dt))
        args = [d(c[k]) for k in ("ref", "offsets", "lidar2img", "U", "Vc")]
        out = torch.empty(A, C, device=dev)
        perm = ops.camera_sorted_order(args[0], args[2], c["pc_range"], c["pad_hw"], spatial=False)
        perm2 = ops.camera_sorted_order(args[0], args[2], c["pc_range"], c["pad_hw"], spatial=True)
        by = agg_bytes(N, S, C, A, 13, 8, 4, ev)
        tab = ops.agg_tables(args[4])
        for variant in (8, 7, 3):
            for name, pm in (("aggregate_fwd", None), ("aggregate_fwd+camsort", perm), ("aggregate_fwd+cam+tile", perm2)):
                fn = lambda: ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out, perm=pm,
                                                   variant=variant, tables=tab)
                t = timeit(fn, a.iters)
                rows.append(dict(kernel="v%d:%s" % (variant, name), dtype=str(dt).split(".")[-1], us=t * 1e6, algorithmic_bytes=by,
                                 achieved_GBps=by / t / 1e9, frac_hbm_peak=by / t / HBM_PEAK))
        # variant 9: kernel 8 + sibling workgroups for the queries two cameras see (the device-side order marks them)
        for extra in (128, 256, 384, 512):
            sp = ops.AggSplit(A, extra, dev)
            pm9 = ops.aggregation_order(args[0], args[2], c["pc_range"], c["pad_hw"], split=sp)
            nsib = int((pm9[A:] != 0x7fffffff).sum().item())
            fn = lambda: ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out, perm=pm9, tables=tab, split=sp)
            t = timeit(fn, a.iters)
            rows.append(dict(kernel="v9:aggregate_fwd+device order+%d sibling slots (%d used)" % (extra, nsib), dtype=str(dt).split(".")[-1], us=t * 1e6,
                             algorithmic_bytes=by, achieved_GBps=by / t / 1e9, frac_hbm_peak=by / t / HBM_PEAK))
        pm8 = ops.aggregation_order(args[0], args[2], c["pc_range"], c["pad_hw"])
        fn = lambda: ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out, perm=pm8, variant=8, tables=tab)
        t = timeit(fn, a.iters)
        rows.append(dict(kernel="v8:aggregate_fwd+device order", dtype=str(dt).split(".")[-1], us=t * 1e6, algorithmic_bytes=by,
                         achieved_GBps=by / t / 1e9, frac_hbm_peak=by / t / HBM_PEAK))
    for r in rows:
        print(json.dumps(r))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
