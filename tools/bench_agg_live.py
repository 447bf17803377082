"""Aggregation kernel variants on the LIVE operands of a benchmark frame (tools/dump_agg_operands.py): device time per launch
(hipGraph of 24 launches, HIP events) per decoder layer, and the max abs difference between variants.

  python tools/bench_agg_live.py gpurun_out/agg_operands.pt [variants: 7 8 ...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far3d_amd import ops  # noqa: E402
from tools.bench_kernels import timeit, agg_bytes  # noqa: E402


def main():
    path = sys.argv[1]
    variants = [int(v) for v in sys.argv[2:]] or [7, 8]
    dev = "cuda:0"
    layers = torch.load(path)["layers"]
    d = lambda t: t.to(dev).contiguous()
    for dt, ev in ((torch.bfloat16, 2), (torch.float32, 4)):
        for li, c in enumerate(layers):
            if dt == torch.float32 and li not in (0, len(layers) - 1):
                continue
            feat = torch.randn(c["feat_shape"], device=dev).to(dt)
            args = [d(c[k]) for k in ("ref", "offsets", "lidar2img", "U", "Vc")]
            perm = d(c["perm"]) if c["perm"] is not None else None
            tab = ops.agg_tables(args[4])
            N, S, C = c["feat_shape"]
            A = args[0].shape[0]
            by = agg_bytes(N, S, C, A, 13, 8, 4, ev)
            outs = {}
            for v in variants:
                out = torch.empty(A, C, device=dev)
                fn = lambda: ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out,
                                                   perm=perm, variant=v, tables=tab)
                t = timeit(fn, 24)
                outs[v] = out.clone()
                print(json.dumps(dict(layer=li, variant=v, dtype=str(dt).split(".")[-1], us=round(t * 1e6, 2), frac_hbm_peak=round(by / t / 8e12, 4),
                                      max_abs_diff_vs_first=round((outs[v] - outs[variants[0]]).abs().max().item(), 8))))
            if os.environ.get("NOPERM") and li == len(layers) - 1:     # the same launch in natural query order (no camera / cell sort)
                v = variants[-1]
                fn = lambda: ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out,
                                                   perm=None, variant=v, tables=tab)
                print(json.dumps(dict(layer=li, variant=v, dtype=str(dt).split(".")[-1], order="natural", us=round(timeit(fn, 24) * 1e6, 2))))


if __name__ == "__main__":
    main()
