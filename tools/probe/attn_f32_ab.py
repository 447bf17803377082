"""A/B of the fp32 attention instantiations (far3d_attention_f32_variant) at the decoder's size, 1544 queries x 2312 keys x 8 heads:
4 / 2 = the LDS-staged kernel with that many key parts, 0 = the default (register-fed, 64-query workgroups x 4 key parts), 14 / 118 other
shapes of the register-fed kernel.  Error against a float64 softmax(q k^T / sqrt(d)) v."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from far3d_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
q = torch.randn(1544, 256, device=dev); k = torch.randn(2312, 256, device=dev); v = torch.randn(2312, 256, device=dev)
o = torch.empty(1544, 256, device=dev)
qh, kh, vh = (t.double().view(-1, 8, 32).transpose(0, 1) for t in (q, k, v))
want = torch.softmax(qh @ kh.transpose(1, 2) / 32 ** 0.5, -1) @ vh


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(ts)[2]


for variant, what in ((2, "staged, 2 key parts (round 5)"), (4, "staged, 4 key parts"), (0, "register-fed 2 x 4 (default)"),
                      (14, "register-fed 1 x 4"), (118, "register-fed 1 x 8, 64 queries per wave")):
    ops.attention_f32_variant(variant)
    ops.attention_forward(q, k, v, num_heads=8, out=o)
    err = (o.double().view(-1, 8, 32).transpose(0, 1) - want).abs().max().item()
    print("variant %3d  %-42s %5.1f us  max err vs float64 %.2e" % (variant, what, timeit(lambda: ops.attention_forward(q, k, v, num_heads=8, out=o)), err), flush=True)
ops.attention_f32_variant(0)
