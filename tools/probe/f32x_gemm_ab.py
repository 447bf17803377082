"""A/B of the exact-fp32 pipelined GEMM tiles (482-486; 487-494 = K groups inside the workgroup) against the register-staged exact kernel (tile 3 / auto) and the split-product
fp32-row tiles (479-481, bf16x3 arithmetic) on the decoder's GEMM shapes (the fp32 attention core: tools/probe/attn_f32_ab.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from far3d_amd import ops
dev = "cuda:0"


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


SHAPES = [("qkv", 1544, 512, 768), ("memkv", 768, 512, 4608), ("out", 1544, 256, 256), ("wl", 1544, 512, 455), ("oproj", 1544, 256, 256),
          ("ffn1", 1544, 256, 1024), ("ffn2", 1544, 1024, 256), ("branch", 9264, 256, 256), ("cls_head", 9264, 256, 26), ("mln", 1024, 256, 512)]
for name, M, K, N in SHAPES:
    x = torch.randn(M, K, device=dev)
    w, b = torch.randn(N, K) * 0.05, torch.randn(N)
    pc = ops.PackedConv(w, b, dtype=torch.float32, device=dev)
    pcs = ops.PackedConv(w, b, dtype=torch.float32, device=dev, compute="bf16x3")
    out = torch.empty(M, N, device=dev)
    line = "%-9s M=%5d K=%4d N=%4d |" % (name, M, K, N)
    for tile in (3, 482, 486, 487, 488, 489, 490, 491, 492, 493, 494):
        line += " t%d %5.1f us |" % (tile, timeit(lambda: ops.linear(x, pc, out=out, tile=tile)))
    line += " split(bf16x3) t480 %5.1f us |" % timeit(lambda: ops.linear(x, pcs, out=out, tile=480))
    print(line, flush=True)
