"""Kernel times of the decoder-sized fp32 linear layers: the exact-fp32 staged kernel against the 3-product kernels (staged tile 3, pipelined fp32-row tiles
479-481).  HIP-event timing of this loop is host-bound (~14 us per ops.linear call): run it under `rocprofv3 --kernel-trace` and read the trace
(profiles/r5/fp32_rows_gemm.txt).  FAR3D_AB_LIB=<name>.so times another build of the library."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd import lib as flib
if os.environ.get("FAR3D_AB_LIB"):
    flib.LIB_PATH = os.path.join(ROOT, "far3d_amd", os.environ["FAR3D_AB_LIB"])
flib.load()
from far3d_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def timeit(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for M, K, N in [(1544, 256, 256), (1544, 256, 768), (2568, 256, 512), (1544, 1024, 256), (1544, 256, 1024), (1544, 512, 256)]:
    x = torch.randn(M, K, generator=g).to(dev)
    pc = ops.PackedConv(torch.randn(N, K, 1, 1, generator=g) * 0.05, torch.randn(N, generator=g), dtype=torch.float32, device=dev)
    out = torch.empty(M, N, device=dev)
    us = timeit(lambda: ops.linear(x, pc, out=out))
    ref = x @ pc.w[:N, :K].float().t() if False else None
    pcx = ops.PackedConv(torch.randn(N, K, 1, 1, generator=g) * 0.05, torch.randn(N, generator=g), dtype=torch.float32, device=dev, compute="bf16x3")
    row = "fp32 linear M=%d K=%d N=%d  exact-fp32 staged %.1f us" % (M, K, N, us)
    for tile in (0, 479, 480, 481, 3):
        try:
            row += " | x3 tile %d: %.1f" % (tile, timeit(lambda: ops.linear(x, pcx, out=out, tile=tile)))
        except Exception as e:   # noqa
            row += " | x3 tile %d: %s" % (tile, str(e)[:40])
    print(row, flush=True)
