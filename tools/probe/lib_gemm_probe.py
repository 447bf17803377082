"""Probe: what the vendor libraries reach on this model's GEMM / conv shapes (targets for the hand-written kernels)."""
import torch
import torch.nn.functional as F
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for name, npix, cout, K in [("s2.cat", 268800, 256, 768), ("s3.cat", 67200, 512, 1312), ("s4.cat", 16800, 768, 1728), ("s5.cat", 4200, 1024, 2144)]:
    x = torch.randn(npix, K, device=dev, dtype=torch.bfloat16); w = torch.randn(cout, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(cout, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: F.linear(x, w, b))
    print("%-8s linear  %8.1f us %7.1f TF/s" % (name, t * 1e6, 2.0 * npix * cout * K / t / 1e12), flush=True)
torch.backends.cudnn.benchmark = True
for name, N, H, W, cin, cout in [("s2.c1", 7, 160, 240, 128, 128), ("s3.c1", 7, 80, 120, 160, 160), ("s4.c1", 7, 40, 60, 192, 192), ("s4.c0", 7, 40, 60, 768, 192), ("s5.c1", 7, 20, 30, 224, 224)]:
    x = torch.randn(N, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    try:
        t = timeit(lambda: F.conv2d(x, w, padding=1))
        print("%-8s miopen  %8.1f us %7.1f TF/s" % (name, t * 1e6, 2.0 * N * H * W * cout * cin * 9 / t / 1e12), flush=True)
    except Exception as ex:
        print(name, "failed", ex)
