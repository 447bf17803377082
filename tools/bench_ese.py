"""Device time of the eSE op (far3d_ese_nhwc: channel sums -> gate -> apply) at the four stage shapes of the benchmarked frame."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from far3d_amd import ops
from tune_conv import timeit
dev = "cuda:0"
tot = 0.0
for (H, W, C, nblk) in ((160, 240, 256, 1), (80, 120, 512, 3), (40, 60, 768, 9), (20, 30, 1024, 3)):
    x = torch.randn(7, H, W, C, device=dev).to(torch.bfloat16)
    idn = torch.randn(7, H, W, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    fcw = torch.randn(C, C, device=dev) * 0.02
    fcb = torch.randn(C, device=dev) * 0.1
    t = timeit(lambda: ops.ese_nhwc(x, fcw, fcb, identity=idn, out=y)) * 1e6
    tot += t * nblk
    print("eSE %dx%dx%d: %.1f us (x%d blocks per frame)" % (H, W, C, t, nblk))
print("per frame: %.1f us" % tot)
