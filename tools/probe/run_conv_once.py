"""Launch a few representative convolutions eagerly (for rocprofv3 --pmc passes).  Cases are told apart by grid size."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from far3d_amd import ops
dev = "cuda:0"
CASES = [("s2.c1", 7, 160, 240, 128, 128, 3, 60), ("s4.c1", 7, 40, 60, 192, 192, 3, 60), ("s4.c0", 7, 40, 60, 768, 192, 3, 64),
         ("s4.cat", 7, 40, 60, 1728, 768, 1, 79), ("s3.cat", 7, 80, 120, 1312, 512, 1, 79), ("s2.cat", 7, 160, 240, 768, 256, 1, 70)]
sel = os.environ.get("CASES")
g = torch.Generator().manual_seed(0)
_warm = (torch.zeros(1024, device=dev) + 1).sum().item()   # let torch launch first (profiler start-up)
for name, N, H, W, cin, cout, k, tile in CASES:
    if sel and name not in sel.split(","):
        continue
    tiles = [int(v) for v in os.environ.get("TILE_" + name.replace(".", "_"), str(tile)).split(",")]
    x = torch.randn(N, H, W, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.05
    pc = ops.PackedConv(w, torch.zeros(cout), stride=1, pad=k // 2, dtype=torch.bfloat16, device=dev)
    y = torch.empty(N, H, W, cout, device=dev, dtype=torch.bfloat16)
    for tile in tiles:
        for _ in range(int(os.environ.get("N_LAUNCH", "3"))):
            ops.conv2d_nhwc(x, pc, out=y, act="relu", tile=tile)
        torch.cuda.synchronize()
        print(name, "tile", tile, "done", flush=True)
