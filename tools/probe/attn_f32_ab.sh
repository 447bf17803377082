for p in 2 4 24 14 118; do
FAR3D_ATTN_F32_PARTS=$p timeout 200 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from far3d_amd import ops
import torch.nn.functional as F
dev = "cuda:0"
torch.manual_seed(0)
q = torch.randn(1544, 256, device=dev); k = torch.randn(2312, 256, device=dev); v = torch.randn(2312, 256, device=dev)
o = torch.empty(1544, 256, device=dev)
ops.attention_forward(q, k, v, num_heads=8, out=o)
qh, kh, vh = (t.double().view(-1, 8, 32).transpose(0, 1) for t in (q, k, v))
want = torch.softmax(qh @ kh.transpose(1, 2) / 32 ** 0.5, -1) @ vh
err = (o.double().view(-1, 8, 32).transpose(0, 1) - want).abs().max().item()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(ts)[2]
print("FAR3D_ATTN_F32_PARTS=%s: %.1f us  max err vs float64 %.2e" % (os.environ["FAR3D_ATTN_F32_PARTS"], timeit(lambda: ops.attention_forward(q, k, v, num_heads=8, out=o)), err), flush=True)
PY
done
