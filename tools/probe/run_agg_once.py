"""Launch the fused aggregation kernel a few times eagerly (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from far3d_amd import ops
from tests import cases
dev = "cuda:0"
c = cases.config2_aggregate_case(seed=0)
d = lambda t: t.to(dev).contiguous()
dt = torch.float32 if "--fp32" in sys.argv else torch.bfloat16
feat = d(c["feat"].to(dt))
args = [d(c[k]) for k in ("ref", "offsets", "lidar2img", "U", "Vc")]
perm = ops.camera_sorted_order(args[0], args[2], c["pc_range"], c["pad_hw"]) if "--camsort" in sys.argv else None
out = torch.empty(c["ref"].shape[0], 256, device=dev)
variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 0
for _ in range(int(os.environ.get("N_LAUNCH", "8"))):
    ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out, perm=perm, variant=variant)
torch.cuda.synchronize()
