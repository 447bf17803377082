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
qbase = None
if "--unsorted" not in sys.argv and perm is None and variant in (0, 8):       # as the engine runs it: sorted mode (operands in launch order)
    A = args[0].shape[0]
    perm, (inv, qbase) = ops.aggregation_order(args[0], args[2], c["pc_range"], c["pad_hw"], sorted_operands=True)
    offs2, U2 = args[1].reshape(A, -1), args[3]
    Os, Us = torch.empty_like(offs2), torch.empty_like(U2)
    Os[inv.long()], Us[inv.long()] = offs2, U2
    args[1], args[3] = Os, Us
for _ in range(int(os.environ.get("N_LAUNCH", "8"))):
    ops.aggregate_forward(feat, *args, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], out=out, perm=perm, variant=variant, qbase=qbase)
torch.cuda.synchronize()
