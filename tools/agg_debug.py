"""Run one (variant, case, dtype) of the aggregation kernel against the oracle in this process (crash isolation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import cases
variant, case, dt = int(sys.argv[1]), sys.argv[2], sys.argv[3]
c = dict(small=lambda: cases.small_aggregate_case(0), config2=lambda: cases.config2_aggregate_case(0), near=lambda: cases.near_aggregate_case(1),
         wide=lambda: cases.aggregate_case(7, (640, 960), 64, seed=3, offset_std=12.0))[case]()
dtype = torch.bfloat16 if dt == "bf16" else torch.float32
err = cases.run_aggregate_case(c, "cuda:0", dtype, variant)
torch.cuda.synchronize()
print("variant %d %s %s err %.3e" % (variant, case, dt, err))
