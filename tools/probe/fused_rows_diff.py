"""fused_rows vs default engine at the benchmark size: frame 0 (no memory), the memory top-k overlap, and frame 1 with the SAME memory
state in both engines -- separates the chains' own numerical difference from what a flipped memory selection does downstream."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from far3d_amd import engine, synth, weights  # noqa: E402

DEV = "cuda:0"
cfg = engine.default_cfg(proposal_topk=92)
sd = weights.init_state_dict(weights.detector_spec(cfg["backbone"]), seed=0)
engs = []
for fused in (False, True):
    e = engine.Far3DEngine(sd, cfg, device=DEV, precision="bf16")
    e.fused_rows = fused
    engs.append(e)
frames = [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=DEV, ego_motion=True) for fi in range(2)]
o = [e.forward_frame(*frames[0]) for e in engs]
d = (o[0]["all_cls_scores"] - o[1]["all_cls_scores"]).abs()
per_layer = [round(d[l].mean().item(), 6) for l in range(d.shape[0])]
k0, k1 = (set(x["memory_topk"].cpu().tolist()) for x in o)
print(json.dumps(dict(frame=0, logits_max=round(d.max().item(), 5), logits_mean=round(d.mean().item(), 6), per_layer_mean=per_layer,
                      logit_scale=round(o[0]["all_cls_scores"].abs().mean().item(), 4), topk_overlap=len(k0 & k1), topk=len(k0),
                      topk_same_order=bool(torch.equal(o[0]["memory_topk"], o[1]["memory_topk"])))))
for k, v in engs[0].mem.items():                      # same streaming memory for frame 1
    engs[1].mem[k].copy_(v)
o = [e.forward_frame(*frames[1]) for e in engs]
d = (o[0]["all_cls_scores"] - o[1]["all_cls_scores"]).abs()
print(json.dumps(dict(frame=1, same_memory=True, logits_max=round(d.max().item(), 5), logits_mean=round(d.mean().item(), 6),
                      per_layer_mean=[round(d[l].mean().item(), 6) for l in range(d.shape[0])])))
