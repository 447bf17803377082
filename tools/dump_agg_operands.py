"""Dump the aggregation kernel's LIVE operands of one steady-state benchmark frame (all decoder layers) and, per layer, the
per-wave phase stamps of the profiling build -- the data the kernel is redesigned against (patch sizes, rows per query, which waves
form the tail).  Value maps are not dumped (45 MB; the kernel's timing depends on addresses, not on values).

  python tools/dump_agg_operands.py [out.pt]      (GPU box; writes gpurun_out/agg_operands.pt by default)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from far3d_amd import engine, ops, synth, weights  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "agg_operands.pt")
    dev = torch.device("cuda", 0)
    cfg = engine.default_cfg(proposal_topk=92)
    sd = weights.init_state_dict(weights.detector_spec(cfg["backbone"]), seed=0)
    eng = engine.Far3DEngine(sd, cfg, device=dev, precision="bf16")
    eng.use_graph = False
    calls = []
    real = ops.aggregate_forward

    def spy(feat, ref, offsets, lidar2img, U, Vc, level_hw, level_start, pc_range, pad_hw, **kw):
        calls.append(dict(ref=ref.detach().float().cpu().clone(), offsets=offsets.detach().float().cpu().clone(),
                          lidar2img=lidar2img.detach().float().cpu().clone(), U=U.detach().float().cpu().clone(),
                          Vc=Vc.detach().float().cpu().clone(), perm=None if kw.get("perm") is None else kw["perm"].cpu().clone(),
                          level_hw=[tuple(x) for x in level_hw], level_start=list(level_start), pc_range=list(pc_range), pad_hw=tuple(pad_hw),
                          feat_shape=tuple(feat.shape), feat_dtype=str(feat.dtype)))
        return real(feat, ref, offsets, lidar2img, U, Vc, level_hw, level_start, pc_range, pad_hw, **kw)

    frames = [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=dev, ego_motion=True) for fi in range(4)]
    for fi in range(4):
        if fi == 3:
            engine.ops.aggregate_forward = spy
        eng.forward_frame(*frames[fi])
        eng.wait_outputs()
    engine.ops.aggregate_forward = real
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    torch.save(dict(layers=calls, frame_index=3, note="bench workload (BASELINE configs[1]), frame 3 of the sequence, bf16 engine"), out_path)
    print("dumped %d aggregation calls to %s" % (len(calls), out_path))


if __name__ == "__main__":
    main()
