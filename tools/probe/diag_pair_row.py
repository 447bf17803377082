"""Why does ONE query of frame 1 come out 0.19 away in the pair-stored (bf16x3) engine when the eSE pooling comes from the GEMM
epilogue?  Runs frames 0-1 of the full-size parity rig, lets the oracle adopt the device's discrete decisions, and compares, for the
worst row of decoder layer 0: the query's inputs (tgt, query_pos, reference point) on both sides, and the conditioning of its key
points in the oracle (depth z along every camera axis: the reference has no behind-camera mask, detr3d_transformer.py:550, so a
key point next to a camera plane turns 1e-7 of input noise into an O(1) sampling change)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far3d_amd import engine, weights  # noqa: E402
from oracle import far3d_oracle  # noqa: E402
from tests import test_engine_full_gpu as T  # noqa: E402


def main():
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    sd = weights.init_state_dict(weights.detector_spec("V-99-eSE"), seed=0)
    eng = engine.Far3DEngine(sd, engine.default_cfg(proposal_topk=T.K), device="cuda:0", precision=sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
    orc = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg(proposal_topk=T.K))
    cap = {}
    real_dec, real_cross = orc.decoder, orc.cross_attn

    def dec(tgt, qpos, feat, ref, *a):
        cap.update(tgt=tgt.clone(), qpos=qpos.clone(), ref=ref.clone(), cross=[])
        return real_dec(tgt, qpos, feat, ref, *a)

    def cross(x, qpos, feat, ref, level_hw, level_start, l2i, pad_hw, lp):
        cap["cross"].append(dict(x=x.clone(), lp=lp, l2i=l2i.clone(), pad_hw=pad_hw))
        return real_cross(x, qpos, feat, ref, level_hw, level_start, l2i, pad_hw, lp)
    orc.decoder, orc.cross_attn = dec, cross
    frames = T._frames()[:2]
    for fi, (data, metas) in enumerate(frames):
        o = eng.forward_frame(data, metas)
        cnt = o["sel_cnt"].cpu().numpy()
        sel = [(n, int(i)) for n in range(7) for i in o["sel_idx"][n, :cnt[n]].cpu().numpy()]
        with torch.no_grad():
            w = orc.simple_test(data, metas, forced_valid=T._resolve_topk_ties(sel, "f%d" % fi, 5e-4),
                                forced_topk=T._resolve_memory_ties(o["memory_topk"].cpu(), "f%d" % fi),
                                forced_depth=T._resolve_depth_ties(o["depth_logit"].float().cpu(), "f%d" % fi, 1e-2))
        d0 = (o["outs_dec"][0].cpu() - w["outs_dec"][0, 0]).abs().max(-1).values
        row = int(d0.argmax())
        A = d0.numel()
        TQ, QP, RF = (eng._bufs[(eng._par, k)][:A].float().cpu() for k in ("tq", "qp", "rf"))
        print("frame %d: worst row %d, decoder-0 error %.4f; logits max %.4f" % (fi, row, d0[row].item(), (o["all_cls_scores"].cpu() - w["all_cls_scores"]).abs().max().item()))
        print("  inputs of that row: |tgt diff| %.3e (max over all rows %.3e), |query_pos diff| %.3e (all rows %.3e), |ref diff| %.3e" %
              ((TQ[row] - cap["tgt"][0, row]).abs().max().item(), (TQ - cap["tgt"][0]).abs().max().item(),
               (QP[row] - cap["qpos"][0, row]).abs().max().item(), (QP - cap["qpos"][0]).abs().max().item(), (RF[row] - cap["ref"][0, row]).abs().max().item()))
        # conditioning of the row's key points in layer 0 of the oracle
        c0 = cap["cross"][0]
        lp = c0["lp"] + "attentions.1."
        pc = orc.P("pts_bbox_head.pc_range")
        x = c0["x"][0, row]
        offs = (orc.P(lp + "learnable_fc.weight") @ x + orc.P(lp + "learnable_fc.bias")).view(13, 3)
        kp = cap["ref"][0, row] * (pc[3:] - pc[:3]) + pc[:3] + offs
        p = torch.einsum("nij,pj->npi", c0["l2i"][0], torch.cat([kp, torch.ones(13, 1)], -1))
        z = p[..., 2]
        u, v = p[..., 0] / z.clamp(min=1e-5) / c0["pad_hw"][1], p[..., 1] / z.clamp(min=1e-5) / c0["pad_hw"][0]
        inside = (u > -0.05) & (u < 1.05) & (v > -0.05) & (v < 1.05)
        print("  key points (13) x cameras (7): min |z| over all %.4f m; pairs inside (or within 5%% of) an image: %d; their z: %s" %
              (z.abs().min().item(), int(inside.sum()), [round(t, 4) for t in z[inside].tolist()][:20]))
        print("  du/dz sensitivity of the inside pairs (|u / z| per metre): %s" % [round(t, 2) for t in (u[inside].abs() / z[inside].abs()).tolist()][:20])


if __name__ == "__main__":
    main()
