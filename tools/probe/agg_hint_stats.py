"""How often does the aggregation kernel's camera hint (far3d_agg_order: the two cameras a query's reference point projects closest to)
cover the cameras the query's key points actually fall into?  Runs the engine on benchmark frames, takes the LIVE sorted-mode operands of
every decoder layer of the last frame (reference points in metres + hint, key-point offsets, lidar2img) and counts, on the host, the visible
(camera, level) items per query and how many of them belong to a camera outside the hint (those form their weights behind the barrier)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from far3d_amd import engine, ops, synth, weights  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    dev = torch.device("cuda", 0)
    cfg = engine.default_cfg(proposal_topk=92)
    sd = weights.init_state_dict(weights.detector_spec(cfg["backbone"]), seed=0)
    eng = engine.Far3DEngine(sd, cfg, device=dev, precision=prec)
    eng.use_graph = False
    calls = []
    real = ops.aggregate_forward

    def spy(feat, ref, offsets, lidar2img, U, Vc, level_hw, level_start, pc_range, pad_hw, **kw):
        if kw.get("qbase") is not None:
            calls.append(dict(offsets=offsets.detach().float().cpu().clone(), l2i=lidar2img.detach().float().cpu().clone(),
                              qbase=kw["qbase"].detach().cpu().clone(), level_hw=[tuple(x) for x in level_hw], pad_hw=tuple(pad_hw)))
        return real(feat, ref, offsets, lidar2img, U, Vc, level_hw, level_start, pc_range, pad_hw, **kw)

    frames = [synth.make_frame(7, (640, 960), seed=0, frame_index=fi, device=dev, ego_motion=True) for fi in range(4)]
    for fi in range(4):
        if fi == 3:
            engine.ops.aggregate_forward = spy
        eng.forward_frame(*frames[fi])
        eng.wait_outputs()
    engine.ops.aggregate_forward = real
    print("%s engine, frame 3, %d aggregation launches in sorted mode" % (prec, len(calls)))
    for li, c in enumerate(calls):
        qb = c["qbase"]
        A = qb.shape[0]
        refm = qb[:, :3].double()
        hint = qb.view(torch.int32)[:, 3]
        cam0, cam1 = (hint & 0xff).long(), ((hint >> 8) & 0xff).long()
        off = c["offsets"].double().reshape(A, -1, 3)
        kp = torch.cat([refm[:, None, :] + off, torch.ones(A, off.shape[1], 1, dtype=torch.float64)], -1)      # (A, P, 4)
        pr = torch.einsum("nij,apj->napi", c["l2i"].double(), kp)                                               # (N, A, P, 4)
        z = pr[..., 2].clamp(min=1e-5)
        u, v = pr[..., 0] / z / c["pad_hw"][1], pr[..., 1] / z / c["pad_hw"][0]
        items = torch.zeros(pr.shape[0], A, dtype=torch.long)
        for (H, W) in c["level_hw"]:
            fx0 = (u.min(-1).values * W - 0.5).floor().clamp(min=0); fx1 = ((u.max(-1).values * W - 0.5).floor() + 1).clamp(max=W - 1)
            fy0 = (v.min(-1).values * H - 0.5).floor().clamp(min=0); fy1 = ((v.max(-1).values * H - 0.5).floor() + 1).clamp(max=H - 1)
            items += ((fx1 >= fx0) & (fy1 >= fy0)).long()
        per_q = items.sum(0)
        n = torch.arange(pr.shape[0])[:, None]
        hinted = (n == cam0[None]) | (n == cam1[None])
        out = (items * (~hinted).long()).sum(0)
        print("  layer %d: items per query mean %.2f (median %d, max %d); items outside the hint %.1f %% of all items; queries with at least one "
              "such item %.1f %%; queries whose items are all in camera cam0 %.1f %%" %
              (li, per_q.float().mean(), per_q.median(), per_q.max(), 100.0 * out.sum() / max(per_q.sum(), 1), 100.0 * (out > 0).float().mean(),
               100.0 * ((items * (n != cam0[None]).long()).sum(0) == 0).float().mean()))


if __name__ == "__main__":
    main()
