"""Seeded input cases shared by the parity tests, smoke() and the fixture generator."""
import torch

from far3d_amd import synth


def aggregate_case(num_cams, pad_hw, A, P=13, G=8, C=256, seed=0, offset_std=2.0):
    g = torch.Generator().manual_seed(seed)
    hw = synth.level_shapes(pad_hw)
    starts, S = synth.level_starts(hw)
    L = len(hw)
    _, _, l2i = synth.ring_cameras(num_cams, pad_hw)
    case = dict(
        feat=torch.randn(num_cams, S, C, generator=g),
        ref=torch.rand(A, 3, generator=g),
        offsets=torch.randn(A, P, 3, generator=g) * offset_std,
        lidar2img=l2i.contiguous(),
        U=torch.randn(A, L * P * G, generator=g),
        Vc=torch.randn(num_cams, L * P * G, generator=g),
        level_hw=hw, level_start=starts, pc_range=list(synth.PC_RANGE), pad_hw=tuple(pad_hw), G=G,
    )
    return case


def small_aggregate_case(seed=0):
    return aggregate_case(num_cams=3, pad_hw=(64, 96), A=37, seed=seed, offset_std=4.0)


def config2_aggregate_case(seed=0):
    """BASELINE.json configs[1]: 7 cams, 640x960, A = 644 + 256 + 644."""
    return aggregate_case(num_cams=7, pad_hw=(640, 960), A=1544, seed=seed)


def oracle_aggregate(case, value_dtype=torch.float32):
    from oracle import sampling
    feat = case["feat"]
    if value_dtype == torch.bfloat16:
        feat = feat.to(torch.bfloat16).float()
    logits = case["U"][:, None, :] + case["Vc"][None, :, :]
    return sampling.aggregation_ref(feat, case["ref"], case["offsets"], case["lidar2img"], logits,
                                    case["level_hw"], case["level_start"], case["pc_range"], case["pad_hw"],
                                    num_groups=case["G"])


def near_aggregate_case(seed=0, A=96):
    """Queries a few metres from the cameras with metre-scale offsets: the 13 points of a query spread over many tokens of
    the fine levels (wide-spread path of the v4 kernel) and over several cameras, some behind the camera."""
    c = aggregate_case(num_cams=7, pad_hw=(640, 960), A=A, seed=seed, offset_std=1.5)
    g = torch.Generator().manual_seed(seed + 77)
    c["ref"] = 0.5 + (torch.rand(A, 3, generator=g) - 0.5) * torch.tensor([0.08, 0.08, 0.4])   # +-12 m in x/y, +-2 m in z
    return c


def hip_aggregate(case, device, value_dtype=torch.float32, variant=0, perm=None):
    from far3d_amd import ops
    d = lambda t: t.to(device).contiguous()
    return ops.aggregate_forward(d(case["feat"].to(value_dtype)), d(case["ref"]), d(case["offsets"]),
                                 d(case["lidar2img"]), d(case["U"]), d(case["Vc"]), case["level_hw"],
                                 case["level_start"], case["pc_range"], case["pad_hw"], num_groups=case["G"], variant=variant,
                                 perm=perm)


def run_aggregate_case(case, device, value_dtype=torch.float32, variant=0):
    want = oracle_aggregate(case, value_dtype)
    got = hip_aggregate(case, device, value_dtype, variant).float().cpu()
    return (got - want).abs().max().item()


def msda_case(bs=2, Q=19, H=4, Dh=8, hw=((6, 9), (3, 5), (2, 2)), P=3, seed=0, spread=0.3):
    g = torch.Generator().manual_seed(seed)
    starts, S = synth.level_starts(hw)
    L = len(hw)
    loc = torch.rand(bs, Q, H, L, P, 2, generator=g) * (1 + 2 * spread) - spread  # some samples fall outside
    w = torch.rand(bs, Q, H, L, P, generator=g)
    return dict(value=torch.randn(bs, S, H, Dh, generator=g),
                shapes=torch.tensor([list(x) for x in hw], dtype=torch.long),
                lsi=torch.tensor(starts, dtype=torch.long), loc=loc, w=w)
