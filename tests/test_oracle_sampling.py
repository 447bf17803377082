"""CPU: the two oracle statements of MSDA agree, and closed-form known answers hold (SURVEY §8(c) KATs)."""
import numpy as np
import torch

from oracle import sampling
from tests import cases


def test_grid_sample_vs_scalar_loops():
    for seed in range(3):
        c = cases.msda_case(seed=seed)
        a = sampling.msda_grid_sample(c["value"], c["shapes"], c["lsi"], c["loc"], c["w"])
        b = sampling.msda_loops(c["value"].numpy(), c["shapes"].tolist(), c["lsi"].tolist(), c["loc"].numpy(),
                                c["w"].numpy())
        assert np.abs(a.numpy() - b).max() < 2e-5


def test_known_answer_pixel_centre():
    # all weight on a sample at the centre of pixel (y=2,x=3) of a 4x6 level returns that pixel
    H, W, Dh = 4, 6, 8
    value = torch.randn(1, H * W, 1, Dh)
    loc = torch.tensor([(3 + 0.5) / W, (2 + 0.5) / H]).view(1, 1, 1, 1, 1, 2)
    w = torch.ones(1, 1, 1, 1, 1)
    out = sampling.msda_grid_sample(value, torch.tensor([[H, W]]), torch.tensor([0]), loc, w)
    assert torch.allclose(out[0, 0], value[0, 2 * W + 3, 0], atol=1e-6)


def test_known_answer_outside_is_zero_and_corner_is_quarter():
    H, W, Dh = 4, 6, 4
    value = torch.ones(1, H * W, 1, Dh)
    locs = torch.tensor([[-0.5, 0.5], [0.0, 0.0]]).view(1, 2, 1, 1, 1, 2)  # far outside; exact image corner
    w = torch.ones(1, 2, 1, 1, 1)
    out = sampling.msda_grid_sample(value, torch.tensor([[H, W]]), torch.tensor([0]), locs, w)
    assert out[0, 0].abs().max() == 0
    assert torch.allclose(out[0, 1], torch.full((Dh,), 0.25))  # only 1 of 4 corners inside, weight .5*.5


def test_aggregation_ref_linear_in_features():
    c = cases.small_aggregate_case(seed=1)
    a = cases.oracle_aggregate(c)
    c2 = dict(c)
    c2["feat"] = c["feat"] * 2.0
    assert torch.allclose(cases.oracle_aggregate(c2), 2 * a, atol=1e-4)
    assert a.abs().max() > 1e-3  # some camera actually sees the points


def test_tent_patch_form_equals_bilinear_corner_form():
    """The algorithm of the v4 HIP aggregation kernel (merge the bilinear taps of a (camera, level) per token through the
    tent form of the interpolation weights, clipped bounding patch, per-corner path for wide spreads) restated on the CPU:
    identical to the reference formulation up to fp32 re-association, for tight, wide and off-image point clouds."""
    seen = dict(patch=0, wide=0)
    for seed, std, hw, A in ((0, 4.0, (64, 96), 37), (1, 0.5, (64, 96), 37), (2, 12.0, (64, 96), 37), (3, 6.0, (256, 384), 12)):
        c = cases.aggregate_case(num_cams=3, pad_hw=hw, A=A, seed=seed, offset_std=std)
        if seed == 3:   # near the cameras: wide spreads on the fine levels
            c["ref"] = 0.5 + (torch.rand(A, 3, generator=torch.Generator().manual_seed(9)) - 0.5) * torch.tensor([0.06, 0.06, 0.3])
        want = cases.oracle_aggregate(c)
        logits = c["U"][:, None, :] + c["Vc"][None, :, :]
        got, st = sampling.aggregation_tent(c["feat"], c["ref"], c["offsets"], c["lidar2img"], logits, c["level_hw"], c["level_start"],
                                            c["pc_range"], c["pad_hw"])
        assert (got - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item()) + 2e-6
        for k in seen:
            seen[k] += st[k]
    assert seen["patch"] > 0 and seen["wide"] > 0, seen


def test_factored_softmax_is_the_softmax():
    """The identity aggregate_v8_kernel rests on (csrc/agg_tables.hpp): with logit[n][j] = U[j] + V[n][j],
        softmax_{n,j}(U + V)[n][j] = eU[j] * eV[n][j] / sum_j eU[j] * EV[j],
        mV[j] = max_n V[n][j],  eV[n][j] = exp(V[n][j] - mV[j]),  EV[j] = sum_n eV[n][j],  eU[j] = exp(U[j] + mV[j] - max_j(U[j] + mV[j])).
    The largest term is exactly 1, so it neither overflows nor underflows where the plain max-subtracted softmax does not -- checked
    on ordinary logits and on logits 200 apart (a naive exp(U) * exp(V) factorisation returns NaN there), in float32."""
    g = torch.Generator().manual_seed(0)
    for scale in (1.0, 30.0, 200.0):
        U = torch.randn(52, 8, generator=g) * scale                       # (L*P, G) query part
        V = torch.randn(7, 52, 8, generator=g) * scale                    # (N, L*P, G) camera part
        want = (U[None] + V).double().reshape(-1, 8).softmax(0).reshape(7, 52, 8)
        mV = V.max(0).values
        eV = torch.exp(V - mV)
        EV = eV.sum(0)
        Up = U + mV
        eU = torch.exp(Up - Up.max(0).values)
        S = (eU * EV).sum(0)
        got = eU[None] * eV / S
        assert torch.isfinite(got).all() and (S >= 1).all()
        assert (got.double() - want).abs().max().item() < 5e-6, scale
        assert abs(got.double().sum().item() - 8.0) < 1e-4               # every group's weights sum to 1
    # the naive factorisation the kernel does NOT use: exp(U - max U) * exp(V - max V) underflows to 0 / 0
    U = torch.tensor([[200.0], [0.0]])
    V = torch.tensor([[[0.0], [200.0]]])
    naive = torch.exp(U - U.max()) * torch.exp(V - V.max())
    assert (naive == 0).all() or not torch.isfinite(naive / naive.sum()).all()
