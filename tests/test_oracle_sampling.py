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
