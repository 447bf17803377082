"""GPU: attention core, LayerNorm, eSE, GroupNorm, MaxPool (C ABI) vs torch CPU fp32."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mha_core(q, k, v, H):
    Aq, E = q.shape
    d = E // H
    qh = (q * d ** -0.5).view(Aq, H, d).transpose(0, 1)
    kh = k.view(-1, H, d).transpose(0, 1)
    vh = v.view(-1, H, d).transpose(0, 1)
    p = torch.softmax(qh @ kh.transpose(1, 2), dim=-1)
    return (p @ vh).transpose(0, 1).reshape(Aq, E)


@pytest.mark.parametrize("variant", [0, 14, 118, 4, 2])      # 0: the default (register-fed kernel); 14 / 118 its other shapes; 4 / 2 the LDS-staged kernel
@pytest.mark.parametrize("Aq,Nk", [(37, 50), (128, 64), (300, 333), (1544, 2312)])
def test_attention_fp32_exact_mode(hip_lib, Aq, Nk, variant):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(Aq)
    q, k, v = (torch.randn(n, 256, generator=g) for n in (Aq, Nk, Nk))
    q = q * 2.0   # make the softmax peaky enough to exercise the running-max rescale
    want = _mha_core(q, k, v, 8)
    prev = ops.attention_f32_variant(variant)
    try:
        got = ops.attention_forward(q.to(DEV), k.to(DEV), v.to(DEV)).cpu()
    finally:
        ops.attention_f32_variant(prev)
    assert ops.attention_f32_variant() == prev
    assert (got - want).abs().max().item() < 2e-5


def test_attention_rescale_branch_forced(hip_lib):
    """One key per query dominates and sits in a LATE tile, so the running max jumps (guide rule 26)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(7)
    Aq, Nk = 96, 400
    q, k, v = torch.randn(Aq, 256, generator=g), torch.randn(Nk, 256, generator=g), torch.randn(Nk, 256, generator=g)
    for i in range(Aq):
        k[100 + 3 * i] += 3.0 * q[i]      # spike in tiles 1..6
    want = _mha_core(q, k, v, 8)
    got = ops.attention_forward(q.to(DEV), k.to(DEV), v.to(DEV)).cpu()
    assert (got - want).abs().max().item() < 5e-5


def test_attention_bf16_mode(hip_lib):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(3)
    Aq, Nk = 1544, 2312
    q, k, v = (torch.randn(n, 256, generator=g).to(torch.bfloat16) for n in (Aq, Nk, Nk))
    want = _mha_core(q.float(), k.float(), v.float(), 8)
    got = ops.attention_forward(q.to(DEV), k.to(DEV), v.to(DEV)).cpu()
    # P is rounded to bf16 before the PV product (2^-9 relative per element, averaged by the sum)
    assert (got - want).abs().max().item() < 1e-2
    assert (got - want).abs().mean().item() < 1e-3


def test_attention_strided_views_of_one_buffer(hip_lib):
    """q/k/v produced by one GEMM into a (rows, 768) buffer are consumed in place."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(200, 768, generator=g)
    want = _mha_core(buf[:150, :256], buf[:, 256:512], buf[:, 512:], 8)
    d = buf.to(DEV)
    got = ops.attention_forward(d[:150, :256], d[:, 256:512], d[:, 512:]).cpu()
    assert (got - want).abs().max().item() < 2e-5


def test_layernorm_variants(hip_lib):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(0)
    for rows, C in ((1544, 256), (7, 256), (5, 512), (3, 1024)):
        x = torch.randn(rows, C, generator=g) * 3 + 1
        w, b, add = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(rows, C, generator=g)
        want = F.layer_norm(x, (C,), w, b, 1e-5)
        y, y2 = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), add=add.to(DEV))
        assert (y.cpu() - want).abs().max().item() < 2e-5
        assert (y2.cpu() - (want + add)).abs().max().item() < 2e-5
        yr = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), act="relu")
        assert (yr.cpu() - want.relu()).abs().max().item() < 2e-5
        yn = ops.layernorm(x.to(DEV), None, None)
        assert (yn.cpu() - F.layer_norm(x, (C,))).abs().max().item() < 2e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ese_with_identity_on_slices(hip_lib, dt):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(1)
    N, H, W, C = 3, 13, 17, 256
    x = torch.randn(N, H, W, C, generator=g).to(dt)
    idn = torch.randn(N, H, W, C + 64, generator=g).to(dt)
    fcw, fcb = torch.randn(C, C, generator=g) * 0.05, torch.randn(C, generator=g)
    mean = x.float().mean(dim=(1, 2))
    gate = F.relu6(mean @ fcw.t() + fcb + 3.0) / 6.0
    want = x.float() * gate[:, None, None, :] + idn.float()[..., :C]
    got = ops.ese_nhwc(x.to(DEV), fcw.to(DEV), fcb.to(DEV), identity=idn.to(DEV)[..., :C]).float().cpu()
    tol = 2e-5 if dt == torch.float32 else 0.03
    assert (got - want).abs().max().item() < tol
    got2 = ops.ese_nhwc(x.to(DEV), fcw.to(DEV), fcb.to(DEV)).float().cpu()
    assert (got2 - x.float() * gate[:, None, None, :]).abs().max().item() < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_groupnorm_relu(hip_lib, dt):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(2)
    N, H, W, C = 2, 20, 30, 256
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.5).to(dt)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    want = F.group_norm(x.float().permute(0, 3, 1, 2), 32, w, b, 1e-5).relu().permute(0, 2, 3, 1)
    got = ops.groupnorm_nhwc(x.to(DEV), w.to(DEV), b.to(DEV)).float().cpu()
    assert (got - want).abs().max().item() < (5e-5 if dt == torch.float32 else 0.05)


@pytest.mark.parametrize("hw", [(160, 240), (17, 23), (8, 9)])
def test_maxpool_ceil_mode(hip_lib, hw):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, hw[0], hw[1], 64, generator=g)
    want = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
    got = ops.maxpool3x3s2_nhwc(x.to(DEV)).cpu()
    assert got.shape == want.shape
    assert torch.equal(got, want)
    gb = ops.maxpool3x3s2_nhwc(x.to(torch.bfloat16).to(DEV)).float().cpu()
    wb = F.max_pool2d(x.to(torch.bfloat16).float().permute(0, 3, 1, 2), 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
    assert torch.equal(gb, wb)


def test_cam_embed_chain_matches_torch(hip_lib):
    """All decoder layers' camera term (cam_embed MLP + LayerNorm + weights_fc) in one launch vs torch fp32."""
    import torch.nn.functional as F
    from far3d_amd import ops
    g = torch.Generator().manual_seed(7)
    L, N, J = 3, 5, 416
    layers = []
    for _ in range(L):
        layers.append((torch.randn(128, 12, generator=g) * 0.3, torch.randn(128, generator=g) * 0.1, torch.randn(256, 128, generator=g) * 0.08,
                       torch.randn(256, generator=g) * 0.1, 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g),
                       torch.randn(J, 256, generator=g) * 0.05, torch.randn(J, generator=g) * 0.1))
    l2i = torch.randn(N, 12, generator=g)
    packed = ops.pack_cam_embed_chain(layers, "cuda:0")
    got = ops.cam_embed_chain(l2i.to("cuda:0"), packed).cpu()
    # the engine hands over lidar2img (N,4,4) itself: rows are read in place with stride 16
    l44 = torch.cat([l2i, torch.randn(N, 4, generator=g)], dim=1).view(N, 4, 4)
    assert torch.equal(ops.cam_embed_chain(l44.to("cuda:0"), packed).cpu(), got)
    for l, (w0, b0, w2, b2, lg, lb, w3, b3) in enumerate(layers):
        e = F.layer_norm(F.relu(F.linear(F.relu(F.linear(l2i, w0, b0)), w2, b2)), (256,), lg, lb)
        want = F.linear(e, w3, b3)
        assert (got[l] - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
