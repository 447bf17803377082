"""CPU: the host side of pair storage (FAR3D_DT_BF16_PAIR) -- layout, exactness, weight packing, tile-id mapping.  The kernels
themselves are covered on the GPU by tests/test_pair_gpu.py."""
import torch

from far3d_amd import ops


def test_pair_storage_roundtrip_layout_and_precision():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 5, 96, generator=g) * torch.logspace(-3, 3, 96)
    p = ops.pair_from_float(x)
    assert p.dtype == torch.bfloat16 and tuple(p.shape) == (2, 3, 5, 192)
    back = ops.pair_to_float(p)
    assert ((back - x).abs() <= x.abs() * 2.0 ** -16).all()                       # 16 significant bits
    hi = x[..., 32:64].to(torch.bfloat16)
    assert torch.equal(p[..., 64:96], hi)                                          # block b: stored [64b, 64b+32) = bf16(x[32b:32b+32])
    assert torch.equal(p[..., 96:128], (x[..., 32:64] - hi.float()).to(torch.bfloat16))
    # hi + lo is exact in fp32, and a stored value survives decode -> re-split -> decode unchanged (what the max-pool store relies on;
    # the two halves themselves may differ when lo sits exactly on a rounding tie of hi)
    assert torch.equal(ops.pair_to_float(ops.pair_from_float(back)), back)
    # values that are already bf16 have a zero lo half
    assert (ops.pair_from_float(hi.float())[..., 32:] == 0).all()


def test_split_weights_are_the_pair_layout_of_the_rows():
    """PackedConv(compute='bf16x3') stores every 32-channel block of a weight row as [32 hi | 32 lo] -- the same layout as pair-stored
    activations, which is what lets the LDS-DMA kernels stream both operands as 64-byte planes."""
    g = torch.Generator().manual_seed(1)
    w = torch.randn(40, 64, 3, 3, generator=g)
    pc = ops.PackedConv(w, None, stride=1, pad=1, dtype=torch.float32, device="cpu", compute="bf16x3")
    assert pc.w.dtype == torch.bfloat16 and pc.w_code == ops.DT_F32_BF16X3 and pc.terms == 3
    rows = w.permute(0, 2, 3, 1).reshape(40, 9 * 64)                                # tap-major K
    assert torch.equal(pc.w[:40], ops.pair_from_float(rows))
    assert (pc.w[40:] == 0).all()                                                  # zero rows for over-reading channel tiles


def test_pair_tile_mapping():
    g = torch.Generator().manual_seed(2)
    pc3 = ops.PackedConv(torch.randn(64, 64, 3, 3, generator=g), None, stride=1, pad=1, dtype=torch.float32, device="cpu", compute="bf16x3")
    pc1 = ops.PackedConv(torch.randn(64, 64, 1, 1, generator=g), None, dtype=torch.float32, device="cpu", compute="bf16x3")
    pcs = ops.PackedConv(torch.randn(64, 64, 3, 3, generator=g), None, stride=2, pad=1, dtype=torch.float32, device="cpu", compute="bf16x3")
    assert ops._pair_tile(pc3, 64, 10 ** 9, 160) == 160 and ops._pair_tile(pc1, 64, 10 ** 9, 179) == 179
    pc3.terms = pc1.terms = pcs.terms = 1
    assert ops._pair_tile(pc3, 64, 10 ** 9, 160) == 260 and ops._pair_tile(pc3, 64, 10 ** 9, 163) == 260      # hi-only kernels
    assert ops._pair_tile(pc1, 64, 10 ** 9, 170) == 279 and ops._pair_tile(pc1, 64, 10 ** 9, 180) == 280
    assert ops._pair_tile(pcs, 64, 10 ** 9, 3) == 3                                 # strided convs keep the register-staged kernel
    x = torch.zeros(1, 4, 4, 128, dtype=torch.bfloat16)
    assert ops._is_pair_input(x, pc3) and not ops._is_pair_input(x.float(), pc3)
