"""GPU: fixed-point channel sums out of the GEMM epilogue (far3d_conv2d_nhwc chan_sums) and their consumer (far3d_ese_nhwc).

The sums are defined on the STORED output: every stored bf16 element v (a pair output: its hi and its lo half) contributes
rint(v * 2^18) to a 64-bit integer -- so the expectation is computed from the output tensor itself and the comparison is bit-exact,
whatever the tile shape, the image boundaries inside tiles or the order of the atomics."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FRAC = 18


def _expected(stored, C, pair):
    """stored: (N,H,W,Cs) bf16 output as written -> (N,C) int64."""
    v = stored.double().cpu()
    q = torch.round(v * 2.0 ** FRAC).to(torch.int64)        # round-half-even, like __float2int_rn
    q = q.sum(dim=(1, 2))
    if pair:
        q = q.reshape(q.shape[0], -1, 2, 32).sum(dim=2).reshape(q.shape[0], C)
    return q


@pytest.mark.parametrize("tile", [70, 73, 75, 79, 80, 81, 85, 110, 116, 120, 122, 123, 125, 128, 140, 142, 143, 145])
def test_bf16_gemm_tiles_accumulate_exact_channel_sums(hip_lib, tile):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(tile)
    # three images of 20 x 30 = 600 pixels: image boundaries fall inside pixel tiles; ragged channel tile (200), K with a half step (96)
    for (N, H, W, Cin, Cout, bias) in ((3, 20, 30, 96, 200, 0.0), (2, 24, 32, 64, 256, 0.0), (3, 20, 30, 32, 72, 90.0)):
        x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(DEV)
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
        b = torch.randn(Cout, generator=g) + bias          # bias 90: outputs far above 64 -> the 64-bit path of the epilogue
        pc = ops.PackedConv(w, b, dtype=torch.bfloat16, device=DEV)
        sums = torch.zeros(N, Cout, dtype=torch.int64, device=DEV)
        y = ops.conv2d_nhwc(x, pc, act="relu", tile=tile, sums=sums)
        y0 = ops.conv2d_nhwc(x, pc, act="relu", tile=tile)
        assert torch.equal(y, y0), "the stored output must not depend on the sums"
        assert torch.equal(sums.cpu(), _expected(y, Cout, False)), (tile, N, H, W, Cin, Cout)
        ops.conv2d_nhwc(x, pc, act="relu", tile=tile, sums=sums)       # sums accumulate
        assert torch.equal(sums.cpu(), 2 * _expected(y, Cout, False))


@pytest.mark.parametrize("tile", [0, 170, 173, 179, 180, 181])
def test_pair_gemm_tiles_accumulate_exact_channel_sums(hip_lib, tile):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(300 + tile)
    for (N, H, W, Cin, Cout) in ((3, 20, 30, 96, 160), (2, 24, 32, 64, 256)):
        x = ops.pair_from_float(torch.randn(N, H, W, Cin, generator=g)).to(DEV)
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
        b = torch.randn(Cout, generator=g)
        pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
        sums = torch.zeros(N, Cout, dtype=torch.int64, device=DEV)
        out = torch.empty(N, H, W, 2 * Cout, dtype=torch.bfloat16, device=DEV)
        ops.conv2d_nhwc(x, pc, out=out, act="relu", tile=tile, sums=sums)
        assert torch.equal(sums.cpu(), _expected(out, Cout, True)), (tile, N, H, W, Cin, Cout)


def test_channel_sums_are_refused_where_they_cannot_be_produced(hip_lib):
    from far3d_amd import ops
    x = torch.randn(2, 8, 8, 64).to(torch.bfloat16).to(DEV)          # 64 pixels per image: smaller than any GEMM tile
    pc = ops.PackedConv(torch.randn(64, 64, 1, 1) * 0.1, torch.zeros(64), dtype=torch.bfloat16, device=DEV)
    sums = torch.zeros(2, 64, dtype=torch.int64, device=DEV)
    assert not ops.conv_can_fuse_sums(x, pc)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, pc, tile=79, sums=sums)                   # map smaller than the tile
    x2 = torch.randn(1, 32, 32, 64).to(torch.bfloat16).to(DEV)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x2, pc, tile=1, sums=sums)                   # not a pipelined GEMM tile
    pc3 = ops.PackedConv(torch.randn(64, 64, 3, 3) * 0.1, torch.zeros(64), pad=1, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x2, pc3, tile=60, sums=sums)                 # 3x3 layer
    assert int(sums.abs().sum()) == 0


@pytest.mark.parametrize("pair", [False, True])
def test_ese_from_epilogue_sums_matches_ese_from_its_own_pooling(hip_lib, pair):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(7 + pair)
    N, H, W, Cin, C = 3, 20, 30, 160, 256
    xin = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(C, Cin, 1, 1, generator=g) * 0.08
    b = torch.randn(C, generator=g) * 0.1
    fcw = (torch.randn(C, C, generator=g) * 0.05).to(DEV)
    fcb = (torch.randn(C, generator=g) * 0.1).to(DEV)
    idn = torch.randn(N, H, W, C, generator=g)
    if pair:
        x = ops.pair_from_float(xin).to(DEV)
        pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
        cat_out = torch.empty(N, H, W, 2 * C, dtype=torch.bfloat16, device=DEV)
        idn_d = ops.pair_from_float(idn).to(DEV)
    else:
        x = xin.to(torch.bfloat16).to(DEV)
        pc = ops.PackedConv(w, b, dtype=torch.bfloat16, device=DEV)
        cat_out = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=DEV)
        idn_d = idn.to(torch.bfloat16).to(DEV)
    assert ops.conv_can_fuse_sums(x, pc) == pair      # an untuned bf16 shape has no measured GEMM tile; the pair default is one
    tile = 0 if pair else 79
    sums = torch.zeros(N, C, dtype=torch.int64, device=DEV)
    ops.conv2d_nhwc(x, pc, out=cat_out, act="relu", sums=sums, tile=tile)
    assert int(sums.abs().sum()) > 0
    fused = ops.ese_nhwc(cat_out, fcw, fcb, identity=idn_d, pair=pair, sums=sums)
    assert int(sums.abs().sum()) == 0, "the consumer returns the sums to zero"
    plain = ops.ese_nhwc(cat_out, fcw, fcb, identity=idn_d, pair=pair)
    f, p = (ops.pair_to_float(fused), ops.pair_to_float(plain)) if pair else (fused.float(), plain.float())
    # same map, same gate arithmetic; only the pooled means differ (fixed point 2^-18 per element vs fp32 partial sums)
    tol = 1e-5 * max(1.0, p.abs().max().item()) + (0.0 if pair else p.abs().max().item() * 2 ** -8)
    assert (f - p).abs().max().item() <= tol
    # run-to-run: bit-identical
    ops.conv2d_nhwc(x, pc, out=cat_out, act="relu", sums=sums, tile=tile)
    again = ops.ese_nhwc(cat_out, fcw, fcb, identity=idn_d, pair=pair, sums=sums)
    assert torch.equal(again, fused)
