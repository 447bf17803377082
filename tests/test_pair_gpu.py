"""GPU: pair-stored activations (FAR3D_DT_BF16_PAIR: [32 hi | 32 lo] bf16 per 32-channel block) through the C ABI.

The pair format is how the "bf16x3" precision mode keeps fp32 data on the LDS-DMA pipelined conv kernels: every product is
hi*hi' + hi*lo' + lo*hi' on the bf16 MFMA with fp32 accumulation.  Each operand keeps 16 significant bits and the dropped lo*lo'
term is 2^-16 relative, so a K-term dot product of O(1) terms deviates from the exact one by ~2^-16 * sqrt(K) * |term|: asserted
at a small multiple of that, ~100x tighter than one bf16 product term could pass -- the test fails if any of the three partial
products, a plane of the patch / weight ring, or a pixel of the halo is wrong.  References are float64 on the CPU.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

TILES_3X3 = [0, 150, 152, 153, 154, 155, 157, 159, 160, 161, 162, 163, 164, 165, 166, 167, 168, 169, 190, 198, 191, 192, 193, 197, 1, 3, 5]
TILES_1X1 = [0, 170, 171, 172, 173, 174, 175, 176, 177, 178, 179, 180, 181, 185, 186, 187, 188, 2, 4]


def _bound(K, x, w):
    scale = (x.abs().mean() * w.abs().mean()).item()
    return 4 * 2.0 ** -16 * K ** 0.5 * scale * 3, 2.0 ** -9 * K ** 0.5 * scale


def test_pair_roundtrip_and_layout():
    from far3d_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 5, 96, generator=g) * torch.logspace(-3, 3, 96)
    p = ops.pair_from_float(x)
    assert p.dtype == torch.bfloat16 and tuple(p.shape) == (2, 3, 5, 192)
    back = ops.pair_to_float(p)
    assert ((back - x).abs() <= x.abs() * 2.0 ** -16).all()             # 16 significant bits
    # block layout: stored channels [64b, 64b+32) are bf16(x[32b:32b+32]), the next 32 the residuals
    assert torch.equal(p[..., 64:96].float(), x[..., 32:64].to(torch.bfloat16).float())
    assert torch.equal(p[..., 96:128].float(), (x[..., 32:64] - x[..., 32:64].to(torch.bfloat16).float()).to(torch.bfloat16).float())


@pytest.mark.parametrize("tile", TILES_3X3)
def test_pair_conv3x3_split_products(hip_lib, tile):
    """3x3/s1/p1 on pair-stored maps: ragged sizes (W % 32, H % TH, Cout % BM), input AND output channel slices of wider buffers
    (the OSA concat layout), pair output through the LDS-transposed 16-byte row stores; neighbours must stay untouched."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(tile)
    for (N, Cin, Cout, H, W) in ((2, 64, 224, 13, 45), (1, 96, 160, 20, 30), (3, 32, 64, 9, 70)):
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1).relu()
        pc = ops.PackedConv(w, b, stride=1, pad=1, dtype=torch.float32, device=DEV, compute="bf16x3")
        xin = torch.full((N, H, W, 2 * (Cin + 64)), 3.0, dtype=torch.bfloat16, device=DEV)
        xin[..., 64:64 + 2 * Cin] = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)     # logical channels [32, 32 + Cin)
        buf = torch.full((N, H, W, 2 * (Cout + 64)), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.conv2d_nhwc(xin[..., 64:64 + 2 * Cin], pc, out=buf[..., 128:128 + 2 * Cout], act="relu", tile=tile)
        got = ops.pair_to_float(buf[..., 128:128 + 2 * Cout]).cpu().permute(0, 3, 1, 2)
        bound, bf16_level = _bound(Cin * 9, x, w)
        err = (got.double() - want).abs().max().item()
        assert err < bound + want.abs().max().item() * 2.0 ** -16, (tile, err, bound)       # + the 16-bit storage of y
        assert bound < 0.1 * bf16_level
        assert (buf[..., :128] == 7.0).all() and (buf[..., 128 + 2 * Cout:] == 7.0).all()
        # f32 output (direct epilogue)
        y32 = ops.conv2d_nhwc(xin[..., 64:64 + 2 * Cin], pc, act="relu", out_dtype=torch.float32, tile=tile)
        assert tuple(y32.shape) == (N, H, W, Cout)
        assert (y32.cpu().permute(0, 3, 1, 2).double() - want).abs().max().item() < bound


TILES_WS = [400, 401, 402, 403, 404, 405, 406, 407, 408, 409, 410, 411, 412, 413, 414, 415, 416, 417, 418, 419, 440, 444, 445, 450, 451, 452, 453, 454, 455, 456, 457, 458, 459]


@pytest.mark.parametrize("tile", TILES_WS)
def test_pair_conv3x3_wave_specialised_persistent_kernel(hip_lib, tile):
    """The persistent wave-specialised 3x3 kernel (csrc/conv_ws.hpp: producer waves issue every LDS-DMA, consumer waves only read LDS and
    run MFMAs, a workgroup walks several tiles as one stream of steps, register epilogue with v_permlane32_swap 16-byte stores): ragged
    sizes (W % 32, H % TH, Cout % BM incl. a channel tile past the packed weight rows), channel slices of wider buffers on both sides,
    more tiles than workgroups' first round (the persistent loop and its cross-tile prefetch), Cin of one chunk and of many.  Checked
    against float64 AND bit for bit against the shipped tile 163 (same products, same order)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(tile)
    for (N, Cin, Cout, H, W, act) in ((2, 64, 224, 13, 45, "relu"), (1, 96, 160, 20, 30, "relu"), (3, 32, 64, 9, 70, None), (7, 128, 128, 160, 240, "swish"),
                                      (2, 160, 192, 40, 60, "relu")):
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        pc = ops.PackedConv(w, b, stride=1, pad=1, dtype=torch.float32, device=DEV, compute="bf16x3")
        xin = torch.full((N, H, W, 2 * (Cin + 64)), 3.0, dtype=torch.bfloat16, device=DEV)
        xin[..., 64:64 + 2 * Cin] = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)     # logical channels [32, 32 + Cin)
        bufs = []
        for tl in (tile, 163):
            buf = torch.full((N, H, W, 2 * (Cout + 64)), 7.0, dtype=torch.bfloat16, device=DEV)
            ops.conv2d_nhwc(xin[..., 64:64 + 2 * Cin], pc, out=buf[..., 128:128 + 2 * Cout], act=act, tile=tl)
            bufs.append(buf)
        buf = bufs[0]
        assert (buf[..., :128] == 7.0).all() and (buf[..., 128 + 2 * Cout:] == 7.0).all()
        assert torch.equal(bufs[0], bufs[1]), "tile %d differs from tile 163 on %s (max %.3e)" % (
            tile, (N, Cin, Cout, H, W), (ops.pair_to_float(bufs[0][..., 128:128 + 2 * Cout].contiguous()) -
                                         ops.pair_to_float(bufs[1][..., 128:128 + 2 * Cout].contiguous())).abs().max().item())
        if N * H * W <= 4096:
            want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
            want = want.relu() if act == "relu" else (want * torch.sigmoid(want) if act == "swish" else want)
            got = ops.pair_to_float(buf[..., 128:128 + 2 * Cout].contiguous()).cpu().permute(0, 3, 1, 2)
            bound, bf16_level = _bound(Cin * 9, x, w)
            err = (got.double() - want).abs().max().item()
            assert err < bound + want.abs().max().item() * 2.0 ** -16, (tile, err, bound)
    # what the kernel does not do is an error of the call, never a silent fallback: f32 output, residual
    x = ops.pair_from_float(torch.randn(1, 8, 32, 32)).to(DEV)
    pc = ops.PackedConv(torch.randn(32, 32, 3, 3), None, stride=1, pad=1, dtype=torch.float32, device=DEV, compute="bf16x3")
    with pytest.raises(Exception):
        ops.conv2d_nhwc(x, pc, out=torch.empty(1, 8, 32, 32, device=DEV), tile=tile)


@pytest.mark.parametrize("tile", TILES_1X1)
def test_pair_conv1x1_split_products(hip_lib, tile):
    """1x1 GEMM over K = 160 / 1056 / 32 logical channels (odd and single step counts), ragged pixel / channel tiles."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(300 + tile)
    for (Cin, Cout, H, W) in ((160, 96, 7, 19), (1056, 512, 5, 9), (32, 64, 23, 31), (64, 288, 4, 5)):
        x = torch.randn(2, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = F.conv2d(x.double(), w.double(), b.double()).relu()
        pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
        xin = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)
        got = ops.pair_to_float(ops.conv2d_nhwc(xin, pc, act="relu", tile=tile)).cpu().permute(0, 3, 1, 2)
        bound, _ = _bound(Cin, x, w)
        assert (got.double() - want).abs().max().item() < bound + want.abs().max().item() * 2.0 ** -16, tile
        y32 = ops.conv2d_nhwc(xin, pc, act="relu", out_dtype=torch.float32, tile=tile).cpu().permute(0, 3, 1, 2)
        assert (y32.double() - want).abs().max().item() < bound, tile


TILES_WS_GEMM = [460, 461, 462, 463, 464, 465, 466, 467, 468, 469, 470, 471, 473, 474, 475, 476]


@pytest.mark.parametrize("tile", TILES_WS_GEMM)
def test_pair_gemm_wave_specialised_persistent_kernel(hip_lib, tile):
    """The persistent wave-specialised 1x1 GEMM (csrc/conv_ws.hpp gemm1x1_ws_kernel): producer waves stream weights AND activation rows
    through one LDS ring, a workgroup walks (pixel tile, channel tile) items as one stream of steps whose hand-over groups straddle tiles
    (odd step counts), pixel tiles that span two images, ragged pixel / channel tiles, more items than the grid's first round, channel
    slices of wider buffers.  Bit for bit the shipped tile 179 (same products, same order) and within the split-product bound of float64."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(tile)
    for (N, Cin, Cout, H, W, act) in ((2, 160, 96, 7, 19, "relu"), (2, 1056, 512, 5, 9, "relu"), (3, 32, 64, 23, 31, None), (7, 768, 256, 40, 60, "relu"),
                                      (2, 96, 288, 150, 161, "swish"), (3, 224, 160, 17, 23, "relu"), (5, 192, 64, 16, 16, None)):
        x = torch.randn(N, Cin, H, W, generator=g)
        if Cout == 64:
            x[:, :, :3] *= 40.0                     # stored values beyond 64: the 64-bit escape of the channel sums
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
        xin = torch.full((N, H, W, 2 * (Cin + 64)), 3.0, dtype=torch.bfloat16, device=DEV)
        xin[..., 64:64 + 2 * Cin] = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)
        # channel sums (the eSE pooling fused into the concat layers): maps of at least one 256-pixel tile, K of at least six steps
        with_sums = H * W >= 256 and Cin >= 192
        bufs, sums = [], []
        for tl in (tile, 179):
            buf = torch.full((N, H, W, 2 * (Cout + 64)), 7.0, dtype=torch.bfloat16, device=DEV)
            sm = torch.zeros(N, Cout, dtype=torch.int64, device=DEV) if with_sums else None
            ops.conv2d_nhwc(xin[..., 64:64 + 2 * Cin], pc, out=buf[..., 128:128 + 2 * Cout], act=act, tile=tl, sums=sm)
            if with_sums:
                ops.conv2d_nhwc(xin[..., 64:64 + 2 * Cin], pc, out=buf[..., 128:128 + 2 * Cout], act=act, tile=tl, sums=sm)      # sums accumulate
            bufs.append(buf)
            sums.append(sm)
        if with_sums:
            assert sums[0].abs().sum().item() > 0 and torch.equal(sums[0], sums[1]), "tile %d: channel sums differ from tile 179 on %s (%d entries)" % (
                tile, (N, Cin, Cout, H, W), (sums[0] != sums[1]).sum().item())
        assert (bufs[0][..., :128] == 7.0).all() and (bufs[0][..., 128 + 2 * Cout:] == 7.0).all()
        assert torch.equal(bufs[0], bufs[1]), "tile %d differs from tile 179 on %s (max %.3e)" % (
            tile, (N, Cin, Cout, H, W), (ops.pair_to_float(bufs[0][..., 128:128 + 2 * Cout].contiguous()) -
                                         ops.pair_to_float(bufs[1][..., 128:128 + 2 * Cout].contiguous())).abs().max().item())
        if N * H * W <= 4096:
            want = F.conv2d(x.double(), w.double(), b.double())
            want = want.relu() if act == "relu" else (want * torch.sigmoid(want) if act == "swish" else want)
            got = ops.pair_to_float(bufs[0][..., 128:128 + 2 * Cout].contiguous()).cpu().permute(0, 3, 1, 2)
            bound, _ = _bound(Cin, x, w)
            assert (got.double() - want).abs().max().item() < bound + want.abs().max().item() * 2.0 ** -16, tile
    # what the kernel does not do is an error of the call: f32 output, channel sums
    x = ops.pair_from_float(torch.randn(1, 8, 32, 32)).to(DEV)
    pc = ops.PackedConv(torch.randn(32, 32, 1, 1), None, dtype=torch.float32, device=DEV, compute="bf16x3")
    with pytest.raises(Exception):
        ops.conv2d_nhwc(x, pc, out=torch.empty(1, 8, 32, 32, device=DEV), tile=tile)
    with pytest.raises(Exception):       # a map smaller than the pixel tile takes no channel sums
        ops.conv2d_nhwc(x[:, :4, :4], pc, tile=tile, sums=torch.zeros(1, 32, dtype=torch.int64, device=DEV))


def test_pair_conv_small_cout_f32_heads_and_swish(hip_lib):
    """The 2D head's last 1x1 convs: Cout = 26 / 5 / 51 (f32 outputs, padded weight rows); Swish towers with pair output."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 10, 15, generator=g)
    xin = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)
    for Cout in (26, 5, 51):
        w = torch.randn(Cout, 256, 1, 1, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
        got = ops.conv2d_nhwc(xin, pc, out_dtype=torch.float32).cpu().permute(0, 3, 1, 2)
        want = F.conv2d(x.double(), w.double(), b.double())
        assert (got.double() - want).abs().max().item() < _bound(256, x, w)[0]
    w = torch.randn(512, 256, 3, 3, generator=g) * 0.03
    b = torch.randn(512, generator=g)
    pc = ops.PackedConv(w, b, stride=1, pad=1, dtype=torch.float32, device=DEV, compute="bf16x3")
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    want = y * torch.sigmoid(y)
    got = ops.pair_to_float(ops.conv2d_nhwc(xin, pc, act="swish")).cpu().permute(0, 3, 1, 2)
    assert (got.double() - want).abs().max().item() < _bound(2304, x, w)[0] + want.abs().max().item() * 2.0 ** -15


def test_pair_conv_stride2(hip_lib):
    """Strided convs (stem 3, the extra FPN level) with pair input: the register-staged kernel (1-5) and the LDS-patch kernel (330, 331)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(12)
    N, Cin, Cout, H, W = 2, 64, 128, 21, 30
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1).relu()
    pc = ops.PackedConv(w, b, stride=2, pad=1, dtype=torch.float32, device=DEV, compute="bf16x3")
    xin = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)
    for tile in (0, 1, 2, 3, 4, 5, 330, 331):          # 330 / 331: the LDS-patch kernel with de-interleaved patch rows (round 5)
        got = ops.pair_to_float(ops.conv2d_nhwc(xin, pc, act="relu", tile=tile)).cpu().permute(0, 3, 1, 2)
        assert tuple(got.shape) == (N, Cout, 11, 15)
        assert (got.double() - want).abs().max().item() < _bound(Cin * 9, x, w)[0] + want.abs().max().item() * 2.0 ** -16, tile


def test_pair_fpn_style_residual_and_token_output(hip_lib):
    """FPN lateral: pair-stored upsampled residual; FPN output conv: pair raw map + modulated f32 token-major second output."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(5)
    N, Cin, C, H, W = 2, 96, 256, 10, 14
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(C, Cin, 1, 1, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    coarse = torch.randn(N, C, 5, 7, generator=g)
    gamma, beta = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    coarse_p = ops.pair_from_float(coarse.permute(0, 2, 3, 1))
    lat = F.conv2d(x.double(), w.double(), b.double()) + F.interpolate(ops.pair_to_float(coarse_p).permute(0, 3, 1, 2).double(), size=(H, W), mode="nearest")
    want2 = gamma[:, :, None, None].double() * lat + beta[:, :, None, None].double()
    pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
    S = H * W + 11
    tokens = torch.zeros(N, S, C, device=DEV)
    y2 = tokens[:, 11:].view(N, H, W, C)
    xin = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)
    y = ops.conv2d_nhwc(xin, pc, res=coarse_p.to(DEV), y2=y2, y2_scale=gamma.to(DEV), y2_shift=beta.to(DEV))
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (N, H, W, 2 * C)
    bound = _bound(Cin, x, w)[0]
    assert (ops.pair_to_float(y).cpu().permute(0, 3, 1, 2).double() - lat).abs().max().item() < bound + lat.abs().max().item() * 2.0 ** -16
    assert (tokens[:, 11:].cpu().view(N, H, W, C).permute(0, 3, 1, 2).double() - want2).abs().max().item() < 4 * bound
    assert tokens[:, :11].abs().max().item() == 0


@pytest.mark.parametrize("tile3,tile1", [(260, 279), (265, 280), (252, 279)])
def test_pair_hi_only_layers_are_single_bf16_products(hip_lib, tile3, tile1):
    """terms = 1: the hi planes only -- exactly the bf16 product of bf16-rounded operands, written back in pair storage."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(tile3)
    N, Cin, Cout, H, W = 2, 64, 96, 11, 37
    x = torch.randn(N, Cin, H, W, generator=g)
    xin = ops.pair_from_float(x.permute(0, 2, 3, 1)).to(DEV)
    xb = x.to(torch.bfloat16).double()
    for k, tile in ((3, tile3), (1, tile1)):
        w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = F.conv2d(xb, w.to(torch.bfloat16).double(), b.double(), padding=k // 2).relu()
        pc = ops.PackedConv(w, b, stride=1, pad=k // 2, dtype=torch.float32, device=DEV, compute="bf16x3")
        pc.terms = 1
        assert ops.conv_tile(xin, pc) in (260, 279)
        got = ops.pair_to_float(ops.conv2d_nhwc(xin, pc, act="relu", tile=tile)).cpu().permute(0, 3, 1, 2)
        tol = 2e-6 * (Cin * k * k) ** 0.5 * max(1.0, want.abs().max().item()) + 1e-5 + want.abs().max().item() * 2.0 ** -16
        assert (got.double() - want).abs().max().item() < tol, (k, tile)
        got0 = ops.pair_to_float(ops.conv2d_nhwc(xin, pc, act="relu")).cpu().permute(0, 3, 1, 2)     # tile chosen through pc.terms
        assert (got0.double() - want).abs().max().item() < tol


def test_pair_elementwise_stages(hip_lib):
    """eSE (+identity, channel slices), GroupNorm+ReLU, MaxPool(3,2,ceil) and the stem im2col on pair storage vs float64."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(1)
    N, H, W, C = 3, 13, 17, 256
    x = torch.randn(N, H, W, C, generator=g)
    idn = torch.randn(N, H, W, C + 64, generator=g)
    fcw, fcb = torch.randn(C, C, generator=g) * 0.05, torch.randn(C, generator=g)
    xp, ip = ops.pair_from_float(x), ops.pair_from_float(idn)
    xr, ir = ops.pair_to_float(xp).double(), ops.pair_to_float(ip).double()
    gate = F.relu6(xr.mean(dim=(1, 2)) @ fcw.double().t() + fcb.double() + 3.0) / 6.0
    want = xr * gate[:, None, None, :] + ir[..., :C]
    out = torch.full((N, H, W, 2 * C + 128), 5.0, dtype=torch.bfloat16, device=DEV)
    ops.ese_nhwc(xp.to(DEV), fcw.to(DEV), fcb.to(DEV), identity=ip.to(DEV)[..., :2 * C], out=out[..., 64:64 + 2 * C], pair=True)
    got = ops.pair_to_float(out[..., 64:64 + 2 * C]).cpu().double()
    assert (got - want).abs().max().item() < 2e-5 + want.abs().max().item() * 2.0 ** -16
    assert (out[..., :64] == 5.0).all() and (out[..., 64 + 2 * C:] == 5.0).all()
    # GroupNorm + ReLU
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    wantg = F.group_norm(xr.permute(0, 3, 1, 2), 32, w.double(), b.double(), 1e-5).relu().permute(0, 2, 3, 1)
    gotg = ops.pair_to_float(ops.groupnorm_nhwc(xp.to(DEV), w.to(DEV), b.to(DEV), pair=True)).cpu().double()
    assert (gotg - wantg).abs().max().item() < 5e-5 + wantg.abs().max().item() * 2.0 ** -16
    # MaxPool: the maximum of exactly representable values is re-split to the same two halves
    wantm = F.max_pool2d(xr.permute(0, 3, 1, 2), 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
    dst = torch.zeros(N, wantm.shape[1], wantm.shape[2], 2 * C + 64, dtype=torch.bfloat16, device=DEV)
    ops.maxpool3x3s2_nhwc(xp.to(DEV), out=dst[..., :2 * C], pair=True)
    assert torch.equal(ops.pair_to_float(dst[..., :2 * C]).cpu().double(), wantm)
    # stem im2col
    img = torch.randn(2, 3, 20, 34, generator=g)
    f32 = ops.stem_im2col(img.to(DEV), torch.float32).cpu()
    pr = ops.stem_im2col(img.to(DEV), pair=True)
    assert tuple(pr.shape) == (2, 10, 17, 64) and torch.equal(pr.cpu(), ops.pair_from_float(f32))


def test_pair_checks(hip_lib):
    from far3d_amd import lib, ops
    pc = ops.PackedConv(torch.randn(48, 32, 1, 1), None, dtype=torch.float32, device=DEV, compute="bf16x3")
    x = ops.pair_from_float(torch.randn(1, 4, 4, 32)).to(DEV)
    with pytest.raises(lib.Far3dHipError):      # pair output needs Cout % 32 == 0
        ops.conv2d_nhwc(x, pc)
    assert tuple(ops.conv2d_nhwc(x, pc, out_dtype=torch.float32).shape) == (1, 4, 4, 48)
    with pytest.raises(ValueError):
        ops.pair_from_float(torch.randn(2, 40))
