"""GPU: implicit-GEMM conv / linear (C ABI far3d_conv2d_nhwc) vs torch CPU fp32 (F.conv2d / F.linear).

Tolerances.  fp32 weights -> exact-fp32 MFMA: 1e-4 abs / 1e-5 rel (reassociation only).  bf16 weights: inputs
and weights are rounded to bf16 ON BOTH SIDES, products are exact in fp32 and accumulation is fp32, so the
same reassociation-level tolerance applies (scaled by K)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rt(t, dt):
    return t.to(dt).float()


def _ref_conv(x_nchw, w, b, stride, pad, act, cdt):
    y = F.conv2d(_rt(x_nchw, cdt), _rt(w, cdt), b, stride=stride, padding=pad)
    if act == "relu":
        y = y.relu()
    elif act == "swish":
        y = y * torch.sigmoid(y)
    return y


def _close(got, want, K):
    tol = 2e-6 * K ** 0.5 * max(1.0, want.abs().max().item()) + 1e-5
    err = (got - want).abs().max().item()
    assert err < tol, (err, tol)


@pytest.mark.parametrize("xdt,wdt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 18, 43, 46, 48])
def test_conv3x3_relu(hip_lib, xdt, wdt, tile):
    from far3d_amd import ops
    if tile > 5 and not (xdt == torch.bfloat16 and wdt == torch.bfloat16):
        pytest.skip("LDS-DMA tiles are bf16-only")
    g = torch.Generator().manual_seed(1)
    N, Cin, Cout, H, W = 2, 64, 96, 17, 23
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    x = _rt(x, xdt)
    want = _ref_conv(x, w, b, 1, 1, "relu", wdt)
    pc = ops.PackedConv(w, b, stride=1, pad=1, dtype=wdt, device=DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(xdt).to(DEV)
    y = ops.conv2d_nhwc(xd, pc, act="relu", out_dtype=torch.float32, tile=tile)
    _close(y.cpu().permute(0, 3, 1, 2), want, Cin * 9)


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
def test_conv_stride2_and_swish_and_bf16_out(hip_lib, wdt):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(2)
    N, Cin, Cout, H, W = 3, 32, 64, 21, 30
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    want = _ref_conv(x, w, b, 2, 1, "swish", wdt)
    pc = ops.PackedConv(w, b, stride=2, pad=1, dtype=wdt, device=DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.conv2d_nhwc(xd, pc, act="swish", out_dtype=torch.float32)
    assert tuple(y.shape) == (N, 11, 15, Cout)
    _close(y.cpu().permute(0, 3, 1, 2), want, Cin * 9)
    yb = ops.conv2d_nhwc(xd, pc, act="swish", out_dtype=torch.bfloat16)
    assert (yb.float().cpu().permute(0, 3, 1, 2) - want).abs().max().item() < 0.02 * max(1.0, want.abs().max().item())


def test_conv1x1_reads_and_writes_channel_slices(hip_lib):
    """OSA concat-free layout: input = channels [32:96) of a 160-wide buffer, output = channels [96:160)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 9, 14
    buf = torch.randn(N, H, W, 160, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    want = F.conv2d(buf[..., 32:96].permute(0, 3, 1, 2), w, None, padding=1).relu()
    d = buf.to(DEV)
    pc = ops.PackedConv(w, None, stride=1, pad=1, dtype=torch.float32, device=DEV)
    keep = d.clone()
    ops.conv2d_nhwc(d[..., 32:96], pc, out=d[..., 96:160], act="relu")
    _close(d[..., 96:160].cpu().permute(0, 3, 1, 2), want, 64 * 9)
    assert torch.equal(d[..., :96], keep[..., :96])  # nothing outside the slice was touched


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
def test_linear_odd_sizes_with_residual(hip_lib, wdt):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(4)
    for (M, K, Nout) in ((37, 257, 256), (1544, 256, 39), (7, 14, 256), (130, 384, 416), (5, 180, 256)):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(Nout, K, generator=g) * 0.1
        b = torch.randn(Nout, generator=g)
        r = torch.randn(M, Nout, generator=g)
        want = F.linear(_rt(x, wdt), _rt(w, wdt), b) + r
        pc = ops.PackedConv(w, b, dtype=wdt, device=DEV)
        y = ops.linear(x.to(DEV), pc, res=r.to(DEV))
        _close(y.cpu(), want, K)
        y2 = ops.linear(x.to(DEV), pc, act="relu")
        _close(y2.cpu(), F.linear(_rt(x, wdt), _rt(w, wdt), b).relu(), K)


def test_fpn_style_upsampled_residual_and_modulated_second_output(hip_lib):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(5)
    N, Cin, C, H, W = 2, 96, 256, 10, 14
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(C, Cin, 1, 1, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    coarse = torch.randn(N, C, 5, 7, generator=g)
    gamma, beta = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    lat = F.conv2d(x, w, b) + F.interpolate(coarse, size=(H, W), mode="nearest")
    want2 = gamma[:, :, None, None] * lat + beta[:, :, None, None]
    pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV)
    S = H * W + 11
    tokens = torch.zeros(N, S, C, device=DEV)            # token-major value maps, this level starts at token 11
    y2 = tokens[:, 11:].view(N, H, W, C)
    y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), pc,
                        res=coarse.permute(0, 2, 3, 1).contiguous().to(DEV), y2=y2,
                        y2_scale=gamma.to(DEV), y2_shift=beta.to(DEV))
    _close(y.cpu().permute(0, 3, 1, 2), lat, Cin)
    _close(tokens[:, 11:].cpu().view(N, H, W, C).permute(0, 3, 1, 2), want2, Cin)
    assert tokens[:, :11].abs().max().item() == 0


def test_conv_asymmetric_weights_catch_transposes(hip_lib):
    """Identity-like probe with asymmetric weights (guide rule: transpose-detecting checks)."""
    from far3d_amd import ops
    Cin, Cout = 32, 64
    w = torch.zeros(Cout, Cin, 1, 1)
    for m in range(Cout):
        w[m, (m * 7 + 3) % Cin, 0, 0] = 1.0 + m
    x = torch.arange(5 * Cin, dtype=torch.float32).view(1, 1, 5, Cin) / 10.0
    for dt in (torch.float32, torch.bfloat16):
        pc = ops.PackedConv(w, None, dtype=dt, device=DEV)
        y = ops.conv2d_nhwc(x.to(DEV), pc, out_dtype=torch.float32).cpu()
        want = F.conv2d(_rt(x, dt).permute(0, 3, 1, 2), w).permute(0, 2, 3, 1)
        assert torch.allclose(y, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tile", [50, 51, 52, 53, 54, 55, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 90, 91, 92, 93, 94, 95, 96, 97, 100, 101, 102, 103, 104, 105, 106, 130, 131, 132, 133, 134, 135, 136, 137, 138, 139])
def test_conv3x3_pipelined_patch_kernel(hip_lib, tile):
    """3x3/s1/p1 bf16 kernel with the LDS-resident halo patch: ragged sizes (W % 32 != 0, H % TH != 0, Cout % BM != 0)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(tile)
    for (N, Cin, Cout, H, W) in ((2, 64, 200, 13, 45), (1, 96, 160, 20, 30), (3, 32, 64, 9, 70)):
        x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16).float()
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = _ref_conv(x, w, b, 1, 1, "relu", torch.bfloat16)
        pc = ops.PackedConv(w, b, stride=1, pad=1, dtype=torch.bfloat16, device=DEV)
        buf = torch.zeros(N, H, W, Cin + 32, dtype=torch.bfloat16, device=DEV)   # read a channel slice of a wider buffer
        buf[..., 16:16 + Cin] = x.permute(0, 2, 3, 1).to(torch.bfloat16).to(DEV)
        y = ops.conv2d_nhwc(buf[..., 16:16 + Cin], pc, act="relu", out_dtype=torch.float32, tile=tile)
        _close(y.cpu().permute(0, 3, 1, 2), want, Cin * 9)


@pytest.mark.parametrize("tile", [30, 31, 32, 33, 34, 35])
def test_conv3x3_stride2_patch_kernel(hip_lib, tile):
    """3x3 / stride 2 / pad 1 on the LDS-patch kernel (patch rows de-interleaved into even and odd input columns): odd and even input
    sizes (the last window column / row falls on the padding or not), ragged tiles, Cout % BM != 0, input read as a channel slice of a
    wider buffer, bf16 output through the coalesced row stores into a channel slice and fp32 output through the direct epilogue."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(900 + tile)
    for (N, Cin, Cout, H, W) in ((2, 64, 128, 22, 38), (1, 32, 200, 17, 67), (3, 96, 64, 9, 130), (1, 64, 128, 64, 96)):
        x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16).float()
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = _ref_conv(x, w, b, 2, 1, "relu", torch.bfloat16)
        pc = ops.PackedConv(w, b, stride=2, pad=1, dtype=torch.bfloat16, device=DEV)
        buf = torch.zeros(N, H, W, Cin + 32, dtype=torch.bfloat16, device=DEV)
        buf[..., 16:16 + Cin] = x.permute(0, 2, 3, 1).to(torch.bfloat16).to(DEV)
        y = ops.conv2d_nhwc(buf[..., 16:16 + Cin], pc, act="relu", out_dtype=torch.float32, tile=tile)
        assert tuple(y.shape) == (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout)
        _close(y.cpu().permute(0, 3, 1, 2), want, Cin * 9)
        ob = torch.full((N, y.shape[1], y.shape[2], Cout + 24), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.conv2d_nhwc(buf[..., 16:16 + Cin], pc, out=ob[..., 8:8 + Cout], act="relu", tile=tile)
        got = ob[..., 8:8 + Cout].float().cpu().permute(0, 3, 1, 2)
        assert (got - want).abs().max().item() < 2e-6 * (Cin * 9) ** 0.5 * max(1.0, want.abs().max().item()) + 1e-5 + want.abs().max().item() * 2 ** -8
        assert (ob[..., :8] == 7.0).all() and (ob[..., 8 + Cout:] == 7.0).all()


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 18, 43, 46, 48, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 110, 111, 112, 113, 114, 115, 116, 117, 120, 121, 122, 123, 124, 125, 126, 127, 128, 129, 140, 141, 142, 143, 144, 145])
def test_conv1x1_bf16_dma_partial_last_step(hip_lib, tile):
    """1x1 conv over K = 160 / 1056 / 2144 channels: K chunk counts (5, 33, 67) that do not divide the chunks-per-step."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(100 + tile)
    for (Cin, Cout, H, W) in ((160, 96, 7, 19), (1056, 512, 5, 9), (2144, 130, 3, 11), (32, 40, 23, 31), (64, 300, 4, 5)):
        x = torch.randn(2, Cin, H, W, generator=g).to(torch.bfloat16).float()
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = _ref_conv(x, w, b, 1, 0, "relu", torch.bfloat16)
        pc = ops.PackedConv(w, b, dtype=torch.bfloat16, device=DEV)
        y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV), pc, act="relu", out_dtype=torch.float32, tile=tile)
        _close(y.cpu().permute(0, 3, 1, 2), want, Cin)


@pytest.mark.parametrize("tile", [50, 52, 53, 57, 59, 60, 61, 62, 63, 64, 65, 66, 67, 90, 92, 93, 95, 100, 101, 102, 103, 104, 105, 106, 130, 131, 132, 133, 134, 135, 136, 137, 138, 139, 70, 71, 72, 73, 74, 75, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 110, 111, 112, 113, 114, 115, 116, 117, 120, 121, 122, 123, 124, 125, 126, 127, 128, 129, 140, 141, 142, 143, 144, 145])
def test_pipelined_kernels_bf16_output_coalesced_rows(hip_lib, tile):
    """bf16 outputs of the pipelined kernels leave through the LDS-transposed 16-byte row stores: ragged tiles
    (Cout % BM != 0, W % 32 != 0, pixel count % BP != 0), output written into a channel slice of a wider buffer whose
    neighbours must stay untouched."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(500 + tile)
    k = 1 if (70 <= tile < 90 or 110 <= tile < 130 or 140 <= tile < 150) else 3
    for (N, Cin, Cout, H, W) in ((2, 64, 200, 13, 45), (3, 96, 40, 9, 37), (1, 32, 64, 5, 33)):
        x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16).float()
        w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = _ref_conv(x, w, b, 1, k // 2, "relu", torch.bfloat16)
        pc = ops.PackedConv(w, b, stride=1, pad=k // 2, dtype=torch.bfloat16, device=DEV)
        buf = torch.full((N, H, W, Cout + 24), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV), pc, out=buf[..., 8:8 + Cout], act="relu", tile=tile)
        got = buf[..., 8:8 + Cout].float().cpu().permute(0, 3, 1, 2)
        tol = 2e-6 * (Cin * k * k) ** 0.5 * max(1.0, want.abs().max().item()) + 1e-5 + want.abs().max().item() * 2 ** -8   # + bf16 rounding of y
        assert (got - want).abs().max().item() < tol
        assert (buf[..., :8] == 7.0).all() and (buf[..., 8 + Cout:] == 7.0).all()


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", ["conv3x3", "stride2_odd_channels", "linear"])
def test_split_bf16_products_track_fp32(hip_lib, tile, case):
    """compute="bf16x3" (FAR3D_DT_F32_BF16X3): fp32 operands split as hi + lo bf16 while staging, three bf16 MFMAs per product
    (hi*hi' + hi*lo' + lo*hi', fp32 accumulation).  Each operand keeps 16 significant bits and the dropped lo*lo' term is 2^-16
    relative, so a K-term dot product of O(1) terms deviates from the exact one by ~2^-16 * sqrt(K) * |term| -- asserted at
    4x that, which is ~100x tighter than plain bf16 could pass and fails if any of the three partial products is missing."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(11)
    if case == "linear":
        M, Kd, Co = 333, 264, 200                     # K not a multiple of 32, odd row count
        x = torch.randn(M, Kd, generator=g)
        w = torch.randn(Co, Kd, generator=g) * 0.1
        b = torch.randn(Co, generator=g)
        want = F.linear(x.double(), w.double(), b.double())
        pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
        got = ops.linear(x.to(DEV), pc, tile=tile).cpu()
        K, scale = Kd, (x.abs().mean() * w.abs().mean()).item()
    else:
        stride = 2 if case == "stride2_odd_channels" else 1
        N, Cin, Cout, H, W = (2, 40, 72, 19, 26) if stride == 2 else (2, 64, 96, 17, 23)
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1).relu()
        pc = ops.PackedConv(w, b, stride=stride, pad=1, dtype=torch.float32, device=DEV, compute="bf16x3")
        got = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), pc, act="relu", tile=tile).cpu().permute(0, 3, 1, 2)
        K, scale = Cin * 9, (x.abs().mean() * w.abs().mean()).item()
    err = (got.double() - want).abs().max().item()
    bound = 4 * 2.0 ** -16 * K ** 0.5 * scale * 3      # 3 error sources of ~2^-17..2^-16 each per product
    bf16_level = 2.0 ** -9 * K ** 0.5 * scale
    assert err < bound, (err, bound)
    assert bound < 0.1 * bf16_level                   # the assertion is far below what one bf16 product term would give


@pytest.mark.parametrize("tile", [0, 479, 480, 481])
@pytest.mark.parametrize("shape", [(1544, 256, 256), (333, 96, 200), (2568, 1024, 256), (70, 32, 40)])
@pytest.mark.parametrize("epilogue", ["plain", "relu_residual_strided"])
def test_fp32_rows_on_the_pipelined_gemm_kernel(hip_lib, tile, shape, epilogue):
    """fp32 activation rows x pre-split weights, K % 32 == 0: the LDS-DMA pipelined GEMM kernel with the hi / lo split of the rows done
    in registers (tiles 479-481, auto) -- the decoder / FarHead GEMMs of the in-tolerance engine.  Same three products as the staged
    kernel: the same error bound as test_split_bf16_products_track_fp32; rows with a stride, a residual and an activation go through
    the generic epilogue."""
    from far3d_amd import ops
    M, Kd, Co = shape
    g = torch.Generator().manual_seed(M + Kd + Co)
    xs = torch.randn(M, Kd + 8, generator=g)
    x = xs[:, 4:4 + Kd]                               # row stride Kd + 8, 16-byte aligned start
    w = torch.randn(Co, Kd, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV, compute="bf16x3")
    xd = xs.to(DEV)[:, 4:4 + Kd]
    want = F.linear(x.double(), w.double(), b.double())
    if epilogue == "plain":
        got = ops.linear(xd, pc, tile=tile).cpu()
    else:
        res = torch.randn(M, Co, generator=g)
        want = want.relu() + res.double()
        got = ops.linear(xd, pc, act="relu", res=res.to(DEV), tile=tile).cpu()
    err = (got.double() - want).abs().max().item()
    scale = (x.abs().mean() * w.abs().mean()).item()
    bound = 4 * 2.0 ** -16 * Kd ** 0.5 * scale * 3
    assert err < bound, (err, bound)


@pytest.mark.parametrize("tile", [482, 483, 484, 485, 486, 487, 488, 489, 490, 491, 492, 493, 494])
@pytest.mark.parametrize("shape", [(1544, 256, 256), (333, 96, 200), (2568, 1024, 256), (70, 32, 40), (1544, 512, 455)])
@pytest.mark.parametrize("epilogue", ["plain", "relu_residual_strided"])
def test_exact_fp32_on_the_pipelined_gemm_kernel(hip_lib, tile, shape, epilogue):
    """fp32 rows x fp32 weights on the LDS-DMA pipelined GEMM kernel with the EXACT fp32 MFMA (tiles 482-486, round 6: the decoder /
    FarHead GEMMs of the in-tolerance and fp32 engines; 487-494: K groups inside the workgroup, partial tiles added in group order -- K of
    one step leaves groups without work, K of three steps uneven ones).  Exact products, fp32 accumulation: fp32 rounding noise against float64, and
    within that noise of the register-staged exact kernel (tile 0) -- only the summation order differs.  Ragged M / N, K of one to 32
    steps, strided rows, residual + activation through the generic epilogue; asymmetric operands (a transposed fragment would show)."""
    from far3d_amd import ops
    M, Kd, Co = shape
    g = torch.Generator().manual_seed(M + Kd + Co + tile)
    xs = torch.randn(M, Kd + 8, generator=g)
    x = xs[:, 4:4 + Kd]                               # row stride Kd + 8, 16-byte aligned start
    w = torch.randn(Co, Kd, generator=g) * 0.1 * torch.linspace(0.5, 1.5, Kd)[None]
    b = torch.randn(Co, generator=g)
    pc = ops.PackedConv(w, b, dtype=torch.float32, device=DEV)
    xd = xs.to(DEV)[:, 4:4 + Kd]
    want = F.linear(x.double(), w.double(), b.double())
    if epilogue == "plain":
        got, old = ops.linear(xd, pc, tile=tile).cpu(), ops.linear(xd, pc, tile=3).cpu()
    else:
        res = torch.randn(M, Co, generator=g)
        want = want.relu() + res.double()
        got, old = ops.linear(xd, pc, act="relu", res=res.to(DEV), tile=tile).cpu(), ops.linear(xd, pc, act="relu", res=res.to(DEV), tile=3).cpu()
    scale = (x.abs().mean() * w.abs().mean()).item()
    bound = 8 * 2.0 ** -24 * Kd ** 0.5 * max(1.0, want.abs().max().item()) + 4 * 2.0 ** -24 * Kd * scale
    assert (got.double() - want).abs().max().item() < bound, ((got.double() - want).abs().max().item(), bound)
    assert (got - old).abs().max().item() < bound
    if epilogue == "plain" and M > 200:
        # a row's bits depend on the tile and on K, not on which rows share the launch (the query-sharded decoder launches row subsets)
        sub = ops.linear(xd[37:37 + 130], pc, tile=tile).cpu()
        assert torch.equal(sub, got[37:37 + 130])


def test_split_mode_rejects_bf16_operands(hip_lib):
    from far3d_amd import lib, ops
    with pytest.raises(ValueError):
        ops.PackedConv(torch.randn(8, 8, 1, 1), None, dtype=torch.bfloat16, device=DEV, compute="bf16x3")
    pc = ops.PackedConv(torch.randn(32, 32, 1, 1), None, dtype=torch.float32, device=DEV, compute="bf16x3")
    with pytest.raises(ValueError):       # a bf16 tensor under split weights is PAIR storage (2 x 32 stored channels expected)
        ops.conv2d_nhwc(torch.randn(1, 4, 4, 32, device=DEV).bfloat16(), pc)
