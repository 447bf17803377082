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


@pytest.mark.parametrize("tile", [0, 170, 173, 179, 180, 181, 185, 186, 187, 188])
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


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("geom", [(3, 20, 30, 256, True), (2, 21, 29, 768, True), (1, 9, 17, 64, False), (7, 40, 60, 512, True)])
def test_fused_ese_pool_is_bitwise_the_apply_and_pool_launches(hip_lib, pair, geom):
    """far3d_ese_fused_nhwc (the gate launch + ONE pass that applies the gates in 16-byte pieces and writes the stage-end max-pool) against
    far3d_ese_nhwc(chan_sums) followed by far3d_maxpool3x3s2_nhwc: same gate arithmetic, taps rounded to the storage format before the
    maximum -> BIT-identical maps; the sums come back zero; repeated launches stay identical; channel slices of wider buffers on every
    operand; the pooled-only form (stage 2 inside the detector) writes no full-resolution map."""
    from far3d_amd import ops
    N, H, W, C, with_idn = geom
    g = torch.Generator().manual_seed(11 + pair + C)
    cs = 2 if pair else 1
    x = torch.randn(N, H, W, C, generator=g).relu()
    idn = torch.randn(N, H, W, C, generator=g) if with_idn else None
    fcw = (torch.randn(C, C, generator=g) * 0.05).to(DEV)
    fcb = (torch.randn(C, generator=g) * 0.1).to(DEV)
    enc = (lambda t: ops.pair_from_float(t)) if pair else (lambda t: t.to(torch.bfloat16))
    xbuf = torch.zeros(N, H, W, (C + 32) * cs, dtype=torch.bfloat16, device=DEV)          # x = channels [32, 32 + C) of a wider buffer
    xd = xbuf[..., 32 * cs:]
    xd.copy_(enc(x).to(DEV))
    ibuf = torch.zeros(N, H, W, (C + 64) * cs, dtype=torch.bfloat16, device=DEV)
    idn_d = None
    if with_idn:
        idn_d = ibuf[..., :C * cs]
        idn_d.copy_(enc(idn).to(DEV))
    # the channel sums the concat convolution would have accumulated: rint(v * 2^18) of every STORED element
    def sums_of(t):
        q = torch.round(t.double() * 2 ** 18).to(torch.int64).sum(dim=(1, 2))
        return (q.reshape(N, -1, 2, 32).sum(dim=2).reshape(N, C) if pair else q).contiguous()
    s0 = sums_of(xd)
    scratch = torch.empty(ops.ese_scratch_floats(N, C), dtype=torch.float32, device=DEV)
    sums = s0.clone()
    want = ops.ese_nhwc(xd, fcw, fcb, identity=idn_d, pair=pair, sums=sums, scratch=scratch)
    want_pool = ops.maxpool3x3s2_nhwc(want, pair=pair)
    Hp, Wp = ops.maxpool_out_hw(H, W)
    assert tuple(want_pool.shape[1:3]) == (Hp, Wp)
    gate = torch.empty(N * C, dtype=torch.float32, device=DEV)
    obuf = torch.full((N, H, W, (C + 32) * cs), 7.0, dtype=torch.bfloat16, device=DEV)
    pbuf = torch.full((N, Hp, Wp, (C + 96) * cs), 7.0, dtype=torch.bfloat16, device=DEV)
    for rep in range(3):
        sums = s0.clone()
        ops.ese_fused_nhwc(xd, fcw, fcb, sums, gate, identity=idn_d, out=obuf[..., :C * cs], pooled=pbuf[..., 32 * cs:(32 + C) * cs], pair=pair)
        assert torch.equal(obuf[..., :C * cs], want), "rep %d: y differs" % rep
        assert torch.equal(pbuf[..., 32 * cs:(32 + C) * cs], want_pool), "rep %d: pooled map differs" % rep
        assert int(sums.abs().sum()) == 0, "the launch returns the sums to zero"
    assert bool((obuf[..., C * cs:] == 7.0).all()) and bool((pbuf[..., :32 * cs] == 7.0).all()) and bool((pbuf[..., (32 + C) * cs:] == 7.0).all())
    # y only, pooled only
    sums = s0.clone()
    y_only = ops.ese_fused_nhwc(xd, fcw, fcb, sums, gate, identity=idn_d, out=torch.empty_like(want), pair=pair)
    assert torch.equal(y_only, want)
    sums = s0.clone()
    p_only = ops.ese_fused_nhwc(xd, fcw, fcb, sums, gate, identity=idn_d, pooled=torch.empty_like(want_pool), pair=pair)
    assert torch.equal(p_only, want_pool)
    # as a hipGraph (the engine's steady state)
    sums = s0.clone()
    static_s = s0.clone()
    gr = torch.cuda.CUDAGraph()
    yg, pg = torch.empty_like(want), torch.empty_like(want_pool)
    with torch.cuda.graph(gr):
        sums.copy_(static_s)
        ops.ese_fused_nhwc(xd, fcw, fcb, sums, gate, identity=idn_d, out=yg, pooled=pg, pair=pair)
    for _ in range(3):
        yg.zero_(); pg.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg, want) and torch.equal(pg, want_pool)


def test_fused_ese_pool_refuses_what_it_cannot_do(hip_lib):
    from far3d_amd import ops, lib as flib
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=DEV)
    fcw, fcb = torch.zeros(64, 64, device=DEV), torch.zeros(64, device=DEV)
    sums, gate = torch.zeros(1, 64, dtype=torch.int64, device=DEV), torch.zeros(64, device=DEV)
    with pytest.raises(ValueError):
        ops.ese_fused_nhwc(x, fcw, fcb, sums, gate)                                   # nothing to write
    with pytest.raises(flib.Far3dHipError):
        ops.ese_fused_nhwc(x, fcw, fcb, sums, gate, pooled=torch.zeros(1, 3, 3, 64, dtype=torch.bfloat16, device=DEV))   # ceil-mode size is 2 x 2
    with pytest.raises(flib.Far3dHipError):
        ops.ese_fused_nhwc(x[..., 4:], fcw[:60, :60].contiguous(), fcb[:60], sums, gate, out=torch.zeros(1, 4, 4, 60, dtype=torch.bfloat16, device=DEV))   # 8-byte pieces
