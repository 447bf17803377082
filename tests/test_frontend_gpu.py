"""GPU: 2D proposal selection (C ABI far3d_proposal_select) -- the static top-K mode against a sort of the kernel's own
peak-weight map (the threshold mode is pinned by the golden sequence in test_engine_gpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("K,hw", [(1, None), (7, None), (92, None), (92, [(80, 120), (40, 60), (20, 30), (10, 15)]), (200, [(80, 120), (40, 60), (20, 30), (10, 15)]),
                                  (300, [(40, 60), (20, 30), (10, 15), (5, 8)])])
def test_proposal_select_topk_matches_stable_sort(hip_lib, K, hw):
    """Static top-K per camera, ties towards the lower index, output in index order.  The small maps have fewer positive peaks than K = 92
    (the K-th place is a tie among zeros: the exact ordered-compaction path); the benchmark-size maps take the windowed bisection +
    counting ranks of round 5 (K <= 256) -- with a quantised camera so that tie groups straddle the cut -- and K = 300 the exact path again."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(3 + K)
    N, ncls = 3, 26
    hw = hw or [(16, 24), (8, 12), (4, 6), (2, 3)]
    cls = [(torch.randn(N, h, w, ncls, generator=g) * 2 - 1).to(DEV) for h, w in hw]
    reg = [torch.randn(N, h, w, 5, generator=g).to(DEV) for h, w in hw]
    # ties: quantise the logits of one camera so that many peaks share a weight
    cls = [c.clone() for c in cls]
    for c in cls:
        c[1] = torch.round(c[1] * 2) / 2
    reg = [r.clone() for r in reg]
    for r in reg:
        r[1, ..., 4] = torch.round(r[1, ..., 4] * 2) / 2
    wgt, sel_idx, sel_cnt = ops.proposal_select(cls, reg, (8, 16, 32, 64), K, thr=0.1, topk=True)
    w = wgt.cpu().numpy()
    idx = sel_idx.cpu().numpy()
    assert (sel_cnt.cpu().numpy() == K).all()
    for n in range(N):
        order = np.argsort(-w[n], kind="stable")[:K]          # largest first, ties -> lower index
        assert np.array_equal(idx[n], np.sort(order)), "camera %d" % n


@pytest.mark.parametrize("geom", [(2, 22, 38), (1, 64, 96), (3, 37, 131), (7, 640, 960)])
def test_stem_conv_from_the_image_is_bitwise_im2col_plus_gemm(hip_lib, geom):
    """far3d_stem_conv (VoVNet stem_1 read straight from the NCHW image, csrc/stem.hip) against far3d_stem_im2col + far3d_conv2d_nhwc:
    same bf16 products in the same order -> bit-identical; and against F.conv2d on bf16-rounded operands; odd sizes (Wo % 32 != 0, odd H / W),
    an output written into a channel slice of a wider buffer whose neighbours must stay untouched."""
    import torch.nn.functional as F
    from far3d_amd import ops
    N, H, W = geom
    g = torch.Generator().manual_seed(H * W)
    img = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g)
    pc = ops.PackedConv(F.pad(w.permute(0, 2, 3, 1).reshape(64, 27), (0, 5)), b, dtype=torch.bfloat16, device=DEV)
    d = img.to(DEV)
    want = ops.conv2d_nhwc(ops.stem_im2col(d, torch.bfloat16), pc, act="relu")
    got = ops.stem_conv(d, pc, act="relu")
    assert got.shape == want.shape and torch.equal(got, want)
    ref = F.conv2d(img.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), b, stride=2, padding=1).relu().permute(0, 2, 3, 1)
    assert (got.float().cpu() - ref).abs().max().item() < 2e-5 * 27 ** 0.5 * max(1.0, ref.abs().max().item()) + ref.abs().max().item() * 2 ** -8
    Ho, Wo = got.shape[1], got.shape[2]
    buf = torch.full((N, Ho, Wo, 96), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.stem_conv(d, pc, act=None, out=buf[..., 16:80])
    assert torch.equal(buf[..., 16:80], ops.conv2d_nhwc(ops.stem_im2col(d, torch.bfloat16), pc)) and bool((buf[..., :16] == 7.0).all()) and bool((buf[..., 80:] == 7.0).all())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_stem_im2col_matches_unfold(hip_lib, dt):
    """Stem conv input: (n, oy, ox, tap*3 + c) = img[n, c, 2*oy-1+ky, 2*ox-1+kx], zero padded, 5 zero pad channels."""
    import torch.nn.functional as F
    from far3d_amd import ops
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 22, 38
    img = torch.randn(N, 3, H, W, generator=g)
    got = ops.stem_im2col(img.to(DEV), dt).float().cpu()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cols = F.unfold(img, 3, padding=1, stride=2).view(N, 3, 9, Ho, Wo)          # (N, c, tap, oy, ox)
    want = torch.zeros(N, Ho, Wo, 32)
    want[..., :27] = cols.permute(0, 3, 4, 2, 1).reshape(N, Ho, Wo, 27)
    if dt == torch.bfloat16:
        want = want.to(torch.bfloat16).float()
    assert got.shape == want.shape and torch.equal(got, want)
