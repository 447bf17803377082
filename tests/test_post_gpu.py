"""GPU: the post-processing / calibration kernels of csrc/post.hip (C ABI far3d_topk, far3d_decode_topk, far3d_camera_prep,
far3d_agg_order, far3d_nan_to_num) and the row-strided variants of far3d_add_cast / far3d_layernorm, against torch on the CPU
and the oracle's decode."""
import numpy as np
import pytest
import torch

from far3d_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,K", [(1544, 256), (37, 5), (40144, 300), (1024, 1024), (3000, 1)])
def test_topk_matches_torch(hip_lib, n, K):
    from far3d_amd import ops
    g = torch.Generator().manual_seed(n + K)
    v = torch.randn(n, generator=g)
    v[::7] = torch.round(v[::7] * 4) / 4          # ties
    v[3] = float("inf")
    v[5] = -float("inf")
    idx, val = ops.topk(v.to(DEV), K, with_values=True)
    idx, val = idx.cpu(), val.cpu()
    want_v, _ = torch.topk(v, K)
    assert torch.equal(val, want_v)
    assert torch.equal(v[idx], val)
    # ties resolved towards the lower index, positions unique
    order = np.lexsort((np.arange(n), -v.numpy()))[:K]
    assert np.array_equal(idx.numpy(), order)


@pytest.mark.parametrize("n,K,levels", [(5000, 100, 3), (40000, 300, 2), (1544, 256, 1), (2000, 700, 5)])
def test_topk_with_huge_tie_groups_takes_the_exact_tie_rule(hip_lib, n, K, levels):
    """Values drawn from a handful of levels: the K-th place sits inside a tie group far larger than the window of the fast path (no cut
    keeps between K and K + max(K/8, 32) entries), so the search runs to its end and the index bisection decides -- lowest indices first."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(n * 3 + K)
    v = torch.randint(0, levels, (n,), generator=g).float() * 0.5 - 1.0
    idx, val = ops.topk(v.to(DEV), K, with_values=True)
    order = np.lexsort((np.arange(n), -v.numpy()))[:K]
    assert np.array_equal(idx.cpu().numpy(), order) and torch.equal(val.cpu(), v[torch.as_tensor(order)])


@pytest.mark.parametrize("layout", ["top_values_only_at_sampled_positions", "top_values_never_sampled", "plain"])
@pytest.mark.parametrize("n,K", [(40960, 300), (40144, 300), (9000, 1000), (4097, 64)])
def test_topk_when_the_strided_sample_misleads(hip_lib, layout, n, K):
    """Large inputs place the candidate threshold from a strided 2048-entry sample (every n/2048-th input).  Layouts whose sample says
    nothing true about the input -- the largest values ONLY at the sampled positions (far fewer than K candidates come back) or NEVER
    there (more candidates than the LDS holds) -- must fall back to the 16-wave bisection over the registers and give the same answer."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(n + K + len(layout))
    v = torch.randn(n, generator=g) * 0.1 - 4.0
    sampled = torch.unique((torch.arange(2048, dtype=torch.int64) * n) // 2048)
    if layout == "top_values_only_at_sampled_positions":
        v[sampled] += 10.0
    elif layout == "top_values_never_sampled":
        mask = torch.ones(n, dtype=torch.bool)
        mask[sampled] = False
        pick = torch.nonzero(mask)[:, 0][::3][:6000]
        v[pick] += 10.0
    idx, val = ops.topk(v.to(DEV), K, with_values=True)
    order = np.lexsort((np.arange(n), -v.numpy()))[:K]
    assert np.array_equal(idx.cpu().numpy(), order) and torch.equal(val.cpu(), v[torch.as_tensor(order)])


def test_decode_topk_matches_oracle_decode(hip_lib):
    from far3d_amd import ops
    from oracle import far3d_oracle
    g = torch.Generator().manual_seed(11)
    A, ncls = 1544, 26
    cls = torch.randn(1, 1, A, ncls, generator=g) * 2
    box = torch.randn(1, 1, A, 8, generator=g)
    box[..., :3] = (torch.rand(1, 1, A, 3, generator=g) - 0.5) * torch.tensor([330.0, 330.0, 12.0])   # some centres out of range
    orc = far3d_oracle.Far3DOracle({}, far3d_oracle.default_cfg())
    want = orc.decode(dict(all_cls_scores=cls, all_bbox_preds=box))
    got = ops.decode_topk(cls[0, 0].to(DEV), box[0, 0].to(DEV), 300, synth.PC_RANGE)
    keep = got["keep"].cpu()
    assert got["labels_3d"].dtype == torch.int64 and int(keep.sum()) == want["scores_3d"].numel() < 300
    assert torch.equal(got["labels_3d"].cpu()[keep], want["labels_3d"])
    assert (got["scores_3d"].cpu()[keep] - want["scores_3d"]).abs().max().item() < 1e-6
    assert (got["boxes_3d"].cpu()[keep] - want["boxes_3d"]).abs().max().item() < 1e-4
    s = got["scores_3d"].cpu()
    assert (s[:-1] >= s[1:]).all()


@pytest.mark.parametrize("A", [1576, 4484, 9000])
def test_decode_topk_large_query_counts_take_the_chunked_path(hip_lib, A):
    """the reference's threshold proposal mode can exceed A * 26 = 40960 logits (M > 675 adaptive queries); the decode
    then ranks 40960-logit chunks in parallel and the chunks' winners in a second launch -- same result as the single launch,
    ties (planted across chunk borders) to the lower flat index, NaN logits first like torch.topk."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(A)
    ncls, K = 26, 300
    cls = torch.randn(A, ncls, generator=g) * 2
    flat = cls.view(-1)
    flat[::4099] = 3.25                                # ties, some in every chunk
    flat[-1] = 3.25
    flat[40961 % flat.numel()] = float("nan")
    box = torch.randn(A, 8, generator=g)
    got = ops.decode_topk(cls.to(DEV), box.to(DEV), K, [-1e3] * 3 + [1e3] * 3)
    key = torch.where(torch.isnan(flat), torch.full_like(flat, float("inf")), flat)
    order = np.lexsort((np.arange(flat.numel()), -key.numpy()))[:K]
    assert np.array_equal(got["labels_3d"].cpu().numpy(), order % ncls)
    want_s = torch.sigmoid(flat[torch.as_tensor(order)])
    gs = got["scores_3d"].cpu()
    assert torch.isnan(gs[0]) and torch.isnan(want_s[0])
    assert (gs[1:] - want_s[1:]).abs().max().item() < 1e-6
    q = torch.as_tensor(order // ncls)
    assert torch.allclose(got["boxes_3d"].cpu()[:, :2], box[q][:, :2])
    assert ops.decode_ws_bytes(A * ncls, K) == (0 if A * ncls <= 40960 else -(-A * ncls // 40960) * K * 8)


@pytest.mark.parametrize("A,ncls,K,mK", [(1544, 26, 300, 256), (1544, 10, 300, 256), (120, 10, 100, 64), (1576, 26, 300, 256), (5000, 26, 300, 256)])
def test_decode_with_the_memory_topk_in_one_launch(hip_lib, A, ncls, K, mK):
    """far3d_decode_topk_mem (round 6): workgroup 0 decodes, workgroup 1 ranks the memory update's scores -- bit for bit the two
    stand-alone calls (ties planted in both inputs; the last two shapes lie outside the fused launch and run as the two calls)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(A + ncls)
    cls = torch.randn(A, ncls, generator=g) * 2
    cls.view(-1)[::977] = 2.5
    box = torch.randn(A, 8, generator=g)
    sc = torch.sigmoid(cls.max(-1).values)
    sc[::7] = sc[3]
    if A == 5000:
        sc = torch.cat([sc, sc])                       # more than 4096 scores
    rng = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    cd, bd, sd = cls.to(DEV), box.to(DEV), sc.to(DEV)
    want, widx = ops.decode_topk(cd, bd, K, rng), ops.topk(sd, mK)
    got, gidx = ops.decode_topk(cd, bd, K, rng, mem_scores=sd, mem_K=mK)
    assert torch.equal(gidx, widx)
    for k in want:
        a, b = got[k], want[k]
        assert torch.equal(a, b) or (k == "scores_3d" and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))), k
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        r2, i2 = ops.decode_topk(cd, bd, K, rng, mem_scores=sd, mem_K=mK)
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(i2, widx) and torch.equal(r2["labels_3d"], want["labels_3d"]) and torch.equal(r2["boxes_3d"], want["boxes_3d"])


def test_decode_topk_keeps_velocity_channels(hip_lib):
    """code_size 10 (the reference nuScenes layout): vx, vy are appended like denormalize_bbox does (core/bbox/util.py:45-50)."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(12)
    cls, box = torch.randn(50, 10, generator=g), torch.randn(50, 10, generator=g)
    got = ops.decode_topk(cls.to(DEV), box.to(DEV), 20, [-1e3] * 3 + [1e3] * 3)
    assert got["boxes_3d"].shape == (20, 9)
    q = torch.div(torch.topk(cls.flatten(), 20).indices, 10, rounding_mode="floor")
    assert torch.allclose(got["boxes_3d"].cpu()[:, 7:], box[q][:, 8:])
    assert torch.allclose(got["boxes_3d"].cpu()[:, 6], torch.atan2(box[q][:, 6], box[q][:, 7]), atol=1e-6)


def test_camera_prep_inverse_and_mln_code(hip_lib):
    from far3d_amd import ops
    intr, extr, l2i = synth.ring_cameras(7, (640, 960))
    i2l, c14 = ops.camera_prep(l2i.to(DEV), intr.to(DEV), extr.to(DEV))
    want = torch.linalg.inv(l2i.double())
    rel = ((i2l.cpu().double() - want).abs() / (want.abs() + 1e-3)).max().item()
    assert rel < 1e-5, rel
    eye = i2l.cpu().double() @ l2i.double()
    assert (eye - torch.eye(4, dtype=torch.float64)).abs().max().item() < 1e-3     # f32 storage of entries up to ~1e3
    want14 = torch.cat([intr[:, 0, 0:1] / 1e3, intr[:, 1, 1:2] / 1e3, extr[:, :3, :].flatten(1)], dim=-1)
    assert torch.allclose(c14.cpu(), want14, atol=1e-6)


def test_agg_order_groups_like_camera_sort(hip_lib):
    from far3d_amd import ops
    from tests import cases
    c = cases.config2_aggregate_case(seed=5)
    ref, l2i = c["ref"].to(DEV), c["lidar2img"].to(DEV)
    perm = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"]).cpu()
    assert sorted(perm.tolist()) == list(range(ref.shape[0]))
    want = ops.camera_sorted_order(ref, l2i, c["pc_range"], c["pad_hw"], spatial=True).cpu()
    # same (camera, cell) grouping in the same group order; the order inside a group is free
    pc = torch.tensor(c["pc_range"])
    pts = c["ref"] * (pc[3:] - pc[:3]) + pc[:3]
    p = torch.einsum("nij,aj->nai", c["lidar2img"][:, :3, :3], pts) + c["lidar2img"][:, :3, 3][:, None, :]
    z = p[..., 2]
    u = p[..., 0] / z.clamp(min=1e-5) / c["pad_hw"][1] - 0.5
    v = p[..., 1] / z.clamp(min=1e-5) / c["pad_hw"][0] - 0.5
    cost = torch.where(z > 1e-5, u * u + v * v, torch.full_like(z, 1e9))
    cam = cost.argmin(0)
    uu, vv = torch.gather(u, 0, cam[None])[0], torch.gather(v, 0, cam[None])[0]
    key = (cam * 8 + ((vv + 0.5).clamp(0, 0.999) * 8).long()) * 8 + ((uu + 0.5).clamp(0, 0.999) * 8).long()
    kp = key[perm.long()]
    mism = int((kp[1:] < kp[:-1]).sum())
    assert mism <= 2, mism        # a point on a cell border may round to the neighbouring cell on the device
    assert torch.equal(torch.sort(key[want.long()]).values, torch.sort(kp).values)


def test_nan_to_num_inplace_and_bf16_copy(hip_lib):
    from far3d_amd import ops
    x = torch.randn(6, 100, 256)
    x[0, 0, 0], x[1, 2, 3], x[2, 5, 7] = float("nan"), float("inf"), -float("inf")
    want = torch.nan_to_num(x)
    d = x.to(DEV)
    xb = ops.nan_to_num_(d, bf16_copy=True)
    assert torch.equal(d.cpu(), want)
    assert torch.equal(xb.float().cpu(), want.to(torch.bfloat16).float())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_add_cast_and_layernorm_write_strided_operand_halves(hip_lib, dt):
    """[x+pos | x] merged-GEMM operand: both producers write the two halves of one (rows, 2C) buffer."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(77, 256, generator=g), torch.randn(77, 256, generator=g)
    buf = torch.zeros(77, 512, dtype=dt, device=DEV)
    ops.add_cast(a.to(DEV), b.to(DEV), dt, out_sum=buf[:, :256], out_a=buf[:, 256:])
    assert torch.equal(buf[:, :256].float().cpu(), (a + b).to(dt).float())
    assert torch.equal(buf[:, 256:].float().cpu(), a.to(dt).float())
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    buf.zero_()
    y = torch.empty(77, 256, device=DEV)
    ops.layernorm(a.to(DEV), gam.to(DEV), bet.to(DEV), out=y, add=b.to(DEV), y2=buf[:, :256], yb=buf[:, 256:])
    want = torch.nn.functional.layer_norm(a, (256,), gam, bet)
    assert (y.cpu() - want).abs().max().item() < 1e-5
    tol = 1e-5 if dt == torch.float32 else 2e-2
    assert (buf[:, :256].float().cpu() - (want + b)).abs().max().item() < tol
    assert (buf[:, 256:].float().cpu() - want).abs().max().item() < tol
