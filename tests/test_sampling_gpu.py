"""GPU: HIP sampling kernels (through the C ABI) vs the CPU oracle.  Tolerances: fp32 values 2e-5 abs on
O(1) outputs (fp32 reassociation only); bf16 values are compared against the oracle run on the SAME
bf16-rounded maps, so the same 2e-5 applies to the arithmetic."""
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seed", [0, 1])
def test_msda_matches_oracle(hip_lib, dtype, seed):
    from far3d_amd import ops
    from oracle import sampling
    c = cases.msda_case(bs=3, Q=53, H=8, Dh=32, hw=((10, 15), (5, 8), (3, 4), (2, 2)), P=13, seed=seed)
    v = c["value"].to(dtype)
    want = sampling.msda_grid_sample(v.float(), c["shapes"], c["lsi"], c["loc"], c["w"])
    got = ops.msda_forward(v.to(DEV), c["shapes"].to(DEV), c["lsi"].to(DEV), c["loc"].to(DEV), c["w"].to(DEV))
    assert got.shape == want.shape
    assert (got.cpu() - want).abs().max().item() < 2e-5


def test_msda_accepts_reference_4d_weights_and_checks_step(hip_lib):
    from far3d_amd import ops
    c = cases.msda_case(bs=2, Q=5, H=2, Dh=4, seed=5)
    args = [c[k].to(DEV) for k in ("value", "shapes", "lsi", "loc")]
    w4 = c["w"].flatten(-2).to(DEV)  # the reference passes (bs,Q,H,L*P), detr3d_transformer.py:541-542
    a = ops.msda_forward(*args, w4)
    b = ops.msda_forward(*args, c["w"].to(DEV))
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        ops.msda_forward(torch.cat([args[0]] * 3)[:3], args[1], args[2], torch.cat([args[3]] * 3)[:3],
                         torch.cat([w4] * 3)[:3], im2col_step=2)


def test_msda_empty_queries(hip_lib):
    from far3d_amd import ops
    c = cases.msda_case(bs=1, Q=1, H=2, Dh=4, seed=2)
    out = ops.msda_forward(c["value"].to(DEV), c["shapes"].to(DEV), c["lsi"].to(DEV), c["loc"][:, :0].contiguous().to(DEV),
                           c["w"][:, :0].contiguous().to(DEV))
    assert out.shape == (1, 0, 8)


VARIANTS = [0, 3, 7, 11]   # far3d_aggregate_forward kernel variants (include/far3d_hip.h): 0 = default (v8: factored softmax, decoupled waves), 3 = round-1 kernel, 7 = round 2/3 kernel, 11 = 7 + VALU reductions / packed FMAs


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aggregate_small(hip_lib, dtype, variant):
    for seed in range(3):
        assert cases.run_aggregate_case(cases.small_aggregate_case(seed), DEV, dtype, variant) < 2e-5


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aggregate_config2_full_size(hip_lib, dtype, variant):
    c = cases.config2_aggregate_case(seed=0)
    assert cases.run_aggregate_case(c, DEV, dtype, variant) < 5e-5


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aggregate_near_queries_wide_spread(hip_lib, dtype, variant):
    """Queries close to the cameras: the key points of one query cover many tokens (the per-corner path of the v4 kernel,
    patches of every aspect ratio, list flushes) and several cameras."""
    assert cases.run_aggregate_case(cases.near_aggregate_case(seed=1), DEV, dtype, variant) < 5e-5
    assert cases.run_aggregate_case(cases.aggregate_case(7, (640, 960), 64, seed=3, offset_std=12.0), DEV, dtype, variant) < 5e-5


def test_aggregate_strided_operands_and_perm(hip_lib):
    """U and the key-point offsets as column blocks of one row-strided buffer (the merged-GEMM output the engine passes),
    bf16 output, and a workgroup order: same rows as the dense call."""
    from far3d_amd import ops
    c = cases.config2_aggregate_case(seed=2)
    A, J = c["U"].shape
    buf = torch.zeros(A, 512)
    buf[:, :J] = c["U"]
    buf[:, J:J + 39] = c["offsets"].reshape(A, 39)
    buf = buf.to(DEV)
    d = lambda t: t.to(DEV).contiguous()
    feat, ref, l2i, Vc = d(c["feat"].to(torch.bfloat16)), d(c["ref"]), d(c["lidar2img"]), d(c["Vc"])
    dense = cases.hip_aggregate(c, DEV, torch.bfloat16)
    perm = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"])
    assert sorted(perm.cpu().tolist()) == list(range(A))
    for variant in VARIANTS:
        got = ops.aggregate_forward(feat, ref, buf[:, J:J + 39], l2i, buf[:, :J], Vc, c["level_hw"], c["level_start"], c["pc_range"],
                                    c["pad_hw"], perm=perm, variant=variant, out_dtype=torch.bfloat16)
        assert got.dtype == torch.bfloat16
        assert (got.float() - dense).abs().max().item() < 1e-2 * max(1.0, dense.abs().max().item())
        got32 = ops.aggregate_forward(feat, ref, buf[:, J:J + 39], l2i, buf[:, :J], Vc, c["level_hw"], c["level_start"], c["pc_range"],
                                      c["pad_hw"], perm=perm, variant=variant)
        assert (got32 - dense).abs().max().item() < 5e-5
        # reproducible bit for bit, launch after launch and whatever the workgroup order (no float atomics anywhere)
        for pm in (perm, None):
            again = ops.aggregate_forward(feat, ref, buf[:, J:J + 39], l2i, buf[:, :J], Vc, c["level_hw"], c["level_start"], c["pc_range"],
                                          c["pad_hw"], perm=pm, variant=variant)
            assert torch.equal(again, got32), "variant %d is not reproducible" % variant


def test_aggregate_equals_unfused_msda_sum(hip_lib):
    """Size-independent property at full size: fused kernel == sum over cameras of the drop-in MSDA op fed
    with the replicated locations / permuted softmax weights the reference materialises."""
    from far3d_amd import ops
    from oracle import sampling
    c = cases.config2_aggregate_case(seed=4)
    N, S, C = c["feat"].shape
    A, P, G, L = c["ref"].shape[0], 13, 8, 4
    pc = torch.tensor(c["pc_range"])
    kp = (c["ref"] * (pc[3:] - pc[:3]) + pc[:3])[None, :, None, :] + c["offsets"][None]
    p2d = sampling.project_points(kp, c["lidar2img"][None], c["pad_hw"]).flatten(end_dim=1)
    loc = p2d[:, :, None, None].repeat(1, 1, G, L, 1, 1).contiguous()
    logits = c["U"][:, None, :] + c["Vc"][None]
    w = logits.reshape(1, A, -1, G).softmax(-2).reshape(1, A, N, -1, G).permute(0, 2, 1, 4, 3).contiguous()[0]
    shapes = torch.tensor([list(x) for x in c["level_hw"]])
    per_cam = ops.msda_forward(c["feat"].view(N, S, G, C // G).to(DEV), shapes.to(DEV),
                               torch.tensor(c["level_start"]).to(DEV), loc.to(DEV), w.to(DEV))
    fused = cases.hip_aggregate(c, DEV)
    assert (per_cam.sum(0) - fused).abs().max().item() < 5e-5


def test_aggregate_invisible_queries_are_zero(hip_lib):
    c = cases.small_aggregate_case(seed=7)
    c["lidar2img"] = c["lidar2img"].clone()
    c["lidar2img"][:, 2, :] = 0.0
    c["lidar2img"][:, 2, 3] = -1.0  # every point "behind" every camera: z clamps to 1e-5, coords explode
    out = cases.hip_aggregate(c, DEV)
    assert out.abs().max().item() == 0.0
    assert cases.oracle_aggregate(c).abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aggregate_generic_point_and_level_counts(hip_lib, dtype):
    """P != 13 / L != 4 take the runtime-sized instantiation of the kernel (no quad-level pass)."""
    c = cases.aggregate_case(num_cams=5, pad_hw=(128, 192), A=45, P=6, seed=8, offset_std=3.0)
    c["level_hw"], c["level_start"] = c["level_hw"][:3], c["level_start"][:3]
    S3 = c["level_start"][2] + c["level_hw"][2][0] * c["level_hw"][2][1]
    g = torch.Generator().manual_seed(8)
    c["feat"] = c["feat"][:, :S3].contiguous()
    c["U"], c["Vc"] = torch.randn(45, 3 * 6 * 8, generator=g), torch.randn(5, 3 * 6 * 8, generator=g)
    assert cases.run_aggregate_case(c, DEV, dtype) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("extra", [384, 7, 1])
def test_aggregate_sibling_workgroups_for_heavy_queries(hip_lib, dtype, extra):
    """Variant 9 (round 5): queries two cameras see are handled by TWO workgroups that deal the items into four shares and merge their
    partial sums in part order.  Against the oracle at the benchmarked size; rows that were not split are BIT-identical to variant 8;
    the marked set is the first `extra` heavy queries in row order (few slots: most heavy queries stay unsplit); launch after launch
    and as a hipGraph the result is bit-identical (the tickets return to zero); sibling slots that are not used do nothing."""
    from far3d_amd import ops
    c = cases.config2_aggregate_case(seed=5)
    A = c["ref"].shape[0]
    d = lambda t: t.to(DEV).contiguous()
    feat = d(c["feat"].to(dtype))
    ref, offs, l2i, U, Vc = d(c["ref"]), d(c["offsets"]), d(c["lidar2img"]), d(c["U"]), d(c["Vc"])
    want = cases.oracle_aggregate(c, dtype)
    tab = ops.agg_tables(Vc)
    perm8 = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"])
    base = ops.aggregate_forward(feat, ref, offs, l2i, U, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm8, variant=8, tables=tab)
    sp = ops.AggSplit(A, extra, DEV)
    perm = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"], split=sp)
    pm = perm.cpu().numpy()
    main, sib = pm[:A], pm[A:]
    assert sorted((main & 0x1fffffff).tolist()) == list(range(A))
    marked = sorted((main[(main & (1 << 29)) != 0] & 0x1fffffff).tolist())
    used = sib[sib != 0x7fffffff]
    assert sorted((used & 0x1fffffff).tolist()) == marked and bool(((used >> 29) & 3 == 3).all()), "every marked query has exactly one sibling entry"
    assert 0 < len(marked) <= extra
    # the choice is by row order: the marked set is a prefix of the heavy rows
    sp_all = ops.AggSplit(A, 4096, DEV)
    pa = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"], split=sp_all).cpu().numpy()[:A]
    heavy = sorted((pa[(pa & (1 << 29)) != 0] & 0x1fffffff).tolist())
    assert marked == heavy[:len(marked)] and (len(marked) == extra or len(marked) == len(heavy))
    run = lambda: ops.aggregate_forward(feat, ref, offs, l2i, U, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab, split=sp)
    got = run()
    assert (got.float().cpu() - want).abs().max().item() < 5e-5
    unsplit = torch.ones(A, dtype=torch.bool)
    unsplit[marked] = False
    assert torch.equal(got[unsplit.to(DEV)], base[unsplit.to(DEV)]), "rows without a sibling take variant 8's path bit for bit"
    assert int(sp.tickets.abs().sum()) == 0
    for _ in range(3):
        assert torch.equal(run(), got) and int(sp.tickets.abs().sum()) == 0
    out = torch.empty_like(got)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.aggregate_forward(feat, ref, offs, l2i, U, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab, split=sp, out=out)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, got)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aggregate_sorted_mode_is_bit_identical(hip_lib, dtype):
    """Round 6: SORTED mode of the default kernel -- far3d_agg_order also emits every row's launch slot (inv) and the layer-independent
    point in metres + a two-camera hint (qbase); U / offsets are handed over in launch order.  Same rows bit for bit as the unsorted call,
    against the oracle, with a hole, launch after launch."""
    from far3d_amd import ops
    c = cases.config2_aggregate_case(seed=6)
    A = c["ref"].shape[0]
    d = lambda t: t.to(DEV).contiguous()
    feat = d(c["feat"].to(dtype))
    ref, offs, l2i, U, Vc = d(c["ref"]), d(c["offsets"]).reshape(A, -1), d(c["lidar2img"]), d(c["U"]), d(c["Vc"])
    tab = ops.agg_tables(Vc)
    want = cases.oracle_aggregate(c, dtype)
    for hole in (None, (torch.tensor([37], dtype=torch.int32, device=DEV), 100, 300)):
        perm, (inv, qbase) = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"], hole=hole, sorted_operands=True)
        pm, iv = perm.cpu().long(), inv.cpu().long()
        rows = torch.where(pm < 0, ~pm, pm)
        assert torch.equal(rows[iv], torch.arange(A)), "perm[inv[a]] names row a"
        # qbase = (the reference point in metres, the camera hint) per slot
        pc = torch.tensor(c["pc_range"], dtype=torch.float64)
        refm3 = c["ref"].double() * (pc[3:] - pc[:3]) + pc[:3]
        got_qb = qbase.cpu()[iv]
        assert (got_qb[:, :3].double() - refm3).abs().max().item() < 1e-4
        hint = qbase.cpu().view(torch.int32)[:, 3][iv]                    # cam0 | cam1 << 8, the two closest cameras
        cam0, cam1 = (hint & 0xff).long(), ((hint >> 8) & 0xff).long()
        assert int(cam0.max()) < 7 and int(cam1.max()) < 7 and bool((cam0 != cam1).all())
        refm = torch.cat([refm3, torch.ones(A, 1, dtype=torch.float64)], 1)
        qb = torch.einsum("nij,aj->ani", c["lidar2img"].double(), refm)[:, :, :3]
        uv = qb[:, :, :2] / qb[:, :, 2:3].clamp(min=1e-5) / torch.tensor([c["pad_hw"][1], c["pad_hw"][0]], dtype=torch.float64) - 0.5
        cost = torch.where(qb[:, :, 2] > 1e-5, (uv ** 2).sum(-1), torch.full_like(qb[:, :, 2], 1e9))
        srt = cost.sort(dim=1)
        clear = (srt.values[:, 1] - srt.values[:, 0]) > 1e-6 * (1 + srt.values[:, 0])           # no near tie for the first place
        assert torch.equal(cam0[clear], srt.indices[clear, 0])
        Us, Os = torch.full_like(U, float("nan")), torch.full_like(offs, float("nan"))
        Us[inv.long()], Os[inv.long()] = U, offs
        base = ops.aggregate_forward(feat, ref, offs, l2i, U, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, variant=8, tables=tab)
        run = lambda: ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], perm=perm, tables=tab,
                                            qbase=qbase)
        got = run()
        assert torch.equal(got, base), "sorted mode differs from the unsorted call"
        assert torch.equal(run(), got)
        if hole is None:
            assert (got.float().cpu() - want).abs().max().item() < 5e-5
        else:
            assert got[137:300].abs().max().item() == 0.0 and (got[:137].float().cpu() - want[:137]).abs().max().item() < 5e-5
    with pytest.raises(ValueError):
        ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], tables=tab, qbase=qbase)


def test_producers_store_in_launch_order(hip_lib):
    """far3d_layernorm_rows and far3d_rowchain_attn_out(ul_rows): the row map moves the GEMM-operand outputs only."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, C = 203, 256
    x, add = torch.randn(rows, C, generator=g).to(DEV), torch.randn(rows, C, generator=g).to(DEV)
    gam, bet = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    inv = torch.randperm(rows, generator=g).to(torch.int32).to(DEV)
    for dt in (torch.float32, torch.bfloat16):
        XW0 = torch.zeros(rows, 2 * C, dtype=dt, device=DEV)
        XW1 = torch.zeros(rows, 2 * C, dtype=dt, device=DEV)
        y0 = ops.layernorm(x, gam, bet, add=add, y2=XW0[:, :C], yb=XW0[:, C:])[0]
        y1 = ops.layernorm(x, gam, bet, add=add, y2=XW1[:, :C], yb=XW1[:, C:], out_rows=inv)[0]
        assert torch.equal(y0, y1)
        assert torch.equal(XW1[inv.long()], XW0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aggregate_two_kernel_split_is_bit_identical(hip_lib, dtype):
    """Variant 13 (round 6, A/B only): the sorted-mode kernel as a list-build launch + a pure gather launch through global lists --
    the same entries in the same order: the rows of the fused kernel (bit for bit on bf16 rows), with a hole, launch after launch."""
    from far3d_amd import ops
    c = cases.config2_aggregate_case(seed=9)
    A = c["ref"].shape[0]
    d = lambda t: t.to(DEV).contiguous()
    feat = d(c["feat"].to(dtype))
    ref, offs, l2i, U, Vc = d(c["ref"]), d(c["offsets"]).reshape(A, -1), d(c["lidar2img"]), d(c["U"]), d(c["Vc"])
    tab = ops.agg_tables(Vc)
    lists = ops.AggLists(A, DEV)
    for hole in (None, (torch.tensor([11], dtype=torch.int32, device=DEV), 50, 90)):
        perm, (inv, qbase) = ops.aggregation_order(ref, l2i, c["pc_range"], c["pad_hw"], hole=hole, sorted_operands=True)
        Us, Os = torch.empty_like(U), torch.empty_like(offs)
        Us[inv.long()], Os[inv.long()] = U, offs
        kw = dict(perm=perm, tables=tab, qbase=qbase)
        fused = ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], **kw)
        for _ in range(2):
            split = ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], variant=13, lists=lists, **kw)
            if dtype == torch.bfloat16:
                assert torch.equal(split, fused)
            else:       # fp32 rows: the stand-alone gather kernel's accumulate is contracted differently by the compiler -- 1 ulp
                assert (split - fused).abs().max().item() <= 2e-7 * fused.abs().max().item()
        assert 0 < int(lists.counts.max()) <= 1024
    with pytest.raises(ValueError):
        ops.aggregate_forward(feat, ref, Os, l2i, Us, Vc, c["level_hw"], c["level_start"], c["pc_range"], c["pad_hw"], variant=13, **kw)
