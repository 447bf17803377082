"""CPU: small pieces of host logic the GPU path relies on (BN folding, weight packing layout, camera-embedding packing)."""
import torch
import torch.nn.functional as F

from far3d_amd import weights


def test_fold_bn_equals_conv_then_eval_batchnorm():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 9, 11, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g)
    bw, bb = torch.rand(7, generator=g) + 0.5, torch.randn(7, generator=g)
    mean, var = torch.randn(7, generator=g), torch.rand(7, generator=g) + 0.1
    for eps in (1e-5, 1e-3):       # VoVNet / YOLOX towers
        want = F.batch_norm(F.conv2d(x, w, padding=1), mean, var, bw, bb, training=False, eps=eps)
        fw, fb = weights.fold_bn(w, bw, bb, mean, var, eps)
        got = F.conv2d(x, fw, fb, padding=1)
        assert (got - want).abs().max().item() < 1e-4


def test_packed_conv_layout_tap_major_zero_padded():
    """PackedConv: rows = channels (zero rows up to the over-read margin), K = (tap, cin_pad) with cin padded to 32."""
    from far3d_amd import ops
    g = torch.Generator().manual_seed(1)
    w = torch.randn(10, 40, 3, 3, generator=g)
    b = torch.randn(10, generator=g)
    pc = ops.PackedConv(w, b, stride=1, pad=1, dtype=torch.float32, device="cpu")
    cin_pad = 64
    assert pc.w.shape[1] == 9 * cin_pad and pc.w.shape[0] >= 10 + 256 and pc.bias.shape[0] == pc.w.shape[0]
    wk = pc.w.view(pc.w.shape[0], 9, cin_pad)
    for tap in range(9):
        assert torch.equal(wk[:10, tap, :40], w[:, :, tap // 3, tap % 3])
    assert wk[:10, :, 40:].abs().max().item() == 0.0 and wk[10:].abs().max().item() == 0.0
    assert torch.equal(pc.bias[:10], b) and pc.bias[10:].abs().max().item() == 0.0
    assert pc.out_hw(17, 23) == (17, 23)
    assert ops.PackedConv(w, None, stride=2, pad=1, dtype=torch.float32, device="cpu").out_hw(17, 23) == (9, 12)


def test_cam_embed_chain_packing_is_transposed_and_stacked():
    from far3d_amd import ops
    g = torch.Generator().manual_seed(2)
    layers = [(torch.randn(128, 12, generator=g), torch.randn(128, generator=g), torch.randn(256, 128, generator=g),
               torch.randn(256, generator=g), torch.randn(256, generator=g), torch.randn(256, generator=g),
               torch.randn(416, 256, generator=g), torch.randn(416, generator=g)) for _ in range(3)]
    p = ops.pack_cam_embed_chain(layers, "cpu")
    assert p["w0t"].shape == (3, 12, 128) and p["w2t"].shape == (3, 128, 256) and p["w3t"].shape == (3, 256, 416)
    assert torch.equal(p["w2t"][1], layers[1][2].t()) and torch.equal(p["b3"][2], layers[2][7])
    assert all(v.is_contiguous() and v.dtype == torch.float32 for v in p.values())


def test_load_checkpoint_unwraps_runner_layout_ddp_prefix_and_shared_branches(tmp_path):
    """§8(f2): a synthetic checkpoint in the layout mmcv's runner writes for the reference (meta + state_dict, DDP `module.`
    prefixes, six copies of the shared cls/reg branches, BN bookkeeping) loads into the schema unchanged."""
    import pytest
    import torch
    from far3d_amd import weights
    spec = weights.detector_spec("V-tiny-eSE", num_query=8, num_propagated=4)
    sd = weights.init_state_dict(spec, seed=3)
    ck = {}
    for k, v in sd.items():
        ck["module." + k] = v.clone()
        for i in range(1, 6):
            for br in ("cls_branches", "reg_branches"):
                if k.startswith("pts_bbox_head.%s.0." % br):
                    ck["module." + k.replace("%s.0." % br, "%s.%d." % (br, i))] = v.clone()
    ck["module.img_backbone.stem.stem_1/norm.num_batches_tracked"] = torch.tensor(5)
    path = str(tmp_path / "iter_1.pth")
    torch.save(dict(meta=dict(iter=1), state_dict=ck, optimizer={}), path)
    got = weights.load_checkpoint(path, strict_schema=spec)
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert "img_backbone.stem.stem_1/conv.weight" in got
    # bare state dict, no prefixes
    torch.save(sd, path)
    assert list(weights.load_checkpoint(path, strict_schema=spec)) == list(sd)
    # a diverging shared-branch copy and a missing key are loud
    bad = dict(ck)
    bad["module.pts_bbox_head.cls_branches.3.6.bias"] = bad["module.pts_bbox_head.cls_branches.3.6.bias"] + 1
    with pytest.raises(ValueError):
        weights.normalize_state_dict(dict(state_dict=bad))
    # ... in whatever order the keys arrive (an alias that preceded its owner used to be overwritten silently)
    with pytest.raises(ValueError):
        weights.normalize_state_dict(dict(state_dict=dict(reversed(list(bad.items())))))
    rev = weights.normalize_state_dict(dict(state_dict=dict(reversed(list(ck.items())))), strict_schema=spec)
    assert all(torch.equal(rev[k], sd[k]) for k in sd)
    del ck["module.pts_bbox_head.reference_points.weight"]
    with pytest.raises(KeyError):
        weights.normalize_state_dict(dict(state_dict=ck), strict_schema=spec)


def test_channel_sum_fusion_is_offered_exactly_where_the_gemm_epilogue_can_produce_it():
    """Host rule of the eSE pooling fusion (ops.conv_can_fuse_sums): the OSA concat layers of the benchmarked frame -- 1x1, bf16 or
    pair-stored, a measured pipelined-GEMM tile, maps of at least one pixel tile -- qualify in every stage and at the per-rank camera
    counts of a sharded run; small maps, fp32 maps, 3x3 layers and untuned shapes do not (the eSE op then pools the map itself)."""
    from far3d_amd import ops
    spec = weights.VOV_SPECS["V-99-eSE"]
    hw = [(160, 240), (80, 120), (40, 60), (20, 30)]
    in_ch = spec["stem"][2]
    for si in range(4):
        sc, oc = spec["stage_conv_ch"][si], spec["stage_out_ch"][si]
        for first in (True, False):
            cin = (in_ch if first else oc) + 5 * sc
            pc = ops.PackedConv(torch.zeros(oc, cin, 1, 1), torch.zeros(oc), dtype=torch.bfloat16, device="cpu")
            for ncam in (7, 4, 2, 1):
                x = torch.empty(ncam, hw[si][0], hw[si][1], cin, dtype=torch.bfloat16)
                assert ops.conv_can_fuse_sums(x, pc), (si, first, ncam)
            assert not ops.conv_can_fuse_sums(torch.empty(7, 8, 8, cin, dtype=torch.bfloat16), pc)            # 64-pixel maps (configs[0])
            assert not ops.conv_can_fuse_sums(torch.empty(7, hw[si][0], hw[si][1], cin, dtype=torch.float32), pc)
            if spec["block_per_stage"][si] == 1:
                break
        in_ch = oc
    pc3 = ops.PackedConv(torch.zeros(128, 128, 3, 3), torch.zeros(128), pad=1, dtype=torch.bfloat16, device="cpu")
    assert not ops.conv_can_fuse_sums(torch.empty(7, 160, 240, 128, dtype=torch.bfloat16), pc3)
    odd = ops.PackedConv(torch.zeros(96, 352, 1, 1), torch.zeros(96), dtype=torch.bfloat16, device="cpu")       # no entry in the table
    assert not ops.conv_can_fuse_sums(torch.empty(7, 40, 60, 352, dtype=torch.bfloat16), odd)


def test_bench_launch_plan():
    """`python bench.py --gpus N` starts its own N ranks (the reference wraps its test script in a launcher the
    same way, tools/dist_test.sh:11-23): one torch.distributed.run process group on 127.0.0.1, the flags passed through; under a
    launcher (WORLD_SIZE set) the process is a rank and must agree with --gpus."""
    import importlib.util
    import os
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("far3d_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launch_plan(1, {}, []) is None
    assert bench.launch_plan(4, {"WORLD_SIZE": "4"}, ["--gpus", "4"]) is None
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {"WORLD_SIZE": "1"}, ["--gpus", "8"])
    cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")


def test_ida_matrix_against_a_stepwise_float32_chain():
    """ida_matrix is a float64 closed form rounded once; the reference (datasets/pipelines/custom_pipeline.py:294-311)
    composes scale, crop, flip and rotation step by step in float32.  Equal when rotate == 0 (the reference's only supported
    value), float32-rounding close otherwise."""
    import numpy as np
    import torch
    from far3d_amd.data_pipeline.preprocess import ida_matrix

    def stepwise(resize, crop, flip, rotate):
        rot = torch.eye(2) * resize
        tran = -torch.tensor(crop[:2], dtype=torch.float32)
        if flip:
            A = torch.tensor([[-1.0, 0.0], [0.0, 1.0]])
            b = torch.tensor([crop[2] - crop[0], 0.0])
            rot, tran = A.matmul(rot), A.matmul(tran) + b
        th = rotate / 180 * np.pi
        A = torch.tensor([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]], dtype=torch.float32)
        b = torch.tensor([crop[2] - crop[0], crop[3] - crop[1]], dtype=torch.float32) / 2
        b = A.matmul(-b) + b
        rot, tran = A.matmul(rot), A.matmul(tran) + b
        M = torch.eye(3)
        M[:2, :2], M[:2, 2] = rot, tran
        return M

    rng = np.random.RandomState(3)
    for _ in range(50):
        resize = float(np.float32(rng.uniform(0.3, 1.2)))
        left, top = int(rng.randint(0, 400)), int(rng.randint(0, 300))
        crop = (left, top, left + 960, top + 640)
        for flip in (False, True):
            assert torch.equal(ida_matrix(resize, crop, flip, 0.0), stepwise(resize, crop, flip, 0.0))
            ang = float(rng.uniform(-25, 25))
            a, b = ida_matrix(resize, crop, flip, ang), stepwise(resize, crop, flip, ang)
            assert (a - b).abs().max().item() <= 4e-6 * b.abs().max().item()


def test_rowchain_weight_packing_is_the_documented_fragment_order():
    """ops.pack_rowchain (include/far3d_hip.h "Packed weights"): element j of lane l of (column tile t, k step s) is
    W[16 t + (l & 15)][32 s + 8 (l >> 4) + j]; the bias covers the padded columns; the geometry gate of the chains."""
    import torch
    from far3d_amd import ops
    g = torch.Generator().manual_seed(0)
    pc = ops.PackedConv(torch.randn(455, 512, generator=g), torch.randn(455, generator=g), dtype=torch.bfloat16, device="cpu")
    frag, bias = ops.pack_rowchain(pc)
    assert tuple(frag.shape) == (29, 16, 64, 8) and tuple(bias.shape) == (464,) and frag.is_contiguous()
    t, s_, l, j = torch.meshgrid(torch.arange(29), torch.arange(16), torch.arange(64), torch.arange(8), indexing="ij")
    assert torch.equal(frag, pc.w[16 * t + (l & 15), 32 * s_ + 8 * (l >> 4) + j])
    assert torch.equal(bias[:455], pc.bias[:455]) and not bias[455:].any() and not frag[28, :, 7:16].any()       # rows past Cout are zero
    two, b2 = ops.pack_rowchain(ops.PackedConv(torch.randn(10, 256, generator=g), None, dtype=torch.bfloat16, device="cpu"), cols=32)
    assert tuple(two.shape) == (2, 8, 64, 8) and tuple(b2.shape) == (32,) and not b2.any() and not two[1].any()
    import pytest
    with pytest.raises(ValueError):
        ops.pack_rowchain(ops.PackedConv(torch.randn(16, 384, generator=g), None, dtype=torch.bfloat16, device="cpu"))     # K % 256
    with pytest.raises(ValueError):
        ops.pack_rowchain(ops.PackedConv(torch.randn(16, 256, generator=g), None, dtype=torch.float32, device="cpu"))      # bf16 only
    mk = lambda co, ci: ops.PackedConv(torch.randn(co, ci, generator=g), torch.zeros(co), dtype=torch.bfloat16, device="cpu")
    ly = dict(out=mk(256, 256), wl=mk(455, 512), oproj=mk(256, 256), ffn1=mk(1024, 256), ffn2=mk(256, 1024), qkv=mk(768, 512))
    assert ops.RowChainLayer.supported(ly, 256, torch.bfloat16)
    assert not ops.RowChainLayer.supported(ly, 256, torch.float32)
    assert not ops.RowChainLayer.supported(dict(ly, wl=mk(300, 512)), 256, torch.bfloat16)          # 19 column tiles: not the 29 the kernel is built for
    assert not ops.RowChainLayer.supported(dict(ly, ffn1=mk(512, 256), ffn2=mk(256, 512)), 256, torch.bfloat16)
    cls, reg = [mk(256, 256), mk(256, 256), mk(26, 256)], [mk(256, 256), mk(256, 256), mk(8, 256)]
    assert ops.RowChainBranches.supported(cls, reg, 256, torch.bfloat16)
    assert not ops.RowChainBranches.supported(cls[:2] + [mk(40, 256)], reg, 256, torch.bfloat16)    # > 32 classes: two tiles do not cover it
