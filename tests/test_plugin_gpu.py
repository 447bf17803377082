"""GPU: the registry-level drop-in surface (Far3D detector, DeformableFeatureAggregationCuda module, VoVNet module,
MultiScaleDeformableAttnFunction) against the reference-generated golden vectors / the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from far3d_amd import config, plugin, synth, weights
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")


def _golden():
    z = np.load(os.path.join(GOLD, "far3d_small_seq.npz"))
    return z, json.loads(bytes(z["recipe"]).decode())


def test_far3d_detector_from_config_reproduces_reference_boxes(hip_lib):
    z, rc = _golden()
    det = plugin.build_detector(config.default_model_cfg(num_cams=rc["num_cams"], num_query=rc["num_query"],
                                                         num_propagated=rc["num_propagated"], memory_len=rc["memory_len"],
                                                         topk_proposals=rc["topk_proposals"]))
    spec = weights.detector_spec(rc["backbone"], num_query=rc["num_query"], num_propagated=rc["num_propagated"])
    det.load_state_dict(weights.init_state_dict(spec, seed=rc["weight_seed"]))
    det.prepare(DEV, precision="fp32")
    for fi in range(rc["frames"]):
        data, metas = synth.recipe_frame(rc, fi)
        res = det(return_loss=False, rescale=True, img_metas=metas, **data)[0]["pts_bbox"]
        from tests.conftest import assert_detections_match
        assert_detections_match(tuple(res[k].cpu().numpy() for k in ("labels_3d", "boxes_3d", "scores_3d")),
                                tuple(z["f%d_%s" % (fi, k)] for k in ("labels_3d", "boxes_3d", "scores_3d")), "frame %d" % fi)
        assert np.abs(det.last_outs["all_cls_scores"].cpu().numpy() - z["f%d_all_cls_scores" % fi]).max() < 1e-3


def test_aggregation_module_reference_signature(hip_lib):
    """DeformableFeatureAggregationCuda.forward(instance_feature, query_pos, feat_flatten, reference_points, spatial_flatten,
    level_start_index, pc_range, lidar2img, img_metas) vs the oracle's restatement of the reference module."""
    from oracle import far3d_oracle
    from tests import cases
    c = cases.aggregate_case(num_cams=3, pad_hw=(64, 96), A=50, seed=11)
    m = plugin.ATTENTION.build(dict(type="DeformableFeatureAggregationCuda", embed_dims=256, num_groups=8, num_levels=4, num_cams=3,
                                    dropout=0.1, num_pts=13, bias=2.0))
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.1))
        m.learnable_fc.weight.mul_(10.0)
    m.precision = "fp32"
    x, qpos = torch.randn(1, 50, 256, generator=g), torch.randn(1, 50, 256, generator=g)
    sd = {"L." + k: v for k, v in m.state_dict().items()}
    orc = far3d_oracle.Far3DOracle({k.replace("L.", "L.attentions.1."): v for k, v in sd.items()},
                                   far3d_oracle.default_cfg(num_cams=3))
    want = orc.cross_attn(x, qpos, c["feat"], c["ref"][None], c["level_hw"], c["level_start"], c["lidar2img"][None], c["pad_hw"], "L.")
    m = m.to(DEV)
    metas = [dict(pad_shape=[(64, 96, 3)] * 3)]
    got = m(x.to(DEV), qpos.to(DEV), c["feat"].to(DEV), c["ref"][None].to(DEV),
            torch.tensor([list(h) for h in c["level_hw"]], device=DEV), torch.tensor(c["level_start"], device=DEV),
            torch.tensor(c["pc_range"], device=DEV), c["lidar2img"][None].to(DEV), metas)
    assert got.shape == (1, 50, 256)
    assert (got.cpu() - want).abs().max().item() < 2e-4


def test_vovnet_module_forward_nchw(hip_lib):
    from oracle import far3d_oracle
    bb = plugin.BACKBONES.build(dict(type="VoVNet", spec_name="V-99-eSE", norm_eval=True, frozen_stages=-1, input_ch=3,
                                     out_features=("stage2", "stage3", "stage4", "stage5")))
    spec = weights.backbone_spec("V-99-eSE")
    sd = weights.init_state_dict(spec, seed=4)
    bb.load_state_dict({k[len("img_backbone."):]: v for k, v in sd.items()})
    bb.precision = "fp32"
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = far3d_oracle.Far3DOracle(sd, far3d_oracle.default_cfg()).backbone(x)
    got = bb(x.to(DEV))
    assert [tuple(g.shape) for g in got] == [tuple(w.shape) for w in want]
    for g, w in zip(got, want):
        assert (g.cpu() - w).abs().max().item() < 1e-3 * max(1.0, w.abs().max().item())


def test_msda_function_apply_is_mmcv_shaped(hip_lib):
    from oracle import sampling
    from tests import cases
    c = cases.msda_case(bs=2, Q=31, H=8, Dh=32, hw=((8, 12), (4, 6)), P=4, seed=9)
    out = plugin.MultiScaleDeformableAttnFunction.apply(c["value"].to(DEV), c["shapes"].to(DEV), c["lsi"].to(DEV), c["loc"].to(DEV),
                                                        c["w"].flatten(-2).to(DEV), 64)
    want = sampling.msda_grid_sample(c["value"], c["shapes"], c["lsi"], c["loc"], c["w"])
    assert (out.cpu() - want).abs().max().item() < 2e-5
    with pytest.raises(RuntimeError):
        plugin.MultiScaleDeformableAttnFunction.apply(c["value"].to(DEV).requires_grad_(), c["shapes"].to(DEV), c["lsi"].to(DEV),
                                                      c["loc"].to(DEV), c["w"].to(DEV), 64)
