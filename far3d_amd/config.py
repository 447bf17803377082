"""Config handling without mmcv: load an OpenMMLab-style python config (the reference's projects/configs/far3d.py loads
unchanged, its un-vendored `_base_` runtime file is simply not needed for inference) and a programmatic default."""
import os

POINT_CLOUD_RANGE = [-152.4, -152.4, -5.0, 152.4, 152.4, 5.0]


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_wrap(v) for v in x)
    return x


def load_config(path):
    """exec() a python config file; `_base_` entries are resolved if the files exist and skipped otherwise (the reference
    config inherits mmdetection3d/configs/_base_/default_runtime.py which only carries logging / runtime settings)."""
    ns = {}
    src = open(path).read()
    exec(compile(src, path, "exec"), ns)
    cfg = {}
    for b in ns.get("_base_", []) if isinstance(ns.get("_base_"), (list, tuple)) else [ns.get("_base_")]:
        if b:
            bp = os.path.normpath(os.path.join(os.path.dirname(path), b))
            if os.path.exists(bp):
                cfg.update(load_config(bp))
    cfg.update({k: v for k, v in ns.items() if not k.startswith("__") and k != "_base_" and not callable(v)})
    return _wrap(cfg)


def default_model_cfg(num_cams=7, num_query=644, num_propagated=256, memory_len=1024, topk_proposals=256, backbone="V-99-eSE",
                      proposal_topk=None, proposal_capacity=None):
    """The Far3D VoV-99 Argoverse-2 model (values of the reference's only config), assembled programmatically."""
    depthnet = dict(type=0, hidden_dim=256, num_depth_bins=50, depth_min=0.1, depth_max=110, stride=8)
    strides = [8, 16, 32, 64]
    self_attn = dict(type="MultiheadAttention", embed_dims=256, num_heads=8, dropout=0.1)
    cross_attn = dict(type="DeformableFeatureAggregationCuda", embed_dims=256, num_groups=8, num_levels=4, num_cams=num_cams,
                      dropout=0.1, num_pts=13, bias=2.0)
    layer = dict(type="Detr3DTemporalDecoderLayer", batch_first=True, attn_cfgs=[self_attn, cross_attn], feedforward_channels=2048,
                 ffn_dropout=0.1, with_cp=True, operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))
    head = dict(type="FarHead", num_classes=26, in_channels=256, num_query=num_query, memory_len=memory_len,
                topk_proposals=topk_proposals, num_propagated=num_propagated, with_dn=True, with_ego_pos=True,
                add_query_from_2d=True, depthnet_config=depthnet, add_multi_depth_proposal=True,
                multi_depth_config=dict(topk=1, range_min=30), return_bbox2d_scores=True, return_context_feat=True, code_size=8,
                code_weights=[1.0] * 8,
                transformer=dict(type="Detr3DTransformer",
                                 decoder=dict(type="Detr3DTransformerDecoder", embed_dims=256, num_layers=6, transformerlayers=layer)),
                bbox_coder=dict(type="NMSFreeCoder", post_center_range=POINT_CLOUD_RANGE, pc_range=POINT_CLOUD_RANGE, max_num=300,
                                voxel_size=[0.2, 0.2, 8], num_classes=26))
    roi = dict(type="YOLOXHeadCustom", num_classes=26, in_channels=256, strides=strides, pred_with_depth=True,
               depthnet_config=depthnet, reg_depth_level="p3", sample_with_score=True, threshold_score=0.1, topk_proposal=None,
               return_context_feat=True)
    return dict(type="Far3D", use_grid_mask=True, stride=strides, position_level=[0, 1, 2, 3],
                img_backbone=dict(type="VoVNet", spec_name=backbone, norm_eval=True, frozen_stages=-1, input_ch=3,
                                  out_features=("stage2", "stage3", "stage4", "stage5")),
                img_neck=dict(type="FPN", start_level=1, add_extra_convs="on_output", relu_before_extra_convs=True,
                              in_channels=[256, 512, 768, 1024], out_channels=256, num_outs=4),
                img_roi_head=roi, pts_bbox_head=head, proposal_topk=proposal_topk, proposal_capacity=proposal_capacity)
