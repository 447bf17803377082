"""Host-side mirror of the reference's `projects/mmdet3d_plugin` registry surface.

The reference resolves every model class from config `type` strings through OpenMMLab registries
(projects/configs/far3d.py:38-159; tools/test.py:134-155).  This package provides the same names -- `Far3D`, `VoVNet`,
`FPN`, `YOLOXHeadCustom`, `FarHead`, `Detr3DTransformer`, `Detr3DTransformerDecoder`, `Detr3DTemporalDecoderLayer`,
`MultiheadAttention`, `DeformableFeatureAggregationCuda` (aliases `SpatialDeformableAttention`,
`PerspectiveAwareAggregation`), `NMSFreeCoder` -- with the reference's constructor kwargs and state-dict keys, plus the
operator-level drop-in `MultiScaleDeformableAttnFunction.apply(...)` (mmcv contract).  mmcv / mmdet are NOT needed.

Modules here own parameters (reference names) and configuration; all arithmetic runs in far3d_amd.engine.Far3DEngine /
far3d_amd.ops on the HIP kernels.  There is no CPU execution path: forward on a box without a HIP device raises.
"""
import copy

import torch
import torch.nn as nn

from .. import engine as _engine
from .. import lib as _lib
from .. import ops, weights


class Registry:
    def __init__(self, name):
        self.name, self._m = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            for n in ([name] if isinstance(name, str) else (name or [cls.__name__])):
                self._m[n] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._m.get(key)

    def __contains__(self, key):
        return key in self._m

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        t = cfg.pop("type")
        cls = self._m.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError("%s is not in the %s registry (known: %s)" % (t, self.name, sorted(self._m)))
        return cls(**cfg)


DETECTORS, BACKBONES, NECKS, HEADS = Registry("detector"), Registry("backbone"), Registry("neck"), Registry("head")
ATTENTION, TRANSFORMER = Registry("attention"), Registry("transformer")
TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE = Registry("transformer_layer"), Registry("transformer_layer_sequence")
BBOX_CODERS = Registry("bbox_coder")
TRAIN_ONLY = {"HungarianAssigner3D", "FocalLoss", "L1Loss", "GIoULoss", "SimOTAAssigner", "FocalLossCost", "BBox3DL1Cost",
              "IoUCost", "CrossEntropyLoss", "IoULoss", "PseudoSampler"}   # tolerated in configs, never built (inference only)


def _register_schema(module, spec, prefix):
    """Create parameters / buffers named exactly like the reference's state dict (names may contain '/')."""
    for full, shape in spec.items():
        if not full.startswith(prefix):
            continue
        parts = full[len(prefix):].split(".")
        m = module
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, nn.Module())
            m = m._modules[p]
        if parts[-1] in ("running_mean", "running_var"):
            m.register_buffer(parts[-1], torch.zeros(shape))
        else:
            m.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


class _SchemaModule(nn.Module):
    """Base: parameters from far3d_amd.weights.detector_spec under `prefix`; inference only."""
    prefix = ""

    def _init_schema(self, spec):
        _register_schema(self, spec, self.prefix)

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("far3d_amd is an inference path; training stays with the reference")
        return super().train(False)

    def init_weights(self):
        pass

    # stand-alone use of one module (a maintainer swapping in a single registry class): a private engine holding only this
    # module's parts, rebuilt when the weights, device or precision change
    precision = "bf16"
    _eng = None

    def _part_engine(self, device, parts, cfg, extra_sd=None):
        ver = (sum(int(p._version) for p in self.parameters()), str(device), self.precision)
        if self._eng is None or self._eng_ver != ver:
            sd = {self.prefix + k: v for k, v in self.state_dict().items()}
            sd.update(extra_sd or {})
            self._eng = _engine.Far3DEngine(sd, cfg, device=device, precision=self.precision, parts=parts)
            self._eng_ver = ver
        return self._eng


def _to_nchw(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ operator drop-in
class MultiScaleDeformableAttnFunction:
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction stand-in (forward only).

    Same call as the reference (models/utils/detr3d_transformer.py:561-563):
        out = MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
                                                     sampling_locations, attention_weights, im2col_step)
    """

    @staticmethod
    def apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step=64):
        for t in (value, sampling_locations, attention_weights):
            if isinstance(t, torch.Tensor) and t.requires_grad:
                raise RuntimeError("far3d_amd MultiScaleDeformableAttnFunction is forward-only (inference path); "
                                   "backward is not implemented")
        return ops.msda_forward(value.contiguous(), value_spatial_shapes.contiguous(), value_level_start_index.contiguous(),
                                sampling_locations.contiguous(), attention_weights.contiguous(), im2col_step)


# ------------------------------------------------------------------------------------------------ attention modules
@ATTENTION.register_module(name=["DeformableFeatureAggregationCuda", "SpatialDeformableAttention", "PerspectiveAwareAggregation"])
class DeformableFeatureAggregationCuda(nn.Module):
    """Reference: models/utils/detr3d_transformer.py:483-569 (same kwargs, same parameter names, same forward args)."""

    def __init__(self, embed_dims=256, num_groups=8, num_levels=4, num_cams=6, dropout=0.1, num_pts=13, im2col_step=64,
                 batch_first=True, bias=1.0):
        super().__init__()
        self.embed_dims, self.num_groups, self.num_levels, self.num_cams = embed_dims, num_groups, num_levels, num_cams
        self.num_pts, self.im2col_step, self.bias, self.batch_first = num_pts, im2col_step, bias, batch_first
        self.group_dims = embed_dims // num_groups
        self.weights_fc = nn.Linear(embed_dims, num_groups * num_levels * num_pts)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.learnable_fc = nn.Linear(embed_dims, num_pts * 3)
        self.cam_embed = nn.Sequential(nn.Linear(12, embed_dims // 2), nn.ReLU(inplace=True),
                                       nn.Linear(embed_dims // 2, embed_dims), nn.ReLU(inplace=True), nn.LayerNorm(embed_dims))
        self.drop = nn.Dropout(dropout)
        self._packed = None
        self.precision = "bf16"
        for p in self.parameters():
            p.requires_grad_(False)

    def init_weight(self):   # reference :517-520
        nn.init.zeros_(self.weights_fc.weight); nn.init.zeros_(self.weights_fc.bias)
        nn.init.xavier_uniform_(self.output_proj.weight); nn.init.zeros_(self.output_proj.bias)
        nn.init.uniform_(self.learnable_fc.bias, -self.bias, self.bias)

    def _pack(self, dev):
        if self._packed is None or self._packed["dev"] != dev or self._packed["prec"] != self.precision:
            dw = _engine.PRECISIONS[self.precision]["dec"]
            pk = lambda lin, bias=True: ops.PackedConv(lin.weight.data, lin.bias.data if bias else None, dtype=dw, device=dev)
            self._packed = dict(dev=dev, prec=self.precision, wfc=pk(self.weights_fc, False), wfc_full=pk(self.weights_fc),
                                lfc=pk(self.learnable_fc), oproj=pk(self.output_proj), ce0=pk(self.cam_embed[0]), ce2=pk(self.cam_embed[2]),
                                ce_ln=(self.cam_embed[4].weight.data.float().to(dev), self.cam_embed[4].bias.data.float().to(dev)))
        return self._packed

    @torch.no_grad()
    def forward(self, instance_feature, query_pos, feat_flatten, reference_points, spatial_flatten, level_start_index,
                pc_range, lidar2img_mat, img_metas):
        _lib.require_device()
        bs, A = reference_points.shape[:2]
        if bs != 1:
            raise NotImplementedError("far3d_amd aggregation runs one sample per call (the reference tests with B=1)")
        dev = instance_feature.device
        P = self._pack(dev)
        x = instance_feature[0].float().contiguous()
        xq = (instance_feature[0] + query_pos[0]).float().contiguous()
        l2i = lidar2img_mat[0].float().contiguous()
        ce = ops.linear(ops.linear(l2i[:, :3, :].flatten(1).contiguous(), P["ce0"], act="relu"), P["ce2"], act="relu")
        ce = ops.layernorm(ce, *P["ce_ln"])
        Vc, U, offs = ops.linear(ce, P["wfc_full"]), ops.linear(xq, P["wfc"]), ops.linear(x, P["lfc"])
        hw = [tuple(int(v) for v in r) for r in spatial_flatten.tolist()]
        starts = [int(v) for v in level_start_index.tolist()]
        pad = img_metas[0]["pad_shape"][0]
        feat = feat_flatten
        if feat.dtype not in (torch.float32, torch.bfloat16):
            feat = feat.float()
        agg = ops.aggregate_forward(feat.contiguous(), reference_points[0].float().contiguous(), offs, l2i, U, Vc, hw, starts,
                                    [float(v) for v in pc_range.tolist()], (pad[0], pad[1]), num_groups=self.num_groups)
        return ops.linear(agg, P["oproj"], res=x)[None]   # dropout is identity at inference


@ATTENTION.register_module()
class MultiheadAttention(nn.Module):
    """mmcv MultiheadAttention wrapper semantics (in-tree statement: models/utils/petr_transformer.py:184-326)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, dropout_layer=None, init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__()
        kwargs.pop("dropout", None)
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, 0.0)   # parameter holder: in_proj_*, out_proj.*
        self.precision = "bf16"
        self._packed = None
        for p in self.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        """mmcv semantics (in-tree statement models/utils/petr_transformer.py:286-326): key/value default to query, identity to
        query, key_pos to query_pos when shapes agree; positions are added to q and k only; out = identity + out_proj(attn).
        One sample (the reference tests with B=1), no masks (inference)."""
        _lib.require_device()
        if attn_mask is not None or key_padding_mask is not None:
            raise NotImplementedError("far3d_amd MultiheadAttention: masks are a training-time (denoising) feature")
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        sq = (lambda t: t[0] if self.batch_first else t[:, 0])
        if (query.shape[0] if self.batch_first else query.shape[1]) != 1:
            raise NotImplementedError("far3d_amd MultiheadAttention runs one sample per call")
        q = sq(query) + (sq(query_pos) if query_pos is not None else 0)
        k = sq(key) + (sq(key_pos) if key_pos is not None else 0)
        v, idn = sq(value), sq(identity).float().contiguous()
        dev, E = q.device, self.embed_dims
        dw = _engine.PRECISIONS[self.precision]["dec"]
        ver = (int(self.attn.in_proj_weight._version), str(dev), self.precision)
        if self._packed is None or self._packed[0] != ver:
            w, b = self.attn.in_proj_weight.data, self.attn.in_proj_bias.data
            pk = lambda ww, bb: ops.PackedConv(ww, bb, dtype=dw, device=dev)
            self._packed = (ver, pk(w[:E], b[:E]), pk(w[E:2 * E], b[E:2 * E]), pk(w[2 * E:], b[2 * E:]),
                            pk(self.attn.out_proj.weight.data, self.attn.out_proj.bias.data))
        _, pq, pkk, pv, po = self._packed
        cast = lambda t: t.to(dw).contiguous()
        Q, K_, V = ops.linear(cast(q), pq, out_dtype=dw), ops.linear(cast(k), pkk, out_dtype=dw), ops.linear(cast(v), pv, out_dtype=dw)
        att = ops.attention_forward(Q, K_, V, num_heads=self.num_heads, out_dtype=dw)
        out = ops.linear(att, po, res=idn)
        return out[None] if self.batch_first else out[:, None]


# ------------------------------------------------------------------------------------------------ transformer containers
@TRANSFORMER_LAYER.register_module()
class Detr3DTemporalDecoderLayer(nn.Module):
    """models/utils/detr3d_transformer.py:192-480.  NB: like the reference, `feedforward_channels` / `ffn_dropout` passed
    at this level are swallowed by **kwargs and the FFN keeps its default hidden width 1024 (SURVEY.md finding 4)."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None, norm_cfg=None, init_cfg=None, batch_first=False,
                 with_cp=True, **kwargs):
        super().__init__()
        ffn_cfgs = ffn_cfgs or dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0.0)
        self.operation_order, self.batch_first = tuple(operation_order), batch_first
        assert self.operation_order == ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"), \
            "the HIP decoder implements the reference's post-norm order (projects/configs/far3d.py:130-131)"
        self.attentions = nn.ModuleList()
        for c in attn_cfgs:
            c = dict(c)
            c.setdefault("batch_first", batch_first)
            self.attentions.append(ATTENTION.build(c))
        self.embed_dims = self.attentions[0].embed_dims
        self.ffn_dim = ffn_cfgs.get("feedforward_channels", 1024)
        ffn = nn.Module()
        ffn.layers = nn.Sequential(nn.Sequential(nn.Linear(self.embed_dims, self.ffn_dim), nn.ReLU(inplace=True), nn.Dropout(0.0)),
                                   nn.Linear(self.ffn_dim, self.embed_dims), nn.Dropout(0.0))
        self.ffns = nn.ModuleList([ffn])
        self.norms = nn.ModuleList([nn.LayerNorm(self.embed_dims) for _ in range(3)])
        for p in self.parameters():
            p.requires_grad_(False)


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class Detr3DTransformerDecoder(nn.Module):
    def __init__(self, embed_dims=256, transformerlayers=None, num_layers=6, init_cfg=None, **kwargs):
        super().__init__()
        self.embed_dims, self.num_layers = embed_dims, num_layers
        self.layers = nn.ModuleList([TRANSFORMER_LAYER.build(copy.deepcopy(transformerlayers)) for _ in range(num_layers)])


@TRANSFORMER.register_module()
class Detr3DTransformer(nn.Module):
    def __init__(self, decoder=None, **kwargs):
        super().__init__()
        self.decoder = TRANSFORMER_LAYER_SEQUENCE.build(decoder)

    def init_weights(self):
        for m in self.modules():
            if hasattr(m, "init_weight"):
                m.init_weight()


# ------------------------------------------------------------------------------------------------ coder
@BBOX_CODERS.register_module()
class NMSFreeCoder:
    """core/bbox/coders/nms_free_coder.py:8-112 (decode_single / decode), on whatever device the tensors live."""

    def __init__(self, pc_range, voxel_size=None, post_center_range=None, max_num=100, score_threshold=None, num_classes=10):
        self.pc_range, self.voxel_size, self.post_center_range = pc_range, voxel_size, post_center_range
        self.max_num, self.score_threshold, self.num_classes = max_num, score_threshold, num_classes

    def decode_single(self, cls_scores, bbox_preds):
        cls_scores = cls_scores.sigmoid()
        scores, idx = cls_scores.view(-1).topk(min(self.max_num, cls_scores.numel()))
        labels = idx % self.num_classes
        b = bbox_preds[torch.div(idx, self.num_classes, rounding_mode="floor")]
        boxes = torch.cat([b[..., 0:3], b[..., 3:6].exp(), torch.atan2(b[..., 6:7], b[..., 7:8])], dim=-1)
        if self.post_center_range is None:
            raise NotImplementedError("post_center_range is required (as in the reference)")
        rng = torch.as_tensor(self.post_center_range, device=scores.device, dtype=boxes.dtype)
        mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
        if self.score_threshold:
            mask &= scores >= self.score_threshold
        return dict(bboxes=boxes[mask], scores=scores[mask], labels=labels[mask])

    def decode(self, preds_dicts):
        cls, box = preds_dicts["all_cls_scores"][-1], preds_dicts["all_bbox_preds"][-1]
        return [self.decode_single(cls[i], box[i]) for i in range(cls.shape[0])]


# ------------------------------------------------------------------------------------------------ per-camera modules
@BACKBONES.register_module()
class VoVNet(_SchemaModule):
    """models/backbones/vovnet.py:276-384.  forward(x NCHW f32) -> list of NCHW stage maps, computed by the HIP engine."""
    prefix = "img_backbone."

    def __init__(self, spec_name, input_ch=3, out_features=None, frozen_stages=-1, norm_eval=True, pretrained=None, init_cfg=None):
        super().__init__()
        if spec_name not in weights.VOV_SPECS:
            raise KeyError("VoVNet spec %s not supported (have %s)" % (spec_name, sorted(weights.VOV_SPECS)))
        self.spec_name, self._out_features = spec_name, tuple(out_features or ("stage2", "stage3", "stage4", "stage5"))
        self._init_schema(weights.backbone_spec(spec_name, input_ch))
        self._eng = None
        self.precision = "bf16"

    @torch.no_grad()
    def forward(self, x):
        _lib.require_device()
        if self._eng is None or self._eng.dev != x.device or self._eng.precision != self.precision:
            sd = {self.prefix + k: v for k, v in self.state_dict().items()}
            self._eng = _engine.Far3DEngine(sd, _engine.default_cfg(backbone=self.spec_name), device=x.device,
                                            precision=self.precision, parts=("backbone",))
        outs = self._eng.backbone(x.float().contiguous())
        names = ["stage2", "stage3", "stage4", "stage5"]
        return [self._eng.act_to_nchw(o) for n, o in zip(names, outs) if n in self._out_features]


@NECKS.register_module()
class FPN(_SchemaModule):
    """mmdet FPN as configured at projects/configs/far3d.py:50-57 (parameter holder; computed inside Far3D)."""
    prefix = "img_neck."

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, **kwargs):
        super().__init__()
        assert start_level == 1 and add_extra_convs == "on_output" and num_outs == len(in_channels), \
            "only the reference's FPN configuration (start_level=1, add_extra_convs='on_output') is implemented"
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        spec = {}
        for i, c in enumerate(in_channels[1:]):
            spec["img_neck.lateral_convs.%d.conv.weight" % i] = (out_channels, c, 1, 1)
            spec["img_neck.lateral_convs.%d.conv.bias" % i] = (out_channels,)
        for i in range(num_outs):
            spec["img_neck.fpn_convs.%d.conv.weight" % i] = (out_channels, out_channels, 3, 3)
            spec["img_neck.fpn_convs.%d.conv.bias" % i] = (out_channels,)
        self._init_schema(spec)

    @torch.no_grad()
    def forward(self, inputs):
        """inputs: the backbone's stage maps (NCHW, the first one is skipped by start_level=1) -> tuple of num_outs NCHW maps,
        computed by the engine's FPN stage (laterals + fused nearest-upsample add + 3x3 output convs + stride-2 extra level)."""
        _lib.require_device()
        assert len(inputs) == len(self.in_channels)
        dev = inputs[0].device
        name = next(k for k, v in weights.VOV_SPECS.items() if tuple(v["stage_out_ch"]) == tuple(self.in_channels))
        eng = self._part_engine(dev, ("neck",), _engine.default_cfg(backbone=name))
        feats = [eng.act_from_nchw(x) for x in inputs]
        n = feats[0].shape[0]
        one, zero = torch.ones(n, self.out_channels, device=dev), torch.zeros(n, self.out_channels, device=dev)
        raw, _, _, _ = eng.fpn(feats, one, zero)
        return tuple(eng.act_to_nchw(r) for r in raw)


@HEADS.register_module()
class YOLOXHeadCustom(_SchemaModule):
    """models/dense_heads/yolox_head.py:25-519 (inference members; parameter holder, computed inside Far3D)."""
    prefix = "img_roi_head."

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=2, strides=(8, 16, 32), pred_with_depth=False,
                 depthnet_config=None, reg_depth_level="p4", sample_with_score=True, threshold_score=0.05, topk_proposal=None,
                 return_context_feat=False, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        assert pred_with_depth and reg_depth_level == "p3" and stacked_convs == 2 and feat_channels == in_channels, \
            "only the reference's 2D-head configuration is implemented"
        self.num_classes, self.strides, self.depthnet_config = num_classes, list(strides), dict(depthnet_config or {})
        self.sample_with_score, self.threshold_score, self.topk_proposal = sample_with_score, threshold_score, topk_proposal
        full = weights.detector_spec(num_classes=num_classes, embed=in_channels, fpn_levels=len(strides),
                                     depth_bins=self.depthnet_config.get("num_depth_bins", 50))
        self._init_schema(full)

    def _engine_for(self, dev):
        cfg = _engine.default_cfg(num_classes=self.num_classes, strides=tuple(self.strides), score_thr=self.threshold_score,
                                  proposal_topk=self.topk_proposal,
                                  depthnet=dict(num_depth_bins=self.depthnet_config.get("num_depth_bins", 50),
                                                depth_min=self.depthnet_config.get("depth_min", 0.1),
                                                depth_max=self.depthnet_config.get("depth_max", 110.0),
                                                stride=self.depthnet_config.get("stride", 8)))
        return self._part_engine(dev, ("roi",), cfg)

    @torch.no_grad()
    def forward(self, locations, **data):
        """ref yolox_head.py:260-341.  data['img_feats']: list of (B,N,C,h,w) FPN maps.  Returns the reference's dict
        (enc_cls_scores / enc_bbox_preds / objectnesses / pred_centers2d_offset lists of (BN,c,h,w), depth_logit, pred_depth,
        topk_indexes=None); the NHWC device maps the HIP proposal kernels consume ride along under '_far3d'."""
        _lib.require_device()
        feats = data["img_feats"]
        dev = feats[0].device
        eng = self._engine_for(dev)
        raw = [eng.act_from_nchw(f.flatten(0, 1)) for f in feats]
        ctr = []
        cls, reg, depth_logit = eng.roi_head(raw, centers2d=ctr)
        out = dict(enc_cls_scores=[_to_nchw(c) for c in cls], enc_bbox_preds=[_to_nchw(r[..., :4]) for r in reg],
                   pred_centers2d_offset=[_to_nchw(c) for c in ctr], objectnesses=[_to_nchw(r[..., 4:5]) for r in reg],
                   topk_indexes=None, depth_logit=_to_nchw(depth_logit))
        out["pred_depth"] = out["depth_logit"].softmax(dim=1)
        out["_far3d"] = dict(cls=cls, reg=reg, depth_logit=depth_logit)
        return out

    @torch.no_grad()
    def get_bboxes(self, preds_dicts):
        """ref yolox_head.py:355-489: peak test, score threshold, 2D box decode.  Returns bbox_list (per camera (M_i,4) cxcywh),
        bbox2d_scores (M,1), valid_indices (BN,S,1) bool -- built from the device-side ordered selection."""
        st = preds_dicts.get("_far3d")
        if st is None:
            raise NotImplementedError("YOLOXHeadCustom.get_bboxes needs the dict returned by this module's forward")
        dev = st["cls"][0].device
        eng = self._engine_for(dev)
        cfg = eng.cfg
        n = st["cls"][0].shape[0]
        S = sum(c.shape[1] * c.shape[2] for c in st["cls"])
        K = cfg["proposal_topk"]
        cap = K if K is not None else S
        wgt, sel_idx, sel_cnt = ops.proposal_select(st["cls"], st["reg"], cfg["strides"], cap, thr=cfg["score_thr"], topk=K is not None)
        cnt = sel_cnt.cpu().tolist()
        valid = torch.zeros((n, S, 1), dtype=torch.bool, device=dev)
        for i in range(n):
            valid[i, sel_idx[i, :cnt[i]].long(), 0] = True
        # 2D boxes: the gather kernel decodes them (identity img2lidar / dummy tokens: only box2d and score are read here)
        eye = torch.eye(4, device=dev)[None].repeat(n, 1, 1).contiguous()
        tok = torch.zeros((n, S, cfg["embed_dims"]), device=dev)
        _, _, box2d, score2d = ops.proposal_gather(st["reg"], cfg["strides"], sel_idx, sel_cnt, wgt, st["depth_logit"], cfg["depthnet"]["stride"],
                                                   cfg["depthnet"], eye, tok, cfg["pc_range"], score_thr=0.1)
        M = sum(cnt)
        off = [0]
        for c in cnt:
            off.append(off[-1] + c)
        st.update(sel_idx=sel_idx, sel_cnt=sel_cnt, peak_weight=wgt)
        return dict(bbox_list=[box2d[off[i]:off[i + 1]] for i in range(n)], bbox2d_scores=score2d[:M, None], valid_indices=valid)


@HEADS.register_module()
class FarHead(_SchemaModule):
    """models/dense_heads/farhead.py:20-1245 (inference members; parameter holder, computed inside Far3D)."""
    prefix = "pts_bbox_head."

    def __init__(self, num_classes, in_channels=256, stride=16, embed_dims=256, num_query=100, memory_len=1024,
                 topk_proposals=256, num_propagated=256, with_dn=True, with_ego_pos=True, add_query_from_2d=False,
                 depthnet_config=None, multi_depth_config=None, return_context_feat=False, return_bbox2d_scores=False,
                 transformer=None, bbox_coder=None, code_size=10, code_weights=None, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        assert with_ego_pos and add_query_from_2d and return_context_feat and return_bbox2d_scores, \
            "only the reference's FarHead configuration (2D-adaptive queries with context + score) is implemented"
        assert (multi_depth_config or {}).get("topk", 1) == 1, "multi-depth proposals with topk > 1 are not implemented"
        self.num_classes, self.embed_dims, self.num_query, self.memory_len = num_classes, embed_dims, num_query, memory_len
        self.topk_proposals, self.num_propagated, self.code_size = topk_proposals, num_propagated, code_size
        self.depthnet_config = dict(depthnet_config or {})
        self.transformer = TRANSFORMER.build(transformer)
        self.bbox_coder = BBOX_CODERS.build(bbox_coder)
        layer0 = self.transformer.decoder.layers[0]
        agg = layer0.attentions[1]
        self.num_layers = self.transformer.decoder.num_layers
        full = weights.detector_spec(num_query=num_query, num_propagated=num_propagated, num_classes=num_classes, embed=embed_dims,
                                     num_layers=self.num_layers, num_levels=agg.num_levels, num_pts=agg.num_pts,
                                     num_groups=agg.num_groups, ffn_dim=layer0.ffn_dim, code_size=code_size)
        own = {k: v for k, v in full.items() if k.startswith(self.prefix) and ".transformer." not in k}
        self._init_schema(own)
        with torch.no_grad():
            self.pc_range.copy_(torch.tensor(self.bbox_coder.pc_range, dtype=torch.float32))
            self.code_weights.fill_(1.0); self.match_costs.fill_(1.0)
        self.agg_cfg = dict(num_cams=agg.num_cams, num_groups=agg.num_groups, num_levels=agg.num_levels, num_pts=agg.num_pts,
                            num_heads=layer0.attentions[0].num_heads, ffn_dim=layer0.ffn_dim)

    def engine_cfg(self, **over):
        a = self.agg_cfg
        cfg = dict(embed_dims=self.embed_dims, num_classes=self.num_classes, num_cams=a["num_cams"], num_query=self.num_query,
                   num_propagated=self.num_propagated, memory_len=self.memory_len, topk_proposals=self.topk_proposals,
                   num_layers=self.num_layers, num_heads=a["num_heads"], num_groups=a["num_groups"], num_levels=a["num_levels"],
                   num_pts=a["num_pts"], ffn_dim=a["ffn_dim"], pc_range=list(self.bbox_coder.pc_range), code_size=self.code_size,
                   max_num=self.bbox_coder.max_num,
                   depthnet=dict(num_depth_bins=self.depthnet_config.get("num_depth_bins", 50), depth_min=self.depthnet_config.get("depth_min", 0.1),
                                 depth_max=self.depthnet_config.get("depth_max", 110.0), stride=self.depthnet_config.get("stride", 8)))
        cfg.update(over)
        return _engine.default_cfg(**cfg)

    def reset_memory(self):
        if self._eng is not None:
            self._eng.reset_memory()

    @torch.no_grad()
    def forward(self, img_metas, outs_roi=None, **data):
        """ref farhead.py:533-693 (inference branch): data['img_feats'] list of (1,N,C,h,w) FPN maps, intrinsics / extrinsics /
        lidar2img (1,N,4,4), ego_pose / ego_pose_inv (1,4,4), timestamp (1,), prev_exists (1,); outs_roi = the dict built by this
        package's YOLOXHeadCustom forward + get_bboxes.  Returns all_cls_scores (layers,1,A,classes), all_bbox_preds
        (layers,1,A,code) and dn_mask_dict=None like the reference; the streaming memory lives in the module's engine."""
        _lib.require_device()
        feats = data["img_feats"]
        dev = feats[0].device
        if outs_roi is None or "_far3d" not in outs_roi or "sel_idx" not in outs_roi["_far3d"]:
            raise NotImplementedError("FarHead.forward needs outs_roi from far3d_amd's YOLOXHeadCustom (forward + get_bboxes): the "
                                      "adaptive queries are built by the HIP proposal kernels from its device-side selection")
        st = outs_roi["_far3d"]
        eng = self._part_engine(dev, ("head",), self.engine_cfg(strides=tuple(2 ** (3 + i) for i in range(len(feats)))))
        N = feats[0].shape[1]
        E = self.embed_dims
        f32 = lambda t: t.to(dev).float().contiguous()
        dd = dict(lidar2img=f32(data["lidar2img"]), intrinsics=f32(data["intrinsics"]), extrinsics=f32(data["extrinsics"]),
                  ego_pose=f32(data["ego_pose"]), ego_pose_inv=f32(data["ego_pose_inv"]), timestamp=data["timestamp"].to(dev).double().contiguous())
        if "prev_exists" in data and float(data["prev_exists"].flatten()[0]) == 0.0:
            eng.reset_memory()
        # camera-aware MLN of every token (farhead.py:553-563) -> token-major value maps
        img2lidar, c14 = ops.camera_prep(dd["lidar2img"][0], dd["intrinsics"][0], dd["extrinsics"][0])
        hh = eng.sa["reduce"](c14, act="relu")
        gamma, beta = eng.sa["gamma"](hh), eng.sa["beta"](hh)
        hw = [(f.shape[3], f.shape[4]) for f in feats]
        from ..synth import level_starts
        starts, S = level_starts(hw)
        tokens = torch.empty((N, S, E), dtype=torch.float32, device=dev)
        for l, f in enumerate(feats):
            x = f[0].float().permute(0, 2, 3, 1).contiguous()          # (N,h,w,C)
            for n in range(N):
                ops.row_affine_ln(x[n].view(-1, E), gamma[n:n + 1], beta[n:n + 1], do_ln=False,
                                  out=tokens[n, starts[l]:starts[l] + hw[l][0] * hw[l][1]])
        tokens = tokens.to(eng.prec["value"])
        M = int(st["sel_cnt"].sum().item())
        pr = ops.proposal_gather(st["reg"], eng.cfg["strides"], st["sel_idx"], st["sel_cnt"], st["peak_weight"], st["depth_logit"],
                                 eng.cfg["depthnet"]["stride"], eng.cfg["depthnet"], img2lidar, tokens, eng.cfg["pc_range"], score_thr=0.1)
        pad_hw = tuple(img_metas[0]["pad_shape"][0][:2])
        outs = eng.head_stage(tokens, pr[0], pr[1], M, dd, img_metas, hw, starts, pad_hw)
        self.last_outs = outs
        return dict(all_cls_scores=outs["all_cls_scores"], all_bbox_preds=outs["all_bbox_preds"], dn_mask_dict=None,
                    reference_points2d=pr[0][:M][None], _far3d_result=outs["result"])

    @torch.no_grad()
    def get_bboxes(self, preds_dicts, img_metas, rescale=False):
        """ref farhead.py:1224-1245 -> [[boxes, scores, labels]] (boxes wrapped in img_metas[0]['box_type_3d'] when given)."""
        r = preds_dicts.get("_far3d_result")
        if r is None:
            r = self._eng.decode(preds_dicts["all_cls_scores"], preds_dicts["all_bbox_preds"])
        keep = r["keep"]
        boxes, scores, labels = r["boxes_3d"][keep], r["scores_3d"][keep], r["labels_3d"][keep]
        box_type = img_metas[0].get("box_type_3d")
        if box_type is not None:
            boxes = box_type(boxes, boxes.size(-1))
        return [[boxes, scores, labels]]


# ------------------------------------------------------------------------------------------------ detector
@DETECTORS.register_module()
class Far3D(nn.Module):
    """models/detectors/far3d.py:21-278 -- inference surface: forward(return_loss=False, ...) / forward_test / simple_test."""

    def __init__(self, use_grid_mask=False, img_backbone=None, img_neck=None, pts_bbox_head=None, img_roi_head=None,
                 train_cfg=None, test_cfg=None, stride=(16,), position_level=(0,), aux_2d_only=True, single_test=False,
                 pretrained=None, proposal_topk=None, proposal_capacity=None, **kwargs):
        super().__init__()
        self.img_backbone = BACKBONES.build(img_backbone)
        self.img_neck = NECKS.build(img_neck)
        self.pts_bbox_head = HEADS.build(pts_bbox_head, train_cfg=None, test_cfg=None)
        self.img_roi_head = HEADS.build(img_roi_head)
        self.stride, self.position_level = list(stride), list(position_level)
        self.use_grid_mask = use_grid_mask     # GridMask is the identity in eval mode (models/utils/grid_mask.py:85)
        self.proposal_topk = proposal_topk     # build-side extension: static K proposals per camera (None = reference)
        # build-side extension: the reference's threshold rule with static shapes -- rows reserved for the adaptive queries, count on
        # the device (None = the legacy form with a host sync on M); see INTEGRATION.md "Proposal modes"
        self.proposal_capacity = proposal_capacity
        self.engine = None
        self.eval()

    # -- weights ------------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        canon = {}
        for k, v in state_dict.items():
            ck = weights.canonical_key(k)
            if ck is not None:
                canon[ck] = v
        self.engine = None
        return super().load_state_dict(canon, strict=strict)

    def init_weights(self, seed=0):
        sd = weights.init_state_dict({k: tuple(v.shape) for k, v in self.state_dict().items()}, seed=seed,
                                     pc_range=self.pts_bbox_head.bbox_coder.pc_range)
        self.load_state_dict(sd)

    def engine_cfg(self):
        h, r = self.pts_bbox_head, self.img_roi_head
        a = h.agg_cfg
        return _engine.default_cfg(
            backbone=self.img_backbone.spec_name, embed_dims=h.embed_dims, num_classes=h.num_classes, strides=tuple(r.strides),
            num_cams=a["num_cams"], num_query=h.num_query, num_propagated=h.num_propagated, memory_len=h.memory_len,
            topk_proposals=h.topk_proposals, num_layers=h.num_layers, num_heads=a["num_heads"], num_groups=a["num_groups"],
            num_levels=a["num_levels"], num_pts=a["num_pts"], ffn_dim=a["ffn_dim"], pc_range=list(h.bbox_coder.pc_range),
            code_size=h.code_size, max_num=h.bbox_coder.max_num,
            depthnet=dict(num_depth_bins=r.depthnet_config.get("num_depth_bins", 50), depth_min=r.depthnet_config.get("depth_min", 0.1),
                          depth_max=r.depthnet_config.get("depth_max", 110.0), stride=r.depthnet_config.get("stride", 8)),
            score_thr=r.threshold_score, proposal_topk=self.proposal_topk, proposal_capacity=self.proposal_capacity)

    def prepare(self, device="cuda:0", precision="bf16"):
        """Fold BN, pack weights for the kernels, upload.  Must be called again after loading new weights."""
        self.engine = _engine.Far3DEngine(self.state_dict(), self.engine_cfg(), device=device, precision=precision)
        return self

    # -- reference inference entry points -------------------------------------------------------------------------------
    def forward(self, return_loss=True, **data):
        if return_loss:
            raise NotImplementedError("far3d_amd implements the inference path only (return_loss=False)")
        return self.forward_test(**data)

    def forward_test(self, img_metas, rescale=True, **data):   # detectors/far3d.py:232-242
        if not isinstance(img_metas, list):
            raise TypeError("img_metas must be a list, but got %s" % type(img_metas))
        if isinstance(img_metas[0], list):   # test-time-augmentation wrapping of the mmdet pipeline
            img_metas = img_metas[0]
            data = {k: (v[0][0].unsqueeze(0) if k not in ("img", "gt_bboxes_3d", "gt_bboxes", "centers2d") else v[0])
                    for k, v in data.items()}
        return self.simple_test(img_metas, **data)

    @torch.no_grad()
    def simple_test(self, img_metas, **data):                   # detectors/far3d.py:268-277
        if self.engine is None:
            raise _lib.Far3dHipError("call Far3D.prepare(device, precision) before inference")
        outs = self.engine.forward_frame(data, img_metas)
        self.engine.wait_outputs()       # pipeline mode produces the outputs on the head stream (no-op otherwise)
        if self.proposal_capacity is not None:
            self.engine.check_proposal_overflow()      # a frame that did not fit the reserved rows is an error, never a silent truncation
        r = outs["result"]
        keep = r["keep"]
        res = dict(boxes_3d=r["boxes_3d"][keep], scores_3d=r["scores_3d"][keep], labels_3d=r["labels_3d"][keep])
        box_type = img_metas[0].get("box_type_3d")
        if box_type is not None:
            res["boxes_3d"] = box_type(res["boxes_3d"], res["boxes_3d"].size(-1))
        self.last_outs = outs
        return [dict(pts_bbox=res)]


def build_detector(cfg, train_cfg=None, test_cfg=None):
    cfg = {k: v for k, v in dict(cfg).items()}
    return DETECTORS.build(cfg)
