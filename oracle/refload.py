"""Load the reference's OWN hot-path files by path, under minimal stand-ins for mmcv / mmdet / mmdet3d.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY: needs /root/reference (absent on the GPU box).  Used by
tools/gen_golden.py to (i) check oracle/far3d_oracle.py against the reference's in-tree code and
(ii) emit the golden fixtures under tests/golden/.  Nothing of the reference is copied: its files are
executed where they lie.

The stand-ins restate the published semantics of the third-party pieces the reference calls but does
not contain (SURVEY.md §8(c) "Third-party arithmetic"): registries, ConvModule(conv->BN->act, Swish),
FFN (x + W2 relu(W1 x)), MultiheadAttention (pos added to q/k only, identity + out), mmdet FPN,
MlvlPointGenerator(offset=0), inverse_sigmoid(eps=1e-5), and MSDA via grid_sample.
"""
import copy
import functools
import importlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("FAR3D_REFERENCE", "/root/reference")
PLUGIN = os.path.join(REF_ROOT, "projects", "mmdet3d_plugin")


def available():
    return os.path.isdir(PLUGIN)


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _to_configdict(x):
    if isinstance(x, dict):
        return ConfigDict({k: _to_configdict(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_to_configdict(v) for v in x)
    return x


class Registry:
    def __init__(self, name):
        self.name, self.module_dict = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, k):
        return self.module_dict.get(k)

    def build(self, cfg, **default):
        cfg = _to_configdict(dict(cfg))
        for k, v in default.items():
            cfg.setdefault(k, v)
        t = cfg.pop("type")
        cls = self.module_dict[t] if isinstance(t, str) else t
        return cls(**cfg)


def build_from_cfg(cfg, registry, default_args=None):
    return registry.build(cfg, **(default_args or {}))


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class _Noop:
    """Stands for any loss / assigner / sampler object the inference path never calls."""
    use_sigmoid = True

    def __init__(self, *a, **k):
        self.__dict__.update(k)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


def build_norm_layer(cfg, num_features, postfix=""):
    t = cfg["type"]
    if t == "LN":
        return "ln" + str(postfix), nn.LayerNorm(num_features, eps=cfg.get("eps", 1e-5))
    if t == "BN":
        return "bn" + str(postfix), nn.BatchNorm2d(num_features, eps=cfg.get("eps", 1e-5), momentum=cfg.get("momentum", 0.1))
    if t == "GN":
        return "gn" + str(postfix), nn.GroupNorm(cfg["num_groups"], num_features)
    raise KeyError(t)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **kw):
        super().__init__()
        if bias == "auto":
            bias = norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.norm_name = None
        if norm_cfg is not None:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        self.activate = None
        if act_cfg is not None:
            self.activate = {"ReLU": nn.ReLU(), "Swish": Swish()}[act_cfg["type"]]

    def forward(self, x):
        x = self.conv(x)
        if self.norm_name:
            x = getattr(self, self.norm_name)(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type="ReLU", inplace=True),
                 ffn_drop=0.0, dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__(init_cfg)
        layers, c = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(c, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
            c = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity
        self.embed_dims = embed_dims

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out


class MultiheadAttention(BaseModule):
    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, dropout_layer=None, init_cfg=None,
                 batch_first=False, **kw):
        super().__init__(init_cfg)
        if "dropout" in kw:
            attn_drop = kw.pop("dropout")
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kw)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kw):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = (t.transpose(0, 1).contiguous() for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1).contiguous()
        return identity + out


ATTENTION, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE = Registry("attention"), Registry("tl"), Registry("tls")
PLUGIN_LAYERS, POSITIONAL_ENCODING, FEEDFORWARD_NETWORK = Registry("plugin"), Registry("pe"), Registry("ffn")
TRANSFORMER, HEADS, BACKBONES, NECKS, DETECTORS, BBOX_CODERS = (Registry(n) for n in
                                                                ("transformer", "heads", "backbones", "necks", "detectors", "coders"))
ATTENTION.register_module(module=MultiheadAttention)
FEEDFORWARD_NETWORK.register_module(module=FFN)


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        cfgs = [copy.deepcopy(transformerlayers) for _ in range(num_layers)] if isinstance(transformerlayers, dict) else transformerlayers
        self.num_layers = num_layers
        self.layers = nn.ModuleList([TRANSFORMER_LAYER.build(copy.deepcopy(c)) for c in cfgs])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


class FPN(BaseModule):
    """mmdet 2.28 FPN arithmetic (lateral 1x1 -> nearest top-down -> 3x3 -> extra stride-2 convs)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__(init_cfg)
        self.in_channels, self.num_outs, self.start_level = in_channels, num_outs, start_level
        self.backbone_end_level = len(in_channels) if end_level == -1 else end_level + 1
        self.add_extra_convs = "on_input" if add_extra_convs is True else add_extra_convs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))
        extra = num_outs - self.backbone_end_level + start_level
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == "on_input") else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))

    def forward(self, inputs):
        lats = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        n = len(lats)
        for i in range(n - 1, 0, -1):
            lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode="nearest")
        outs = [self.fpn_convs[i](lats[i]) for i in range(n)]
        if self.num_outs > len(outs):
            src = {"on_input": inputs[self.backbone_end_level - 1], "on_lateral": lats[-1], "on_output": outs[-1]}[self.add_extra_convs]
            outs.append(self.fpn_convs[n](src))
            for i in range(n + 1, self.num_outs):
                outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)


NECKS.register_module(module=FPN)


class MlvlPointGenerator:
    def __init__(self, strides, offset=0.5):
        self.strides, self.offset = [(s, s) if isinstance(s, int) else s for s in strides], offset

    def grid_priors(self, featmap_sizes, dtype=torch.float32, device="cpu", with_stride=False):
        return [self.single_level_grid_priors(s, i, dtype, device, with_stride) for i, s in enumerate(featmap_sizes)]

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32, device="cpu", with_stride=False):
        h, w = featmap_size
        sw, sh = self.strides[level_idx]
        sx = ((torch.arange(0, w, device=device) + self.offset) * sw).to(dtype)
        sy = ((torch.arange(0, h, device=device) + self.offset) * sh).to(dtype)
        xx, yy = sx.repeat(h), sy.view(-1, 1).repeat(1, w).view(-1)
        if not with_stride:
            return torch.stack([xx, yy], -1)
        return torch.stack([xx, yy, xx.new_full((xx.shape[0],), sw), xx.new_full((xx.shape[0],), sh)], -1)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def bbox_xyxy_to_cxcywh(b):
    x1, y1, x2, y2 = b.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def multi_apply(func, *args, **kwargs):
    res = map(functools.partial(func, **kwargs) if kwargs else func, *args)
    return tuple(map(list, zip(*res)))


def _deco_factory(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


class _MSDA:
    @staticmethod
    def apply(value, shapes, lsi, loc, w, im2col_step):
        from oracle import sampling
        return sampling.msda_grid_sample(value, shapes, lsi, loc, w)


class AnchorFreeHead(BaseModule):
    def __init__(self, num_classes, in_channels, init_cfg=None, **kw):
        super().__init__(init_cfg)


class BaseDenseHead(BaseModule):
    pass


class BBoxTestMixin:
    pass


class BaseBBoxCoder:
    pass


class LiDARBoxes:
    """mmdet3d LiDARInstance3DBoxes stand-in: (x, y, z_bottom, w, l, h, yaw, ...) rows; gravity_center = bottom centre + h/2."""

    def __init__(self, tensor, box_dim=7):
        self.tensor, self.box_dim = tensor, box_dim

    @property
    def gravity_center(self):
        t = self.tensor
        return torch.cat([t[:, :2], t[:, 2:3] + t[:, 5:6] * 0.5], dim=1)


def bbox3d2result(bboxes, scores, labels):
    return dict(boxes_3d=bboxes, scores_3d=scores.cpu(), labels_3d=labels.cpu())


class MVXTwoStageDetector(BaseModule):
    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None, pts_fusion_layer=None,
                 img_backbone=None, pts_backbone=None, img_neck=None, pts_neck=None, pts_bbox_head=None,
                 img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.img_backbone = BACKBONES.build(img_backbone)
        self.img_neck = NECKS.build(img_neck)
        self.pts_bbox_head = HEADS.build(pts_bbox_head, train_cfg=None, test_cfg=None)
        self.img_roi_head = HEADS.build(img_roi_head)

    @property
    def with_img_neck(self):
        return self.img_neck is not None


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_installed = False


def install():
    """Seed sys.modules with the stand-ins and with path-only parents for the reference packages."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference checkout not found at %s" % REF_ROOT)
    if REF_ROOT not in sys.path:
        pass  # nothing of the reference is put on sys.path; packages below carry explicit __path__
    build_any = lambda cfg, *a, **k: _Noop(**(cfg if isinstance(cfg, dict) else {}))
    _mod("mmcv")
    _mod("mmcv.cnn", xavier_init=lambda m, **k: None, constant_init=lambda m, **k: None, build_norm_layer=build_norm_layer,
         Linear=nn.Linear, bias_init_with_prob=lambda p: float(-torch.log(torch.tensor((1 - p) / p))), Scale=nn.Identity,
         ConvModule=ConvModule, DepthwiseSeparableConvModule=ConvModule)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.transformer", BaseTransformerLayer=BaseModule, TransformerLayerSequence=TransformerLayerSequence,
         build_transformer_layer_sequence=lambda cfg: TRANSFORMER_LAYER_SEQUENCE.build(cfg),
         build_attention=lambda cfg: ATTENTION.build(cfg),
         build_feedforward_network=lambda cfg, default=None: FEEDFORWARD_NETWORK.build(cfg),
         build_positional_encoding=build_any, POSITIONAL_ENCODING=POSITIONAL_ENCODING, FFN=FFN)
    _mod("mmcv.cnn.bricks.registry", ATTENTION=ATTENTION, TRANSFORMER_LAYER=TRANSFORMER_LAYER,
         TRANSFORMER_LAYER_SEQUENCE=TRANSFORMER_LAYER_SEQUENCE, PLUGIN_LAYERS=PLUGIN_LAYERS)
    _mod("mmcv.ops")
    _mod("mmcv.ops.multi_scale_deform_attn", MultiScaleDeformableAttnFunction=_MSDA)
    _mod("mmcv.ops.nms", batched_nms=None)
    _mod("mmcv.runner", force_fp32=_deco_factory, auto_fp16=_deco_factory, BaseModule=BaseModule)
    _mod("mmcv.runner.base_module", BaseModule=BaseModule)
    _mod("mmcv.utils", deprecated_api_warning=_deco_factory, ConfigDict=ConfigDict, build_from_cfg=build_from_cfg)
    _mod("mmdet")
    _mod("mmdet.core", build_assigner=build_any, build_sampler=build_any, multi_apply=multi_apply, reduce_mean=lambda x: x,
         MlvlPointGenerator=MlvlPointGenerator, bbox_xyxy_to_cxcywh=bbox_xyxy_to_cxcywh)
    _mod("mmdet.core.bbox", BaseBBoxCoder=BaseBBoxCoder)
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=BBOX_CODERS)
    _mod("mmdet.models", HEADS=HEADS, build_loss=build_any, DETECTORS=DETECTORS)
    _mod("mmdet.models.builder", HEADS=HEADS, build_loss=build_any, BACKBONES=BACKBONES)
    _mod("mmdet.models.utils", build_transformer=lambda cfg: TRANSFORMER.build(cfg), NormedLinear=nn.Linear)
    _mod("mmdet.models.utils.builder", TRANSFORMER=TRANSFORMER)
    _mod("mmdet.models.utils.transformer", inverse_sigmoid=inverse_sigmoid)
    _mod("mmdet.models.dense_heads")
    _mod("mmdet.models.dense_heads.anchor_free_head", AnchorFreeHead=AnchorFreeHead)
    _mod("mmdet.models.dense_heads.base_dense_head", BaseDenseHead=BaseDenseHead)
    _mod("mmdet.models.dense_heads.dense_test_mixins", BBoxTestMixin=BBoxTestMixin)
    _mod("mmdet3d")
    _mod("mmdet3d.core", bbox3d2result=bbox3d2result)
    _mod("mmdet3d.core.bbox")
    _mod("mmdet3d.core.bbox.coders", build_bbox_coder=lambda cfg: BBOX_CODERS.build(cfg))
    _mod("mmdet3d.models")
    _mod("mmdet3d.models.detectors")
    _mod("mmdet3d.models.detectors.mvx_two_stage", MVXTwoStageDetector=MVXTwoStageDetector)
    # reference packages: parents carry __path__ only, their __init__.py (which star-imports datasets, flash-attn
    # wrappers, ...) is never executed; leaf modules are imported normally from the read-only checkout.
    for pkg, rel in (("projects", ".."), ("projects.mmdet3d_plugin", ""), ("projects.mmdet3d_plugin.models", "models"),
                     ("projects.mmdet3d_plugin.models.utils", "models/utils"),
                     ("projects.mmdet3d_plugin.models.dense_heads", "models/dense_heads"),
                     ("projects.mmdet3d_plugin.models.backbones", "models/backbones"),
                     ("projects.mmdet3d_plugin.models.detectors", "models/detectors"),
                     ("projects.mmdet3d_plugin.core", "core"), ("projects.mmdet3d_plugin.core.bbox", "core/bbox"),
                     ("projects.mmdet3d_plugin.core.bbox.coders", "core/bbox/coders")):
        _mod(pkg).__path__ = [os.path.normpath(os.path.join(PLUGIN, rel))]
    _installed = True


class _AnyMeta(type):
    def __getattr__(cls, k):      # enum-like members (AffinityType.CENTER, ...) evaluate to their own name
        if k.startswith("__"):
            raise AttributeError(k)
        return k

    def __iter__(cls):            # constants used as sequences at import time (ORDERED_CUBOID_COL_NAMES, ...)
        return iter(())


class _AnyModule(types.ModuleType):
    """A module whose every attribute exists (a throw-away class): for packages the reference's dataset files import names from
    but whose functions the fixture generator never calls (av2, kornia, refile)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = _AnyMeta(k, (), {})
        setattr(self, k, v)
        return v


_data_installed = False


def install_data_stubs(class_names):
    """Stand-ins for what the reference's pipeline / dataset / sampler files import (mmcv image ops restated from mmcv 1.6.2's
    published semantics; av2 0.2.1 / kornia / refile as name-only shells).  Test-only, build container only."""
    global _data_installed
    install()
    if _data_installed:
        return
    import enum
    import numpy as np

    def imnormalize(img, mean, std, to_rgb=True):      # mmcv.image.photometric.imnormalize: float32 image, float64 scalars via cv2
        img = img.copy().astype(np.float32)
        mean32 = np.float64(mean.reshape(1, -1)).astype(np.float32)
        stdinv32 = (1 / np.float64(std.reshape(1, -1))).astype(np.float32)
        if to_rgb:
            img = img[..., ::-1]
        return ((img - mean32) * stdinv32).astype(np.float32)

    def impad(img, *, shape=None, padding=None, pad_val=0, padding_mode="constant"):   # mmcv.image.geometric.impad (shape mode)
        out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
        out[:img.shape[0], :img.shape[1]] = img
        return out

    class _DC:      # mmcv.parallel.DataContainer
        def __init__(self, data, **kw):
            self.data = data

    _mod("mmcv", imnormalize=imnormalize, impad=impad, imread=None, track_iter_progress=lambda it: it,
         mkdir_or_exist=lambda p: os.makedirs(p, exist_ok=True))
    _mod("mmcv.parallel", DataContainer=_DC)
    _mod("mmcv.utils.registry", Registry=Registry, build_from_cfg=build_from_cfg)
    PIPELINES, DATASETS = Registry("pipeline"), Registry("dataset")
    _mod("mmdet.datasets", DATASETS=DATASETS)
    _mod("mmdet.datasets.builder", PIPELINES=PIPELINES)
    _mod("mmdet3d.datasets")
    _mod("mmdet3d.datasets.builder", PIPELINES=PIPELINES)
    _mod("mmdet3d.datasets.custom_3d", Custom3DDataset=type("Custom3DDataset", (), {}))
    _mod("mmdet3d.core.points", BasePoints=object, get_points_type=lambda *a: None)
    _mod("mmdet3d.core.bbox", LiDARInstance3DBoxes=LiDARBoxes)
    import importlib.abc
    import importlib.machinery

    class _ShellFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):    # any av2.* / kornia.* / refile import -> name-only shell
        def find_spec(self, fullname, path=None, target=None):
            if fullname.split(".")[0] in ("av2", "kornia", "refile"):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            m = _AnyModule(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass
    sys.meta_path.insert(0, _ShellFinder())
    cats = enum.Enum("CompetitionCategories", {c: c for c in class_names}, type=str)
    consts = importlib.import_module("av2.evaluation.detection.constants")
    consts.CompetitionCategories = cats
    for k in ("MAX_NORMALIZED_ASE", "MAX_SCALE_ERROR", "MAX_YAW_RAD_ERROR", "MIN_AP", "MIN_CDS", "NUM_DECIMALS", "EPS"):
        setattr(consts, k, 0.0)
    for pkg, rel in (("projects.mmdet3d_plugin.datasets", "datasets"), ("projects.mmdet3d_plugin.datasets.pipelines", "datasets/pipelines"),
                     ("projects.mmdet3d_plugin.datasets.samplers", "datasets/samplers")):
        _mod(pkg).__path__ = [os.path.normpath(os.path.join(PLUGIN, rel))]
    _data_installed = True


def ref(module):
    """Import `projects.mmdet3d_plugin.<module>` from the reference checkout."""
    install()
    return importlib.import_module("projects.mmdet3d_plugin." + module)


def reference_model_cfg(num_cams=7, num_query=644, num_propagated=256, memory_len=1024, topk_proposals=256):
    """The `model` dict of the reference config (exec'd where it lies; `_base_` is ignored), with the camera /
    query counts optionally shrunk for small fixtures."""
    ns = {}
    exec(compile(open(os.path.join(REF_ROOT, "projects", "configs", "far3d.py")).read(), "far3d.py", "exec"), ns)
    cfg = copy.deepcopy(ns["model"])
    h = cfg["pts_bbox_head"]
    h.update(num_query=num_query, num_propagated=num_propagated, memory_len=memory_len, topk_proposals=topk_proposals)
    for a in h["transformer"]["decoder"]["transformerlayers"]["attn_cfgs"]:
        if a["type"] == "DeformableFeatureAggregationCuda":
            a["num_cams"] = num_cams
    return cfg, ns


def build_reference_detector(cfg):
    """Instantiate the reference Far3D detector (eval mode) from its own classes."""
    ref("models.utils.detr3d_transformer")
    ref("core.bbox.coders.nms_free_coder")
    ref("models.backbones.vovnet")
    ref("models.dense_heads.yolox_head")
    ref("models.dense_heads.farhead")
    det = ref("models.detectors.far3d")
    cfg = copy.deepcopy(cfg)
    cfg.pop("type")
    cfg.pop("train_cfg", None)
    cfg.pop("test_cfg", None)
    cfg["pts_bbox_head"]["transformer"]["decoder"]["transformerlayers"] = ConfigDict(
        cfg["pts_bbox_head"]["transformer"]["decoder"]["transformerlayers"])
    keep = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0   # the (training-only) DDNLoss asks for it in __init__; no GPU here
    try:
        model = det.Far3D(**cfg)
    finally:
        torch.cuda.current_device = keep
    return model.eval()
