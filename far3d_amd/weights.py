"""State-dict schema of the Far3D detector (the reference's own parameter names, so a released checkpoint
loads unchanged) plus a deterministic, non-degenerate random initialiser for tests / benchmarks.

Key names follow the reference modules: VoVNet `stem_1/conv`, `OSA2_1_0/norm`, ... (models/backbones/vovnet.py:114-139,
188-273 -- note the '/' inside names), mmdet FPN `lateral_convs.N.conv`, YOLOX ConvModule `conv`/`bn`
(models/dense_heads/yolox_head.py:163-232), DepthPredictor (models/depth_predictor/depth_predictor.py:41-60), FarHead
(models/dense_heads/farhead.py:228-282) and the decoder (models/utils/detr3d_transformer.py:270-307,503-512).
cls/reg branches are ONE module shared by the 6 decoder layers (farhead.py:248-251): a checkpoint lists them 6 times
(`cls_branches.0..5`); the schema keeps index 0 as the owner and treats 1..5 as aliases.
"""
import math
from collections import OrderedDict

import torch

VOV_SPECS = {
    "V-99-eSE": dict(stem=(64, 64, 128), stage_conv_ch=(128, 160, 192, 224), stage_out_ch=(256, 512, 768, 1024),
                     layer_per_block=5, block_per_stage=(1, 3, 9, 3)),
    "V-57-eSE": dict(stem=(64, 64, 128), stage_conv_ch=(128, 160, 192, 224), stage_out_ch=(256, 512, 768, 1024),
                     layer_per_block=5, block_per_stage=(1, 1, 4, 3)),
    "V-39-eSE": dict(stem=(64, 64, 128), stage_conv_ch=(128, 160, 192, 224), stage_out_ch=(256, 512, 768, 1024),
                     layer_per_block=5, block_per_stage=(1, 1, 2, 2)),
    "V-19-eSE": dict(stem=(64, 64, 128), stage_conv_ch=(128, 160, 192, 224), stage_out_ch=(256, 512, 768, 1024),
                     layer_per_block=3, block_per_stage=(1, 1, 1, 1)),
    # build-owned miniature with the same topology, for plumbing-scale fixtures (BASELINE configs[0] scale)
    "V-tiny-eSE": dict(stem=(32, 32, 64), stage_conv_ch=(32, 32, 64, 64), stage_out_ch=(64, 128, 192, 256),
                       layer_per_block=2, block_per_stage=(1, 1, 2, 1)),
}


def _bn(spec, name, c):
    for k in ("weight", "bias", "running_mean", "running_var"):
        spec[name + "." + k] = (c,)


def backbone_spec(spec_name, input_ch=3):
    s = VOV_SPECS[spec_name]
    spec = OrderedDict()
    chans = [input_ch] + list(s["stem"])
    for i in range(3):
        n = "img_backbone.stem.stem_%d" % (i + 1)
        spec[n + "/conv.weight"] = (chans[i + 1], chans[i], 3, 3)
        _bn(spec, n + "/norm", chans[i + 1])
    in_ch = s["stem"][2]
    for si in range(4):
        k = si + 2
        sc, oc = s["stage_conv_ch"][si], s["stage_out_ch"][si]
        for b in range(s["block_per_stage"][si]):
            name = "OSA%d_%d" % (k, b + 1)
            p = "img_backbone.stage%d.%s" % (k, name)
            c = in_ch
            for i in range(s["layer_per_block"]):
                spec["%s.layers.%d.%s_%d/conv.weight" % (p, i, name, i)] = (sc, c, 3, 3)
                _bn(spec, "%s.layers.%d.%s_%d/norm" % (p, i, name, i), sc)
                c = sc
            spec["%s.concat.%s_concat/conv.weight" % (p, name)] = (oc, in_ch + s["layer_per_block"] * sc, 1, 1)
            _bn(spec, "%s.concat.%s_concat/norm" % (p, name), oc)
            spec[p + ".ese.fc.weight"] = (oc, oc, 1, 1)
            spec[p + ".ese.fc.bias"] = (oc,)
            in_ch = oc
    return spec


def detector_spec(backbone="V-99-eSE", num_query=644, num_propagated=256, num_classes=26, embed=256, num_layers=6,
                  num_levels=4, num_pts=13, num_groups=8, ffn_dim=1024, code_size=8, depth_bins=50, fpn_levels=4):
    s = VOV_SPECS[backbone]
    spec = backbone_spec(backbone)
    fin = s["stage_out_ch"][1:]
    for i, c in enumerate(fin):
        spec["img_neck.lateral_convs.%d.conv.weight" % i] = (embed, c, 1, 1)
        spec["img_neck.lateral_convs.%d.conv.bias" % i] = (embed,)
    for i in range(fpn_levels):
        spec["img_neck.fpn_convs.%d.conv.weight" % i] = (embed, embed, 3, 3)
        spec["img_neck.fpn_convs.%d.conv.bias" % i] = (embed,)
    h = "pts_bbox_head."
    spec[h + "code_weights"] = (code_size,)
    spec[h + "match_costs"] = (code_size,)
    spec[h + "pc_range"] = (6,)
    for i in range(num_layers):
        lp = h + "transformer.decoder.layers.%d." % i
        spec[lp + "attentions.0.attn.in_proj_weight"] = (3 * embed, embed)
        spec[lp + "attentions.0.attn.in_proj_bias"] = (3 * embed,)
        spec[lp + "attentions.0.attn.out_proj.weight"] = (embed, embed)
        spec[lp + "attentions.0.attn.out_proj.bias"] = (embed,)
        c = lp + "attentions.1."
        for n, (o, k) in (("weights_fc", (num_groups * num_levels * num_pts, embed)), ("output_proj", (embed, embed)),
                          ("learnable_fc", (num_pts * 3, embed)), ("cam_embed.0", (embed // 2, 12)),
                          ("cam_embed.2", (embed, embed // 2))):
            spec[c + n + ".weight"] = (o, k)
            spec[c + n + ".bias"] = (o,)
        spec[c + "cam_embed.4.weight"] = (embed,)
        spec[c + "cam_embed.4.bias"] = (embed,)
        spec[lp + "ffns.0.layers.0.0.weight"] = (ffn_dim, embed)
        spec[lp + "ffns.0.layers.0.0.bias"] = (ffn_dim,)
        spec[lp + "ffns.0.layers.1.weight"] = (embed, ffn_dim)
        spec[lp + "ffns.0.layers.1.bias"] = (embed,)
        for j in range(3):
            spec[lp + "norms.%d.weight" % j] = (embed,)
            spec[lp + "norms.%d.bias" % j] = (embed,)
    for j in (0, 3):
        spec[h + "cls_branches.0.%d.weight" % j] = (embed, embed)
        spec[h + "cls_branches.0.%d.bias" % j] = (embed,)
        spec[h + "cls_branches.0.%d.weight" % (j + 1)] = (embed,)     # LayerNorm
        spec[h + "cls_branches.0.%d.bias" % (j + 1)] = (embed,)
    spec[h + "cls_branches.0.6.weight"] = (num_classes, embed)
    spec[h + "cls_branches.0.6.bias"] = (num_classes,)
    for j in (0, 2):
        spec[h + "reg_branches.0.%d.weight" % j] = (embed, embed)
        spec[h + "reg_branches.0.%d.bias" % j] = (embed,)
    spec[h + "reg_branches.0.4.weight"] = (code_size, embed)
    spec[h + "reg_branches.0.4.bias"] = (code_size,)
    spec[h + "reference_points.weight"] = (num_query, 3)
    if num_propagated > 0:
        spec[h + "pseudo_reference_points.weight"] = (num_propagated, 3)

    def mln(name, cdim):
        spec[h + name + ".reduce.0.weight"] = (embed, cdim)
        spec[h + name + ".reduce.0.bias"] = (embed,)
        for g in ("gamma", "beta"):
            spec[h + name + "." + g + ".weight"] = (embed, embed)
            spec[h + name + "." + g + ".bias"] = (embed,)
    mln("spatial_alignment", 14)
    for n, cin in (("context_embed", embed + 1), ("query_embedding", embed * 3 // 2)):
        spec[h + n + ".0.weight"] = (embed, cin)
        spec[h + n + ".0.bias"] = (embed,)
        spec[h + n + ".2.weight"] = (embed, embed)
        spec[h + n + ".2.bias"] = (embed,)
    spec[h + "time_embedding.0.weight"] = (embed, embed)
    spec[h + "time_embedding.0.bias"] = (embed,)
    spec[h + "time_embedding.1.weight"] = (embed,)
    spec[h + "time_embedding.1.bias"] = (embed,)
    mln("ego_pose_pe", 180)
    mln("ego_pose_memory", 180)
    r = "img_roi_head."
    for l in range(fpn_levels):
        for t in ("cls", "reg"):
            for i in range(2):
                p = r + "multi_level_%s_convs.%d.%d." % (t, l, i)
                spec[p + "conv.weight"] = (embed, embed, 3, 3)
                _bn(spec, p + "bn", embed)
        for n, o in (("cls", num_classes), ("reg", 4), ("obj", 1), ("centers2d", 2)):
            spec[r + "multi_level_conv_%s.%d.weight" % (n, l)] = (o, embed, 1, 1)
            spec[r + "multi_level_conv_%s.%d.bias" % (n, l)] = (o,)
    for i in range(2):
        spec[r + "depthnet.depth_head.%d.0.weight" % i] = (embed, embed, 3, 3)
        spec[r + "depthnet.depth_head.%d.0.bias" % i] = (embed,)
        spec[r + "depthnet.depth_head.%d.1.weight" % i] = (embed,)
        spec[r + "depthnet.depth_head.%d.1.bias" % i] = (embed,)
    spec[r + "depthnet.depth_classifier.weight"] = (depth_bins + 1, embed, 1, 1)
    spec[r + "depthnet.depth_classifier.bias"] = (depth_bins + 1,)
    return spec


SHARED_ALIASES = [("pts_bbox_head.cls_branches.%d." % i, "pts_bbox_head.cls_branches.0.") for i in range(1, 6)] + \
                 [("pts_bbox_head.reg_branches.%d." % i, "pts_bbox_head.reg_branches.0.") for i in range(1, 6)]


def canonical_key(k):
    """Map a checkpoint key to the schema key (shared-branch aliases, BN bookkeeping dropped -> None)."""
    if k.endswith("num_batches_tracked"):
        return None
    for alias, owner in SHARED_ALIASES:
        if k.startswith(alias):
            return owner + k[len(alias):]
    return k


def init_state_dict(spec, seed=0, pc_range=(-152.4, -152.4, -5.0, 152.4, 152.4, 5.0)):
    """Deterministic non-degenerate weights: He-scaled conv/linear weights (activations stay O(1) through the 60-conv
    backbone), BN/LN scales U(0.75,1.25), small biases, `weights_fc` NOT zero (the reference's own init zeroes it,
    detr3d_transformer.py:518, which would hide attention-weight bugs), reference points U(0,1)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shape in spec.items():
        leaf = k.rsplit(".", 1)[-1]
        if k.endswith("pc_range"):
            v = torch.tensor(pc_range, dtype=torch.float32)
        elif k.endswith("code_weights") or k.endswith("match_costs"):
            v = torch.ones(shape)
        elif "reference_points" in k:
            v = torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            v = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            v = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif len(shape) == 1 and leaf == "weight":           # BN / LN / GN scale
            v = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif leaf == "bias" and "learnable_fc" in k:
            v = (torch.rand(shape, generator=g) * 2 - 1) * 2.0   # U(-2, 2) metres, see the weight branch below
        elif leaf == "bias" or leaf == "in_proj_bias":
            v = torch.randn(shape, generator=g) * 0.05
        else:                                                 # conv / linear weight
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 1.0 if ("attn" in k or "weights_fc" in k or "learnable_fc" in k or "branches" in k) else math.sqrt(2.0)
            v = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
            if "learnable_fc" in k:
                # the reference's own initialiser for the key-point offsets (detr3d_transformer.py:505,517-520): nn.Linear's
                # default weight U(-1/sqrt(fan_in), 1/sqrt(fan_in)); the bias (below) U(-bias, bias) with bias = 2 m from the
                # config (projects/configs/far3d.py:117-125).  Round 1 used N(0, 4/sqrt(fan_in)) weights, i.e. ~3x wider clouds.
                v = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
            if k.startswith("img_neck."):
                v = v * 0.5                                   # no ReLU/BN in the FPN: keep its outputs O(1)
            if "multi_level_conv_" in k:
                v = v * (0.1 if "conv_reg" in k else 0.5)     # 2D logits ~N(0,1.5): no sigmoid saturation, finite log-odds
        sd[k] = v
    return sd


def fold_bn(conv_w, bn_w, bn_b, mean, var, eps):
    """conv (no bias) + eval-mode BN  ->  conv weight / bias (vovnet.py norm_eval=True; YOLOX BN eps 1e-3)."""
    scale = bn_w / torch.sqrt(var + eps)
    return conv_w * scale.view(-1, 1, 1, 1), bn_b - mean * scale


# ------------------------------------------------------------------------------------------ checkpoint compatibility (§8 f2)
def normalize_state_dict(ckpt, strict_schema=None):
    """Whatever mmcv's `load_checkpoint(model, path, map_location='cpu')` accepts (ref tools/test.py:208) -> a flat
    {schema key: tensor} dict: unwraps `{'state_dict': ...}` / `{'model': ...}` containers (mmcv runner checkpoints carry
    'meta', 'state_dict', 'optimizer'), strips DDP `module.` prefixes (tools/test.py:229-232 wraps the model before saving in
    training), keeps the '/'-containing VoVNet names (vovnet.py:128-139), resolves the shared cls/reg branch aliases
    (farhead.py:248-251) after checking that the six copies are identical, and drops BN `num_batches_tracked`.
    strict_schema: optional detector_spec(); then missing / unexpected / mis-shaped keys raise with the offending names."""
    sd = ckpt
    for k in ("state_dict", "model"):
        if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
            sd = sd[k]
    if not isinstance(sd, dict) or not all(isinstance(k, str) for k in sd):
        raise TypeError("checkpoint does not contain a state dict")
    out = OrderedDict()
    first_name = {}                      # canonical key -> the checkpoint key whose tensor is stored in `out`
    for k, v in sd.items():
        if not isinstance(v, torch.Tensor):
            continue
        while k.startswith("module."):
            k = k[len("module."):]
        ck = canonical_key(k)
        if ck is None:
            continue
        if ck in out:
            # every copy of a shared branch is compared with the one seen first, whatever the order of the keys (an alias that
            # precedes its owner used to be overwritten without a comparison: ADVICE r2)
            if not torch.equal(out[ck], v):
                raise ValueError("checkpoint keys %s and %s are copies of one shared branch (%s) but disagree" % (first_name[ck], k, ck))
            if ck == k:
                first_name[ck] = k       # the owner's own tensor object is the one kept
                out[ck] = v
            continue
        out[ck] = v
        first_name[ck] = k
    if strict_schema is not None:
        missing = [k for k in strict_schema if k not in out]
        unexpected = [k for k in out if k not in strict_schema]
        bad = [k for k in strict_schema if k in out and tuple(out[k].shape) != tuple(strict_schema[k])]
        if missing or unexpected or bad:
            raise KeyError("checkpoint does not match the Far3D schema: %d missing (e.g. %s), %d unexpected (e.g. %s), %d mis-shaped (e.g. %s)"
                           % (len(missing), missing[:3], len(unexpected), unexpected[:3], len(bad), bad[:3]))
    return out


def load_checkpoint(path, strict_schema=None, map_location="cpu"):
    """torch.load + normalize_state_dict.  Works for the released `iter_82548.pth` layout (mmcv runner checkpoint) and for a
    bare state dict."""
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    except TypeError:
        ckpt = torch.load(path, map_location=map_location)
    return normalize_state_dict(ckpt, strict_schema)
