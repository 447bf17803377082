"""CPU fp32 restatement of the reference Far3D per-frame inference path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional, plain torch, driven by a state dict that
uses the reference's own parameter names, so a released checkpoint -- or the reference model built by
oracle/refload.py -- can be fed in unchanged.  Pinned against the reference's files by tools/gen_golden.py
(fixtures in tests/golden/).  Citations are file:line under projects/mmdet3d_plugin/ of the reference.
"""
import math

import torch
import torch.nn.functional as F

from . import sampling

VOV99 = dict(stem=(64, 64, 128), stage_conv_ch=(128, 160, 192, 224), stage_out_ch=(256, 512, 768, 1024),
             layer_per_block=5, block_per_stage=(1, 3, 9, 3))          # models/backbones/vovnet.py:79-87
VOV_TINY = dict(stem=(32, 32, 64), stage_conv_ch=(32, 32, 64, 64), stage_out_ch=(64, 128, 192, 256),
                layer_per_block=2, block_per_stage=(1, 1, 2, 1))       # build-owned small spec for fixtures (C1 scale)


def default_cfg(**over):
    cfg = dict(
        backbone=VOV99, fpn_in=(512, 768, 1024), embed_dims=256, num_classes=26, strides=(8, 16, 32, 64),
        num_cams=7, num_query=644, num_propagated=256, memory_len=1024, topk_proposals=256,
        num_layers=6, num_heads=8, num_groups=8, num_levels=4, num_pts=13, ffn_dim=1024,
        pc_range=[-152.4, -152.4, -5.0, 152.4, 152.4, 5.0], code_size=8, max_num=300,
        depth_bins=50, depth_min=0.1, depth_max=110.0, score_thr=0.1,
        proposal_topk=None,   # None: reference behaviour (score > thr, data-dependent M); int K: build-defined static mode,
    )                         # the K best peaks per camera, kept in the reference's flat-index order
    cfg.update(over)
    return cfg


def inverse_sigmoid(x, eps=1e-5):  # mmdet.models.utils.transformer.inverse_sigmoid
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def pos2posemb(pos, num_pos_feats, temperature=10000):
    """One coordinate -> interleaved sin/cos (models/utils/positional_encoding.py:13-36)."""
    dim_t = torch.arange(num_pos_feats, dtype=pos.dtype if pos.dtype == torch.float64 else torch.float32, device=pos.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    p = (pos * (2 * math.pi))[..., None] / dim_t
    return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)


def pos2posemb3d(pos):   # order (y, x, z): positional_encoding.py:24
    return torch.cat([pos2posemb(pos[..., 1], 128), pos2posemb(pos[..., 0], 128), pos2posemb(pos[..., 2], 128)], dim=-1)


def nerf_encoding(t, n=6):   # positional_encoding.py:38-80 (log sampling, no input)
    out = []
    for f in 2.0 ** torch.linspace(0.0, n - 1, n, dtype=t.dtype, device=t.device):
        out += [torch.sin(t * f), torch.cos(t * f)]
    return torch.cat(out, dim=-1)


class Far3DOracle:
    def __init__(self, state_dict, cfg=None, dtype=torch.float32):
        """dtype=torch.float64 runs the same restatement in double precision: the yardstick for how much of a deviation is the
        fp32 rounding noise of the reference arithmetic itself (tests/test_engine_full_gpu.py); inputs must then be double too."""
        self.sd = {k: v.detach().to(dtype) if v.is_floating_point() else v for k, v in state_dict.items()}
        self.dtype = dtype
        self.cfg = cfg or default_cfg()
        self.reset_memory()
        self.prev_scene = None

    def P(self, k):
        return self.sd[k]

    # ------------------------------------------------------------------ a2: VoVNet (vovnet.py:188-273,349-360)
    def _cbr(self, x, prefix, stride=1, pad=1):
        w = self.P(prefix + "/conv.weight")
        x = F.conv2d(x, w, None, stride=stride, padding=pad)
        n = prefix + "/norm."
        x = F.batch_norm(x, self.P(n + "running_mean"), self.P(n + "running_var"), self.P(n + "weight"), self.P(n + "bias"),
                         False, 0.0, 1e-5)
        return F.relu(x)

    def _osa(self, x, prefix, name, identity):
        spec = self.cfg["backbone"]
        outs = [x]
        h = x
        for i in range(spec["layer_per_block"]):
            h = self._cbr(h, "%s.layers.%d.%s_%d" % (prefix, i, name, i))
            outs.append(h)
        xt = self._cbr(torch.cat(outs, dim=1), "%s.concat.%s_concat" % (prefix, name), pad=0)
        g = F.conv2d(xt.mean(dim=(2, 3), keepdim=True), self.P(prefix + ".ese.fc.weight"), self.P(prefix + ".ese.fc.bias"))
        xt = xt * (F.relu6(g + 3.0) / 6.0)
        return xt + x if identity else xt

    def backbone(self, img):
        spec = self.cfg["backbone"]
        x = self._cbr(img, "img_backbone.stem.stem_1", stride=2)
        x = self._cbr(x, "img_backbone.stem.stem_2")
        x = self._cbr(x, "img_backbone.stem.stem_3", stride=2)
        outs = []
        for si in range(4):
            k = si + 2
            if k != 2:
                x = F.max_pool2d(x, 3, 2, ceil_mode=True)
            for b in range(spec["block_per_stage"][si]):
                name = "OSA%d_%d" % (k, b + 1)
                x = self._osa(x, "img_backbone.stage%d.%s" % (k, name), name, identity=b > 0)
            outs.append(x)
        return outs

    # ------------------------------------------------------------------ a3: mmdet FPN (cfg configs/far3d.py:50-57)
    def fpn(self, feats):
        ins = feats[1:]   # start_level=1
        lat = [F.conv2d(x, self.P("img_neck.lateral_convs.%d.conv.weight" % i), self.P("img_neck.lateral_convs.%d.conv.bias" % i))
               for i, x in enumerate(ins)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [F.conv2d(lat[i], self.P("img_neck.fpn_convs.%d.conv.weight" % i), self.P("img_neck.fpn_convs.%d.conv.bias" % i), padding=1)
                for i in range(len(lat))]
        n = len(lat)
        outs.append(F.conv2d(outs[-1], self.P("img_neck.fpn_convs.%d.conv.weight" % n), self.P("img_neck.fpn_convs.%d.conv.bias" % n),
                             stride=2, padding=1))   # add_extra_convs='on_output': first extra conv sees outs[-1] with no ReLU
        return outs

    # ------------------------------------------------------------------ a4: YOLOX head + depth (yolox_head.py:241-341)
    def _tower(self, x, prefix):
        for i in range(2):
            x = F.conv2d(x, self.P("%s.%d.conv.weight" % (prefix, i)), None, padding=1)
            n = "%s.%d.bn." % (prefix, i)
            x = F.batch_norm(x, self.P(n + "running_mean"), self.P(n + "running_var"), self.P(n + "weight"), self.P(n + "bias"),
                             False, 0.0, 1e-3)
            x = x * torch.sigmoid(x)
        return x

    def roi_head(self, feats):
        r = "img_roi_head."
        cls, reg, obj = [], [], []
        for l, x in enumerate(feats):
            cf = self._tower(x, r + "multi_level_cls_convs.%d" % l)
            rf = self._tower(x, r + "multi_level_reg_convs.%d" % l)
            cls.append(F.conv2d(cf, self.P(r + "multi_level_conv_cls.%d.weight" % l), self.P(r + "multi_level_conv_cls.%d.bias" % l)))
            reg.append(F.conv2d(rf, self.P(r + "multi_level_conv_reg.%d.weight" % l), self.P(r + "multi_level_conv_reg.%d.bias" % l)))
            obj.append(F.conv2d(rf, self.P(r + "multi_level_conv_obj.%d.weight" % l), self.P(r + "multi_level_conv_obj.%d.bias" % l)))
        d = feats[0]   # reg_depth_level='p3' (configs/far3d.py:67; depth_predictor.py:80-81)
        for i in range(2):
            d = F.conv2d(d, self.P(r + "depthnet.depth_head.%d.0.weight" % i), self.P(r + "depthnet.depth_head.%d.0.bias" % i), padding=1)
            d = F.relu(F.group_norm(d, 32, self.P(r + "depthnet.depth_head.%d.1.weight" % i), self.P(r + "depthnet.depth_head.%d.1.bias" % i)))
        logit = F.conv2d(d, self.P(r + "depthnet.depth_classifier.weight"), self.P(r + "depthnet.depth_classifier.bias"))
        return dict(enc_cls_scores=cls, enc_bbox_preds=reg, objectnesses=obj, depth_logit=logit, pred_depth=logit.softmax(dim=1))

    # ------------------------------------------------------------------ a5: 2D proposals (yolox_head.py:355-501)
    def get_bboxes(self, outs, forced_valid=None):
        """forced_valid: optional (BN,S,1) bool mask that replaces the peak selection (test rigs use it to resolve K-th-place
        near-ties of the build-defined static top-K mode the same way as the device did; the weights are returned for the check),
        or a tuple (mask, peak mask (BN,S,1) bool): the cells of `mask` outside `peak mask` weigh zero (zero-weight fillers)."""
        cls, reg, obj = outs["enc_cls_scores"], outs["enc_bbox_preds"], outs["objectnesses"]
        n_img = cls[0].shape[0]
        priors, weights, raws = [], [], []
        for l, c in enumerate(cls):
            h, w = c.shape[2:]
            s = float(self.cfg["strides"][l])
            xs = torch.arange(w, dtype=self.dtype) * s
            ys = torch.arange(h, dtype=self.dtype) * s
            priors.append(torch.stack([xs.repeat(h), ys.view(-1, 1).repeat(1, w).view(-1),
                                       torch.full((h * w,), s, dtype=self.dtype), torch.full((h * w,), s, dtype=self.dtype)], dim=-1))
            sw = obj[l].sigmoid() * c.topk(1, dim=1).values.sigmoid()
            nms = F.max_pool2d(sw, (3, 3), stride=1, padding=1).permute(0, 2, 3, 1).reshape(n_img, -1, 1)
            sw = sw.permute(0, 2, 3, 1).reshape(n_img, -1, 1)
            raws.append(sw)
            weights.append(sw * (sw == nms).to(sw.dtype))
        weight = torch.cat(weights, dim=1)                    # (BN, S, 1)
        K = self.cfg["proposal_topk"]
        if isinstance(forced_valid, tuple):
            # (selection mask, peak mask): the rig also hands over the device's outcome of the 3x3 peak test for the selected cells
            # (an equality test on near-equal scores: a cell the device counts as a zero-weight filler must weigh zero here too)
            forced_valid, peak = forced_valid
            weight = torch.cat(raws, dim=1) * peak.to(self.dtype)
        if forced_valid is not None:
            valid = forced_valid
        elif K is None:
            valid = weight > self.cfg["score_thr"]
        else:
            idx = weight[..., 0].topk(K, dim=1).indices
            valid = torch.zeros_like(weight, dtype=torch.bool)
            valid.scatter_(1, idx[..., None], True)
        preds = torch.cat([r.permute(0, 2, 3, 1).reshape(n_img, -1, 4) for r in reg], dim=1)
        pri = torch.cat(priors)
        xy = preds[..., :2] * pri[:, 2:] + pri[:, :2]
        wh = preds[..., 2:].exp() * pri[:, 2:]
        boxes = torch.cat([xy - wh / 2, xy + wh / 2], dim=-1)  # xyxy
        bbox_list = []
        for i in range(n_img):
            b = boxes[i][valid[i].repeat(1, 4)].reshape(-1, 4)
            bbox_list.append(torch.cat([(b[:, :2] + b[:, 2:]) / 2, b[:, 2:] - b[:, :2]], dim=-1))   # cxcywh
        return dict(bbox_list=bbox_list, bbox2d_scores=weight[valid].reshape(-1, 1), valid_indices=valid, peak_weight=weight,
                    raw_weight=torch.cat(raws, dim=1))

    # ------------------------------------------------------------------ FarHead helpers
    def _lin(self, x, name):
        return F.linear(x, self.P(name + ".weight"), self.P(name + ".bias"))

    def _mln(self, x, c, name):   # models/utils/misc.py:182-190 (use_ln handled by caller)
        h = F.relu(self._lin(c, name + ".reduce.0"))
        return self._lin(h, name + ".gamma") * x + self._lin(h, name + ".beta")

    def _mlp2(self, x, name):
        return self._lin(F.relu(self._lin(x, name + ".0")), name + ".2")

    def reset_memory(self):
        self.mem = None

    def _bin_to_depth(self, idx):   # farhead.py:521-527
        c = self.cfg
        bin_size = 2 * (c["depth_max"] - c["depth_min"]) / (c["depth_bins"] * (1 + c["depth_bins"]))
        return c["depth_min"] + bin_size / 8 * (torch.square(idx / 0.5 + 1) - 1)

    def _pre_update_memory(self, data, x):   # farhead.py:453-477
        h = "pts_bbox_head."
        c = self.cfg
        pc = self.P(h + "pc_range")
        if self.mem is None:
            self.mem = dict(emb=x.new_zeros(1, c["memory_len"], c["embed_dims"]), ref=x.new_zeros(1, c["memory_len"], 3),
                            ts=x.new_zeros(1, c["memory_len"], 1), pose=x.new_zeros(1, c["memory_len"], 4, 4),
                            velo=x.new_zeros(1, c["memory_len"], 2))
        else:
            m, L = self.mem, c["memory_len"]
            m["ts"] = m["ts"] + data["timestamp"].unsqueeze(-1).unsqueeze(-1)
            m["pose"] = data["ego_pose_inv"].unsqueeze(1) @ m["pose"]
            m["ref"] = self._transform_ref(m["ref"], data["ego_pose_inv"])
            for k in ("ts", "ref", "emb", "pose", "velo"):
                m[k] = m[k][:, :L] * x.view(-1, *([1] * (m[k].dim() - 1)))
        np_ = c["num_propagated"]
        if np_ > 0:
            pseudo = self.P(h + "pseudo_reference_points.weight") * (pc[3:6] - pc[0:3]) + pc[0:3]
            m = self.mem
            m["ref"] = torch.cat([m["ref"][:, :np_] + (1 - x).view(1, 1, 1) * pseudo, m["ref"][:, np_:]], dim=1)
            m["pose"] = torch.cat([m["pose"][:, :np_] + (1 - x).view(1, 1, 1, 1) * torch.eye(4, dtype=self.dtype), m["pose"][:, np_:]], dim=1)

    @staticmethod
    def _transform_ref(ref, pose):   # misc.py:193-202
        r = torch.cat([ref, torch.ones_like(ref[..., :1])], dim=-1)
        return (pose.unsqueeze(1) @ r.unsqueeze(-1)).squeeze(-1)[..., :3]

    def _proposals(self, outs_roi, feat_flatten, data, pad_hw, forced_depth=None):   # farhead.py:571-610,710-827 (topk=1 path)
        h = "pts_bbox_head."
        pc = self.P(h + "pc_range")
        pred_depth = outs_roi["pred_depth"]
        depth_idx = torch.argmax(pred_depth.permute(0, 2, 3, 1), dim=-1, keepdim=True)   # (BN,H,W,1)
        if forced_depth is not None:    # test rigs: resolve near-ties of the depth-bin argmax the way the device did
            depth_idx = forced_depth(pred_depth, depth_idx)
        valid = outs_roi["valid_indices"]
        C = feat_flatten.shape[-1]
        ctx = feat_flatten[valid.repeat(1, 1, C)].reshape(-1, C)
        bbox_list, scores = outs_roi["bbox_list"], outs_roi["bbox2d_scores"]
        nums = [len(b) for b in bbox_list]
        if sum(nums) == 0:
            return None, None
        boxes = torch.cat(bbox_list, dim=0).to(self.dtype)
        ds = int(pad_hw[0] / pred_depth.shape[2])
        hmax, wmax = pred_depth.shape[2:]
        depths = []
        for i, b in enumerate(bbox_list):
            if nums[i] == 0:
                continue
            dm = depth_idx[i].flatten(0, 1)                      # (HW,1)
            c2 = (b[:, :2] / ds).round().long()
            c2[c2 < 0] = 0
            c2[:, 0][c2[:, 0] >= wmax] = wmax - 1
            c2[:, 1][c2[:, 1] >= hmax] = hmax - 1
            flat = c2[:, 1] * (pad_hw[1] / ds) + c2[:, 0]
            depths.append(torch.gather(dm, 0, flat.long().unsqueeze(1)))
        depths = torch.cat(depths, dim=0)                        # (M,1) bin indices (topk = 1 -> argmax bin)
        thr = torch.tensor([0.1], dtype=self.dtype)
        # build-defined static top-K mode may pad a camera with zero-weight cells: keep their log-odds finite.  A peak that
        # passed the reference's `> 0.1` test is never touched by this clamp, so threshold mode is the reference as is.
        scores = scores.clamp(min=1e-6)
        log_odds = torch.log(scores / (1 - scores)) - torch.log(thr / (1 - thr))
        ctx = torch.cat([ctx, log_odds], dim=-1)                 # (M, C+1)
        d = self._bin_to_depth(depths)
        coords = torch.cat([boxes[:, :2], d], dim=1)
        coords = torch.cat([coords, torch.ones_like(coords[..., :1])], dim=-1)
        coords[..., :2] = coords[..., :2] * torch.maximum(coords[..., 2:3], torch.ones_like(coords[..., 2:3]) * 1e-5)
        i2l = data["lidar2img"].inverse().view(-1, 1, 4, 4)
        i2l = torch.cat([i2l[k].repeat(n, 1, 1) for k, n in enumerate(nums)], dim=0)
        c3 = torch.matmul(i2l, coords.unsqueeze(-1)).squeeze(-1)[..., :3]
        c3 = (c3 - pc[0:3]) / (pc[3:6] - pc[0:3])
        return c3.unsqueeze(0), ctx.unsqueeze(0)

    def _temporal_alignment(self, query_pos, tgt, ref):   # farhead.py:284-313
        h = "pts_bbox_head."
        c, m = self.cfg, self.mem
        pc = self.P(h + "pc_range")
        temp_ref = (m["ref"] - pc[:3]) / (pc[3:6] - pc[0:3])
        temp_pos = self._mlp2(pos2posemb3d(temp_ref), h + "query_embedding")
        temp_mem = m["emb"]
        A = query_pos.size(1)
        eye = torch.eye(4, dtype=self.dtype).unsqueeze(0).unsqueeze(0).repeat(1, A, 1, 1)
        rec_motion = nerf_encoding(torch.cat([torch.zeros_like(ref[..., :3]), eye[..., :3, :].flatten(-2)], dim=-1))
        tgt = self._mln(F.layer_norm(tgt, (256,)), rec_motion, h + "ego_pose_memory")
        query_pos = self._mln(F.layer_norm(query_pos, (256,)), rec_motion, h + "ego_pose_pe")
        mem_motion = nerf_encoding(torch.cat([m["velo"], m["ts"], m["pose"][..., :3, :].flatten(-2)], dim=-1).to(self.dtype))
        temp_pos = self._mln(F.layer_norm(temp_pos, (256,)), mem_motion, h + "ego_pose_pe")
        temp_mem = self._mln(F.layer_norm(temp_mem, (256,)), mem_motion, h + "ego_pose_memory")
        te = lambda t: F.layer_norm(self._lin(t, h + "time_embedding.0"), (256,), self.P(h + "time_embedding.1.weight"),
                                    self.P(h + "time_embedding.1.bias"))
        query_pos = query_pos + te(pos2posemb(torch.zeros_like(ref[..., 0]), 256))
        temp_pos = temp_pos + te(pos2posemb(m["ts"][..., 0], 256).to(self.dtype))
        np_ = c["num_propagated"]
        if np_ > 0:
            tgt = torch.cat([tgt, temp_mem[:, :np_]], dim=1)
            query_pos = torch.cat([query_pos, temp_pos[:, :np_]], dim=1)
            ref = torch.cat([ref, temp_ref[:, :np_]], dim=1)
            eye = torch.eye(4, dtype=self.dtype).unsqueeze(0).unsqueeze(0).repeat(1, A + np_, 1, 1)
            temp_mem, temp_pos = temp_mem[:, np_:], temp_pos[:, np_:]
        return tgt, query_pos, ref, temp_mem, temp_pos, eye

    # ------------------------------------------------------------------ a7/a8: decoder (detr3d_transformer.py:311-422,522-569)
    def _self_attn(self, x, qpos, mem, mempos, lp):
        a = lp + "attentions.0.attn."
        key = torch.cat([x, mem], dim=1)
        kpos = torch.cat([qpos, mempos], dim=1)
        q = (x + qpos).transpose(0, 1)
        k = (key + kpos).transpose(0, 1)
        v = key.transpose(0, 1)
        out = F.multi_head_attention_forward(q, k, v, 256, self.cfg["num_heads"], self.P(a + "in_proj_weight"), self.P(a + "in_proj_bias"),
                                             None, None, False, 0.0, self.P(a + "out_proj.weight"), self.P(a + "out_proj.bias"),
                                             training=False, need_weights=False)[0]
        return x + out.transpose(0, 1)

    def cross_attn(self, x, qpos, feat_flatten, ref, level_hw, level_start, lidar2img, pad_hw, lp, detail=None):
        c = lp + "attentions.1."
        cfg = self.cfg
        A = x.shape[1]
        offsets = self._lin(x, c + "learnable_fc").reshape(A, cfg["num_pts"], 3)
        l2i = lidar2img[0, :, :3, :].flatten(-2)
        ce = F.relu(self._lin(F.relu(self._lin(l2i, c + "cam_embed.0")), c + "cam_embed.2"))
        ce = F.layer_norm(ce, (256,), self.P(c + "cam_embed.4.weight"), self.P(c + "cam_embed.4.bias"))      # (N,256)
        feat_pos = (x + qpos)[0][:, None, :] + ce[None]                                                     # (A,N,256)
        logits = self._lin(feat_pos, c + "weights_fc")                                                       # (A,N,416)
        agg = sampling.aggregation_ref(feat_flatten, ref[0], offsets, lidar2img[0], logits, level_hw, level_start,
                                       cfg["pc_range"], pad_hw, num_groups=cfg["num_groups"])
        if detail is not None:
            detail.update(offsets=offsets, logits=logits, cam_embed=ce, agg=agg)
        return self._lin(agg[None], c + "output_proj") + x

    def decoder(self, tgt, qpos, feat_flatten, ref, level_hw, level_start, mem, mempos, lidar2img, pad_hw):
        outs, x = [], tgt
        for i in range(self.cfg["num_layers"]):
            lp = "pts_bbox_head.transformer.decoder.layers.%d." % i
            ln = lambda t, j: F.layer_norm(t, (256,), self.P(lp + "norms.%d.weight" % j), self.P(lp + "norms.%d.bias" % j))
            x = ln(self._self_attn(x, qpos, mem, mempos, lp), 0)
            x = ln(self.cross_attn(x, qpos, feat_flatten, ref, level_hw, level_start, lidar2img, pad_hw, lp), 1)
            f = lp + "ffns.0.layers."
            x = ln(x + self._lin(F.relu(self._lin(x, f + "0.0")), f + "1"), 2)
            outs.append(x)
        return torch.stack(outs)

    # ------------------------------------------------------------------ a6 + a10 + a11: FarHead.forward (farhead.py:533-693)
    def head_forward(self, mlvl_feats, outs_roi, data, prev_exists, pad_hw, forced_topk=None, forced_depth=None):
        h = "pts_bbox_head."
        cfg = self.cfg
        self._pre_update_memory(data, prev_exists)
        pc = self.P(h + "pc_range")
        N = mlvl_feats[0].shape[0]
        intr = data["intrinsics"] / 1e3
        extr = data["extrinsics"][..., :3, :]
        mln_in = torch.cat([intr[..., 0, 0:1], intr[..., 1, 1:2], extr.flatten(-2)], dim=-1).flatten(0, 1).unsqueeze(1)   # (N,1,14)
        feats, level_hw = [], []
        for f in mlvl_feats:
            n, C, H, W = f.shape
            feats.append(self._mln(f.reshape(n, C, -1).transpose(1, 2), mln_in, h + "spatial_alignment"))
            level_hw.append((H, W))
        feat_flatten = torch.cat(feats, dim=1)
        level_start = [0]
        for hh, ww in level_hw[:-1]:
            level_start.append(level_start[-1] + hh * ww)
        ref = self.P(h + "reference_points.weight")[None]
        query_pos = self._mlp2(pos2posemb3d(ref), h + "query_embedding")
        ref2d, ctx = self._proposals(outs_roi, feat_flatten, data, pad_hw, forced_depth)
        M = 0
        if ref2d is not None:
            M = ref2d.shape[1]
            query_pos = torch.cat([query_pos, self._mlp2(pos2posemb3d(ref2d), h + "query_embedding")], dim=1)
            ref = torch.cat([ref, ref2d], dim=1)
        tgt = torch.zeros_like(query_pos)
        if ctx is not None:
            tgt[:, -M:, :] = self._mlp2(ctx, h + "context_embed")
        tgt, query_pos, ref, temp_mem, temp_pos, rec_pose = self._temporal_alignment(query_pos, tgt, ref)
        outs_dec = self.decoder(tgt, query_pos, feat_flatten, ref, level_hw, level_start, temp_mem, temp_pos, data["lidar2img"], pad_hw)
        outs_dec = torch.nan_to_num(outs_dec)
        cb, rb = h + "cls_branches.0.", h + "reg_branches.0."   # shared across layers (farhead.py:248-251)
        ln = lambda t, j: F.layer_norm(t, (256,), self.P(cb + "%d.weight" % j), self.P(cb + "%d.bias" % j))
        c1 = F.relu(ln(self._lin(outs_dec, cb + "0"), 1))
        c2 = F.relu(ln(self._lin(c1, cb + "3"), 4))
        all_cls = self._lin(c2, cb + "6")
        r = self._lin(F.relu(self._lin(F.relu(self._lin(outs_dec, rb + "0")), rb + "2")), rb + "4")
        xyz = (r[..., 0:3] + inverse_sigmoid(ref)[None]).sigmoid() * (pc[3:6] - pc[0:3]) + pc[0:3]
        all_box = torch.cat([xyz, r[..., 3:]], dim=-1)
        # post_update_memory (farhead.py:479-508)
        score = all_cls[-1].sigmoid().topk(1, dim=-1).values[..., 0:1]
        _, idx = torch.topk(score, cfg["topk_proposals"], dim=1)
        if forced_topk is not None:     # test rigs: resolve near-ties of the memory selection the way the device did
            idx = forced_topk(score[0, :, 0], idx[0, :, 0]).view(1, -1, 1)
        g = lambda t: torch.gather(t, 1, idx.view(1, -1, *([1] * (t.dim() - 2))).repeat(1, 1, *t.shape[2:]))
        m = self.mem
        m["emb"] = torch.cat([g(outs_dec[-1]), m["emb"]], dim=1)
        m["ts"] = torch.cat([g(torch.zeros_like(score, dtype=torch.float64)), m["ts"]], dim=1)
        m["pose"] = torch.cat([g(rec_pose), m["pose"]], dim=1)
        m["ref"] = torch.cat([g(all_box[-1][..., :3]), m["ref"]], dim=1)
        m["velo"] = torch.cat([g(all_box[-1][..., -2:]), m["velo"]], dim=1)
        m["ref"] = self._transform_ref(m["ref"], data["ego_pose"])
        m["ts"] = m["ts"] - data["timestamp"].unsqueeze(-1).unsqueeze(-1)
        m["pose"] = data["ego_pose"].unsqueeze(1) @ m["pose"]
        return dict(all_cls_scores=all_cls, all_bbox_preds=all_box, outs_dec=outs_dec, feat_flatten=feat_flatten,
                    reference_points=ref, query_pos=query_pos, tgt=tgt, num_adaptive=M, level_hw=level_hw, level_start=level_start)

    # ------------------------------------------------------------------ a12: box decode (nms_free_coder.py:39-112; farhead.py:1224-1245)
    def decode(self, outs):
        cfg = self.cfg
        cls = outs["all_cls_scores"][-1][0].sigmoid()
        box = outs["all_bbox_preds"][-1][0]
        scores, idx = cls.view(-1).topk(min(cfg["max_num"], cls.numel()))
        labels = idx % cfg["num_classes"]
        b = box[torch.div(idx, cfg["num_classes"], rounding_mode="floor")]
        rot = torch.atan2(b[..., 6:7], b[..., 7:8])
        b = torch.cat([b[..., 0:3], b[..., 3:6].exp(), rot], dim=-1)
        rng = torch.tensor(cfg["pc_range"], dtype=self.dtype)
        mask = (b[..., :3] >= rng[:3]).all(1) & (b[..., :3] <= rng[3:]).all(1)
        b, scores, labels = b[mask], scores[mask], labels[mask]
        b = b.clone()
        b[:, 2] = b[:, 2] - b[:, 5] * 0.5
        return dict(boxes_3d=b, scores_3d=scores, labels_3d=labels)

    # ------------------------------------------------------------------ a1: one frame (detectors/far3d.py:64-99,244-277)
    def simple_test(self, data, img_metas, forced_valid=None, forced_topk=None, forced_depth=None, camera_cache=None):
        """forced_*: hooks of the test rigs (tests/test_engine_full_gpu.py) that make the oracle adopt the device's choice where a
        discrete decision sits on a near-tie: forced_valid (2D peak selection), forced_topk (memory top-k), forced_depth
        (callable(pred_depth (BN,D,H,W), depth_idx (BN,H,W,1)) -> depth_idx: the per-cell depth-bin argmax).
        camera_cache: optional (dict, key) of a test rig that runs the SAME frame through several oracles of the same weights and
        dtype (decisions adopted or not): the per-camera stages (backbone, FPN, 2D head: 90 % of a frame's CPU time) depend on the
        images only, not on the streaming state or on any decision, so their outputs are computed once per key and re-used."""
        img = data["img"]
        B, N = img.shape[:2]
        assert B == 1
        pad_hw = img_metas[0]["pad_shape"][0][:2]
        hit = camera_cache[0].get(camera_cache[1]) if camera_cache is not None else None
        if hit is not None:
            feats, outs_roi = list(hit[0]), dict(hit[1])
        else:
            feats = self.fpn(self.backbone(img.reshape(B * N, *img.shape[2:])))
            outs_roi = self.roi_head(feats)
            if camera_cache is not None:
                camera_cache[0][camera_cache[1]] = (list(feats), dict(outs_roi))
        if callable(forced_valid):
            forced_valid = forced_valid(self.get_bboxes(outs_roi))
        outs_roi.update(self.get_bboxes(outs_roi, forced_valid))
        if img_metas[0]["scene_token"] != self.prev_scene:
            self.prev_scene = img_metas[0]["scene_token"]
            prev = img.new_zeros(1)
            self.reset_memory()
        else:
            prev = img.new_ones(1)
        outs = self.head_forward(feats, outs_roi, data, prev, pad_hw, forced_topk, forced_depth)
        outs["roi"] = outs_roi
        outs["feat_levels"] = feats
        outs["result"] = self.decode(outs)
        return outs
