"""Far3DEngine: one 7-camera frame through the HIP kernels, batch 1, inference only.

Host-side orchestration only: every heavy step is a C-ABI kernel from libfar3d_hip.so (far3d_amd.ops).  torch is used
for device memory, the stream, and a handful of O(1k)-element index/glue ops (sin/cos position codes, top-k, 4x4 pose
products) that SURVEY.md marks negligible; there is NO CPU path -- constructing the engine without a HIP device raises.

Data layout (HBM): all feature maps are NHWC (channels-last) in the activation dtype (bf16 in 'bf16' mode, fp32 in
'fp32' parity mode); an OSA block's input and its five 3x3 outputs live in ONE (N,H,W,Cin+5*Cs) buffer and every conv
writes its channel slice in place, so torch.cat never runs; FPN outputs are written twice by the same conv epilogue:
raw (for the 2D head) and camera-modulated, token-major (N, S, 256), which IS `feat_flatten` of the reference.
"""
import math

import torch
import torch.nn.functional as F

from . import lib as _lib
from . import ops, weights
from .synth import level_shapes, level_starts

PRECISIONS = {
    # activations/weights of the convs, value maps, decoder GEMM weights, attention operands
    "bf16": dict(act=torch.bfloat16, value=torch.bfloat16, dec_w=torch.bfloat16, attn=torch.bfloat16),
    "fp32": dict(act=torch.float32, value=torch.float32, dec_w=torch.float32, attn=torch.float32),
}


def default_cfg(**over):
    cfg = dict(
        backbone="V-99-eSE", embed_dims=256, num_classes=26, strides=(8, 16, 32, 64),
        num_cams=7, num_query=644, num_propagated=256, memory_len=1024, topk_proposals=256,
        num_layers=6, num_heads=8, num_groups=8, num_levels=4, num_pts=13, ffn_dim=1024,
        pc_range=[-152.4, -152.4, -5.0, 152.4, 152.4, 5.0], code_size=8, max_num=300,
        depthnet=dict(num_depth_bins=50, depth_min=0.1, depth_max=110.0, stride=8), score_thr=0.1,
        proposal_topk=None,      # None = reference (score > thr, dynamic M, one host sync); K = static K best per camera
        proposal_cap=512,        # per-camera capacity in threshold mode
    )
    cfg.update(over)
    return cfg


def pos2posemb(pos, num_pos_feats, temperature=10000):
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=pos.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    p = (pos * (2 * math.pi))[..., None] / dim_t
    return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)


def pos2posemb3d(pos):   # reference order (y, x, z): models/utils/positional_encoding.py:13-25
    return torch.cat([pos2posemb(pos[..., 1], 128), pos2posemb(pos[..., 0], 128), pos2posemb(pos[..., 2], 128)], dim=-1)


def nerf_encoding(t, n=6):   # positional_encoding.py:38-80
    out = []
    for k in range(n):
        f = float(2.0 ** k)
        out += [torch.sin(t * f), torch.cos(t * f)]
    return torch.cat(out, dim=-1)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class _Lin:
    """A Linear layer packed for far3d_conv2d_nhwc."""

    def __init__(self, sd, name, dtype, device, bias=True, rows=None):
        w = sd[name + ".weight"]
        b = sd[name + ".bias"] if bias else None
        if rows is not None:
            w = w[rows]
            b = b[rows] if b is not None else None
        self.pc = ops.PackedConv(w, b, dtype=dtype, device=device)

    def __call__(self, x, act=None, res=None, out=None, out_dtype=torch.float32):
        return ops.linear(x, self.pc, act=act, res=res, out=out, out_dtype=out_dtype)


class Far3DEngine:
    def __init__(self, state_dict, cfg=None, device="cuda:0", precision="bf16", parts=("backbone", "neck", "roi", "head")):
        _lib.require_device()
        self.parts = tuple(parts)
        self.cfg = cfg or default_cfg()
        self.dev = torch.device(device)
        self.prec = PRECISIONS[precision]
        self.precision = precision
        self.sd = {}
        for k, v in state_dict.items():
            ck = weights.canonical_key(k)
            if ck is not None:
                self.sd[ck] = v.detach().float()
        self.spec = weights.VOV_SPECS[self.cfg["backbone"]]
        self._bufs = {}
        self._cam_ids = {}
        self._in = None             # static input buffers (graph replay reads them)
        self._graph = None
        self._graph_outs = None
        self.use_graph = False      # set True to capture the steady-state frame into one hipGraph (static proposal mode only)
        # Optional forked HIP streams for independent small branches (FAR3D_MS=roi,dec,fork).  Off by default: inside a
        # captured hipGraph the parallel branches measured SLOWER on ROCm 7.2 (8.14 -> 8.8-9.0 ms per frame), see DESIGN.md.
        import os as _os
        self.ms_parts = set(_os.environ.get("FAR3D_MS", "none").split(","))
        self.multi_stream = True
        self._side = None
        self._side_dec = None
        self.fuse_ese = _os.environ.get("FAR3D_FUSE_ESE", "0") == "1"   # measured slower (8.13 -> 8.53 ms): off
        self.eye4 = None
        self.kernel_events = None   # set to {} to collect (start, stop) HIP event pairs around selected kernels
        self.after_fpn = None       # hook(stage_dict) called right after the FPN (camera-sharded mode starts its gather)
        self.reset_memory()
        self.prev_scene = None
        self._prepare()

    # ------------------------------------------------------------------------------------------ weights
    def _conv_bn(self, prefix, conv="/conv", norm="/norm", eps=1e-5, stride=1, pad=1):
        sd = self.sd
        n = prefix + norm + "."
        w, b = weights.fold_bn(sd[prefix + conv + ".weight"], sd[n + "weight"], sd[n + "bias"], sd[n + "running_mean"],
                               sd[n + "running_var"], eps)
        return w, b, stride, pad

    def _pack(self, w, b, stride=1, pad=0):
        return ops.PackedConv(w, b, stride=stride, pad=pad, dtype=self.prec["act"], device=self.dev)

    def _prepare(self):
        sd, cfg, dev = self.sd, self.cfg, self.dev
        f32 = lambda t: t.to(dev).float().contiguous()
        if "backbone" in self.parts:
            self._prepare_backbone()
        if "neck" in self.parts:
            self._prepare_neck_roi()
        if "head" in self.parts:
            self._prepare_head()
        torch.cuda.synchronize(dev)

    def _prepare_backbone(self):
        sd, cfg, dev = self.sd, self.cfg, self.dev
        f32 = lambda t: t.to(dev).float().contiguous()
        # ---- backbone (a2)
        bb = {}
        w, b, _, _ = self._conv_bn("img_backbone.stem.stem_1")
        bb["stem1"] = self._pack(F.pad(w.permute(0, 2, 3, 1).reshape(w.shape[0], 27), (0, 5)), b)   # 1x1 over the 32-ch im2col
        bb["stem2"] = self._pack(*self._conv_bn("img_backbone.stem.stem_2"))
        bb["stem3"] = self._pack(*self._conv_bn("img_backbone.stem.stem_3", stride=2))
        stages = []
        for si in range(4):
            k = si + 2
            blocks = []
            for bi in range(self.spec["block_per_stage"][si]):
                name = "OSA%d_%d" % (k, bi + 1)
                p = "img_backbone.stage%d.%s" % (k, name)
                convs = [self._pack(*self._conv_bn("%s.layers.%d.%s_%d" % (p, i, name, i))) for i in range(self.spec["layer_per_block"])]
                cw, cb, _, _ = self._conv_bn("%s.concat.%s_concat" % (p, name), pad=0)
                blocks.append(dict(convs=convs, concat=self._pack(cw, cb), fcw=f32(sd[p + ".ese.fc.weight"].flatten(1)),
                                   fcb=f32(sd[p + ".ese.fc.bias"])))
            stages.append(blocks)
        bb["stages"] = stages
        self.bb = bb

    def _prepare_neck_roi(self):
        sd, cfg, dev = self.sd, self.cfg, self.dev
        f32 = lambda t: t.to(dev).float().contiguous()
        # ---- FPN (a3)
        nl = len(self.spec["stage_out_ch"]) - 1
        self.fpn_lat = [self._pack(sd["img_neck.lateral_convs.%d.conv.weight" % i], sd["img_neck.lateral_convs.%d.conv.bias" % i]) for i in range(nl)]
        self.fpn_out = [self._pack(sd["img_neck.fpn_convs.%d.conv.weight" % i], sd["img_neck.fpn_convs.%d.conv.bias" % i],
                                   stride=2 if i == nl else 1, pad=1) for i in range(nl + 1)]
        # ---- 2D head + depth (a4)
        r = "img_roi_head."
        self.roi = []
        for l in range(len(cfg["strides"])):
            lv = {}
            for t in ("cls", "reg"):
                lv[t] = [self._pack(*self._conv_bn(r + "multi_level_%s_convs.%d.%d" % (t, l, i), conv=".conv", norm=".bn", eps=1e-3)) for i in range(2)]
            # the two towers' first convs read the same map: one 256 -> 512 conv (cls channels first), half the launches and
            # twice the workgroups on the small levels
            wc, bc, _, _ = self._conv_bn(r + "multi_level_cls_convs.%d.0" % l, conv=".conv", norm=".bn", eps=1e-3)
            wr, br, _, _ = self._conv_bn(r + "multi_level_reg_convs.%d.0" % l, conv=".conv", norm=".bn", eps=1e-3)
            lv["tower0"] = self._pack(torch.cat([wc, wr]), torch.cat([bc, br]), 1, 1)
            lv["cls_head"] = self._pack(sd[r + "multi_level_conv_cls.%d.weight" % l], sd[r + "multi_level_conv_cls.%d.bias" % l])
            lv["reg_head"] = self._pack(torch.cat([sd[r + "multi_level_conv_reg.%d.weight" % l], sd[r + "multi_level_conv_obj.%d.weight" % l]]),
                                        torch.cat([sd[r + "multi_level_conv_reg.%d.bias" % l], sd[r + "multi_level_conv_obj.%d.bias" % l]]))
            self.roi.append(lv)
        self.depth = dict(
            convs=[self._pack(sd[r + "depthnet.depth_head.%d.0.weight" % i], sd[r + "depthnet.depth_head.%d.0.bias" % i], pad=1) for i in range(2)],
            gn=[(f32(sd[r + "depthnet.depth_head.%d.1.weight" % i]), f32(sd[r + "depthnet.depth_head.%d.1.bias" % i])) for i in range(2)],
            cls=self._pack(sd[r + "depthnet.depth_classifier.weight"], sd[r + "depthnet.depth_classifier.bias"]))

    def _prepare_head(self):
        sd, cfg, dev = self.sd, self.cfg, self.dev
        f32 = lambda t: t.to(dev).float().contiguous()
        # ---- FarHead (a6, a10)
        h = "pts_bbox_head."
        dw = self.prec["dec_w"]
        L = lambda name, **kw: _Lin(sd, h + name, dw, dev, **kw)
        self.pc_range = f32(sd[h + "pc_range"])
        self.pc_lo, self.pc_span = self.pc_range[:3], self.pc_range[3:6] - self.pc_range[:3]
        self.sa = dict(reduce=L("spatial_alignment.reduce.0"), gamma=L("spatial_alignment.gamma"), beta=L("spatial_alignment.beta"))
        self.qe = (L("query_embedding.0"), L("query_embedding.2"))
        self.ce = (L("context_embed.0"), L("context_embed.2"))
        self.te = L("time_embedding.0")
        self.te_ln = (f32(sd[h + "time_embedding.1.weight"]), f32(sd[h + "time_embedding.1.bias"]))
        self.mln = {n: dict(reduce=L(n + ".reduce.0"), gamma=L(n + ".gamma"), beta=L(n + ".beta")) for n in ("ego_pose_pe", "ego_pose_memory")}
        self.cls_b = (L("cls_branches.0.0"), L("cls_branches.0.3"), L("cls_branches.0.6"))
        self.cls_ln = [(f32(sd[h + "cls_branches.0.%d.weight" % j]), f32(sd[h + "cls_branches.0.%d.bias" % j])) for j in (1, 4)]
        self.reg_b = (L("reg_branches.0.0"), L("reg_branches.0.2"), L("reg_branches.0.4"))
        self.ref_fixed = f32(sd[h + "reference_points.weight"])
        self.pseudo_ref = f32(sd[h + "pseudo_reference_points.weight"]) if cfg["num_propagated"] > 0 else None
        # ---- decoder layers (a7, a8)
        E = cfg["embed_dims"]
        self.layers = []
        for i in range(cfg["num_layers"]):
            lp = h + "transformer.decoder.layers.%d." % i
            a, c = lp + "attentions.0.attn.", lp + "attentions.1."
            ipw, ipb = sd[a + "in_proj_weight"], sd[a + "in_proj_bias"]
            pk = lambda w, b: ops.PackedConv(w, b, dtype=dw, device=dev)
            ly = dict(
                qk=pk(ipw[:2 * E], ipb[:2 * E]), k=pk(ipw[E:2 * E], ipb[E:2 * E]), v=pk(ipw[2 * E:], ipb[2 * E:]),
                out=pk(sd[a + "out_proj.weight"], sd[a + "out_proj.bias"]),
                wfc=pk(sd[c + "weights_fc.weight"], None), wfc_full=pk(sd[c + "weights_fc.weight"], sd[c + "weights_fc.bias"]),
                lfc=pk(sd[c + "learnable_fc.weight"], sd[c + "learnable_fc.bias"]),
                oproj=pk(sd[c + "output_proj.weight"], sd[c + "output_proj.bias"]),
                ce0=pk(sd[c + "cam_embed.0.weight"], sd[c + "cam_embed.0.bias"]), ce2=pk(sd[c + "cam_embed.2.weight"], sd[c + "cam_embed.2.bias"]),
                ce_ln=(f32(sd[c + "cam_embed.4.weight"]), f32(sd[c + "cam_embed.4.bias"])),
                ffn1=pk(sd[lp + "ffns.0.layers.0.0.weight"], sd[lp + "ffns.0.layers.0.0.bias"]),
                ffn2=pk(sd[lp + "ffns.0.layers.1.weight"], sd[lp + "ffns.0.layers.1.bias"]),
                norms=[(f32(sd[lp + "norms.%d.weight" % j]), f32(sd[lp + "norms.%d.bias" % j])) for j in range(3)])
            self.layers.append(ly)
        cl = []
        for i in range(cfg["num_layers"]):
            c = h + "transformer.decoder.layers.%d.attentions.1." % i
            cl.append((sd[c + "cam_embed.0.weight"], sd[c + "cam_embed.0.bias"], sd[c + "cam_embed.2.weight"], sd[c + "cam_embed.2.bias"],
                       sd[c + "cam_embed.4.weight"], sd[c + "cam_embed.4.bias"], sd[c + "weights_fc.weight"], sd[c + "weights_fc.bias"]))
        self.cam_chain = ops.pack_cam_embed_chain(cl, dev)
        # ---- frame-invariant pieces of temporal_alignment (farhead.py:284-303): the current frame's ego motion is the
        # identity, so its MLN(180) codes are constants, and the 644 learned queries never change.
        mk_dim_t = lambda n: (10000 ** (2 * torch.div(torch.arange(n, dtype=torch.float32, device=dev), 2, rounding_mode="floor") / n)).contiguous()
        self.dim_t128, self.dim_t256 = mk_dim_t(128), mk_dim_t(256)
        eye = torch.eye(4, device=dev)[:3, :].flatten()
        rec = nerf_encoding(torch.cat([torch.zeros(3, device=dev), eye])[None])              # (1,180)
        self.rec_code = {}
        for n in ("ego_pose_pe", "ego_pose_memory"):
            hh = self.mln[n]["reduce"](rec, act="relu")
            self.rec_code[n] = (self.mln[n]["gamma"](hh), self.mln[n]["beta"](hh))             # (1,256) each
        self.time0 = ops.layernorm(self.te(pos2posemb(torch.zeros(1, device=dev), 256)), *self.te_ln)   # (1,256)
        qp = self._query_pos(self.ref_fixed)
        self.qpos_fixed = ops.row_affine_ln(qp, *self.rec_code["ego_pose_pe"], add=self.time0)
        self.tgt_fixed = ops.row_affine_ln(torch.zeros_like(qp), *self.rec_code["ego_pose_memory"])

    def _query_pos(self, ref):
        return self.qe[1](self.qe[0](ops.posemb3d(ref.contiguous(), self.dim_t128), act="relu"))

    def _fork(self, fns):
        """Run independent branches concurrently: fns[0] on the current stream, the rest on side streams that fork from it
        and join back (inside a hipGraph capture this becomes parallel graph branches).  Returns the branches' results."""
        if not self.multi_stream or len(fns) == 1 or "fork" not in self.ms_parts:
            return [f() for f in fns]
        cur = torch.cuda.current_stream(self.dev)
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.dev) for _ in range(8)]
        outs = [None] * len(fns)
        used = []
        for i, f in enumerate(fns[1:]):
            st = self._side[i % len(self._side)]
            if st not in used:
                st.wait_stream(cur)
                used.append(st)
            with torch.cuda.stream(st):
                outs[i + 1] = f()
        outs[0] = fns[0]()
        for st in used:
            cur.wait_stream(st)

        def keep(o):
            if isinstance(o, torch.Tensor):
                o.record_stream(cur)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    keep(v)
        for o in outs[1:]:
            keep(o)
        return outs

    def _buf(self, key, shape, dtype):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = torch.empty(shape, dtype=dtype, device=self.dev)
            self._bufs[key] = b
        return b

    # ------------------------------------------------------------------------------------------ a2: backbone
    def backbone(self, img):
        """img (N,3,H,W) f32 NCHW on device -> [stage2..stage5] dense NHWC maps."""
        act, spec = self.prec["act"], self.spec
        N = img.shape[0]
        Lb = spec["layer_per_block"]
        # eSE workspaces (channel sums + gates, N*C*3 floats per block): one slab, zeroed once per frame
        nblk = sum(len(b) for b in self.bb["stages"])
        slab = self._buf(("ese_slab",), (nblk, N * max(spec["stage_out_ch"]) * (2 * ops.ESE_REPLICAS + 1)), torch.float32)
        slab.zero_()
        ese_i = 0
        x = ops.stem_im2col(img, act)
        x = ops.conv2d_nhwc(x, self.bb["stem1"], act="relu")
        x = ops.conv2d_nhwc(x, self.bb["stem2"], act="relu")
        H, W = self.bb["stem3"].out_hw(x.shape[1], x.shape[2])
        in_ch = spec["stem"][2]
        outs = []
        stage_in = None   # dense input of the stage (stem3 output is written straight into the first concat buffer)
        for si, blocks in enumerate(self.bb["stages"]):
            sc, oc = spec["stage_conv_ch"][si], spec["stage_out_ch"][si]
            if si > 0:
                Hp, Wp = H, W
                H, W = -(-(Hp - 3) // 2) + 1, -(-(Wp - 3) // 2) + 1
                if (H - 1) * 2 >= Hp:
                    H -= 1
                if (W - 1) * 2 >= Wp:
                    W -= 1
            cat = self._buf(("cat", si, 0), (N, H, W, in_ch + Lb * sc), act)
            if si == 0:
                ops.conv2d_nhwc(x, self.bb["stem3"], out=cat[..., :in_ch], act="relu")
            else:
                ops.maxpool3x3s2_nhwc(stage_in, out=cat[..., :in_ch])
            cur_in = in_ch
            for bi, blk in enumerate(blocks):
                last = bi == len(blocks) - 1
                src = cat[..., :cur_in]
                for i, pc in enumerate(blk["convs"]):
                    dst = cat[..., cur_in + i * sc: cur_in + (i + 1) * sc]
                    ops.conv2d_nhwc(src, pc, out=dst, act="relu")
                    src = dst
                # bf16 path: the concat conv's epilogue also accumulates the eSE average pool (per-XCD replicas in the slab)
                fused = self.fuse_ese and act == torch.bfloat16 and ops.conv_tile(cat, blk["concat"]) >= 50
                xt = ops.conv2d_nhwc(cat, blk["concat"], out=self._buf(("xt", si), (N, H, W, oc), act), act="relu",
                                     chan_sum=slab[ese_i] if fused else None)
                if last:
                    out = self._buf(("stage", si), (N, H, W, oc), act)
                    nxt = None
                else:
                    nxt = self._buf(("cat", si, 1 + (bi % 2)), (N, H, W, oc + Lb * sc), act)
                    out = nxt[..., :oc]
                ops.ese_nhwc(xt, blk["fcw"], blk["fcb"], identity=cat[..., :cur_in] if bi > 0 else None, out=out,
                             scratch=slab[ese_i], sums_state="ready" if fused else "zeroed")
                ese_i += 1
                if not last:
                    cat, cur_in = nxt, oc
            stage_in = self._bufs[("stage", si)]
            outs.append(stage_in)
            in_ch = oc
        return outs

    # ------------------------------------------------------------------------------------------ a3 + MLN: FPN
    def fpn(self, feats, mln_scale, mln_shift):
        """Returns (raw levels [NHWC act], feat_flatten (N,S,256) value dtype, level_hw, level_start)."""
        act, val = self.prec["act"], self.prec["value"]
        ins = feats[1:]
        N = ins[0].shape[0]
        n = len(ins)
        hw = [(f.shape[1], f.shape[2]) for f in ins]
        hw.append(self.fpn_out[n].out_hw(*hw[-1]))
        starts, S = level_starts(hw)
        tokens = self._buf(("tokens",), (N, S, 256), val)
        lat = [None] * n
        for i in range(n - 1, -1, -1):   # top-down: laterals[i-1] += nearest_upsample(laterals[i])
            lat[i] = ops.conv2d_nhwc(ins[i], self.fpn_lat[i], out=self._buf(("lat", i), (N, hw[i][0], hw[i][1], 256), act),
                                     res=lat[i + 1] if i + 1 < n else None)
        raw = []
        for i in range(n + 1):
            src = lat[i] if i < n else raw[n - 1]
            y2 = tokens[:, starts[i]: starts[i] + hw[i][0] * hw[i][1]].view(N, hw[i][0], hw[i][1], 256)
            raw.append(ops.conv2d_nhwc(src, self.fpn_out[i], out=self._buf(("fpn", i), (N, hw[i][0], hw[i][1], 256), act),
                                       y2=y2, y2_scale=mln_scale, y2_shift=mln_shift))
        return raw, tokens, hw, starts

    # ------------------------------------------------------------------------------------------ a4: 2D head + depth
    def roi_head(self, raw):
        """YOLOX towers + depth head.  The four pyramid levels and the depth branch are independent: the small levels run
        on side streams underneath the stride-8 level, which alone fills the GPU."""
        def level(l):
            def run():
                x, lv = raw[l], self.roi[l]
                t0 = ops.conv2d_nhwc(x, lv["tower0"], act="swish")                 # (N,h,w,512): cls | reg
                half = t0.shape[-1] // 2
                cf = ops.conv2d_nhwc(t0[..., :half], lv["cls"][1], act="swish")
                c = ops.conv2d_nhwc(cf, lv["cls_head"], out_dtype=torch.float32)
                rf = ops.conv2d_nhwc(t0[..., half:], lv["reg"][1], act="swish")
                r = ops.conv2d_nhwc(rf, lv["reg_head"], out_dtype=torch.float32)
                return c, r
            return run

        def depth():
            d = raw[0]
            for i in range(2):
                d = ops.conv2d_nhwc(d, self.depth["convs"][i])
                d = ops.groupnorm_nhwc(d, *self.depth["gn"][i], groups=32, relu=True)
            return ops.conv2d_nhwc(d, self.depth["cls"], out_dtype=torch.float32)

        br = [level(0)] + [level(l) for l in range(1, len(raw))] + [depth]
        res = self._fork(br) if "roi" in self.ms_parts else [f() for f in br]
        cls, reg = [r[0] for r in res[:-1]], [r[1] for r in res[:-1]]
        return cls, reg, res[-1]

    # ------------------------------------------------------------------------------------------ memory (a6/a11, tiny)
    def reset_memory(self):
        self.mem = None

    def _mem_init(self):
        """Persistent streaming-memory buffers (fixed shapes so that a captured hipGraph can read/write them in place).
        The reference grows the queue to 1280 and truncates to memory_len at the next frame (farhead.py:467-471,501-505);
        truncating right after the update is the same thing."""
        cfg, dev = self.cfg, self.dev
        Lm, E = cfg["memory_len"], cfg["embed_dims"]
        self.mem = dict(emb=torch.zeros(1, Lm, E, device=dev), ref=torch.zeros(1, Lm, 3, device=dev),
                        ts=torch.zeros(1, Lm, 1, device=dev, dtype=torch.float64), pose=torch.zeros(1, Lm, 4, 4, device=dev),
                        velo=torch.zeros(1, Lm, 2, device=dev))

    def _pre_update_memory(self, data, prev_exists, fresh):   # farhead.py:453-477
        cfg, dev = self.cfg, self.dev
        P_ = cfg["num_propagated"]
        x = prev_exists
        s = self.mem
        if fresh:
            m = {k: v.clone() for k, v in s.items()}
        else:
            m = dict(ts=s["ts"] + data["timestamp"].unsqueeze(-1).unsqueeze(-1),
                     pose=data["ego_pose_inv"].unsqueeze(1) @ s["pose"],
                     ref=self._transform_ref(s["ref"], data["ego_pose_inv"]), emb=s["emb"], velo=s["velo"])
            m = {k: v * x.view(-1, *([1] * (v.dim() - 1))).to(v.dtype) for k, v in m.items()}
        if P_ > 0:
            pseudo = self.pseudo_ref * self.pc_span + self.pc_lo
            m["ref"] = torch.cat([m["ref"][:, :P_] + (1 - x).view(1, 1, 1) * pseudo, m["ref"][:, P_:]], dim=1)
            m["pose"] = torch.cat([m["pose"][:, :P_] + (1 - x).view(1, 1, 1, 1) * self.eye4, m["pose"][:, P_:]], dim=1)
        return m

    @staticmethod
    def _transform_ref(ref, pose):
        r = torch.cat([ref, torch.ones_like(ref[..., :1])], dim=-1)
        return (pose.unsqueeze(1) @ r.unsqueeze(-1)).squeeze(-1)[..., :3]

    def _mln_rows(self, x, code, name, add=None):
        m = self.mln[name]
        hh = m["reduce"](code, act="relu")
        return ops.row_affine_ln(x, m["gamma"](hh), m["beta"](hh), add=add)

    # ------------------------------------------------------------------------------------------ a7/a8: decoder
    def decoder(self, tgt, qpos, tokens, ref, hw, starts, mem, mempos, lidar2img, pad_hw):
        cfg = self.cfg
        A, E = tgt.shape
        Km = mem.shape[0]
        at = self.prec["attn"]
        fast = at == torch.bfloat16     # bf16 mode: GEMM operands are handed over as bf16 copies (LDS-DMA GEMM path)
        x = tgt
        xqb, xb = ops.add_cast(x, qpos, at, at if fast else None)
        memkb, memb = ops.add_cast(mem, mempos, at, at if fast else None)
        if not fast:
            xb, memb = x, mem
        l2i = lidar2img[:, :3, :].flatten(1).contiguous()             # (N,12)
        outs = torch.empty((cfg["num_layers"], A, E), dtype=torch.float32, device=self.dev)
        # reference points are fixed across the 6 layers: one camera-sorted workgroup order per frame (scheduling only)
        perm = ops.aggregation_order(ref, lidar2img, cfg["pc_range"], pad_hw)
        ln_kw = dict(add=qpos, add_dtype=at, bf16_copy=fast)
        L = len(self.layers)
        qks = [self._buf(("qk", li), (A + Km, 2 * E), at) for li in range(L)]
        vbs = [self._buf(("v", li), (A + Km, E), at) for li in range(L)]

        # Work that does not depend on the evolving queries -- the memory rows' K/V projections and the camera-embedding
        # chain (lidar2img -> cam_embed -> camera part of the attention logits) of every layer -- is issued up front on a
        # side stream, layer 0 first; each layer waits on its own event, so all of it hides under the main chain.
        Vcs = [None] * L
        side_ev = [None] * L
        vc_all = ops.cam_embed_chain(l2i, self.cam_chain)             # all layers' camera terms: one launch (fp32 weights)

        def side_layer(li):
            ly = self.layers[li]
            ops.linear(memkb, ly["k"], out=qks[li][A:, E:])
            ops.linear(memb, ly["v"], out=vbs[li][A:])
            Vcs[li] = vc_all[li]                                        # (N,416) camera part + bias

        cur = torch.cuda.current_stream(self.dev)
        ms_dec = self.multi_stream and "dec" in self.ms_parts
        if ms_dec:
            if self._side_dec is None:
                self._side_dec = torch.cuda.Stream(device=self.dev)
            sst = self._side_dec
            sst.wait_stream(cur)
            with torch.cuda.stream(sst):
                for li in range(L):
                    side_layer(li)
                    side_ev[li] = torch.cuda.Event()
                    side_ev[li].record(sst)
                    Vcs[li].record_stream(cur)
        for li, ly in enumerate(self.layers):
            qk, vb = qks[li], vbs[li]
            if not ms_dec:
                side_layer(li)
            # self-attention: q = x+pos, k = cat[x,mem]+cat[pos,mempos], v = cat[x,mem] (detr3d_transformer.py:378-396)
            self._fork([lambda: ops.linear(xqb, ly["qk"], out=qk[:A]), lambda: ops.linear(xb, ly["v"], out=vb[:A])])
            if side_ev[li] is not None:
                cur.wait_event(side_ev[li])
            att = ops.attention_forward(qk[:A, :E], qk[:, E:], vb, num_heads=cfg["num_heads"], out_dtype=at)
            y = ops.linear(att, ly["out"], res=x)
            r = ops.layernorm(y, *ly["norms"][0], **ln_kw)
            x, xqb = r[0], r[1]
            xb = r[2] if fast else x
            # cross-attention: fused perspective-aware aggregation (detr3d_transformer.py:522-569)
            U, offs = self._fork([lambda: ops.linear(xqb, ly["wfc"]),   # (A,416) query part of the logits
                                  lambda: ops.linear(xb, ly["lfc"])])   # (A,39) learnable 3D offsets
            Vc = Vcs[li]
            agg = ops.aggregate_forward(tokens, ref, offs, lidar2img, U, Vc, hw, starts, cfg["pc_range"], pad_hw,
                                        num_groups=cfg["num_groups"], perm=perm, out_dtype=at)
            self.last_agg = (tokens, ref, offs, lidar2img, U, Vc, hw, starts, pad_hw, perm)   # for isolated kernel timing
            y = ops.linear(agg, ly["oproj"], res=x)
            if fast:
                x, xb = ops.layernorm(y, *ly["norms"][1], bf16_copy=True)
            else:
                x = ops.layernorm(y, *ly["norms"][1])
                xb = x
            # FFN: x + W2 relu(W1 x), hidden 1024 (SURVEY.md finding 4)
            hdn = ops.linear(xb, ly["ffn1"], act="relu", out_dtype=at)
            y = ops.linear(hdn, ly["ffn2"], res=x)
            r = ops.layernorm(y, *ly["norms"][2], out=outs[li], **ln_kw)
            x, xqb = r[0], r[1]
            xb = r[2] if fast else x
        if ms_dec:
            cur.wait_stream(self._side_dec)
        return outs

    # ------------------------------------------------------------------------------------------ one frame
    def camera_stage(self, img, dd, cam_ids, pad_hw):
        """Everything that is independent per camera: backbone, FPN (+MLN), 2D head, depth, proposal
        selection and adaptive-query construction.  img (n,3,H,W) on device for the cameras `cam_ids` (global indices)."""
        cfg, dev = self.cfg, self.dev
        n = img.shape[0]
        key = tuple(cam_ids)
        if key not in self._cam_ids:
            self._cam_ids[key] = torch.as_tensor(list(key), device=dev)
        ids = self._cam_ids[key]
        lidar2img = dd["lidar2img"][0].float()[ids].contiguous()
        intr = dd["intrinsics"][0].float()[ids] / 1e3
        extr = dd["extrinsics"][0].float()[ids][:, :3, :]
        c14 = torch.cat([intr[:, 0, 0:1], intr[:, 1, 1:2], extr.flatten(1)], dim=-1).contiguous()   # farhead.py:553-556
        hh = self.sa["reduce"](c14, act="relu")
        mln_scale, mln_shift = self.sa["gamma"](hh), self.sa["beta"](hh)
        ev = self.kernel_events
        if ev is not None:
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
        feats = self.backbone(img)
        if ev is not None:
            b1.record()
            ev.setdefault("backbone", []).append((b0, b1))
        raw, tokens, hw, starts = self.fpn(feats, mln_scale, mln_shift)
        st = dict(tokens=tokens, hw=hw, starts=starts, raw=raw, lidar2img=lidar2img)
        if self.after_fpn is not None:
            self.after_fpn(st)                      # multi-GPU: start the all-gather of the value maps here
        cls, reg, depth_logit = self.roi_head(raw)
        K = cfg["proposal_topk"]
        cap = K if K is not None else min(cfg["proposal_cap"], tokens.shape[1])
        wgt, sel_idx, sel_cnt = ops.proposal_select(cls, reg, cfg["strides"], cap, thr=cfg["score_thr"], topk=K is not None)
        row_off = (torch.cumsum(sel_cnt, 0) - sel_cnt).to(torch.int32)
        img2lidar = dd["img2lidar"][ids].contiguous()   # inverse(lidar2img), computed when the inputs are staged
        ref2d, ctx, box2d, score2d = ops.proposal_gather(reg, cfg["strides"], sel_idx, sel_cnt, row_off, wgt, depth_logit,
                                                         cfg["depthnet"]["stride"], cfg["depthnet"], img2lidar, tokens,
                                                         cfg["pc_range"], score_thr=0.1)
        st.update(ref2d=ref2d, ctx=ctx, box2d=box2d, score2d=score2d, sel_idx=sel_idx, sel_cnt=sel_cnt, depth_logit=depth_logit)
        return st

    def head_stage(self, tokens, ref2d, ctx, M, dd, img_metas, hw, starts, pad_hw):
        """The cross-camera part: streaming memory, query construction, 6-layer decoder, heads, decode (a6-a12)."""
        cfg, dev = self.cfg, self.dev
        lidar2img = dd["lidar2img"][0].float().contiguous()
        ref2d, ctx = ref2d[:M], ctx[:M]
        # ---- scene change / memory (detectors/far3d.py:252-257)
        fresh = img_metas[0]["scene_token"] != self.prev_scene or self.mem is None
        if fresh:
            self.prev_scene = img_metas[0]["scene_token"]
            self._mem_init()
        # ---- a6: memory pre-update + temporal codes (one kernel), then the adaptive / propagated queries
        P_ = cfg["num_propagated"]
        m, temp_ref, mem_code, tpos = ops.memory_prepare(self.mem, dd["ego_pose_inv"], dd["timestamp"], self.pseudo_ref, self.dim_t256,
                                                         0.0 if fresh else 1.0, fresh, cfg["pc_range"], P_)
        if M > 0:
            qpos_a = ops.row_affine_ln(self._query_pos(ref2d), *self.rec_code["ego_pose_pe"], add=self.time0)
            tgt_a = ops.row_affine_ln(self.ce[1](self.ce[0](ctx, act="relu")), *self.rec_code["ego_pose_memory"])
        t_emb = ops.layernorm(self.te(tpos), *self.te_ln)
        temp_pos = self._mln_rows(self._query_pos(temp_ref), mem_code, "ego_pose_pe", add=t_emb)
        temp_mem = self._mln_rows(m["emb"][0], mem_code, "ego_pose_memory")
        parts_t, parts_q, parts_r = [self.tgt_fixed], [self.qpos_fixed], [self.ref_fixed]
        if M > 0:
            parts_t.append(tgt_a); parts_q.append(qpos_a); parts_r.append(ref2d)
        parts_t.append(temp_mem[:P_]); parts_q.append(temp_pos[:P_]); parts_r.append(temp_ref[:P_])
        tgt, qpos, ref = torch.cat(parts_t).contiguous(), torch.cat(parts_q).contiguous(), torch.cat(parts_r).contiguous()
        A = tgt.shape[0]
        outs_dec = self.decoder(tgt, qpos, tokens, ref, hw, starts, temp_mem[P_:].contiguous(), temp_pos[P_:].contiguous(),
                                lidar2img, pad_hw)
        outs_dec = torch.nan_to_num(outs_dec)
        # ---- a10: shared heads over all 6 layers at once
        flat = outs_dec.view(-1, outs_dec.shape[-1])
        c1 = ops.layernorm(self.cls_b[0](flat), *self.cls_ln[0], act="relu")
        c2 = ops.layernorm(self.cls_b[1](c1), *self.cls_ln[1], act="relu")
        all_cls = self.cls_b[2](c2).view(cfg["num_layers"], 1, A, cfg["num_classes"])
        rr = self.reg_b[2](self.reg_b[1](self.reg_b[0](flat, act="relu"), act="relu"))
        nl = cfg["num_layers"]
        box_flat, sc = ops.head_finalize(rr, ref, all_cls[-1][0], cfg["pc_range"], nl, cfg["num_classes"])
        all_box = box_flat.view(nl, 1, A, cfg["code_size"])
        # ---- a11: memory post-update (farhead.py:479-508): top-k by max-class score, push, truncate, ego warp -- in place
        idx = torch.topk(sc, cfg["topk_proposals"], dim=0).indices
        ops.memory_post_update(m, idx, outs_dec[-1], all_box[-1][0], dd["ego_pose"], dd["timestamp"], self.mem)
        outs = dict(all_cls_scores=all_cls, all_bbox_preds=all_box, outs_dec=outs_dec, num_adaptive=M, feat_flatten=tokens,
                    reference_points=ref)
        outs["result"] = self.decode(all_cls, all_box)
        return outs

    def _stage_inputs(self, data):
        """Copy the frame's inputs into static device buffers (so a captured graph can be replayed on them) and derive
        img2lidar = inverse(lidar2img) (farhead.py:798) -- the only step torch.linalg does for us, outside the graph."""
        dev = self.dev
        img = data["img"]
        if img.dim() == 5:
            assert img.shape[0] == 1, "batch 1 per engine (one scene stream per GPU)"
            img = img[0]
        keys = ("lidar2img", "intrinsics", "extrinsics", "ego_pose", "ego_pose_inv", "timestamp")
        if self._in is None or tuple(self._in["img"].shape) != tuple(img.shape):
            self._in = dict(img=torch.empty(tuple(img.shape), dtype=torch.float32, device=dev))
            for k in keys:
                self._in[k] = torch.empty(tuple(data[k].shape), dtype=torch.float64 if k == "timestamp" else torch.float32, device=dev)
            self._in["img2lidar"] = torch.empty(tuple(data["lidar2img"].shape[1:]), dtype=torch.float32, device=dev)
            self._graph = None
        self._in["img"].copy_(img, non_blocking=True)
        for k in keys:
            self._in[k].copy_(data[k], non_blocking=True)
        self._in["img2lidar"].copy_(torch.linalg.inv(self._in["lidar2img"][0]))
        if self.eye4 is None:
            self.eye4 = torch.eye(4, device=dev)
        return self._in

    def _frame_body(self, dd, img_metas, pad_hw):
        cfg = self.cfg
        img = dd["img"]
        N = img.shape[0]
        st = self.camera_stage(img, dd, range(N), pad_hw)
        K = cfg["proposal_topk"]
        M = N * K if K is not None else int(st["sel_cnt"].sum().item())   # the reference's data-dependent M: one host sync
        outs = self.head_stage(st["tokens"], st["ref2d"], st["ctx"], M, dd, img_metas, st["hw"], st["starts"], pad_hw)
        outs.update(fpn=st["raw"], depth_logit=st["depth_logit"], bbox2d=st["box2d"][:M], bbox2d_scores=st["score2d"][:M],
                    sel_idx=st["sel_idx"], sel_cnt=st["sel_cnt"])
        return outs

    @torch.no_grad()
    def forward_frame(self, data, img_metas):
        """data: the reference's per-frame dict (img (1,N,3,H,W), lidar2img, intrinsics, extrinsics, ego_pose(_inv),
        timestamp); tensors may live on the host (they are uploaded) or already on the device.  With `use_graph` the
        steady-state frame (same scene, static proposal mode) is captured once into a hipGraph and replayed."""
        dd = self._stage_inputs(data)
        pad_hw = tuple(img_metas[0]["pad_shape"][0][:2])
        steady = img_metas[0]["scene_token"] == self.prev_scene and self.mem is not None
        if self.use_graph and steady and self.cfg["proposal_topk"] is not None and self.kernel_events is None:
            if self._graph is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._graph_outs = self._frame_body(dd, img_metas, pad_hw)
                self._graph = g
            self._graph.replay()
            return self._graph_outs
        return self._frame_body(dd, img_metas, pad_hw)

    # ------------------------------------------------------------------------------------------ a12: NMS-free decode
    def decode(self, all_cls, all_box):   # core/bbox/coders/nms_free_coder.py:39-112; farhead.py:1224-1245
        cfg = self.cfg
        cls = all_cls[-1][0].sigmoid()
        box = all_box[-1][0]
        scores, idx = cls.view(-1).topk(min(cfg["max_num"], cls.numel()))
        labels = idx % cfg["num_classes"]
        b = box[torch.div(idx, cfg["num_classes"], rounding_mode="floor")]
        b = torch.cat([b[..., 0:3], b[..., 3:6].exp(), torch.atan2(b[..., 6:7], b[..., 7:8])], dim=-1)
        mask = (b[..., :3] >= self.pc_range[:3]).all(1) & (b[..., :3] <= self.pc_range[3:]).all(1)
        b = torch.cat([b[:, :2], b[:, 2:3] - b[:, 5:6] * 0.5, b[:, 3:]], dim=-1)
        return dict(boxes_3d=b, scores_3d=scores, labels_3d=labels, keep=mask)
