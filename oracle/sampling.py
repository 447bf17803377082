"""Oracle for the deformable-sampling rows.  Test infrastructure only.

msda_grid_sample  -- mmcv's own CPU fallback formulation (`multi_scale_deformable_attn_pytorch`):
                     per level F.grid_sample(bilinear, zeros, align_corners=False) at 2*loc-1.
                     The reference has an in-tree statement of the same sampling at
                     models/utils/sparse_blocks.py:234-255.
msda_loops        -- independent scalar restatement of the CUDA kernel semantics
                     (x*W-0.5 pixel convention, 4-corner zero padding); small cases only.
aggregation_ref   -- DeformableFeatureAggregationCuda.forward minus the two Linear layers around it,
                     written exactly along models/utils/detr3d_transformer.py:522-569.
"""
import numpy as np
import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """value (bs,S,H,Dh); spatial_shapes (L,2) [(h,w)]; sampling_locations (bs,Q,H,L,P,2);
    attention_weights (bs,Q,H,L,P) or (bs,Q,H,L*P).  Returns (bs,Q,H*Dh)."""
    bs, S, H, Dh = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    attention_weights = attention_weights.reshape(bs, Q, H, L, P)
    shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
    value = value if value.dtype == torch.float64 else value.float()
    grids = 2 * sampling_locations - 1
    out_lv = []
    for l, (h, w) in enumerate(shapes):
        st = int(level_start_index[l])
        v = value[:, st:st + h * w].flatten(2).transpose(1, 2).reshape(bs * H, Dh, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)  # (bs*H, Q, P, 2)
        out_lv.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * H, 1, Q, L * P)
    out = (torch.stack(out_lv, dim=-2).flatten(-2) * aw).sum(-1).view(bs, H * Dh, Q)
    return out.transpose(1, 2).contiguous()


def msda_loops(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """Scalar numpy restatement (float64 accumulation of float32 products is NOT used: plain f32)."""
    value = np.asarray(value, dtype=np.float32)
    loc = np.asarray(sampling_locations, dtype=np.float32)
    bs, S, H, Dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    aw = np.asarray(attention_weights, dtype=np.float32).reshape(bs, Q, H, L, P)
    out = np.zeros((bs, Q, H, Dh), dtype=np.float32)
    f = np.float32
    for b in range(bs):
        for q in range(Q):
            for h in range(H):
                acc = np.zeros(Dh, dtype=np.float32)
                for l in range(L):
                    Hl, Wl = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                    st = int(level_start_index[l])
                    for p in range(P):
                        h_im = loc[b, q, h, l, p, 1] * f(Hl) - f(0.5)
                        w_im = loc[b, q, h, l, p, 0] * f(Wl) - f(0.5)
                        if not (h_im > -1 and w_im > -1 and h_im < Hl and w_im < Wl):
                            continue
                        h_low, w_low = int(np.floor(h_im)), int(np.floor(w_im))
                        lh, lw = f(h_im - f(h_low)), f(w_im - f(w_low))
                        hh, hw = f(1) - lh, f(1) - lw
                        val = np.zeros(Dh, dtype=np.float32)
                        for (yy, xx, wt) in ((h_low, w_low, hh * hw), (h_low, w_low + 1, hh * lw),
                                             (h_low + 1, w_low, lh * hw), (h_low + 1, w_low + 1, lh * lw)):
                            if 0 <= yy <= Hl - 1 and 0 <= xx <= Wl - 1:
                                val += f(wt) * value[b, st + yy * Wl + xx, h]
                        acc += aw[b, q, h, l, p] * val
                out[b, q, h] = acc
    return out.reshape(bs, Q, H * Dh)


def project_points(key_points, lidar2img, pad_hw):
    """detr3d_transformer.py:547-552.  key_points (B,A,P,3) metres, lidar2img (B,N,4,4) -> (B,N,A,P,2) in [0,1]."""
    pts = torch.cat([key_points, torch.ones_like(key_points[..., :1])], dim=-1)
    p2d = torch.matmul(lidar2img[:, :, None, None], pts[:, None, ..., None]).squeeze(-1)
    p2d = p2d[..., :2] / torch.clamp(p2d[..., 2:3], min=1e-5)
    p2d = torch.stack([p2d[..., 0] / pad_hw[1], p2d[..., 1] / pad_hw[0]], dim=-1)
    return p2d


def aggregation_ref(feat_flatten, ref, offsets, lidar2img, logits, level_hw, level_start, pc_range, pad_hw,
                    num_groups=8):
    """One-sample restatement of feature_sampling + the weight softmax/permute.

    feat_flatten (N,S,C); ref (A,3) in [0,1]; offsets (A,P,3); lidar2img (N,4,4);
    logits (A,N,L*P*G) = weights_fc(feat_pos) BEFORE softmax, flattened exactly like the reference
    (index = (l*P+p)*G+g).  Returns (A,C)."""
    N, S, C = feat_flatten.shape
    A, P = offsets.shape[0], offsets.shape[1]
    L, G = len(level_hw), num_groups
    pc = torch.as_tensor(pc_range, dtype=ref.dtype)
    ref_m = ref * (pc[3:6] - pc[0:3]) + pc[0:3]                                   # get_global_pos :27-29
    key_points = ref_m[None, :, None, :] + offsets[None]                          # :525
    # _get_weights :540-542
    w = logits.reshape(1, A, -1, G).softmax(dim=-2)
    w = w.reshape(1, A, N, -1, G).permute(0, 2, 1, 4, 3).contiguous().flatten(end_dim=1)  # (N,A,G,L*P)
    p2d = project_points(key_points, lidar2img[None], pad_hw).flatten(end_dim=1)  # (N,A,P,2)
    p2d = p2d[:, :, None, None, :, :].repeat(1, 1, G, L, 1, 1)                    # :555
    value = feat_flatten.reshape(N, S, G, C // G)
    shapes = torch.as_tensor([list(x) for x in level_hw], dtype=torch.long)
    starts = torch.as_tensor(list(level_start), dtype=torch.long)
    out = msda_grid_sample(value, shapes, starts, p2d, w)                         # (N,A,C)
    return out.reshape(1, N, A, C).sum(1)[0]


def aggregation_tent(feat_flatten, ref, offsets, lidar2img, logits, level_hw, level_start, pc_range, pad_hw, num_groups=8,
                     max_patch=64):
    """Scalar restatement of the *algorithm* of the v4 HIP kernel (far3d_amd/csrc/sampling.hip `aggregate_v4_kernel`), used
    to prove on the CPU that its tent-weight / token-patch reformulation equals `aggregation_ref`:
    bilinear weight of token (tx,ty) for a sample at (px,py) = max(0,1-|px-tx|)*max(0,1-|py-ty|); per (camera, level) the
    P points are merged into per-token weights over the clipped bounding patch when it has <= max_patch tokens (power-of-two
    width), else every (point, corner) is taken on its own.  Small cases only (python loops)."""
    f = np.float32
    feat = np.asarray(feat_flatten, dtype=np.float32)
    N, S, C = feat.shape
    A, P = offsets.shape[0], offsets.shape[1]
    L, G = len(level_hw), num_groups
    pc = np.asarray(pc_range, dtype=np.float32)
    refm = np.asarray(ref, dtype=np.float32) * (pc[3:6] - pc[0:3]) + pc[0:3]
    kp = refm[:, None, :] + np.asarray(offsets, dtype=np.float32)                       # (A,P,3)
    l2i = np.asarray(lidar2img, dtype=np.float32)
    lg = np.asarray(logits, dtype=np.float32).reshape(A, N * L * P, G)
    w = np.exp(lg - lg.max(axis=1, keepdims=True))
    w = (w / w.sum(axis=1, keepdims=True)).reshape(A, N, L, P, G)
    out = np.zeros((A, C), dtype=np.float32)
    stats = dict(patch=0, wide=0, rows=0)
    for a in range(A):
        for n in range(N):
            m = l2i[n]
            xyz = kp[a] @ m[:3, :3].T + m[:3, 3]
            zc = np.maximum(xyz[:, 2], f(1e-5))
            u = xyz[:, 0] / zc / f(pad_hw[1])
            v = xyz[:, 1] / zc / f(pad_hw[0])
            for l, (Hl, Wl) in enumerate(level_hw):
                px, py = u * f(Wl) - f(0.5), v * f(Hl) - f(0.5)
                fx0, fx1 = max(np.floor(px.min()), 0.0), min(np.floor(px.max()) + 1.0, Wl - 1.0)
                fy0, fy1 = max(np.floor(py.min()), 0.0), min(np.floor(py.max()) + 1.0, Hl - 1.0)
                if not (fx1 >= fx0 and fy1 >= fy0):
                    continue
                x0, y0, sx, sy = int(fx0), int(fy0), int(fx1 - fx0) + 1, int(fy1 - fy0) + 1
                base = level_start[l]
                pw = 1
                while pw < sx:
                    pw *= 2
                acc = np.zeros((G, C // G), dtype=np.float32)
                if pw <= max_patch and sy <= max_patch // pw:
                    stats["patch"] += 1
                    for ty in range(y0, y0 + sy):
                        for tx in range(x0, x0 + sx):
                            tw = np.maximum(1 - np.abs(px - tx), 0) * np.maximum(1 - np.abs(py - ty), 0)   # (P,)
                            if tw.max() > 0:
                                stats["rows"] += 1
                                wg = (tw[:, None] * w[a, n, l]).sum(0)                                      # (G,)
                                acc += wg[:, None] * feat[n, base + ty * Wl + tx].reshape(G, C // G)
                else:
                    stats["wide"] += 1
                    for p in range(P):
                        if not (py[p] > -1 and px[p] > -1 and py[p] < Hl and px[p] < Wl):
                            continue
                        hl, wl = int(np.floor(py[p])), int(np.floor(px[p]))
                        lh, lw = py[p] - hl, px[p] - wl
                        for (yy, xx, bw) in ((hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw),
                                             (hl + 1, wl, lh * (1 - lw)), (hl + 1, wl + 1, lh * lw)):
                            if 0 <= yy <= Hl - 1 and 0 <= xx <= Wl - 1 and bw != 0:
                                stats["rows"] += 1
                                acc += (bw * w[a, n, l, p])[:, None] * feat[n, base + yy * Wl + xx].reshape(G, C // G)
                out[a] += acc.reshape(C)
    return torch.from_numpy(out), stats
