"""Tensor-level wrappers over the C-ABI: checks + allocation + raw pointers, nothing else.

Every function launches on torch's *current* HIP stream of the tensors' device (so torch.cuda
events / graphs see the kernels) and never synchronises.
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib

DT_F32, DT_BF16 = 0, 1


def _dt(t):
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError("far3d_amd: unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)


def _chk(t, name, dtype=None, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.Far3dHipError("%s must live on a HIP device (got %s); there is no CPU path" % (name, t.device))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims (got %s)" % (name, ndim, tuple(t.shape)))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _host_i32(vals):
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.int32))
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _host_f32(vals):
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.float32))
    return a, a.ctypes.data_as(ctypes.c_void_p)


def msda_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step=64):
    """mmcv `ms_deform_attn_forward` contract (reference call site detr3d_transformer.py:561-563).

    value (bs,S,H,Dh) f32|bf16; spatial_shapes (L,2) i64; level_start_index (L,) i64;
    sampling_locations (bs,Q,H,L,P,2) f32; attention_weights (bs,Q,H,L,P) or (bs,Q,H,L*P) f32.
    Returns (bs,Q,H*Dh) f32.
    """
    lib = _lib.require_device()
    _chk(value, "value", ndim=4)
    _chk(spatial_shapes, "value_spatial_shapes", torch.int64, 2)
    _chk(level_start_index, "value_level_start_index", torch.int64, 1)
    _chk(sampling_locations, "sampling_locations", torch.float32, 6)
    _chk(attention_weights, "attention_weights", torch.float32)
    bs, S, H, Dh = value.shape
    _, Q, H2, L, P, two = sampling_locations.shape
    if H2 != H or two != 2 or sampling_locations.shape[0] != bs:
        raise ValueError("sampling_locations shape %s inconsistent with value %s" %
                         (tuple(sampling_locations.shape), tuple(value.shape)))
    if attention_weights.numel() != bs * Q * H * L * P:
        raise ValueError("attention_weights shape %s != (bs,Q,H,L*P)" % (tuple(attention_weights.shape),))
    if spatial_shapes.shape[0] != L or level_start_index.shape[0] != L:
        raise ValueError("spatial_shapes/level_start_index must have L=%d rows" % L)
    step = min(bs, im2col_step)
    if step > 0 and bs % step != 0:  # mmcv asserts the same
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (bs, step))
    out = torch.empty((bs, Q, H * Dh), dtype=torch.float32, device=value.device)
    _lib.check(lib.far3d_msda_forward(_ptr(value), _dt(value), _ptr(spatial_shapes), _ptr(level_start_index),
                                      _ptr(sampling_locations), _ptr(attention_weights), _ptr(out),
                                      bs, S, H, Dh, L, Q, P, _stream(value)), "far3d_msda_forward")
    return out


def aggregate_forward(feat, ref, offsets, lidar2img, U, Vc, level_hw, level_start, pc_range, pad_hw,
                      num_groups=8, out=None):
    """Fused perspective-aware aggregation for ONE sample (B=1).

    feat (N,S,256) f32|bf16 token-major value maps; ref (A,3) f32 normalised reference points;
    offsets (A,P,3) f32 = learnable_fc(x); lidar2img (N,4,4) f32; U (A,L*P*G) f32 query part of the
    attention logits; Vc (N,L*P*G) f32 camera part (incl. bias); level_hw [(H,W)]*L; level_start [L];
    pc_range 6 floats; pad_hw (H,W) of the padded image.  Returns (A,256) f32 = sum over cameras of
    MSDA(feat_n, project_n(ref+offsets), softmax_{n,l,p}(U+Vc)).
    """
    lib = _lib.require_device()
    _chk(feat, "feat", ndim=3)
    _chk(ref, "ref", torch.float32, 2)
    _chk(offsets, "offsets", torch.float32)
    _chk(lidar2img, "lidar2img", torch.float32, 3)
    _chk(U, "U", torch.float32, 2)
    _chk(Vc, "Vc", torch.float32, 2)
    N, S, C = feat.shape
    A = ref.shape[0]
    L = len(level_hw)
    G = num_groups
    P = offsets.numel() // max(A * 3, 1) if A > 0 else U.shape[1] // (L * G)
    if U.shape != (A, L * P * G) or Vc.shape != (N, L * P * G) or lidar2img.shape != (N, 4, 4):
        raise ValueError("aggregate_forward: inconsistent shapes U%s Vc%s l2i%s (A=%d N=%d L=%d P=%d G=%d)" %
                         (tuple(U.shape), tuple(Vc.shape), tuple(lidar2img.shape), A, N, L, P, G))
    if out is None:
        out = torch.empty((A, C), dtype=torch.float32, device=feat.device)
    hw_keep, hw_p = _host_i32([list(x) for x in level_hw])
    st_keep, st_p = _host_i32(list(level_start))
    pc_keep, pc_p = _host_f32(list(pc_range))
    _lib.check(lib.far3d_aggregate_forward(_ptr(feat), _dt(feat), _ptr(ref), _ptr(offsets), _ptr(lidar2img),
                                           _ptr(U), _ptr(Vc), _ptr(out), A, N, S, C, G, P, L, hw_p, st_p, pc_p,
                                           float(pad_hw[0]), float(pad_hw[1]), _stream(feat)),
               "far3d_aggregate_forward")
    return out
