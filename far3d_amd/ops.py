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


# --------------------------------------------------------------------------------------------------
# implicit-GEMM convolution / linear
# --------------------------------------------------------------------------------------------------
ACT = {None: 0, "none": 0, "relu": 1, "swish": 2}


class PackedConv:
    """Weights of one conv / linear layer in the kernel's layout (built once at model-prepare time).

    w (rows=ceil(Cout/128)*128, taps*cin_pad) in the compute dtype (bf16 -> bf16 MFMA, fp32 -> exact fp32
    MFMA), tap-major K; bias (rows,) f32 or None.
    """

    def __init__(self, weight, bias=None, stride=1, pad=0, dtype=torch.bfloat16, device=None):
        if weight.dim() == 2:
            weight = weight[:, :, None, None]
        Cout, Cin, KH, KW = weight.shape
        device = device if device is not None else weight.device
        cin_pad = (Cin + 31) // 32 * 32
        rows = (Cout + 127) // 128 * 128
        w = torch.zeros(rows, KH * KW, cin_pad, dtype=torch.float32, device=weight.device)
        w[:Cout, :, :Cin] = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, KH * KW, Cin)
        self.w = w.reshape(rows, KH * KW * cin_pad).to(dtype).to(device).contiguous()
        self.bias = None
        if bias is not None:
            b = torch.zeros(rows, dtype=torch.float32, device=weight.device)
            b[:Cout] = bias.detach().float()
            self.bias = b.to(device)
        self.Cout, self.Cin, self.KH, self.KW, self.stride, self.pad = Cout, Cin, KH, KW, stride, pad

    def out_hw(self, H, W):
        return ((H + 2 * self.pad - self.KH) // self.stride + 1, (W + 2 * self.pad - self.KW) // self.stride + 1)


def _nhwc_view(t, name):
    """(N,H,W,C) view with unit channel stride and dense rows: returns (ld, img_stride)."""
    if t.dim() != 4 or t.stride(3) != 1 or t.stride(1) != t.shape[2] * t.stride(2):
        raise ValueError("%s must be an NHWC view with unit channel stride and dense rows, got shape %s stride %s" %
                         (name, tuple(t.shape), t.stride()))
    if not t.is_cuda:
        raise _lib.Far3dHipError("%s must live on a HIP device; there is no CPU path" % name)
    return t.stride(2), t.stride(0)


def conv2d_nhwc(x, pc, out=None, act=None, out_dtype=None, res=None, y2=None, y2_scale=None, y2_shift=None, tile=0):
    """x: (N,H,W,Cin) NHWC view (f32|bf16).  pc: PackedConv.  out: optional (N,Ho,Wo,Cout) NHWC view to write
    into (e.g. a channel slice of an OSA concat buffer).  Returns out."""
    lib = _lib.require_device()
    N, H, W, Cin = x.shape
    if Cin != pc.Cin:
        raise ValueError("conv2d_nhwc: input has %d channels, layer expects %d" % (Cin, pc.Cin))
    ldx, xs = _nhwc_view(x, "x")
    Ho, Wo = pc.out_hw(H, W)
    if out is None:
        out = torch.empty((N, Ho, Wo, pc.Cout), dtype=out_dtype or x.dtype, device=x.device)
    if tuple(out.shape) != (N, Ho, Wo, pc.Cout):
        raise ValueError("conv2d_nhwc: out shape %s != %s" % (tuple(out.shape), (N, Ho, Wo, pc.Cout)))
    ldy, ys = _nhwc_view(out, "out")
    rp, rdt, ldr, rs, Hr, Wr = None, 0, 0, 0, Ho, Wo
    if res is not None:
        ldr, rs = _nhwc_view(res, "res")
        if res.shape[0] != N or res.shape[3] != pc.Cout:
            raise ValueError("conv2d_nhwc: residual shape %s incompatible" % (tuple(res.shape),))
        rp, rdt, Hr, Wr = _ptr(res), _dt(res), res.shape[1], res.shape[2]
    y2p, y2dt, ldy2, y2s, sp, hp = None, 0, 0, 0, None, None
    if y2 is not None:
        ldy2, y2s = _nhwc_view(y2, "y2")
        _chk(y2_scale, "y2_scale", torch.float32)
        _chk(y2_shift, "y2_shift", torch.float32)
        if y2_scale.numel() != N * pc.Cout or y2_shift.numel() != N * pc.Cout or tuple(y2.shape) != tuple(out.shape):
            raise ValueError("conv2d_nhwc: y2/scale/shift shapes inconsistent")
        y2p, y2dt, sp, hp = _ptr(y2), _dt(y2), _ptr(y2_scale), _ptr(y2_shift)
    _lib.check(lib.far3d_conv2d_nhwc(
        _ptr(x), _dt(x), _ptr(pc.w), _dt(pc.w), _ptr(pc.bias) if pc.bias is not None else None, _ptr(out), _dt(out),
        N, H, W, Cin, ldx, xs, Ho, Wo, pc.Cout, ldy, ys, pc.KH, pc.KW, pc.stride, pc.pad, ACT[act],
        rp, rdt, ldr, rs, Hr, Wr, y2p, y2dt, ldy2, y2s, sp, hp, tile, _stream(x)), "far3d_conv2d_nhwc")
    return out


def linear(x, pc, act=None, res=None, out=None, out_dtype=torch.float32, tile=0):
    """y = act(x @ W^T + b) (+ res).  x (M,K) with unit inner stride (row stride free); returns (M,Cout)."""
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("linear: x must be (M,K) with unit inner stride")
    M = x.shape[0]
    xv = x.as_strided((1, 1, M, x.shape[1]), (M * x.stride(0), M * x.stride(0), x.stride(0), 1))
    if out is None:
        out = torch.empty((M, pc.Cout), dtype=out_dtype, device=x.device)
    ov = out.as_strided((1, 1, M, pc.Cout), (M * out.stride(0), M * out.stride(0), out.stride(0), 1))
    rv = None
    if res is not None:
        rv = res.as_strided((1, 1, M, pc.Cout), (M * res.stride(0), M * res.stride(0), res.stride(0), 1))
    conv2d_nhwc(xv, pc, out=ov, act=act, res=rv, tile=tile)
    return out


# --------------------------------------------------------------------------------------------------
# attention / normalisation / pooling
# --------------------------------------------------------------------------------------------------
def attention_forward(q, k, v, num_heads=8, out=None):
    """softmax(q k^T / sqrt(d)) v per head.  q (Aq,E), k/v (Nk,E) f32|bf16 with unit inner stride; out (Aq,E) f32."""
    lib = _lib.require_device()
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
            raise ValueError("attention_forward: %s must be a 2-D device tensor with unit inner stride" % n)
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError("attention_forward: q/k/v dtypes differ")
    Aq, E = q.shape
    Nk = k.shape[0]
    hd = E // num_heads
    if out is None:
        out = torch.empty((Aq, E), dtype=torch.float32, device=q.device)
    _lib.check(lib.far3d_attention_forward(_ptr(q), _ptr(k), _ptr(v), _dt(q), _ptr(out), Aq, Nk, num_heads, hd,
                                           q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                           float(hd) ** -0.5, _stream(q)), "far3d_attention_forward")
    return out


def layernorm(x, gamma, beta, eps=1e-5, act=None, add=None, out=None):
    """Returns LN(x) (and LN(x)+add when `add` is given)."""
    lib = _lib.require_device()
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32:
        raise ValueError("layernorm: x must be (rows,C) f32 with unit inner stride")
    rows, C = x.shape
    y = out if out is not None else torch.empty((rows, C), dtype=torch.float32, device=x.device)
    y2 = torch.empty_like(y) if add is not None else None
    _lib.check(lib.far3d_layernorm(_ptr(x), _ptr(gamma) if gamma is not None else None,
                                   _ptr(beta) if beta is not None else None, _ptr(y), rows, C, x.stride(0), y.stride(0),
                                   float(eps), 1 if act == "relu" else 0,
                                   _ptr(add) if add is not None else None, add.stride(0) if add is not None else 0,
                                   _ptr(y2) if y2 is not None else None, y2.stride(0) if y2 is not None else 0,
                                   _stream(x)), "far3d_layernorm")
    return (y, y2) if add is not None else y


def ese_nhwc(x, fcw, fcb, identity=None, out=None, scratch=None):
    """x * hsigmoid(fc(mean_hw x)) (+ identity) on NHWC views (any channel slice / pixel stride)."""
    lib = _lib.require_device()
    N, H, W, C = x.shape
    ldx, xs = _nhwc_view(x, "x")
    if out is None:
        out = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    ldy, ys = _nhwc_view(out, "out")
    ldi, isd, ip = 0, 0, None
    if identity is not None:
        ldi, isd = _nhwc_view(identity, "identity")
        ip = _ptr(identity)
    if scratch is None:
        scratch = torch.empty(N * C * 3, dtype=torch.float32, device=x.device)
    _lib.check(lib.far3d_ese_nhwc(_ptr(x), _dt(x), _ptr(fcw), _ptr(fcb), ip, _ptr(out), _ptr(scratch), N, H * W, C,
                                  ldx, xs, ldi, isd, ldy, ys, _stream(x)), "far3d_ese_nhwc")
    return out


def groupnorm_nhwc(x, gamma, beta, groups=32, eps=1e-5, relu=True, out=None, scratch=None):
    lib = _lib.require_device()
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    if scratch is None:
        scratch = torch.empty(N * C * 2 + N * groups * 2, dtype=torch.float32, device=x.device)
    _lib.check(lib.far3d_groupnorm_nhwc(_ptr(x), _dt(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(scratch), N, H * W, C,
                                        groups, float(eps), 1 if relu else 0, _stream(x)), "far3d_groupnorm_nhwc")
    return out


def maxpool3x3s2_nhwc(x, out=None):
    lib = _lib.require_device()
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    Ho, Wo = -(-(H - 3) // 2) + 1, -(-(W - 3) // 2) + 1
    if (Ho - 1) * 2 >= H:
        Ho -= 1
    if (Wo - 1) * 2 >= W:
        Wo -= 1
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    _lib.check(lib.far3d_maxpool3x3s2_nhwc(_ptr(x), _dt(x), _ptr(out), N, H, W, C, Ho, Wo, _stream(x)),
               "far3d_maxpool3x3s2_nhwc")
    return out
