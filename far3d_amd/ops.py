"""Tensor-level wrappers over the C-ABI: checks + allocation + raw pointers, nothing else.

Every function launches on torch's *current* HIP stream of the tensors' device (so torch.cuda
events / graphs see the kernels) and never synchronises.
"""
import contextlib
import ctypes
import threading

import numpy as np
import torch

from . import lib as _lib

DT_F32, DT_BF16 = 0, 1
DT_F32_BF16X3 = 2       # FAR3D_DT_F32_BF16X3: conv weight code "fp32 data, two-term bf16 split products"
DT_BF16_PAIR = 3        # FAR3D_DT_BF16_PAIR: activations stored split, [32 hi | 32 lo] bf16 per 32-channel block


# ---- pair storage (include/far3d_hip.h FAR3D_DT_BF16_PAIR): a map of C logical channels (C % 32 == 0) is a bf16 tensor with 2C
# channels; every 32-channel block is 32 hi = bf16(x) then 32 lo = bf16(x - hi), so hi + lo carries 16 significant bits.
def pair_from_float(t):
    """(..., C) f32 -> (..., 2C) bf16 pair storage."""
    C = t.shape[-1]
    if C % 32:
        raise ValueError("pair storage needs C %% 32 == 0 (got %d)" % C)
    b = t.float().reshape(t.shape[:-1] + (C // 32, 32))
    hi = b.to(torch.bfloat16)
    lo = (b - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).reshape(t.shape[:-1] + (2 * C,)).contiguous()


def pair_to_float(t):
    """(..., 2C) bf16 pair storage -> (..., C) f32 (exact: hi + lo)."""
    C2 = t.shape[-1]
    b = t.reshape(t.shape[:-1] + (C2 // 64, 2, 32)).float()
    return (b[..., 0, :] + b[..., 1, :]).reshape(t.shape[:-1] + (C2 // 2,))


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


def _hole(hole):
    """hole: None or (count int32 device tensor (1,), start, end): the query rows [start + count, end) hold no query (fixed-capacity
    proposal mode, far3d_hip.h far3d_proposal_gather).  -> (pointer | None, start, end) C arguments."""
    if hole is None:
        return None, 0, 0
    cnt, start, end = hole
    _chk(cnt, "hole count", torch.int32)
    return _ptr(cnt), int(start), int(end)


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


def agg_tables(Vc, out=None):
    """Softmax factors of the camera part of the aggregation logits (far3d_agg_tables): Vc (layers, N, J) or (N, J) f32 ->
    (layers, 2 + N, J) / (2 + N, J) f32 = [max over cameras | sum of exp | exp(Vc - max)].  Per frame, for all layers at once."""
    lib = _lib.require_device()
    _chk(Vc, "Vc", torch.float32)
    v3 = Vc if Vc.dim() == 3 else Vc[None]
    Ls, N, J = v3.shape
    tab = out if out is not None else torch.empty((Ls, 2 + N, J), dtype=torch.float32, device=Vc.device)
    _lib.check(lib.far3d_agg_tables(_ptr(v3), _ptr(tab), Ls, N, J, _stream(Vc)), "far3d_agg_tables")
    return tab if Vc.dim() == 3 else tab[0]


class AggSplit:
    """Workspace of far3d_aggregate_forward variant 9 (sibling workgroups for heavy queries): partial sums [rows][2][256] f32, arrival
    tickets [rows] int32 (zero at rest) and the number of sibling slots far3d_agg_order may hand out."""

    def __init__(self, rows, extra=384, device="cuda:0"):
        self.rows, self.extra = int(rows), int(extra)
        self.partials = torch.empty((self.rows, 2, 256), dtype=torch.float32, device=device)
        self.tickets = torch.zeros((self.rows,), dtype=torch.int32, device=device)


class AggLists:
    """Workspace of far3d_aggregate_forward variant 13 (the two-kernel split, A/B only): FAR3D_AGG_LISTS_FLOATS(A) floats of row lists +
    softmax denominators, 2 * 8 * ceil(A / 8) int32 of per-wave entry counts."""

    def __init__(self, A, device):
        slots = -(-A // 8) * 8
        self.rows = A
        self.buf = torch.zeros((slots * (2 * 1024 * 9 + 8),), dtype=torch.float32, device=device)
        self.counts = torch.zeros((slots * 2,), dtype=torch.int32, device=device)


def aggregate_forward(feat, ref, offsets, lidar2img, U, Vc, level_hw, level_start, pc_range, pad_hw,
                      num_groups=8, out=None, perm=None, out_dtype=torch.float32, variant=0, tables=None, split=None, qbase=None, lists=None):
    """Fused perspective-aware aggregation for ONE sample (B=1).

    feat (N,S,256) f32|bf16 token-major value maps; ref (A,3) f32 normalised reference points;
    offsets (A,P,3) f32 = learnable_fc(x); lidar2img (N,4,4) f32; U (A,L*P*G) f32 query part of the
    attention logits; Vc (N,L*P*G) f32 camera part (incl. bias); level_hw [(H,W)]*L; level_start [L];
    pc_range 6 floats; pad_hw (H,W) of the padded image.  Returns (A,256) f32 = sum over cameras of
    MSDA(feat_n, project_n(ref+offsets), softmax_{n,l,p}(U+Vc)).  U / offsets may be row-strided views (unit inner stride)
    only through .contiguous() -- the kernel reads dense rows.  variant: see include/far3d_hip.h (0 = default).
    tables: agg_tables(Vc) (2+N, L*P*G) -- the per-frame softmax factors the default kernel reads instead of Vc; computed here
    (one more launch) when the default kernel runs and the caller did not pass them.
    qbase: aggregation_order(..., sorted_operands=True)'s (A, 4) table -- SORTED mode of the default kernel: U and offsets hold the
    query of perm entry e at row e (their producers stored through the order's `inv`), ref is not read; same bits as the unsorted call.
    """
    lib = _lib.require_device()
    _chk(feat, "feat", ndim=3)
    _chk(ref, "ref", torch.float32, 2)
    _chk(lidar2img, "lidar2img", torch.float32, 3)
    _chk(Vc, "Vc", torch.float32, 2)
    N, S, C = feat.shape
    A = ref.shape[0]
    L = len(level_hw)
    G = num_groups
    if offsets.dim() == 3:
        offsets = offsets.reshape(A, -1)
    for t, n in ((U, "U"), (offsets, "offsets")):     # row-strided views are fine (column blocks of a merged-GEMM output)
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and (A == 0 or t.stride(1) == 1)):
            raise ValueError("aggregate_forward: %s must be a 2-D f32 device tensor with unit inner stride" % n)
    P = offsets.shape[1] // 3
    if U.shape != (A, L * P * G) or Vc.shape != (N, L * P * G) or lidar2img.shape != (N, 4, 4):
        raise ValueError("aggregate_forward: inconsistent shapes U%s Vc%s l2i%s (A=%d N=%d L=%d P=%d G=%d)" %
                         (tuple(U.shape), tuple(Vc.shape), tuple(lidar2img.shape), A, N, L, P, G))
    if out is None:
        out = torch.empty((A, C), dtype=out_dtype, device=feat.device)
    nperm = None
    if perm is not None:
        _chk(perm, "perm", torch.int32, 1)
        nperm = perm.numel() - (split.extra if split is not None else 0)        # main entries (the sibling entries follow them)
        if nperm > A or nperm < 0:
            raise ValueError("aggregate_forward: perm has more entries than there are rows")
    if split is not None:
        if perm is None or split.rows < A or variant not in (0, 9):
            raise ValueError("aggregate_forward: split needs the perm of aggregation_order(split=...), a workspace of >= A rows and variant 0 / 9")
        variant = 9
    if tables is None and variant in (0, 8, 9, 12, 13) and A > 0 and N <= 8 and P <= 16:
        tables = agg_tables(Vc)
    if qbase is not None:
        _chk(qbase, "qbase", torch.float32, 2)
        if perm is None or split is not None or variant not in (0, 8, 13) or tuple(qbase.shape) != (nperm, 4) or nperm != A:
            raise ValueError("aggregate_forward: qbase (sorted mode) needs the full perm of aggregation_order(sorted_operands=True), "
                             "variant 0 / 8 / 13, no split and shape (A, 4)")
        variant = 13 if variant == 13 else 8
    if variant == 13:
        if qbase is None or lists is None or lists.rows < A:
            raise ValueError("aggregate_forward: variant 13 (two-kernel split) needs the sorted mode's qbase and an AggLists workspace of >= A rows")
    if tables is not None:
        _chk(tables, "tables", torch.float32, 2)
        if tuple(tables.shape) != (2 + N, L * P * G):
            raise ValueError("aggregate_forward: tables must be (2+N, L*P*G) = %s, got %s" % ((2 + N, L * P * G), tuple(tables.shape)))
    hw_keep, hw_p = _host_i32([list(x) for x in level_hw])
    st_keep, st_p = _host_i32(list(level_start))
    pc_keep, pc_p = _host_f32(list(pc_range))
    _lib.check(lib.far3d_aggregate_forward(_ptr(feat), _dt(feat), _ptr(ref), _ptr(offsets), _ptr(lidar2img),
                                           _ptr(U), _ptr(Vc), _ptr(tables) if tables is not None else None,
                                           _ptr(perm) if perm is not None else None, _ptr(out), _dt(out),
                                           A if perm is None else nperm, N, S, C, G, P, L, hw_p, st_p, pc_p,
                                           float(pad_hw[0]), float(pad_hw[1]), U.stride(0) if A > 0 else 0,
                                           offsets.stride(0) if A > 0 else 0, int(variant),
                                           _ptr(split.partials) if split is not None else (_ptr(lists.buf) if lists is not None else None),
                                           _ptr(split.tickets) if split is not None else (_ptr(lists.counts) if lists is not None else None),
                                           split.extra if split is not None else 0, _ptr(qbase) if qbase is not None else None,
                                           _stream(feat)),
               "far3d_aggregate_forward")
    return out


# --------------------------------------------------------------------------------------------------
# implicit-GEMM convolution / linear
# --------------------------------------------------------------------------------------------------
ACT = {None: 0, "none": 0, "relu": 1, "swish": 2}
_TUNING = {}


# Tile tables conv2d_nhwc / linear consult when the caller passes tile = 0.  The two names below are the PROCESS DEFAULTS (what a
# stand-alone op or a plugin module gets); an engine selects its own tables for the duration of its entry points with
# use_tile_tables(), which is thread-local and restores the previous selection on exit -- an engine that ran earlier, or one driven
# from another thread, never changes what somebody else's launches look up (the tables used to be re-assigned module globals).
BF16_TILE_TABLE = "tuning_mi355x.json"
PAIR_TILE_TABLE = "tuning_mi355x_pair.json"   # pair-stored activations (the bf16x3 engine mode)
_PAIR_TABLE_OF = {"tuning_mi355x.json": "tuning_mi355x_pair.json", "tuning_mi355x_tput.json": "tuning_mi355x_pair_tput.json"}
_TABLE_SEL = threading.local()


def pair_table_for(bf16_table):
    """The pair-storage tile table that goes with a bf16 one: the shipped pairs by name, `<stem>_pair.json` beside a custom table when
    that file exists, else the default pair table -- never the bf16 table itself."""
    if bf16_table in _PAIR_TABLE_OF:
        return _PAIR_TABLE_OF[bf16_table]
    import os
    cand = bf16_table[:-5] + "_pair.json" if bf16_table.endswith(".json") else bf16_table + "_pair"
    if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", cand)):
        return cand
    return "tuning_mi355x_pair.json"


def tile_tables():
    """(bf16 table, pair table) in force for this thread."""
    return getattr(_TABLE_SEL, "cur", None) or (BF16_TILE_TABLE, PAIR_TILE_TABLE)


@contextlib.contextmanager
def use_tile_tables(bf16_table, pair_table=None):
    """Select the tile tables for the launches issued (or captured into a graph) inside the block, for this thread only."""
    prev = getattr(_TABLE_SEL, "cur", None)
    _TABLE_SEL.cur = (bf16_table, pair_table or pair_table_for(bf16_table))
    try:
        yield
    finally:
        _TABLE_SEL.cur = prev


WS_TILES = range(400, 478)      # persistent wave-specialised kernels (csrc/conv_ws.hpp; 400-459 3x3, 460-477 1x1 GEMM on pair maps; 479-481 are fp32-row GEMM tiles): bias + activation + pair / bf16 store only


def _tuned_tile(Cout, Cin, k, stride, npix, table=None, ws_ok=False):
    """Workgroup tile measured fastest on MI355X for this conv shape (tools/tune_conv.py); 0 = kernel heuristic.
    Shapes that were not swept (e.g. fewer cameras per rank in camera-sharded mode) borrow the entry of the same layer
    geometry with the closest pixel count.  table: tuning_mi355x.json (bf16) or tuning_mi355x_bf16x3.json (split mode).
    An entry may be a pair [ws tile, other tile]: the wave-specialised kernel where the call allows it (ws_ok: no residual, second
    output or channel sums, same storage in and out), else the fastest of the general kernels -- an explicit choice per call, the
    library itself never substitutes a kernel."""
    if table is None:
        table = tile_tables()[0]
    tab = _TUNING.get(table)
    if tab is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", table)
        raw = json.load(open(path)) if os.path.exists(path) else {}
        tab = _TUNING[table] = {}
        for key, tile in raw.items():
            co, ci, kk, st, npx = (int(v) for v in key.split(","))
            tab.setdefault((co, ci, kk, st), []).append((npx, tuple(int(t) for t in tile) if isinstance(tile, (list, tuple)) else int(tile)))
    cands = tab.get((Cout, Cin, k, stride))
    if not cands:
        return 0
    npx, tile = min(cands, key=lambda c: abs(c[0] - npix))
    if isinstance(tile, tuple):
        # a persistent tile was measured at ITS pixel count (its grid is one workgroup per CU walking npix / tile items): a borrowed entry
        # (camera-sharded ranks see 1/2 .. 1/7 of the pixels) keeps the general tile
        return tile[0] if (ws_ok and 4 * abs(npx - npix) <= npx) else tile[1]
    return tile


class PackedConv:
    """Weights of one conv / linear layer in the kernel's layout (built once at model-prepare time).

    w (rows=ceil(Cout/128)*128, taps*cin_pad) in the compute dtype (bf16 -> bf16 MFMA, fp32 -> exact fp32
    MFMA), tap-major K; bias (rows,) f32 or None.  compute="bf16x3" (fp32 weights and activations only): products as a
    two-term bf16 split on the bf16 MFMA with fp32 accumulation (FAR3D_DT_F32_BF16X3).
    """

    def __init__(self, weight, bias=None, stride=1, pad=0, dtype=torch.bfloat16, device=None, compute=None):
        if compute not in (None, "bf16x3") or (compute == "bf16x3" and dtype != torch.float32):
            raise ValueError("PackedConv: compute=%r needs dtype=float32 (got %s)" % (compute, dtype))
        if weight.dim() == 2:
            weight = weight[:, :, None, None]
        Cout, Cin, KH, KW = weight.shape
        device = device if device is not None else weight.device
        cin_pad = (Cin + 31) // 32 * 32
        rows = (Cout + 255) // 256 * 256 + 256   # zero rows so that ANY channel tile (<= 256 rows) may over-read
        w = torch.zeros(rows, KH * KW, cin_pad, dtype=torch.float32, device=weight.device)
        w[:Cout, :, :Cin] = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, KH * KW, Cin)
        if compute == "bf16x3":    # pre-split rows: per 32-channel block [32 hi | 32 lo] bf16 (include/far3d_hip.h)
            blk = w.reshape(rows, KH * KW * cin_pad // 32, 32)
            hi = blk.to(torch.bfloat16)
            lo = (blk - hi.float()).to(torch.bfloat16)
            self.w = torch.cat([hi, lo], dim=2).reshape(rows, 2 * KH * KW * cin_pad).to(device).contiguous()
        else:
            self.w = w.reshape(rows, KH * KW * cin_pad).to(dtype).to(device).contiguous()
        self.bias = None
        if bias is not None:
            b = torch.zeros(rows, dtype=torch.float32, device=weight.device)
            b[:Cout] = bias.detach().float()
            self.bias = b.to(device)
        self.Cout, self.Cin, self.KH, self.KW, self.stride, self.pad = Cout, Cin, KH, KW, stride, pad
        self.w_code = DT_F32_BF16X3 if compute == "bf16x3" else _dt(self.w)
        # pair-stored inputs only: 3 = split products (default); 1 = hi halves only, i.e. a single-bf16 layer inside a pair-stored
        # network (the per-layer precision assignment measured by tools/precision_sweep.py)
        self.terms = 3

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


SUMS_MAX_PARTS = 32     # FAR3D_SUMS_MAX_PARTS in include/far3d_hip.h


def ese_scratch_floats(N, C):
    """FAR3D_SUMS_SCRATCH_FLOATS(N, C): workspace of ese_nhwc / groupnorm_nhwc (never needs zeroing)."""
    return N * C * (2 * SUMS_MAX_PARTS + 1)


def _is_pair_input(x, pc):
    return pc.w_code == DT_F32_BF16X3 and x.dtype == torch.bfloat16


_HI_ONLY_TILE = {160: 260, 165: 265, 152: 252, 179: 279, 180: 280}


def _pair_tile(pc, Cin, npix, tile, ws_ok=False):
    """Tile id for a pair-stored input: the measured split-product tile (far3d_amd/data/tuning_mi355x_pair.json, ids 150+, 0 = the
    library default), mapped to a hi-planes-only kernel when the layer is assigned a single bf16 product (pc.terms == 1)."""
    fast = pc.stride == 1 and ((pc.KH == 3 and pc.pad == 1) or (pc.KH == 1 and pc.pad == 0))
    if tile == 0:
        tile = _tuned_tile(pc.Cout, Cin, pc.KH, pc.stride, npix, tile_tables()[1], ws_ok=ws_ok and pc.terms != 1)
    if pc.terms == 1 and fast and (tile == 0 or 150 <= tile < 200):
        tile = _HI_ONLY_TILE.get(tile, 260 if pc.KH == 3 else 279)
    return tile


def conv_tile(x, pc):
    """The tile conv2d_nhwc(tile=0) will use for this input (0 = kernel heuristic)."""
    N, H, W, Cin = x.shape
    if _is_pair_input(x, pc):
        return _pair_tile(pc, Cin // 2, N * H * W, 0)
    return _tuned_tile(pc.Cout, Cin, pc.KH, pc.stride, N * H * W) if x.dtype == torch.bfloat16 else 0


def conv2d_nhwc(x, pc, out=None, act=None, out_dtype=None, res=None, y2=None, y2_scale=None, y2_shift=None, tile=0, sums=None):
    """x: (N,H,W,Cin) NHWC view (f32|bf16).  pc: PackedConv.  out: optional (N,Ho,Wo,Cout) NHWC view to write
    into (e.g. a channel slice of an OSA concat buffer).  Returns out.
    Pair storage: with split weights (compute="bf16x3") a bf16 `x` is a pair-stored map (N,H,W,2*Cin) (pair_from_float); a bf16
    `out` / `res` is then pair-stored too ((..., 2*Cout)), an f32 `out` / `y2` is plain.
    sums: optional zeroed int64 (N, Cout) device tensor -- the layer's epilogue adds the fixed-point channel sums of its stored output
    (far3d_hip.h, chan_sums; see conv_can_fuse_sums); ese_nhwc(sums=...) consumes them."""
    lib = _lib.require_device()
    N, H, W, Cx = x.shape
    pair_in = _is_pair_input(x, pc)
    Cin = Cx // 2 if pair_in else Cx
    if Cin != pc.Cin or (pair_in and Cx % 64):
        raise ValueError("conv2d_nhwc: input has %d channels, layer expects %d" % (Cin, pc.Cin))
    ldx, xs = _nhwc_view(x, "x")
    Ho, Wo = pc.out_hw(H, W)
    if out is None:
        odt = out_dtype or x.dtype
        out = torch.empty((N, Ho, Wo, pc.Cout * (2 if pair_in and odt == torch.bfloat16 else 1)), dtype=odt, device=x.device)
    pair_out = pair_in and out.dtype == torch.bfloat16
    cso = 2 if pair_out else 1
    if tuple(out.shape) != (N, Ho, Wo, pc.Cout * cso):
        raise ValueError("conv2d_nhwc: out shape %s != %s" % (tuple(out.shape), (N, Ho, Wo, pc.Cout * cso)))
    ldy, ys = _nhwc_view(out, "out")
    rp, rdt, ldr, rs, Hr, Wr = None, 0, 0, 0, Ho, Wo
    if res is not None:
        ldr, rs = _nhwc_view(res, "res")
        pair_res = pair_in and res.dtype == torch.bfloat16
        if res.shape[0] != N or res.shape[3] != pc.Cout * (2 if pair_res else 1):
            raise ValueError("conv2d_nhwc: residual shape %s incompatible" % (tuple(res.shape),))
        rp, rdt, Hr, Wr = _ptr(res), DT_BF16_PAIR if pair_res else _dt(res), res.shape[1], res.shape[2]
    y2p, y2dt, ldy2, y2s, sp, hp = None, 0, 0, 0, None, None
    if y2 is not None:
        ldy2, y2s = _nhwc_view(y2, "y2")
        _chk(y2_scale, "y2_scale", torch.float32)
        _chk(y2_shift, "y2_shift", torch.float32)
        if y2_scale.numel() != N * pc.Cout or y2_shift.numel() != N * pc.Cout or tuple(y2.shape) != (N, Ho, Wo, pc.Cout):
            raise ValueError("conv2d_nhwc: y2/scale/shift shapes inconsistent")
        y2p, y2dt, sp, hp = _ptr(y2), _dt(y2), _ptr(y2_scale), _ptr(y2_shift)
    # what the wave-specialised kernels cover (anything else takes the table's general tile)
    ws_ok = (res is None and y2 is None and pc.stride == 1 and
             ((pc.KH == 3 and pc.KW == 3 and pc.pad == 1 and sums is None) or
              (pc.KH == 1 and pc.KW == 1 and pc.pad == 0 and pair_in and (sums is None or (Ho * Wo >= 256 and Cin >= 192)))) and
             Cin % 32 == 0 and pc.Cout % 32 == 0 and out.dtype == torch.bfloat16 and ldx % 8 == 0 and ldy % 8 == 0 and
             x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and xs % 8 == 0 and ys % 8 == 0)
    if pair_in:
        tile = _pair_tile(pc, Cin, N * H * W, tile, ws_ok)
    elif tile == 0 and x.dtype == torch.bfloat16:
        tile = _tuned_tile(pc.Cout, Cin, pc.KH, pc.stride, N * H * W, ws_ok=ws_ok)
    elif tile == 0 and pc.w_code == DT_F32_BF16X3:
        tile = _tuned_tile(pc.Cout, Cin, pc.KH, pc.stride, N * H * W, "tuning_mi355x_bf16x3.json")
    _lib.check(lib.far3d_conv2d_nhwc(
        _ptr(x), DT_BF16_PAIR if pair_in else _dt(x), _ptr(pc.w), pc.w_code, _ptr(pc.bias) if pc.bias is not None else None, _ptr(out),
        DT_BF16_PAIR if pair_out else _dt(out),
        N, H, W, Cin, ldx, xs, Ho, Wo, pc.Cout, ldy, ys, pc.KH, pc.KW, pc.stride, pc.pad, ACT[act],
        rp, rdt, ldr, rs, Hr, Wr, y2p, y2dt, ldy2, y2s, sp, hp, _sums_ptr(sums, N, pc.Cout), tile,
        _stream(x)), "far3d_conv2d_nhwc")
    return out


# linear(): tile for fp32 rows x fp32 weights (exact fp32 MFMA on the pipelined GEMM kernel, far3d_hip.h tiles 482-494); 0 = the
# register-staged kernel of rounds 1-5.  482 = 64 x 64, 2 stages: the fastest of 482-486 on every decoder shape; 487-494 split K between
# wave groups INSIDE the workgroup (a 32 x 32 tile over K = 1 024 is a 16 us serial MFMA chain however idle the chip is): 493 (32 x 64, 2
# groups) wins the K <= 512 shapes of up to 768 columns, 489 (32 x 64, 4 groups) the K = 1 024 one, 482 stays for >= 1 024 columns where
# the tiles already fill the chip (tools/probe/f32x_gemm_ab.py, profiles/r6/f32x_gemm_ab.txt).
F32X_LINEAR_TILE = 482


def f32x_linear_tile(cout, K):
    """The exact-fp32 tile of linear() for a (cout, K) weight.  A function of the WEIGHT's shape only, never of the number of rows: the
    tile fixes the order in which a row's products are added, and the query-sharded decoder (row subsets) reproduces the replicated one
    bit for bit."""
    if not F32X_LINEAR_TILE:
        return 0
    if K >= 1024:
        return 489
    return F32X_LINEAR_TILE if cout >= 1024 else 493
SUMS_FRAC_BITS = 18      # FAR3D_SUMS_FRAC_BITS in include/far3d_hip.h
# GEMM tiles that exist AND leave LDS for the channel-sum scratch (114 / 115 fill the 160 KB with their ring; 118 / 119 are no kernels)
_GEMM_TILES = set(range(70, 90)) | {110, 111, 112, 113, 116, 117} | set(range(120, 130)) | set(range(140, 146)) | set(range(170, 182)) | set(range(185, 189)) | {279, 280} | set(range(460, 478))
_GEMM_TILE_PIXELS = 512  # no GEMM tile holds more pixels


def _sums_ptr(sums, N, C):
    if sums is None:
        return None
    if sums.dtype != torch.int64 or sums.numel() < N * C or not sums.is_contiguous():
        raise ValueError("channel sums: contiguous int64 tensor of at least N*C = %d elements" % (N * C))
    return _ptr(sums)


def conv_can_fuse_sums(x, pc, out_dtype=torch.bfloat16):
    """True if conv2d_nhwc(x, pc, sums=...) is possible: a 1x1 / stride-1 layer whose measured tile is one of the pipelined GEMM
    kernels, bf16 or pair output, and a map of at least one pixel tile (otherwise the eSE op pools the map itself)."""
    N, H, W, Cx = x.shape
    if pc.KH != 1 or pc.KW != 1 or pc.stride != 1 or pc.pad != 0 or out_dtype != torch.bfloat16 or x.dtype != torch.bfloat16:
        return False
    if H * W < _GEMM_TILE_PIXELS or pc.Cout % 8:
        return False
    pair_in = _is_pair_input(x, pc)
    Cin = Cx // 2 if pair_in else Cx
    if Cin % 32:
        return False
    tile = _pair_tile(pc, Cin, N * H * W, 0) if pair_in else _tuned_tile(pc.Cout, Cin, 1, 1, N * H * W)
    return tile in _GEMM_TILES or (pair_in and tile == 0)


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
    if tile == 0 and x.dtype == torch.bfloat16 and pc.w.dtype == torch.bfloat16 and x.shape[1] % 32 == 0 and \
            _tuned_tile(pc.Cout, x.shape[1], 1, 1, M) == 0:
        tile = 80     # untuned small GEMM: 64x64 pipelined tile (the winner on every decoder-sized shape of the table)
    if tile == 0 and F32X_LINEAR_TILE and x.dtype == torch.float32 and pc.w.dtype == torch.float32 and pc.w_code == DT_F32 and \
            x.shape[1] % 32 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and M * x.stride(0) * 4 < 2 ** 31 - 1:
        # exact fp32 on the LDS-DMA pipelined kernel (round 6): the GEMMs of the fp32 / in-tolerance engines' decoder and FarHead.  The
        # choice must NOT depend on the number of rows: the query-sharded decoder launches row subsets and has to reproduce the replicated
        # decoder bit for bit (a row's bits depend on the kernel, not on which rows share its launch)
        tile = f32x_linear_tile(pc.Cout, x.shape[1])
    conv2d_nhwc(xv, pc, out=ov, act=act, res=rv, tile=tile)
    return out


# --------------------------------------------------------------------------------------------------
# row-resident decoder chains (csrc/rowchain.hip)
# --------------------------------------------------------------------------------------------------
ROWCHAIN_E, ROWCHAIN_FF, ROWCHAIN_WL_TILES = 256, 1024, 29


def pack_rowchain(pc, cols=None):
    """A PackedConv's (rows, K) bf16 weight in the MFMA fragment order the row chains stream (include/far3d_hip.h):
    [ceil(cols / 16)][K / 32][64][8], element j of lane l of (tile t, step s) = W[16 t + (l & 15)][32 s + 8 (l >> 4) + j].
    Returns (packed weight, f32 bias covering the padded columns)."""
    w = pc.w
    if w.dtype != torch.bfloat16 or pc.KH * pc.KW != 1 or pc.w_code != DT_BF16:
        raise ValueError("pack_rowchain: plain bf16 linear weights only")
    K = w.shape[1]
    if K % 256 != 0:
        raise ValueError("pack_rowchain: K=%d must be a multiple of 256" % K)
    cols = pc.Cout if cols is None else cols
    nt = -(-cols // 16)
    if nt * 16 > w.shape[0]:
        raise ValueError("pack_rowchain: %d columns exceed the packed weight's %d rows" % (nt * 16, w.shape[0]))
    frag = w[:nt * 16].reshape(nt, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()      # [t][s][l >> 4][l & 15][j]
    bias = pc.bias[:nt * 16].contiguous() if pc.bias is not None else torch.zeros(nt * 16, dtype=torch.float32, device=w.device)
    return frag.reshape(nt, K // 32, 64, 8), bias


class RowChainLayer:
    """The packed operands of one decoder layer's two row chains.  ly: the engine's layer dict (PackedConvs out / wl / oproj /
    ffn1 / ffn2 / qkv and the three (gamma, beta) pairs)."""

    def __init__(self, ly):
        self.out, self.b_out = pack_rowchain(ly["out"])
        self.wl, self.b_wl = pack_rowchain(ly["wl"])
        self.n_wl = ly["wl"].Cout
        self.oproj, self.b_oproj = pack_rowchain(ly["oproj"])
        self.ffn1, self.b_ffn1 = pack_rowchain(ly["ffn1"])
        self.ffn2, self.b_ffn2 = pack_rowchain(ly["ffn2"])
        self.qkv, self.b_qkv = pack_rowchain(ly["qkv"])
        self.norms = [(g.contiguous(), b.contiguous()) for g, b in ly["norms"]]

    @staticmethod
    def supported(ly, E, dtype):
        """The chains are built for the benchmark's decoder geometry (E = 256, FFN 1024, 449..464 aggregation outputs, bf16)."""
        return (dtype == torch.bfloat16 and E == ROWCHAIN_E and ly["ffn1"].Cout == ROWCHAIN_FF and ly["ffn2"].Cin == ROWCHAIN_FF and
                16 * (ROWCHAIN_WL_TILES - 1) < ly["wl"].Cout <= 16 * ROWCHAIN_WL_TILES and ly["wl"].Cin == 2 * E and
                ly["qkv"].Cin == 2 * E and ly["qkv"].Cout == 3 * E and all(ly[k].w.dtype == torch.bfloat16 and ly[k].w_code == DT_BF16
                                                                           for k in ("out", "wl", "oproj", "ffn1", "ffn2", "qkv")))


class RowChainBranches:
    """Packed operands of far3d_rowchain_branches.  cls / reg: the three PackedConvs of each branch; cls_ln: two (gamma, beta)."""

    def __init__(self, cls, cls_ln, reg):
        (self.c0, self.b_c0), (self.c1, self.b_c1) = (pack_rowchain(pc) for pc in cls[:2])
        (self.r0, self.b_r0), (self.r1, self.b_r1) = (pack_rowchain(pc) for pc in reg[:2])
        (self.c2, self.b_c2), (self.r2, self.b_r2) = pack_rowchain(cls[2], cols=32), pack_rowchain(reg[2], cols=32)   # two tiles each
        self.cls_ln = [(g.contiguous(), b.contiguous()) for g, b in cls_ln]
        self.n_cls, self.n_reg = cls[2].Cout, reg[2].Cout

    @staticmethod
    def supported(cls, reg, E, dtype):
        pcs = list(cls) + list(reg)
        return (dtype == torch.bfloat16 and E == ROWCHAIN_E and all(pc.w.dtype == torch.bfloat16 and pc.w_code == DT_BF16 and pc.Cin == E for pc in pcs) and
                all(pc.Cout == E for pc in (cls[0], cls[1], reg[0], reg[1])) and 1 <= cls[2].Cout <= 32 and 1 <= reg[2].Cout <= 32)


def _rows(t, name, dtype, cols):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != dtype or t.shape[1] < cols or not t.is_cuda:
        raise ValueError("%s must be a (M, >=%d) %s device tensor with unit inner stride" % (name, cols, dtype))
    return t


def rowchain_attn_out(att, x, qpos, rc, x1, ul, eps=1e-5, ul_rows=None):
    """x1 = LN0(att @ W_out^T + b + x); ul[:, :n_wl] = [x1 + qpos | x1] @ W_wl^T + b  -- one launch (far3d_rowchain_attn_out).
    att (M,E) bf16; x, qpos (M,E) f32; x1 (M,E) f32 out; ul (M, >= n_wl) f32 out.  rc: RowChainLayer.
    ul_rows (M) int32: row i of ul is stored at row ul_rows[i] (aggregation_order's inv: the aggregation kernel's launch order)."""
    lib = _lib.require_device()
    E = ROWCHAIN_E
    M = att.shape[0]
    _rows(att, "att", torch.bfloat16, E); _rows(x, "x", torch.float32, E); _rows(qpos, "qpos", torch.float32, E)
    _rows(x1, "x1", torch.float32, E); _rows(ul, "ul", torch.float32, rc.n_wl)
    for t in (x, qpos, x1, ul):
        if t.shape[0] != M:
            raise ValueError("rowchain_attn_out: row counts differ")
    g0, be0 = rc.norms[0]
    if ul_rows is not None:
        _chk(ul_rows, "ul_rows", torch.int32, 1)
        if ul_rows.numel() != M:
            raise ValueError("rowchain_attn_out: ul_rows needs one entry per row")
    _lib.check(lib.far3d_rowchain_attn_out(_ptr(att), att.stride(0), _ptr(x), x.stride(0), _ptr(qpos), qpos.stride(0),
                                           _ptr(rc.out), _ptr(rc.b_out), _ptr(g0), _ptr(be0), _ptr(rc.wl), _ptr(rc.b_wl), rc.n_wl,
                                           _ptr(x1), x1.stride(0), _ptr(ul), ul.stride(0), _ptr(ul_rows) if ul_rows is not None else None,
                                           M, float(eps), _stream(att)),
               "far3d_rowchain_attn_out")
    return x1, ul


def rowchain_ffn(agg, x1, qpos, rc, out, nxt=None, qkv=None, xop=None, eps=1e-5):
    """x2 = LN1(agg @ W_o^T + b + x1); out = LN2(relu(x2 @ W_1^T + b) @ W_2^T + b + x2); with nxt (the NEXT layer's
    RowChainLayer): qkv = [out + qpos | out] @ W_qkv^T + b (bf16, (M, 3E)) -- one launch (far3d_rowchain_ffn).
    xop: optional (M, 2E) bf16 out, [out + qpos | out]."""
    lib = _lib.require_device()
    E = ROWCHAIN_E
    M = agg.shape[0]
    _rows(agg, "agg", torch.bfloat16, E); _rows(x1, "x1", torch.float32, E); _rows(qpos, "qpos", torch.float32, E)
    _rows(out, "out", torch.float32, E)
    if (nxt is None) != (qkv is None):
        raise ValueError("rowchain_ffn: nxt and qkv go together")
    if qkv is not None:
        _rows(qkv, "qkv", torch.bfloat16, 3 * E)
    if xop is not None:
        _rows(xop, "xop", torch.bfloat16, 2 * E)
    for t in (x1, qpos, out, qkv, xop):
        if t is not None and t.shape[0] != M:
            raise ValueError("rowchain_ffn: row counts differ")
    (g1, be1), (g2, be2) = rc.norms[1], rc.norms[2]
    _lib.check(lib.far3d_rowchain_ffn(_ptr(agg), agg.stride(0), _ptr(x1), x1.stride(0), _ptr(qpos), qpos.stride(0),
                                      _ptr(rc.oproj), _ptr(rc.b_oproj), _ptr(g1), _ptr(be1), _ptr(rc.ffn1), _ptr(rc.b_ffn1),
                                      _ptr(rc.ffn2), _ptr(rc.b_ffn2), _ptr(g2), _ptr(be2),
                                      _ptr(nxt.qkv) if nxt is not None else None, _ptr(nxt.b_qkv) if nxt is not None else None,
                                      _ptr(out), out.stride(0), _ptr(qkv) if qkv is not None else None,
                                      qkv.stride(0) if qkv is not None else 0, _ptr(xop) if xop is not None else None,
                                      xop.stride(0) if xop is not None else 0, M, float(eps), _stream(agg)), "far3d_rowchain_ffn")
    return out


def rowchain_qkv(x, qpos, rc, qkv):
    """qkv (M,3E) bf16 = [x + qpos | x] @ W_qkv^T + b of layer `rc` (RowChainLayer) -- bit-identical to what rowchain_ffn(..., nxt=rc)
    writes for the same rows (far3d_rowchain_qkv)."""
    lib = _lib.require_device()
    E = ROWCHAIN_E
    M = x.shape[0]
    _rows(x, "x", torch.float32, E); _rows(qpos, "qpos", torch.float32, E); _rows(qkv, "qkv", torch.bfloat16, 3 * E)
    if qpos.shape[0] != M or qkv.shape[0] != M:
        raise ValueError("rowchain_qkv: row counts differ")
    _lib.check(lib.far3d_rowchain_qkv(_ptr(x), x.stride(0), _ptr(qpos), qpos.stride(0), _ptr(rc.qkv), _ptr(rc.b_qkv), _ptr(qkv), qkv.stride(0),
                                      M, _stream(x)), "far3d_rowchain_qkv")
    return qkv


def rowchain_branches(h, rb, cls_out, reg_out, eps=1e-5):
    """cls_out (M, n_cls), reg_out (M, n_reg) f32 = the classification / regression branches of h (M,E) bf16 in one launch
    (far3d_rowchain_branches).  rb: RowChainBranches."""
    lib = _lib.require_device()
    M = h.shape[0]
    _rows(h, "h", torch.bfloat16, ROWCHAIN_E); _rows(cls_out, "cls_out", torch.float32, rb.n_cls); _rows(reg_out, "reg_out", torch.float32, rb.n_reg)
    if cls_out.shape[0] != M or reg_out.shape[0] != M:
        raise ValueError("rowchain_branches: row counts differ")
    (g0, be0), (g1, be1) = rb.cls_ln
    _lib.check(lib.far3d_rowchain_branches(_ptr(h), h.stride(0), _ptr(rb.c0), _ptr(rb.b_c0), _ptr(g0), _ptr(be0), _ptr(rb.c1), _ptr(rb.b_c1),
                                           _ptr(g1), _ptr(be1), _ptr(rb.c2), _ptr(rb.b_c2), rb.n_cls, _ptr(rb.r0), _ptr(rb.b_r0),
                                           _ptr(rb.r1), _ptr(rb.b_r1), _ptr(rb.r2), _ptr(rb.b_r2), rb.n_reg, _ptr(cls_out), cls_out.stride(0),
                                           _ptr(reg_out), reg_out.stride(0), M, float(eps), _stream(h)), "far3d_rowchain_branches")
    return cls_out, reg_out


# --------------------------------------------------------------------------------------------------
# attention / normalisation / pooling
# --------------------------------------------------------------------------------------------------
def attention_f32_variant(variant=-1):
    """A/B switch of the fp32 attention instantiation (include/far3d_hip.h); returns the previous value, -1 only reads."""
    return _lib.load().far3d_attention_f32_variant(int(variant))


def attention_forward(q, k, v, num_heads=8, out=None, out_dtype=torch.float32, hole=None):
    """softmax(q k^T / sqrt(d)) v per head.  q (Aq,E), k/v (Nk,E) f32|bf16 with unit inner stride; out (Aq,E) f32.
    hole: keys [start + count, end) are masked (see _hole)."""
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
        out = torch.empty((Aq, E), dtype=out_dtype, device=q.device)
    _lib.check(lib.far3d_attention_forward(_ptr(q), _ptr(k), _ptr(v), _dt(q), _ptr(out), _dt(out), Aq, Nk, num_heads, hd,
                                           q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                           float(hd) ** -0.5, *_hole(hole), _stream(q)), "far3d_attention_forward")
    return out


def layernorm(x, gamma, beta, eps=1e-5, act=None, add=None, out=None, add_dtype=torch.float32, bf16_copy=False, y2=None, yb=None,
              out_rows=None):
    """Returns LN(x); with `add`: (LN(x), LN(x)+add [add_dtype]); with bf16_copy also a bf16 copy of LN(x) (last).
    out_rows (rows) int32: row i of y2 / yb is stored at row out_rows[i] (far3d_layernorm_rows; LN(x) itself stays in place).
    y2 / yb: optional preallocated (rows,C) outputs (f32|bf16, unit inner stride, any row stride) for LN(x)+add and the
    copy of LN(x) -- e.g. the two halves of one (rows,2C) [x+pos | x] merged-GEMM operand; they are returned in place of
    freshly allocated ones."""
    lib = _lib.require_device()
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32:
        raise ValueError("layernorm: x must be (rows,C) f32 with unit inner stride")
    rows, C = x.shape
    y = out if out is not None else torch.empty((rows, C), dtype=torch.float32, device=x.device)
    if y2 is None and add is not None:
        y2 = torch.empty((rows, C), dtype=add_dtype, device=x.device)
    if yb is None and bf16_copy:
        yb = torch.empty((rows, C), dtype=torch.bfloat16, device=x.device)
    if y2 is not None and add is None:
        raise ValueError("layernorm: y2 needs add")
    for t in (y, y2, yb):
        if t is not None and (t.stride(1) != 1 or tuple(t.shape) != (rows, C)):
            raise ValueError("layernorm: outputs must be (rows,C) with unit inner stride")
    args = (_ptr(x), _ptr(gamma) if gamma is not None else None,
            _ptr(beta) if beta is not None else None, _ptr(y), rows, C, x.stride(0), y.stride(0),
            float(eps), 1 if act == "relu" else 0,
            _ptr(add) if add is not None else None, add.stride(0) if add is not None else 0,
            _ptr(y2) if y2 is not None else None, y2.stride(0) if y2 is not None else 0,
            _dt(y2) if y2 is not None else 0, _ptr(yb) if yb is not None else None,
            yb.stride(0) if yb is not None else 0, _dt(yb) if yb is not None else 0)
    if out_rows is not None:
        _chk(out_rows, "out_rows", torch.int32, 1)
        if out_rows.numel() != rows:
            raise ValueError("layernorm: out_rows needs one entry per row")
        _lib.check(lib.far3d_layernorm_rows(*args, _ptr(out_rows), _stream(x)), "far3d_layernorm_rows")
    else:
        _lib.check(lib.far3d_layernorm(*args, _stream(x)), "far3d_layernorm")
    res = (y,) + ((y2,) if y2 is not None else ()) + ((yb,) if yb is not None else ())
    return res if len(res) > 1 else y


def ese_nhwc(x, fcw, fcb, identity=None, out=None, scratch=None, pair=False, sums=None):
    """x * hsigmoid(fc(mean_hw x)) (+ identity) on NHWC views (any channel slice / pixel stride).  scratch: optional
    ese_scratch_floats(N, C) f32 workspace (never needs zeroing; the pooling is deterministic).  pair: x / identity / out are
    pair-stored bf16 maps (2C stored channels).  sums: the int64 channel sums conv2d_nhwc(..., sums=) accumulated while it produced
    x -- the pooling pass is skipped and the sums come back zeroed."""
    lib = _lib.require_device()
    N, H, W, C = x.shape
    if pair:
        if x.dtype != torch.bfloat16 or C % 64:
            raise ValueError("ese_nhwc: pair storage is bf16 with a multiple of 64 stored channels")
        C //= 2
    ldx, xs = _nhwc_view(x, "x")
    if out is None:
        out = torch.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    ldy, ys = _nhwc_view(out, "out")
    ldi, isd, ip = 0, 0, None
    if identity is not None:
        ldi, isd = _nhwc_view(identity, "identity")
        ip = _ptr(identity)
    if scratch is None:
        scratch = torch.empty(ese_scratch_floats(N, C), dtype=torch.float32, device=x.device)
    if scratch.numel() < ese_scratch_floats(N, C):
        raise ValueError("ese_nhwc: scratch needs %d floats" % ese_scratch_floats(N, C))
    _lib.check(lib.far3d_ese_nhwc(_ptr(x), DT_BF16_PAIR if pair else _dt(x), _ptr(fcw), _ptr(fcb), ip, _ptr(out), _ptr(scratch), N, H * W, C,
                                  ldx, xs, ldi, isd, ldy, ys, _sums_ptr(sums, N, C), _stream(x)), "far3d_ese_nhwc")
    return out


def ese_fused_ok(x, pair=False):
    """Whether ese_fused_nhwc takes this map: bf16 / pair storage, 8-channel (16-byte) pieces everywhere."""
    C = x.shape[-1] // (2 if pair else 1)
    return x.dtype == torch.bfloat16 and C % (32 if pair else 8) == 0 and C <= 1024 and x.stride(2) % 8 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0


def maxpool_out_hw(H, W):
    """MaxPool2d(3, 2, ceil_mode=True) output size (ref models/backbones/vovnet.py:249-250)."""
    Ho, Wo = -(-(H - 3) // 2) + 1, -(-(W - 3) // 2) + 1
    if (Ho - 1) * 2 >= H:
        Ho -= 1
    if (Wo - 1) * 2 >= W:
        Wo -= 1
    return Ho, Wo


def ese_fused_nhwc(x, fcw, fcb, sums, gate, identity=None, out=None, pooled=None, pair=False):
    """eSE with the stage-end pooling fused into the apply pass (far3d_ese_fused_nhwc): gates from the fixed-point channel sums `sums` of
    the concat convolution, y = x * gate (+ identity) into `out` (None: not written) and, optionally, MaxPool2d(3, 2, ceil_mode=True)(y) into
    `pooled` (an NHWC view, e.g. the input slice of the next stage's concat buffer).  gate: >= N*C floats of workspace."""
    lib = _lib.require_device()
    N, H, W, C = x.shape
    if pair:
        C //= 2
    if out is None and pooled is None:
        raise ValueError("ese_fused_nhwc: nothing to write (out and pooled are both None)")
    ldx, xs = _nhwc_view(x, "x")
    ldy, ys, yp = 0, 0, None
    if out is not None:
        ldy, ys = _nhwc_view(out, "out")
        yp = _ptr(out)
    ldi, isd, ip = 0, 0, None
    if identity is not None:
        ldi, isd = _nhwc_view(identity, "identity")
        ip = _ptr(identity)
    Hp = Wp = ldp = ps = 0
    pp = None
    if pooled is not None:
        Hp, Wp = pooled.shape[1], pooled.shape[2]
        ldp, ps = _nhwc_view(pooled, "pooled")
        pp = _ptr(pooled)
    if gate.numel() < N * C or gate.dtype != torch.float32:
        raise ValueError("ese_fused_nhwc: gate needs N*C floats")
    _lib.check(lib.far3d_ese_fused_nhwc(_ptr(x), DT_BF16_PAIR if pair else _dt(x), _ptr(fcw), _ptr(fcb), ip, yp, pp, _ptr(gate), N, H, W, C,
                                        ldx, xs, ldi, isd, ldy, ys, Hp, Wp, ldp, ps, _sums_ptr(sums, N, C), _stream(x)), "far3d_ese_fused_nhwc")
    return out if out is not None else pooled


def groupnorm_nhwc(x, gamma, beta, groups=32, eps=1e-5, relu=True, out=None, scratch=None, pair=False):
    lib = _lib.require_device()
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    if pair:
        if x.dtype != torch.bfloat16 or C % 64:
            raise ValueError("groupnorm_nhwc: pair storage is bf16 with a multiple of 64 stored channels")
        C //= 2
    if out is None:
        out = torch.empty_like(x)
    if scratch is None:
        scratch = torch.empty(ese_scratch_floats(N, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.far3d_groupnorm_nhwc(_ptr(x), DT_BF16_PAIR if pair else _dt(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(scratch), N, H * W, C,
                                        groups, float(eps), 1 if relu else 0, _stream(x)), "far3d_groupnorm_nhwc")
    return out


def maxpool3x3s2_nhwc(x, out=None, pair=False):
    lib = _lib.require_device()
    _chk(x, "x", ndim=4)
    N, H, W, C = x.shape
    Cs = C
    if pair:
        if x.dtype != torch.bfloat16 or C % 64:
            raise ValueError("maxpool3x3s2_nhwc: pair storage is bf16 with a multiple of 64 stored channels")
        C //= 2
    Ho, Wo = -(-(H - 3) // 2) + 1, -(-(W - 3) // 2) + 1
    if (Ho - 1) * 2 >= H:
        Ho -= 1
    if (Wo - 1) * 2 >= W:
        Wo -= 1
    if out is None:
        out = torch.empty((N, Ho, Wo, Cs), dtype=x.dtype, device=x.device)
    ldy, ys = _nhwc_view(out, "out")
    _lib.check(lib.far3d_maxpool3x3s2_nhwc(_ptr(x), DT_BF16_PAIR if pair else _dt(x), _ptr(out), N, H, W, C, Ho, Wo, ldy, ys, _stream(x)),
               "far3d_maxpool3x3s2_nhwc")
    return out


# --------------------------------------------------------------------------------------------------
# front-end glue: stem im2col, 2D proposals, MLN apply
# --------------------------------------------------------------------------------------------------
def stem_im2col(img, out_dtype=torch.bfloat16, out=None, pair=False):
    """(N,3,H,W) f32 NCHW -> (N,Ho,Wo,32) NHWC im2col of the stride-2 3x3 stem conv (pair: (N,Ho,Wo,64) bf16 pair storage)."""
    lib = _lib.require_device()
    _chk(img, "img", torch.float32, 4)
    N, C, H, W = img.shape
    if C != 3:
        raise ValueError("stem_im2col expects 3 input channels")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, 64 if pair else 32), dtype=torch.bfloat16 if pair else out_dtype, device=img.device)
    _lib.check(lib.far3d_stem_im2col(_ptr(img), _ptr(out), DT_BF16_PAIR if pair else _dt(out), N, H, W, _stream(img)), "far3d_stem_im2col")
    return out


def stem_conv(img, pc, act="relu", out=None):
    """VoVNet's first stem convolution (3 -> 64, 3x3, stride 2, pad 1) straight from the (N,3,H,W) f32 NCHW image: bf16 NHWC output,
    bit-identical to conv2d_nhwc(stem_im2col(img), pc).  pc: the layer packed as a 1x1 conv over the 32-channel im2col rows (bf16)."""
    lib = _lib.require_device()
    _chk(img, "img", torch.float32, 4)
    N, C, H, W = img.shape
    if C != 3 or pc.Cout != 64 or pc.Cin != 32 or pc.KH != 1 or pc.w.dtype != torch.bfloat16:
        raise ValueError("stem_conv: 3-channel image and the 64 x 32 bf16 packing of the stem layer")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, 64), dtype=torch.bfloat16, device=img.device)
    ldy, ys = _nhwc_view(out, "out")
    _lib.check(lib.far3d_stem_conv(_ptr(img), _ptr(pc.w), _ptr(pc.bias) if pc.bias is not None else None, _ptr(out), N, H, W, ldy, ys,
                                   {None: 0, "relu": 1}[act], _stream(img)), "far3d_stem_conv")
    return out


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def proposal_select(cls_maps, reg_maps, strides, cap, thr=0.1, topk=False):
    """cls_maps[l] (N,h,w,ncls) f32, reg_maps[l] (N,h,w,>=5) f32 (dx,dy,logw,logh,obj).  Returns
    (weights (N,S) f32, sel_idx (N,cap) i32, sel_cnt (N) i32) -- all on device, no sync."""
    lib = _lib.require_device()
    L = len(cls_maps)
    for t in list(cls_maps) + list(reg_maps):
        _chk(t, "cls/reg map", torch.float32, 4)
    N = cls_maps[0].shape[0]
    hw = [(int(c.shape[1]), int(c.shape[2])) for c in cls_maps]
    S = sum(h * w for h, w in hw)
    dev = cls_maps[0].device
    sw = torch.empty((N, S), dtype=torch.float32, device=dev)
    wgt = torch.empty((N, S), dtype=torch.float32, device=dev)
    sel_idx = torch.empty((N, cap), dtype=torch.int32, device=dev)     # entries >= sel_cnt[n] are never read
    sel_cnt = torch.empty((N,), dtype=torch.int32, device=dev)
    hk, hp = _host_i32([list(x) for x in hw])
    sk, sp = _host_i32(list(strides))
    ca, ra = _ptr_array(cls_maps), _ptr_array(reg_maps)
    _lib.check(lib.far3d_proposal_select(ca, ra, cls_maps[0].shape[3], reg_maps[0].shape[3], N, L, hp, sp, _ptr(sw), _ptr(wgt),
                                         _ptr(sel_idx), _ptr(sel_cnt), cap, float(thr), 1 if topk else 0, _stream(wgt)),
               "far3d_proposal_select")
    return wgt, sel_idx, sel_cnt


def proposal_gather(reg_maps, strides, sel_idx, sel_cnt, weights, depth_logit, depth_stride, depth_cfg, img2lidar,
                    feat, pc_range, score_thr=0.1, out=None, rows_total=0, m_out=None, overflow_out=None):
    """Returns ref2d (N*cap,3), ctx (N*cap,C+1), box2d (N*cap,4), score (N*cap); rows >= sum(sel_cnt) are NOT written
    (callers slice [:M]).  out: optional (ref2d, ctx, box2d, score) buffers to write into.
    rows_total > 0: fixed-capacity mode -- the buffers have rows_total rows, rows past M = sum(sel_cnt) are zero-filled, m_out /
    overflow_out (int32 device tensors of one element) receive min(M, rows_total) and the dropped-proposal flag."""
    lib = _lib.require_device()
    L = len(reg_maps)
    N, cap = sel_idx.shape
    hw = [(int(c.shape[1]), int(c.shape[2])) for c in reg_maps]
    _chk(depth_logit, "depth_logit", torch.float32, 4)
    _chk(img2lidar, "img2lidar", torch.float32, 3)
    _chk(feat, "feat", ndim=3)
    C = feat.shape[2]
    dev = feat.device
    M = N * cap if not rows_total else int(rows_total)
    if out is not None:
        ref2d, ctx, box2d, score = out
    else:
        ref2d = torch.empty((M, 3), dtype=torch.float32, device=dev)
        ctx = torch.empty((M, C + 1), dtype=torch.float32, device=dev)
        box2d = torch.empty((M, 4), dtype=torch.float32, device=dev)
        score = torch.empty((M,), dtype=torch.float32, device=dev)
    if ref2d.stride(0) != 3 or ctx.stride(0) != C + 1:
        raise ValueError("proposal_gather: ref2d / ctx must be dense rows")
    if rows_total and min(ref2d.shape[0], ctx.shape[0], box2d.shape[0], score.shape[0]) < rows_total:
        raise ValueError("proposal_gather: the output buffers need rows_total=%d rows" % rows_total)
    for t in (m_out, overflow_out):
        if t is not None:
            _chk(t, "m_out / overflow_out", torch.int32)
    hk, hp = _host_i32([list(x) for x in hw])
    sk, sp = _host_i32(list(strides))
    pk, pp = _host_f32(list(pc_range))
    ra = _ptr_array(reg_maps)
    _, hd, wd, nd = depth_logit.shape
    _lib.check(lib.far3d_proposal_gather(ra, reg_maps[0].shape[3], N, L, hp, sp, _ptr(sel_idx), _ptr(sel_cnt), cap,
                                         _ptr(weights), _ptr(depth_logit), hd, wd, nd, int(depth_stride),
                                         float(depth_cfg["depth_min"]), float(depth_cfg["depth_max"]),
                                         int(depth_cfg["num_depth_bins"]), _ptr(img2lidar), _ptr(feat), _dt(feat), C, pp,
                                         float(score_thr), _ptr(ref2d), _ptr(ctx), _ptr(box2d), _ptr(score), int(rows_total),
                                         _ptr(m_out) if m_out is not None else None,
                                         _ptr(overflow_out) if overflow_out is not None else None, _stream(feat)),
               "far3d_proposal_gather")
    return ref2d, ctx, box2d, score


def compact_rows(src, counts, dst, m_out, overflow_out):
    """src (nblocks, rows_per_block, D) f32 with counts[b] valid rows in block b -> dst (dst_rows, D): the valid rows in block order,
    the rest zero; m_out = min(sum counts, dst_rows); overflow_out |= (sum counts > dst_rows).  One launch, no sync."""
    lib = _lib.require_device()
    _chk(src, "src", torch.float32, 3)
    _chk(counts, "counts", torch.int32, 1)
    _chk(dst, "dst", torch.float32, 2)
    _chk(m_out, "m_out", torch.int32)
    _chk(overflow_out, "overflow_out", torch.int32)
    nb, rpb, D = src.shape
    if counts.numel() != nb or dst.shape[1] != D:
        raise ValueError("compact_rows: inconsistent shapes")
    _lib.check(lib.far3d_compact_rows(_ptr(src), _ptr(counts), nb, rpb, D, _ptr(dst), dst.shape[0], _ptr(m_out), _ptr(overflow_out),
                                      _stream(src)), "far3d_compact_rows")
    return dst


def row_affine_ln(x, gamma, beta, add=None, do_ln=True, eps=1e-5, out=None):
    """out[r] = gamma[r] * LN(x[r]) + beta[r] (+ add[r]); gamma/beta/add may be a single row (broadcast).  C = 256."""
    lib = _lib.require_device()
    rows, C = x.shape
    bc = lambda t: 0 if (t.dim() == 1 or t.shape[0] == 1) else t.stride(0)
    if out is None:
        out = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    if bc(gamma) != bc(beta):
        raise ValueError("row_affine_ln: gamma and beta must broadcast alike")
    _lib.check(lib.far3d_row_affine_ln(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(add) if add is not None else None, _ptr(out),
                                       rows, C, x.stride(0), bc(gamma), bc(add) if add is not None else 0, out.stride(0),
                                       float(eps), 1 if do_ln else 0, _stream(x)), "far3d_row_affine_ln")
    return out



def camera_sorted_order(ref, lidar2img, pc_range, pad_hw, spatial=True):
    """Query order for aggregate_forward's `perm`: queries sorted by the camera whose image centre their reference point
    projects closest to, then (spatial=True) by image tile (8 x 8 coarse cells, row-major) inside that camera, so that
    consecutive workgroups of one XCD touch neighbouring pixels.  Pure scheduling hint -- results do not depend on it."""
    pc = pc_range if isinstance(pc_range, torch.Tensor) else torch.as_tensor(pc_range, dtype=torch.float32, device=ref.device)
    pts = ref * (pc[3:6] - pc[0:3]) + pc[0:3]
    p = torch.einsum("nij,aj->nai", lidar2img[:, :3, :3], pts) + lidar2img[:, :3, 3][:, None, :]       # (N,A,3)
    z = p[..., 2]
    u = p[..., 0] / z.clamp(min=1e-5) / pad_hw[1] - 0.5
    v = p[..., 1] / z.clamp(min=1e-5) / pad_hw[0] - 0.5
    cost = torch.where(z > 1e-5, u * u + v * v, torch.full_like(z, 1e9))
    cam = cost.argmin(dim=0)
    key = cam
    if spatial:
        uu = torch.gather(u, 0, cam[None])[0]
        vv = torch.gather(v, 0, cam[None])[0]
        ub = ((uu + 0.5).clamp(0, 0.999) * 8).long()
        vb = ((vv + 0.5).clamp(0, 0.999) * 8).long()
        key = (cam * 8 + vb) * 8 + ub
    return torch.sort(key, stable=True).indices.to(torch.int32).contiguous()


# --------------------------------------------------------------------------------------------------
# fused FarHead bookkeeping (glue.hip)
# --------------------------------------------------------------------------------------------------
def posemb3d(pos, dim_t128):
    lib = _lib.require_device()
    _chk(pos, "pos", torch.float32, 2)
    out = torch.empty((pos.shape[0], 384), dtype=torch.float32, device=pos.device)
    _lib.check(lib.far3d_posemb3d(_ptr(pos), _ptr(dim_t128), _ptr(out), pos.shape[0], _stream(pos)), "far3d_posemb3d")
    return out


def memory_prepare(state, ego_pose_inv, timestamp, pseudo_ref, dim_t256, prev_exists, fresh, pc_range, num_propagated,
                   temp_ref_out=None):
    """state: dict emb (1,L,E) ref (1,L,3) ts (1,L,1) f64 pose (1,L,4,4) velo (1,L,2).  Returns (m dict, temp_ref, nerf, tpos)."""
    lib = _lib.require_device()
    L, E = state["emb"].shape[1], state["emb"].shape[2]
    dev = state["emb"].device
    m = {k: torch.empty_like(v) for k, v in state.items()}
    temp_ref = temp_ref_out if temp_ref_out is not None else torch.empty((L, 3), dtype=torch.float32, device=dev)
    if tuple(temp_ref.shape) != (L, 3) or not temp_ref.is_contiguous():
        raise ValueError("memory_prepare: temp_ref_out must be a contiguous (L,3) tensor")
    nerf = torch.empty((L, 180), dtype=torch.float32, device=dev)
    tpos = torch.empty((L, 256), dtype=torch.float32, device=dev)
    pk, pp = _host_f32(list(pc_range))
    _lib.check(lib.far3d_memory_prepare(_ptr(state["emb"]), _ptr(state["ref"]), _ptr(state["ts"]), _ptr(state["pose"]), _ptr(state["velo"]),
                                        _ptr(ego_pose_inv), _ptr(timestamp), _ptr(pseudo_ref) if pseudo_ref is not None else None,
                                        _ptr(dim_t256), float(prev_exists), 1 if fresh else 0, pp, L, E, int(num_propagated),
                                        _ptr(m["emb"]), _ptr(m["ref"]), _ptr(m["ts"]), _ptr(m["pose"]), _ptr(m["velo"]),
                                        _ptr(temp_ref), _ptr(nerf), _ptr(tpos), _stream(temp_ref)), "far3d_memory_prepare")
    return m, temp_ref, nerf, tpos


def head_finalize(reg, ref, cls_all, pc_range, layers, num_classes, hole=None):
    """reg (layers*A, code) -> box (layers*A, code) with metres in [:3]; score (A,) = max-class sigmoid of the last layer.
    cls_all: the contiguous (layers, ..., A, num_classes) logits.  hole: rows without a query get score -inf and -inf logits in
    every layer, written into cls_all in place (see _hole)."""
    lib = _lib.require_device()
    A, code = ref.shape[0], reg.shape[1]
    _chk(cls_all, "cls_all", torch.float32)
    if cls_all.numel() != layers * A * num_classes:
        raise ValueError("head_finalize: cls_all must hold layers * A * num_classes logits")
    box = torch.empty_like(reg)
    score = torch.empty((A,), dtype=torch.float32, device=reg.device)
    pk, pp = _host_f32(list(pc_range))
    _lib.check(lib.far3d_head_finalize(_ptr(reg), _ptr(ref), _ptr(cls_all), _ptr(box), _ptr(score), layers, A, code, num_classes, pp,
                                       *_hole(hole), _stream(reg)), "far3d_head_finalize")
    return box, score


def memory_post_update(m, topk_idx, dec_last, box_last, ego_pose, timestamp, state):
    lib = _lib.require_device()
    L, E = state["emb"].shape[1], state["emb"].shape[2]
    _chk(topk_idx, "topk_idx", torch.int64, 1)
    _lib.check(lib.far3d_memory_post_update(_ptr(m["emb"]), _ptr(m["ref"]), _ptr(m["ts"]), _ptr(m["pose"]), _ptr(m["velo"]), _ptr(topk_idx),
                                            _ptr(dec_last), _ptr(box_last), _ptr(ego_pose), _ptr(timestamp), L, E, topk_idx.numel(),
                                            box_last.shape[1], _ptr(state["emb"]), _ptr(state["ref"]), _ptr(state["ts"]),
                                            _ptr(state["pose"]), _ptr(state["velo"]), _stream(dec_last)), "far3d_memory_post_update")


def add_cast(a, b, sum_dtype, a_dtype=None, out_sum=None, out_a=None):
    """Returns (a+b as sum_dtype, a as a_dtype or None) in one pass; a,b (rows,C) f32 contiguous.  out_sum / out_a: optional
    (rows,C) destinations with unit inner stride and any row stride (e.g. halves of one (rows,2C) operand buffer)."""
    lib = _lib.require_device()
    _chk(a, "a", torch.float32, 2)
    _chk(b, "b", torch.float32, 2)
    rows, C = a.shape
    osum = out_sum if out_sum is not None else torch.empty(a.shape, dtype=sum_dtype, device=a.device)
    oa = out_a if out_a is not None else (torch.empty(a.shape, dtype=a_dtype, device=a.device) if a_dtype is not None else None)
    for t in (osum, oa):
        if t is not None and (tuple(t.shape) != (rows, C) or t.stride(1) != 1):
            raise ValueError("add_cast: outputs must be (rows,C) with unit inner stride")
    _lib.check(lib.far3d_add_cast(_ptr(a), _ptr(b), _ptr(osum), _dt(osum), _ptr(oa) if oa is not None else None,
                                  _dt(oa) if oa is not None else 0, rows, C, osum.stride(0), oa.stride(0) if oa is not None else 0,
                                  _stream(a)), "far3d_add_cast")
    return osum, oa


def aggregation_order(ref, lidar2img, pc_range, pad_hw, out=None, hole=None, rows=None, Vc=None, tables_out=None, split=None,
                      sorted_operands=None):
    """Query order for aggregate_forward's `perm` (camera, then 8x8 image cell), one single-workgroup launch.  Groups the
    same way as camera_sorted_order(spatial=True); the order inside a cell is arbitrary (scheduling only).  hole: rows without a
    query are entered as ~a so that aggregate_forward writes zero rows for them (see _hole).  rows=(a0, a1): order only the rows
    [a0, a1) of ref; the entries are absolute row indices (pass the full-size buffers and this perm to aggregate_forward).
    Vc (layers, N, J): the same launch also computes agg_tables(Vc) (into tables_out when given); returns (perm, tables) then.
    sorted_operands: True, or a preallocated (inv (A) int32, qbase (A, 4) f32) pair -- the same launch also writes the operands of
    aggregate_forward's sorted mode (include/far3d_hip.h); they are appended to the return value as (inv, qbase)."""
    lib = _lib.require_device()
    _chk(ref, "ref", torch.float32, 2)
    a0, a1 = (0, ref.shape[0]) if rows is None else rows
    A = a1 - a0
    extra = split.extra if split is not None else 0        # AggSplit: A main entries + `extra` sibling entries (aggregate_forward variant 9)
    perm = out if out is not None else torch.empty((A + extra,), dtype=torch.int32, device=ref.device)
    if perm.numel() != A + extra:
        raise ValueError("aggregation_order: perm needs %d entries (A = %d + %d sibling slots), got %d" % (A + extra, A, extra, perm.numel()))
    pk, pp = _host_f32(list(pc_range))
    tab, layers, J = None, 0, 0
    if Vc is not None:
        _chk(Vc, "Vc", torch.float32, 3)
        layers, Nv, J = Vc.shape
        if Nv != lidar2img.shape[0]:
            raise ValueError("aggregation_order: Vc has %d cameras, lidar2img %d" % (Nv, lidar2img.shape[0]))
        tab = tables_out if tables_out is not None else torch.empty((layers, 2 + Nv, J), dtype=torch.float32, device=ref.device)
    inv = qbase = None
    if sorted_operands is not None and sorted_operands is not False:
        if split is not None:
            raise ValueError("aggregation_order: sorted operands and sibling workgroups (split) do not combine")
        if sorted_operands is True:
            inv = torch.empty((A,), dtype=torch.int32, device=ref.device)
            qbase = torch.empty((A, 4), dtype=torch.float32, device=ref.device)
        else:
            inv, qbase = sorted_operands
        _chk(inv, "inv", torch.int32, 1)
        _chk(qbase, "qbase", torch.float32, 2)
        if inv.numel() != A or tuple(qbase.shape) != (A, 4):
            raise ValueError("aggregation_order: inv needs A = %d entries and qbase the shape (A, 4)" % A)
    _lib.check(lib.far3d_agg_order(_ptr(ref), _ptr(lidar2img), _ptr(perm), A, lidar2img.shape[0], pp, float(pad_hw[0]), float(pad_hw[1]),
                                   *_hole(hole), int(a0), _ptr(Vc) if Vc is not None else None, _ptr(tab) if tab is not None else None,
                                   layers, J, int(extra), _ptr(inv) if inv is not None else None, _ptr(qbase) if qbase is not None else None,
                                   _stream(ref)), "far3d_agg_order")
    res = (perm,) + ((tab,) if Vc is not None else ()) + (((inv, qbase),) if inv is not None else ())
    return res if len(res) > 1 else perm


def topk(vals, K, with_values=False):
    """Descending top-K indices (int64) of a 1-D f32 device tensor (n <= 40960, K <= 1024), ties -> lower index."""
    lib = _lib.require_device()
    _chk(vals, "vals", torch.float32, 1)
    idx = torch.empty((K,), dtype=torch.int64, device=vals.device)
    out = torch.empty((K,), dtype=torch.float32, device=vals.device) if with_values else None
    _lib.check(lib.far3d_topk(_ptr(vals), vals.numel(), int(K), _ptr(idx), _ptr(out) if out is not None else None, _stream(vals)),
               "far3d_topk")
    return (idx, out) if with_values else idx


DECODE_CHUNK = 40960     # logits one workgroup ranks (csrc/post.hip)


def decode_ws_bytes(n, K):
    """FAR3D_DECODE_WS_BYTES: workspace far3d_decode_topk needs for n = A * num_classes logits (0: single-launch path)."""
    return 0 if n <= DECODE_CHUNK else -(-n // DECODE_CHUNK) * K * 8


def decode_topk(cls_last, box_last, K, post_center_range, workspace=None, mem_scores=None, mem_K=None):
    """NMS-free decode of the last layer: returns dict(boxes_3d (K,code-1), scores_3d (K), labels_3d (K) i64, keep (K) bool).
    workspace: optional uint8 device buffer of decode_ws_bytes(A * ncls, K) bytes (allocated here when needed and not given).
    mem_scores (n) f32 + mem_K: the same launch also ranks them (far3d_decode_topk_mem: topk(mem_scores, mem_K) in a second workgroup);
    returns (dict, idx (mem_K) int64) then."""
    lib = _lib.require_device()
    _chk(cls_last, "cls_last", torch.float32, 2)
    _chk(box_last, "box_last", torch.float32, 2)
    A, ncls = cls_last.shape
    code = box_last.shape[1]
    dev = cls_last.device
    boxes = torch.empty((K, code - 1), dtype=torch.float32, device=dev)
    scores = torch.empty((K,), dtype=torch.float32, device=dev)
    labels = torch.empty((K,), dtype=torch.int64, device=dev)
    keep = torch.empty((K,), dtype=torch.bool, device=dev)
    rk, rp = _host_f32(list(post_center_range))
    need = decode_ws_bytes(A * ncls, int(K))
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    res = dict(boxes_3d=boxes, scores_3d=scores, labels_3d=labels, keep=keep)
    if mem_scores is not None:
        _chk(mem_scores, "mem_scores", torch.float32, 1)
        idx = torch.empty((int(mem_K),), dtype=torch.int64, device=dev)
        _lib.check(lib.far3d_decode_topk_mem(_ptr(cls_last), _ptr(box_last), A, ncls, code, int(K), rp, _ptr(boxes), _ptr(scores), _ptr(labels),
                                             _ptr(keep), _ptr(workspace) if need else None, need, _ptr(mem_scores), mem_scores.numel(), int(mem_K),
                                             _ptr(idx), _stream(cls_last)), "far3d_decode_topk_mem")
        return res, idx
    _lib.check(lib.far3d_decode_topk(_ptr(cls_last), _ptr(box_last), A, ncls, code, int(K), rp, _ptr(boxes), _ptr(scores), _ptr(labels),
                                     _ptr(keep), _ptr(workspace) if need else None, need, _stream(cls_last)), "far3d_decode_topk")
    return res


def camera_prep(lidar2img, intrinsics=None, extrinsics=None):
    """lidar2img / intrinsics / extrinsics (N,4,4) f32 -> (img2lidar (N,4,4) = inverse(lidar2img), c14 (N,14) | None)."""
    lib = _lib.require_device()
    _chk(lidar2img, "lidar2img", torch.float32, 3)
    N, dev = lidar2img.shape[0], lidar2img.device
    i2l = torch.empty((N, 4, 4), dtype=torch.float32, device=dev)
    c14 = None
    if intrinsics is not None:
        _chk(intrinsics, "intrinsics", torch.float32, 3)
        _chk(extrinsics, "extrinsics", torch.float32, 3)
        c14 = torch.empty((N, 14), dtype=torch.float32, device=dev)
    _lib.check(lib.far3d_camera_prep(_ptr(lidar2img), _ptr(intrinsics) if c14 is not None else None,
                                     _ptr(extrinsics) if c14 is not None else None, _ptr(i2l),
                                     _ptr(c14) if c14 is not None else None, N, _stream(lidar2img)), "far3d_camera_prep")
    return i2l, c14


def nan_to_num_(x, bf16_copy=False):
    """In-place torch.nan_to_num on a contiguous f32 tensor; optionally also returns a bf16 copy of the result."""
    lib = _lib.require_device()
    _chk(x, "x", torch.float32)
    xb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if bf16_copy else None
    _lib.check(lib.far3d_nan_to_num(_ptr(x), _ptr(xb) if xb is not None else None, x.numel(), _stream(x)), "far3d_nan_to_num")
    return xb


def cam_embed_chain(l2i, packed, eps=1e-5):
    """Camera term of the aggregation logits for all decoder layers in one launch.  l2i: lidar2img (N,4,4) f32 (read in
    place) or its (N,12) top-three-rows flattening; packed: dict of stacked, transposed fp32 weights (see
    pack_cam_embed_chain).  Returns (L,N,J) f32."""
    lib = _lib.require_device()
    N = l2i.shape[0]
    l2i = _chk(l2i.reshape(N, -1), "l2i", torch.float32, 2)
    L, E, J = packed["w3t"].shape
    Hd = packed["w0t"].shape[2]
    if E != 256 or l2i.shape[1] not in (12, 16) or tuple(packed["w2t"].shape) != (L, Hd, 256) or Hd > 256:
        raise ValueError("cam_embed_chain: expects 12 -> Hd (<=256) -> 256 -> J weights, got %s" % {k: tuple(v.shape) for k, v in packed.items()})
    out = torch.empty((L, N, J), dtype=torch.float32, device=l2i.device)
    _lib.check(lib.far3d_cam_embed_chain(_ptr(l2i), _ptr(packed["w0t"]), _ptr(packed["b0"]), _ptr(packed["w2t"]),
                                         _ptr(packed["b2"]), _ptr(packed["ln_g"]), _ptr(packed["ln_b"]), _ptr(packed["w3t"]),
                                         _ptr(packed["b3"]), _ptr(out), N, L, J, Hd, eps, l2i.shape[1], _stream(l2i)), "far3d_cam_embed_chain")
    return out


def pack_cam_embed_chain(layers, device):
    """layers: list of (W0 (Hd,12), b0, W2 (256,Hd), b2, ln_g, ln_b, W3 (J,256), b3) per decoder layer (state-dict tensors)."""
    f = lambda ts: torch.stack([t.detach().float() for t in ts]).contiguous().to(device)
    return dict(w0t=f([l[0].t() for l in layers]), b0=f([l[1] for l in layers]), w2t=f([l[2].t() for l in layers]),
                b2=f([l[3] for l in layers]), ln_g=f([l[4] for l in layers]), ln_b=f([l[5] for l in layers]),
                w3t=f([l[6].t() for l in layers]), b3=f([l[7] for l in layers]))
