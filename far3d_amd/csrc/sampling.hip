// Deformable sampling kernels for the Far3D cross-attention.
//
//  * far3d_msda_forward      -- drop-in for mmcv's ms_deform_attn_forward as called at
//                               reference detr3d_transformer.py:561-563 (operator boundary).
//  * far3d_aggregate_forward -- the fused perspective-aware aggregation: everything in
//                               DeformableFeatureAggregationCuda.feature_sampling + _get_weights'
//                               softmax (detr3d_transformer.py:535-569) in ONE launch:
//                               projection of the 13 key points into the 7 cameras, visibility
//                               culling, 364-way softmax per group, bilinear gather over 4 levels,
//                               weighting, and the cross-camera sum.  No (B*N,A,8,4,13,2) replicated
//                               location tensor, no (B*N,A,8,52) permuted weight tensor, no
//                               (B*N,A,256) per-camera output ever reaches HBM.
//
// Layout facts the kernels are built around (gfx950, wave = 64 lanes):
//   value maps are token-major (camera, token, 256 channels); one token row is 1 KiB fp32 / 512 B
//   bf16, so ONE wave-wide 16 B(8 B)-per-lane load fetches a whole token row for all 8 groups:
//   lane = group*8 + chunk, 4 channels per lane.  All 8 groups sample the same pixel (the reference
//   replicates locations over groups, :555), so a bilinear sample is 4 fully-coalesced row loads.
#include "common.hpp"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------
// Generic MSDA (mmcv contract).  value (bs,S,H,Dh) TV; shapes (L,2) i64 (h,w); lsi (L) i64;
// loc (bs,Q,H,L,P,2) f32 in [0,1] (x,y); attn (bs,Q,H,L,P) f32; out (bs,Q,H*Dh) f32.
// One thread owns 4 consecutive channels of one (b,q,h): neighbouring lanes read neighbouring
// 16-B pieces of the same token row -> coalesced.
// ------------------------------------------------------------------------------------------
template <typename TV>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const TV* __restrict__ value,
                                                       const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi,
                                                       const float* __restrict__ loc,
                                                       const float* __restrict__ attn,
                                                       float* __restrict__ out, int bs, int S, int H,
                                                       int Dh, int L, int Q, int P) {
  const int c4n = Dh >> 2;
  const long total = (long)bs * Q * H * c4n;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    long t = idx / c4n;
    const int h = (int)(t % H);
    t /= H;
    const int q = (int)(t % Q);
    const int b = (int)(t / Q);
    const long bqh = ((long)b * Q + q) * H + h;
    const float* locp = loc + bqh * L * P * 2;
    const float* wp = attn + bqh * L * P;
    const TV* vb = value + (long)b * S * H * Dh + h * Dh + c4 * 4;
    const long row = (long)H * Dh;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      const TV* vl = vb + lsi[l] * row;
      for (int p = 0; p < P; ++p) {
        const float lx = locp[(l * P + p) * 2], ly = locp[(l * P + p) * 2 + 1];
        const float aw = wp[l * P + p];
        const float h_im = ly * Hl - 0.5f, w_im = lx * Wl - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < Hl && w_im < Wl) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const float lh = h_im - h_low, lw = w_im - w_low;
          const float hh = 1.f - lh, hw = 1.f - lw;
          float4 v1 = make_float4(0, 0, 0, 0), v2 = v1, v3 = v1, v4 = v1;
          if (h_low >= 0 && w_low >= 0) v1 = load4(vl + ((long)h_low * Wl + w_low) * row);
          if (h_low >= 0 && w_high <= Wl - 1) v2 = load4(vl + ((long)h_low * Wl + w_high) * row);
          if (h_high <= Hl - 1 && w_low >= 0) v3 = load4(vl + ((long)h_high * Wl + w_low) * row);
          if (h_high <= Hl - 1 && w_high <= Wl - 1) v4 = load4(vl + ((long)h_high * Wl + w_high) * row);
          const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          acc.x += aw * (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x);
          acc.y += aw * (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y);
          acc.z += aw * (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z);
          acc.w += aw * (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w);
        }
      }
    }
    *reinterpret_cast<float4*>(out + bqh * Dh + c4 * 4) = acc;
  }
}

extern "C" int far3d_msda_forward(const void* value, int value_dtype, const int64_t* spatial_shapes,
                                  const int64_t* level_start_index, const float* sampling_loc,
                                  const float* attn_weight, float* out, int bs, int S, int H, int Dh,
                                  int L, int Q, int P, void* stream) {
  FAR3D_CHECK_ARG(bs >= 0 && S > 0 && H > 0 && Dh > 0 && L > 0 && Q >= 0 && P > 0,
                  "far3d_msda_forward: bad sizes bs=%d S=%d H=%d Dh=%d L=%d Q=%d P=%d", bs, S, H, Dh, L, Q, P);
  FAR3D_CHECK_ARG((Dh & 3) == 0, "far3d_msda_forward: head dim %d must be a multiple of 4", Dh);
  FAR3D_CHECK_ARG(value_dtype == FAR3D_DT_F32 || value_dtype == FAR3D_DT_BF16,
                  "far3d_msda_forward: unsupported value dtype %d", value_dtype);
  const long total = (long)bs * Q * H * (Dh / 4);
  if (total == 0) return FAR3D_OK;  // empty query set: nothing to write (pointers may be null)
  FAR3D_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
                  "far3d_msda_forward: null pointer argument");
  long blocks = (total + 255) / 256;
  if (blocks > 256L * 16) blocks = 256L * 16;
  hipStream_t st = (hipStream_t)stream;
  if (value_dtype == FAR3D_DT_F32)
    hipLaunchKernelGGL(msda_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, out, bs, S, H, Dh, L, Q, P);
  else
    hipLaunchKernelGGL(msda_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, out, bs, S, H, Dh, L, Q, P);
  FAR3D_CHECK_LAUNCH("far3d_msda_forward");
  return FAR3D_OK;
}

// ------------------------------------------------------------------------------------------
// Fused perspective-aware aggregation.  One 256-thread workgroup (4 waves) per query.
//   phase 1: 91 (camera, point) projections -> LDS (u,v) + per-level in-range bit mask
//   phase 2: 364-way softmax for the 8 groups (logit = U[a] + Vc[n]; see aggregate.py) -> LDS
//   phase 3: each wave walks the visible (camera,point) pairs; per level 4 coalesced row loads
//   phase 4: 4-wave LDS reduction, 1 KiB coalesced store
// ------------------------------------------------------------------------------------------
#define AGG_MAX_L 4
#define AGG_MAX_NLP 384  // 12 softmax items per thread
#define AGG_DEDUP_RCAP 88   // de-duplicated gather for queries with at most this many visible samples (4*RCAP taps) ...
#define AGG_DEDUP_LCAP 192  // ... and at most this many distinct token rows
struct AggParams {
  int A, N, S, P, L;
  int H[AGG_MAX_L], W[AGG_MAX_L], start[AGG_MAX_L];
  float pc_lo[3], pc_span[3];
  float pad_w, pad_h;
  int q_per_xcd;
  int out_dt;
};

template <typename TV, int ABL = 0>   // ABL: timing ablations (1: no gather, 2: no softmax, 3: neither) -- never shipped results
__global__ __launch_bounds__(256) void aggregate_fwd_kernel(const TV* __restrict__ feat,
                                                            const float* __restrict__ ref,
                                                            const float* __restrict__ offs,
                                                            const float* __restrict__ l2i,
                                                            const float* __restrict__ U,
                                                            const float* __restrict__ Vc,
                                                            const int* __restrict__ perm,
                                                            float* __restrict__ out, AggParams prm) {
  // XCD-aware query mapping: workgroup b lands on XCD b%8 (observed dispatch order); give each XCD a
  // contiguous range of the (optionally camera-sorted) query order so that one XCD's L2 serves 1-2 cameras.  Speed only.
  int a = (blockIdx.x & 7) * prm.q_per_xcd + (blockIdx.x >> 3);
  if (a >= prm.A) return;
  if (perm) a = perm[a];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NP = prm.N * prm.P;
  const int NLP = NP * prm.L;
  const int J = prm.L * prm.P;            // logits per camera per group
  float* w_s = reinterpret_cast<float*>(smem);                 // [NLP][8]
  float* red_s = w_s + NLP * 8;                                 // [4][256]
  float* uv_s = red_s + 1024;                                   // [NP][2]
  float* stat_s = uv_s + 2 * NP;                                // [2][4][8]
  int* mask_s = reinterpret_cast<int*>(stat_s + 64);            // [NP]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;

  // ---- phase 1: projection (detr3d_transformer.py:524-525,547-552)
  if (t < NP) {
    const int n = t / prm.P, p = t - n * prm.P;
    float k[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      k[d] = (ref[a * 3 + d] * prm.pc_span[d] + prm.pc_lo[d]) + offs[(long)a * prm.P * 3 + p * 3 + d];
    const float* m = l2i + n * 16;
    const float x = m[0] * k[0] + m[1] * k[1] + m[2] * k[2] + m[3];
    const float y = m[4] * k[0] + m[5] * k[1] + m[6] * k[2] + m[7];
    const float z = m[8] * k[0] + m[9] * k[1] + m[10] * k[2] + m[11];
    const float zc = fmaxf(z, 1e-5f);
    const float u = (x / zc) / prm.pad_w, v = (y / zc) / prm.pad_h;
    int mask = 0;
#pragma unroll
    for (int l = 0; l < AGG_MAX_L; ++l) {
      if (l < prm.L) {
        const float h_im = v * prm.H[l] - 0.5f, w_im = u * prm.W[l] - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < prm.H[l] && w_im < prm.W[l]) mask |= 1 << l;
      }
    }
    uv_s[2 * t] = u;
    uv_s[2 * t + 1] = v;
    mask_s[t] = mask;
  }

  // ---- phase 2: softmax over cams x levels x points per group (detr3d_transformer.py:539-540)
  if constexpr ((ABL & 2) == 0) {
    const int g = t & 7, r = t >> 3;
    float lg[AGG_MAX_NLP / 32];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < AGG_MAX_NLP / 32; ++i) {
      const int idx = r + 32 * i;
      lg[i] = -INFINITY;
      if (idx < NLP) {
        const int n = idx / J, j = idx - n * J;
        lg[i] = U[(long)a * J * 8 + j * 8 + g] + Vc[n * J * 8 + j * 8 + g];
      }
      mx = fmaxf(mx, lg[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (lane < 8) stat_s[wv * 8 + lane] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat_s[g], stat_s[8 + g]), fmaxf(stat_s[16 + g], stat_s[24 + g]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < AGG_MAX_NLP / 32; ++i) {
      lg[i] = (r + 32 * i < NLP) ? expf(lg[i] - mx) : 0.f;
      sum += lg[i];
    }
    sum += __shfl_xor(sum, 8);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (lane < 8) stat_s[32 + wv * 8 + lane] = sum;
    __syncthreads();
    sum = (stat_s[32 + g] + stat_s[40 + g]) + (stat_s[48 + g] + stat_s[56 + g]);
#pragma unroll
    for (int i = 0; i < AGG_MAX_NLP / 32; ++i) {
      const int idx = r + 32 * i;
      if (idx < NLP) w_s[idx * 8 + g] = lg[i] / sum;
    }
  }
  __syncthreads();

  // ---- phase 3: gather (mmcv ms_deform_attn bilinear semantics; zeros outside)
  const int g = lane >> 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const TV* fbase = feat + lane * 4;
  for (int pair = wv; pair < ((ABL & 1) ? 0 : NP); pair += 4) {
    const int mask = mask_s[pair];
    if (mask == 0) continue;  // wave-uniform
    const int n = pair / prm.P, p = pair - n * prm.P;
    const float u = uv_s[2 * pair], v = uv_s[2 * pair + 1];
#pragma unroll
    for (int l = 0; l < AGG_MAX_L; ++l) {
      if (l >= prm.L || !((mask >> l) & 1)) continue;
      const int Hl = prm.H[l], Wl = prm.W[l];
      const float h_im = v * Hl - 0.5f, w_im = u * Wl - 0.5f;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const float lh = h_im - h_low, lw = w_im - w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= Hl - 1;
      const bool l_ok = w_low >= 0, r_ok = w_low + 1 <= Wl - 1;
      const int hl_c = t_ok ? h_low : 0, hh_c = b_ok ? h_low + 1 : Hl - 1;
      const int wl_c = l_ok ? w_low : 0, wh_c = r_ok ? w_low + 1 : Wl - 1;
      const TV* lvl = fbase + ((long)n * prm.S + prm.start[l]) * 256;
      float4 v1 = load4(lvl + (long)(hl_c * Wl + wl_c) * 256);
      float4 v2 = load4(lvl + (long)(hl_c * Wl + wh_c) * 256);
      float4 v3 = load4(lvl + (long)(hh_c * Wl + wl_c) * 256);
      float4 v4 = load4(lvl + (long)(hh_c * Wl + wh_c) * 256);
      const float aw = w_s[((n * prm.L + l) * prm.P + p) * 8 + g];
      const float w1 = (t_ok && l_ok) ? hh * hw : 0.f;
      const float w2 = (t_ok && r_ok) ? hh * lw : 0.f;
      const float w3 = (b_ok && l_ok) ? lh * hw : 0.f;
      const float w4 = (b_ok && r_ok) ? lh * lw : 0.f;
      if (!(t_ok && l_ok)) v1 = make_float4(0, 0, 0, 0);
      if (!(t_ok && r_ok)) v2 = make_float4(0, 0, 0, 0);
      if (!(b_ok && l_ok)) v3 = make_float4(0, 0, 0, 0);
      if (!(b_ok && r_ok)) v4 = make_float4(0, 0, 0, 0);
      acc.x += aw * (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x);
      acc.y += aw * (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y);
      acc.z += aw * (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z);
      acc.w += aw * (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w);
    }
  }

  // ---- phase 4: cross-wave (= cross-camera/point) reduction, coalesced store
  *reinterpret_cast<float4*>(red_s + wv * 256 + lane * 4) = acc;
  __syncthreads();
  out[(long)a * 256 + t] = (red_s[t] + red_s[256 + t]) + (red_s[512 + t] + red_s[768 + t]);
}


// ------------------------------------------------------------------------------------------
// v3 of the fused aggregation (the shipped kernel).  PMC counters on v1 showed the kernel VALU-bound (2.5k VALU
// instructions per wave, L2 hit 60 %, HBM fetch == the 46 MB of value maps): every lane recomputed wave-uniform sample
// geometry.  v3 computes each visible (camera, point, level) sample ONCE (one thread per camera-point in phase 1), stores a
// 32-byte record {clamped top-left offset, dx, dy, weight row, 4 bilinear weights} in a compacted LDS list (deterministic
// block scan), and the gather loop is reduced to: broadcast record read, 4 coalesced row loads, 16 FMAs -- NB samples in
// flight per wave.  Softmax uses the hardware exp2 path and incremental (camera, logit) indices.
// ------------------------------------------------------------------------------------------
template <typename TV, int NB, int ABL = 0, int RP = 0>   // RP: bf16 row-pair gather (16-byte loads, both bilinear columns per instruction)
__global__ __launch_bounds__(256) void aggregate_v3_kernel(const TV* __restrict__ feat, const float* __restrict__ ref,
                                                           const float* __restrict__ offs, const float* __restrict__ l2i,
                                                           const float* __restrict__ U, const float* __restrict__ Vc,
                                                           const int* __restrict__ perm, void* __restrict__ out, AggParams prm) {
  int a = (blockIdx.x & 7) * prm.q_per_xcd + (blockIdx.x >> 3);
  if (a >= prm.A) return;
  if (perm) a = perm[a];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NP = prm.N * prm.P, NLP = NP * prm.L, J = prm.L * prm.P;
  float* w_s = reinterpret_cast<float*>(smem);                 // [NLP][8] softmax weights; reused as [4][256] reduction buffer
  float* stat_s = w_s + (NLP * 8 > 1024 ? NLP * 8 : 1024);     // [64]
  int* scan_s = reinterpret_cast<int*>(stat_s + 64);           // [16]
  int4* rec_s = reinterpret_cast<int4*>(scan_s + 16);          // [NLP][2] sample records (compacted: only [0, nsamp) are live;
                                                               //  the de-duplicating gather keeps its tables behind record RCAP)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;

  // ---- phase 1: one thread per (camera, point): projection, per-level geometry (detr3d_transformer.py:524-525,547-552)
  int mask = 0;
  float u = 0.f, v = 0.f;
  int n1 = 0, p1 = 0;
  if (t < NP) {
    n1 = t / prm.P; p1 = t - n1 * prm.P;
    float k[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      k[d] = (ref[a * 3 + d] * prm.pc_span[d] + prm.pc_lo[d]) + offs[(long)a * prm.P * 3 + p1 * 3 + d];
    const float* m = l2i + n1 * 16;
    const float x = m[0] * k[0] + m[1] * k[1] + m[2] * k[2] + m[3];
    const float y = m[4] * k[0] + m[5] * k[1] + m[6] * k[2] + m[7];
    const float z = m[8] * k[0] + m[9] * k[1] + m[10] * k[2] + m[11];
    const float zc = fmaxf(z, 1e-5f);
    u = (x / zc) / prm.pad_w; v = (y / zc) / prm.pad_h;
#pragma unroll
    for (int l = 0; l < AGG_MAX_L; ++l) {
      if (l < prm.L) {
        const float h_im = v * prm.H[l] - 0.5f, w_im = u * prm.W[l] - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < prm.H[l] && w_im < prm.W[l]) mask |= 1 << l;
      }
    }
  }
  // deterministic compaction: exclusive scan of the per-thread visible-level counts
  {
    const int c = __popc(mask);
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int nb = __shfl_up(inc, o);
      if (lane >= o) inc += nb;
    }
    if (lane == 63) scan_s[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += scan_s[k];
    int pos = base + inc - c;
    if (t == 255) scan_s[4] = base + inc;
#pragma unroll
    for (int l = 0; l < AGG_MAX_L; ++l) {
      if ((mask >> l) & 1) {
        const int Hl = prm.H[l], Wl = prm.W[l];
        const float h_im = v * Hl - 0.5f, w_im = u * Wl - 0.5f;
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lw = w_im - w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= Hl - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= Wl - 1;
        const int y0 = t_ok ? h_low : 0, y1 = b_ok ? h_low + 1 : Hl - 1;
        const int x0 = l_ok ? w_low : 0, x1 = r_ok ? w_low + 1 : Wl - 1;
        int4 r0;
        r0.x = ((n1 * prm.S + prm.start[l]) + y0 * Wl + x0) * 256;   // element offset of the (clamped) top-left pixel
        r0.y = (x1 - x0) * 256;                                      // 0 when the right column is clamped away
        r0.z = (y1 - y0) * Wl * 256;
        r0.w = ((n1 * prm.L + l) * prm.P + p1) * 8;                  // row of this sample in w_s
        float4 wq;
        wq.x = (t_ok && l_ok) ? hh * hw : 0.f;
        wq.y = (t_ok && r_ok) ? hh * lw : 0.f;
        wq.z = (b_ok && l_ok) ? lh * hw : 0.f;
        wq.w = (b_ok && r_ok) ? lh * lw : 0.f;
        rec_s[2 * pos] = r0;
        rec_s[2 * pos + 1] = make_int4(__float_as_int(wq.x), __float_as_int(wq.y), __float_as_int(wq.z), __float_as_int(wq.w));
        ++pos;
      }
    }
  }

  // ---- phase 2: softmax over cams x levels x points per group (detr3d_transformer.py:539-540)
  if constexpr ((ABL & 2) == 0) {
    const int gg = t & 7, r = t >> 3;
    float lg[AGG_MAX_NLP / 32];
    float mx = -INFINITY;
    int n = 0, j = r;   // idx = r + 32*i  ->  (camera n, logit j) tracked incrementally (J > 32)
    while (j >= J) { j -= J; ++n; }
    const float* Ua = U + (long)a * J * 8 + gg;
    const float* Vg = Vc + gg;
#pragma unroll
    for (int i = 0; i < AGG_MAX_NLP / 32; ++i) {
      lg[i] = -INFINITY;
      if (r + 32 * i < NLP) lg[i] = Ua[j * 8] + Vg[(n * J + j) * 8];
      mx = fmaxf(mx, lg[i]);
      j += 32;
      while (j >= J) { j -= J; ++n; }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (lane < 8) stat_s[wv * 8 + lane] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat_s[gg], stat_s[8 + gg]), fmaxf(stat_s[16 + gg], stat_s[24 + gg]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < AGG_MAX_NLP / 32; ++i) {
      lg[i] = (r + 32 * i < NLP) ? __expf(lg[i] - mx) : 0.f;
      sum += lg[i];
    }
    sum += __shfl_xor(sum, 8);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (lane < 8) stat_s[32 + wv * 8 + lane] = sum;
    __syncthreads();
    sum = (stat_s[32 + gg] + stat_s[40 + gg]) + (stat_s[48 + gg] + stat_s[56 + gg]);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < AGG_MAX_NLP / 32; ++i) {
      const int idx = r + 32 * i;
      if (idx < NLP) w_s[idx * 8 + gg] = lg[i] * inv;
    }
  }
  __syncthreads();
  const int nsamp = (ABL & 1) ? 0 : scan_s[4];

  if constexpr (RP == 2 && sizeof(TV) == 2) {
    // ---- phase 2.5 + 3 (bf16, de-duplicated gather).  The 13 points of a query land on few distinct pixels of the coarse
    // pyramid levels (measured on the benchmark case: 228 non-zero bilinear taps -> 85 distinct token rows per query), so
    // the taps are merged per token row first: an LDS hash table (open addressing, atomicCAS) elects one leader tap per row,
    // leaders get compact slots (deterministic block scan), every tap adds softmax weight x bilinear weight of its 8 groups
    // into its leader's weight vector (LDS float atomics), and the gather reads each distinct row ONCE.  Identical math up
    // to fp32 reassociation.  Queries with more than RCAP visible samples or LCAP distinct rows keep the row-pair gather.
    if (nsamp <= AGG_DEDUP_RCAP) {          // workgroup-uniform
      constexpr int TCAP = 4 * AGG_DEDUP_RCAP, LCAP = AGG_DEDUP_LCAP, HT = 512;
      int* key_s = reinterpret_cast<int*>(rec_s + 2 * AGG_DEDUP_RCAP);   // [TCAP] token-row element offset of tap i (-1: zero weight)
      int* lead_s = key_s + TCAP;                           // [TCAP] compact slot of the tap's leader
      int* tab_s = lead_s + TCAP;                           // [HT]   hash table: leader tap per token row
      int* loff_s = tab_s + HT;                             // [LCAP] element offset of compact slot c
      float* wl_s = reinterpret_cast<float*>(loff_s + LCAP);   // [LCAP][8] merged weights of compact slot c
      const int T = 4 * nsamp;
      for (int i = t; i < T; i += 256) {
        const int4 r0 = rec_s[2 * (i >> 2)];
        const int4 r1 = rec_s[2 * (i >> 2) + 1];
        const int q = i & 3;
        const int bw = q == 0 ? r1.x : q == 1 ? r1.y : q == 2 ? r1.z : r1.w;
        key_s[i] = (__int_as_float(bw) != 0.f) ? r0.x + ((q & 1) ? r0.y : 0) + ((q & 2) ? r0.z : 0) : -1;
      }
      for (int i = t; i < HT; i += 256) tab_s[i] = -1;
      for (int i = t; i < LCAP * 8; i += 256) wl_s[i] = 0.f;
      __syncthreads();
      int ld[2] = {-1, -1}, flag[2] = {0, 0};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = t + 256 * h;
        if (i < T) {
          const int k = key_s[i];
          if (k >= 0) {
            unsigned hs = ((unsigned)(k >> 8) * 2654435761u) >> 23;      // token index -> 9 bits
            for (;;) {
              const int old = atomicCAS(&tab_s[hs], -1, i);
              if (old == -1) { ld[h] = i; break; }
              if (key_s[old] == k) { ld[h] = old; break; }
              hs = (hs + 1) & (HT - 1);
            }
            flag[h] = ld[h] == i ? 1 : 0;
          }
        }
      }
      {
        const int c = flag[0] + flag[1];
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int nb = __shfl_up(inc, o);
          if (lane >= o) inc += nb;
        }
        if (lane == 63) scan_s[8 + wv] = inc;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < wv; ++k) base += scan_s[8 + k];
        int pos = base + inc - c;
        if (t == 255) scan_s[12] = base + inc;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = t + 256 * h;
          if (i < T && flag[h]) {
            lead_s[i] = pos;
            if (pos < LCAP) loff_s[pos] = key_s[i];
            ++pos;
          }
        }
      }
      __syncthreads();
      const int nlead = scan_s[12];
      if (nlead <= LCAP) {                    // workgroup-uniform
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = t + 256 * h;
          if (i < T && ld[h] >= 0 && !flag[h]) lead_s[i] = lead_s[ld[h]];   // leaders' entries already hold their slot
        }
        __syncthreads();
        for (int idx = t; idx < T * 8; idx += 256) {
          const int i = idx >> 3, gq = idx & 7;
          if (key_s[i] < 0) continue;
          const int4 r0 = rec_s[2 * (i >> 2)];
          const int4 r1 = rec_s[2 * (i >> 2) + 1];
          const int q = i & 3;
          const int bw = q == 0 ? r1.x : q == 1 ? r1.y : q == 2 ? r1.z : r1.w;
          atomicAdd(&wl_s[lead_s[i] * 8 + gq], w_s[r0.w + gq] * __int_as_float(bw));
        }
        __syncthreads();
        // gather: one instruction = two distinct token rows (lanes 0-31 / 32-63), 16 bytes (8 channels) per lane
        typedef unsigned dd_u32x4 __attribute__((ext_vector_type(4)));
        const int l31 = lane & 31, hi = lane >> 5, g = l31 >> 2;
        float a8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] = 0.f;
        const bf16_t* fb = reinterpret_cast<const bf16_t*>(feat) + l31 * 8;
        for (int c0 = wv * 2; c0 < nlead; c0 += 8 * NB) {
          dd_u32x4 vv[NB];
          float ww[NB];
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const int c = c0 + 8 * i + hi;
            const bool ok = c < nlead;
            vv[i] = *reinterpret_cast<const dd_u32x4*>(fb + loff_s[ok ? c : 0]);
            ww[i] = ok ? wl_s[c * 8 + g] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              a8[2 * k] += ww[i] * __uint_as_float(vv[i][k] << 16);
              a8[2 * k + 1] += ww[i] * __uint_as_float(vv[i][k] & 0xffff0000u);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] += __shfl_xor(a8[k], 32);
        __syncthreads();
        float* red_s = w_s;
        if (hi == 0) {
          *reinterpret_cast<float4*>(red_s + wv * 256 + l31 * 8) = make_float4(a8[0], a8[1], a8[2], a8[3]);
          *reinterpret_cast<float4*>(red_s + wv * 256 + l31 * 8 + 4) = make_float4(a8[4], a8[5], a8[6], a8[7]);
        }
        __syncthreads();
        const float res = (red_s[t] + red_s[256 + t]) + (red_s[512 + t] + red_s[768 + t]);
        if (prm.out_dt == FAR3D_DT_F32) reinterpret_cast<float*>(out)[(long)a * 256 + t] = res;
        else reinterpret_cast<bf16_t*>(out)[(long)a * 256 + t] = f32_to_bf16(res);
        return;
      }
    }
  }
  if constexpr (RP != 0 && sizeof(TV) == 2) {
    // ---- phase 3 (bf16): lanes 0-31 take the left bilinear column, lanes 32-63 the right one; a lane owns 8 channels
    // (16 bytes) of its pixel, so one instruction fetches the two adjacent 512-byte token rows of a sample's top (then
    // bottom) tap pair.  Half the load instructions of the 8-byte layout, twice the samples in flight per VGPR; the two
    // column partial sums are added once at the end (the weighted sum is linear).
    typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));
    const int l31 = lane & 31, hi = lane >> 5, g = l31 >> 2;
    float a8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] = 0.f;
    const bf16_t* fb = reinterpret_cast<const bf16_t*>(feat) + l31 * 8;
    for (int b0 = wv; b0 < nsamp; b0 += 4 * NB) {
      rp_u32x4 vt[NB], vb[NB];
      float wt[NB], wb[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int si = b0 + 4 * i;
        const bool ok = si < nsamp;
        const int4 r0 = rec_s[2 * (ok ? si : 0)];
        const int4 r1 = rec_s[2 * (ok ? si : 0) + 1];
        const bf16_t* p = fb + r0.x + (hi ? r0.y : 0);
        vt[i] = *reinterpret_cast<const rp_u32x4*>(p);
        vb[i] = *reinterpret_cast<const rp_u32x4*>(p + r0.z);
        const float aw = ok ? w_s[r0.w + g] : 0.f;
        wt[i] = aw * __int_as_float(hi ? r1.y : r1.x);
        wb[i] = aw * __int_as_float(hi ? r1.w : r1.z);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a8[2 * k] += wt[i] * __uint_as_float(vt[i][k] << 16) + wb[i] * __uint_as_float(vb[i][k] << 16);
          a8[2 * k + 1] += wt[i] * __uint_as_float(vt[i][k] & 0xffff0000u) + wb[i] * __uint_as_float(vb[i][k] & 0xffff0000u);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] += __shfl_xor(a8[k], 32);
    __syncthreads();
    float* red_s = w_s;
    if (hi == 0) {
      *reinterpret_cast<float4*>(red_s + wv * 256 + l31 * 8) = make_float4(a8[0], a8[1], a8[2], a8[3]);
      *reinterpret_cast<float4*>(red_s + wv * 256 + l31 * 8 + 4) = make_float4(a8[4], a8[5], a8[6], a8[7]);
    }
    __syncthreads();
    const float res = (red_s[t] + red_s[256 + t]) + (red_s[512 + t] + red_s[768 + t]);
    if (prm.out_dt == FAR3D_DT_F32) reinterpret_cast<float*>(out)[(long)a * 256 + t] = res;
    else reinterpret_cast<bf16_t*>(out)[(long)a * 256 + t] = f32_to_bf16(res);
    return;
  }
  // ---- phase 3: gather.  lane = group*8 + chunk (4 channels); a sample = 4 coalesced token-row loads.
  const int g = lane >> 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const TV* fbase = feat + lane * 4;
  for (int b0 = wv; b0 < nsamp; b0 += 4 * NB) {
    float4 v1[NB], v2[NB], v3[NB], v4[NB];
    float cw[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int si = b0 + 4 * i;
      const bool ok = si < nsamp;
      const int4 r0 = rec_s[2 * (ok ? si : 0)];
      const int4 r1 = rec_s[2 * (ok ? si : 0) + 1];
      const TV* p00 = fbase + r0.x;
      v1[i] = load4(p00);
      v2[i] = load4(p00 + r0.y);
      v3[i] = load4(p00 + r0.z);
      v4[i] = load4(p00 + r0.z + r0.y);
      const float aw = ok ? w_s[r0.w + g] : 0.f;
      cw[i][0] = aw * __int_as_float(r1.x); cw[i][1] = aw * __int_as_float(r1.y);
      cw[i][2] = aw * __int_as_float(r1.z); cw[i][3] = aw * __int_as_float(r1.w);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      acc.x += (cw[i][0] * v1[i].x + cw[i][1] * v2[i].x) + (cw[i][2] * v3[i].x + cw[i][3] * v4[i].x);
      acc.y += (cw[i][0] * v1[i].y + cw[i][1] * v2[i].y) + (cw[i][2] * v3[i].y + cw[i][3] * v4[i].y);
      acc.z += (cw[i][0] * v1[i].z + cw[i][1] * v2[i].z) + (cw[i][2] * v3[i].z + cw[i][3] * v4[i].z);
      acc.w += (cw[i][0] * v1[i].w + cw[i][1] * v2[i].w) + (cw[i][2] * v3[i].w + cw[i][3] * v4[i].w);
    }
  }

  // ---- phase 4: cross-wave (= cross-camera/point) reduction through the (now dead) weight buffer; coalesced store
  __syncthreads();
  float* red_s = w_s;
  *reinterpret_cast<float4*>(red_s + wv * 256 + lane * 4) = acc;
  __syncthreads();
  const float res = (red_s[t] + red_s[256 + t]) + (red_s[512 + t] + red_s[768 + t]);
  if (prm.out_dt == FAR3D_DT_F32) reinterpret_cast<float*>(out)[(long)a * 256 + t] = res;
  else reinterpret_cast<bf16_t*>(out)[(long)a * 256 + t] = f32_to_bf16(res);
}

extern "C" int far3d_aggregate_forward(const void* feat, int feat_dtype, const float* ref,
                                       const float* offsets, const float* lidar2img, const float* U,
                                       const float* Vc, const int32_t* perm, void* out, int out_dt, int A, int N, int S, int C, int G,
                                       int P, int L, const int32_t* level_hw, const int32_t* level_start,
                                       const float* pc_range, float pad_h, float pad_w, void* stream) {
  FAR3D_CHECK_ARG(feat && ref && offsets && lidar2img && U && Vc && out && level_hw && level_start && pc_range,
                  "far3d_aggregate_forward: null pointer argument");
  FAR3D_CHECK_ARG(C == 256 && G == 8, "far3d_aggregate_forward: fused kernel is built for C=256,G=8 (got C=%d G=%d)", C, G);
  FAR3D_CHECK_ARG(L >= 1 && L <= AGG_MAX_L, "far3d_aggregate_forward: L=%d out of range [1,%d]", L, AGG_MAX_L);
  FAR3D_CHECK_ARG(N >= 1 && P >= 1 && N * P <= 256 && N * P * L <= AGG_MAX_NLP,
                  "far3d_aggregate_forward: N*P=%d (<=256) or N*P*L=%d (<=%d) too large", N * P, N * P * L, AGG_MAX_NLP);
  FAR3D_CHECK_ARG(feat_dtype == FAR3D_DT_F32 || feat_dtype == FAR3D_DT_BF16,
                  "far3d_aggregate_forward: unsupported feature dtype %d", feat_dtype);
  FAR3D_CHECK_ARG(A >= 0 && S > 0 && (long)N * S * C < (1L << 31), "far3d_aggregate_forward: bad sizes A=%d S=%d (N*S*C must fit int32)", A, S);
  if (A == 0) return FAR3D_OK;
  AggParams prm;
  prm.A = A; prm.N = N; prm.S = S; prm.P = P; prm.L = L;
  for (int l = 0; l < AGG_MAX_L; ++l) {
    prm.H[l] = l < L ? level_hw[2 * l] : 1;
    prm.W[l] = l < L ? level_hw[2 * l + 1] : 1;
    prm.start[l] = l < L ? level_start[l] : 0;
    if (l < L)
      FAR3D_CHECK_ARG(prm.H[l] > 0 && prm.W[l] > 0 && prm.start[l] >= 0 && prm.start[l] + prm.H[l] * prm.W[l] <= S,
                      "far3d_aggregate_forward: level %d (%dx%d @%d) exceeds S=%d", l, prm.H[l], prm.W[l], prm.start[l], S);
  }
  for (int d = 0; d < 3; ++d) { prm.pc_lo[d] = pc_range[d]; prm.pc_span[d] = pc_range[3 + d] - pc_range[d]; }
  prm.pad_w = pad_w; prm.pad_h = pad_h;
  prm.q_per_xcd = cdiv(A, 8);
  prm.out_dt = out_dt;
  FAR3D_CHECK_ARG(out_dt == FAR3D_DT_F32 || out_dt == FAR3D_DT_BF16, "far3d_aggregate_forward: unsupported output dtype %d", out_dt);
  const size_t lds = (size_t)(N * P * L * 8 + 1024 + 2 * N * P + 64) * 4 + (size_t)N * P * 4;
  dim3 grid(8 * prm.q_per_xcd), block(256);
  hipStream_t st = (hipStream_t)stream;
  const char* abl_env = getenv("FAR3D_AGG_ABLATE");   // profiling aid only
  const int abl = abl_env ? atoi(abl_env) : 0;
  const char* v1_env = getenv("FAR3D_AGG_V1");   // A/B aid: the simple v1 kernel
  if (!(v1_env && atoi(v1_env))) {
    const int nlp8 = N * P * L * 8 > 1024 ? N * P * L * 8 : 1024;
    const size_t lds3_base = (size_t)nlp8 * 4 + 64 * 4 + 64 + (size_t)N * P * L * 32;
    // de-duplication tables live behind record RCAP: keys + slots (4*RCAP ints each), 512-entry hash table, LCAP offsets + weight vectors
    const size_t dd_tail = (size_t)AGG_DEDUP_RCAP * 32 + (size_t)4 * AGG_DEDUP_RCAP * 8 + 512 * 4 + (size_t)AGG_DEDUP_LCAP * (4 + 32);
    const size_t lds3_dedup = (size_t)nlp8 * 4 + 64 * 4 + 64 + ((size_t)N * P * L * 32 > dd_tail ? (size_t)N * P * L * 32 : dd_tail);
    // opt-in (FAR3D_AGG_DEDUP=1): measured SLOWER than the plain row-pair gather (30.6 vs 22.2 us per launch) although it reads
    // 2.7x fewer token rows -- the extra barrier phases, CAS probes and LDS float atomics cost more than the gather saves
    static const int no_dedup = !(getenv("FAR3D_AGG_DEDUP") && atoi(getenv("FAR3D_AGG_DEDUP")));
    const size_t lds3 = lds3_base;
    if (feat_dtype == FAR3D_DT_F32)
      hipLaunchKernelGGL((aggregate_v3_kernel<float, 2>), grid, block, lds3, st, (const float*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 1)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 4, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 2)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 4, 2>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 3)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 4, 3>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 5)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 2>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 6)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 4)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 4>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 7)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 2, 0, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 8)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 8, 0, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 10)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 1, 0, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (abl == 9)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 6, 0, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else if (!no_dedup && abl == 0)   // de-duplicated gather (row-pair gather for the rare queries above AGG_DEDUP_RCAP samples)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 2, 0, 2>), grid, block, lds3_dedup, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    else   // row-pair gather, 2 samples (4 x 1 KB row-pair loads) in flight per wave: measured best (more in flight thrashes L2)
      hipLaunchKernelGGL((aggregate_v3_kernel<bf16_t, 2, 0, 1>), grid, block, lds3, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, out, prm);
    FAR3D_CHECK_LAUNCH("far3d_aggregate_forward");
    return FAR3D_OK;
  }
  if (feat_dtype == FAR3D_DT_F32)
    hipLaunchKernelGGL(aggregate_fwd_kernel<float>, grid, block, lds, st, (const float*)feat, ref, offsets,
                       lidar2img, U, Vc, perm, (float*)out, prm);
  else if (abl == 1)
    hipLaunchKernelGGL((aggregate_fwd_kernel<bf16_t, 1>), grid, block, lds, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, (float*)out, prm);
  else if (abl == 2)
    hipLaunchKernelGGL((aggregate_fwd_kernel<bf16_t, 2>), grid, block, lds, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, (float*)out, prm);
  else if (abl == 3)
    hipLaunchKernelGGL((aggregate_fwd_kernel<bf16_t, 3>), grid, block, lds, st, (const bf16_t*)feat, ref, offsets, lidar2img, U, Vc, perm, (float*)out, prm);
  else
    hipLaunchKernelGGL(aggregate_fwd_kernel<bf16_t>, grid, block, lds, st, (const bf16_t*)feat, ref, offsets,
                       lidar2img, U, Vc, perm, (float*)out, prm);
  FAR3D_CHECK_LAUNCH("far3d_aggregate_forward");
  return FAR3D_OK;
}
