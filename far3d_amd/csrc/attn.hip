// Decoder self-attention core: softmax(q k^T / sqrt(32)) v for 8 heads of dim 32,
// 1544 queries x 2312 keys, flash-style (the 1544 x 2312 x 8 score tensor never exists).
// Replaces the bmm/softmax/bmm inside torch.nn.MultiheadAttention as wrapped by mmcv MultiheadAttention
// (reference cfg projects/configs/far3d.py:112-116; in-tree statement models/utils/petr_transformer.py:286-326).
//
// gfx950 mapping.  Workgroup = 4 waves = 128 queries of one head; a wave owns 32 queries.  Scores are computed
// TRANSPOSED, S^T = K Q^T, so the MFMA C layout puts one query per lane column: lane (q = lane&31, hi = lane>>5)
// holds 16 of the 32 keys of a tile for ITS query, the partner lane q+32 the other 16 -> the row max / row sum
// are 15 in-register ops + one cross-half shuffle, and the running max / rescale are per-lane scalars.
// O^T = V^T P^T accumulates with the same column = query mapping.  K/V tiles of 64 keys go through LDS once per
// workgroup (K row-major, padded rows; V transposed with the key order permuted inside 16-key groups so that the
// bf16 MFMA A-fragment is ONE ds_read_b128); the next tile's global loads are in flight during the MFMAs.
// T = bf16: v_mfma_f32_32x32x16_bf16.  T = float: exact-fp32 v_mfma_f32_32x32x2_f32 (parity mode).
#include "common.hpp"
#include <atomic>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define ATT_D 32
#define ATT_KT 64

template <typename T> struct ACfg;
template <> struct ACfg<bf16_t> { static constexpr int KROW = 80, VROW = 144, KSUB = 2, CH = 1; };   // bytes
template <> struct ACfg<float> { static constexpr int KROW = 144, VROW = 144, KSUB = 4, CH = 2; };

template <typename T>
__device__ __forceinline__ void mma_att(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
}

// position of key k inside a permuted 16-key group: quads 1 and 2 swapped (matches the S^T register order)
__device__ __forceinline__ int vt_pos(int key) {
  const int quad = (key >> 2) & 3;
  const int sw = (quad == 1) ? 2 : ((quad == 2) ? 1 : quad);
  return (key & ~15) | (sw << 2) | (key & 3);
}

template <typename T, int NS>     // NS: key-range parts per workgroup (2 waves each)
__global__ __launch_bounds__(128 * NS) void attn_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                            const T* __restrict__ V, void* __restrict__ O, int out_dt, int Aq,
                                                            int Nk, int ldq, int ldk, int ldv, int ldo, float scale,
                                                            const int* __restrict__ hole_count, int hole_start, int hole_end) {
  // Workgroup = 64 queries of one head, 2*NS waves.  Waves 2p, 2p+1 walk part p of the keys for queries [0,32) / [32,64)
  // (flash-decoding style split); the NS partial (max, sum, O) states are merged through LDS at the end.  The loop is a chain
  // of latencies (global load -> LDS -> MFMA -> exp -> MFMA, ~1.7 us per 64-key tile), and 200 workgroups of 4 waves leave most
  // SIMDs with a single wave: more parts = fewer serial tiles per wave and more waves per SIMD (bf16: NS = 4, 9 tiles per wave
  // at 2312 keys; fp32 keeps NS = 2, its tiles do not fit four times into the static LDS budget).
  constexpr int NT = 128 * NS;
  constexpr int KROW = ACfg<T>::KROW, VROW = ACfg<T>::VROW, KSUB = ACfg<T>::KSUB, CH = ACfg<T>::CH;
  constexpr int E = 16 / sizeof(T);
  constexpr int KBYTES = ATT_KT * KROW, VBYTES = (sizeof(T) == 2 ? ATT_D : ATT_KT) * VROW;
  // static LDS up to the 64 KB static limit (the shipped NS = 2 / 4 instantiations, unchanged); the 8-part variant's 78 KB are dynamic
  constexpr int LDS_TOTAL = NS * KBYTES + NS * VBYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  __shared__ __attribute__((aligned(16))) unsigned char smem_st[LDS_TOTAL <= 65536 ? LDS_TOTAL : 16];
  unsigned char* smem = LDS_TOTAL <= 65536 ? smem_st : smem_dyn;
  static_assert((NS - 1) * 2 * 18 * 64 * 4 <= NS * (KBYTES + VBYTES), "merge buffer must fit the tile buffers");
  unsigned char* KsAll = smem;
  unsigned char* VsAll = smem + NS * KBYTES;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, hi = lane >> 5;
  const int half = wv >> 1;                     // key part of this wave, 0..NS-1
  const int head = blockIdx.y;
  const int q = blockIdx.x * 64 + (wv & 1) * 32 + l31;
  const bool q_ok = q < Aq;
  const int ntiles = (Nk + ATT_KT - 1) / ATT_KT;
  const int htiles = (ntiles + NS - 1) / NS;    // tiles per part; the last parts may have fewer (or none)
  // keys [hole_lo, hole_end) hold no query (fixed-capacity proposal mode): masked like keys past Nk; their K / V rows must be
  // finite (the producer zero-fills them) because a masked probability is an exact 0 that still multiplies V
  const int hole_lo = hole_count ? hole_start + min(max(*hole_count, 0), hole_end - hole_start) : hole_end;
  unsigned char* Ks = KsAll + half * KBYTES;
  unsigned char* Vs = VsAll + half * VBYTES;

  // Q fragments (B operand: column = query), held in registers for the whole key loop
  u32x4_t qf[KSUB];
#pragma unroll
  for (int kk = 0; kk < KSUB; ++kk) {
    qf[kk] = u32x4_t{0u, 0u, 0u, 0u};
    if (q_ok) {
      const T* src = Q + (long)q * ldq + head * ATT_D + kk * 2 * E + hi * E;
      qf[kk] = *reinterpret_cast<const u32x4_t*>(src);
      if constexpr (sizeof(T) == 4) {  // torch scales q before the product (F.multi_head_attention_forward)
        qf[kk].x = __float_as_uint(__uint_as_float(qf[kk].x) * scale);
        qf[kk].y = __float_as_uint(__uint_as_float(qf[kk].y) * scale);
        qf[kk].z = __float_as_uint(__uint_as_float(qf[kk].z) * scale);
        qf[kk].w = __float_as_uint(__uint_as_float(qf[kk].w) * scale);
      }
    }
  }

  // cooperative staging of NS K/V tiles per iteration (one per part): chunk id -> (part, key, 16-B piece)
  constexpr int CPK = ATT_D * sizeof(T) / 16;  // 16-B chunks per key row: 4 (bf16) / 8 (f32)
  constexpr int CH2 = 2 * CH;
  u32x4_t kreg[CH2], vreg[CH2];
  auto gload = [&](int it) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH2; ++c) {
      const int id = t + c * NT, h = id / (ATT_KT * CPK), rem = id % (ATT_KT * CPK), key = rem / CPK, ck = rem % CPK;
      const int tl = it + h * htiles;
      const int gk = tl * ATT_KT + key;
      kreg[c] = u32x4_t{0u, 0u, 0u, 0u};
      vreg[c] = u32x4_t{0u, 0u, 0u, 0u};
      if (tl < ntiles && gk < Nk) {
        kreg[c] = *reinterpret_cast<const u32x4_t*>(K + (long)gk * ldk + head * ATT_D + ck * E);
        vreg[c] = *reinterpret_cast<const u32x4_t*>(V + (long)gk * ldv + head * ATT_D + ck * E);
      }
    }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH2; ++c) {
      const int id = t + c * NT, h = id / (ATT_KT * CPK), rem = id % (ATT_KT * CPK), key = rem / CPK, ck = rem % CPK;
      *reinterpret_cast<u32x4_t*>(KsAll + h * KBYTES + key * KROW + ck * 16) = kreg[c];
      unsigned char* vs = VsAll + h * VBYTES;
      if constexpr (sizeof(T) == 2) {
        const int pos = vt_pos(key);
        const uint32_t w[4] = {vreg[c].x, vreg[c].y, vreg[c].z, vreg[c].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint16_t val = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
          *reinterpret_cast<uint16_t*>(vs + (ck * 8 + e) * VROW + pos * 2) = val;
        }
      } else {
        *reinterpret_cast<u32x4_t*>(vs + key * VROW + ck * 16) = vreg[c];
      }
    }
  };

  f32x16_t o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  gload(0);
  lstore();
  __syncthreads();
  for (int it = 0; it < htiles; ++it) {
    const bool more = it + 1 < htiles;
    if (more) gload(it + 1);
    const int tile = it + half * htiles;
    if (tile < ntiles && !(tile * ATT_KT >= hole_lo && (tile + 1) * ATT_KT <= hole_end)) {   // wave-uniform; tiles inside the hole are skipped
      // ---- S^T = K Q^T for the two 32-key sub-tiles
      f32x16_t s[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
          const u32x4_t a = *reinterpret_cast<const u32x4_t*>(Ks + (u * 32 + l31) * KROW + kk * 32 + hi * 16);
          mma_att<T>(s[u], a, qf[kk]);
        }
      }
      // ---- online softmax (per-lane query)
      const int kbase = tile * ATT_KT + 4 * hi;
      float mloc = -INFINITY;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = s[u][r];
          if constexpr (sizeof(T) == 2) v *= scale;
          const int key = kbase + u * 32 + (r & 3) + 8 * (r >> 2);
          v = (key < Nk && !(key >= hole_lo && key < hole_end)) ? v : -INFINITY;
          s[u][r] = v;
          mloc = fmaxf(mloc, v);
        }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
      const float m_new = fmaxf(m_run, mloc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = (sizeof(T) == 2) ? __expf(m_run - m_use) : expf(m_run - m_use);   // m_run = -inf -> 0
      float psum = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = (sizeof(T) == 2) ? __expf(s[u][r] - m_use) : expf(s[u][r] - m_use);
          s[u][r] = p;
          psum += p;
        }
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;

      // ---- O^T += V^T P^T
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            u32x4_t b;
            b.x = pack_bf16x2(s[u][8 * g + 0], s[u][8 * g + 1]);
            b.y = pack_bf16x2(s[u][8 * g + 2], s[u][8 * g + 3]);
            b.z = pack_bf16x2(s[u][8 * g + 4], s[u][8 * g + 5]);
            b.w = pack_bf16x2(s[u][8 * g + 6], s[u][8 * g + 7]);
            const u32x4_t a = *reinterpret_cast<const u32x4_t*>(Vs + l31 * VROW + ((u * 32 + 16 * g) + hi * 8) * 2);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), o, 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float a = *reinterpret_cast<const float*>(Vs + key * VROW + l31 * 4);
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[u][r], o, 0, 0, 0);
          }
      }
    }
    __syncthreads();
    if (more) lstore();
    __syncthreads();
  }

  // ---- merge the key parts: waves of parts 1..NS-1 publish (m, l, O) per lane, waves 0,1 combine in part order and store
  float l_tot = l_run + __shfl_xor(l_run, 32);   // both lane halves of a query share m_run
  float* mb = reinterpret_cast<float*>(smem);    // [NS-1 parts][2 waves][18][64 lanes]
  if (half >= 1) {
    float* dst = mb + ((half - 1) * 2 + (wv & 1)) * 18 * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64] = o[r];
    dst[16 * 64] = m_run;
    dst[17 * 64] = l_tot;
  }
  __syncthreads();
  if (half == 0 && q_ok) {
    float m = m_run;
#pragma unroll
    for (int p = 1; p < NS; ++p) m = fmaxf(m, mb[((p - 1) * 2 + (wv & 1)) * 18 * 64 + lane + 16 * 64]);
    const float mu = (m == -INFINITY) ? 0.f : m;
    float ap[NS];                                  // a part that saw no key has m = -inf -> weight 0
    ap[0] = expf(m_run - mu);
    float den = l_tot * ap[0];
#pragma unroll
    for (int p = 1; p < NS; ++p) {
      const float* src = mb + ((p - 1) * 2 + (wv & 1)) * 18 * 64 + lane;
      ap[p] = expf(src[16 * 64] - mu);
      den += src[17 * 64] * ap[p];
    }
    const float inv = 1.f / den;
    const long off = (long)q * ldo + head * ATT_D + 4 * hi;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = o[4 * qd + e] * ap[0];
#pragma unroll
      for (int p = 1; p < NS; ++p) {
        const float* src = mb + ((p - 1) * 2 + (wv & 1)) * 18 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += src[(4 * qd + e) * 64] * ap[p];
      }
      const float4 r = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
      if (out_dt == FAR3D_DT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off + 8 * qd) = r;
      else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(O) + off + 8 * qd) = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
    }
  }
}

// fp32, round 6: the key loop WITHOUT LDS and without barriers.  The staged kernel above walks its 64-key tiles in lockstep (two workgroup
// barriers per tile: every wave of a SIMD is in its MFMA phase, then every wave in its exp phase), so the exact-fp32 MFMA pipe -- the
// bound of this kernel: 3.7 GFLOP at 256 FLOP / cycle / CU -- idles half the time.  Here a wave owns (32 queries, one part of the keys)
// and feeds the MFMAs from registers it loads itself: the A operand of S^T = K Q^T is lane (key, hi)'s 16 bytes of its key row, the A
// operand of O^T += V^T P^T one float of a value row per lane (128 contiguous bytes per half-wave) -- both are plain coalesced global
// loads of L2-resident rows (a head's K and V are 590 KB; workgroup id % heads = head, so with 8 heads an XCD's L2 holds ONE head).  The
// next 32-key tile's 20 loads are in flight under the current tile's 32 MFMAs; waves run free of each other until the merge of the key
// parts (one barrier).  Same arithmetic as the staged kernel per 32-key tile: exact products, fp32 accumulation, online softmax.
template <int QB, int NS, int QW = 1>     // QB query blocks per workgroup, NS key parts (waves = QB * NS); a wave owns QW x 32 queries
__global__ __launch_bounds__(64 * QB * NS) void attn_f32_direct_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                      const float* __restrict__ V, void* __restrict__ O, int out_dt, int Aq,
                                                                      int Nk, int heads, int ldq, int ldk, int ldv, int ldo, float scale,
                                                                      const int* __restrict__ hole_count, int hole_start, int hole_end) {
  // QW = 2: one set of K / V fragments feeds two independent MFMA chains (queries l31 and 32 + l31 of the block): half the loads per
  // product and a second chain to issue while the first one's result is on its way
  __shared__ float mb[(NS - 1) * QB * QW * 18 * 64];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, hi = lane >> 5;
  const int part = wv / QB, qb = wv % QB;
  const int head = blockIdx.x % heads, qblk = blockIdx.x / heads;
  const int q0 = (qblk * QB + qb) * 32 * QW + l31;          // this lane's query of chain w: q0 + 32 w
  const int ntiles = (Nk + 31) >> 5;                       // 32-key tiles
  const int ptiles = (ntiles + NS - 1) / NS;
  const int t0 = part * ptiles, t1 = min(ntiles, t0 + ptiles);
  const int hole_lo = hole_count ? hole_start + min(max(*hole_count, 0), hole_end - hole_start) : hole_end;
  u32x4_t qf[QW][4];
#pragma unroll
  for (int w = 0; w < QW; ++w)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[w][kk] = u32x4_t{0u, 0u, 0u, 0u};
      if (q0 + 32 * w < Aq) {
        const float4 v = *reinterpret_cast<const float4*>(Q + (long)(q0 + 32 * w) * ldq + head * ATT_D + kk * 8 + hi * 4);
        qf[w][kk] = u32x4_t{__float_as_uint(v.x * scale), __float_as_uint(v.y * scale), __float_as_uint(v.z * scale), __float_as_uint(v.w * scale)};
      }
    }
  const float* Kh = K + head * ATT_D + hi * 4;
  const float* Vh = V + head * ATT_D + l31;
  u32x4_t kf[2][4];
  float vf[2][16];
  auto gload = [&](int tile, u32x4_t (&kd)[4], float (&vd)[16]) __attribute__((always_inline)) {
    const int k0 = tile * 32;
    const float* kr = Kh + (long)min(k0 + l31, Nk - 1) * ldk;           // rows past Nk: a valid row, masked below
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kd[kk] = *reinterpret_cast<const u32x4_t*>(kr + kk * 8);
#pragma unroll
    for (int r = 0; r < 16; ++r) vd[r] = Vh[(long)min(k0 + (r & 3) + 8 * (r >> 2) + 4 * hi, Nk - 1) * ldv];
  };
  f32x16_t o[QW];
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int w = 0; w < QW; ++w) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[w][r] = 0.f;
    m_run[w] = -INFINITY; l_run[w] = 0.f;
  }
  auto compute = [&](int tile, const u32x4_t (&kd)[4], const float (&vd)[16]) __attribute__((always_inline)) {
    if (tile * 32 >= hole_lo && (tile + 1) * 32 <= hole_end) return;      // wave-uniform: a tile inside the hole
    f32x16_t s[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[w][r] = 0.f;
    // exact-fp32 MFMAs of the chains interleaved: consecutive instructions do not wait for each other's accumulator
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float ka[4] = {__uint_as_float(kd[kk].x), __uint_as_float(kd[kk].y), __uint_as_float(kd[kk].z), __uint_as_float(kd[kk].w)};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int w = 0; w < QW; ++w) {
          const float qe = __uint_as_float(e == 0 ? qf[w][kk].x : e == 1 ? qf[w][kk].y : e == 2 ? qf[w][kk].z : qf[w][kk].w);
          s[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qe, s[w], 0, 0, 0);
        }
    }
    const int kbase = tile * 32 + 4 * hi;
    const bool ragged = (tile + 1) * 32 > Nk || ((tile + 1) * 32 > hole_lo && tile * 32 < hole_end);     // wave-uniform: only a ragged tile is masked
    // e^(s - m) as the hardware's 2^x of one fma (1 ulp, the accuracy of expf without its ~18 instructions: per 32-key tile the exp phase
    // was as long as the 32 exact-fp32 MFMAs, profiles/r6/attn_f32_ab.txt)
    constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      if (ragged) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2);
          s[w][r] = (key < Nk && !(key >= hole_lo && key < hole_end)) ? s[w][r] : -INFINITY;
        }
      }
      float mloc = fmaxf(fmaxf(fmaxf(s[w][0], s[w][1]), fmaxf(s[w][2], s[w][3])), fmaxf(fmaxf(s[w][4], s[w][5]), fmaxf(s[w][6], s[w][7])));
      mloc = fmaxf(mloc, fmaxf(fmaxf(fmaxf(s[w][8], s[w][9]), fmaxf(s[w][10], s[w][11])), fmaxf(fmaxf(s[w][12], s[w][13]), fmaxf(s[w][14], s[w][15]))));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
      const float m_new = fmaxf(m_run[w], mloc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float mneg = -m_use * LOG2E;
      const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[w], LOG2E, mneg));      // m_run = -inf -> 0
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[w][r], LOG2E, mneg));
        s[w][r] = p;
        psum += p;
      }
      l_run[w] = l_run[w] * alpha + psum;
      m_run[w] = m_new;
      if (__ballot(alpha != 1.f)) {                                                // the running maximum moved for some query of the wave
#pragma unroll
        for (int r = 0; r < 16; ++r) o[w][r] *= alpha;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int w = 0; w < QW; ++w) o[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(vd[r], s[w][r], o[w], 0, 0, 0);
  };
  if (t0 < t1) {
    gload(t0, kf[0], vf[0]);
    for (int tile = t0; tile < t1; tile += 2) {
      if (tile + 1 < t1) gload(tile + 1, kf[1], vf[1]);
      compute(tile, kf[0], vf[0]);
      if (tile + 1 >= t1) break;
      if (tile + 2 < t1) gload(tile + 2, kf[0], vf[0]);
      compute(tile + 1, kf[1], vf[1]);
    }
  }
  // ---- merge the key parts in part order (as the staged kernel)
  float l_tot[QW];
#pragma unroll
  for (int w = 0; w < QW; ++w) l_tot[w] = l_run[w] + __shfl_xor(l_run[w], 32);
  if (part >= 1) {
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      float* dst = mb + (((part - 1) * QB + qb) * QW + w) * 18 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = o[w][r];
      dst[16 * 64] = m_run[w];
      dst[17 * 64] = l_tot[w];
    }
  }
  __syncthreads();
  if (part != 0) return;
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    const int q = q0 + 32 * w;
    if (q >= Aq) continue;
    float m = m_run[w];
#pragma unroll
    for (int p = 1; p < NS; ++p) m = fmaxf(m, mb[(((p - 1) * QB + qb) * QW + w) * 18 * 64 + lane + 16 * 64]);
    const float mu = (m == -INFINITY) ? 0.f : m;
    float ap[NS];
    ap[0] = expf(m_run[w] - mu);
    float den = l_tot[w] * ap[0];
#pragma unroll
    for (int p = 1; p < NS; ++p) {
      const float* src = mb + (((p - 1) * QB + qb) * QW + w) * 18 * 64 + lane;
      ap[p] = expf(src[16 * 64] - mu);
      den += src[17 * 64] * ap[p];
    }
    const float inv = 1.f / den;
    const long off = (long)q * ldo + head * ATT_D + 4 * hi;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = o[w][4 * qd + e] * ap[0];
#pragma unroll
      for (int p = 1; p < NS; ++p) {
        const float* src = mb + (((p - 1) * QB + qb) * QW + w) * 18 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += src[(4 * qd + e) * 64] * ap[p];
      }
      const float4 r = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
      if (out_dt == FAR3D_DT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(O) + off + 8 * qd) = r;
      else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(O) + off + 8 * qd) = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
    }
  }
}

template <int QB, int NS, int QW = 1>
static void launch_attn_f32_direct(const float* q, const float* k, const float* v, void* out, int out_dt, int Aq, int Nk, int heads, int ldq,
                                   int ldk, int ldv, int ldo, float scale, const int* hole_count, int hole_start, int hole_end, hipStream_t st) {
  const int nqb = (Aq + 32 * QB * QW - 1) / (32 * QB * QW);
  hipLaunchKernelGGL((attn_f32_direct_kernel<QB, NS, QW>), dim3(nqb * heads), dim3(64 * QB * NS), 0, st, q, k, v, out, out_dt, Aq, Nk, heads, ldq,
                     ldk, ldv, ldo, scale, hole_count, hole_start, hole_end);
}

// A/B switch of the fp32 instantiation (far3d_attention_f32_variant; tools/probe/attn_f32_ab.py): 0 = default
static std::atomic<int> g_attn_f32_variant{0};
extern "C" int far3d_attention_f32_variant(int variant) {
  if (variant < 0) return g_attn_f32_variant.load(std::memory_order_relaxed);
  return g_attn_f32_variant.exchange(variant, std::memory_order_relaxed);
}

extern "C" int far3d_attention_forward(const void* q, const void* k, const void* v, int dtype, void* out, int out_dt, int Aq,
                                       int Nk, int heads, int head_dim, int ldq, int ldk, int ldv, int ldo,
                                       float scale, const int32_t* hole_count, int hole_start, int hole_end, void* stream) {
  FAR3D_CHECK_ARG(q && k && v && out, "far3d_attention_forward: null pointer argument");
  FAR3D_CHECK_ARG(head_dim == ATT_D, "far3d_attention_forward: head_dim must be %d (got %d)", ATT_D, head_dim);
  FAR3D_CHECK_ARG(Aq > 0 && Nk > 0 && heads > 0, "far3d_attention_forward: bad sizes Aq=%d Nk=%d heads=%d", Aq, Nk, heads);
  FAR3D_CHECK_ARG(dtype == FAR3D_DT_F32 || dtype == FAR3D_DT_BF16, "far3d_attention_forward: unsupported dtype %d", dtype);
  FAR3D_CHECK_ARG(!hole_count || (0 <= hole_start && hole_start <= hole_end && hole_end <= Nk), "far3d_attention_forward: bad key hole [%d, %d) for Nk=%d",
                  hole_start, hole_end, Nk);
  if (!hole_count) hole_start = hole_end = 0;
  const int es = dtype == FAR3D_DT_F32 ? 4 : 2;
  FAR3D_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                  (ldq * es) % 16 == 0 && (ldk * es) % 16 == 0 && (ldv * es) % 16 == 0 && ldo % 4 == 0 &&
                  (out_dt == FAR3D_DT_F32 || out_dt == FAR3D_DT_BF16),
                  "far3d_attention_forward: pointers / row strides must be 16-byte aligned");
  dim3 grid((Aq + 63) / 64, heads);
  hipStream_t st = (hipStream_t)stream;
  // key parts per workgroup (2 waves each): fp32 2, bf16 4 (an 8-part bf16 instantiation was measured in round 4 -- 27.4 vs 28 us,
  // profiles/r4/attn_parts8.txt: the per-tile MFMA -> exp -> MFMA chain is what a wave waits on -- and removed)
  if (dtype == FAR3D_DT_F32) {
    // fp32 (the in-tolerance engine's decoder): the register-fed kernel (attn_f32_direct_kernel) since round 6; the staged kernel with 4
    // key parts (its four fp32 K / V tile pairs, 72 KB, are dynamic LDS) or 2 (round 5) stays selectable for A/B.
    // far3d_attention_f32_variant (A/B): 0 the default; 10 * QB + NS (+ 100: 64 queries per wave) a register-fed instantiation; 2 / 4 the
    // staged kernel with that many key parts (round 5 / earlier in round 6)
    const int parts = g_attn_f32_variant.load(std::memory_order_relaxed);
    const float* qf_ = (const float*)q; const float* kf_ = (const float*)k; const float* vf_ = (const float*)v;
#define FAR3D_ATTN_DIRECT(QB_, NS_, QW_) launch_attn_f32_direct<QB_, NS_, QW_>(qf_, kf_, vf_, out, out_dt, Aq, Nk, heads, ldq, ldk, ldv, ldo, scale, \
                                                                              (const int*)hole_count, hole_start, hole_end, st)
    // measured at 1544 x 2312 x 8 heads (profiles/r6/attn_f32_ab.txt): staged 4 parts 80 us; register-fed with expf 75; with the one-fma
    // exp2 62-63 in every shape tried (2 x 4, 1 x 4, 1 x 8 with 64 queries per wave) -- the exact-fp32 MFMA runs at the VECTOR fp32 rate
    // (MI355X guide) and shares its time with the softmax's VALU work; 200-392 workgroups on 256 CUs leave a quarter of the chip idle
    if (parts == 0 || parts == 24) FAR3D_ATTN_DIRECT(2, 4, 1);
    else if (parts == 14) FAR3D_ATTN_DIRECT(1, 4, 1);
    else if (parts == 118) FAR3D_ATTN_DIRECT(1, 8, 2);      // 64 queries per wave
#undef FAR3D_ATTN_DIRECT
    else if (parts == 4) {
      constexpr int lds = 4 * (ATT_KT * ACfg<float>::KROW + ATT_KT * ACfg<float>::VROW);
      static std::atomic<unsigned long long> lds_ok{0};
      if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&attn_fwd_kernel<float, 4>), lds, lds_ok, "far3d_attention_forward")) return rc;
      hipLaunchKernelGGL((attn_fwd_kernel<float, 4>), grid, dim3(512), lds, st, (const float*)q, (const float*)k, (const float*)v, out, out_dt,
                         Aq, Nk, ldq, ldk, ldv, ldo, scale, (const int*)hole_count, hole_start, hole_end);
    } else {
      FAR3D_CHECK_ARG(parts == 2, "far3d_attention_forward: far3d_attention_f32_variant(%d) names no fp32 instantiation", parts);
      hipLaunchKernelGGL((attn_fwd_kernel<float, 2>), grid, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, out, out_dt,
                         Aq, Nk, ldq, ldk, ldv, ldo, scale, (const int*)hole_count, hole_start, hole_end);
    }
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 4>), grid, dim3(512), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       out, out_dt, Aq, Nk, ldq, ldk, ldv, ldo, scale, (const int*)hole_count, hole_start, hole_end);
  }
  FAR3D_CHECK_LAUNCH("far3d_attention_forward");
  return FAR3D_OK;
}
