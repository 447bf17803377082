// Row-resident chains of the decoder layer: everything between the attention core and the
// aggregation kernel, and between the aggregation kernel and the next layer's attention core, is ROW-LOCAL -- projections,
// residuals, LayerNorms, the FFN -- so a workgroup that owns 16 query rows can run the whole sequence with the rows resident
// in LDS and only the weights streaming by:
//
//   far3d_rowchain_attn_out   x1 = LN0(att W_out^T + b + x);  UL = [x1 + pos | x1] W_wl^T + b         (3 launches -> 1)
//     ref models/utils/detr3d_transformer.py:385-400 (self-attention output, norm), :522-531 (attention-weight and
//     key-point-offset linears of the perspective-aware aggregation, merged into one weight as in the unfused path)
//   far3d_rowchain_ffn        x2 = LN1(agg W_o^T + b + x1);  out = LN2(relu(x2 W_1^T + b) W_2^T + b + x2);
//                             QKV' = [out + pos | out] W_qkv'^T + b   (next layer's in-projection)    (6 launches -> 1)
//     ref detr3d_transformer.py:566-569 (output projection), :398-422 (norm, FFN, norm), :378-384 (next layer's q/k/v)
//
// The unfused path (far3d_conv2d_nhwc + far3d_layernorm, 11 launches per decoder layer, each 5-9 us of mostly latency at
// 1544 rows) stays the default until the chains have been measured; far3d_amd.engine selects them with fused_rows=True.
//
// gfx950 mapping.  Workgroup = 16 rows (one MFMA row tile; 97 workgroups at 1544 rows), 8 waves.  A GEMM step is
// D[16 x N] = A[16 x K] W^T: A is the row block in LDS as bf16 (row stride K + 8 elements: the 16 lanes of a ds_read_b128
// phase hit 16 different bank quads), read once per 256-wide K chunk into 8 fragments; W arrives straight from L2 in MFMA
// FRAGMENT ORDER (far3d_amd.ops.pack_rowchain: [column tile][k step][lane][8 bf16], so one global_load_dwordx4 of a wave is one
// contiguous KiB = 8 full lines and IS the B operand of a v_mfma_f32_16x16x32_bf16) -- no LDS staging, no barrier inside a
// GEMM; a wave owns the column tiles t = wave + 8 i and keeps the next tile's 8 fragments in flight under the current
// tile's 8 MFMAs.  fp32 accumulators; epilogues follow the unfused kernels' order (bias -> activation -> residual); the
// LayerNorm is far3d_layernorm's arithmetic instruction for instruction (one wave per row, float4 per lane, the same
// shuffle tree), so given the same GEMM result the normalised rows are bit-identical.  Rows never interact: a row's
// output does not depend on which block or launch carries it (the query-sharded decoder relies on that).
// bf16 operands only (the exact-fp32 / split-bf16 decoders keep the unfused path).
#include "common.hpp"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

#define RC_R 16
#define RC_WAVES 8
#define RC_THREADS 512
#define RC_E 256                       // embed dims
#define RC_FF 1024                     // FFN hidden
#define RC_FLD (RC_E + 4)              // floats per row of the fp32 row buffer
#define RC_A0LD (2 * RC_E + 8)         // bf16 per row of the operand buffer ([x + pos | x])
#define RC_A1LD (RC_FF + 8)            // bf16 per row of the hidden / staging buffer
#define RC_F_BYTES (RC_R * RC_FLD * 4)
#define RC_A0_BYTES (RC_R * RC_A0LD * 2)
#define RC_A1_BYTES (RC_R * RC_A1LD * 2)
#define RC_LDS (RC_F_BYTES + RC_A0_BYTES + RC_A1_BYTES)
#define RC_WL_TILES 29                 // column tiles of the merged aggregation linear: 449..464 outputs (benchmark: 416 + 39)
#define RC_WL_SLD 468                  // floats per staged row of its output

__device__ __forceinline__ float rc_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// D[16 x 16*NTILES] = A[16 x K] W^T for this wave's column tiles t = wave + 8 i (i < TW); tiles past NTILES repeat the last
// valid one (their accumulators are never stored) so that the whole step is one basic block.  An "iteration" is one column
// tile x one 256-wide K chunk = 8 weight fragments (8 KiB per wave) and 8 MFMAs; the fragments of the next RC_PF iterations are
// in flight while one computes (the wave streams ~240 KiB of weights per chain at ~1 us per L2 round trip under load: the
// depth of this ring, not the MFMA rate, sets the pace).  The scheduling barriers keep the compiler from sinking the
// prefetch next to its use (it otherwise interleaves loads and MFMAs with 2-4 loads in flight).
#define RC_PF 2
#define RC_NB (RC_PF + 1)
template <int K, int NTILES>
__device__ __forceinline__ void rc_fetch(const uint4* __restrict__ Wp, int it, uint4 (&dst)[8], int wave, int lane) {
  constexpr int KS = K / 32, TW = (NTILES + RC_WAVES - 1) / RC_WAVES;
  const int c1 = it / TW, t1 = wave + RC_WAVES * (it % TW);
  const int tile = t1 < NTILES ? t1 : NTILES - 1;
  const uint4* src = Wp + ((long)tile * KS + c1 * 8) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 8; ++s) dst[s] = src[s * 64];
}
// the first RC_PF iterations' fragments: issued BEFORE the barrier / LayerNorm phase that precedes the GEMM (weights do not
// depend on the row block), so that a GEMM step does not open with an exposed L2 round trip
template <int K, int NTILES>
__device__ __forceinline__ void rc_prefetch(const uint4* __restrict__ Wp, uint4 (&b)[RC_NB][8], int wave, int lane) {
  constexpr int NIT = (K / 256) * ((NTILES + RC_WAVES - 1) / RC_WAVES);
#pragma unroll
  for (int it = 0; it < RC_PF && it < NIT; ++it) rc_fetch<K, NTILES>(Wp, it, b[it % RC_NB], wave, lane);
  __builtin_amdgcn_sched_barrier(0);
}
template <int K, int NTILES, int LDA>
__device__ __forceinline__ void rc_gemm(const bf16_t* __restrict__ As, const uint4* __restrict__ Wp, uint4 (&b)[RC_NB][8],
                                        f32x4_t (&acc)[(NTILES + RC_WAVES - 1) / RC_WAVES], int wave, int lane) {
  static_assert(K % 256 == 0, "K chunks of 256");
  constexpr int SPC = 8, TW = (NTILES + RC_WAVES - 1) / RC_WAVES, NIT = (K / 256) * TW;
#pragma unroll
  for (int i = 0; i < TW; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bf16_t* arow = As + (lane & 15) * LDA + 8 * (lane >> 4);
  uint4 a[SPC];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it / TW, i = it % TW;
    if (it + RC_PF < NIT) rc_fetch<K, NTILES>(Wp, it + RC_PF, b[(it + RC_PF) % RC_NB], wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (i == 0) {
#pragma unroll
      for (int s = 0; s < SPC; ++s) a[s] = *reinterpret_cast<const uint4*>(arow + (c * SPC + s) * 32);
    }
#pragma unroll
    for (int s = 0; s < SPC; ++s)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[s]), __builtin_bit_cast(bf16x8_t, b[it % RC_NB][s]),
                                                       acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// accumulator element j of tile t: row 4 * (lane >> 4) + j, column 16 t + (lane & 15)
// F[r][c] = (acc + bias) + F[r][c]   (E columns: 16 tiles, 2 per wave; every (row, column) has exactly one owner lane)
__device__ __forceinline__ void rc_epi_residual(float* F, const f32x4_t (&acc)[2], const float (&bv)[2], int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (wave + RC_WAVES * i) * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = F + (4 * (lane >> 4) + j) * RC_FLD + c;
      *p = (acc[i][j] + bv[i]) + *p;
    }
  }
}
// this lane's bias values of its TW column tiles (loaded before the GEMM so that the round trip is hidden under it)
template <int NTILES>
__device__ __forceinline__ void rc_bias(const float* __restrict__ bias, float (&bv)[(NTILES + RC_WAVES - 1) / RC_WAVES], int wave, int lane) {
#pragma unroll
  for (int i = 0; i < (NTILES + RC_WAVES - 1) / RC_WAVES; ++i) {
    const int t = wave + RC_WAVES * i;
    bv[i] = bias[(t < NTILES ? t : NTILES - 1) * 16 + (lane & 15)];
  }
}

// one wave, one row: far3d_layernorm's arithmetic (norm.hip layernorm_kernel<1>, C = 256)
__device__ __forceinline__ float4 rc_layernorm_row(const float* Frow, const float4& g, const float4& bt, float eps, int lane) {
  const float4 v = *reinterpret_cast<const float4*>(Frow + lane * 4);
  const float s = (v.x + v.y) + (v.z + v.w);
  const float mean = rc_wave_sum(s) / RC_E;
  const float a = v.x - mean, b = v.y - mean, d = v.z - mean, e = v.w - mean;
  const float q = (a * a + b * b) + (d * d + e * e);
  const float rstd = 1.f / sqrtf(rc_wave_sum(q) / RC_E + eps);
  float4 o;
  o.x = (v.x - mean) * rstd * g.x + bt.x;
  o.y = (v.y - mean) * rstd * g.y + bt.y;
  o.z = (v.z - mean) * rstd * g.z + bt.z;
  o.w = (v.w - mean) * rstd * g.w + bt.w;
  return o;
}

__device__ __forceinline__ uint2 rc_pack4(const float4& v) { return make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
__device__ __forceinline__ float4 rc_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// The block's input rows: 16 x 256 bf16 (GEMM operand) -> A0, 16 x 256 f32 (residual) -> F.  Rows past M read row M - 1 and are
// zeroed (unconditional loads: a conditional one becomes a branch with its own wait); all three loads are in flight together.
__device__ __forceinline__ void rc_load_block(bf16_t* A0, float* F, const bf16_t* __restrict__ srcb, int ldb, const float* __restrict__ srcf,
                                              int ldf, int row0, int M, int tid) {
  const int rb = tid >> 5, cb = (tid & 31) * 8;
  const int r0 = tid >> 6, r1 = r0 + 8, cf = (tid & 63) * 4;
  uint4 vb = *reinterpret_cast<const uint4*>(srcb + (long)min(row0 + rb, M - 1) * ldb + cb);
  float4 v0 = rc_ld4(srcf + (long)min(row0 + r0, M - 1) * ldf + cf);
  float4 v1 = rc_ld4(srcf + (long)min(row0 + r1, M - 1) * ldf + cf);
  const bool okb = row0 + rb < M, ok0 = row0 + r0 < M, ok1 = row0 + r1 < M;      // per component: a select of whole vectors goes through scratch
  vb.x = okb ? vb.x : 0u; vb.y = okb ? vb.y : 0u; vb.z = okb ? vb.z : 0u; vb.w = okb ? vb.w : 0u;
  v0.x = ok0 ? v0.x : 0.f; v0.y = ok0 ? v0.y : 0.f; v0.z = ok0 ? v0.z : 0.f; v0.w = ok0 ? v0.w : 0.f;
  v1.x = ok1 ? v1.x : 0.f; v1.y = ok1 ? v1.y : 0.f; v1.z = ok1 ? v1.z : 0.f; v1.w = ok1 ? v1.w : 0.f;
  *reinterpret_cast<uint4*>(A0 + rb * RC_A0LD + cb) = vb;
  *reinterpret_cast<float4*>(F + r0 * RC_FLD + cf) = v0;
  *reinterpret_cast<float4*>(F + r1 * RC_FLD + cf) = v1;
}

struct RowAttnOutParams {
  const bf16_t* att; int ld_att;        // (M, E) attention-core output
  const float* x; int ldx;              // (M, E) residual: the layer's input rows
  const float* qpos; int ldq;           // (M, E) query position codes
  const uint4* w_out; const float* b_out;
  const float* g0; const float* be0;
  const uint4* w_wl; const float* b_wl; int n_wl;
  float* x1; int ldx1;                  // (M, E) out: LN0 rows
  float* ul; int ldu;                   // (M, n_wl) out: aggregation logits | key-point offsets
  const int* ul_rows;                   // optional (M): row i of ul is stored at row ul_rows[i] (far3d_agg_order's inv)
  int M; float eps;
};

__global__ __launch_bounds__(RC_THREADS) void rowchain_attn_out_kernel(RowAttnOutParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* F = reinterpret_cast<float*>(smem);
  bf16_t* A0 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES);
  float* S = reinterpret_cast<float*>(smem + RC_F_BYTES + RC_A0_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RC_R;
  rc_load_block(A0, F, P.att, P.ld_att, P.x, P.ldx, row0, P.M, tid);
  // operands of the later phases, requested now: position codes of this wave's two LayerNorm rows, the affine terms, the biases,
  // the destination row of this thread's piece of the logits / offsets
  const int ulrow_ld = P.ul_rows ? P.ul_rows[min(row0 + (tid >> 5), P.M - 1)] : 0;
  float4 pos[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) pos[rr] = rc_ld4(P.qpos + (long)min(row0 + 2 * wave + rr, P.M - 1) * P.ldq + lane * 4);
  const float4 g0 = rc_ld4(P.g0 + lane * 4), be0 = rc_ld4(P.be0 + lane * 4);
  float bo[2], bw[(RC_WL_TILES + RC_WAVES - 1) / RC_WAVES];
  rc_bias<16>(P.b_out, bo, wave, lane);
  rc_bias<RC_WL_TILES>(P.b_wl, bw, wave, lane);
  uint4 wb[RC_NB][8];
  rc_prefetch<RC_E, 16>(P.w_out, wb, wave, lane);
  __syncthreads();
  {
    f32x4_t acc[2];
    rc_gemm<RC_E, 16, RC_A0LD>(A0, P.w_out, wb, acc, wave, lane);
    rc_epi_residual(F, acc, bo, wave, lane);
  }
  rc_prefetch<2 * RC_E, RC_WL_TILES>(P.w_wl, wb, wave, lane);
  __syncthreads();                                     // F complete; all reads of A0 done
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr, row = row0 + r;
    const float4 o = rc_layernorm_row(F + r * RC_FLD, g0, be0, P.eps, lane);
    if (row < P.M) *reinterpret_cast<float4*>(P.x1 + (long)row * P.ldx1 + lane * 4) = o;
    const float4 op = make_float4(o.x + pos[rr].x, o.y + pos[rr].y, o.z + pos[rr].z, o.w + pos[rr].w);
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + lane * 4) = rc_pack4(op);
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + RC_E + lane * 4) = rc_pack4(o);
  }
  __syncthreads();
  {
    constexpr int TW = (RC_WL_TILES + RC_WAVES - 1) / RC_WAVES;
    f32x4_t acc[TW];
    rc_gemm<2 * RC_E, RC_WL_TILES, RC_A0LD>(A0, P.w_wl, wb, acc, wave, lane);
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const int t = wave + RC_WAVES * i;
      if (t < RC_WL_TILES) {
        const int c = t * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) S[(4 * (lane >> 4) + j) * RC_WL_SLD + c] = acc[i][j] + bw[i];
      }
    }
  }
  __syncthreads();
  {
    const int r = tid >> 5, row = row0 + r;
    if (row < P.M) {
      float* dst = P.ul + (long)(P.ul_rows ? ulrow_ld : row) * P.ldu;
      for (int c = (tid & 31) * 4; c < P.n_wl; c += 128) {
        const float4 v = *reinterpret_cast<const float4*>(S + r * RC_WL_SLD + c);
        if (c + 3 < P.n_wl) {
          *reinterpret_cast<float4*>(dst + c) = v;
        } else {
          dst[c] = v.x;
          if (c + 1 < P.n_wl) dst[c + 1] = v.y;
          if (c + 2 < P.n_wl) dst[c + 2] = v.z;
        }
      }
    }
  }
}

// [q | k | v] = A0[16 x 2E] W_qkv^T + b -> bf16 rows of the layer's column block (staged through A1 for 16-byte row stores).
// Shared by far3d_rowchain_ffn's tail and far3d_rowchain_qkv; the caller has issued rc_prefetch for w_qkv and a barrier after
// writing A0.
__device__ __forceinline__ void rc_qkv_tail(const bf16_t* A0, bf16_t* A1, const uint4* __restrict__ w_qkv, const float* __restrict__ b_qkv,
                                            uint4 (&wb)[RC_NB][8], bf16_t* __restrict__ qkv, int ldqkv, int row0, int M, int tid) {
  constexpr int NT = 3 * RC_E / 16, TW = NT / RC_WAVES, SLD = 3 * RC_E + 8;
  const int lane = tid & 63, wave = tid >> 6;
  float bq[TW];
  rc_bias<NT>(b_qkv, bq, wave, lane);
  f32x4_t acc[TW];
  rc_gemm<2 * RC_E, NT, RC_A0LD>(A0, w_qkv, wb, acc, wave, lane);
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    const int c = (wave + RC_WAVES * i) * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) A1[(4 * (lane >> 4) + j) * SLD + c] = f32_to_bf16(acc[i][j] + bq[i]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; ++j) {                        // 16 rows x 96 pieces of 16 bytes
    const int idx = tid + RC_THREADS * j, r = idx / 96, ch = idx - r * 96;
    if (row0 + r < M)
      *reinterpret_cast<uint4*>(qkv + (long)(row0 + r) * ldqkv + ch * 8) = *reinterpret_cast<const uint4*>(A1 + r * SLD + ch * 8);
  }
}

struct RowFfnParams {
  const bf16_t* agg; int ld_agg;        // (M, E) aggregation output
  const float* x1; int ldx1;            // (M, E) residual: LN0 rows
  const float* qpos; int ldq;
  const uint4* w_o; const float* b_o;
  const float* g1; const float* be1;
  const uint4* w_1; const float* b_1;
  const uint4* w_2; const float* b_2;
  const float* g2; const float* be2;
  const uint4* w_qkv; const float* b_qkv;   // next layer's in-projection over [out + pos | out], or null
  float* out; int ldo;                  // (M, E) out: the layer's output rows (LN2)
  bf16_t* qkv; int ldqkv;               // (M, 3E) out: next layer's [q | k | v] rows
  bf16_t* xop; int ldxop;               // optional (M, 2E) out: [out + pos | out] as bf16 (the unfused path's GEMM operand)
  int M; float eps;
};

__global__ __launch_bounds__(RC_THREADS) void rowchain_ffn_kernel(RowFfnParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* F = reinterpret_cast<float*>(smem);
  bf16_t* A0 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES);
  bf16_t* A1 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES + RC_A0_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RC_R;
  rc_load_block(A0, F, P.agg, P.ld_agg, P.x1, P.ldx1, row0, P.M, tid);
  float4 pos[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) pos[rr] = rc_ld4(P.qpos + (long)min(row0 + 2 * wave + rr, P.M - 1) * P.ldq + lane * 4);
  const float4 g1 = rc_ld4(P.g1 + lane * 4), be1 = rc_ld4(P.be1 + lane * 4);
  const float4 g2 = rc_ld4(P.g2 + lane * 4), be2 = rc_ld4(P.be2 + lane * 4);
  float bo[2], b2[2];
  rc_bias<16>(P.b_o, bo, wave, lane);
  rc_bias<16>(P.b_2, b2, wave, lane);
  uint4 wb[RC_NB][8];
  rc_prefetch<RC_E, 16>(P.w_o, wb, wave, lane);
  __syncthreads();
  {
    f32x4_t acc[2];
    rc_gemm<RC_E, 16, RC_A0LD>(A0, P.w_o, wb, acc, wave, lane);
    rc_epi_residual(F, acc, bo, wave, lane);
  }
  rc_prefetch<RC_E, RC_FF / 16>(P.w_1, wb, wave, lane);
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {                     // LN1: x2 stays in F (the FFN's residual), its bf16 copy is the FFN's operand
    const int r = 2 * wave + rr;
    const float4 o = rc_layernorm_row(F + r * RC_FLD, g1, be1, P.eps, lane);
    *reinterpret_cast<float4*>(F + r * RC_FLD + lane * 4) = o;
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + lane * 4) = rc_pack4(o);
  }
  __syncthreads();
  {
    constexpr int TW = RC_FF / 16 / RC_WAVES;
    float b1[TW];
    rc_bias<RC_FF / 16>(P.b_1, b1, wave, lane);
    f32x4_t acc[TW];
    rc_gemm<RC_E, RC_FF / 16, RC_A0LD>(A0, P.w_1, wb, acc, wave, lane);
    rc_prefetch<RC_FF, 16>(P.w_2, wb, wave, lane);
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const int c = (wave + RC_WAVES * i) * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) A1[(4 * (lane >> 4) + j) * RC_A1LD + c] = f32_to_bf16(fmaxf(acc[i][j] + b1[i], 0.f));
    }
  }
  __syncthreads();
  {
    f32x4_t acc[2];
    rc_gemm<RC_FF, 16, RC_A1LD>(A1, P.w_2, wb, acc, wave, lane);
    rc_epi_residual(F, acc, b2, wave, lane);
  }
  if (P.w_qkv) rc_prefetch<2 * RC_E, 3 * RC_E / 16>(P.w_qkv, wb, wave, lane);
  __syncthreads();                                     // F complete; A0 and A1 free
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr, row = row0 + r;
    const float4 o = rc_layernorm_row(F + r * RC_FLD, g2, be2, P.eps, lane);
    if (row < P.M) *reinterpret_cast<float4*>(P.out + (long)row * P.ldo + lane * 4) = o;
    const float4 op = make_float4(o.x + pos[rr].x, o.y + pos[rr].y, o.z + pos[rr].z, o.w + pos[rr].w);
    const uint2 bp = rc_pack4(op), bq = rc_pack4(o);
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + lane * 4) = bp;
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + RC_E + lane * 4) = bq;
    if (P.xop && row < P.M) {
      *reinterpret_cast<uint2*>(P.xop + (long)row * P.ldxop + lane * 4) = bp;
      *reinterpret_cast<uint2*>(P.xop + (long)row * P.ldxop + RC_E + lane * 4) = bq;
    }
  }
  if (!P.w_qkv) return;                                // uniform over the launch
  __syncthreads();
  rc_qkv_tail(A0, A1, P.w_qkv, P.b_qkv, wb, P.qkv, P.ldqkv, row0, P.M, tid);
}

struct RowQkvParams {
  const float* x; int ldx;              // (M, E) layer input rows (the previous layer's LN2 output)
  const float* qpos; int ldq;
  const uint4* w_qkv; const float* b_qkv;
  bf16_t* qkv; int ldqkv;
  int M;
};

// The in-projection alone: qkv = [x + pos | x] W_qkv^T + b with the operand built exactly as far3d_rowchain_ffn builds it from its
// LN2 rows and the GEMM / store code shared with that kernel's tail -- given the same x and pos the result is BIT-IDENTICAL to the
// tail's.  The query-sharded decoder needs it: the layer outputs of all ranks exist only after the exchange.
__global__ __launch_bounds__(RC_THREADS) void rowchain_qkv_kernel(RowQkvParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* A0 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES);
  bf16_t* A1 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES + RC_A0_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RC_R;
  uint4 wb[RC_NB][8];
  rc_prefetch<2 * RC_E, 3 * RC_E / 16>(P.w_qkv, wb, wave, lane);
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr, rowc = min(row0 + r, P.M - 1);
    float4 o = rc_ld4(P.x + (long)rowc * P.ldx + lane * 4);
    const float4 pos = rc_ld4(P.qpos + (long)rowc * P.ldq + lane * 4);
    const bool ok = row0 + r < P.M;
    o.x = ok ? o.x : 0.f; o.y = ok ? o.y : 0.f; o.z = ok ? o.z : 0.f; o.w = ok ? o.w : 0.f;
    const float4 op = make_float4(o.x + pos.x, o.y + pos.y, o.z + pos.z, o.w + pos.w);
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + lane * 4) = rc_pack4(op);
    *reinterpret_cast<uint2*>(A0 + r * RC_A0LD + RC_E + lane * 4) = rc_pack4(o);
  }
  __syncthreads();
  rc_qkv_tail(A0, A1, P.w_qkv, P.b_qkv, wb, P.qkv, P.ldqkv, row0, P.M, tid);
}

struct RowBranchParams {
  const bf16_t* h; int ldh;             // (M, E) decoder outputs of all layers, bf16
  const uint4* w_c0; const float* b_c0; const float* g_c0; const float* be_c0;
  const uint4* w_c1; const float* b_c1; const float* g_c1; const float* be_c1;
  const uint4* w_c2; const float* b_c2; int n_cls;
  const uint4* w_r0; const float* b_r0;
  const uint4* w_r1; const float* b_r1;
  const uint4* w_r2; const float* b_r2; int n_reg;
  float* cls; int ld_cls;               // (M, n_cls) out
  float* reg; int ld_reg;               // (M, n_reg) out
  int M; float eps;
};

// F[r][c] = acc + bias (no residual)
__device__ __forceinline__ void rc_epi_store(float* F, const f32x4_t (&acc)[2], const float (&bv)[2], int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (wave + RC_WAVES * i) * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) F[(4 * (lane >> 4) + j) * RC_FLD + c] = acc[i][j] + bv[i];
  }
}
// dst[r][c] = bf16(relu(acc + bias)), E columns
__device__ __forceinline__ void rc_epi_relu_bf16(bf16_t* dst, int ldd, const f32x4_t (&acc)[2], const float (&bv)[2], int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (wave + RC_WAVES * i) * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(4 * (lane >> 4) + j) * ldd + c] = f32_to_bf16(fmaxf(acc[i][j] + bv[i], 0.f));
  }
}
// the <= 32-column output of a branch's last linear (two column tiles: waves 0 and 1), straight to global
#define RC_SMALL_TILES 2
__device__ __forceinline__ void rc_epi_small(float* __restrict__ out, int ld, int n, const f32x4_t& acc, float bv, int row0, int M, int wave,
                                             int lane) {
  if (wave >= RC_SMALL_TILES) return;
  const int c = wave * 16 + (lane & 15);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = row0 + 4 * (lane >> 4) + j;
    if (c < n && row < M) out[(long)row * ld + c] = acc[j] + bv;
  }
}

// The classification and regression branches over the decoder outputs of all layers (ref models/dense_heads/farhead.py:
// 230-245 branch definitions, :646-664 application): cls = L(relu(LN(L(relu(LN(L(h))))))), reg = L(relu(L(relu(L(h))))) with
// the row block resident -- 8 launches of the unfused path in one.
__global__ __launch_bounds__(RC_THREADS) void rowchain_branches_kernel(RowBranchParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* F = reinterpret_cast<float*>(smem);
  bf16_t* A0 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES);          // columns [0, E): h; [E, 2E): the branch's current hidden rows
  bf16_t* A1 = reinterpret_cast<bf16_t*>(smem + RC_F_BYTES + RC_A0_BYTES);
  bf16_t* Hd = A0 + RC_E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RC_R;
  {
    const int rb = tid >> 5, cb = (tid & 31) * 8;
    uint4 vb = *reinterpret_cast<const uint4*>(P.h + (long)min(row0 + rb, P.M - 1) * P.ldh + cb);
    const bool ok = row0 + rb < P.M;
    vb.x = ok ? vb.x : 0u; vb.y = ok ? vb.y : 0u; vb.z = ok ? vb.z : 0u; vb.w = ok ? vb.w : 0u;
    *reinterpret_cast<uint4*>(A0 + rb * RC_A0LD + cb) = vb;
  }
  const float4 g0 = rc_ld4(P.g_c0 + lane * 4), be0 = rc_ld4(P.be_c0 + lane * 4);
  const float4 g1 = rc_ld4(P.g_c1 + lane * 4), be1 = rc_ld4(P.be_c1 + lane * 4);
  float bc0[2], bc1[2], br0[2], br1[2], bc2[1], br2[1];
  rc_bias<16>(P.b_c0, bc0, wave, lane); rc_bias<16>(P.b_c1, bc1, wave, lane);
  rc_bias<16>(P.b_r0, br0, wave, lane); rc_bias<16>(P.b_r1, br1, wave, lane);
  rc_bias<RC_SMALL_TILES>(P.b_c2, bc2, wave, lane); rc_bias<RC_SMALL_TILES>(P.b_r2, br2, wave, lane);
  uint4 wb[RC_NB][8];
  rc_prefetch<RC_E, 16>(P.w_c0, wb, wave, lane);
  __syncthreads();
  auto ln_relu = [&](const float4& g, const float4& b) {             // this wave's two rows of F -> relu(LN) -> Hd (bf16)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;
      float4 o = rc_layernorm_row(F + r * RC_FLD, g, b, P.eps, lane);
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      *reinterpret_cast<uint2*>(Hd + r * RC_A0LD + lane * 4) = rc_pack4(o);
    }
  };
  {                                                                    // ---- classification branch
    f32x4_t acc[2];
    rc_gemm<RC_E, 16, RC_A0LD>(A0, P.w_c0, wb, acc, wave, lane);
    rc_prefetch<RC_E, 16>(P.w_c1, wb, wave, lane);
    rc_epi_store(F, acc, bc0, wave, lane);
  }
  __syncthreads();
  ln_relu(g0, be0);
  __syncthreads();
  {
    f32x4_t acc[2];
    rc_gemm<RC_E, 16, RC_A0LD>(Hd, P.w_c1, wb, acc, wave, lane);
    rc_prefetch<RC_E, RC_SMALL_TILES>(P.w_c2, wb, wave, lane);
    rc_epi_store(F, acc, bc1, wave, lane);                            // F was last read before the previous barrier
  }
  __syncthreads();                                                     // F complete; reads of Hd done
  ln_relu(g1, be1);
  __syncthreads();
  {
    f32x4_t acc[1];
    rc_gemm<RC_E, RC_SMALL_TILES, RC_A0LD>(Hd, P.w_c2, wb, acc, wave, lane);
    rc_prefetch<RC_E, 16>(P.w_r0, wb, wave, lane);
    rc_epi_small(P.cls, P.ld_cls, P.n_cls, acc[0], bc2[0], row0, P.M, wave, lane);
  }
  {                                                                    // ---- regression branch (h is still in A0[:, :E])
    f32x4_t acc[2];
    rc_gemm<RC_E, 16, RC_A0LD>(A0, P.w_r0, wb, acc, wave, lane);
    rc_prefetch<RC_E, 16>(P.w_r1, wb, wave, lane);
    rc_epi_relu_bf16(A1, RC_A1LD, acc, br0, wave, lane);
  }
  __syncthreads();                                                     // A1 complete; every wave is past its reads of Hd
  {
    f32x4_t acc[2];
    rc_gemm<RC_E, 16, RC_A1LD>(A1, P.w_r1, wb, acc, wave, lane);
    rc_prefetch<RC_E, RC_SMALL_TILES>(P.w_r2, wb, wave, lane);
    rc_epi_relu_bf16(Hd, RC_A0LD, acc, br1, wave, lane);
  }
  __syncthreads();
  {
    f32x4_t acc[1];
    rc_gemm<RC_E, RC_SMALL_TILES, RC_A0LD>(Hd, P.w_r2, wb, acc, wave, lane);
    rc_epi_small(P.reg, P.ld_reg, P.n_reg, acc[0], br2[0], row0, P.M, wave, lane);
  }
}

static bool rc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int far3d_rowchain_attn_out(const void* att, int ld_att, const float* x, int ldx, const float* qpos, int ldq,
                                       const void* w_out, const float* b_out, const float* g0, const float* be0,
                                       const void* w_wl, const float* b_wl, int n_wl, float* x1, int ldx1, float* ul, int ldu,
                                       const int32_t* ul_rows, int M, float eps, void* stream) {
  FAR3D_CHECK_ARG(att && x && qpos && w_out && b_out && g0 && be0 && w_wl && b_wl && x1 && ul, "far3d_rowchain_attn_out: null pointer");
  FAR3D_CHECK_ARG(M >= 0, "far3d_rowchain_attn_out: M=%d", M);
  FAR3D_CHECK_ARG(n_wl > 16 * (RC_WL_TILES - 1) && n_wl <= 16 * RC_WL_TILES && ldu >= n_wl,
                  "far3d_rowchain_attn_out: n_wl=%d must be in (%d, %d] (ldu=%d)", n_wl, 16 * (RC_WL_TILES - 1), 16 * RC_WL_TILES, ldu);
  FAR3D_CHECK_ARG(ld_att % 8 == 0 && ldx % 4 == 0 && ldq % 4 == 0 && ldx1 % 4 == 0 && ldu % 4 == 0 && ld_att >= RC_E && ldx >= RC_E &&
                      ldq >= RC_E && ldx1 >= RC_E,
                  "far3d_rowchain_attn_out: row strides must cover %d columns in 16-byte units", RC_E);
  FAR3D_CHECK_ARG(rc_aligned16(att) && rc_aligned16(x) && rc_aligned16(qpos) && rc_aligned16(w_out) && rc_aligned16(w_wl) &&
                      rc_aligned16(x1) && rc_aligned16(ul) && rc_aligned16(g0) && rc_aligned16(be0),
                  "far3d_rowchain_attn_out: pointers must be 16-byte aligned");
  if (M == 0) return FAR3D_OK;
  RowAttnOutParams P;
  P.att = (const bf16_t*)att; P.ld_att = ld_att; P.x = x; P.ldx = ldx; P.qpos = qpos; P.ldq = ldq;
  P.w_out = (const uint4*)w_out; P.b_out = b_out; P.g0 = g0; P.be0 = be0;
  P.w_wl = (const uint4*)w_wl; P.b_wl = b_wl; P.n_wl = n_wl; P.x1 = x1; P.ldx1 = ldx1; P.ul = ul; P.ldu = ldu; P.ul_rows = (const int*)ul_rows; P.M = M; P.eps = eps;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int rc_ = far3d_allow_lds(reinterpret_cast<const void*>(&rowchain_attn_out_kernel), RC_LDS, lds_ok, "rowchain_attn_out_kernel")) return rc_;
  hipLaunchKernelGGL(rowchain_attn_out_kernel, dim3((M + RC_R - 1) / RC_R), dim3(RC_THREADS), RC_LDS, (hipStream_t)stream, P);
  FAR3D_CHECK_LAUNCH("far3d_rowchain_attn_out");
  return FAR3D_OK;
}

extern "C" int far3d_rowchain_ffn(const void* agg, int ld_agg, const float* x1, int ldx1, const float* qpos, int ldq,
                                  const void* w_o, const float* b_o, const float* g1, const float* be1,
                                  const void* w_1, const float* b_1, const void* w_2, const float* b_2,
                                  const float* g2, const float* be2, const void* w_qkv, const float* b_qkv,
                                  float* out, int ldo, void* qkv, int ldqkv, void* xop, int ldxop, int M, float eps, void* stream) {
  FAR3D_CHECK_ARG(agg && x1 && qpos && w_o && b_o && g1 && be1 && w_1 && b_1 && w_2 && b_2 && g2 && be2 && out, "far3d_rowchain_ffn: null pointer");
  FAR3D_CHECK_ARG(M >= 0, "far3d_rowchain_ffn: M=%d", M);
  FAR3D_CHECK_ARG(!w_qkv || (b_qkv && qkv && ldqkv >= 3 * RC_E && ldqkv % 8 == 0 && rc_aligned16(qkv) && rc_aligned16(w_qkv)),
                  "far3d_rowchain_ffn: w_qkv needs a 16-byte aligned qkv output with ldqkv >= %d", 3 * RC_E);
  FAR3D_CHECK_ARG(!xop || (ldxop >= 2 * RC_E && ldxop % 4 == 0 && (reinterpret_cast<uintptr_t>(xop) & 7) == 0),
                  "far3d_rowchain_ffn: xop needs ldxop >= %d", 2 * RC_E);
  FAR3D_CHECK_ARG(ld_agg % 8 == 0 && ldx1 % 4 == 0 && ldq % 4 == 0 && ldo % 4 == 0 && ld_agg >= RC_E && ldx1 >= RC_E && ldo >= RC_E &&
                      ldq >= RC_E,
                  "far3d_rowchain_ffn: row strides must cover %d columns in 16-byte units", RC_E);
  FAR3D_CHECK_ARG(rc_aligned16(agg) && rc_aligned16(x1) && rc_aligned16(qpos) && rc_aligned16(w_o) && rc_aligned16(w_1) &&
                      rc_aligned16(w_2) && rc_aligned16(out) && rc_aligned16(g1) && rc_aligned16(be1) && rc_aligned16(g2) && rc_aligned16(be2),
                  "far3d_rowchain_ffn: pointers must be 16-byte aligned");
  if (M == 0) return FAR3D_OK;
  RowFfnParams P;
  P.agg = (const bf16_t*)agg; P.ld_agg = ld_agg; P.x1 = x1; P.ldx1 = ldx1; P.qpos = qpos; P.ldq = ldq;
  P.w_o = (const uint4*)w_o; P.b_o = b_o; P.g1 = g1; P.be1 = be1; P.w_1 = (const uint4*)w_1; P.b_1 = b_1;
  P.w_2 = (const uint4*)w_2; P.b_2 = b_2; P.g2 = g2; P.be2 = be2; P.w_qkv = (const uint4*)w_qkv; P.b_qkv = b_qkv;
  P.out = out; P.ldo = ldo; P.qkv = (bf16_t*)qkv; P.ldqkv = ldqkv; P.xop = (bf16_t*)xop; P.ldxop = ldxop; P.M = M; P.eps = eps;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int rc_ = far3d_allow_lds(reinterpret_cast<const void*>(&rowchain_ffn_kernel), RC_LDS, lds_ok, "rowchain_ffn_kernel")) return rc_;
  hipLaunchKernelGGL(rowchain_ffn_kernel, dim3((M + RC_R - 1) / RC_R), dim3(RC_THREADS), RC_LDS, (hipStream_t)stream, P);
  FAR3D_CHECK_LAUNCH("far3d_rowchain_ffn");
  return FAR3D_OK;
}

extern "C" int far3d_rowchain_branches(const void* h, int ldh, const void* w_c0, const float* b_c0, const float* g_c0, const float* be_c0,
                                       const void* w_c1, const float* b_c1, const float* g_c1, const float* be_c1,
                                       const void* w_c2, const float* b_c2, int n_cls, const void* w_r0, const float* b_r0,
                                       const void* w_r1, const float* b_r1, const void* w_r2, const float* b_r2, int n_reg,
                                       float* cls, int ld_cls, float* reg, int ld_reg, int M, float eps, void* stream) {
  FAR3D_CHECK_ARG(h && w_c0 && b_c0 && g_c0 && be_c0 && w_c1 && b_c1 && g_c1 && be_c1 && w_c2 && b_c2 && w_r0 && b_r0 && w_r1 && b_r1 &&
                      w_r2 && b_r2 && cls && reg,
                  "far3d_rowchain_branches: null pointer");
  FAR3D_CHECK_ARG(M >= 0 && n_cls >= 1 && n_cls <= 16 * RC_SMALL_TILES && n_reg >= 1 && n_reg <= 16 * RC_SMALL_TILES && ld_cls >= n_cls &&
                      ld_reg >= n_reg,
                  "far3d_rowchain_branches: M=%d n_cls=%d n_reg=%d (1..%d outputs per branch)", M, n_cls, n_reg, 16 * RC_SMALL_TILES);
  FAR3D_CHECK_ARG(ldh % 8 == 0 && ldh >= RC_E, "far3d_rowchain_branches: ldh=%d must be a multiple of 8, >= %d", ldh, RC_E);
  FAR3D_CHECK_ARG(rc_aligned16(h) && rc_aligned16(w_c0) && rc_aligned16(w_c1) && rc_aligned16(w_c2) && rc_aligned16(w_r0) &&
                      rc_aligned16(w_r1) && rc_aligned16(w_r2) && rc_aligned16(g_c0) && rc_aligned16(be_c0) && rc_aligned16(g_c1) &&
                      rc_aligned16(be_c1),
                  "far3d_rowchain_branches: pointers must be 16-byte aligned");
  if (M == 0) return FAR3D_OK;
  RowBranchParams P;
  P.h = (const bf16_t*)h; P.ldh = ldh;
  P.w_c0 = (const uint4*)w_c0; P.b_c0 = b_c0; P.g_c0 = g_c0; P.be_c0 = be_c0;
  P.w_c1 = (const uint4*)w_c1; P.b_c1 = b_c1; P.g_c1 = g_c1; P.be_c1 = be_c1;
  P.w_c2 = (const uint4*)w_c2; P.b_c2 = b_c2; P.n_cls = n_cls;
  P.w_r0 = (const uint4*)w_r0; P.b_r0 = b_r0; P.w_r1 = (const uint4*)w_r1; P.b_r1 = b_r1;
  P.w_r2 = (const uint4*)w_r2; P.b_r2 = b_r2; P.n_reg = n_reg;
  P.cls = cls; P.ld_cls = ld_cls; P.reg = reg; P.ld_reg = ld_reg; P.M = M; P.eps = eps;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int rc_ = far3d_allow_lds(reinterpret_cast<const void*>(&rowchain_branches_kernel), RC_LDS, lds_ok, "rowchain_branches_kernel")) return rc_;
  hipLaunchKernelGGL(rowchain_branches_kernel, dim3((M + RC_R - 1) / RC_R), dim3(RC_THREADS), RC_LDS, (hipStream_t)stream, P);
  FAR3D_CHECK_LAUNCH("far3d_rowchain_branches");
  return FAR3D_OK;
}

extern "C" int far3d_rowchain_qkv(const float* x, int ldx, const float* qpos, int ldq, const void* w_qkv, const float* b_qkv,
                                  void* qkv, int ldqkv, int M, void* stream) {
  FAR3D_CHECK_ARG(x && qpos && w_qkv && b_qkv && qkv, "far3d_rowchain_qkv: null pointer");
  FAR3D_CHECK_ARG(M >= 0 && ldx % 4 == 0 && ldq % 4 == 0 && ldx >= RC_E && ldq >= RC_E && ldqkv >= 3 * RC_E && ldqkv % 8 == 0,
                  "far3d_rowchain_qkv: M=%d, row strides %d / %d / %d", M, ldx, ldq, ldqkv);
  FAR3D_CHECK_ARG(rc_aligned16(x) && rc_aligned16(qpos) && rc_aligned16(w_qkv) && rc_aligned16(qkv),
                  "far3d_rowchain_qkv: pointers must be 16-byte aligned");
  if (M == 0) return FAR3D_OK;
  RowQkvParams P;
  P.x = x; P.ldx = ldx; P.qpos = qpos; P.ldq = ldq; P.w_qkv = (const uint4*)w_qkv; P.b_qkv = b_qkv; P.qkv = (bf16_t*)qkv; P.ldqkv = ldqkv;
  P.M = M;
  static std::atomic<unsigned long long> lds_ok{0};
  if (const int rc_ = far3d_allow_lds(reinterpret_cast<const void*>(&rowchain_qkv_kernel), RC_LDS, lds_ok, "rowchain_qkv_kernel")) return rc_;
  hipLaunchKernelGGL(rowchain_qkv_kernel, dim3((M + RC_R - 1) / RC_R), dim3(RC_THREADS), RC_LDS, (hipStream_t)stream, P);
  FAR3D_CHECK_LAUNCH("far3d_rowchain_qkv");
  return FAR3D_OK;
}
