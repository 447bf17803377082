// Implicit-GEMM convolution / linear layer on the CDNA4 matrix cores.
//
//   Out[m][p] = epilogue( sum_k W[m][k] * X[p][k] )      m: output channel, p: output pixel (n,oy,ox),
//                                                        k: (tap, input channel), tap-major.
// One kernel serves the VoVNet 3x3 / 1x1 convolutions (folded BN + ReLU), the FPN, the YOLOX towers and
// every nn.Linear of the decoder (a Linear is a 1x1 "conv" over rows).  Activations are NHWC so that a
// K-slice of one pixel is contiguous; reading / writing through (pointer, pixel-stride) pairs lets an OSA
// block's five 3x3 convs write straight into their channel slice of the concat buffer (no torch.cat).
//
// gfx950 mapping: 256 threads = 4 waves in a 2x2 grid; each wave owns WM x WN tiles of 32x32 computed with
// v_mfma_f32_32x32x16_bf16 (TC = bf16) or the exact-fp32 v_mfma_f32_32x32x2_f32 (TC = float, parity mode).
// The weight tile is the MFMA A operand (rows -> output channels) so every lane ends up holding 4
// consecutive channels of one pixel per accumulator quad -> 8/16-byte stores.  BK = 32; LDS rows are padded
// by 16 B which makes the 16-lane ds_read_b128 groups conflict-free (row stride 80 B / 144 B).
// Global -> register -> LDS staging with the next tile's loads in flight during the MFMAs, two LDS buffers,
// one barrier per K step.
#pragma once
#include "common.hpp"
#include <stdlib.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_SWISH 2

// Timing-only ablations of the pipelined kernels (tools/conv_ablation.py builds libfar3d_hip_abl<k>.so with -DFAR3D_ABLATE=k;
// the results are WRONG by construction, only the launch time means something).  The shipped library is built with 0.
// A bit mask:  1 no epilogue   2 no LDS-DMA inside the K loop   4 no MFMA (and no fragment reads)   8 no fragment reads (MFMA on the
//   step's first fragments)   16 no barrier / vmcnt wait at the top of a step;  e.g. 27 = the bare MFMA loop
#ifndef FAR3D_ABLATE
#define FAR3D_ABLATE 0
#endif
struct IgemmParams;
template <int WM, int WN, typename PT>
__device__ __forceinline__ void ablate_epilogue(const PT& P, f32x16_t (&acc)[WM][WN]) {   // keeps the MFMAs alive
  float ssum = 0.f;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ssum += acc[i][j][r];
  if (ssum == 123456.789f) reinterpret_cast<float*>(P.y)[0] = ssum;
}

struct IgemmParams {
  const void* x;        // input activations (TIn), NHWC with pixel stride ldx, channel offset pre-applied
  const void* w;        // packed weights (TC) [Cout_pad][taps][cin_pad]
  const float* bias;    // [Cout_pad] or null
  void* y;              // output, dtype y_dt, pixel stride ldy
  void* y2;             // optional second output: y2 = scale[n][m]*v + shift[n][m]  (camera-aware MLN)
  const float* y2_scale;
  const float* y2_shift;
  const void* res;      // optional residual, NHWC (Hr x Wr, nearest-neighbour upsampled to Ho x Wo)
  long x_img_stride, y_img_stride, y2_img_stride, res_img_stride;  // elements between images
  int N, H, W, Cin, ldx;
  int Ho, Wo, Cout, ldy;
  int KH, KW, stride, pad;
  int cin_pad, nsteps;
  int act, y_dt, y2_dt, ldy2;
  int res_dt, ldr, Hr, Wr;
  int x_vec, y_vec, y2_vec, res_vec;
  int y_rows16;         // plain bf16 output whose pixel rows take 16-byte stores (LDS-transposed coalesced epilogue)
  long long* chan_sums; // optional [N][Cout] fixed-point sums over the pixels of the STORED output (GEMM kernels, rows16 epilogue)
  int sums_hw;          // Ho * Wo (>= the pixel tile: a tile then spans at most two images)
#ifdef FAR3D_PROFILING
  unsigned long long* prof;   // tools/conv_phase_times.py: 8 x 64-bit stamps per workgroup (never in libfar3d_hip.so)
#endif
};

// Profiling build only (-DFAR3D_PROFILING, libfar3d_hip_prof.so): thread 0 of every workgroup records s_memtime at the phase boundaries
// -- slot 0 hardware id (HW_ID | XCC_ID << 32), 1 kernel entry, 2 set-up done (addresses, descriptors, first DMA issued), 3 first step's
// operands in LDS, 4 K loop done, 5 epilogue stores retired, 6 s_memrealtime at entry (100 MHz: clock rate and cross-CU ordering).
#ifdef FAR3D_PROFILING
#define FAR3D_CONV_TS(P, slot)                                                                                          \
  do {                                                                                                                  \
    if ((P).prof && threadIdx.x == 0) {                                                                                 \
      const unsigned long long wg_ = (unsigned long long)blockIdx.x + (unsigned long long)blockIdx.y * gridDim.x;        \
      (P).prof[wg_ * 8 + (slot)] = (unsigned long long)__builtin_amdgcn_s_memtime();                                    \
    }                                                                                                                   \
  } while (0)
#define FAR3D_CONV_TS_ENTRY(P)                                                                                          \
  do {                                                                                                                  \
    if ((P).prof && threadIdx.x == 0) {                                                                                 \
      const unsigned long long wg_ = (unsigned long long)blockIdx.x + (unsigned long long)blockIdx.y * gridDim.x;        \
      unsigned hw_, xcc_;                                                                                               \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                 \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                               \
      (P).prof[wg_ * 8 + 0] = (unsigned long long)hw_ | ((unsigned long long)xcc_ << 32);                               \
      (P).prof[wg_ * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_memrealtime();                                     \
      (P).prof[wg_ * 8 + 1] = (unsigned long long)__builtin_amdgcn_s_memtime();                                         \
    }                                                                                                                   \
  } while (0)
#define FAR3D_CONV_TS_FIRST(P) do { wait_vmcnt_all_(); __syncthreads(); FAR3D_CONV_TS(P, 3); } while (0)
#define FAR3D_CONV_TS_END(P) do { wait_vmcnt_all_(); FAR3D_CONV_TS(P, 5); } while (0)
__device__ __forceinline__ void wait_vmcnt_all_() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
#define FAR3D_CONV_TS(P, slot) do { } while (0)
#define FAR3D_CONV_TS_ENTRY(P) do { } while (0)
#define FAR3D_CONV_TS_FIRST(P) do { } while (0)
#define FAR3D_CONV_TS_END(P) do { } while (0)
#endif

// pixel index -> (image, pixel of the image).  p < 2^31 (checked at launch): a 32-bit division -- the 64-bit one this replaced costs
// ~150 VALU instructions per call, and the row-store epilogue of the GEMM kernels called it once per 16-byte piece (16-32 times per thread
// of a 256 x 256 tile: a third of that epilogue, tools/conv_phase_times.py, round 5)
__device__ __forceinline__ void pix_split(long p, int HoWo, int& n, int& rem) {
  const unsigned u = (unsigned)p;
  n = (int)(u / (unsigned)HoWo);
  rem = (int)(u - (unsigned)n * (unsigned)HoWo);
}

// Bytes of the packed weight rows a BM-row tile at row m0 may touch: the packed matrix holds ceil(Cout / 128) * 128 rows (ops.PackedConv),
// so a 256-row tile whose second half lies past them must not describe it -- rows beyond the descriptor read as zeros.
__device__ __forceinline__ long w_tile_bytes(int BM, int m0, int Cout, int Ktot) {
  const int rows = ((Cout + 127) / 128) * 128 - m0;
  return (long)(rows < BM ? (rows > 0 ? rows : 0) : BM) * Ktot * 2;
}

template <typename TC> struct Cfg;
template <> struct Cfg<bf16_t> { static constexpr int E = 8, ROWB = 80, KSUB = 2; };
template <> struct Cfg<float> { static constexpr int E = 4, ROWB = 144, KSUB = 4; };
// fp32 data computed as a two-term bf16 split: x = hi + lo (both round-to-nearest-even bf16, 16 significant bits together),
// x*w ~= hi*hi' + hi*lo' + lo*hi' on the bf16 MFMA with fp32 accumulation (the dropped lo*lo' term is 2^-16 relative).
// 3 bf16 MFMAs of K=16 replace 16 fp32 MFMAs of K=2: ~5x the fp32 matrix rate at ~1e-5 relative operand error.
// LDS rows: [32 hi | 32 lo] bf16 (same 128 B as 32 floats) + 16 B padding.  Activations are split once, while staging;
// weights are split when they are packed (host side) and arrive in the LDS row layout.
struct split_t { float v; };
template <> struct Cfg<split_t> { static constexpr int E = 4, ROWB = 144, KSUB = 2; };
template <typename TC> struct is_split { static constexpr bool value = false; };
template <> struct is_split<split_t> { static constexpr bool value = true; };
template <typename T> struct is_pair { static constexpr bool value = false; };
template <> struct is_pair<pair_t> { static constexpr bool value = true; };   // activations already stored split (common.hpp)

// 4 floats -> (4 hi bf16, 4 lo bf16)
__device__ __forceinline__ void split4(const u32x4_t& c, uint2& h, uint2& l) {
  const float x0 = __uint_as_float(c.x), x1 = __uint_as_float(c.y), x2 = __uint_as_float(c.z), x3 = __uint_as_float(c.w);
  h.x = pack_bf16x2(x0, x1); h.y = pack_bf16x2(x2, x3);
  l.x = pack_bf16x2(x0 - __uint_as_float(h.x << 16), x1 - __uint_as_float(h.x & 0xffff0000u));
  l.y = pack_bf16x2(x2 - __uint_as_float(h.y << 16), x3 - __uint_as_float(h.y & 0xffff0000u));
}

// Load E consecutive input channels (E = elements per 16-B LDS chunk) converting TIn -> TC.
template <typename TIn, typename TC>
__device__ __forceinline__ u32x4_t load_chunk(const TIn* src, int nvalid, bool vec) {
  constexpr int E = Cfg<TC>::E;
  u32x4_t r = {0u, 0u, 0u, 0u};
  if (nvalid <= 0) return r;
  if (vec && nvalid >= E) {
    if constexpr (sizeof(TIn) == sizeof(TC)) {
      r = *reinterpret_cast<const u32x4_t*>(src);
    } else {  // float -> bf16
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 b = *reinterpret_cast<const float4*>(src + 4);
      r.x = pack_bf16x2(a.x, a.y); r.y = pack_bf16x2(a.z, a.w);
      r.z = pack_bf16x2(b.x, b.y); r.w = pack_bf16x2(b.z, b.w);
    }
    return r;
  }
  float v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = e < nvalid ? LoadCvt<TIn>::ld(src + e) : 0.f;
  if constexpr (sizeof(TC) == 4) {
    r.x = __float_as_uint(v[0]); r.y = __float_as_uint(v[1]); r.z = __float_as_uint(v[2]); r.w = __float_as_uint(v[3]);
  } else {
    r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]); r.w = pack_bf16x2(v[6], v[7]);
  }
  return r;
}

template <typename TC>
__device__ __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (sizeof(TC) == 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
}

// off: element offset of the pixel in the output; m: logical channel of v[0].  Pair storage (FAR3D_DT_BF16_PAIR: strides count
// bf16 elements, Cout % 32 == 0 so nv == 4): hi at chan_off(m), lo 32 elements further.
__device__ __forceinline__ void store4(void* base, int dt, long off, int m, const float* v, int nv, bool vec) {
  if (dt == FAR3D_DT_F32) {
    float* p = reinterpret_cast<float*>(base) + off + m;
    if (vec && nv == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) p[e] = v[e];
    }
  } else if (dt == FAR3D_DT_BF16_PAIR) {
    store4(reinterpret_cast<pair_t*>(base) + off + chan_off<pair_t>(m), make_float4(v[0], v[1], v[2], v[3]));
  } else {
    bf16_t* p = reinterpret_cast<bf16_t*>(base) + off + m;
    if (vec && nv == 4) *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) p[e] = f32_to_bf16(v[e]);
    }
  }
}

// Shared epilogue: lane (l31, hi) of wave (wm, wn) holds, per 32x32 tile (i, j), 4 consecutive channels x 1 pixel per quad.
template <int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_px(const IgemmParams& P, f32x16_t (&acc)[WM][WN], const int (&pn)[WN],
                                                  const int (&ppix)[WN], int m0, int wm, int hi) {
  // ---- epilogue: bias -> activation -> residual -> store (+ optional modulated second output)
  // pn[j] / ppix[j]: image index and linear pixel index (oy*Wo+ox) of this lane's pixel in tile column j (pn < 0: none)
  long yoff[WN], roff[WN], y2off[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = pn[j] < 0 ? 0 : pn[j], pix = ppix[j];
    yoff[j] = (long)n * P.y_img_stride + (long)pix * P.ldy;
    y2off[j] = (long)n * P.y2_img_stride + (long)pix * P.ldy2;
    roff[j] = 0;
    if (P.res && pn[j] >= 0) {
      const int oy = pix / P.Wo, ox = pix - oy * P.Wo;
      const int ry = (P.Hr == P.Ho) ? oy : min((int)floorf(oy * ((float)P.Hr / P.Ho)), P.Hr - 1);
      const int rx = (P.Wr == P.Wo) ? ox : min((int)floorf(ox * ((float)P.Wr / P.Wo)), P.Wr - 1);
      roff[j] = (long)n * P.res_img_stride + ((long)ry * P.Wr + rx) * P.ldr;
    }
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + (wm * WM + i) * 32 + 8 * q + 4 * hi;
      if (m >= P.Cout) continue;
      const int nv = min(4, P.Cout - m);
      // bias rows are padded (include/far3d_hip.h) and 16-byte aligned: one vector load, hoisted out of the pixel loop
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + m);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        if (pn[j] < 0) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        if (P.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (P.act == ACT_SWISH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.f / (1.f + expf(-v[e])));
        }
        if (P.res) {
          if (P.res_dt == FAR3D_DT_F32) {
            const float* rp = reinterpret_cast<const float*>(P.res) + roff[j] + m;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) v[e] += rp[e];
          } else if (P.res_dt == FAR3D_DT_BF16_PAIR) {
            const float4 r4 = load4(reinterpret_cast<const pair_t*>(P.res) + roff[j] + chan_off<pair_t>(m));
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
          } else {
            const bf16_t* rp = reinterpret_cast<const bf16_t*>(P.res) + roff[j] + m;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) v[e] += bf16_to_f32(rp[e]);
          }
        }
        store4(P.y, P.y_dt, yoff[j], m, v, nv, P.y_vec != 0);
        if (P.y2) {
          const long so = (long)pn[j] * P.Cout + m;
          float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) u[e] = P.y2_scale[so + e] * v[e] + P.y2_shift[so + e];
          store4(P.y2, P.y2_dt, y2off[j], m, u, nv, P.y2_vec != 0);
        }
      }
    }
  }
}

template <int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& P, f32x16_t (&acc)[WM][WN], int p0, int m0, int wm, int wn,
                                               int l31, int hi, int HoWo, long Npix) {
  int pn[WN], ppix[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const long p = (long)p0 + (wn * WN + j) * 32 + l31;
    pn[j] = -1; ppix[j] = 0;
    if (p < Npix) pix_split(p, HoWo, pn[j], ppix[j]);
  }
  igemm_epilogue_px<WM, WN>(P, acc, pn, ppix, m0, wm, hi);
}

// Coalesced epilogue for plain bf16 outputs: the workgroup's BM x BPX tile is transposed through LDS (rows of BM channels,
// +16 B of padding: conflict-free 8-byte writes from the MFMA layout, 16-byte row reads) and leaves as 16-byte stores
// that cover whole pixel rows (the MFMA layout alone gives 8-byte pieces 2*ldy bytes apart).  pix_off(pl) -> element
// offset of local pixel pl in y, or -1.  smem must hold BPX * (2*BM + 16) bytes and is free once every wave has passed
// the first barrier.
// Channel sums (P.chan_sums, GEMM kernels): every STORED element v (bf16, or the hi and the lo half of a pair) adds
// rint(v * 2^FAR3D_SUMS_FRAC_BITS) to a 64-bit integer sum of its (image, channel).  Integer addition is associative, so the result
// does not depend on tile shapes, on which workgroup finishes first or on how many images a launch holds: run-to-run and
// sharded-vs-single-rank results are bit-identical although the sums are accumulated with atomics (fp32 partial sums would depend
// on the grouping).  The quantisation is per element (|error| <= 2^-19; a bf16 value >= 2^-10 is represented exactly).
// sums_p0: flattened index of the tile's first pixel; lsum_off: byte offset of a [2][BM] int64 scratch behind the staged tile.
__device__ __forceinline__ void sums_add8(int (&s)[8], float& mx, const u32x4_t& ch) {
  const unsigned w[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = __uint_as_float(w[k] << 16), b = __uint_as_float(w[k] & 0xffff0000u);
    mx = fmaxf(mx, fmaxf(fabsf(a), fabsf(b)));
    s[2 * k] += __float2int_rn(a * (float)(1 << FAR3D_SUMS_FRAC_BITS));
    s[2 * k + 1] += __float2int_rn(b * (float)(1 << FAR3D_SUMS_FRAC_BITS));
  }
}
__device__ __forceinline__ void sums_add8_wide(long long (&s)[8], const u32x4_t& ch) {     // any finite magnitude (rare path)
  const unsigned w[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s[2 * k] += __double2ll_rn((double)__uint_as_float(w[k] << 16) * (double)(1 << FAR3D_SUMS_FRAC_BITS));
    s[2 * k + 1] += __double2ll_rn((double)__uint_as_float(w[k] & 0xffff0000u) * (double)(1 << FAR3D_SUMS_FRAC_BITS));
  }
}

template <int NW, int WM, int WN, int BM, int BPX, bool PO = false, typename PixFn>   // PO: pair-storage output ([32 hi | 32 lo] blocks)
__device__ __forceinline__ void epilogue_rows16(const IgemmParams& P, unsigned char* smem, f32x16_t (&acc)[WM][WN], int m0,
                                                int wm, int wn, int l31, int hi, PixFn pix_off, long sums_p0 = -1, int lsum_off = 0) {
  constexpr int EB = PO ? 4 : 2;                       // bytes per logical channel of a staged row
  constexpr int RS = BM * EB + 16, CPP = BM * EB / 16, NT = 64 * NW;
  // a thread keeps one 16-byte chunk column over its pixels, and the lanes of a wave that share a column are a power-of-two stride apart
  // (every GEMM tile that takes sums; launch_gemm1x1_pipe refuses the others)
  constexpr bool SUMS_OK = NT % CPP == 0 && (CPP & (CPP - 1)) == 0 && CPP <= 64;
  const bool do_sums = SUMS_OK && P.chan_sums != nullptr && sums_p0 >= 0;     // workgroup-uniform
  long long* lsum = reinterpret_cast<long long*>(smem + lsum_off);
  if (do_sums)
    for (int i = threadIdx.x; i < 2 * BM; i += NT) lsum[i] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = (wm * WM + i) * 32 + 8 * q + 4 * hi;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + m0 + cl);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int pl = (wn * WN + j) * 32 + l31;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        if (P.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (P.act == ACT_SWISH) {
          if constexpr (PO) {              // 16 significant bits are kept: full-precision exp / divide
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.f / (1.f + expf(-v[e])));
          } else {                         // bf16 output: hardware exp2 / rcp (rel. error ~1e-6, far below the bf16 rounding)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * __frcp_rn(1.f + __expf(-v[e]));
          }
        }
        if constexpr (PO) {
          uint2 h, l;
          split4f(v[0], v[1], v[2], v[3], h, l);
          unsigned char* d = smem + pl * RS + (cl >> 5) * 128 + (cl & 31) * 2;
          *reinterpret_cast<uint2*>(d) = h;
          *reinterpret_cast<uint2*>(d + 64) = l;
        } else {
          *reinterpret_cast<uint2*>(smem + pl * RS + cl * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
      }
    }
  }
  __syncthreads();
  bf16_t* y = reinterpret_cast<bf16_t*>(P.y);
  constexpr int ES = EB / 2;                            // stored bf16 elements per logical channel
  // The copy loops below run in batches of RB pieces per thread: the RB staged rows are read unconditionally (clamped index) BEFORE any
  // of them is stored, so that RB LDS reads are in flight together -- one piece at a time the loop is a chain of LDS round trips, and the
  // stamps of tools/conv_phase_times.py put this epilogue at a fifth (3x3) to a third (256 x 256 GEMM) of a workgroup's life.
  // Measured NEUTRAL on the backbone (4.395 ms before and after: profiles/r5/bench.json vs bench_epilogue_batched.json): the second resident workgroup's
  // K loop already covers this epilogue's LDS latency.  Kept because it is not slower and bounds the chain for single-workgroup CUs.
  constexpr int RBATCH = 4, NPIECE = BPX * CPP;
  if (!do_sums) {
    for (int base = threadIdx.x; base < NPIECE; base += RBATCH * NT) {
      u32x4_t ch[RBATCH];
      long off[RBATCH];
      int cu[RBATCH];
#pragma unroll
      for (int u = 0; u < RBATCH; ++u) {
        const int idx = min(base + u * NT, NPIECE - 1);
        const int pl = idx / CPP, c = idx - pl * CPP;
        ch[u] = *reinterpret_cast<const u32x4_t*>(smem + pl * RS + c * 16);
        off[u] = (base + u * NT < NPIECE && m0 * ES + c * 8 < P.Cout * ES) ? pix_off(pl) : -1L;
        cu[u] = c;
      }
#pragma unroll
      for (int u = 0; u < RBATCH; ++u)
        if (off[u] >= 0) *reinterpret_cast<u32x4_t*>(y + off[u] + m0 * ES + cu[u] * 8) = ch[u];
    }
    return;
  }
  // ---- stores + channel sums.  NT % CPP == 0: a thread keeps ONE 16-byte chunk column c (8 stored channels) over all its pixels
  if constexpr (SUMS_OK) {
  const int cam0 = (int)(sums_p0 / P.sums_hw);
  const long bnd = (long)(cam0 + 1) * P.sums_hw;       // first pixel of the next image; a tile holds at most two (sums_hw >= BPX)
  const int nslots = sums_p0 + BPX > bnd ? 2 : 1;       // workgroup-uniform
  for (int slot = 0; slot < nslots; ++slot) {
    int s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mx = 0.f;
    int cc = -1;
    for (int base = threadIdx.x; base < NPIECE; base += RBATCH * NT) {
      u32x4_t ch[RBATCH];
      long off[RBATCH];
      int cu[RBATCH], plu[RBATCH];
#pragma unroll
      for (int u = 0; u < RBATCH; ++u) {
        const int idx = min(base + u * NT, NPIECE - 1);
        const int pl = idx / CPP, c = idx - pl * CPP;
        ch[u] = *reinterpret_cast<const u32x4_t*>(smem + pl * RS + c * 16);
        off[u] = (base + u * NT < NPIECE && m0 * ES + c * 8 < P.Cout * ES) ? pix_off(pl) : -1L;
        cu[u] = c; plu[u] = pl;
      }
#pragma unroll
      for (int u = 0; u < RBATCH; ++u) {
        if (off[u] < 0) continue;
        if (slot == 0) *reinterpret_cast<u32x4_t*>(y + off[u] + m0 * ES + cu[u] * 8) = ch[u];
        if ((sums_p0 + plu[u] >= bnd) == (slot == 1)) { sums_add8(s, mx, ch[u]); cc = cu[u]; }
      }
    }
    // a magnitude the 32-bit per-wave sums cannot hold (|v| >= 2^(24 - FRAC_BITS) = 64 over up to 64 pixels): redo this slot in 64 bits
    const bool wide = __any(mx >= (float)(1 << (24 - FAR3D_SUMS_FRAC_BITS)));
    if (wide) {                                                                   // wave-uniform, practically never taken
      long long sw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int idx = threadIdx.x; idx < BPX * CPP; idx += NT) {
        const int pl = idx / CPP, c = idx - pl * CPP;
        if (m0 * ES + c * 8 >= P.Cout * ES || pix_off(pl) < 0) continue;
        if ((sums_p0 + pl >= bnd) == (slot == 1)) sums_add8_wide(sw, *reinterpret_cast<const u32x4_t*>(smem + pl * RS + c * 16));
      }
      if (cc >= 0) {
        const int chn = PO ? (cc >> 3) * 32 + (cc & 3) * 8 : cc * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (sw[e]) atomicAdd(reinterpret_cast<unsigned long long*>(lsum + slot * BM + chn + e), (unsigned long long)sw[e]);
      }
    } else {
      // lanes with the same chunk column (lane % CPP) add up first: one LDS atomic per (wave, channel)
#pragma unroll
      for (int o = 32; o >= CPP && o >= 1; o >>= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += __shfl_xor(s[e], o);
      }
      const int lane = threadIdx.x & 63;
      const int c = threadIdx.x % CPP;               // == cc wherever this thread saw a pixel
      if (lane < CPP && m0 * ES + c * 8 < P.Cout * ES) {
        const int chn = PO ? (c >> 3) * 32 + (c & 3) * 8 : c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (s[e]) atomicAdd(reinterpret_cast<unsigned long long*>(lsum + slot * BM + chn + e), (unsigned long long)(long long)s[e]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nslots * BM; i += NT) {
    const int slot = i / BM, chn = i - slot * BM, cam = cam0 + slot;
    const long long v = lsum[i];
    if (v != 0 && cam < P.N && m0 + chn < P.Cout)
      atomicAdd(reinterpret_cast<unsigned long long*>(P.chan_sums + (long)cam * P.Cout + m0 + chn), (unsigned long long)v);
  }
  }
}

// WGM x WGN: arrangement of the 4 waves over (channels, pixels); WM x WN: 32x32 MFMA tiles per wave.
template <typename TIn, typename TC, int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams P) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int E = Cfg<TC>::E, ROWB = Cfg<TC>::ROWB, KSUB = Cfg<TC>::KSUB;
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr int CPR = 32 / E;                 // 16-B chunks per tile row
  constexpr int A_CH = BM * CPR / 256;        // weight chunks per thread
  constexpr int B_CH = BP * CPR / 256;        // pixel chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                               // [2][BM][ROWB]
  unsigned char* Bs = smem + 2 * BM * ROWB;               // [2][BP][ROWB]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  const int p0 = blockIdx.x * BP, m0 = blockIdx.y * BM;
  const int HoWo = P.Ho * P.Wo;
  const long Npix = (long)P.N * HoWo;
  const int Ktot = P.KH * P.KW * P.cin_pad;

  // ---- per-thread staging coordinates (fixed for the whole K loop)
  constexpr bool SPLIT = is_split<TC>::value;
  constexpr bool PAIR_IN = is_pair<TIn>::value;
  static_assert(sizeof(TC) == 2 || sizeof(TIn) == 4 || (PAIR_IN && SPLIT), "fp32 compute types take fp32 (or pair-stored) activations");
  static_assert(!PAIR_IN || SPLIT, "pair-stored activations feed the split products");
  const TC* wsrc[A_CH];
  int arow[A_CH], acol[A_CH];
#pragma unroll
  for (int c = 0; c < A_CH; ++c) {
    const int id = t + c * 256;
    arow[c] = id / CPR; acol[c] = id % CPR;
    wsrc[c] = reinterpret_cast<const TC*>(P.w) + (long)(m0 + arow[c]) * Ktot + acol[c] * E;
  }
  int brow[B_CH], bcol[B_CH], bn[B_CH], boy[B_CH], box[B_CH];
#pragma unroll
  for (int c = 0; c < B_CH; ++c) {
    const int id = t + c * 256;
    brow[c] = id / CPR; bcol[c] = id % CPR;
    const long p = (long)p0 + brow[c];
    if (p < Npix) {
      int n, rem;
      pix_split(p, HoWo, n, rem);
      bn[c] = n; boy[c] = (rem / P.Wo) * P.stride - P.pad; box[c] = (rem % P.Wo) * P.stride - P.pad;
    } else {
      bn[c] = -1; boy[c] = 0; box[c] = 0;
    }
  }
  u32x4_t areg[A_CH], breg[B_CH];
  auto gload = [&](int step) __attribute__((always_inline)) {
    const int kb = step * 32;
    const int tap = kb / P.cin_pad, c0 = kb - tap * P.cin_pad;
    const int ky = tap / P.KW, kx = tap - ky * P.KW;
#pragma unroll
    for (int c = 0; c < A_CH; ++c) areg[c] = *reinterpret_cast<const u32x4_t*>(wsrc[c] + kb);
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      const int iy = boy[c] + ky, ix = box[c] + kx, ch = c0 + bcol[c] * E;
      const bool ok = bn[c] >= 0 && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      if constexpr (PAIR_IN) {      // 4 hi + 4 lo bf16 of 4 logical channels, already split (Cin % 32 == 0: no ragged chunk)
        u32x4_t r = {0u, 0u, 0u, 0u};
        if (ok && ch < P.Cin) {
          const pair_t* src = reinterpret_cast<const pair_t*>(P.x) + (long)bn[c] * P.x_img_stride + ((long)iy * P.W + ix) * P.ldx + chan_off<pair_t>(ch);
          const uint2 h = *reinterpret_cast<const uint2*>(src), l = *reinterpret_cast<const uint2*>(src + 32);
          r.x = h.x; r.y = h.y; r.z = l.x; r.w = l.y;
        }
        breg[c] = r;
      } else {
        const TIn* src = reinterpret_cast<const TIn*>(P.x) + (long)bn[c] * P.x_img_stride + ((long)iy * P.W + ix) * P.ldx + ch;
        breg[c] = load_chunk<TIn, TC>(src, ok ? P.Cin - ch : 0, P.x_vec != 0);
      }
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    if constexpr (SPLIT) {
      uint2 h, l;
#pragma unroll
      for (int c = 0; c < A_CH; ++c)        // weights arrive pre-split (rows of [32 hi | 32 lo] bf16 per 32-channel block): plain copy
        *reinterpret_cast<u32x4_t*>(As + (buf * BM + arow[c]) * ROWB + acol[c] * 16) = areg[c];
#pragma unroll
      for (int c = 0; c < B_CH; ++c) {
        if constexpr (PAIR_IN) { h = make_uint2(breg[c].x, breg[c].y); l = make_uint2(breg[c].z, breg[c].w); }
        else split4(breg[c], h, l);
        unsigned char* d = Bs + (buf * BP + brow[c]) * ROWB + bcol[c] * 8;
        *reinterpret_cast<uint2*>(d) = h; *reinterpret_cast<uint2*>(d + 64) = l;
      }
    } else {
#pragma unroll
      for (int c = 0; c < A_CH; ++c)
        *reinterpret_cast<u32x4_t*>(As + (buf * BM + arow[c]) * ROWB + acol[c] * 16) = areg[c];
#pragma unroll
      for (int c = 0; c < B_CH; ++c)
        *reinterpret_cast<u32x4_t*>(Bs + (buf * BP + brow[c]) * ROWB + bcol[c] * 16) = breg[c];
    }
  };

  f32x16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(0);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < P.nsteps; ++step) {
    const bool more = step + 1 < P.nsteps;
    if (more) gload(step + 1);
    const unsigned char* Ab = As + (buf * BM + wm * WM * 32 + l31) * ROWB + hi * 16;
    const unsigned char* Bb = Bs + (buf * BP + wn * WN * 32 + l31) * ROWB + hi * 16;
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
      u32x4_t af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(Ab + i * 32 * ROWB + kk * 32);
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(Bb + j * 32 * ROWB + kk * 32);
      if constexpr (SPLIT) {
        u32x4_t al[WM], bl[WN];        // the lo halves sit 64 B further in the row
#pragma unroll
        for (int i = 0; i < WM; ++i) al[i] = *reinterpret_cast<const u32x4_t*>(Ab + i * 32 * ROWB + kk * 32 + 64);
#pragma unroll
        for (int j = 0; j < WN; ++j) bl[j] = *reinterpret_cast<const u32x4_t*>(Bb + j * 32 * ROWB + kk * 32 + 64);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {       // small terms first
            mma<bf16_t>(acc[i][j], al[i], bf[j]);
            mma<bf16_t>(acc[i][j], af[i], bl[j]);
            mma<bf16_t>(acc[i][j], af[i], bf[j]);
          }
      } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) mma<TC>(acc[i][j], af[i], bf[j]);
      }
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  igemm_epilogue<WM, WN>(P, acc, p0, m0, wm, wn, l31, hi, HoWo, Npix);
}

template <typename TIn, typename TC, int WGM, int WGN, int WM, int WN>
static int launch_igemm(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  const long Npix = (long)P.N * P.Ho * P.Wo;
  dim3 grid((unsigned)((Npix + BP - 1) / BP), (unsigned)((P.Cout + BM - 1) / BM));
  const size_t lds = (size_t)2 * (BM + BP) * Cfg<TC>::ROWB;
  hipLaunchKernelGGL((igemm_kernel<TIn, TC, WGM, WGN, WM, WN>), grid, dim3(256), lds, st, P);
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// bf16 fast path: same tiling / epilogue, but the operands stream global -> LDS directly (LDS-DMA, `global_load_lds`
// 16 B per lane, no VGPR staging) through an NS-deep ring, issued NS-1 K-steps ahead with counted `s_waitcnt vmcnt` and ONE
// raw `s_barrier` per step.  An LDS-DMA instruction writes 1 KiB = 16 rows x 64 B linearly, so the 64-byte (BK = 32 bf16)
// rows are unpadded; bank conflicts of the 16-lane `ds_read_b128` groups are removed by an XOR swizzle applied on the
// SOURCE side: the lane that fills physical 16-B chunk pc of row r fetches logical chunk pc ^ ((r>>2)&3), and fragment
// reads use the same involution.  Out-of-image taps read a 64-B zero page instead of branching.
// ------------------------------------------------------------------------------------------------------------------
static __device__ __attribute__((aligned(64))) unsigned int g_zero_page[32];

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int U, int MAXK> __device__ __forceinline__ void wait_vmcnt_units(int k) {      // s_waitcnt vmcnt(U * k), k wave-uniform
  if constexpr (MAXK <= 0) {
    wait_vmcnt<0>();
  } else {
    if (k >= MAXK) wait_vmcnt<U * MAXK>();
    else wait_vmcnt_units<U, MAXK - 1>(k);
  }
}

template <int WGM, int WGN, int WM, int WN, int NS, int KPS = 1>   // KPS: 32-channel K chunks per barrier step
__global__ __launch_bounds__(256) void igemm_dma_kernel(IgemmParams P) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr int GA = BM / 16, GB = BP / 16, GT = GA + GB;      // 16-row groups = LDS-DMA instructions per chunk
  constexpr int GLW = (GT + 3) / 4;                              // per wave (slots past GT re-fetch an earlier group: benign)
  constexpr int CHUNK = (BM + BP) * 64, STAGE = CHUNK * KPS;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  const int p0 = blockIdx.x * BP, m0 = blockIdx.y * BM;
  const int HoWo = P.Ho * P.Wo;
  const long Npix = (long)P.N * HoWo;
  const int Ktot = P.KH * P.KW * P.cin_pad;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // lane -> (row within the 16-row group, physical chunk) -> logical chunk it must fetch
  const int rg = lane >> 2, pc = lane & 3;
  const int lc = pc ^ ((rg >> 2) & 3);          // group bases are multiples of 16 rows, so (row>>2)&3 == (rg>>2)&3
  const bf16_t* sbase_[GLW];                     // A slot: weight row pointer; B slot: image base (nullptr = past the end)
  int soy[GLW], sox[GLW], sdst[GLW];
  bool sisA[GLW];
#pragma unroll
  for (int i = 0; i < GLW; ++i) {
    int q = wv + 4 * i;
    if (q >= GT) q -= GT;
    sisA[i] = q < GA;
    if (sisA[i]) {
      const int row = q * 16 + rg;
      sbase_[i] = reinterpret_cast<const bf16_t*>(P.w) + (long)(m0 + row) * Ktot + lc * 8;
      soy[i] = 0; sox[i] = 0;
      sdst[i] = q * 1024;
    } else {
      const int row = (q - GA) * 16 + rg;
      const long p = (long)p0 + row;
      sdst[i] = BM * 64 + (q - GA) * 1024;
      if (p < Npix) {
        int n, rem;
      pix_split(p, HoWo, n, rem);
        sbase_[i] = reinterpret_cast<const bf16_t*>(P.x) + (long)n * P.x_img_stride + lc * 8;
        soy[i] = (rem / P.Wo) * P.stride - P.pad; sox[i] = (rem % P.Wo) * P.stride - P.pad;
      } else {
        sbase_[i] = nullptr; soy[i] = 0; sox[i] = 0;
      }
    }
  }
  int i_tap = 0, i_c0 = 0;   // (tap, first channel) of the next K chunk to be issued
  const int ntaps = P.KH * P.KW;
  auto issue = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int kc = 0; kc < KPS; ++kc) {
      const bool live = i_tap < ntaps;     // chunks past the end of K (last, partial step) are fed from the zero page
      const int ky = i_tap / P.KW, kx = i_tap - ky * P.KW;
      const int kb = i_tap * P.cin_pad + i_c0;
      unsigned char* sb = smem + stage * STAGE + kc * CHUNK;
#pragma unroll
      for (int i = 0; i < GLW; ++i) {
        const bf16_t* src;
        if (sisA[i]) {
          src = live ? sbase_[i] + kb : zero;
        } else {
          const int iy = soy[i] + ky, ix = sox[i] + kx;
          const bool ok = live && sbase_[i] != nullptr && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
          src = ok ? sbase_[i] + ((long)iy * P.W + ix) * P.ldx + i_c0 : zero;
        }
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(sb + sdst[i]), 16, 0, 0);
      }
      i_c0 += 32;
      if (i_c0 >= P.cin_pad) { i_c0 = 0; ++i_tap; }
    }
  };

  f32x16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (P.nsteps + KPS - 1) / KPS;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nsteps) issue(s);
  // fragment read addresses: row r = tile row, logical chunk (kk*2 + hi) -> physical chunk ^ ((r>>2)&3)
  int aoff[WM], boff[WN], aswz[WM], bswz[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) { const int r = (wm * WM + i) * 32 + l31; aoff[i] = r * 64; aswz[i] = (r >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < WN; ++j) { const int r = (wn * WN + j) * 32 + l31; boff[j] = BM * 64 + r * 64; bswz[j] = (r >> 2) & 3; }

  int stage = 0;
  for (int step = 0; step < nsteps; ++step) {
    if (NS > 2 && step + NS - 1 <= nsteps) wait_vmcnt<GLW * KPS * (NS > 2 ? NS - 2 : 0)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (step + NS - 1 < nsteps) {
      int st2 = stage + NS - 1; if (st2 >= NS) st2 -= NS;
      issue(st2);
    }
#pragma unroll
    for (int kc = 0; kc < KPS; ++kc) {
      const unsigned char* sb = smem + stage * STAGE + kc * CHUNK;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4_t af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(sb + aoff[i] + (((kk * 2 + hi) ^ aswz[i]) << 4));
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(sb + boff[j] + (((kk * 2 + hi) ^ bswz[j]) << 4));
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[i], bf[j]);
      }
    }
    if (++stage == NS) stage = 0;
  }
  igemm_epilogue<WM, WN>(P, acc, p0, m0, wm, wn, l31, hi, HoWo, Npix);
}

template <int WGM, int WGN, int WM, int WN, int NS, int KPS = 1>
static int launch_igemm_dma(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  const long Npix = (long)P.N * P.Ho * P.Wo;
  dim3 grid((unsigned)((Npix + BP - 1) / BP), (unsigned)((P.Cout + BM - 1) / BM));
  constexpr size_t lds = (size_t)NS * KPS * (BM + BP) * 64;
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds > 65536)
    if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&igemm_dma_kernel<WGM, WGN, WM, WN, NS, KPS>), (int)lds, lds_ok, "far3d_conv2d_nhwc")) return rc;
  hipLaunchKernelGGL((igemm_dma_kernel<WGM, WGN, WM, WN, NS, KPS>), grid, dim3(256), lds, st, P);
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 bf16 convolution with an LDS-resident input patch, software pipelined.
// A workgroup owns a TH x 32 pixel rectangle of one image: for every 32-channel slice of Cin the (TH+2) x 34 halo patch is
// DMA'd to LDS ONCE (double buffered) and the 9 taps are shifted LDS reads of it; only the weights stream, one kernel row
// (3 taps) per barrier step, through a 2- or 3-deep ring.  The first version of this kernel (global_load_lds, addresses
// recomputed per tap) ran the matrix pipe 30 % busy with waves parked 42 % of the time: the compiler issued each tap's
// `ds_read_b128`s, then `s_waitcnt lgkmcnt(0)`, then the MFMAs, six times per step, and spent ~9 VALU per MFMA on swizzled
// LDS addresses (profiles/r1).  Here
//   * every fragment address is a per-lane VGPR computed once + a compile-time immediate (chunk parity, ring stage, tap
//     are compile-time: the loop body is two chunks = six steps, fully unrolled),
//   * fragments are double-buffered in registers: the reads of tap/k-half i+1 are issued before the MFMAs of i,
//   * the next chunk's patch is issued AFTER the next step's weights so that `vmcnt(GPL)` keeps it in flight for two steps.
// ------------------------------------------------------------------------------------------------------------------
// Pair-stored activations (FAR3D_DT_BF16_PAIR, common.hpp) and pre-split weights run through the same pipeline: a 32-channel
// chunk is then TWO 64-byte planes (hi, lo) of every patch pixel / weight row, and every (tap, k-half) does the three products
// lo*hi' + hi*lo' + hi*hi' (NT = 3) from four fragment sets -- 3 MFMAs per 2+2 fragment reads instead of 1 per 1+1, so the
// split mode is LESS LDS-bound than plain bf16.  NT = 1 with PAIR reads only the hi planes (a single-bf16 layer inside a
// pair-stored network; the per-layer precision sweep of tools/precision_sweep.py).
template <int WM, int WN, int KY, int POFF, int WOFF, int TAPB, int NTAPS, int NT, int PPS, int WPS, int STRIDE = 1, typename Dma>   // NTAPS: 3 = one kernel row, 9 = all taps
__device__ __forceinline__ void patch_step_pipelined(const unsigned char* smem, const int (&aaddr)[2][WM],
                                                     const int (&baddr)[2][STRIDE * (WN - 1) + 3][3], f32x16_t (&acc)[WM][WN], Dma dma) {
  constexpr int PL = NT == 3 ? 2 : 1;          // planes read per chunk: [0] hi (or plain bf16), [1] lo
  u32x4_t af[2][PL][WM], bf[2][PL][WN];
#pragma unroll
  for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
    for (int i = 0; i < WM; ++i) af[0][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + WOFF + pl * WPS + aaddr[0][i]);
#pragma unroll
    for (int j = 0; j < WN; ++j) bf[0][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + POFF + pl * PPS + baddr[0][STRIDE * j + KY][0]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < 2 * NTAPS; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    if (it + 1 < 2 * NTAPS && !(FAR3D_ABLATE & 8)) {
      const int tp = (it + 1) >> 1, kk = (it + 1) & 1;     // tap tp of this step = kernel row KY + tp / 3, column tp % 3
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[nxt][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + WOFF + tp * TAPB + pl * WPS + aaddr[kk][i]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[nxt][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + POFF + pl * PPS + baddr[kk][STRIDE * j + KY + tp / 3][tp % 3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next fragments' reads ahead of this iteration's MFMAs
    constexpr int fb = (FAR3D_ABLATE & 8) ? 0 : -1;      // ablation 4: every MFMA reads the step's first fragments
    const int rb = fb < 0 ? cur : fb;
    if constexpr (!(FAR3D_ABLATE & 4)) {
    if constexpr (NT == 3) {               // small terms first; term-major so that consecutive MFMAs hit different accumulators
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[rb][PL - 1][i], bf[rb][0][j]);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[rb][0][i], bf[rb][PL - 1][j]);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[rb][0][i], bf[rb][0][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(FAR3D_ABLATE & 2)) dma(it);        // this iteration's share of the next step's LDS-DMA pieces: their issue cost hides under the MFMAs above
    __builtin_amdgcn_sched_barrier(0);
  }
}

// NSW: weight ring depth (steps prefetched ahead + 1).  RPS: kernel rows per barrier step -- 1 (3 taps) or 3 (the whole 32-channel
// chunk, 9 taps: a third of the barriers, for the layers too small to fill the chip with more than one workgroup per CU).
// NT / PAIR: see patch_step_pipelined; with PAIR the pointers / strides of x, w (and of y when y_dt is the pair code) count bf16
// elements of the 2C-wide stored rows.
// STRIDE 2 (round 5; 3x3 / stride 2 / pad 1: VoVNet stem_3, the extra FPN level): the patch is (2 TH + 1) rows of 65 input columns, and
// the DMA lands every patch row DE-INTERLEAVED -- 33 even columns, then 32 odd ones (the source address of an LDS slot is free: a DMA piece
// is lane-linear on the LDS side only) -- so that the 32 output pixels of a fragment read 32 CONSECUTIVE slots for every tap (kx = 0: even
// block at l, kx = 1: odd block at l, kx = 2: even block at l + 1), exactly the conflict-free pattern of the stride-1 kernel.
template <int WGM, int WGN, int WM, int WN, int NSW = 2, int RPS = 1, int NT = 1, bool PAIR = false, int STRIDE = 1>
__global__ __launch_bounds__(64 * WGM * WGN) void conv3x3_pipe_kernel(IgemmParams P, int tiles_x, int tiles_y) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist only in the device pass
  FAR3D_CONV_TS_ENTRY(P);
  constexpr int NW = WGM * WGN;                      // waves per workgroup (4, 8 or 16)
  static_assert(STRIDE == 1 || (STRIDE == 2 && RPS == 1), "stride 2: one kernel row per step");
  constexpr int BM = 32 * WGM * WM, TH = WGN * WN, PW = STRIDE == 1 ? 34 : 66, PH = STRIDE * TH + 3 - STRIDE, PPIX = PW * PH;
  constexpr int RB = STRIDE * (WN - 1) + 3;          // patch rows a wave's WN output rows touch
  constexpr int PG = (PPIX + 15) / 16, PATCH_B = PG * 1024, GPL = (PG + NW - 1) / NW;
  constexpr int TPSN = 3 * RPS;                      // taps per barrier step
  constexpr int GA = BM / 16, WSLOTS = TPSN * GA, GWL = (WSLOTS + NW - 1) / NW, WST = BM * 64;
  constexpr int PLD = NT == 3 ? 2 : 1;               // 64-byte planes of a chunk brought to LDS
  constexpr int PLS = PAIR ? 2 : 1;                  // planes a chunk occupies in memory
  constexpr int TAPB = PLD * WST;                    // one tap of a weight stage: [plane][BM rows x 64 B]
  static_assert(RPS == 3 ? (NSW >= 2 && NSW <= 5) : (NSW == 2 || NSW == 3), "weight ring of 2 or 3 steps; whole-chunk rings of 2..5 chunks");
  static_assert(NT == 1 || (NT == 3 && PAIR && RPS == 1), "split products need pair-stored operands and one kernel row per step");
  constexpr int NPB = RPS == 3 ? NSW : 2;            // patch buffers: whole-chunk steps keep a patch per ring stage
  constexpr int WBASE = NPB * PLD * PATCH_B;         // patches: [NPB buffers][PLD planes][PATCH_B]; then the weight ring [NSW][TPSN][TAPB]
  static_assert(WBASE + NSW * TPSN * TAPB <= 163840, "LDS budget");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int x0 = tx * 32, y0 = ty * TH, m0 = blockIdx.y * BM;
  const int Ktot = 9 * P.cin_pad * PLS;
  const int rg = lane >> 2, pc = lane & 3;
  const int lc = pc ^ ((rg >> 2) & 3);

  // LDS-DMA through buffer descriptors (`buffer_load_dwordx4 ... offen lds`): per-lane byte offset in a VGPR computed once,
  // the K position in an SGPR, out-of-image halo pixels get an out-of-range offset and the hardware writes zeros.
  // (MUBUF rather than `global_load_lds`: the compiler treats the latter as a FLAT access that may touch LDS and
  // then degrades every `s_waitcnt lgkmcnt(n)` to lgkmcnt(0), which serialises the fragment pipeline below.)
  constexpr unsigned OOB = 0x80000000u;
  const long img_bytes = (long)P.H * P.W * P.ldx * 2;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const bf16_t*>(P.x) + (long)n * P.x_img_stride), 0, (int)(img_bytes < 0x7fffffffL ? img_bytes : 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const bf16_t*>(P.w) + (long)m0 * Ktot), 0, (int)w_tile_bytes(BM, m0, P.Cout, Ktot), 0x00020000);
  unsigned pvoff[GPL], wvoff[GWL];
  int pdst[GPL], wdst[GWL];
#pragma unroll
  for (int i = 0; i < GPL; ++i) {       // every wave issues exactly GPL patch slots (slots past PG re-fetch an earlier one)
    int q = wv + NW * i;
    if (q >= PG) q -= PG;
    pdst[i] = q * 1024;
    const int idx = q * 16 + rg;
    const int py = idx / PW, px = idx - py * PW;
    // stride 2: slot px of a patch row holds input column 2 px (px < 33: the even block) or 2 (px - 33) + 1 (the odd block; slot 65 is unused)
    const int pcol = STRIDE == 1 ? px : (px < 33 ? 2 * px : 2 * (px - 33) + 1);
    const int iy = STRIDE * y0 - 1 + py, ix = STRIDE * x0 - 1 + pcol;
    const bool ok = idx < PPIX && pcol <= 64 && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
    pvoff[i] = ok ? (unsigned)((((long)iy * P.W + ix) * P.ldx + lc * 8) * 2) : OOB;
  }
  int wtap[GWL];                        // weight slot = (tap of the kernel row, 16-row group); slots past WSLOTS are skipped
#pragma unroll
  for (int i = 0; i < GWL; ++i) {
    const int q = wv + NW * i;
    wtap[i] = q < WSLOTS ? q / GA : -1;
    const int g = q % GA;
    wdst[i] = (q / GA) * TAPB + g * 1024;
    wvoff[i] = (unsigned)(((g * 16 + rg) * Ktot + lc * 8) * 2);
  }
  // a slot = PLD pieces (the chunk's planes are 64 bytes apart in memory, PATCH_B / WST apart in LDS)
  auto patch_piece = [&](int i, int pl, int chunk, int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(smem + (buf * PLD + pl) * PATCH_B + pdst[i]), 16, pvoff[i],
                                             chunk * (64 * PLS) + pl * 64, 0, 0);
  };
  auto w_piece = [&](int i, int pl, int chunk, int ky, int stage) __attribute__((always_inline)) {   // kernel row ky: taps 3*ky .. 3*ky+2
    if (wtap[i] < 0) return;            // wave-uniform
    const int kb = ((ky * 3 + wtap[i]) * P.cin_pad + chunk * 32) * PLS;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + WBASE + stage * TPSN * TAPB + wdst[i] + pl * WST), 16, wvoff[i],
                                             kb * 2 + pl * 64, 0, 0);
  };
  auto issue_patch = [&](int chunk, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < GPL; ++i)
#pragma unroll
      for (int pl = 0; pl < PLD; ++pl) patch_piece(i, pl, chunk, buf);
  };
  auto issue_w = [&](int chunk, int ky, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < GWL; ++i)
#pragma unroll
      for (int pl = 0; pl < PLD; ++pl) w_piece(i, pl, chunk, ky, stage);
  };

  f32x16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addresses (bytes from the plane base): [k-half][...]; the k-half flips bit 5 of the swizzled chunk
  int aaddr[2][WM], baddr[2][RB][3];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    aaddr[0][i] = r * 64 + ((hi ^ ((r >> 2) & 3)) << 4);
    aaddr[1][i] = aaddr[0][i] ^ 32;
  }
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int slot = STRIDE == 1 ? l31 + kx : (kx == 1 ? 33 + l31 : l31 + (kx >> 1));
      const int idx = (STRIDE * wn * WN + r) * PW + slot;
      baddr[0][r][kx] = idx * 64 + ((hi ^ ((idx >> 2) & 3)) << 4);
      baddr[1][r][kx] = baddr[0][r][kx] ^ 32;
    }

  const int nchunks = P.cin_pad / 32;
  // `s_waitcnt vmcnt` counts this wave's own pieces, oldest first: PLD * nwv weight pieces per step (nwv = GWL or GWL-1 slots),
  // PLD * GPL patch pieces
  const int nwv = (WSLOTS - wv + NW - 1) / NW;
  const bool wfull = nwv == GWL;
#define FAR3D_WAITC(NWS, PA)                                                                                          \
  {                                                                                                                    \
    if (wfull) { if (PA) wait_vmcnt<PLD * ((NWS) * GWL + GPL)>(); else wait_vmcnt<PLD * (NWS) * GWL>(); }              \
    else       { if (PA) wait_vmcnt<PLD * ((NWS) * (GWL - 1) + GPL)>(); else wait_vmcnt<PLD * (NWS) * (GWL - 1)>(); }  \
  }
#define FAR3D_DMA_PIECES(DO_W, WCALL, DO_P, PCALL)                                                                     \
        _Pragma("unroll")                                                                                              \
        for (int k = it * PPI; k < (it + 1) * PPI && k < PIECES; ++k) {                                                \
          if (k < GWL * PLD) { if (DO_W) { const int sl = k / PLD, pl = k % PLD; WCALL; } }                            \
          else if (DO_P) { const int sl = (k - GWL * PLD) / PLD, pl = (k - GWL * PLD) % PLD; PCALL; }                  \
        }
  if constexpr (RPS == 3) {
    // whole-chunk steps over an NSW-deep ring: chunk c lives in stage c % NSW = [patch | all 9 taps]; NSW - 1 chunks are in flight
    // ahead of the one being computed and a step waits for the OLDEST only (counted vmcnt: a wave's LDS-DMAs retire in order).
    // The K loop of the small maps is 5-7 chunks of ~0.3 us of MFMA work each against ~1.5 us of L2 / Infinity-Cache latency per
    // DMA round trip: with the 2-deep ring every step waited for a round trip (stage 4/5 of VoV-99 ran at 640 / 310 TF/s); with
    // most of the K loop issued up front the launch pays the latency once (profiles/r4).  The barrier at the top of step c also
    // says that every wave is done with step c - 1, whose stage chunk c + NSW - 1 goes into.
    constexpr int AHEAD = NSW - 1;
#pragma unroll
    for (int c = 0; c < AHEAD; ++c)
      if (c < nchunks) { issue_w(c, 0, c); issue_patch(c, c); }
    FAR3D_CONV_TS(P, 2);
    FAR3D_CONV_TS_FIRST(P);
#define FAR3D_PIPE_CHUNK(STG, CH)                                                                                      \
  {                                                                                                                    \
    const bool more = (CH) + AHEAD < nchunks;                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if constexpr (!(FAR3D_ABLATE & 16)) {                                                                              \
    int ahead_ = nchunks - 1 - (CH);                        /* chunks issued after CH: they may stay in flight */      \
    if (ahead_ > AHEAD - 1) ahead_ = AHEAD - 1;                                                                        \
    if (wfull) wait_vmcnt_units<PLD * (GWL + GPL), AHEAD - 1>(ahead_);                                                 \
    else wait_vmcnt_units<PLD * (GWL - 1 + GPL), AHEAD - 1>(ahead_);                                                   \
    __builtin_amdgcn_s_barrier();                                                                                      \
    }                                                                                                                  \
    asm volatile("" ::: "memory");                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    constexpr int NXT = ((STG) + AHEAD) % NSW;                                                                         \
    constexpr int PIECES = (GWL + GPL) * PLD, PPI = (PIECES + 17) / 18;                                                \
    patch_step_pipelined<WM, WN, 0, (STG) * PLD * PATCH_B, WBASE + (STG) * TPSN * TAPB, TAPB, 9, NT, PATCH_B, WST, STRIDE>(    \
      smem, aaddr, baddr, acc, [&](int it) __attribute__((always_inline)) {                                            \
        FAR3D_DMA_PIECES(more, w_piece(sl, pl, (CH) + AHEAD, 0, NXT), more, patch_piece(sl, pl, (CH) + AHEAD, NXT))    \
      });                                                                                                              \
  }
    for (int c = 0; c < nchunks; c += NSW) {
      FAR3D_PIPE_CHUNK(0, c)
      if (c + 1 < nchunks) FAR3D_PIPE_CHUNK(1, c + 1)
      if constexpr (NSW > 2) { if (c + 2 < nchunks) FAR3D_PIPE_CHUNK(2, c + 2) }
      if constexpr (NSW > 3) { if (c + 3 < nchunks) FAR3D_PIPE_CHUNK(3, c + 3) }
      if constexpr (NSW > 4) { if (c + 4 < nchunks) FAR3D_PIPE_CHUNK(4, c + 4) }
    }
#undef FAR3D_PIPE_CHUNK
  } else {
  if (NSW == 2) {
    issue_patch(0, 0);
    issue_w(0, 0, 0);
  } else {
    issue_w(0, 0, 0);
    issue_patch(0, 0);
    issue_w(0, 1, 1);
  }
  FAR3D_CONV_TS(P, 2);
  FAR3D_CONV_TS_FIRST(P);
  // step s = 3c + ky reads weight stage s % NSW and patch buffer c & 1; after its barrier it issues the weights of step
  // s + NSW - 1 and (ky == 0) the next chunk's patch.  Pieces issued after W(s), which may stay in flight at the top of
  // step s:  NSW 2: the patch when ky == 1;  NSW 3: W(s+1), plus the patch when ky != 0.
#define FAR3D_PIPE_STEP(PAR, KY, CH)                                                                                   \
  {                                                                                                                    \
    constexpr int STG = NSW == 2 ? ((PAR * 3 + KY) & 1) : KY;                                                          \
    const bool more = (CH) + 1 < nchunks;                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if constexpr (!(FAR3D_ABLATE & 16)) {                                                                                 \
    if (NSW == 2) {                                                                                                    \
      if (KY == 1 && more) wait_vmcnt<PLD * GPL>(); else wait_vmcnt<0>();                                              \
    } else {                                                                                                           \
      const bool have_next = KY < 2 || more, patch_after = KY != 0 && more;                                            \
      if (have_next) FAR3D_WAITC(1, patch_after) else wait_vmcnt<0>();                                                 \
    }                                                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                                      \
    }                                                                                                                  \
    asm volatile("" ::: "memory");                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    /* next pieces: weights of step s + NSW - 1 (chunk wc, row wk, stage ws) first, then (ky == 0) the next patch */   \
    const int wc = (NSW == 2 ? (KY < 2) : (KY == 0)) ? (CH) : (CH) + 1;                                                \
    constexpr int wk = (KY + NSW - 1) % 3, ws = NSW == 2 ? (STG ^ 1) : wk;                                             \
    const bool do_w = wc < nchunks, do_p = KY == 0 && more;                                                            \
    constexpr int PIECES = (GWL + GPL) * PLD, PPI = (PIECES + 5) / 6;                                                  \
    patch_step_pipelined<WM, WN, KY, (PAR) * PLD * PATCH_B, WBASE + STG * 3 * TAPB, TAPB, 3, NT, PATCH_B, WST, STRIDE>(        \
      smem, aaddr, baddr, acc, [&](int it) __attribute__((always_inline)) {                                            \
        FAR3D_DMA_PIECES(do_w, w_piece(sl, pl, wc, wk, ws), do_p, patch_piece(sl, pl, (CH) + 1, (PAR) ^ 1))            \
      });                                                                                                              \
  }
  for (int c = 0; c < nchunks; c += 2) {
    FAR3D_PIPE_STEP(0, 0, c)
    FAR3D_PIPE_STEP(0, 1, c)
    FAR3D_PIPE_STEP(0, 2, c)
    if (c + 1 < nchunks) {
      FAR3D_PIPE_STEP(1, 0, c + 1)
      FAR3D_PIPE_STEP(1, 1, c + 1)
      FAR3D_PIPE_STEP(1, 2, c + 1)
    }
  }
#undef FAR3D_PIPE_STEP
  }   // RPS == 1
#undef FAR3D_WAITC
#undef FAR3D_DMA_PIECES
  FAR3D_CONV_TS(P, 4);
  if constexpr (FAR3D_ABLATE & 1) { ablate_epilogue<WM, WN>(P, acc); return; }
  if (P.y_rows16) {
    epilogue_rows16<NW, WM, WN, BM, TH * 32, PAIR>(P, smem, acc, m0, wm, wn, l31, hi, [&](int pl) -> long {
      const int y = y0 + (pl >> 5), x = x0 + (pl & 31);
      return (y < P.Ho && x < P.Wo) ? (long)n * P.y_img_stride + ((long)y * P.Wo + x) * P.ldy : -1L;
    });
    FAR3D_CONV_TS_END(P);
    return;
  }
  int pn[WN], ppix[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int y = y0 + wn * WN + j, x = x0 + l31;
    pn[j] = (y < P.Ho && x < P.Wo) ? n : -1;
    ppix[j] = y * P.Wo + x;
  }
  igemm_epilogue_px<WM, WN>(P, acc, pn, ppix, m0, wm, hi);
#endif
}

template <int WGM, int WGN, int WM, int WN, int NSW = 2, int RPS = 1, int NT = 1, bool PAIR = false, int STRIDE = 1>
static int launch_conv3x3_pipe(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, TH = WGN * WN, PLD = NT == 3 ? 2 : 1;
  constexpr int PG = ((STRIDE == 1 ? 34 : 66) * (STRIDE * TH + 3 - STRIDE) + 15) / 16;
  constexpr size_t lds_ring = (size_t)(RPS == 3 ? NSW : 2) * PLD * PG * 1024 + (size_t)NSW * 3 * RPS * PLD * BM * 64;
  constexpr size_t lds_out = (size_t)TH * 32 * (BM * (PAIR ? 4 : 2) + 16);
  constexpr size_t lds = lds_ring > lds_out ? lds_ring : lds_out;
  static_assert(lds <= 163840, "LDS budget");
  const int tiles_x = (P.Wo + 31) / 32, tiles_y = (P.Ho + TH - 1) / TH;
  dim3 grid((unsigned)(P.N * tiles_x * tiles_y), (unsigned)((P.Cout + BM - 1) / BM));
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds > 65536)
    if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&conv3x3_pipe_kernel<WGM, WGN, WM, WN, NSW, RPS, NT, PAIR, STRIDE>), (int)lds, lds_ok, "far3d_conv2d_nhwc")) return rc;
  hipLaunchKernelGGL((conv3x3_pipe_kernel<WGM, WGN, WM, WN, NSW, RPS, NT, PAIR, STRIDE>), grid, dim3(64 * WGM * WGN), lds, st, P, tiles_x, tiles_y);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 1x1 / stride 1 convolution (= GEMM  y[pix][ch] = sum_k x[pix][k] w[ch][k]) with the same pipeline as conv3x3_pipe_kernel:
// buffer-descriptor LDS-DMA (one VGPR offset per piece, K position in an SGPR, pixels past the end read as zeros),
// 64 channels of K per barrier step (two 32-channel sub-tiles in the proven 64-byte-row XOR-swizzled layout), 2-deep ring,
// fragments double-buffered in registers with counted lgkmcnt, 4 / 8 / 16 waves per workgroup.
// Workgroups are numbered so that one XCD (round-robin dispatch: id % 8) owns a pixel tile for ALL of its channel tiles
// back to back: the activation tile is fetched from HBM once and re-read from that XCD's L2.
// ------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int SOFF, int BM, int SUBB, int NIT>
__device__ __forceinline__ void gemm_step_pipelined(const unsigned char* smem, const int (&aaddr)[2][WM],
                                                    const int (&baddr)[2][WN], f32x16_t (&acc)[WM][WN]) {
  // iteration it = (sub-tile it>>1, k-half it&1); sub-tile s lives at SOFF + s * SUBB, activations after the BM weight rows
  u32x4_t af[2][WM], bf[2][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) af[0][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + aaddr[0][i]);
#pragma unroll
  for (int j = 0; j < WN; ++j) bf[0][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + baddr[0][j]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    if (it + 1 < NIT && !(FAR3D_ABLATE & 8)) {
      const int sb = ((it + 1) >> 1) * SUBB, kk = (it + 1) & 1;
#pragma unroll
      for (int i = 0; i < WM; ++i) af[nxt][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + sb + aaddr[kk][i]);
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[nxt][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + sb + baddr[kk][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int rb = (FAR3D_ABLATE & 8) ? 0 : cur;
    if constexpr (!(FAR3D_ABLATE & 4)) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[rb][i], bf[rb][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Split products on pair-stored operands: the two 64-byte sub-tiles of a step are the hi and the lo plane of ONE 32-channel block
// (the DMA pattern is that of the bf16 kernel); per k-half: lo*hi' + hi*lo' + hi*hi'.
template <int WM, int WN, int SOFF, int BM, int SUBB>
__device__ __forceinline__ void gemm_step_split(const unsigned char* smem, const int (&aaddr)[2][WM], const int (&baddr)[2][WN],
                                                f32x16_t (&acc)[WM][WN]) {
  u32x4_t af[2][2][WM], bf[2][2][WN];       // [register buffer][plane][tile]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
    for (int i = 0; i < WM; ++i) af[0][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + pl * SUBB + aaddr[0][i]);
#pragma unroll
    for (int j = 0; j < WN; ++j) bf[0][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + pl * SUBB + baddr[0][j]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    if (kk == 0) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[1][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + pl * SUBB + aaddr[1][i]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[1][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + pl * SUBB + baddr[1][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[kk][1][i], bf[kk][0][j]);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[kk][0][i], bf[kk][1][j]);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[kk][0][i], bf[kk][0][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// fp32 activation rows (F32B, round 5: the exact-data decoder GEMMs of the in-tolerance engine on the pipelined kernel): 32 floats are
// the same 128 bytes as a pair-stored 32-channel block, so the DMA pattern is the pair kernel's -- the two 64-byte sub-tiles of a step
// are floats 0-15 and 16-31 -- and only the fragment differs: the 8 floats a lane feeds into k-half kk are logical chunks 2*hi and
// 2*hi + 1 of sub-tile kk, split into their hi / lo bf16 halves in registers (round-to-nearest-even, as pair_from_float stores them).
// baddr[c][j]: chunk 2*hi + c of the lane's row in pixel tile j.  Products and their order are those of gemm_step_split.
template <int WM, int WN, int SOFF, int BM, int SUBB>
__device__ __forceinline__ void gemm_step_split_f32b(const unsigned char* smem, const int (&aaddr)[2][WM], const int (&baddr)[2][WN],
                                                     f32x16_t (&acc)[WM][WN]) {
  u32x4_t af[2][2][WM], bx[2][2][WN];       // [register buffer][plane | chunk][tile]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
    for (int i = 0; i < WM; ++i) af[0][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + pl * SUBB + aaddr[0][i]);
#pragma unroll
    for (int j = 0; j < WN; ++j) bx[0][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + baddr[pl][j]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    if (kk == 0) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[1][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + pl * SUBB + aaddr[1][i]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bx[1][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + SUBB + baddr[pl][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4_t bh[WN], bl[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      uint2 h0, l0, h1, l1;
      split4(bx[kk][0][j], h0, l0);
      split4(bx[kk][1][j], h1, l1);
      bh[j] = u32x4_t{h0.x, h0.y, h1.x, h1.y};
      bl[j] = u32x4_t{l0.x, l0.y, l1.x, l1.y};
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[kk][1][i], bh[j]);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[kk][0][i], bl[j]);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[kk][0][i], bh[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// EXACT fp32 on the pipelined kernel (F32X, round 6: the in-tolerance engine's decoder / FarHead GEMMs ran on the register-staged
// exact-fp32 kernel at 12-19 us a launch, three times their MFMA time -- no LDS-DMA ring, every K step a global-load round trip).  fp32
// rows on BOTH sides: 32 floats = 128 bytes = the DMA pattern of the pair kernel (sub-tile 0: floats 0-15, sub-tile 1: floats 16-31 of a
// 32-channel block).  A lane feeds v_mfma_f32_32x32x2_f32 with ONE float per operand and MFMA: lane (row, hi) reads logical chunks 2 hi
// and 2 hi + 1 (4 floats each) of its row in a sub-tile, on the weight side and on the activation side alike, and the eight MFMAs of a
// sub-tile pair element e of chunk c of the two half-waves -- a permutation of K, the same on both operands.  Exact products, fp32
// accumulation: only the summation order differs from the staged kernel.
template <int WM, int WN, int SOFF, int BM, int SUBB>
__device__ __forceinline__ void gemm_step_exact_f32(const unsigned char* smem, const int (&aaddr)[2][WM], const int (&baddr)[2][WN],
                                                    f32x16_t (&acc)[WM][WN]) {
  u32x4_t ax[2][2][WM], bx[2][2][WN];       // [register buffer = sub-tile][chunk c][tile]
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int i = 0; i < WM; ++i) ax[0][c][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + aaddr[c][i]);
#pragma unroll
    for (int j = 0; j < WN; ++j) bx[0][c][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + baddr[c][j]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    if (kk == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < WM; ++i) ax[1][c][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + SUBB + aaddr[c][i]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bx[1][c][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 64 + SUBB + baddr[c][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) mma<float>(acc[i][j], ax[kk][c][i], bx[kk][c][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// NT / PAIR as in conv3x3_pipe_kernel: pair-stored activations + pre-split weights; NT = 3 split products, NT = 1 hi planes only.
// NS: LDS ring depth in 64-channel steps.  NS = 2: the step that follows the one being computed is in flight (each barrier waits
// for vmcnt(0)).  NS >= 3: NS - 1 steps are in flight and a barrier waits for the OLDEST one only (counted vmcnt: the LDS-DMAs of a
// wave retire in order), so a step's HBM / L2 latency has NS - 1 steps of MFMA work to hide under -- for the K-short GEMMs of the
// decoder (4..16 steps) that is most of the K loop issued up front.

// F32B: the activations are fp32 rows handed over as if pair-stored (the caller doubles ldx / x_img_stride: same bytes), weights pre-split.
// F32X: fp32 rows on both sides, exact fp32 MFMA (gemm_step_exact_f32); instantiated with NT = 3, PAIR = true, F32B = true for the byte geometry.
// KS (round 6, F32X only): K groups INSIDE the workgroup.  The exact-fp32 MFMA runs a 32 x 32 tile through 32 floats of K in 1 024 cycles, so
// a wave that owns a tile over K = 1 024 is a 33 k-cycle serial chain (16 us) however idle the chip is -- and the decoder's GEMMs (1 544
// rows) hand out 100-400 tiles to 1 024 SIMDs.  With KS > 1 the workgroup has KS x WGM x WGN waves: group kg takes the K steps kg, kg + KS,
// ... of the SAME output tile through its own LDS ring (the groups share nothing but the workgroup barriers), the groups' partial tiles are
// added in group order through LDS and group 0 runs the epilogue.  Which K steps a group takes depends on K and KS alone: a row's bits do
// not depend on the number of rows in the launch (the query-sharded decoder reproduces the replicated one).
template <int WGM, int WGN, int WM, int WN, int NT = 1, bool PAIR = false, int NS = 2, bool F32B = false, bool F32X = false, int KS = 1>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void gemm1x1_pipe_kernel(IgemmParams P, int npt, int nct) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WGM * WGN;                                 // waves of one K group
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr int GA = BM / 16, GB = BP / 16;                     // 1 KB DMA pieces per 32-channel sub-tile
  constexpr int AW = (GA + NW - 1) / NW, BW = (GB + NW - 1) / NW;
  constexpr bool A_EXACT = GA % NW == 0, B_EXACT = GB % NW == 0;
  constexpr int SUB = (BM + BP) * 64, STAGE = 2 * SUB;
  static_assert(KS == 1 || F32X, "K groups: the exact fp32 mode only");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_all[];
  const int t = threadIdx.x, lane = t & 63, wv_all = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kg = KS > 1 ? wv_all / NW : 0, wv = KS > 1 ? wv_all % NW : wv_all;
  unsigned char* const smem = smem_all + kg * (NS * STAGE);     // this group's ring
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  // XCD-aware numbering: id % 8 = XCD; inside an XCD channel tiles are consecutive for a fixed pixel tile
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= npt) return;
  const int p0 = pt * BP, m0 = ct * BM;
  const int HoWo = P.Ho * P.Wo;
  const long Npix = (long)P.N * HoWo;
  static_assert(NT == 1 || (NT == 3 && PAIR), "split products need pair-stored operands");
  static_assert(!F32B || (NT == 3 && PAIR), "fp32 activation rows: split products, weights pre-split");
  static_assert(!F32X || F32B, "exact fp32: the fp32-row geometry");
  constexpr int PLS = PAIR ? 2 : 1;
  constexpr int KSTR = (PAIR && NT == 1) ? 128 : 64;      // bytes along K between the sub-tiles this kernel consumes
  const int Ktot = P.cin_pad * PLS;
  const int rg = lane >> 2, pc = lane & 3;
  const int lc = pc ^ ((rg >> 2) & 3);
  constexpr unsigned OOB = 0x80000000u;
  const long x_bytes = ((long)(P.N - 1) * P.x_img_stride + (long)HoWo * P.ldx) * 2;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)(x_bytes < 0x7fffffffL ? x_bytes : 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const bf16_t*>(P.w) + (long)m0 * Ktot), 0, (int)w_tile_bytes(BM, m0, P.Cout, Ktot), 0x00020000);
  unsigned avoff[AW], bvoff[BW];
#pragma unroll
  for (int i = 0; i < AW; ++i) {
    const int g = wv + NW * i;
    avoff[i] = (unsigned)(((g * 16 + rg) * Ktot + lc * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < BW; ++i) {
    const int g = wv + NW * i;
    const long p = (long)p0 + g * 16 + rg;
    if (p < Npix) {
      int n, rem;
      pix_split(p, HoWo, n, rem);
      bvoff[i] = (unsigned)(((long)n * P.x_img_stride + (long)rem * P.ldx + lc * 8) * 2);
    } else {
      bvoff[i] = OOB;
    }
  }
  // one step = NSUBS 32-channel sub-tiles starting at sub-tile index 2*step
  auto issue = [&](int step, int stage_off, int nsubs) __attribute__((always_inline)) {
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      if (sidx >= nsubs) break;
      const int kb = (2 * (step * KS + kg) + sidx) * KSTR;   // bytes along K (step: the group's own step count)
#pragma unroll
      for (int i = 0; i < AW; ++i) {
        const int g = wv + NW * i;
        if (A_EXACT || g < GA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + stage_off + sidx * SUB + g * 1024), 16, avoff[i], kb, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BW; ++i) {
        const int g = wv + NW * i;
        if (B_EXACT || g < GB)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(smem + stage_off + sidx * SUB + BM * 64 + g * 1024), 16, bvoff[i], kb, 0, 0);
      }
    }
  };

  f32x16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aaddr[2][WM], baddr[2][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    if constexpr (F32X) {     // fp32 weight rows: [c] = logical chunk 2 * hi + c, like the activation side
      aaddr[0][i] = r * 64 + (((2 * hi) ^ ((r >> 2) & 3)) << 4);
      aaddr[1][i] = aaddr[0][i] ^ 16;
    } else {
      aaddr[0][i] = r * 64 + ((hi ^ ((r >> 2) & 3)) << 4);
      aaddr[1][i] = aaddr[0][i] ^ 32;
    }
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int r = (wn * WN + j) * 32 + l31;
    if constexpr (F32B) {     // [c]: logical chunk 2 * hi + c (4 floats) of a sub-tile row
      baddr[0][j] = r * 64 + (((2 * hi) ^ ((r >> 2) & 3)) << 4);
      baddr[1][j] = baddr[0][j] ^ 16;
    } else {
      baddr[0][j] = r * 64 + ((hi ^ ((r >> 2) & 3)) << 4);
      baddr[1][j] = baddr[0][j] ^ 32;
    }
  }
  // sub-tiles of 64 bytes along K: NT 1: one per 32 channels, a step takes two; NT 3: the hi and lo plane of one 32-channel block
  const int nsub = NT == 3 ? P.cin_pad / 16 : P.cin_pad / 32;
  const int nfull_all = nsub >> 1, tail = KS > 1 ? 0 : (nsub & 1);   // full steps + an optional half one (F32X: whole blocks only)
  const int nfull = KS > 1 ? (nfull_all - kg + KS - 1) / KS : nfull_all;      // this group's steps
  constexpr int D = NS - 1;                     // steps in flight ahead of the one being computed
  constexpr int UNIT = AW + BW;                 // LDS-DMA instructions of one 32-channel sub-tile, per wave
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  static_assert(NS == 2 || (A_EXACT && B_EXACT), "counted vmcnt needs the same number of DMA instructions in every wave");
  static_assert(NS == 2 || NT == 1 || F32X, "deep rings: plain bf16 products, or the exact fp32 mode (always whole 32-channel blocks: no half step)");
  const int nsteps = nfull + tail;              // step s < nfull: two sub-tiles; step nfull (if tail): one
#pragma unroll
  for (int s = 0; s < D; ++s) {
    if (s < nfull) issue(s, s * STAGE, 2);
    else if (s < nsteps) issue(s, s * STAGE, 1);
  }
  // wait until step S has landed: the steps S+1 .. S+D-1 that were already issued may stay in flight
#define FAR3D_GEMM_SYNC(S)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if constexpr (FAR3D_ABLATE & 16) {                                                                       \
    } else if constexpr (D == 1) {                                                                           \
      wait_vmcnt<0>();                                                                                       \
    } else {                                                                                                 \
      int units_ = 0;                                                                                        \
      _Pragma("unroll")                                                                                      \
      for (int q_ = 1; q_ < D; ++q_)                                                                         \
        if ((S) + q_ < nsteps) units_ += ((S) + q_ < nfull) ? 2 : 1;                                         \
      wait_vmcnt_units<UNIT, 2 * (D - 1)>(units_);                                                           \
    }                                                                                                        \
    if constexpr (!(FAR3D_ABLATE & 16)) __builtin_amdgcn_s_barrier();                                           \
    asm volatile("" ::: "memory");                                                                           \
    __builtin_amdgcn_sched_barrier(0);
  // the barrier also says that every wave is done with step S-1, whose stage is the one step S+D goes into
#define FAR3D_GEMM_STEP(STG, S)                                                                              \
  {                                                                                                          \
    FAR3D_GEMM_SYNC(S)                                                                                       \
    if constexpr (!(FAR3D_ABLATE & 2)) {                                                                     \
    if ((S) + D < nfull) issue((S) + D, (((STG) + D) % NS) * STAGE, 2);                                      \
    else if ((S) + D < nsteps) issue((S) + D, (((STG) + D) % NS) * STAGE, 1);                                \
    }                                                                                                        \
    if constexpr (F32X) gemm_step_exact_f32<WM, WN, (STG) * STAGE, BM, SUB>(smem, aaddr, baddr, acc);        \
    else if constexpr (F32B) gemm_step_split_f32b<WM, WN, (STG) * STAGE, BM, SUB>(smem, aaddr, baddr, acc);  \
    else if constexpr (NT == 3) gemm_step_split<WM, WN, (STG) * STAGE, BM, SUB>(smem, aaddr, baddr, acc);    \
    else gemm_step_pipelined<WM, WN, (STG) * STAGE, BM, SUB, 4>(smem, aaddr, baddr, acc);                    \
  }
  for (int s0 = 0; s0 < nfull; s0 += NS) {
    FAR3D_GEMM_STEP(0, s0)
    if (s0 + 1 < nfull) FAR3D_GEMM_STEP(1, s0 + 1)
    if constexpr (NS > 2) { if (s0 + 2 < nfull) FAR3D_GEMM_STEP(2, s0 + 2) }
    if constexpr (NS > 3) { if (s0 + 3 < nfull) FAR3D_GEMM_STEP(3, s0 + 3) }
  }
  if (tail) {
    FAR3D_GEMM_SYNC(nfull)
    const int ts = nfull % NS;
    if (ts == 0)                            gemm_step_pipelined<WM, WN, 0, BM, SUB, 2>(smem, aaddr, baddr, acc);
    else if (ts == 1)                       gemm_step_pipelined<WM, WN, STAGE, BM, SUB, 2>(smem, aaddr, baddr, acc);
    else if constexpr (NS > 2) {
      if (ts == 2)                          gemm_step_pipelined<WM, WN, 2 * STAGE, BM, SUB, 2>(smem, aaddr, baddr, acc);
      else if constexpr (NS > 3)            gemm_step_pipelined<WM, WN, 3 * STAGE, BM, SUB, 2>(smem, aaddr, baddr, acc);
    }
  }
#undef FAR3D_GEMM_STEP
#undef FAR3D_GEMM_SYNC
  if constexpr (KS > 1) {
    // every wave passes the same number of barriers (the groups' step counts differ by at most one), then the rings are dead: partial
    // tiles of groups 1 .. KS-1 -> LDS [group][wave][tile register][lane], group 0 adds them in group order and stores
    if (nfull < (nfull_all + KS - 1) / KS) __builtin_amdgcn_s_barrier();
    __syncthreads();
    constexpr int PART = WM * WN * 16 * 64;
    float* red = reinterpret_cast<float*>(smem_all);
    if (kg > 0) {
      float* dst = red + ((kg - 1) * NW + wv) * PART + lane;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((i * WN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float* src = red + ((g - 1) * NW + wv) * PART + lane;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * WN + j) * 16 + r) * 64];
    }
  }
  if constexpr (FAR3D_ABLATE & 1) { ablate_epilogue<WM, WN>(P, acc); return; }
  if (KS == 1 && P.y_rows16) {
    constexpr int RING = NS * 2 * (BM + BP) * 64, OUTB = BP * (BM * (PAIR ? 4 : 2) + 16);
    int ep_n0, ep_r0;
    pix_split((long)p0, HoWo, ep_n0, ep_r0);
    epilogue_rows16<NW, WM, WN, BM, BP, PAIR>(P, smem, acc, m0, wm, wn, l31, hi, [&](int pl) -> long {
      if ((long)p0 + pl >= Npix) return -1L;
      int n = ep_n0, rem = ep_r0 + pl;                 // (image, pixel) of the tile's first pixel + pl: no division per piece
      while (rem >= HoWo) { rem -= HoWo; ++n; }
      return (long)n * P.y_img_stride + (long)rem * P.ldy;
    }, (long)p0, RING > OUTB ? RING : OUTB);
    return;
  }
  igemm_epilogue<WM, WN>(P, acc, p0, m0, wm, wn, l31, hi, HoWo, Npix);
#endif
}

// far3d_conv2d_nhwc with channel sums: what a GEMM tile needs (a refusal is an error of the call, never a silent fallback)
template <int BM, int BP, int NTHREADS>
static bool gemm_sums_ok(const IgemmParams& P, size_t lds_sums) {
  static_assert(BP / (NTHREADS / 64) <= 64, "32-bit per-wave sums hold 64 pixels of |v| < 2^(24 - FRAC_BITS)");
  if (!P.y_rows16) { far3d_set_error("far3d_conv2d_nhwc: channel sums need the coalesced bf16 / pair output path (16-byte aligned rows, no residual / second output)"); return false; }
  if (P.sums_hw < BP) { far3d_set_error("far3d_conv2d_nhwc: channel sums need Ho*Wo (%d) >= the tile's %d pixels", P.sums_hw, BP); return false; }
  if (lds_sums > 163840) { far3d_set_error("far3d_conv2d_nhwc: this tile has no LDS left for the channel-sum scratch"); return false; }
  return true;
}

template <int WGM, int WGN, int WM, int WN, int NT = 1, bool PAIR = false, int NS = 2, bool F32B = false, bool F32X = false, int KS = 1>
static int launch_gemm1x1_pipe(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  const long Npix = (long)P.N * P.Ho * P.Wo;
  const int npt = (int)((Npix + BP - 1) / BP), nct = (P.Cout + BM - 1) / BM;
  constexpr size_t lds_part = (size_t)(KS - 1) * WGM * WGN * WM * WN * 4096;      // K groups: the partial tiles of groups 1 .. KS-1
  constexpr size_t lds_ring = (size_t)KS * NS * 2 * (BM + BP) * 64 > lds_part ? (size_t)KS * NS * 2 * (BM + BP) * 64 : lds_part;
  constexpr size_t lds_out = KS > 1 ? 0 : (size_t)BP * (BM * (PAIR ? 4 : 2) + 16);
  static_assert(lds_ring <= 163840, "LDS budget");
  static_assert(64 * WGM * WGN * KS <= 1024, "workgroup size");
  constexpr size_t lds0 = lds_ring > lds_out ? lds_ring : lds_out, lds_sums = lds0 + 2 * BM * sizeof(long long);
  if constexpr (KS > 1) {
    if (P.chan_sums) { far3d_set_error("far3d_conv2d_nhwc: the K-group tiles of the exact-fp32 GEMM take no channel sums"); return FAR3D_ERR_ARG; }
  }
  const size_t lds = P.chan_sums ? lds_sums : lds0;
  if (P.chan_sums && !gemm_sums_ok<BM, BP, 64 * WGM * WGN>(P, lds_sums)) return FAR3D_ERR_ARG;
  {
    constexpr int CPP = BM * (PAIR ? 4 : 2) / 16;        // 16-byte chunks of a staged output row (epilogue_rows16)
    if (P.chan_sums && !((64 * WGM * WGN) % CPP == 0 && (CPP & (CPP - 1)) == 0 && CPP <= 64)) {
      far3d_set_error("far3d_conv2d_nhwc: this tile's %d-channel rows do not take channel sums (power-of-two channel tiles only)", BM);
      return FAR3D_ERR_ARG;
    }
  }
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds_sums > 65536)
    if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&gemm1x1_pipe_kernel<WGM, WGN, WM, WN, NT, PAIR, NS, F32B, F32X, KS>),
                                       (int)(lds_sums <= 163840 ? lds_sums : lds0), lds_ok, "far3d_conv2d_nhwc")) return rc;
  const unsigned blocks = (unsigned)((npt + 7) / 8 * 8) * (unsigned)nct;
  hipLaunchKernelGGL((gemm1x1_pipe_kernel<WGM, WGN, WM, WN, NT, PAIR, NS, F32B, F32X, KS>), dim3(blocks), dim3(64 * WGM * WGN * KS), lds, st, P, npt, nct);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The same GEMM with FULL-LINE LDS-DMA pieces: one `buffer_load ... lds` brings 8 rows x 128 contiguous bytes (64 channels) instead
// of 16 rows x 64 bytes, so a 128-byte line of x / w is requested once instead of twice (MI355X guide: half-line, fragment-shaped
// requests cost up to 2x texture-addresser time at identical L2 / HBM traffic; tools/conv_ablation.py shows these GEMMs run at the
// speed of their DMA stream alone).  LDS image: rows of 128 bytes, the 16-byte slot of global column c of tile row r is
// c ^ ((r >> 1) & 7) -- the permutation is applied to the SOURCE address (a DMA lands lane-linear), and it makes the four 16-lane
// groups of a ds_read_b128 conflict-free.  A step is one 64-channel row; K % 64 == 32 ends with a half step whose upper slots are
// fetched out of range (hardware zero fill).  Plain bf16 only.
// ------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int SOFF, int BM, int NIT>
__device__ __forceinline__ void gemm_step_wide(const unsigned char* smem, const int (&aaddr)[4][WM], const int (&baddr)[4][WN],
                                               f32x16_t (&acc)[WM][WN]) {
  u32x4_t af[2][WM], bf[2][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) af[0][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + aaddr[0][i]);
#pragma unroll
  for (int j = 0; j < WN; ++j) bf[0][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 128 + baddr[0][j]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    if (it + 1 < NIT) {
#pragma unroll
      for (int i = 0; i < WM; ++i) af[nxt][i] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + aaddr[it + 1][i]);
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[nxt][j] = *reinterpret_cast<const u32x4_t*>(smem + SOFF + BM * 128 + baddr[it + 1][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[cur][i], bf[cur][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int WGM, int WGN, int WM, int WN, int NS = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm1x1_wide_kernel(IgemmParams P, int npt, int nct) {
#if defined(__HIP_DEVICE_COMPILE__)
  FAR3D_CONV_TS_ENTRY(P);
  constexpr int NW = WGM * WGN;
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr int GA = BM / 8, GB = BP / 8;                       // 1 KB DMA pieces (8 rows x 128 B) per 64-channel step
  static_assert(GA % NW == 0 && GB % NW == 0, "every wave issues the same number of pieces (counted vmcnt)");
  constexpr int AW = GA / NW, BW = GB / NW;
  constexpr int STAGE = (BM + BP) * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;          // XCD-aware numbering as in gemm1x1_pipe_kernel
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= npt) return;
  const int p0 = pt * BP, m0 = ct * BM;
  const int HoWo = P.Ho * P.Wo;
  const long Npix = (long)P.N * HoWo;
  const int Ktot = P.cin_pad;
  const int rr = lane >> 3, sl = lane & 7;                       // row of the piece, 16-byte slot of the 128-byte LDS row
  constexpr unsigned OOB = 0x80000000u;
  const long x_bytes = ((long)(P.N - 1) * P.x_img_stride + (long)HoWo * P.ldx) * 2;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)(x_bytes < 0x7fffffffL ? x_bytes : 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const bf16_t*>(P.w) + (long)m0 * Ktot), 0, (int)w_tile_bytes(BM, m0, P.Cout, Ktot), 0x00020000);
  unsigned avoff[AW], bvoff[BW], avoff_h[AW], bvoff_h[BW];       // _h: the half step (upper four slots out of range)
#pragma unroll
  for (int i = 0; i < AW; ++i) {
    const int row = (wv + NW * i) * 8 + rr, col = sl ^ ((row >> 1) & 7);
    avoff[i] = (unsigned)((row * Ktot + col * 8) * 2);
    avoff_h[i] = col < 4 ? avoff[i] : OOB;
  }
#pragma unroll
  for (int i = 0; i < BW; ++i) {
    const int row = (wv + NW * i) * 8 + rr, col = sl ^ ((row >> 1) & 7);
    const long p = (long)p0 + row;
    if (p < Npix) {
      int n, rem;
      pix_split(p, HoWo, n, rem);
      bvoff[i] = (unsigned)(((long)n * P.x_img_stride + (long)rem * P.ldx + col * 8) * 2);
    } else {
      bvoff[i] = OOB;
    }
    bvoff_h[i] = col < 4 ? bvoff[i] : OOB;
  }
  auto issue = [&](int step, int stage_off, bool half) __attribute__((always_inline)) {
    const int kb = step * 128;                                   // bytes along K
    if constexpr (!(FAR3D_ABLATE & 32)) {                        // ablation 32: no weight stream
#pragma unroll
      for (int i = 0; i < AW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + stage_off + (wv + NW * i) * 1024), 16, half ? avoff_h[i] : avoff[i], kb, 0, 0);
    }
    if constexpr (!(FAR3D_ABLATE & 64)) {                        // ablation 64: no activation stream
#pragma unroll
      for (int i = 0; i < BW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(smem + stage_off + BM * 128 + (wv + NW * i) * 1024), 16, half ? bvoff_h[i] : bvoff[i], kb, 0, 0);
    }
  };

  f32x16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aaddr[4][WM], baddr[4][WN];                                // [k16 sub-step of the 64-channel row]
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    aaddr[0][i] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) aaddr[kk][i] = aaddr[0][i] ^ (kk << 5);
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int r = (wn * WN + j) * 32 + l31;
    baddr[0][j] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk) baddr[kk][j] = baddr[0][j] ^ (kk << 5);
  }
  const int nsub = P.cin_pad / 32, nfull = nsub >> 1, tail = nsub & 1, nsteps = nfull + tail;
  constexpr int D = NS - 1, UNIT = AW + BW;
  static_assert(NS >= 2 && NS <= 4, "ring depth");
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nsteps) issue(s, s * STAGE, s >= nfull);
  FAR3D_CONV_TS(P, 2);
  FAR3D_CONV_TS_FIRST(P);
#define FAR3D_WIDE_SYNC(S)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if constexpr (FAR3D_ABLATE & 16) {                                                                       \
    } else if constexpr (D == 1) {                                                                           \
      wait_vmcnt<0>();                                                                                       \
    } else {                                                                                                 \
      int units_ = 0;                                                                                        \
      _Pragma("unroll")                                                                                      \
      for (int q_ = 1; q_ < D; ++q_)                                                                         \
        if ((S) + q_ < nsteps) units_ += 1;                                                                  \
      wait_vmcnt_units<UNIT, D - 1>(units_);                                                                 \
    }                                                                                                        \
    if constexpr (!(FAR3D_ABLATE & 16)) __builtin_amdgcn_s_barrier();                                        \
    asm volatile("" ::: "memory");                                                                           \
    __builtin_amdgcn_sched_barrier(0);
#define FAR3D_WIDE_STEP(STG, S)                                                                              \
  {                                                                                                          \
    FAR3D_WIDE_SYNC(S)                                                                                       \
    if constexpr (!(FAR3D_ABLATE & 2)) {                                                                     \
    if ((S) + D < nsteps) issue((S) + D, (((STG) + D) % NS) * STAGE, (S) + D >= nfull);                      \
    }                                                                                                        \
    if constexpr (!(FAR3D_ABLATE & 4)) gemm_step_wide<WM, WN, (STG) * STAGE, BM, 4>(smem, aaddr, baddr, acc);   \
  }
  for (int s0 = 0; s0 < nfull; s0 += NS) {
    FAR3D_WIDE_STEP(0, s0)
    if (s0 + 1 < nfull) FAR3D_WIDE_STEP(1, s0 + 1)
    if constexpr (NS > 2) { if (s0 + 2 < nfull) FAR3D_WIDE_STEP(2, s0 + 2) }
    if constexpr (NS > 3) { if (s0 + 3 < nfull) FAR3D_WIDE_STEP(3, s0 + 3) }
  }
  if (tail) {
    FAR3D_WIDE_SYNC(nfull)
    const int ts = nfull % NS;
    if (ts == 0)                            gemm_step_wide<WM, WN, 0, BM, 2>(smem, aaddr, baddr, acc);
    else if (ts == 1)                       gemm_step_wide<WM, WN, STAGE, BM, 2>(smem, aaddr, baddr, acc);
    else if constexpr (NS > 2) {
      if (ts == 2)                          gemm_step_wide<WM, WN, 2 * STAGE, BM, 2>(smem, aaddr, baddr, acc);
      else if constexpr (NS > 3)            gemm_step_wide<WM, WN, 3 * STAGE, BM, 2>(smem, aaddr, baddr, acc);
    }
  }
#undef FAR3D_WIDE_STEP
#undef FAR3D_WIDE_SYNC
  FAR3D_CONV_TS(P, 4);
  if constexpr (FAR3D_ABLATE & 1) { ablate_epilogue<WM, WN>(P, acc); return; }
  if (P.y_rows16) {
    constexpr int RING = NS * (BM + BP) * 128, OUTB = BP * (BM * 2 + 16);
    int ep_n0, ep_r0;
    pix_split((long)p0, HoWo, ep_n0, ep_r0);
    epilogue_rows16<NW, WM, WN, BM, BP, false>(P, smem, acc, m0, wm, wn, l31, hi, [&](int pl) -> long {
      if ((long)p0 + pl >= Npix) return -1L;
      int n = ep_n0, rem = ep_r0 + pl;                 // (image, pixel) of the tile's first pixel + pl: no division per piece
      while (rem >= HoWo) { rem -= HoWo; ++n; }
      return (long)n * P.y_img_stride + (long)rem * P.ldy;
    }, (long)p0, RING > OUTB ? RING : OUTB);
    FAR3D_CONV_TS_END(P);
    return;
  }
  igemm_epilogue<WM, WN>(P, acc, p0, m0, wm, wn, l31, hi, HoWo, Npix);
#endif
}

template <int WGM, int WGN, int WM, int WN, int NS = 2>
static int launch_gemm1x1_wide(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  const long Npix = (long)P.N * P.Ho * P.Wo;
  const int npt = (int)((Npix + BP - 1) / BP), nct = (P.Cout + BM - 1) / BM;
  constexpr size_t lds_ring = (size_t)NS * (BM + BP) * 128, lds_out = (size_t)BP * (BM * 2 + 16);
  static_assert(lds_ring <= 163840, "LDS budget");
  constexpr size_t lds0 = lds_ring > lds_out ? lds_ring : lds_out, lds_sums = lds0 + 2 * BM * sizeof(long long);
  const size_t lds = P.chan_sums ? lds_sums : lds0;
  if (P.chan_sums && !gemm_sums_ok<BM, BP, 64 * WGM * WGN>(P, lds_sums)) return FAR3D_ERR_ARG;
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds_sums > 65536)
    if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&gemm1x1_wide_kernel<WGM, WGN, WM, WN, NS>),
                                       (int)(lds_sums <= 163840 ? lds_sums : lds0), lds_ok, "far3d_conv2d_nhwc")) return rc;
  const unsigned blocks = (unsigned)((npt + 7) / 8 * 8) * (unsigned)nct;
  hipLaunchKernelGGL((gemm1x1_wide_kernel<WGM, WGN, WM, WN, NS>), dim3(blocks), dim3(64 * WGM * WGN), lds, st, P, npt, nct);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the full-line GEMM with SPLIT RINGS and WAVE-SPECIALISED LDS-DMA issue.
// The concat GEMMs run at the speed of their DMA stream, and the stream is bound by latency x bytes in flight (profiles/r3-r4: a
// 256 x 256 tile spends 3.4 us per 64-channel step against 0.86 us of MFMA work; the weights are L2 hits, ~0.5 us, the activation
// slab comes from HBM / Infinity Cache, 2+ us).  One ring for both operands gives both the same depth, and the LDS holds only two
// 64 KiB stages of a 256 x 256 tile.  Here the weight ring is NSA stages deep and the activation ring NSB (2 + 3 stages of 32 KiB =
// all 160 KiB for a 256 x 256 tile): two activation steps in flight instead of one.  A wave's `vmcnt` retires in order, so one wave
// cannot keep an old activation piece in flight behind a younger weight piece -- therefore half of the waves issue ONLY weight pieces
// and the other half ONLY activation pieces, each waiting on its own counter for its own ring depth before the step's barrier.
// Everything else (128-byte LDS rows, source-side slot permutation, fragment double buffering, epilogue, channel sums, XCD-aware
// numbering) is gemm1x1_wide_kernel's.  The K loop is unrolled over lcm(NSA, NSB) steps so that every LDS offset is an immediate.
// ------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int NIT, typename Dma>
__device__ __forceinline__ void gemm_step_wide_rt(const unsigned char* smem, const int (&aaddr)[WM], const int (&baddr)[WN], int aoff, int boff,
                                                  f32x16_t (&acc)[WM][WN], Dma dma) {
  // Stage offsets at RUN time (wave-uniform, multiples of 1 KiB): WM + WN adds per step, and the K loop stays ONE rolled loop -- no
  // per-stage copies of the step, no conditional blocks around the MFMAs (every such block is a merge point for all accumulators and
  // cost the unrolled form of this kernel hundreds of spills at 2 x 4 tiles per wave).  k16 sub-step kk of the 64-channel row flips
  // bits 5-6 of the slot-permuted byte address: one v_xor per fragment read instead of four address registers per tile row.
  int aa[WM], bb[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) aa[i] = aaddr[i] + aoff;
#pragma unroll
  for (int j = 0; j < WN; ++j) bb[j] = baddr[j] + boff;
  u32x4_t af[2][WM], bf[2][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) af[0][i] = *reinterpret_cast<const u32x4_t*>(smem + aa[i]);
#pragma unroll
  for (int j = 0; j < WN; ++j) bf[0][j] = *reinterpret_cast<const u32x4_t*>(smem + bb[j]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int cur = it & 1, nxt = cur ^ 1;
    if (it + 1 < NIT) {
#pragma unroll
      for (int i = 0; i < WM; ++i) af[nxt][i] = *reinterpret_cast<const u32x4_t*>(smem + (aa[i] ^ ((it + 1) << 5)));
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[nxt][j] = *reinterpret_cast<const u32x4_t*>(smem + (bb[j] ^ ((it + 1) << 5)));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], af[cur][i], bf[cur][j]);
    __builtin_amdgcn_sched_barrier(0);
    dma(it);           // this sub-step's share of the LDS-DMA pieces of a later step: their issue cost hides under the MFMAs above
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int V> struct ic_t { static constexpr int value = V; };
constexpr int far3d_gcd(int a, int b) { return b == 0 ? a : far3d_gcd(b, a % b); }
// f(ic_t<I>{}) for I = 0 .. N-1, unconditionally (compile-time loop: I is a constant inside f)
template <int I, int N, typename F>
__device__ __forceinline__ void far3d_static_for(F&& f) {
  if constexpr (I < N) {
    f(ic_t<I>{});
    far3d_static_for<I + 1, N>(f);
  }
}

template <int WGM, int WGN, int WM, int WN, int NSA, int NSB>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm1x1_split_kernel(IgemmParams P, int npt, int nct) {
#if defined(__HIP_DEVICE_COMPILE__)
  FAR3D_CONV_TS_ENTRY(P);
  constexpr int NW = WGM * WGN, NWA = NW / 2, NWB = NW - NWA;
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr int GA = BM / 8, GB = BP / 8;                       // 1 KB DMA pieces (8 rows x 128 B) per 64-channel step
  static_assert(NW >= 2 && GA % NWA == 0 && GB % NWB == 0, "every wave of a role issues the same number of pieces (counted vmcnt)");
  static_assert(NWA % 2 == 0 && NWB % 2 == 0, "a lane's slot permutation must not depend on the piece index (rows 8 * nr apart, nr even)");
  constexpr int APW = GA / NWA, BPW = GB / NWB, PWMAX = APW > BPW ? APW : BPW;
  constexpr int ASTG = BM * 128, BSTG = BP * 128, BBASE = NSA * ASTG;
  constexpr int DA = NSA - 1, DB = NSB - 1;
  static_assert(NSA >= 2 && NSB >= 2 && NSA <= 4 && NSB <= 5, "ring depths");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;          // XCD-aware numbering as in gemm1x1_pipe_kernel
  const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
  if (pt >= npt) return;
  const int p0 = pt * BP, m0 = ct * BM;
  const int HoWo = P.Ho * P.Wo;
  const long Npix = (long)P.N * HoWo;
  const int Ktot = P.cin_pad;
  const int rr = lane >> 3, sl = lane & 7;                       // row of the piece, 16-byte slot of the 128-byte LDS row
  constexpr unsigned OOB = 0x80000000u;
  const bool is_a = wv < NWA;                                    // wave-uniform role: weight pieces / activation pieces
  const int dw = is_a ? wv : wv - NWA, nr = is_a ? NWA : NWB, npw = is_a ? APW : BPW;
  const long x_bytes = ((long)(P.N - 1) * P.x_img_stride + (long)HoWo * P.ldx) * 2;
  const long w_bytes = w_tile_bytes(BM, m0, P.Cout, Ktot);
  const void* base = is_a ? (const void*)(reinterpret_cast<const bf16_t*>(P.w) + (long)m0 * Ktot) : P.x;
  const long nbytes = is_a ? w_bytes : x_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(nbytes < 0x7fffffffL ? nbytes : 0x7fffffffL), 0x00020000);
  unsigned voff[PWMAX];
  const int col = sl ^ ((dw * 4 + (rr >> 1)) & 7);               // = sl ^ ((row >> 1) & 7) for every piece of this lane (nr even)
  const bool lowcol = col < 4;                                   // the half step fetches slots 0..3 only (upper half out of range -> zeros)
#pragma unroll
  for (int i = 0; i < PWMAX; ++i) {
    const int row = (dw + nr * i) * 8 + rr;
    if (is_a) {
      voff[i] = (unsigned)((row * Ktot + col * 8) * 2);
    } else {
      const long p = (long)p0 + row;
      if (p < Npix) {
        int n, rem;
      pix_split(p, HoWo, n, rem);
        voff[i] = (unsigned)(((long)n * P.x_img_stride + (long)rem * P.ldx + col * 8) * 2);
      } else {
        voff[i] = OOB;
      }
    }
  }
  // pieces [i0, i1) of this wave's share of `step`; so: LDS byte offset of the stage of this wave's ring
  auto issue_part = [&](int step, bool half, int so, int i0, int i1) __attribute__((always_inline)) {
    const int kb = step * 128;                                   // bytes along K
#pragma unroll
    for (int i = 0; i < PWMAX; ++i)
      if (i >= i0 && i < i1 && i < npw)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + so + (dw + nr * i) * 1024), 16, (half && !lowcol) ? OOB : voff[i], kb, 0, 0);
  };
  auto issue = [&](int step, bool half, int so) __attribute__((always_inline)) { issue_part(step, half, so, 0, PWMAX); };

  f32x16_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int aaddr[WM], baddr[WN];                                      // k16 sub-step 0 of the 64-channel row (sub-step kk: ^ kk << 5)
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    aaddr[i] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int r = (wn * WN + j) * 32 + l31;
    baddr[j] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
  }
  const int nsub = P.cin_pad / 32, nfull = nsub >> 1, tail = nsub & 1, nsteps = nfull + tail;
  const int dmine = is_a ? DA : DB;
#pragma unroll
  for (int s = 0; s < (DA > DB ? DA : DB); ++s)
    if (s < dmine && s < nsteps) issue(s, s >= nfull, is_a ? s * ASTG : BBASE + s * BSTG);
  FAR3D_CONV_TS(P, 2);
  // ONE rolled loop over the steps; the half step (K % 64 == 32) runs the full four sub-steps on a row whose upper half the DMA filled
  // with zeros (out-of-range slots), so it needs no code of its own
  const int ring_lo = is_a ? 0 : BBASE, ring_hi = is_a ? NSA * ASTG : BBASE + NSB * BSTG, ring_stg = is_a ? ASTG : BSTG;
  int iss_off = is_a ? (DA % NSA) * ASTG : BBASE + (DB % NSB) * BSTG;        // stage of THIS wave's ring the next issue goes into
  int sa_off = 0, sb_off = BBASE;                                            // stages of the step being computed
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    // wait until this wave's pieces of step s have landed: the steps s+1 .. s+D-1 it issued later may stay in flight
    __builtin_amdgcn_sched_barrier(0);
    {
      int ahead = nsteps - 1 - s;
      if (ahead > dmine - 1) ahead = dmine - 1;
      if (is_a) wait_vmcnt_units<APW, DA - 1>(ahead);
      else wait_vmcnt_units<BPW, DB - 1>(ahead);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the barrier also says that every wave is done with step s-1, whose stages the steps s+DA / s+DB go into
    // ... issued in four shares between the MFMA groups of this step (PWMAX / 4 pieces after each k16 sub-step)
    const bool more = s + dmine < nsteps;
    const int nstep = s + dmine, ioff = iss_off;
    constexpr int PPI = (PWMAX + 3) / 4;
    gemm_step_wide_rt<WM, WN, 4>(smem, aaddr, baddr, sa_off, sb_off, acc, [&](int it) __attribute__((always_inline)) {
      if (more) issue_part(nstep, nstep >= nfull, ioff, it * PPI, (it + 1) * PPI);
    });
    iss_off += ring_stg;
    if (iss_off == ring_hi) iss_off = ring_lo;
    sa_off += ASTG;
    if (sa_off == NSA * ASTG) sa_off = 0;
    sb_off += BSTG;
    if (sb_off == BBASE + NSB * BSTG) sb_off = BBASE;
  }
  if (P.y_rows16) {
    constexpr int RING = NSA * ASTG + NSB * BSTG, OUTB = BP * (BM * 2 + 16);
    FAR3D_CONV_TS(P, 4);
    __syncthreads();         // the ring may fill the whole LDS: the epilogue's sum scratch (zeroed before its first barrier) lies inside it
    int ep_n0, ep_r0;
    pix_split((long)p0, HoWo, ep_n0, ep_r0);
    epilogue_rows16<NW, WM, WN, BM, BP, false>(P, smem, acc, m0, wm, wn, l31, hi, [&](int pl) -> long {
      if ((long)p0 + pl >= Npix) return -1L;
      int n = ep_n0, rem = ep_r0 + pl;                 // (image, pixel) of the tile's first pixel + pl: no division per piece
      while (rem >= HoWo) { rem -= HoWo; ++n; }
      return (long)n * P.y_img_stride + (long)rem * P.ldy;
    }, (long)p0, OUTB);      // the sum scratch sits right behind the staged tile (the ring is dead by then)
    (void)RING;
    FAR3D_CONV_TS_END(P);
    return;
  }
  igemm_epilogue<WM, WN>(P, acc, p0, m0, wm, wn, l31, hi, HoWo, Npix);
#endif
}

template <int WGM, int WGN, int WM, int WN, int NSA, int NSB>
static int launch_gemm1x1_split(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  const long Npix = (long)P.N * P.Ho * P.Wo;
  const int npt = (int)((Npix + BP - 1) / BP), nct = (P.Cout + BM - 1) / BM;
  constexpr size_t lds_ring = (size_t)NSA * BM * 128 + (size_t)NSB * BP * 128, lds_out = (size_t)BP * (BM * 2 + 16);
  static_assert(lds_ring <= 163840, "LDS budget");
  constexpr size_t lds_out_sums = lds_out + 2 * BM * sizeof(long long);
  static_assert(lds_out_sums <= 163840, "LDS budget (epilogue)");
  constexpr size_t lds = lds_ring > lds_out_sums ? lds_ring : lds_out_sums;
  if (P.chan_sums && !gemm_sums_ok<BM, BP, 64 * WGM * WGN>(P, lds)) return FAR3D_ERR_ARG;
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds > 65536)
    if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&gemm1x1_split_kernel<WGM, WGN, WM, WN, NSA, NSB>), (int)lds, lds_ok, "far3d_conv2d_nhwc")) return rc;
  const unsigned blocks = (unsigned)((npt + 7) / 8 * 8) * (unsigned)nct;
  hipLaunchKernelGGL((gemm1x1_split_kernel<WGM, WGN, WM, WN, NSA, NSB>), dim3(blocks), dim3(64 * WGM * WGN), lds, st, P, npt, nct);
  return 0;
}
