// Row-wise normalisations and reductions of the path (HBM-bound; one wave per row, 16-B accesses).
//   far3d_layernorm        -- nn.LayerNorm(C) (+ optional ReLU) (+ optional second output y2 = y + add, used to
//                             hand the next GEMM "query + query_pos" without another pass).  Decoder norms
//                             (ref models/utils/detr3d_transformer.py:304-307,398-400), cls branch LN
//                             (ref models/dense_heads/farhead.py:230-239), cam_embed LN (:506-512), time_embedding LN.
//   far3d_groupnorm_nhwc   -- GroupNorm(32, C) + ReLU on NHWC maps (ref models/depth_predictor/depth_predictor.py:43-45).
//   far3d_ese_nhwc         -- VoVNet eSE: x * hsigmoid(fc(avgpool(x))) (+ identity) (ref models/backbones/vovnet.py:173-185,232-236).
//   far3d_maxpool3x3s2_nhwc-- MaxPool2d(3, 2, ceil_mode=True) (ref models/backbones/vovnet.py:249-250).
#include "common.hpp"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------- LayerNorm
template <int MAXV>  // float4 per lane; C <= 256 * MAXV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, int rows,
                                                        int C, int ldx, int ldy, float eps, int act,
                                                        const float* __restrict__ add, int lda, void* __restrict__ y2,
                                                        int ldy2, int y2_dt, void* __restrict__ yb, int ldyb, int yb_dt,
                                                        const int* __restrict__ out_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int orow = out_rows ? out_rows[row] : row;      // row of y2 / yb (far3d_layernorm_rows); requested with the row itself
  const float* xr = x + (long)row * ldx;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, d = v[i].z - mean, e = v[i].w - mean;
      q += (a * a + b * b) + (d * d + e * e);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      float4 g = make_float4(1, 1, 1, 1), b = make_float4(0, 0, 0, 0);
      if (gamma) g = *reinterpret_cast<const float4*>(gamma + c);
      if (beta) b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *reinterpret_cast<float4*>(y + (long)row * ldy + c) = o;
      if (yb) {
        if (yb_dt == FAR3D_DT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(yb) + (long)orow * ldyb + c) = o;
        else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(yb) + (long)orow * ldyb + c) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
      }
      if (y2) {
        const float4 a = *reinterpret_cast<const float4*>(add + (long)row * lda + c);
        const float4 q2 = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
        if (y2_dt == FAR3D_DT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(y2) + (long)orow * ldy2 + c) = q2;
        else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y2) + (long)orow * ldy2 + c) = make_uint2(pack_bf16x2(q2.x, q2.y), pack_bf16x2(q2.z, q2.w));
      }
    }
  }
}

static int layernorm_launch(const float* x, const float* gamma, const float* beta, float* y, int rows, int C,
                            int ldx, int ldy, float eps, int act, const float* add, int lda, void* y2, int ldy2,
                            int y2_dt, void* yb, int ldyb, int yb_dt, const int32_t* out_rows, void* stream) {
  FAR3D_CHECK_ARG(x && y, "far3d_layernorm: null x/y");
  FAR3D_CHECK_ARG(rows >= 0 && C > 0 && (C % 4) == 0 && C <= 1024, "far3d_layernorm: C=%d must be a multiple of 4, <= 1024", C);
  FAR3D_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && (!y2 || (add && lda % 4 == 0 && ldy2 % 4 == 0)) && (!yb || ldyb % 4 == 0),
                  "far3d_layernorm: row strides must be multiples of 4 elements; y2 needs add");
  if (rows == 0) return FAR3D_OK;
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 256)
    hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, x, gamma, beta, y, rows, C, ldx, ldy, eps, act, add, lda, y2, ldy2, y2_dt, yb, ldyb, yb_dt, (const int*)out_rows);
  else
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, x, gamma, beta, y, rows, C, ldx, ldy, eps, act, add, lda, y2, ldy2, y2_dt, yb, ldyb, yb_dt, (const int*)out_rows);
  FAR3D_CHECK_LAUNCH("far3d_layernorm");
  return FAR3D_OK;
}

extern "C" int far3d_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C,
                               int ldx, int ldy, float eps, int act, const float* add, int lda, void* y2, int ldy2,
                               int y2_dt, void* yb, int ldyb, int yb_dt, void* stream) {
  return layernorm_launch(x, gamma, beta, y, rows, C, ldx, ldy, eps, act, add, lda, y2, ldy2, y2_dt, yb, ldyb, yb_dt, nullptr, stream);
}

extern "C" int far3d_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int rows, int C,
                                    int ldx, int ldy, float eps, int act, const float* add, int lda, void* y2, int ldy2,
                                    int y2_dt, void* yb, int ldyb, int yb_dt, const int32_t* out_rows, void* stream) {
  FAR3D_CHECK_ARG(out_rows && (y2 || yb), "far3d_layernorm_rows: needs out_rows and at least one of y2 / yb");
  return layernorm_launch(x, gamma, beta, y, rows, C, ldx, ldy, eps, act, add, lda, y2, ldy2, y2_dt, yb, ldyb, yb_dt, out_rows, stream);
}

// ---------------------------------------------------------------- per-(image, channel) sums over H*W
// Used by eSE (channel means) and GroupNorm (sum and sum of squares).  x NHWC (T), C channels.
// DETERMINISTIC: workgroup (blk, n) writes its partial sums to part[n][blk][C][2] with plain stores; the consumer
// (ese_gate_kernel / gn_stats_kernel) adds the `nblk` partials in index order.  No atomics, no memset -- run-to-run and
// hipGraph-vs-eager results are bit-identical.
template <typename T>
__global__ __launch_bounds__(256) void chan_sums_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                        int ldx, long img_stride, int rows_per_block) {
  // thread -> 4 consecutive channels; the 256 threads cover rstep = 256*4/C rows per pass; 8 passes are kept in flight.
  __shared__ float red[2][1024];
  const int n = blockIdx.y;
  const int cq = C / 4;
  const int tcol = threadIdx.x % cq, trow = threadIdx.x / cq;
  const int rstep = 256 / cq;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(HW, r0 + rows_per_block);
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  if (trow < rstep) {
    const T* base = x + (long)n * img_stride + chan_off<T>(tcol * 4);
    int r = r0 + trow;
    for (; r + 7 * rstep < r1; r += 8 * rstep) {          // 8 row loads in flight per thread: the loop is latency bound
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = load4(base + (long)(r + k * rstep) * ldx);
      s.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
      s.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
      s.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
      s.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
      q.x += ((v[0].x * v[0].x + v[1].x * v[1].x) + (v[2].x * v[2].x + v[3].x * v[3].x)) + ((v[4].x * v[4].x + v[5].x * v[5].x) + (v[6].x * v[6].x + v[7].x * v[7].x));
      q.y += ((v[0].y * v[0].y + v[1].y * v[1].y) + (v[2].y * v[2].y + v[3].y * v[3].y)) + ((v[4].y * v[4].y + v[5].y * v[5].y) + (v[6].y * v[6].y + v[7].y * v[7].y));
      q.z += ((v[0].z * v[0].z + v[1].z * v[1].z) + (v[2].z * v[2].z + v[3].z * v[3].z)) + ((v[4].z * v[4].z + v[5].z * v[5].z) + (v[6].z * v[6].z + v[7].z * v[7].z));
      q.w += ((v[0].w * v[0].w + v[1].w * v[1].w) + (v[2].w * v[2].w + v[3].w * v[3].w)) + ((v[4].w * v[4].w + v[5].w * v[5].w) + (v[6].w * v[6].w + v[7].w * v[7].w));
    }
    for (; r < r1; r += rstep) {
      const float4 a = load4(base + (long)r * ldx);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      q.x += a.x * a.x; q.y += a.y * a.y; q.z += a.z * a.z; q.w += a.w * a.w;
    }
  }
  // cross-row reduction through LDS: rstep*C <= 1024 floats
  if (trow < rstep) {
    *reinterpret_cast<float4*>(&red[0][trow * C + tcol * 4]) = s;
    *reinterpret_cast<float4*>(&red[1][trow * C + tcol * 4]) = q;
  }
  __syncthreads();
  const int nblk = gridDim.x;
  float* dst = part + ((long)n * nblk + blockIdx.x) * C * 2;
  for (int c = threadIdx.x; c < C; c += 256) {
    float ss = 0.f, qq = 0.f;
    for (int k = 0; k < rstep; ++k) { ss += red[0][k * C + c]; qq += red[1][k * C + c]; }
    *reinterpret_cast<float2*>(dst + 2 * c) = make_float2(ss, qq);
  }
}

// number of partial-sum workgroups per image: >= 64 Ki elements per workgroup, at most FAR3D_SUMS_MAX_PARTS
static int chan_sums_parts(int HW, int C) {
  long n = ((long)HW * C) >> 16;
  if (n < 1) n = 1;
  if (n > FAR3D_SUMS_MAX_PARTS) n = FAR3D_SUMS_MAX_PARTS;
  const int rstep = 256 / (C / 4);
  int rows_per_block = (int)((HW + n - 1) / n);
  rows_per_block = ((rows_per_block + 8 * rstep - 1) / (8 * rstep)) * (8 * rstep);
  return (HW + rows_per_block - 1) / rows_per_block;
}

static int launch_chan_sums(const void* x, int dt, float* part, int N, int HW, int C, int ldx, long img_stride, hipStream_t st) {
  const int nblk = chan_sums_parts(HW, C);
  const int rstep = 256 / (C / 4);
  int rows_per_block = (HW + nblk - 1) / nblk;
  rows_per_block = ((rows_per_block + 8 * rstep - 1) / (8 * rstep)) * (8 * rstep);
  dim3 grid(nblk, N), block(256);
  if (dt == FAR3D_DT_F32)
    hipLaunchKernelGGL(chan_sums_kernel<float>, grid, block, 0, st, (const float*)x, part, HW, C, ldx, img_stride, rows_per_block);
  else if (dt == FAR3D_DT_BF16_PAIR)
    hipLaunchKernelGGL(chan_sums_kernel<pair_t>, grid, block, 0, st, (const pair_t*)x, part, HW, C, ldx, img_stride, rows_per_block);
  else
    hipLaunchKernelGGL(chan_sums_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, part, HW, C, ldx, img_stride, rows_per_block);
  return nblk;
}

// ---------------------------------------------------------------- eSE
// Partial sums of image n, channel c: part[(n * nparts + b) * C + c] (float2: sum, sum of squares), b = 0..nparts-1; added in
// index order (deterministic).  A thread sums NCH channels (c0, c0 + cstep, ...) at once with 8 partials of each in flight:
// nparts <= 32 costs 4 dependent round trips for all of them (summing them one channel / one tail element at a time cost ~20 round
// trips and made the 16 gate launches of a frame 15 us each).  Slots past nparts / C load nothing and add +0.0.
template <int NCH>
__device__ __forceinline__ void sum_parts(const float2* __restrict__ src, int nparts, int C, int c0, int cstep, float2 (&out)[NCH]) {
  float ss[NCH], qq[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) { ss[j] = 0.f; qq[j] = 0.f; }
  for (int b = 0; b < nparts; b += 8) {
    float2 v[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        v[j][k] = src[(long)min(b + k, nparts - 1) * C + min(c0 + j * cstep, C - 1)];     // clamped, never predicated: a load under a
                                                                                          // condition waits for the one before it
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (!(c0 + j * cstep < C && b + k < nparts)) v[j][k] = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) { ss[j] += v[j][k].x; qq[j] += v[j][k].y; }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) out[j] = make_float2(ss[j], qq[j]);
}

// gate[n][c] = hsigmoid( sum_k fcw[c][k] * mean[n][k] + fcb[c] ).  A workgroup owns 16 output channels of one image: it first
// adds the per-workgroup partial channel sums of the pooling kernel (every workgroup of the image does; they are L2 hits), then
// each wave does 4 rows of the mat-vec.
// fixed != null: the channel sums arrive as 64-bit fixed-point integers [N][C] (far3d_conv2d_nhwc's chan_sums) instead of partials.
__global__ __launch_bounds__(256) void ese_gate_kernel(const float* __restrict__ part, const float* __restrict__ fcw,
                                                       const float* __restrict__ fcb, float* __restrict__ gate, int C,
                                                       float inv_hw, int nparts, const long long* __restrict__ fixed) {
  __shared__ float mean[1024];
  const int n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float2* src = reinterpret_cast<const float2*>(part) + (long)n * nparts * C;
  // each wave: 4 output channels at once (4 independent weight-row streams), float4 loads.  The weight rows do not depend on the
  // pooled means: their loads (C <= 1024: at most 4 per row and lane) go out first and overlap the partial-sum round trips.
  const int c0 = blockIdx.x * 16 + wv * 4;
  float4 w[4][4];
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = fcb[min(c0 + j, C - 1)];          // clamped (a predicated load waits for the one before it);
                                                                             // channels past C are never stored
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane * 4 + i * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[i][j] = (k < C && c0 + j < C) ? *reinterpret_cast<const float4*>(fcw + (long)(c0 + j) * C + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (fixed) {
    long long fx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fx[j] = fixed[(long)n * C + min((int)threadIdx.x + j * 256, C - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (threadIdx.x + j * 256 < C)
        mean[threadIdx.x + j * 256] = (float)((double)fx[j] * (1.0 / (1 << FAR3D_SUMS_FRAC_BITS)) * (double)inv_hw);
  } else {
    float2 sm[4];                                  // C <= 1024: channels t, t + 256, t + 512, t + 768
    sum_parts<4>(src, nparts, C, threadIdx.x, 256, sm);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (threadIdx.x + j * 256 < C) mean[threadIdx.x + j * 256] = sm[j].x * inv_hw;
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane * 4 + i * 256;
    if (k < C) {
      const float4 m = *reinterpret_cast<const float4*>(mean + k);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += (w[i][j].x * m.x + w[i][j].y * m.y) + (w[i][j].z * m.z + w[i][j].w * m.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = wave_sum(acc[j]);
    if (lane == 0 && c0 + j < C) {
      const float z = a + bias[j];
      gate[(long)n * C + c0 + j] = fminf(fmaxf(z + 3.f, 0.f), 6.f) / 6.f;   // F.relu6(x + 3) / 6
    }
  }
}

// x * g rounded, THEN + d (two fp32 operations as in the reference, vovnet.py:185,232-236): never contracted into an fma, so that
// far3d_ese_nhwc and far3d_ese_fused_nhwc -- two kernels evaluating the same expression -- give the same bits whatever the compiler
// would have chosen to fuse in each (HIP's __fmul_rn is a plain `*` and contracts like one)
__device__ __forceinline__ float ese_mul(float x, float g) {
#pragma clang fp contract(off)
  return x * g;
}
__device__ __forceinline__ float ese_mul_add(float x, float g, float d) {
#pragma clang fp contract(off)
  const float p = x * g;
  return p + d;
}

template <typename T>
__global__ __launch_bounds__(256) void ese_apply_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                        const T* __restrict__ idn, T* __restrict__ y, long total4, int C,
                                                        int HW, int ldx, long xs, int ldi, long is, int ldy, long ys,
                                                        long long* __restrict__ consumed, int nconsumed) {
  // the fixed-point channel sums the gate kernel (an earlier launch) read are returned to zero for their next producer
  if (consumed)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nconsumed; i += (long)gridDim.x * blockDim.x) consumed[i] = 0;
  const int cq = C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const long pr = i / cq;
    const int n = (int)(pr / HW);
    const long pix = pr - (long)n * HW;
    const int co = chan_off<T>(c);
    float4 v = load4(x + n * xs + pix * ldx + co);
    const float4 g = *reinterpret_cast<const float4*>(gate + (long)n * C + c);
    if (idn) {
      const float4 d = load4(idn + n * is + pix * ldi + co);
      v.x = ese_mul_add(v.x, g.x, d.x); v.y = ese_mul_add(v.y, g.y, d.y); v.z = ese_mul_add(v.z, g.z, d.z); v.w = ese_mul_add(v.w, g.w, d.w);
    } else {
      v.x = ese_mul(v.x, g.x); v.y = ese_mul(v.y, g.y); v.z = ese_mul(v.z, g.z); v.w = ese_mul(v.w, g.w);
    }
    store4(y + n * ys + pix * ldy + co, v);
  }
}

// scratch (floats): [N][FAR3D_SUMS_MAX_PARTS][C][2] partial sums, then [N][C] gates.  Size FAR3D_SUMS_SCRATCH_FLOATS(N, C); never
// needs zeroing, and one workspace (sized for the widest layer) serves a whole stream of calls.
extern "C" int far3d_ese_nhwc(const void* x, int dt, const float* fcw, const float* fcb, const void* identity, void* y,
                              float* scratch, int N, int HW, int C, int ldx, long x_img_stride, int ldi,
                              long i_img_stride, int ldy, long y_img_stride, long long* chan_sums, void* stream) {
  FAR3D_CHECK_ARG(x && fcw && fcb && y && scratch, "far3d_ese_nhwc: null pointer argument");
  FAR3D_CHECK_ARG(!chan_sums || dt != FAR3D_DT_F32, "far3d_ese_nhwc: fixed-point channel sums describe a bf16 / pair map");
  FAR3D_CHECK_ARG(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0 && (!identity || ldi % 4 == 0),
                  "far3d_ese_nhwc: bad sizes (C=%d must be a multiple of 4, <= 1024)", C);
  FAR3D_CHECK_ARG(dt == FAR3D_DT_F32 || dt == FAR3D_DT_BF16 || dt == FAR3D_DT_BF16_PAIR, "far3d_ese_nhwc: unsupported dtype");
  FAR3D_CHECK_ARG(dt != FAR3D_DT_BF16_PAIR || C % 32 == 0, "far3d_ese_nhwc: pair storage needs C %% 32 == 0");
  hipStream_t st = (hipStream_t)stream;
  float* sums = scratch;
  float* gate = scratch + (long)N * FAR3D_SUMS_MAX_PARTS * C * 2;
  const int nparts = chan_sums ? 0 : launch_chan_sums(x, dt, sums, N, HW, C, ldx, x_img_stride, st);
  hipLaunchKernelGGL(ese_gate_kernel, dim3((C + 15) / 16, N), dim3(256), 0, st, sums, fcw, fcb, gate, C, 1.f / HW, nparts, (const long long*)chan_sums);
  const long total4 = (long)N * HW * (C / 4);
  long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dt == FAR3D_DT_F32)
    hipLaunchKernelGGL(ese_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, gate,
                       (const float*)identity, (float*)y, total4, C, HW, ldx, x_img_stride, ldi, i_img_stride, ldy, y_img_stride, chan_sums, N * C);
  else if (dt == FAR3D_DT_BF16_PAIR)
    hipLaunchKernelGGL(ese_apply_kernel<pair_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const pair_t*)x, gate,
                       (const pair_t*)identity, (pair_t*)y, total4, C, HW, ldx, x_img_stride, ldi, i_img_stride, ldy, y_img_stride, chan_sums, N * C);
  else
    hipLaunchKernelGGL(ese_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, gate,
                       (const bf16_t*)identity, (bf16_t*)y, total4, C, HW, ldx, x_img_stride, ldi, i_img_stride, ldy, y_img_stride, chan_sums, N * C);
  FAR3D_CHECK_LAUNCH("far3d_ese_nhwc");
  return FAR3D_OK;
}

// ---------------------------------------------------------------- eSE apply + stage-end max-pool in one pass (round 5)
// far3d_ese_fused_nhwc = the gate launch of far3d_ese_nhwc(chan_sums) + ONE streaming launch that
//   * applies y = x * g (+ identity) in 16-byte pieces: a thread keeps ONE group of 8 channels (its gates live in registers, no division
//     in the loop, the next piece's loads in flight while the current one is stored);
//   * pooled != null: also writes maxpool3x3s2(ceil)(y), computed from x directly -- every tap is re-evaluated as x * g (+ identity) and
//     ROUNDED TO THE STORAGE FORMAT first, so the result is bit-identical to pooling the stored y; y == null: only the pooled map is wanted
//     (VoVNet stage 2 inside the detector: nothing else reads its output) and the full-resolution map is never written;
//   * returns the consumed channel sums to zero.
// Built first as ONE launch -- the gate workgroups at the head of the grid, the data workgroups spinning on a device-scope counter until
// the gates were published -- and measured 5x SLOWER than the two launches (profiles/r5/ese_handover.txt): on this 8-XCD part the
// hand-over needs either agent-scope fences (`buffer_wbl2 sc1` + `buffer_inv sc1`: a write-back and an invalidation of the XCD's whole L2,
// once per workgroup) or sc1 loads / stores that go past the L2 for every gate of every thread, and a thousand resident workgroups
// polling seven counters serialise on their cache lines.  A kernel boundary is the cheap device-wide release / acquire here.
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  struct Raw { uint4 a; };                             // the 16 bytes as loaded: conversion can wait until every load is issued
  static __device__ __forceinline__ Raw ldraw(const bf16_t* p) { Raw r; r.a = *reinterpret_cast<const uint4*>(p); return r; }
  static __device__ __forceinline__ void cvt(const Raw& q, float (&v)[8]) {
    const uint4 r = q.a;
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
  static __device__ __forceinline__ void round(float (&v)[8]) {      // what a store followed by a load returns
#pragma unroll
    for (int e = 0; e < 8; e += 2) { const uint32_t q = pack_bf16x2(v[e], v[e + 1]); v[e] = __uint_as_float(q << 16); v[e + 1] = __uint_as_float(q & 0xffff0000u); }
  }
};
template <> struct Vec8<pair_t> {        // p -> hi of 8 consecutive channels of one 32-block; lo 32 elements on
  struct Raw { uint4 h, l; };
  static __device__ __forceinline__ Raw ldraw(const pair_t* p) {
    Raw r;
    r.h = *reinterpret_cast<const uint4*>(p);
    r.l = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p) + 32);
    return r;
  }
  static __device__ __forceinline__ void cvt(const Raw& q, float (&v)[8]) {
    float h[8], l[8];
    Vec8<bf16_t>::Raw a, b;
    a.a = q.h; b.a = q.l;
    Vec8<bf16_t>::cvt(a, h);
    Vec8<bf16_t>::cvt(b, l);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = h[e] + l[e];
  }
  static __device__ __forceinline__ void ld(const pair_t* p, float (&v)[8]) {
    float h[8], l[8];
    Vec8<bf16_t>::ld(reinterpret_cast<const bf16_t*>(p), h);
    Vec8<bf16_t>::ld(reinterpret_cast<const bf16_t*>(p) + 32, l);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = h[e] + l[e];
  }
  static __device__ __forceinline__ void split(const float (&v)[8], uint4& h, uint4& l) {
    uint2 h0, l0, h1, l1;
    split4f(v[0], v[1], v[2], v[3], h0, l0);
    split4f(v[4], v[5], v[6], v[7], h1, l1);
    h = make_uint4(h0.x, h0.y, h1.x, h1.y); l = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
  static __device__ __forceinline__ void st(pair_t* p, const float (&v)[8]) {
    uint4 h, l;
    split(v, h, l);
    *reinterpret_cast<uint4*>(p) = h;
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p) + 32) = l;
  }
  static __device__ __forceinline__ void round(float (&v)[8]) {
    uint4 h, l;
    split(v, h, l);
    const uint32_t hh[4] = {h.x, h.y, h.z, h.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __uint_as_float(hh[e] << 16) + __uint_as_float(ll[e] << 16);
      v[2 * e + 1] = __uint_as_float(hh[e] & 0xffff0000u) + __uint_as_float(ll[e] & 0xffff0000u);
    }
  }
};

struct EseFusedParams {
  const void* x; const void* idn; void* y; void* pooled;
  const float* gate; long long* sums;
  int N, H, W, C, ldx, ldi, ldy, ldp, Hp, Wp, DB;
  long xs, is, ys, ps;
};

template <typename T, bool POOL>      // POOL = false: the launches without a pooled output (13 of the 16 of a frame) keep their small register file
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void ese_fused_kernel(EseFusedParams P) {
  const int n = blockIdx.y, tid = threadIdx.x, C = P.C;
  const int b = blockIdx.x, cq = C / 8;
  const int HW = P.H * P.W;
  const int stride = P.DB * 256;                           // a multiple of cq: a thread's channel group is the same for all its items
  const int i0 = b * 256 + tid;
  const int cg = i0 % cq, pstep = stride / cq;
  const int co = chan_off<T>(cg * 8);
  const T* x = reinterpret_cast<const T*>(P.x) + (long)n * P.xs + co;
  const T* idn = P.idn ? reinterpret_cast<const T*>(P.idn) + (long)n * P.is + co : nullptr;
  T* y = P.y ? reinterpret_cast<T*>(P.y) + (long)n * P.ys + co : nullptr;
  int pix = i0 / cq;
  float g[8];
  {
    const float4 g0 = *reinterpret_cast<const float4*>(P.gate + (long)n * C + cg * 8), g1 = *reinterpret_cast<const float4*>(P.gate + (long)n * C + cg * 8 + 4);
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
  }
  float xv[8], dv[8];
  if (y && pix < HW) {
    Vec8<T>::ld(x + (long)pix * P.ldx, xv);
    if (idn) Vec8<T>::ld(idn + (long)pix * P.ldi, dv);
  }
  // the gate launch has read the channel sums: return them to zero for their next producer
  for (int i = i0; i < C; i += stride) P.sums[(long)n * C + i] = 0;
  if (y) {
    for (; pix < HW; pix += pstep) {
      const int nxt = pix + pstep;
      float xn[8], dn[8];
      if (nxt < HW) {
        Vec8<T>::ld(x + (long)nxt * P.ldx, xn);
        if (idn) Vec8<T>::ld(idn + (long)nxt * P.ldi, dn);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = idn ? ese_mul_add(xv[e], g[e], dv[e]) : ese_mul(xv[e], g[e]);
      Vec8<T>::st(y + (long)pix * P.ldy, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) { xv[e] = xn[e]; dv[e] = dn[e]; }
    }
  }
  if constexpr (POOL) {
    T* pd = reinterpret_cast<T*>(P.pooled) + (long)n * P.ps + co;
    const int HWp = P.Hp * P.Wp;
    for (int q = i0 / cq; q < HWp; q += pstep) {
      const int oy = q / P.Wp, ox = q - oy * P.Wp;
      // taps past the edge are clamped onto the last row / column, which lies inside the window (ceil mode: a window starts inside the
      // input): a duplicate tap leaves the maximum as it is, and the loads are UNCONDITIONAL -- under `if (iy >= H) continue` every
      // load waited for the one before it (nine dependent round trips per pooled pixel).  A kernel row (3 taps of x and of the
      // identity) is in flight at a time, kept as loaded (16-byte registers) until all of them are issued.
      float m[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) {                 // not unrolled: the compiler otherwise hoists all 18 loads and spills
        const int iy = min(oy * 2 + ky, P.H - 1);
        typename Vec8<T>::Raw tv[3], td[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const long pp = (long)iy * P.W + min(ox * 2 + kx, P.W - 1);
          tv[kx] = Vec8<T>::ldraw(x + pp * P.ldx);
          if (idn) td[kx] = Vec8<T>::ldraw(idn + pp * P.ldi);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float v[8], d[8];
          Vec8<T>::cvt(tv[kx], v);
          if (idn) Vec8<T>::cvt(td[kx], d);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = idn ? ese_mul_add(v[e], g[e], d[e]) : ese_mul(v[e], g[e]);
          Vec8<T>::round(v);
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
        }
      }
      Vec8<T>::st(pd + (long)q * P.ldp, m);
    }
  }
}

extern "C" int far3d_ese_fused_nhwc(const void* x, int dt, const float* fcw, const float* fcb, const void* identity, void* y, void* pooled,
                                    float* gate, int N, int H, int W, int C, int ldx, long x_img_stride, int ldi,
                                    long i_img_stride, int ldy, long y_img_stride, int Hp, int Wp, int ldp, long p_img_stride,
                                    long long* chan_sums, void* stream) {
  FAR3D_CHECK_ARG(x && fcw && fcb && gate && chan_sums && (y || pooled), "far3d_ese_fused_nhwc: null pointer argument (y or pooled must be given)");
  FAR3D_CHECK_ARG(dt == FAR3D_DT_BF16 || dt == FAR3D_DT_BF16_PAIR, "far3d_ese_fused_nhwc: bf16 or pair-stored maps (the fixed-point channel sums describe those)");
  FAR3D_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C <= 1024 && C % (dt == FAR3D_DT_BF16_PAIR ? 32 : 8) == 0, "far3d_ese_fused_nhwc: bad sizes (C=%d)", C);
  auto al = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  FAR3D_CHECK_ARG(al(x) && ldx % 8 == 0 && x_img_stride % 8 == 0 && (!identity || (al(identity) && ldi % 8 == 0 && i_img_stride % 8 == 0)) &&
                  (!y || (al(y) && ldy % 8 == 0 && y_img_stride % 8 == 0)) && (!pooled || (al(pooled) && ldp % 8 == 0 && p_img_stride % 8 == 0)) && al(gate),
                  "far3d_ese_fused_nhwc: pointers and strides must be multiples of 16 bytes");
  if (pooled) {      // ceil_mode output size: ceil((H-3)/2)+1, last window must start inside the input
    int eh = (H - 3 + 1) / 2 + 1; if ((eh - 1) * 2 >= H) --eh;
    int ew = (W - 3 + 1) / 2 + 1; if ((ew - 1) * 2 >= W) --ew;
    FAR3D_CHECK_ARG(Hp == eh && Wp == ew, "far3d_ese_fused_nhwc: pooled size %dx%d != ceil-mode size %dx%d", Hp, Wp, eh, ew);
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ese_gate_kernel, dim3((C + 15) / 16, N), dim3(256), 0, st, (const float*)nullptr, fcw, fcb, gate, C, 1.f / ((float)H * W), 0,
                     (const long long*)chan_sums);
  EseFusedParams P;
  P.x = x; P.idn = identity; P.y = y; P.pooled = pooled; P.gate = gate; P.sums = chan_sums;
  P.N = N; P.H = H; P.W = W; P.C = C; P.ldx = ldx; P.ldi = ldi; P.ldy = ldy; P.ldp = ldp; P.Hp = Hp; P.Wp = Wp;
  P.xs = x_img_stride; P.is = i_img_stride; P.ys = y_img_stride; P.ps = p_img_stride;
  // workgroups per image: one 8-channel piece per thread and ~4 passes, the thread stride a multiple of C / 8 (unit: the smallest
  // workgroup count whose 256 * unit threads are a multiple of C / 8 -- 1 for C = 256 / 512 / 1024, 3 for C = 768)
  const int cq = C / 8;
  int unit = 1;
  while ((256 * unit) % cq) ++unit;
  const long items = (long)(y ? H * W : Hp * Wp) * cq;
  long db = (items + 4 * 256 - 1) / (4 * 256);
  if (db > 1200) db = 1200;
  db = (db + unit - 1) / unit * unit;
  P.DB = (int)db;
  dim3 grid((unsigned)P.DB, (unsigned)N);
  if (dt == FAR3D_DT_BF16_PAIR) {
    if (pooled) hipLaunchKernelGGL((ese_fused_kernel<pair_t, true>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((ese_fused_kernel<pair_t, false>), grid, dim3(256), 0, st, P);
  } else {
    if (pooled) hipLaunchKernelGGL((ese_fused_kernel<bf16_t, true>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((ese_fused_kernel<bf16_t, false>), grid, dim3(256), 0, st, P);
  }
  FAR3D_CHECK_LAUNCH("far3d_ese_fused_nhwc");
  return FAR3D_OK;
}

// ---------------------------------------------------------------- GroupNorm + ReLU (NHWC)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ part, float* __restrict__ stat, int C,
                                                       int groups, float inv_cnt, float eps, int nparts) {
  // one workgroup per image: per-channel sums first (all partial loads in flight), then thread g adds its group's channels in
  // ascending order -- the same additions in the same order as a per-group serial loop
  __shared__ float2 cs[1024];
  const int n = blockIdx.x, cpg = C / groups;
  const float2* src = reinterpret_cast<const float2*>(part) + (long)n * nparts * C;
  float2 sm[4];
  sum_parts<4>(src, nparts, C, threadIdx.x, 256, sm);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (threadIdx.x + j * 256 < C) cs[threadIdx.x + j * 256] = sm[j];
  __syncthreads();
  for (int g = threadIdx.x; g < groups; g += 256) {
    float s = 0.f, q = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += cs[c].x; q += cs[c].y; }
    const float mean = s * inv_cnt;
    const float var = fmaxf(q * inv_cnt - mean * mean, 0.f);
    stat[2 * ((long)n * groups + g)] = mean;
    stat[2 * ((long)n * groups + g) + 1] = 1.f / sqrtf(var + eps);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stat,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, long total4, int C, int groups, int HW, int relu) {
  const int cq = C / 4, cpg = C / groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const long pr = i / cq;
    const int n = (int)(pr / HW);
    const long eo = pr * (C * ChanScale<T>::v) + chan_off<T>(c);
    float4 v = load4(x + eo);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c + e) / cpg;
      const float mean = stat[((long)n * groups + g) * 2], rstd = stat[((long)n * groups + g) * 2 + 1];
      o[e] = (o[e] - mean) * rstd * gamma[c + e] + beta[c + e];
      if (relu) o[e] = fmaxf(o[e], 0.f);
    }
    store4(y + eo, make_float4(o[0], o[1], o[2], o[3]));
  }
}

extern "C" int far3d_groupnorm_nhwc(const void* x, int dt, const float* gamma, const float* beta, void* y, float* scratch,
                                    int N, int HW, int C, int groups, float eps, int relu, void* stream) {
  FAR3D_CHECK_ARG(x && gamma && beta && y && scratch, "far3d_groupnorm_nhwc: null pointer argument");
  FAR3D_CHECK_ARG(N > 0 && HW > 0 && C % 4 == 0 && C <= 1024 && groups > 0 && C % groups == 0, "far3d_groupnorm_nhwc: bad sizes C=%d groups=%d", C, groups);
  FAR3D_CHECK_ARG(dt == FAR3D_DT_F32 || dt == FAR3D_DT_BF16 || (dt == FAR3D_DT_BF16_PAIR && C % 32 == 0), "far3d_groupnorm_nhwc: unsupported dtype");
  const int cs = dt == FAR3D_DT_BF16_PAIR ? 2 : 1;      // stored elements per logical channel
  hipStream_t st = (hipStream_t)stream;
  float* sums = scratch;                                              // [N][nparts][C][2]
  float* stat = scratch + (long)N * FAR3D_SUMS_MAX_PARTS * C * 2;     // [N][groups][2]
  const int nparts = launch_chan_sums(x, dt, sums, N, HW, C, C * cs, (long)HW * C * cs, st);
  FAR3D_CHECK_ARG(groups * 2 <= C, "far3d_groupnorm_nhwc: groups*2 must be <= C");
  hipLaunchKernelGGL(gn_stats_kernel, dim3(N), dim3(256), 0, st, sums, stat, C, groups,
                     1.f / ((float)HW * (C / groups)), eps, nparts);
  const long total4 = (long)N * HW * (C / 4);
  long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dt == FAR3D_DT_F32)
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, stat, gamma, beta,
                       (float*)y, total4, C, groups, HW, relu);
  else if (dt == FAR3D_DT_BF16_PAIR)
    hipLaunchKernelGGL(gn_apply_kernel<pair_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const pair_t*)x, stat, gamma, beta,
                       (pair_t*)y, total4, C, groups, HW, relu);
  else
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, stat, gamma, beta,
                       (bf16_t*)y, total4, C, groups, HW, relu);
  FAR3D_CHECK_LAUNCH("far3d_groupnorm_nhwc");
  return FAR3D_OK;
}

// ---------------------------------------------------------------- MaxPool 3x3 stride 2, ceil_mode (NHWC)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, long total4, int C, int H,
                                                      int W, int Ho, int Wo, int ldy, long ys) {
  const int cq = C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    long pr = i / cq;
    const int ox = (int)(pr % Wo); pr /= Wo;
    const int oy = (int)(pr % Ho);
    const int n = (int)(pr / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky;
      if (iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx;
        if (ix >= W) continue;
        const float4 v = load4(x + (((long)n * H + iy) * W + ix) * (C * ChanScale<T>::v) + chan_off<T>(c));
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    store4(y + (long)n * ys + ((long)oy * Wo + ox) * ldy + chan_off<T>(c), m);
  }
}

extern "C" int far3d_maxpool3x3s2_nhwc(const void* x, int dt, void* y, int N, int H, int W, int C, int Ho, int Wo, int ldy,
                                       long y_img_stride, void* stream) {
  FAR3D_CHECK_ARG(x && y, "far3d_maxpool3x3s2_nhwc: null pointer argument");
  FAR3D_CHECK_ARG(dt == FAR3D_DT_F32 || dt == FAR3D_DT_BF16 || (dt == FAR3D_DT_BF16_PAIR && C % 32 == 0), "far3d_maxpool3x3s2_nhwc: unsupported dtype");
  FAR3D_CHECK_ARG(N > 0 && H > 0 && W > 0 && C % 4 == 0 && ldy >= C * (dt == FAR3D_DT_BF16_PAIR ? 2 : 1) && ldy % 4 == 0, "far3d_maxpool3x3s2_nhwc: bad sizes");
  // ceil_mode output size: ceil((H-3)/2)+1, last window must start inside the input
  int eh = (H - 3 + 1) / 2 + 1; if ((eh - 1) * 2 >= H) --eh;
  int ew = (W - 3 + 1) / 2 + 1; if ((ew - 1) * 2 >= W) --ew;
  FAR3D_CHECK_ARG(Ho == eh && Wo == ew, "far3d_maxpool3x3s2_nhwc: output %dx%d != ceil-mode size %dx%d", Ho, Wo, eh, ew);
  const long total4 = (long)N * Ho * Wo * (C / 4);
  long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FAR3D_DT_F32)
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)y, total4, C, H, W, Ho, Wo, ldy, y_img_stride);
  else if (dt == FAR3D_DT_BF16_PAIR)
    hipLaunchKernelGGL(maxpool_kernel<pair_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const pair_t*)x, (pair_t*)y, total4, C, H, W, Ho, Wo, ldy, y_img_stride);
  else
    hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, total4, C, H, W, Ho, Wo, ldy, y_img_stride);
  FAR3D_CHECK_LAUNCH("far3d_maxpool3x3s2_nhwc");
  return FAR3D_OK;
}
