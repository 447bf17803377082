// VoVNet stem, first convolution (3 -> 64 channels, 3x3, stride 2, pad 1, folded BN + ReLU; ref models/backbones/vovnet.py:306-311) read
// straight from the NCHW fp32 image: the im2col map of far3d_stem_im2col (69 MB written and read back per 7-camera frame) never exists.
//
// A wave owns 32 consecutive output pixels of one output row x all 64 channels: K = 27 (+5 zero) = two v_mfma_f32_32x32x16_bf16 per 32-row
// channel tile.  The MFMA's B operand wants, per lane (pixel l31, half hi), the 8 im2col values k = hi*8 .. hi*8+7 of each K half -- 16
// image samples per lane, gathered with plain 4-byte loads (neighbouring lanes = neighbouring output pixels = every second image column;
// the taps of one lane fill the gaps, the rows stay in L1 / L2) whose image offsets are per-lane constants computed once; the other
// waves of the SIMD (5-6 per SIMD at ~90 registers) cover a tile's load latency.  Same products in the same order as the
// im2col + GEMM path (k 0-15, then 16-31, fp32 accumulation from zero, bias, ReLU, round to bf16): bit-identical output.
// Epilogue: the wave's 32 x 64 tile is transposed through 4.5 KiB of LDS and leaves as 16-byte pieces covering whole 128-byte pixel rows.
#include "igemm_kernels.hpp"

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void stem_conv_kernel(const float* __restrict__ img, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                        bf16_t* __restrict__ y, int N, int H, int W, int Ho, int Wo, int ldy, long y_img_stride,
                                                        int tiles_x, int ntiles, int relu) {
  constexpr int RS = 144;                                   // LDS row: 64 channels x 2 B + 16 B of padding
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32 * RS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
  unsigned char* my = smem + wv * 32 * RS;
  // weights: A operand, row tile i, K half h: 8 bf16 of row i*32 + l31 at k = h*16 + hi*8
  u32x4_t af[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) af[i][h] = *reinterpret_cast<const u32x4_t*>(w + (long)(i * 32 + l31) * 32 + h * 16 + hi * 8);
  // this lane's 16 im2col slots: k = h*16 + hi*8 + j -> (tap, channel) -> image offset relative to (n, 2*oy - 1, 2*ox - 1), channel 0;
  // one register per slot: offset (24 bits) | dy << 24 | dx << 26 | padding slot << 28
  int kinfo[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = (s >> 3) * 16 + hi * 8 + (s & 7);
    const int tap = (k * 11) >> 5, c = k - 3 * tap;          // k / 3, k % 3 (k < 32)
    const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
    kinfo[s] = ((c * H + dy) * W + dx) | (dy << 24) | (dx << 26) | ((k >= 27 ? 1 : 0) << 28);
  }
  auto gather = [&](int t, float (&v)[16]) __attribute__((always_inline)) {
    const int tx = t % tiles_x, r = t / tiles_x, oy = r % Ho, n = r / Ho;
    const int ox = tx * 32 + l31;
    const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
    const float* base = img + ((long)n * 3 * H + iy0) * W + ix0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int iy = iy0 + ((kinfo[s] >> 24) & 3), ix = ix0 + ((kinfo[s] >> 26) & 3);
      const bool ok = ox < Wo && !(kinfo[s] >> 28) && iy >= 0 && iy < H && ix >= 0 && ix < W;
      v[s] = ok ? base[kinfo[s] & 0xffffff] : 0.f;
    }
  };
  const int wstride = gridDim.x * 4;
  int t = blockIdx.x * 4 + wv;
  for (; t < ntiles; t += wstride) {
    float cur[16];
    gather(t, cur);
    u32x4_t bf0, bf1;
    bf0.x = pack_bf16x2(cur[0], cur[1]); bf0.y = pack_bf16x2(cur[2], cur[3]); bf0.z = pack_bf16x2(cur[4], cur[5]); bf0.w = pack_bf16x2(cur[6], cur[7]);
    bf1.x = pack_bf16x2(cur[8], cur[9]); bf1.y = pack_bf16x2(cur[10], cur[11]); bf1.z = pack_bf16x2(cur[12], cur[13]); bf1.w = pack_bf16x2(cur[14], cur[15]);
    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      mma<bf16_t>(acc[i], af[i][0], bf0);
      mma<bf16_t>(acc[i], af[i][1], bf1);
    }
    // wave-private transposition: lane (pixel l31, half hi) holds channels i*32 + 8q + 4hi .. +3 of its pixel
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + i * 32 + 8 * q + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);      // L1 hits
        float v0 = acc[i][4 * q] + b4.x, v1 = acc[i][4 * q + 1] + b4.y, v2 = acc[i][4 * q + 2] + b4.z, v3 = acc[i][4 * q + 3] + b4.w;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        *reinterpret_cast<uint2*>(my + l31 * RS + (i * 32 + 8 * q + 4 * hi) * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int tx = t % tiles_x, r = t / tiles_x, oy = r % Ho, n = r / Ho;
    bf16_t* yrow = y + (long)n * y_img_stride + ((long)oy * Wo + tx * 32) * ldy;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int px = it * 8 + (lane >> 3), ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(my + px * RS + ch * 16);
      if (tx * 32 + px < Wo) *reinterpret_cast<uint4*>(yrow + (long)px * ldy + ch * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tile's LDS reads are done before the next tile overwrites it
    __builtin_amdgcn_wave_barrier();
  }
}

// See include/far3d_hip.h for the argument contract.
extern "C" int far3d_stem_conv(const float* img, const void* w, const float* bias, void* y, int N, int H, int W, int ldy, long y_img_stride,
                               int act, void* stream) {
  FAR3D_CHECK_ARG(img && w && y, "far3d_stem_conv: null pointer argument");
  FAR3D_CHECK_ARG(N > 0 && H > 1 && W > 1 && (long)3 * H * W < (1L << 24), "far3d_stem_conv: bad image size N=%d H=%d W=%d (3*H*W < 2^24)", N, H, W);
  FAR3D_CHECK_ARG(act == ACT_NONE || act == ACT_RELU, "far3d_stem_conv: activation none or ReLU");
  FAR3D_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)w % 16) == 0 && ldy >= 64 && ldy % 8 == 0 && y_img_stride % 8 == 0,
                  "far3d_stem_conv: y / w must be 16-byte aligned, pixel stride a multiple of 8 elements (>= 64)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tiles_x = (Wo + 31) / 32, ntiles = N * Ho * tiles_x;
  int blocks = (ntiles + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, (const bf16_t*)w, bias, (bf16_t*)y, N, H, W, Ho, Wo,
                     ldy, y_img_stride, tiles_x, ntiles, act == ACT_RELU ? 1 : 0);
  FAR3D_CHECK_LAUNCH("far3d_stem_conv");
  return FAR3D_OK;
}
