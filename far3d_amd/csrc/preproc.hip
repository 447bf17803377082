// Device image pre-processing: the step immediately before the hot path.
// Replaces, per camera, AV2ResizeCropFlipRotImageV2._img_transform (ref datasets/pipelines/custom_pipeline.py:277-311:
// PIL Image.resize -> crop -> optional horizontal flip; rotation limits are (0,0) in the only config and asserted so at :69),
// NormalizeMultiviewImage (ref datasets/pipelines/transform_3d.py:89-101 -> mmcv.imnormalize) and AV2PadMultiViewImage
// (ref custom_pipeline.py:358-378), plus the HWC -> CHW transpose of the format bundle.
//
// Image.resize of an 8-bit image is Pillow's two-pass separable resampling (libImaging/Resample.c): horizontal pass to an
// 8-bit intermediate, then vertical pass, each output = clip8((2^21 + sum_k pixel_k * coeff_k) >> 22) with the filter
// coefficients pre-normalised and rounded to 22-bit fixed point on the host (far3d_amd/data_pipeline/resample.py builds
// them exactly like precompute_coeffs / normalize_coeffs_8bpc).  Doing the same integer arithmetic here makes the result
// BIT-IDENTICAL to Pillow; only the rows / columns that survive the crop are computed.
#include "common.hpp"

#define PREPROC_PRECISION_BITS 22

// Horizontal pass: dst[r][x][c] for r in [0,rows), x in [0,outw): input row (row0 + r), output column (x0 + x).
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, long src_pitch, unsigned char* __restrict__ dst,
                                                         const int* __restrict__ bounds, const int* __restrict__ coeffs, int ksize,
                                                         int row0, int rows, int x0, int outw) {
  const long total = (long)rows * outw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % outw), r = (int)(i / outw);
    const int xo = x0 + x;
    const int xmin = bounds[2 * xo], xcnt = bounds[2 * xo + 1];
    const int* k = coeffs + (long)xo * ksize;
    const unsigned char* p = src + (long)(row0 + r) * src_pitch + (long)xmin * 3;
    int s0 = 1 << (PREPROC_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int j = 0; j < xcnt; ++j) {
      const int w = k[j];
      s0 += p[3 * j] * w; s1 += p[3 * j + 1] * w; s2 += p[3 * j + 2] * w;
    }
    unsigned char* o = dst + i * 3;
    o[0] = (unsigned char)min(max(s0 >> PREPROC_PRECISION_BITS, 0), 255);
    o[1] = (unsigned char)min(max(s1 >> PREPROC_PRECISION_BITS, 0), 255);
    o[2] = (unsigned char)min(max(s2 >> PREPROC_PRECISION_BITS, 0), 255);
  }
}

struct ResampleVParams {
  const unsigned char* tmp;     // (rows, outw, 3) horizontal-pass output; its row 0 is input row `row0`
  const int* bounds; const int* coeffs; int ksize;
  int row0, y0, outh, outw;     // output rows [y0, y0+outh) of the resized image
  int flip;                     // horizontal flip after the crop (Image.FLIP_LEFT_RIGHT)
  int mode;                     // 0: u8 HWC (outh,outw,3) -- an intermediate image; 1: normalised planar float into a padded canvas
  void* out; int out_dt;        // mode 1: (3, padH, padW) f32 | bf16
  int padH, padW;
  float mean[3], stdinv[3];
  int to_rgb;                   // swap channels 0 and 2 before normalising (mmcv.imnormalize(to_rgb=True))
};

__global__ __launch_bounds__(256) void resample_v_kernel(ResampleVParams p) {
  const int W = p.mode == 1 ? p.padW : p.outw, H = p.mode == 1 ? p.padH : p.outh;
  const long total = (long)H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)(i / W);
    int v[3] = {0, 0, 0};
    const bool inside = y < p.outh && x < p.outw;
    if (inside) {
      const int yo = p.y0 + y;
      const int ymin = p.bounds[2 * yo], ycnt = p.bounds[2 * yo + 1];
      const int* k = p.coeffs + (long)yo * p.ksize;
      const int xs = p.flip ? p.outw - 1 - x : x;
      const unsigned char* q = p.tmp + ((long)(ymin - p.row0) * p.outw + xs) * 3;
      int s0 = 1 << (PREPROC_PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int j = 0; j < ycnt; ++j) {
        const int w = k[j];
        const unsigned char* r = q + (long)j * p.outw * 3;
        s0 += r[0] * w; s1 += r[1] * w; s2 += r[2] * w;
      }
      v[0] = min(max(s0 >> PREPROC_PRECISION_BITS, 0), 255);
      v[1] = min(max(s1 >> PREPROC_PRECISION_BITS, 0), 255);
      v[2] = min(max(s2 >> PREPROC_PRECISION_BITS, 0), 255);
    }
    if (p.mode == 0) {
      unsigned char* o = reinterpret_cast<unsigned char*>(p.out) + i * 3;
      o[0] = (unsigned char)v[0]; o[1] = (unsigned char)v[1]; o[2] = (unsigned char)v[2];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int cs = p.to_rgb ? 2 - c : c;
        // mmcv.imnormalize: cv2.subtract(img, mean) then cv2.multiply(img, 1/std), both in float32; the pad value is 0
        const float f = inside ? ((float)v[cs] - p.mean[c]) * p.stdinv[c] : 0.f;
        const long o = ((long)c * p.padH + y) * p.padW + x;
        if (p.out_dt == FAR3D_DT_F32) reinterpret_cast<float*>(p.out)[o] = f;
        else reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(f);
      }
    }
  }
}

extern "C" int far3d_image_resample_h(const unsigned char* src, long src_pitch, int in_h, int in_w, unsigned char* dst,
                                      const int32_t* bounds, const int32_t* coeffs, int ksize, int row0, int rows, int x0,
                                      int outw, void* stream) {
  FAR3D_CHECK_ARG(src && dst && bounds && coeffs, "far3d_image_resample_h: null pointer argument");
  FAR3D_CHECK_ARG(in_h > 0 && in_w > 0 && src_pitch >= (long)in_w * 3 && ksize > 0 && row0 >= 0 && rows > 0 && row0 + rows <= in_h && x0 >= 0 && outw > 0,
                  "far3d_image_resample_h: bad geometry");
  const long total = (long)rows * outw;
  long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, src_pitch, dst, bounds, coeffs,
                     ksize, row0, rows, x0, outw);
  FAR3D_CHECK_LAUNCH("far3d_image_resample_h");
  return FAR3D_OK;
}

extern "C" int far3d_image_resample_v(const unsigned char* tmp, int row0, int rows, int outw, const int32_t* bounds,
                                      const int32_t* coeffs, int ksize, int y0, int outh, int flip, int mode, void* out,
                                      int out_dt, int pad_h, int pad_w, const float* mean, const float* stdinv, int to_rgb,
                                      void* stream) {
  FAR3D_CHECK_ARG(tmp && bounds && coeffs && out, "far3d_image_resample_v: null pointer argument");
  FAR3D_CHECK_ARG(rows > 0 && outw > 0 && outh > 0 && ksize > 0 && y0 >= 0 && (mode == 0 || mode == 1), "far3d_image_resample_v: bad geometry");
  FAR3D_CHECK_ARG(mode == 0 || (mean && stdinv && pad_h >= outh && pad_w >= outw && (out_dt == FAR3D_DT_F32 || out_dt == FAR3D_DT_BF16)),
                  "far3d_image_resample_v: mode 1 needs mean/stdinv and a canvas at least as large as the crop");
  ResampleVParams p;
  p.tmp = tmp; p.bounds = bounds; p.coeffs = coeffs; p.ksize = ksize; p.row0 = row0; p.y0 = y0; p.outh = outh; p.outw = outw;
  p.flip = flip; p.mode = mode; p.out = out; p.out_dt = out_dt; p.padH = pad_h; p.padW = pad_w; p.to_rgb = to_rgb;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean ? mean[c] : 0.f; p.stdinv[c] = stdinv ? stdinv[c] : 1.f; }
  const long total = mode == 1 ? (long)pad_h * pad_w : (long)outh * outw;
  long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  FAR3D_CHECK_LAUNCH("far3d_image_resample_v");
  return FAR3D_OK;
}
