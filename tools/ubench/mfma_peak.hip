// Micro-benchmark: the bf16 MFMA rate this chip SUSTAINS (v_mfma_f32_32x32x16_bf16, independent accumulators, no memory traffic) and
// the shader clock it runs at meanwhile (s_memtime ticks / wall time).  The nominal peak (2.5 PFLOP/s) assumes 2.4 GHz; under a
// chip-wide MFMA load the clock is lower, and that sustained figure -- not the nominal one -- is what the bare MFMA loop of the
// convolution kernels (tools/conv_ablation.py, variant 27) should be compared with.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_bin/mfma_peak tools/ubench/mfma_peak.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* sink, long long* ticks) {
  f32x16_t acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8_t x, y;
#pragma unroll
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 3); y[e] = (__bf16)1.0f; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NACC>
static void run(int wgs_per_cu, int iters, float* sink, long long* ticks) {
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, iters, sink, ticks);
  hipDeviceSynchronize();
  float best = 1e30f; long long tk = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, iters, sink, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t; hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    if (ms < best) { best = ms; tk = t; }
  }
  const double flops = (double)grid * 4 * iters * NACC * 32768.0;
  // the stamped wave runs for ~the whole launch (one round of workgroups): ticks / time ~ the shader clock during the launch
  printf("{\"accumulators_per_wave\": %d, \"waves_per_simd\": %d, \"mfma_per_wave\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_2.5PF\": %.3f, \"s_memtime_ticks\": %lld, \"ticks_per_us\": %.1f}\n",
         NACC, wgs_per_cu, iters * NACC, best, flops / best / 1e9, flops / best / 1e9 / 2500.0, tk, tk / (best * 1e3));
}

int main() {
  float* sink; long long* ticks;
  hipMalloc(&sink, 64); hipMalloc(&ticks, 64);
  run<1>(1, 20000, sink, ticks);      // dependent chain: one accumulator, one wave per SIMD
  run<2>(1, 10000, sink, ticks);
  run<4>(1, 5000, sink, ticks);
  run<2>(2, 10000, sink, ticks);      // two waves per SIMD, two accumulators each (the 1 x 2 wave tiles of the conv kernels)
  run<2>(4, 10000, sink, ticks);      // four waves per SIMD
  run<4>(2, 20000, sink, ticks);      // long run: sustained clock
  run<4>(2, 100000, sink, ticks);
  return 0;
}
