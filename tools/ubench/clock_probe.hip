// What does s_memtime count, and how fast does the shader clock run while a kernel is busy?  (the conv / GEMM
// kernels' per-workgroup stamps, tools/conv_phase_times.py, come out at 1.6 - 2.1 GHz against the 100 MHz s_memrealtime.)
//   1. an idle chip: one wave sleeps a KNOWN number of shader cycles (s_sleep 127 = 127 * 64 cycles, 2000 times) and stamps both
//      counters around it -> ticks per slept cycle (1.0 = s_memtime is the shader clock) and MHz;
//   2. the same probe wave while every CU runs (a) a pure MFMA loop, (b) an MFMA loop fed by LDS reads and global loads -- the clock the
//      probe sees is the clock the busy kernel runs at.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_probe.hip -o tools/ubench/_bin/clock_probe && tools/ubench/_bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe_kernel(unsigned long long* out, int reps) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

// mode 0: MFMA only; mode 1: MFMA + LDS fragment reads + a global load stream (what a conv workgroup does)
__global__ __launch_bounds__(512) void load_kernel(const float4* __restrict__ src, float* sink, long n4, int iters, int mode) {
  __shared__ float4 lds[4096];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) % n4;
  for (int it = 0; it < iters; ++it) {
    if (mode) {
      const float4 l = lds[(threadIdx.x * 7 + it) & 4095];
      const float4 v = src[idx];
      idx += (long)gridDim.x * blockDim.x; if (idx >= n4) idx -= n4;
      g.x += l.x + v.x; g.y += l.y + v.y;
      a[0] = (__bf16)g.x;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = g.x + g.y;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) sink[0] = s;
}

int main() {
  unsigned long long* out; hipMalloc(&out, 16);
  float* sink; hipMalloc(&sink, 4);
  const long n4 = (512L << 20) / 16;                       // 512 MB stream: past the Infinity Cache
  float4* src; hipMalloc(&src, n4 * 16); hipMemset(src, 0, n4 * 16);
  hipStream_t s_load, s_probe; hipStreamCreate(&s_load); hipStreamCreateWithPriority(&s_probe, hipStreamNonBlocking, -1);
  const int reps = 2000;                                    // 2000 * 127 * 64 = 16.256 M cycles (~7-10 ms)
  const char* names[3] = {"idle chip", "under a pure MFMA loop on every CU", "under MFMA + LDS reads + a global load stream on every CU"};
  for (int trial = 0; trial < 3; ++trial) {
    hipDeviceSynchronize();
    if (trial) hipLaunchKernelGGL(load_kernel, dim3(256 * 2), dim3(512), 0, s_load, src, sink, n4, trial == 1 ? 120000 : 60000, trial - 1);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s_probe, out, reps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStreamSynchronize(s_probe);
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipDeviceSynchronize();
    const double slept = (double)reps * 127 * 64;
    printf("%-62s s_memtime ticks %10llu  s_memrealtime ticks %8llu (100 MHz)  -> s_memtime runs at %7.1f MHz, %.3f ticks per slept cycle\n",
           names[trial], h[0], h[1], 100.0 * (double)h[0] / (double)h[1], (double)h[0] / slept);
  }
  return 0;
}
