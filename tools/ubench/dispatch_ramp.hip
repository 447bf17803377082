// How long does the chip take to START a one-round launch, and does it depend on the workgroup shape?  The aggregation kernel runs
// 1544 workgroups of 128 threads (one query each), all co-resident; its launch lasts one wave's life plus the dispatch ramp
// (DESIGN 3.1).  This probe launches Q "queries" of work as workgroups of 128 / 256 / 512 threads (1 / 2 / 4 queries per
// workgroup), each wave spinning for a fixed time, with the aggregation kernel's LDS (14.6 KB per query) and register footprint
// (launch bounds of 4 waves per SIMD), and reports the launch span (HIP events over a graph-free stream of back-to-back launches)
// and the spread of the workgroups' start stamps (s_memrealtime, 100 MHz, comparable across CUs).
//   hipcc --offload-arch=gfx950 -O3 -o _bin/dispatch_ramp dispatch_ramp.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int THREADS>
__global__ __launch_bounds__(THREADS, 4) void spin_kernel(long long* stamps, int spin_ticks) {
  extern __shared__ float lds[];
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  if (threadIdx.x < 32) lds[threadIdx.x] = (float)t0;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0 && stamps) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime() + (long long)lds[1] * 0;
  }
}

template <int THREADS>
static void run(int queries, int spin_ticks, long long* d_stamps) {
  const int qpw = THREADS / 128, grid = (queries + qpw - 1) / qpw;
  const size_t lds = 14656 * qpw;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(spin_kernel<THREADS>, dim3(grid), dim3(THREADS), lds, 0, nullptr, spin_ticks);
  hipDeviceSynchronize();
  const int iters = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(spin_kernel<THREADS>, dim3(grid), dim3(THREADS), lds, 0, nullptr, spin_ticks);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipLaunchKernelGGL(spin_kernel<THREADS>, dim3(grid), dim3(THREADS), lds, 0, d_stamps, spin_ticks);
  hipDeviceSynchronize();
  std::vector<long long> st(2 * grid);
  hipMemcpy(st.data(), d_stamps, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost);
  std::vector<long long> s0(grid), s1(grid);
  for (int i = 0; i < grid; ++i) { s0[i] = st[2 * i]; s1[i] = st[2 * i + 1]; }
  std::sort(s0.begin(), s0.end()); std::sort(s1.begin(), s1.end());
  printf("%4d threads x %4d workgroups (%d queries each), spin %.1f us: launch %.2f us back to back; first start -> median start %.2f us, "
         "-> p90 %.2f, -> last start %.2f us; first start -> last end %.2f us\n",
         THREADS, grid, qpw, spin_ticks * 0.01, ms * 1e3 / iters, (s0[grid / 2] - s0[0]) * 0.01, (s0[grid * 9 / 10] - s0[0]) * 0.01,
         (s0[grid - 1] - s0[0]) * 0.01, (s1[grid - 1] - s0[0]) * 0.01);
}

int main(int argc, char** argv) {
  const int queries = argc > 1 ? atoi(argv[1]) : 1544;
  long long* d_stamps;
  hipMalloc(&d_stamps, sizeof(long long) * 2 * queries);
  for (int spin : {0, 500, 1000}) {
    run<128>(queries, spin, d_stamps);
    run<256>(queries, spin, d_stamps);
    run<512>(queries, spin, d_stamps);
  }
  return 0;
}
