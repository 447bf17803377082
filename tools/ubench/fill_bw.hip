// Micro-benchmark: global -> LDS / global -> VGPR fill throughput per CU on gfx950 (sets the ceiling for the conv tiles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// mode 0: LDS-DMA 16 B/lane ; mode 1: global_load_dwordx4 -> VGPR (xor-reduced) ; mode 2: VGPR + ds_write_b128
template <int MODE, int LDSB>
__global__ __launch_bounds__(256) void fill_kernel(const unsigned char* src, size_t region, int iters, unsigned* sink, int stride_blocks) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  // each workgroup walks its own 16 KB tiles through the region (region small -> L2 hits; large -> HBM/MALL)
  size_t off = ((size_t)blockIdx.x * 16384) % region;
  u32x4_t accv = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const unsigned char* p = src + off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // 4 x 1 KB per wave = 16 KB per WG per iteration
      const unsigned char* g = p + (wv * 4 + i) * 1024 + lane * 16;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)(smem + ((it & (LDSB / 16384 - 1)) * 16384) + (wv * 4 + i) * 1024), 16, 0, 0);
      } else {
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(g);
        if (MODE == 2) *reinterpret_cast<u32x4_t*>(smem + ((it & (LDSB / 16384 - 1)) * 16384) + (wv * 4 + i) * 1024 + lane * 16) = v;
        else accv ^= v;
      }
    }
    if (MODE == 0 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    off = (off + (size_t)stride_blocks * 16384) % region;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned r = accv[0] ^ accv[1] ^ accv[2] ^ accv[3];
  if (MODE != 1) r ^= reinterpret_cast<unsigned*>(smem)[t];
  if (r == 0x12345678u) sink[0] = r;
}

template <int MODE, int LDSB>
static void run(const char* name, const unsigned char* src, size_t region, int blocks, int iters, unsigned* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<MODE, LDSB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((fill_kernel<MODE, LDSB>), dim3(blocks), dim3(256), LDSB, 0, src, region, iters, sink, blocks);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)blocks * iters * 16384;
  printf("%-28s region=%6.1f MB blocks=%5d lds/WG=%3d KB  %8.1f us  %6.2f TB/s  (%5.1f B/clk/CU @2.1GHz)\n", name, region / 1048576.0, blocks, LDSB / 1024,
         ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.1e9);
}

int main() {
  unsigned char* src; unsigned* sink;
  const size_t cap = (size_t)512 << 20;
  hipMalloc(&src, cap); hipMemset(src, 1, cap); hipMalloc(&sink, 64);
  for (size_t region : {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)512 << 20}) {
    for (int wgs_per_cu : {1, 2, 4}) {
      const int blocks = 256 * wgs_per_cu, iters = 2048 / wgs_per_cu;
      run<0, 32768>("lds-dma", src, region, blocks, iters, sink);
      run<1, 16384>("vgpr", src, region, blocks, iters, sink);
      run<2, 32768>("vgpr+ds_write", src, region, blocks, iters, sink);
    }
  }
  run<0, 65536>("lds-dma 64KB ring", src, (size_t)1 << 20, 512, 1024, sink);
  return 0;
}
