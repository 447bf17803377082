// Micro-benchmark for one hypothesis about the conv / GEMM kernels' DMA streams: the activations are NHWC, so
// a K slab of a pixel tile is `P` pieces of B contiguous bytes, one per pixel, S bytes apart (S = 2 * channels) -- 128 B per 1536 B for
// the stage-2 concat GEMM, 64 B per 256 B for the stage-2 3x3 convolutions -- and the concat GEMMs fill LDS at a third of the rate the
// contiguous fill benchmark (fill_bw.hip) reaches from HBM.  This kernel streams exactly that access pattern -- a workgroup walks its own
// pixel tiles, K slab after K slab, with ONE stage in flight under the stage being "consumed" (the shipped double buffer) or with
// DEPTH stages in flight -- for piece sizes B = 64 .. 1024 bytes at a given row stride S, from a region that fits L2 and from one that
// does not.  If the fill rate from HBM rises with B, a channel-blocked activation layout ([C / 64][H][W][64]) is what the concat layers
// and the 3x3 layers are missing; if it does not, the stream is bound by latency x bytes in flight and only deeper rings help.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/strided_fill.hip -o tools/ubench/_bin/strided_fill && tools/ubench/_bin/strided_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 256 threads; a stage = P pixels x B bytes = 32 KiB (P = 32768 / B); ring of DEPTH + 1 stages.
template <int B, int DEPTH>
__global__ __launch_bounds__(256) void strided_fill_kernel(const unsigned char* src, unsigned long long region, int S, int tiles_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int P = 32768 / B, STAGE = 32768, RPI = 1024 / B, IPW = 8;      // rows per wave-instruction; instructions per wave per stage
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int slabs = S / B;                                                   // K slabs of a pixel tile
  const int row_in_instr = lane / (B / 16), col = (lane % (B / 16)) * 16;
  const unsigned long long tile_bytes = (unsigned long long)P * S;
  const unsigned long long ntiles_region = region / tile_bytes ? region / tile_bytes : 1;
  unsigned acc = 0;
  int stage = 0, issued = 0;
  const int total = tiles_per_wg * slabs;
  auto issue = [&](int step) {
    const unsigned long long tile = ((unsigned long long)blockIdx.x * tiles_per_wg + step / slabs) % ntiles_region;
    const unsigned char* base = src + tile * tile_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)tile_bytes, 0x00020000);
    const int kb = (step % slabs) * B;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int row = (wv * IPW + i) * RPI + row_in_instr;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + stage * STAGE + (wv * IPW + i) * 1024), 16, row * S + kb + col, 0, 0, 0);
    }
    stage = (stage + 1) % (DEPTH + 1);
    ++issued;
  };
  for (int s = 0; s < DEPTH && s < total; ++s) issue(s);
  for (int s = 0; s < total; ++s) {
    if (s + DEPTH < total) issue(s + DEPTH);
    // wait until stage s has landed: at most min(DEPTH, remaining) younger stages stay in flight
    const int younger = (s + DEPTH < total) ? DEPTH : (total - 1 - s);
    if (younger >= 3) wait_vmcnt<3 * IPW>();
    else if (younger == 2) wait_vmcnt<2 * IPW>();
    else if (younger == 1) wait_vmcnt<IPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    acc ^= reinterpret_cast<const unsigned*>(smem + (s % (DEPTH + 1)) * STAGE)[t];          // "consume": one LDS read per thread
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345678u) sink[0] = acc + issued;
}

template <int B, int DEPTH>
static void run(const unsigned char* src, size_t region, int S, unsigned* sink) {
  constexpr int LDSB = (DEPTH + 1) * 32768;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&strided_fill_kernel<B, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  const int blocks = 256 * (LDSB <= 81920 ? 2 : 1);          // as many workgroups per CU as the LDS footprint allows (<= 2)
  const int P = 32768 / B;
  const int tiles_per_wg = (int)(((size_t)192 << 20) / ((size_t)P * S) / blocks) + 1;       // ~192 MB of pixel rows walked per launch
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((strided_fill_kernel<B, DEPTH>), dim3(blocks), dim3(256), LDSB, 0, src, (unsigned long long)region, S, tiles_per_wg, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  const double bytes = (double)blocks * tiles_per_wg * (S / B) * 32768.0;
  printf("piece %4d B  stride %5d B  in flight %d x 32 KB  region %6.0f MB  WG/CU %d  %8.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", B, S, DEPTH,
         region / 1048576.0, blocks / 256, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
  hipEventDestroy(a); hipEventDestroy(b);
}

template <int B>
static void sweep(const unsigned char* src, int S, unsigned* sink) {
  if (S % B) return;
  for (size_t region : {(size_t)4 << 20, (size_t)1024 << 20}) {
    run<B, 1>(src, region, S, sink);
    run<B, 2>(src, region, S, sink);
    run<B, 3>(src, region, S, sink);
  }
}

int main() {
  unsigned char* src; unsigned* sink;
  const size_t cap = (size_t)1024 << 20;
  if (hipMalloc(&src, cap) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, cap);
  for (int S : {256, 1536, 3456}) {          // 128-channel rows (stage-2 3x3), the stage-2 concat rows (768 ch), the stage-4 concat rows (1728 ch)
    sweep<64>(src, S, sink);
    sweep<128>(src, S, sink);
    sweep<256>(src, S, sink);
    sweep<512>(src, S, sink);
  }
  // the contiguous reference: a pixel tile's whole rows (piece = stride)
  sweep<512>(src, 512, sink);
  sweep<1024>(src, 1024, sink);
  return 0;
}
