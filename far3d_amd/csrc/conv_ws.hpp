// Persistent, wave-specialised 3x3 / stride 1 / pad 1 convolution (round 6).
//
// What the stamps and ablations of rounds 4-5 said about conv3x3_pipe_kernel (profiles/r5/conv_ablation.txt, conv_phase_times.txt):
// its parts ADD instead of overlapping -- bare MFMA loop 49 us + bare DMA stream 43 us + epilogue 12 us ~ the 93 us of the whole
// (s2.c1, bf16) -- because every wave does everything in turn and all waves of the one resident workgroup meet at a barrier per step:
// a wave that sits at a `buffer_load ... lds` the memory pipeline has not accepted yet cannot issue its MFMAs, and a workgroup's
// set-up, first fill and epilogue (a quarter to a third of its life) run with the matrix pipe idle.  This kernel separates the roles:
//
//   * NP PRODUCER waves issue every LDS-DMA (`buffer_load ... lds`, 16 B per lane): one tap of the weights per step into a 3-deep
//     ring, the (TH+2) x 34 halo patch of the next 32-channel chunk into the other half of a double buffer.  They wait (counted
//     vmcnt: a wave's DMAs retire in order) only for the step the consumers need next, and they never touch the matrix pipe;
//   * WGM x WGN CONSUMER waves never issue a vector-memory instruction inside the K loop: fragments come from LDS (double-buffered
//     in registers, the loads of one k-half under the MFMAs of the previous one, ACROSS the step barrier), WM x WN tiles of 32 x 32
//     per wave on v_mfma_f32_32x32x16_bf16;
//   * the workgroup is PERSISTENT: it walks its tiles (static round-robin over the launch's tile list) as ONE stream of steps, so
//     the producers fill the ring for the first steps of tile t+1 while the consumers finish tile t; the epilogue has no LDS
//     staging and no barrier -- bias / activation / hi-lo split in registers, v_permlane32_swap pairs the half-waves' 8-byte pieces
//     into 16-byte stores -- so it costs its own instructions only and the next tile's operands are already resident when it ends.
//
// One barrier per step (= one tap of one 32-channel chunk), all waves.  What the barrier at the top of step S says:
//   producers -> consumers: the weights of step S (ring stage S % 3) and the patch of its chunk have landed (their vmcnt wait);
//   consumers -> producers: every fragment READ of step S - 1 has returned (s_waitcnt lgkmcnt(0)): stage (S - 1) % 3 and, at a
//   chunk boundary, the previous chunk's patch buffer may be overwritten -- which is where the producers put step S + 2 / chunk + 1.
// Arithmetic: products and their order are those of conv3x3_pipe_kernel (chunk-major, tap, k-half; pair mode: lo*hi', hi*lo', hi*hi'
// per k-half), so results are bit-identical to it (tests/test_igemm_gpu.py).
#pragma once
#include "igemm_kernels.hpp"
#include <type_traits>

// A buffer descriptor whose four words are PROVABLY wave-uniform (v_readfirstlane of the pointer halves and the size): hipcc otherwise keeps
// a descriptor that lives across loop iterations in VGPRs and wraps every `buffer_load ... lds` that uses it in a waterfall loop
// (MI355X guide T20) -- ten instructions and a serialisation per DMA piece.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, long nbytes) {
  const unsigned long long a = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int nb = __builtin_amdgcn_readfirstlane((int)(nbytes < 0x7fffffffL ? nbytes : 0x7fffffffL));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {      // s_waitcnt vmcnt(n), n wave-uniform, 0 <= n (clamped to 48: waiting for more than asked is always safe)
#define FAR3D_WVC(k) case k: wait_vmcnt<k>(); break;
  switch (n) {
    FAR3D_WVC(0) FAR3D_WVC(1) FAR3D_WVC(2) FAR3D_WVC(3) FAR3D_WVC(4) FAR3D_WVC(5) FAR3D_WVC(6) FAR3D_WVC(7) FAR3D_WVC(8)
    FAR3D_WVC(9) FAR3D_WVC(10) FAR3D_WVC(11) FAR3D_WVC(12) FAR3D_WVC(13) FAR3D_WVC(14) FAR3D_WVC(15) FAR3D_WVC(16)
    FAR3D_WVC(17) FAR3D_WVC(18) FAR3D_WVC(19) FAR3D_WVC(20) FAR3D_WVC(21) FAR3D_WVC(22) FAR3D_WVC(23) FAR3D_WVC(24)
    FAR3D_WVC(25) FAR3D_WVC(26) FAR3D_WVC(27) FAR3D_WVC(28) FAR3D_WVC(29) FAR3D_WVC(30) FAR3D_WVC(31) FAR3D_WVC(32)
    FAR3D_WVC(33) FAR3D_WVC(34) FAR3D_WVC(35) FAR3D_WVC(36) FAR3D_WVC(37) FAR3D_WVC(38) FAR3D_WVC(39) FAR3D_WVC(40)
    FAR3D_WVC(41) FAR3D_WVC(42) FAR3D_WVC(43) FAR3D_WVC(44) FAR3D_WVC(45) FAR3D_WVC(46) FAR3D_WVC(47)
    default: wait_vmcnt<48>(); break;
  }
#undef FAR3D_WVC
}

// Timing-only ablation switch of the probes (tools/probe/ws_conv_ab.py; results are WRONG when set): bit 0 the producers issue no DMA,
// bit 1 the consumers skip fragment reads and MFMAs.  PROFILING BUILD ONLY (libfar3d_hip_prof.so, -DFAR3D_PROFILING): the shipped
// library has neither the switch nor the setter, its kernels get a constant 0.
#ifdef FAR3D_PROFILING
extern std::atomic<int> g_ws_ablate;
#define FAR3D_WS_ABLATE_ARG g_ws_ablate.load(std::memory_order_relaxed)
#else
#define FAR3D_WS_ABLATE_ARG 0
#endif

// Profiling build: per-workgroup record of the persistent kernel (tools/probe/ws_conv_prof.py; 16 x uint64 per workgroup): slot 0 hardware
// id, 1 s_memtime at entry, 2 first step's operands landed (consumer wave 0 past the first barrier), 3 s_memtime at the end of consumer
// wave 0, 4 tiles processed, 5 steps, 6 / 7 s_memrealtime (100 MHz) at entry / end, 8 cycles consumer wave 0 spent between arriving at a
// step's barrier and leaving it, 9 cycles it spent in the steps' bodies (fragment reads + MFMA issue), 10 cycles in the epilogues.
template <int WGM, int WGN, int WM, int WN, int NP, bool PAIR, bool DBUF, int NSW = 3, bool FLAGS = false, int GRP = 1>
__global__ __launch_bounds__(64 * (WGM * WGN + NP)) void conv3x3_ws_kernel(IgemmParams P, int tiles_x, int tiles_y, int n_mt, int n_items, int ablate) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NWC = WGM * WGN;                      // consumer waves
  // NSW: weight ring depth, one tap per stage; the producers run NSW - 1 steps ahead of the consumers.  (Round 6, first measurement:
  // with 3 stages the persistent kernel only TIED the shipped one -- a step took 1.18 us against 0.73 us of MFMA work: two steps
  // (32 KB) in flight per CU do not cover the LDS-DMA round trip under load.)
  static_assert(NSW >= 3, "ring depth");
  // GRP: taps per hand-over (1, or 3 = one kernel row).  The per-step stamps of round 6 put the hand-over itself -- every wave of the
  // workgroup drained at a barrier, restart -- at ~345 cycles, against 770 (64 channels x 4 rows per consumer pair) to 1 540 cycles
  // of MFMA work per tap: with one barrier per kernel row that price is paid a third as often.  The ring then holds NSW / GRP groups
  // of GRP taps and the producers run NSW / GRP - 1 groups ahead.
  static_assert(GRP == 1 || (GRP == 3 && (NSW == 6 || NSW == 9) && !FLAGS), "one barrier per tap, or per kernel row with a ring of 2 or 3 rows");
  constexpr int NG = NSW / GRP;                       // groups in the ring
  constexpr int PLD = PAIR ? 2 : 1;                   // 64-byte planes per 32-channel chunk (pair storage: hi, lo)
  constexpr int BM = 32 * WGM * WM, TH = WGN * WN, PW = 34, PH = TH + 2, PPIX = PW * PH;
  constexpr int PG = (PPIX + 15) / 16, PATCH_B = PG * 1024;
  constexpr int GA = BM / 16, WST = BM * 64, WSTAGE = PLD * WST;
  constexpr int PBASE = NSW * WSTAGE, PBUF = PLD * PATCH_B;
  constexpr int RB = WN + 2;
  // FLAGS: the step hand-over through LDS counters instead of a workgroup barrier.  The barrier forces all consumer waves into lockstep
  // -- every step ends with the matrix pipes of the whole CU drained, 345 of 2120 cycles per step in the round-6 stamps -- although no
  // consumer ever needs another consumer: a consumer needs the producers' data (prod_done[p] = steps whose pieces of producer p have
  // landed), a producer needs the consumers' releases (cons_done[i] = steps whose fragment reads of consumer i have returned).  Both
  // are monotonic counters in LDS, written by one lane of their owner, polled by the other side (a read ahead of time, s_sleep in the
  // rare wait); every wait is bounded and a wave that gives up stops waiting for good (garbage out, never a hang).
  constexpr int FLAGB = PBASE + 2 * PBUF;               // [16] prod_done, [16] cons_done (u32)
  static_assert(FLAGB + (FLAGS ? 128 : 0) <= 163840, "LDS budget");
  static_assert(!FLAGS || (NP <= 4 && NWC <= 8), "flag words: one b128 read per 4 counters");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  [[maybe_unused]] volatile unsigned* const flg = reinterpret_cast<volatile unsigned*>(smem + FLAGB);
  if constexpr (FLAGS) {
    if (threadIdx.x < 32) flg[threadIdx.x] = 0u;
    __syncthreads();
  }
  [[maybe_unused]] bool gave_up = false;
  constexpr int SPIN_LIMIT = 1 << 18;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nchunks = P.cin_pad / 32;
  const int spt = 9 * nchunks;                                                   // steps per tile
  const int my_items = (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total_steps = my_items * spt, total_chunks = my_items * nchunks;
  const int Ktot = 9 * P.cin_pad * PLD;               // elements of a packed weight row
#ifdef FAR3D_PROFILING
  if (P.prof && t == 0) {
    unsigned hw_, xcc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    P.prof[(long)blockIdx.x * 16 + 0] = (unsigned long long)hw_ | ((unsigned long long)xcc_ << 32);
    P.prof[(long)blockIdx.x * 16 + 6] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    P.prof[(long)blockIdx.x * 16 + 1] = (unsigned long long)__builtin_amdgcn_s_memtime();
    P.prof[(long)blockIdx.x * 16 + 4] = (unsigned long long)my_items;
  }
#endif

  if (wv >= NWC) {
    // ============================================================ PRODUCER: every LDS-DMA of the workgroup
    const int pw = wv - NWC;
    constexpr int GPL = (PG + NP - 1) / NP, GWL = (GA + NP - 1) / NP;
    const int rg = lane >> 2, pc = lane & 3;
    const int lc = pc ^ ((rg >> 2) & 3);               // logical 16-byte chunk this lane fetches for physical chunk pc (XOR swizzle on the source)
    int npq = (PG - pw + NP - 1) / NP, nwq = (GA - pw + NP - 1) / NP;      // this wave's patch / weight slots (wave-uniform)
    npq = npq < 0 ? 0 : npq; nwq = nwq < 0 ? 0 : nwq;
    const int np = npq * PLD, nw = nwq * PLD;          // pieces per patch / per tap
    constexpr unsigned OOB = 0x80000000u;
    unsigned wvoff[GWL];
#pragma unroll
    for (int i = 0; i < GWL; ++i) wvoff[i] = (unsigned)((((pw + NP * i) * 16 + rg) * Ktot + lc * 8) * 2);
    const long img_bytes = (long)P.H * P.W * P.ldx * 2;
    // issue cursors: the weights of step SW (tile tiW) and the patch of global chunk GP (tile tiP) are what comes next
    unsigned pvoff[GPL];
    const bf16_t* x_base = nullptr;                    // image of the patch cursor's tile
    const bf16_t* w_base = nullptr;                    // first weight row of the weight cursor's tile
    long w_bytes = 0;
    auto decode = [&](int k, int& n, int& y0, int& x0, int& m0) __attribute__((always_inline)) {
      const int item = (int)blockIdx.x + k * (int)gridDim.x;
      const int mt = item % n_mt, pt = item / n_mt;
      const int tx = pt % tiles_x, r = pt / tiles_x;
      const int ty = r % tiles_y;
      n = __builtin_amdgcn_readfirstlane(r / tiles_y); y0 = __builtin_amdgcn_readfirstlane(ty * TH);
      x0 = __builtin_amdgcn_readfirstlane(tx * 32); m0 = __builtin_amdgcn_readfirstlane(mt * BM);
    };
    auto set_w_tile = [&](int k) __attribute__((always_inline)) {
      int n, y0, x0, m0;
      decode(k, n, y0, x0, m0);
      w_base = reinterpret_cast<const bf16_t*>(P.w) + (long)m0 * Ktot;
      w_bytes = w_tile_bytes(BM, m0, P.Cout, Ktot);
    };
    auto set_p_tile = [&](int k) __attribute__((always_inline)) {
      int n, y0, x0, m0;
      decode(k, n, y0, x0, m0);
      x_base = reinterpret_cast<const bf16_t*>(P.x) + (long)n * P.x_img_stride;
#pragma unroll
      for (int i = 0; i < GPL; ++i) {
        const int idx = (pw + NP * i) * 16 + rg;
        const int py = idx / PW, px = idx - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = idx < PPIX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;     // halo outside the image: hardware zero fill
        pvoff[i] = ok ? (unsigned)((((long)iy * P.W + ix) * P.ldx + lc * 8) * 2) : OOB;
      }
    };
    auto issue_w = [&](int step_in_tile, int stage) __attribute__((always_inline)) {      // one tap of one chunk: BM rows x 64 B per plane
      const int c = step_in_tile / 9, tap = step_in_tile - 9 * c;
      const int kb = __builtin_amdgcn_readfirstlane((tap * P.cin_pad + c * 32) * PLD * 2);      // byte offset of the tap's chunk inside a weight row
      const __amdgpu_buffer_rsrc_t rs_w = uniform_rsrc(w_base, w_bytes);
      const int sbase = __builtin_amdgcn_readfirstlane(stage * WSTAGE);
#pragma unroll
      for (int i = 0; i < GWL; ++i) {
        if (i < nwq) {                                                                // wave-uniform
#pragma unroll
          for (int pl = 0; pl < PLD; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + sbase + pl * WST + (pw + NP * i) * 1024), 16, wvoff[i],
                                                     kb + pl * 64, 0, 0);
        }
      }
    };
    auto issue_p = [&](int chunk_in_tile, int buf) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rs_x = uniform_rsrc(x_base, img_bytes);
      const int pbase = __builtin_amdgcn_readfirstlane(PBASE + buf * PBUF), coff = __builtin_amdgcn_readfirstlane(chunk_in_tile * (64 * PLD));
#pragma unroll
      for (int i = 0; i < GPL; ++i) {
        if (i < npq) {
#pragma unroll
          for (int pl = 0; pl < PLD; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(smem + pbase + pl * PATCH_B + (pw + NP * i) * 1024), 16, pvoff[i],
                                                     coff + pl * 64, 0, 0);
        }
      }
    };
    // cursors
    int sw_step = 0, sw_tile = 0, sw_stage = 0;        // next weight issue: step-in-tile, tile ordinal, ring stage
    int gp_chunk = 0, gp_tile = 0, gp_buf = 0;         // next patch issue: chunk-in-tile, tile ordinal, buffer
    int w_left = total_steps, p_left = total_chunks;
    set_w_tile(0); set_p_tile(0);
    auto next_w = [&]() __attribute__((always_inline)) {
      if (!(ablate & 1)) issue_w(sw_step, sw_stage);
      --w_left;
      sw_stage = sw_stage == NSW - 1 ? 0 : sw_stage + 1;
      if (++sw_step == spt) { sw_step = 0; ++sw_tile; if (w_left > 0) set_w_tile(sw_tile); }
    };
    auto next_p = [&]() __attribute__((always_inline)) {
      if (!(ablate & 1)) issue_p(gp_chunk, gp_buf);
      --p_left;
      gp_buf ^= 1;
      if (++gp_chunk == nchunks) { gp_chunk = 0; ++gp_tile; if (p_left > 0) set_p_tile(gp_tile); }
    };
    // prologue: P(0), then the weights of the first NG - 1 groups  (the patch first: what stays in flight at the first barrier are later groups only)
    next_p();
#pragma unroll
    for (int i = 0; i < (NG - 1) * GRP; ++i)
      if (w_left > 0) next_w();
    const int total_groups = total_steps / GRP, gpc = 9 / GRP;      // groups per chunk
    int gic = 0;                                         // group of the chunk (0: the chunk starts here)
    unsigned phist = 0;                                  // bit d: a patch was issued in iteration g - 1 - d
    for (int g = 0; g < total_groups; ++g) {
      // pieces issued after the last tap of group g, which may stay in flight: the weights of groups g+1 .. g+NG-2, and the patches issued
      // in iterations g-NG+1 (there after group g's weights) .. g-1
      int ahead = total_groups - 1 - g;
      ahead = ahead > NG - 2 ? NG - 2 : ahead;
      const int cnt = ahead * GRP * nw + __builtin_popcount(phist & ((1u << (NG - 1)) - 1u)) * np;
      const int S = g * GRP;
      if (ablate & 1) wait_vmcnt<0>(); else wait_vmcnt_dyn(cnt);
      if constexpr (FLAGS) {
        if (lane == 0) flg[pw] = (unsigned)(S + 1);      // this wave's pieces of steps <= S (and of S's patch) are in LDS
        // stage (S - 1) % NSW and, at a chunk boundary, the previous chunk's patch buffer are free once EVERY consumer has released step S - 1
        if (S > 0 && !(ablate & 8)) {
          for (int spins = 0; !gave_up; ++spins) {
            const u32x4_t c0 = *reinterpret_cast<volatile u32x4_t*>(smem + FLAGB + 64);
            u32x4_t c1 = c0;
            if constexpr (NWC > 4) c1 = *reinterpret_cast<volatile u32x4_t*>(smem + FLAGB + 80);
            unsigned m = c0.x;
            if constexpr (NWC > 1) m = min(m, c0.y);
            if constexpr (NWC > 2) m = min(m, c0.z);
            if constexpr (NWC > 3) m = min(m, c0.w);
            if constexpr (NWC > 4) m = min(m, c1.x);
            if constexpr (NWC > 5) m = min(m, c1.y);
            if constexpr (NWC > 6) m = min(m, c1.z);
            if constexpr (NWC > 7) m = min(m, c1.w);
            if (__builtin_amdgcn_readfirstlane(m) >= (unsigned)S) break;
            if (spins > SPIN_LIMIT) gave_up = true;
            __builtin_amdgcn_s_sleep(2);
          }
        }
      } else {
#ifdef FAR3D_PROFILING
        if (!(ablate & 8))
#endif
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < GRP; ++i)
        if (w_left > 0) next_w();                        // group g+NG-1 into the stages group g-1 has just released
      phist <<= 1;
      if (gic == 0 && p_left > 0) { next_p(); phist |= 1u; }      // P(chunk+1) into the buffer the previous chunk has released
      gic = gic == gpc - 1 ? 0 : gic + 1;
    }
    return;
  }

  // ============================================================== CONSUMER: LDS fragments -> MFMA -> register epilogue
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  // fragment read addresses for k-half 0 (bytes from the start of LDS, plane 0, stage 0 / current patch buffer); k-half 1 = ^ 32
  int aaddr[WM], baddr[RB][3];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    aaddr[i] = r * 64 + ((hi ^ ((r >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int idx = (wn * WN + r) * PW + l31 + kx;
      baddr[r][kx] = PBASE + idx * 64 + ((hi ^ ((idx >> 2) & 3)) << 4);
    }
  f32x16_t acc[WM][WN];
  u32x4_t fa[2][PLD][WM], fb[2][PLD][WN];              // fragment sets X = [0], Y = [1]

  // Fragment loads of one k-half, split into the weight (A) and the patch (B) part, and the products of one k-half split by term: a step
  // interleaves them so that no more than 12 ds_read_b128 are ever outstanding -- lgkmcnt is a 4-bit counter, and with the 16 reads of
  // two full sets in flight hipcc falls back to `s_waitcnt lgkmcnt(0)` before the first MFMA of the newer set, which exposed a whole LDS
  // round trip per step (~240 cycles of a 770-cycle body: tools/probe/ws_conv_prof.py, round 6).
  auto load_a = [&](int set, int soff, int kk) __attribute__((always_inline)) {      // soff: byte offset of the ring stage (wave-uniform)
#ifdef FAR3D_PROFILING
    if (ablate & 4) return;                              // timing-only: MFMAs on stale registers
#endif
    const int x32 = kk * 32;
#pragma unroll
    for (int pl = 0; pl < PLD; ++pl)
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[set][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + pl * WST + ((aaddr[i] ^ x32) + soff));
  };
  auto load_b = [&](int set, int tap, int kk) __attribute__((always_inline)) {
#ifdef FAR3D_PROFILING
    if (ablate & 4) return;
#endif
    const int ky = tap / 3, kx = tap - 3 * ky, x32 = kk * 32;
#pragma unroll
    for (int pl = 0; pl < PLD; ++pl)
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[set][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + pl * PATCH_B + (baddr[j + ky][kx] ^ x32));
  };
  auto load_set = [&](int set, int soff, int tap, int kk) __attribute__((always_inline)) { load_a(set, soff, kk); load_b(set, tap, kk); };
  // pair mode: term 0 = lo*hi', 1 = hi*lo', 2 = hi*hi' (small terms first; term-major so that consecutive MFMAs hit different accumulators)
  auto mma_term = [&](int set, int term) __attribute__((always_inline)) {
    const int pa = (PAIR && term == 0) ? 1 : 0, pb = (PAIR && term == 1) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], fa[set][pa][i], fb[set][pb][j]);
  };
  auto mma_set = [&](int set) __attribute__((always_inline)) {
    if constexpr (PAIR) { mma_term(set, 0); mma_term(set, 1); }
    mma_term(set, 2);
  };

  int G = 0;                                             // global chunk ordinal of this workgroup (patch buffer = G & 1)
  int soff = 0;                                          // ring stage of the current step, in bytes
  [[maybe_unused]] int Sg = 0;                           // FLAGS: global step ordinal
  [[maybe_unused]] u32x4_t pfl = {0u, 0u, 0u, 0u};       // FLAGS: the producers' counters as last read
#ifdef FAR3D_PROFILING
  unsigned long long pf_ta = 0, pf_tb = 0, pf_wait = 0, pf_body = 0, pf_epi = 0;
  const bool pf_on = P.prof != nullptr && wv == 0;
#endif
  for (int k = 0; k < my_items; ++k) {
    const int item = (int)blockIdx.x + k * (int)gridDim.x;
    const int mt = item % n_mt, pt = item / n_mt;
    const int tx = pt % tiles_x, rr = pt / tiles_x;
    const int ty = rr % tiles_y, n = rr / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, m0 = mt * BM;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int c = 0; c < nchunks; ++c, ++G) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int stage = soff;
        // every fragment read of the previous step has returned (the producers overwrite its stage / patch after this barrier)
        // (the BUILTIN, not inline asm: hipcc's wait-count pass then knows that nothing is outstanding at the top of a step and emits
        // counted lgkmcnt waits inside it; behind an opaque asm wait it re-waits lgkmcnt(0) before the first MFMA)
        if (tap % GRP == 0) {                            // hand-over points: the first tap of a group (compile-time after unrolling)
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0), vmcnt / expcnt untouched
#ifdef FAR3D_PROFILING
        if (pf_on) {                                     // stamps are read here, where the wave waits for lgkmcnt(0) anyway
          const unsigned long long now = __builtin_amdgcn_s_memtime();
          if (pf_tb) { pf_wait += pf_tb - pf_ta; pf_body += now - pf_tb; }
          pf_ta = now;
        }
#endif
        if constexpr (FLAGS) {
          // every fragment read of steps < S has returned: release them; then wait for the producers' pieces of step S (the counters
          // were read ahead, in the middle of the previous step: the common case takes no LDS round trip here)
          if (lane == 0) flg[16 + wv] = (unsigned)Sg;
          if (!(ablate & 8)) {
            for (int spins = 0; !gave_up; ++spins) {
              unsigned m = pfl.x;
              if constexpr (NP > 1) m = min(m, pfl.y);
              if constexpr (NP > 2) m = min(m, pfl.z);
              if constexpr (NP > 3) m = min(m, pfl.w);
              if (__builtin_amdgcn_readfirstlane(m) >= (unsigned)(Sg + 1)) break;
              if (spins > SPIN_LIMIT) gave_up = true;
              __builtin_amdgcn_s_sleep(1);
              pfl = *reinterpret_cast<volatile u32x4_t*>(smem + FLAGB);
            }
          }
          ++Sg;
        } else {
#ifdef FAR3D_PROFILING
          if (!(ablate & 8))
#endif
          __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef FAR3D_PROFILING
        if (pf_on) pf_tb = __builtin_amdgcn_s_memtime();
        if (tap == 0 && c == 0 && k == 0 && P.prof && t == 0) P.prof[(long)blockIdx.x * 16 + 2] = (unsigned long long)__builtin_amdgcn_s_memtime();
#endif
        }
        if (ablate & 2) { soff = soff == (NSW - 1) * WSTAGE ? 0 : soff + WSTAGE; continue; }      // timing-only: barriers alone
        if constexpr (DBUF) {
          load_set(0, stage, tap, 0);                    // X <- (S, k-half 0): 2 PLD (WM + WN) reads
          __builtin_amdgcn_sched_barrier(0);
          if (tap > 0 || c > 0) mma_set(1);              // Y = (S - 1, k-half 1): its latency-free MFMAs cover X's LDS round trip
          __builtin_amdgcn_sched_barrier(0);
          load_a(1, stage, 1);                           // Y's weight fragments; X has landed by now, the counter never holds two full sets
          if constexpr (FLAGS) pfl = *reinterpret_cast<volatile u32x4_t*>(smem + FLAGB);      // the producers' counters, for the next step's check
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (PAIR) {
            mma_term(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(1, tap, 1);                           // Y's patch fragments under X's remaining products
            __builtin_amdgcn_sched_barrier(0);
            mma_term(0, 1);
            mma_term(0, 2);
          } else {
            load_b(1, tap, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_term(0, 2);
          }
          __builtin_amdgcn_sched_barrier(0);
          soff = soff == (NSW - 1) * WSTAGE ? 0 : soff + WSTAGE;
        } else {
          load_set(0, stage, tap, 0);
          __builtin_amdgcn_sched_barrier(0);
          mma_set(0);
          __builtin_amdgcn_sched_barrier(0);
          load_set(0, stage, tap, 1);
          if constexpr (FLAGS) pfl = *reinterpret_cast<volatile u32x4_t*>(smem + FLAGB);
          __builtin_amdgcn_sched_barrier(0);
          mma_set(0);
          __builtin_amdgcn_sched_barrier(0);
          soff = soff == (NSW - 1) * WSTAGE ? 0 : soff + WSTAGE;
        }
      }
      // the next chunk's patch lives in the other buffer
      const int d = (G & 1) ? -PBUF : PBUF;
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) baddr[r][kx] += d;
    }
    if constexpr (DBUF) mma_set(1);                      // the last step's second k-half
    __builtin_amdgcn_sched_barrier(0);
#ifdef FAR3D_PROFILING
    unsigned long long pf_e0 = 0;
    if (pf_on) {
      pf_e0 = __builtin_amdgcn_s_memtime();
      if (pf_tb) { pf_wait += pf_tb - pf_ta; pf_body += pf_e0 - pf_tb; pf_tb = 0; }
    }
#endif

    // ---- epilogue in registers: lane (l31, hi) holds, per 32 x 32 tile and quad q, channels 8q + 4hi .. + 3 of pixel (row, l31).
    // v_permlane32_swap hands the upper half-wave's quad q to the lower lanes and the lower's quad q + 1 to the upper ones: every lane
    // then owns 8 consecutive channels = one 16-byte store per plane and quad pair.
    {
      const int px = x0 + l31;
      unsigned char* yb = reinterpret_cast<unsigned char*>(P.y);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int py = y0 + wn * WN + j;
        const bool ok = py < P.Ho && px < P.Wo;
        const long pix_b = ((long)n * P.y_img_stride + ((long)py * P.Wo + px) * P.ldy) * 2;      // bytes
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const int cb = m0 + (wm * WM + i) * 32;        // first channel of this 32-channel block
          if (cb >= P.Cout) continue;                    // wave-uniform (Cout % 32 == 0)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {               // quad pairs (0,1) and (2,3)
            uint2 h[2], l[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int q = 2 * qp + u;
              float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
              if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + cb + 8 * q + 4 * hi);
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
              if (P.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
              } else if (P.act == ACT_SWISH) {
                if constexpr (PAIR) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.f / (1.f + expf(-v[e])));
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = v[e] * __frcp_rn(1.f + __expf(-v[e]));
                }
              }
              if constexpr (PAIR) split4f(v[0], v[1], v[2], v[3], h[u], l[u]);
              else { h[u] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); l[u] = make_uint2(0u, 0u); }
            }
            // lower lanes: [own quad 2qp | upper's quad 2qp]; upper lanes: [lower's quad 2qp+1 | own quad 2qp+1]
            u32x4_t oh, ol;
            {
              const auto sx = __builtin_amdgcn_permlane32_swap(h[0].x, h[1].x, false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(h[0].y, h[1].y, false, false);
              oh = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            }
            const int qoff = (2 * qp + hi) * 16;           // byte offset of the lane's 8 channels inside a 64-byte plane row
            if constexpr (PAIR) {
              const auto sx = __builtin_amdgcn_permlane32_swap(l[0].x, l[1].x, false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(l[0].y, l[1].y, false, false);
              ol = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
              if (ok) {
                unsigned char* d = yb + pix_b + (long)(cb >> 5) * 128 + qoff;
                *reinterpret_cast<u32x4_t*>(d) = oh;
                *reinterpret_cast<u32x4_t*>(d + 64) = ol;
              }
            } else {
              if (ok) *reinterpret_cast<u32x4_t*>(yb + pix_b + (long)cb * 2 + qoff) = oh;
            }
          }
        }
      }
    }
#ifdef FAR3D_PROFILING
    if (pf_on) pf_epi += __builtin_amdgcn_s_memtime() - pf_e0;
#endif
  }
#ifdef FAR3D_PROFILING
  if (P.prof && t == 0) {
    P.prof[(long)blockIdx.x * 16 + 3] = (unsigned long long)__builtin_amdgcn_s_memtime();
    P.prof[(long)blockIdx.x * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    P.prof[(long)blockIdx.x * 16 + 5] = (unsigned long long)total_steps;
    P.prof[(long)blockIdx.x * 16 + 8] = pf_wait; P.prof[(long)blockIdx.x * 16 + 9] = pf_body; P.prof[(long)blockIdx.x * 16 + 10] = pf_epi;
  }
#endif
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// The same division of labour for the 1x1 / stride 1 layers on pair-stored maps (the OSA concat GEMMs and the FPN laterals: 3.3 ms of the
// in-tolerance frame at 260-330 useful TFLOP/s against the 3x3 layers' 395).  A GEMM has no patch to re-use: EVERY 32-channel step
// brings (BM + BP) x 128 bytes through LDS-DMA, 32 pieces for a 128 x 128 tile, and in gemm1x1_pipe_kernel each of the 8 waves pays
// the issue cost of its 4 pieces (100-185 cycles apiece among MFMAs and fragment reads, MI355X guide) on top of its 384 cycles of MFMA
// work per step, then meets the others at a barrier.  Here NP producer waves issue all of them -- weights and activation rows into one
// ring of NSW stages, GRP steps per hand-over -- and the consumers do what the 3x3 consumers do: fragments from LDS (two register sets,
// skewed across the hand-over), MFMAs, a register epilogue with 16-byte stores.  The workgroup is persistent over (pixel tile, channel
// tile) items in the XCD-aware order of gemm1x1_pipe_kernel; its steps form ONE stream across tiles (a hand-over group may straddle two
// tiles).  Products and their order are those of gemm_step_split: bit-identical results.
// Channel sums (P.chan_sums: the eSE pooling of the concat layers, fixed point as in epilogue_rows16 -- rint(v * 2^FRAC_BITS) of every
// STORED hi and lo value, integer addition, so the result is bit-identical to every other tile's): a consumer sums its quantised values
// over its WN sub-tiles per lane, over the 32 pixels of the half-wave on the VALU (four DPP row rotations + v_permlane16_swap; 32 bits
// while every |v| < 64, else two 16-bit limbs), and lanes 0 / 32 add the wave's 16 WM channel sums into a per-workgroup LDS array
// lsum[tile parity][image slot][BM] (64-bit LDS atomics).  Every consumer runs a tile's epilogue between the SAME two hand-over barriers and drains its LDS counter before a
// barrier, so lsum[T & 1] is complete once the barrier that follows tile T's last step has passed: right behind it every consumer wave
// flushes ITS slice of the array (one global atomic per non-zero (image, channel)) and zeroes it, long before tile T + 2 adds to it again
// (a tile is >= 2 hand-over groups).  The producers stay out of it: an atomic in their vector-memory queue would break the counted
// vmcnt their ring rests on.  No residual or second output: far3d_conv2d_nhwc refuses the tile for a call that needs them.
template <int WGM, int WGN, int WM, int WN, int NP, int NSW, int GRP, bool SUMS>
__global__ __launch_bounds__(64 * (WGM * WGN + NP)) void gemm1x1_ws_kernel(IgemmParams P, int npt, int nct, int ablate) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NWC = WGM * WGN;
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr int GA = BM / 16, GB = BP / 16, SLOTS = GA + GB;      // 1 KB DMA pieces per 64-byte plane of a step
  static_assert(SLOTS % NP == 0, "every producer issues the same number of pieces per step (counted vmcnt)");
  constexpr int SPW = SLOTS / NP;                                 // slots per producer wave and step; a slot = the hi and the lo piece
  constexpr int SUB = (BM + BP) * 64, STAGE = 2 * SUB;            // a stage: [hi plane: A rows | B rows][lo plane: the same]
  static_assert(NSW % GRP == 0 && NSW / GRP >= 2, "ring = at least two hand-over groups");
  constexpr int NG = NSW / GRP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int SUMB = NSW * STAGE;                               // lsum[2 tile parities][2 image slots][BM] int64 behind the ring
  static_assert(SUMB + 4 * BM * 8 <= 163840, "LDS budget");
  [[maybe_unused]] long long* const lsum = reinterpret_cast<long long*>(smem + SUMB);
  constexpr bool do_sums = SUMS;                                  // a second instantiation: the sums cost ~25 VGPRs the plain kernel keeps
  if (do_sums) {
    for (int i = threadIdx.x; i < 4 * BM; i += 64 * (NWC + NP)) lsum[i] = 0;
    __syncthreads();
  }
  const int HoWo = P.Ho * P.Wo;
  const long Npix = (long)P.N * HoWo;
  const int nsteps = P.cin_pad / 32;                              // steps per tile (pair storage: one 32-channel block = 2 planes)
  const int Ktot = P.cin_pad * 2;
  // items of this workgroup: L = blockIdx.x + k * gridDim.x (gridDim.x % 8 == 0), xcd = L & 7, slot = L >> 3, channel tile = slot % nct,
  // pixel tile = (slot / nct) * 8 + xcd; the valid ones (pixel tile < npt) are the first my_items of the sequence
  const int xcd = (int)blockIdx.x & 7, slot0 = (int)blockIdx.x >> 3, dslot = (int)gridDim.x >> 3;
  const int vslots = ((npt - xcd + 7) >> 3) * nct;                // slots with a valid pixel tile on this XCD
  const int my_items = vslots > slot0 ? (vslots - slot0 + dslot - 1) / dslot : 0;
  const int total_steps = my_items * nsteps;
  const int total_groups = (total_steps + GRP - 1) / GRP;
  auto decode = [&](int k, int& p0, int& m0) __attribute__((always_inline)) {
    const int slot = slot0 + k * dslot;
    const int ct = slot % nct, pt = (slot / nct) * 8 + xcd;
    p0 = __builtin_amdgcn_readfirstlane(pt * BP); m0 = __builtin_amdgcn_readfirstlane(ct * BM);
  };

  if (wv >= NWC) {
    // ============================================================ PRODUCER
    const int pw = wv - NWC;
    const int rg = lane >> 2, pc = lane & 3;
    const int lc = pc ^ ((rg >> 2) & 3);
    constexpr unsigned OOB = 0x80000000u;
    const long x_bytes = ((long)(P.N - 1) * P.x_img_stride + (long)HoWo * P.ldx) * 2;
    unsigned voff[SPW];                                // per-lane source byte offsets of this wave's slots (weights: constant; pixels: per tile)
    int sdst[SPW];                                     // slot's byte offset inside a plane of a stage
    bool isw[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int q = pw + NP * i;                       // pw is wave-uniform: the weight / activation branches below are scalar
      isw[i] = q < GA;
      const int g = isw[i] ? q : q - GA;
      sdst[i] = (isw[i] ? 0 : BM * 64) + g * 1024;
      voff[i] = isw[i] ? (unsigned)(((g * 16 + rg) * Ktot + lc * 8) * 2) : 0u;
    }
    const bf16_t* w_base = nullptr;
    long w_bytes = 0;
    int cur_tile = 0, cur_step = 0, cur_stage = 0, left = total_steps;
    auto set_tile = [&](int k) __attribute__((always_inline)) {
      int p0, m0;
      decode(k, p0, m0);
      w_base = reinterpret_cast<const bf16_t*>(P.w) + (long)m0 * Ktot;
      w_bytes = w_tile_bytes(BM, m0, P.Cout, Ktot);
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        if (!isw[i]) {
          const int g = pw + NP * i - GA;
          const long p = (long)p0 + g * 16 + rg;
          if (p < Npix) {
            int n, rem;
            pix_split(p, HoWo, n, rem);
            voff[i] = (unsigned)(((long)n * P.x_img_stride + (long)rem * P.ldx + lc * 8) * 2);
          } else {
            voff[i] = OOB;
          }
        }
      }
    };
    auto next = [&]() __attribute__((always_inline)) {            // issue the cursor's step into the cursor's stage, advance
      if (!(ablate & 1)) {
        const __amdgpu_buffer_rsrc_t rs_w = uniform_rsrc(w_base, w_bytes);
        const __amdgpu_buffer_rsrc_t rs_x = uniform_rsrc(P.x, x_bytes);
        const int sbase = __builtin_amdgcn_readfirstlane(cur_stage * STAGE);
        const int kb = __builtin_amdgcn_readfirstlane(cur_step * 128);
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            if (isw[i])
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + sbase + pl * SUB + sdst[i]), 16, voff[i], kb + pl * 64, 0, 0);
            else
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t*)(smem + sbase + pl * SUB + sdst[i]), 16, voff[i], kb + pl * 64, 0, 0);
          }
        }
      }
      --left;
      cur_stage = cur_stage == NSW - 1 ? 0 : cur_stage + 1;
      if (++cur_step == nsteps) { cur_step = 0; ++cur_tile; if (left > 0) set_tile(cur_tile); }
    };
    if (my_items > 0) set_tile(0);
#pragma unroll
    for (int i = 0; i < (NG - 1) * GRP; ++i)
      if (left > 0) next();
    for (int g = 0; g < total_groups; ++g) {
      // steps issued after the last step of group g (they may stay in flight): those of groups g+1 .. g+NG-2
      int later = total_steps - (g + 1) * GRP;
      later = later < 0 ? 0 : (later > (NG - 2) * GRP ? (NG - 2) * GRP : later);
      if (ablate & 1) wait_vmcnt<0>(); else wait_vmcnt_dyn(later * SPW * 2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < GRP; ++i)
        if (left > 0) next();                            // group g+NG-1 into the stages group g-1 has just released
    }
    if (do_sums) __builtin_amdgcn_s_barrier();           // the consumers' matching barrier follows their last epilogue (they flush the last sums)
    return;
  }

  // ============================================================== CONSUMER
  const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
  int aaddr[WM], baddr[WN];                              // k-half 0, hi plane, stage 0; k-half 1 = ^ 32; lo plane = + SUB
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int r = (wm * WM + i) * 32 + l31;
    aaddr[i] = r * 64 + ((hi ^ ((r >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int r = (wn * WN + j) * 32 + l31;
    baddr[j] = BM * 64 + r * 64 + ((hi ^ ((r >> 2) & 3)) << 4);
  }
  f32x16_t acc[WM][WN];
  u32x4_t fa[2][2][WM], fb[2][2][WN];                    // [set X / Y][plane][tile]
  auto load_a = [&](int set, int soff, int kk) __attribute__((always_inline)) {
    const int x32 = kk * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[set][pl][i] = *reinterpret_cast<const u32x4_t*>(smem + pl * SUB + ((aaddr[i] ^ x32) + soff));
  };
  auto load_b = [&](int set, int soff, int kk) __attribute__((always_inline)) {
    const int x32 = kk * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[set][pl][j] = *reinterpret_cast<const u32x4_t*>(smem + pl * SUB + ((baddr[j] ^ x32) + soff));
  };
  auto mma_term = [&](int set, int term) __attribute__((always_inline)) {      // 0: lo * hi', 1: hi * lo', 2: hi * hi' (gemm_step_split's order)
    const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) mma<bf16_t>(acc[i][j], fa[set][pa][i], fb[set][pb][j]);
  };
  int soff = 0, gs = 0;                                  // ring stage of the current step (bytes); step ordinal inside the hand-over group
  [[maybe_unused]] int flush_tile = 0, gcount = 0;       // sums: next tile to flush; hand-over barriers passed
  auto flush = [&](int T) __attribute__((always_inline)) {      // this wave's slice of lsum[T & 1] -> P.chan_sums, zeroed
    int fp0, fm0;
    decode(T, fp0, fm0);
    const int fcam0 = (int)((long)fp0 / P.sums_hw);
    long long* src = lsum + (T & 1) * 2 * BM;
    for (int i = wv * 64 + lane; i < 2 * BM; i += 64 * NWC) {
      const int sl = i / BM, chn = i - sl * BM, cam = fcam0 + sl;
      const long long v = src[i];
      if (v != 0) {
        src[i] = 0;
        if (cam < P.N && fm0 + chn < P.Cout)
          atomicAdd(reinterpret_cast<unsigned long long*>(P.chan_sums + (long)cam * P.Cout + fm0 + chn), (unsigned long long)v);
      }
    }
  };
  for (int k = 0; k < my_items; ++k) {
    int p0, m0;
    decode(k, p0, m0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < nsteps; ++s) {
      const int stage = soff;
      if (gs == 0) {                                     // hand-over: every fragment read of the previous group has returned
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (do_sums) {
          // this was the barrier of group gcount: the tiles whose last step lies in an earlier group have their sums complete
          while (flush_tile < k && ((flush_tile + 1) * nsteps - 1) / GRP < gcount) { flush(flush_tile); ++flush_tile; }
          ++gcount;
        }
      }
      gs = gs == GRP - 1 ? 0 : gs + 1;
      if (ablate & 2) { soff = soff == (NSW - 1) * STAGE ? 0 : soff + STAGE; continue; }
      load_a(0, stage, 0); load_b(0, stage, 0);          // X <- (s, k-half 0)
      __builtin_amdgcn_sched_barrier(0);
      if (s > 0) { mma_term(1, 0); mma_term(1, 1); mma_term(1, 2); }      // Y = (s - 1, k-half 1)
      __builtin_amdgcn_sched_barrier(0);
      load_a(1, stage, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_term(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_b(1, stage, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_term(0, 1);
      mma_term(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      soff = soff == (NSW - 1) * STAGE ? 0 : soff + STAGE;
    }
    if (!(ablate & 2)) { mma_term(1, 0); mma_term(1, 1); mma_term(1, 2); }
    __builtin_amdgcn_sched_barrier(0);

    // ---- register epilogue (the 3x3 kernel's): bias / activation / hi-lo split, v_permlane32_swap pairs the half-waves' 8-byte pieces
    {
      unsigned char* yb = reinterpret_cast<unsigned char*>(P.y);
      // channel sums: the tile's pixels lie in at most two images (sums_hw >= BP, checked at launch)
      const int cam0 = do_sums ? (int)((long)p0 / P.sums_hw) : 0;
      const long bnd = (long)(cam0 + 1) * P.sums_hw;
      const bool two = do_sums && (long)p0 + BP > bnd;       // workgroup-uniform: the tile spans an image boundary
      long pix_b[WN];
      bool okj[WN], secj[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const long p = (long)p0 + (wn * WN + j) * 32 + l31;
        okj[j] = p < Npix;
        secj[j] = p >= bnd;                                  // this lane's pixel belongs to image cam0 + 1
        pix_b[j] = 0;
        if (okj[j]) {
          int n, rem;
          pix_split(p, HoWo, n, rem);
          pix_b[j] = ((long)n * P.y_img_stride + (long)rem * P.ldy) * 2;
        }
      }
      // the stored hi / lo words of quad q of tile (i, j): channels cb + 8 q + 4 hi .. + 3 at this lane's pixel
      auto stored = [&](int i, int j, int q, int cb, uint2& h, uint2& l) __attribute__((always_inline)) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias) b4 = *reinterpret_cast<const float4*>(P.bias + cb + 8 * q + 4 * hi);
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
        split4f(v[0], v[1], v[2], v[3], h, l);
      };
      // this lane's quantised values of quad q: qs[4 q + e] += rint(hi * 2^F) + rint(lo * 2^F)   (epilogue_rows16's sums_add8, per element)
      auto quantise = [&](const uint2& h, const uint2& l, int q, int (&qs)[16], float& mx) __attribute__((always_inline)) {
        const unsigned w4[4] = {h.x, h.y, l.x, l.y};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float h0 = __uint_as_float(w4[e] << 16), h1 = __uint_as_float(w4[e] & 0xffff0000u);
          const float l0 = __uint_as_float(w4[2 + e] << 16), l1 = __uint_as_float(w4[2 + e] & 0xffff0000u);
          mx = fmaxf(mx, fmaxf(fabsf(h0), fabsf(h1)));
          qs[4 * q + 2 * e] += __float2int_rn(h0 * (float)(1 << FAR3D_SUMS_FRAC_BITS)) + __float2int_rn(l0 * (float)(1 << FAR3D_SUMS_FRAC_BITS));
          qs[4 * q + 2 * e + 1] += __float2int_rn(h1 * (float)(1 << FAR3D_SUMS_FRAC_BITS)) + __float2int_rn(l1 * (float)(1 << FAR3D_SUMS_FRAC_BITS));
        }
      };
      // the half-wave's 32 pixels added up on the VALU -- four DPP row rotations, then v_permlane16_swap pairs the two 16-lane rows of the
      // half -- lanes 0 / 32 -> lsum[tile parity][slot][channel] (64-bit LDS atomics)
      auto half_sum = [&](int x) __attribute__((always_inline)) -> int {
        x += __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true);      // row_ror:8
        x += __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, true);      // row_ror:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, true);      // row_ror:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, true);      // row_ror:1
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        const u2_t r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        const unsigned rx = r.x, ry = r.y;
        return (int)(rx + ry);
      };
      auto reduce_add = [&](const int (&qs)[16], float mx, int sl, int cb) __attribute__((always_inline)) {
        // a |v| >= 64 could overflow the 32-bit sum over the half-wave (2 WN values of < 2^24 per lane, 32 lanes): then two 16-bit limbs
        // -- wave-uniform, practically never taken (epilogue_rows16's sums_add8_wide is the same escape).  |v| >= 2^12 saturates the
        // 32-bit quantisation itself: the caller re-quantises such a tile in 64 bits (quantise_wide)
        const bool wide = WN > 2 || __any(mx >= (float)(1 << (24 - FAR3D_SUMS_FRAC_BITS)));
        long long* dst = lsum + ((k & 1) * 2 + sl) * BM + (cb - m0) + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          long long tot;
          if (!wide) {
            tot = (long long)half_sum(qs[r]);
          } else {
            const int lo16 = qs[r] & 0xffff, hi16 = qs[r] >> 16;      // qs = hi16 * 65536 + lo16, lo16 in [0, 65535]: limb sums cannot overflow
            tot = (long long)half_sum(hi16) * 65536ll + (long long)half_sum(lo16);
          }
          if (l31 == 0 && tot != 0) atomicAdd(reinterpret_cast<unsigned long long*>(dst + 8 * (r >> 2) + (r & 3)), (unsigned long long)tot);
        }
      };
      // one 32-channel block of the wave's tile (an explicit instantiation per block: left to `#pragma unroll` the body is too large for
      // the unroller, and a dynamic index into the accumulators would put them into scratch memory)
      auto channel_block = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int cb = m0 + (wm * WM + i) * 32;              // first channel of this 32-channel block
        if (cb >= P.Cout) return;                            // wave-uniform (Cout % 32 == 0)
        int qs[16];
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) qs[r] = 0;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            uint2 h[2], l[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              stored(i, j, 2 * qp + u, cb, h[u], l[u]);
              if (do_sums && okj[j] && !secj[j]) quantise(h[u], l[u], 2 * qp + u, qs, mx);
            }
            // lower lanes: [own quad 2qp | upper's quad 2qp]; upper lanes: [lower's quad 2qp+1 | own quad 2qp+1]
            u32x4_t oh, ol;
            {
              const auto sx = __builtin_amdgcn_permlane32_swap(h[0].x, h[1].x, false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(h[0].y, h[1].y, false, false);
              oh = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            }
            {
              const auto sx = __builtin_amdgcn_permlane32_swap(l[0].x, l[1].x, false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(l[0].y, l[1].y, false, false);
              ol = u32x4_t{sx[0], sy[0], sx[1], sy[1]};
            }
            if (okj[j]) {
              unsigned char* d = yb + pix_b[j] + (long)(cb >> 5) * 128 + (2 * qp + hi) * 16;
              *reinterpret_cast<u32x4_t*>(d) = oh;
              *reinterpret_cast<u32x4_t*>(d + 64) = ol;
            }
          }
        }
        if constexpr (do_sums) {
          // |v| >= 2^11: a lane's 32-bit sum of 2 WN quantised values could overflow (and from 2^13 the 32-bit quantisation saturates):
          // such a tile is summed again from the accumulators in 64 bits, element by element (any finite magnitude; never taken in practice)
          auto exact_add = [&](int sl) __attribute__((always_inline)) {
            long long* dst = lsum + ((k & 1) * 2 + sl) * BM + (cb - m0) + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              long long x[4] = {0, 0, 0, 0};
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                uint2 h, l;
                stored(i, j, q, cb, h, l);
                if (okj[j] && secj[j] == (sl == 1)) {
                  const unsigned w4[4] = {h.x, h.y, l.x, l.y};
#pragma unroll
                  for (int e = 0; e < 2; ++e) {
                    const float h0 = __uint_as_float(w4[e] << 16), h1 = __uint_as_float(w4[e] & 0xffff0000u);
                    const float l0 = __uint_as_float(w4[2 + e] << 16), l1 = __uint_as_float(w4[2 + e] & 0xffff0000u);
                    x[2 * e] += __double2ll_rn((double)h0 * (double)(1 << FAR3D_SUMS_FRAC_BITS)) + __double2ll_rn((double)l0 * (double)(1 << FAR3D_SUMS_FRAC_BITS));
                    x[2 * e + 1] += __double2ll_rn((double)h1 * (double)(1 << FAR3D_SUMS_FRAC_BITS)) + __double2ll_rn((double)l1 * (double)(1 << FAR3D_SUMS_FRAC_BITS));
                  }
                }
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {                  // four 16-bit limbs (the top one signed): limb sums over 32 lanes cannot overflow
                const long long v = x[e];
                const long long tot = (long long)half_sum((int)(v & 0xffff)) + ((long long)half_sum((int)((v >> 16) & 0xffff)) << 16) +
                                      ((long long)half_sum((int)((v >> 32) & 0xffff)) << 32) + ((long long)half_sum((int)(v >> 48)) << 48);
                if (l31 == 0 && tot != 0) atomicAdd(reinterpret_cast<unsigned long long*>(dst + 8 * q + e), (unsigned long long)tot);
              }
            }
          };
          const bool huge = __any(mx >= 2048.f);               // wave-uniform
          if (huge) exact_add(0); else reduce_add(qs, mx, 0, cb);
          if (two) {                                         // the pixels of image cam0 + 1: their values once more from the accumulators (a few tiles per launch)
            float mx1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) qs[r] = 0;
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                uint2 h, l;
                stored(i, j, q, cb, h, l);
                if (okj[j] && secj[j]) quantise(h, l, q, qs, mx1);
              }
            if (__any(mx1 >= 2048.f)) exact_add(1); else reduce_add(qs, mx1, 1, cb);
          }
        }
      };
      static_assert(WM <= 4, "channel blocks per consumer");
      channel_block(std::integral_constant<int, 0>{});
      if constexpr (WM > 1) channel_block(std::integral_constant<int, 1>{});
      if constexpr (WM > 2) channel_block(std::integral_constant<int, 2>{});
      if constexpr (WM > 3) channel_block(std::integral_constant<int, 3>{});
    }
  }
  if constexpr (do_sums) {                               // the producers' matching barrier: the last tiles' sums are complete behind it
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    while (flush_tile < my_items) { flush(flush_tile); ++flush_tile; }
  }
#endif
}

template <int WGM, int WGN, int WM, int WN, int NP, int NSW, int GRP>
static int launch_gemm1x1_ws(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, BP = 32 * WGN * WN;
  constexpr size_t lds = (size_t)NSW * 2 * (BM + BP) * 64 + 4 * BM * 8;      // ring + lsum[2][2][BM]
  static_assert(lds <= 163840, "LDS budget");
  const long Npix = (long)P.N * P.Ho * P.Wo;
  const int npt = (int)((Npix + BP - 1) / BP), nct = (P.Cout + BM - 1) / BM;
  static std::atomic<unsigned long long> lds_ok_s{0};
  static std::atomic<int> n_cu{0};
  int cus = n_cu.load(std::memory_order_relaxed);
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    n_cu.store(cus, std::memory_order_relaxed);
  }
  constexpr int wg_per_cu = lds <= 81920 && 64 * (WGM * WGN + NP) <= 1024 ? 2 : 1;
  const long items = (long)((npt + 7) / 8 * 8) * nct;
  long grid = (long)cus * wg_per_cu;
  grid = (grid + 7) / 8 * 8;
  if (items < grid) grid = items;                       // items is a multiple of 8
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds > 65536)
    if (const int rc = far3d_allow_lds(P.chan_sums ? reinterpret_cast<const void*>(&gemm1x1_ws_kernel<WGM, WGN, WM, WN, NP, NSW, GRP, true>)
                                                   : reinterpret_cast<const void*>(&gemm1x1_ws_kernel<WGM, WGN, WM, WN, NP, NSW, GRP, false>),
                                       (int)lds, P.chan_sums ? lds_ok_s : lds_ok, "far3d_conv2d_nhwc")) return rc;
  if (P.chan_sums) {
    if (P.sums_hw < BP || P.cin_pad / 32 < 2 * GRP) {
      far3d_set_error("far3d_conv2d_nhwc: channel sums on this wave-specialised GEMM tile need Ho*Wo (%d) >= %d pixels and Cin >= %d", P.sums_hw, BP, 64 * GRP);
      return FAR3D_ERR_ARG;
    }
    hipLaunchKernelGGL((gemm1x1_ws_kernel<WGM, WGN, WM, WN, NP, NSW, GRP, true>), dim3((unsigned)grid), dim3(64 * (WGM * WGN + NP)), lds, st, P, npt, nct, FAR3D_WS_ABLATE_ARG);
  } else {
    hipLaunchKernelGGL((gemm1x1_ws_kernel<WGM, WGN, WM, WN, NP, NSW, GRP, false>), dim3((unsigned)grid), dim3(64 * (WGM * WGN + NP)), lds, st, P, npt, nct, FAR3D_WS_ABLATE_ARG);
  }
  return 0;
}

template <int WGM, int WGN, int WM, int WN, int NP, bool PAIR, bool DBUF, int NSW = 3, bool FLAGS = false, int GRP = 1>
static int launch_conv3x3_ws(const IgemmParams& P, hipStream_t st) {
  constexpr int BM = 32 * WGM * WM, TH = WGN * WN, PLD = PAIR ? 2 : 1;
  constexpr int PG = (34 * (TH + 2) + 15) / 16;
  constexpr size_t lds = (size_t)NSW * PLD * BM * 64 + (size_t)2 * PLD * PG * 1024 + (FLAGS ? 128 : 0);
  static_assert(lds <= 163840, "LDS budget");
  const int tiles_x = (P.Wo + 31) / 32, tiles_y = (P.Ho + TH - 1) / TH, n_mt = (P.Cout + BM - 1) / BM;
  const int n_items = P.N * tiles_x * tiles_y * n_mt;
  static std::atomic<int> n_cu{0};
  int cus = n_cu.load(std::memory_order_relaxed);
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    n_cu.store(cus, std::memory_order_relaxed);
  }
  constexpr int wg_per_cu = lds <= 81920 && 64 * (WGM * WGN + NP) <= 1024 ? 2 : 1;
  const int grid = n_items < cus * wg_per_cu ? n_items : cus * wg_per_cu;
  static std::atomic<unsigned long long> lds_ok{0};
  if constexpr (lds > 65536)
    if (const int rc = far3d_allow_lds(reinterpret_cast<const void*>(&conv3x3_ws_kernel<WGM, WGN, WM, WN, NP, PAIR, DBUF, NSW, FLAGS, GRP>), (int)lds, lds_ok, "far3d_conv2d_nhwc")) return rc;
  hipLaunchKernelGGL((conv3x3_ws_kernel<WGM, WGN, WM, WN, NP, PAIR, DBUF, NSW, FLAGS, GRP>), dim3((unsigned)grid), dim3(64 * (WGM * WGN + NP)), lds, st, P, tiles_x, tiles_y, n_mt,
                     n_items, FAR3D_WS_ABLATE_ARG);
  return 0;
}
