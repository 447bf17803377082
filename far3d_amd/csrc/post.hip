// Post-processing and per-frame calibration kernels: the pieces of the frame that used to
// run as chains of ATen / rocPRIM / rocSOLVER launches (topk, sort, exp/atan2/cat, linalg.inv, nan_to_num).  All are
// latency-bound single-workgroup (or tiny) kernels: the point is ONE graph node each instead of 5-20.
//
//   far3d_topk            descending top-K of n <= 40960 floats in one workgroup (bisection on the ordered bit pattern with the
//                         values held in registers, ordered compaction, bitonic sort of the K winners).
//                         Replaces torch.topk in post_update_memory, ref models/dense_heads/farhead.py:488-491.
//   far3d_decode_topk     NMSFreeCoder.decode_single + FarHead.get_bboxes: sigmoid, top-max_num over A*num_classes, label/query
//                         split, denormalize_bbox (exp, atan2), post_center_range mask, z -= h/2.
//                         ref core/bbox/coders/nms_free_coder.py:39-91, core/bbox/util.py:25-52, farhead.py:1224-1245.
//   far3d_camera_prep     img2lidar = inverse(lidar2img) (ref farhead.py:798) and the 14-float MLN camera code
//                         (ref farhead.py:553-556).
//   far3d_agg_order       camera / image-cell sorted query order for far3d_aggregate_forward (scheduling only).
//   far3d_nan_to_num      torch.nan_to_num on the stacked decoder outputs (ref farhead.py:646) (+ optional bf16 copy).
#include "common.hpp"
#include "agg_tables.hpp"

#ifdef FAR3D_PROFILING
// tools/topk_phase_times.py: stage stamps (s_memtime) of block_topk_sorted, written by thread 0 of workgroup 0: 8 x uint64
__device__ unsigned long long* d_topk_ts = nullptr;
extern "C" int far3d_prof_set_topk_timestamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(d_topk_ts), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#define TOPK_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && d_topk_ts) d_topk_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define TOPK_NOTE(i, v) do { if (threadIdx.x == 0 && blockIdx.x == 0 && d_topk_ts) d_topk_ts[i] = (unsigned long long)(v); } while (0)
#else
#define TOPK_STAMP(i) do { } while (0)
#define TOPK_NOTE(i, v) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------ block-wide top-K
// order-preserving map float -> uint (larger float <-> larger uint; -0 == +0; every NaN, whatever its sign bit, maps to the
// largest key and so sorts above +inf, like torch.topk)
__device__ __forceinline__ unsigned ord_key(float f) {
  if (f != f) return 0xffffffffu;
  unsigned b = __float_as_uint(f);
  if (b == 0x80000000u) b = 0u;      // -0 == +0 (torch.topk compares values)
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_val(unsigned k) {     // inverse of ord_key (the NaN key comes back as a NaN)
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

#define TOPK_THREADS 1024
#define TOPK_MAXV 40          // values per thread held in registers: n <= 40960
#define TOPK_MAXK 1024
#define TOPK_CAND 4096        // candidates the single-wave bisection ranks out of LDS
#define TOPK_SAMPLE 2048      // sample (two register slots per thread) that places the candidate threshold of a large input

// ---- block-wide top-K (round 5, second form).  Measured on the frame's shapes (tools/topk_phase_times.py, profiles/r5): with 40
// values per thread EVERY pass over the registers costs ~3 k cycles on the one CU (16 waves on 4 SIMDs), and a 16-wave bisection
// pass has a floor of ~1.2 k cycles (LDS partials + barrier) even with one value per thread -- 15-21 passes were half the kernel,
// the 40 serialised global loads (one `s_waitcnt vmcnt(0)` each: the bounds check kept them apart) another third.  Now:
//   1. all loads in flight together (clamped index, padding applied afterwards);
//   2. n > TOPK_CAND: a 2048-key sample (every n/2048-th input, LDS) is bisected by ONE wave -- no barrier, no partial sums: ballot + s_bcnt1 over
//      `ds_read_b128`s -- to a threshold T whose sample rank promises ~2-4 K keys >= T; ONE pass over the registers gathers those
//      candidates (key, index) into LDS.  n <= TOPK_CAND: every key is a candidate;
//   3. wave 0 bisects the candidates to a cut that keeps between K and CAPW of them (or, when a tie group straddles every such cut,
//      to the exact K-th key and the index bound inside its tie group);
//   4. the kept candidates are ordered by a counting rank (all 16 waves) and the first K are the result.
// The candidate set holds EVERY key >= T, so steps 3-4 see exactly the order the whole input has; when the sample misleads (fewer
// than K or more than TOPK_CAND keys >= T: heavy ties, adversarial layouts) the 16-wave bisection over the registers -- the previous
// form, kept whole -- takes over.  Result and tie rule (lower index first) are those of torch.topk on every path.

// keys[0 .. 1024 * n1024) >= thr, counted by one wave: four b128 per lane and iteration in flight, wave-uniform result
// (the trip count is in whole groups so that the body can be unrolled by hand: a loop around a ballot is not unrolled by the compiler)
__device__ __forceinline__ int wave_count_ge(const unsigned* keys, int n1024, unsigned thr, int lane) {
  int c = 0;
  for (int j = 0; j < n1024; ++j) {
    uint4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const uint4*>(keys + j * 1024 + u * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      c += __popcll(__ballot(q[u].x >= thr)) + __popcll(__ballot(q[u].y >= thr)) + __popcll(__ballot(q[u].z >= thr)) + __popcll(__ballot(q[u].w >= thr));
  }
  return c;
}
// #{ j : keys[j] == key && idx[j] <= bound }
__device__ __forceinline__ int wave_count_eq_idx(const unsigned* keys, const unsigned* idx, int nk256, unsigned key, unsigned bound, int lane) {
  int c = 0;
  for (int j = 0; j < nk256; ++j) {
    const uint4 q = *reinterpret_cast<const uint4*>(keys + j * 256 + lane * 4), d = *reinterpret_cast<const uint4*>(idx + j * 256 + lane * 4);
    c += __popcll(__ballot(q.x == key && d.x <= bound)) + __popcll(__ballot(q.y == key && d.y <= bound)) +
         __popcll(__ballot(q.z == key && d.z <= bound)) + __popcll(__ballot(q.w == key && d.w <= bound));
  }
  return c;
}

// wave-wide max of an unsigned (all lanes get it): DPP inside the 16-lane rows, permlane swaps across them -- no LDS round trips
template <int CTRL> __device__ __forceinline__ unsigned topk_dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  v = max(v, topk_dpp<0xB1>(v));      // quad_perm:[1,0,3,2]
  v = max(v, topk_dpp<0x4E>(v));      // quad_perm:[2,3,0,1]
  v = max(v, topk_dpp<0x124>(v));     // row_ror:4
  v = max(v, topk_dpp<0x128>(v));     // row_ror:8
  u2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  unsigned rx = r.x, ry = r.y;
  v = max(rx, ry);
  r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  rx = r.x; ry = r.y;
  return max(rx, ry);
}
// smallest non-zero and largest key of keys[0 .. 1024 * n1024), by one wave (0 = padding; every lane gets both)
__device__ __forceinline__ void wave_minmax(const unsigned* keys, int n1024, int lane, unsigned& mn, unsigned& mx) {
  unsigned hi = 0u, lo_inv = 0u;            // lo_inv = max over (~key) of the real keys
  for (int j = 0; j < n1024; ++j) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 q = *reinterpret_cast<const uint4*>(keys + j * 1024 + u * 256 + lane * 4);
      hi = max(max(hi, q.x), max(q.y, max(q.z, q.w)));
      lo_inv = max(lo_inv, q.x ? ~q.x : 0u); lo_inv = max(lo_inv, q.y ? ~q.y : 0u);
      lo_inv = max(lo_inv, q.z ? ~q.z : 0u); lo_inv = max(lo_inv, q.w ? ~q.w : 0u);
    }
  }
  mx = wave_max_u32(hi);
  mn = ~wave_max_u32(lo_inv);
}
// Binary search by ONE wave (no barriers) for a cut with Klo <= count(key >= cut) <= Khi over keys[0 .. 1024 * n1024), started from the
// keys' own range [min, max + 1) instead of [1, 2^32): the frame's logits sit in a band 2^23 keys wide, which is 9 passes less.
// window = true: lo = such a cut.  Otherwise hi - lo == 1 with count(key >= lo) >= Klo > count(key >= hi) = cnt_hi.
// Needs count(key != 0) >= Klo.  Wave-uniform results.
__device__ __forceinline__ void wave_search(const unsigned* keys, int n1024, int Klo, int Khi, int lane, unsigned long long& lo,
                                            unsigned long long& hi, int& cnt_hi, bool& window) {
  unsigned mn, mx;
  wave_minmax(keys, n1024, lane, mn, mx);
  lo = (unsigned long long)__builtin_amdgcn_readfirstlane(mn);
  hi = (unsigned long long)__builtin_amdgcn_readfirstlane(mx) + 1ull;
  cnt_hi = 0;
  window = false;
  while (hi - lo > 1ull) {
    const unsigned mid = (unsigned)(lo + ((hi - lo) >> 1));
    const int c = wave_count_ge(keys, n1024, mid, lane);
    if (c >= Klo && c <= Khi) { lo = mid; window = true; return; }
    if (c > Khi) lo = mid; else { hi = mid; cnt_hi = c; }
  }
}

// Common tail: sel[0..c) holds the kept entries (any order) -> sorted descending.  Counting rank: entry e goes to position
// #{e' > e} (entries are distinct: the index is part of them).
// Stage 1 ranks by the 32-bit keys alone -- 4 keys per 16-byte broadcast read, one v_cmp + one add each -- with all 1024 threads
// at work: a thread keeps TWO entries (e, e + cp/2; cp = 2^m >= c) against one of 2048 / cp key ranges, the partial ranks meet in
// LDS atomics.  (The transposed form -- a lane keeps keys, a wave walks entries whose key is a scalar, ballot + s_bcnt1 count 64
// compares per v_cmp -- has a third of the VALU work and measured 1.6x SLOWER: 11-13 k cycles against 7-8 k, the v_cmp -> s_bcnt1
// -> s_add chains of one wave do not overlap; profiles/r5/topk_phase_times.txt.)
// Stage 2: entries with equal keys share their stage-1 position r, and the g of them own sel[r .. r + g): each parks itself in a
// staging copy of that range (arrival order), then ranks itself among the g by index (lower index = larger entry first).
// All threads call; s_k32: >= 16 KB of scratch (keys, staging); s_rank / s_slot zeroed by the caller.
__device__ __forceinline__ void topk_rank_place(unsigned long long* sel, int c, unsigned* s_k32, int* s_rank, int* s_slot) {
  const int t = threadIdx.x;
  unsigned long long* stage = reinterpret_cast<unsigned long long*>(s_k32 + 2 * TOPK_MAXK);      // [TOPK_MAXK] entries
  const unsigned long long mine = t < c ? sel[t] : 0ull;
  s_k32[t] = (unsigned)(mine >> 32);                  // 0 past the entries: below every real key
  __syncthreads();
  {
    int lg = 6;
    while ((1 << lg) < c) ++lg;                       // cp = 2^lg >= c, 64 .. 1024
    const int half = 1 << (lg - 1), P = (2 * TOPK_THREADS) >> lg;
    const int e0 = t & (half - 1), e1 = e0 + half, part = t >> (lg - 1);
    const int len = ((c + P - 1) / P + 15) & ~15;
    const int j0 = part * len, j1 = min(j0 + len, (c + 15) & ~15);
    if (e0 < c && j0 < j1) {
      const unsigned k0 = s_k32[e0], k1 = e1 < c ? s_k32[e1] : 0xffffffffu;
      int r0 = 0, r1 = 0;
      for (int j = j0; j < j1; j += 16) {
        const uint4 a = *reinterpret_cast<const uint4*>(s_k32 + j), b = *reinterpret_cast<const uint4*>(s_k32 + j + 4);
        const uint4 d = *reinterpret_cast<const uint4*>(s_k32 + j + 8), f = *reinterpret_cast<const uint4*>(s_k32 + j + 12);
        const unsigned q[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w, f.x, f.y, f.z, f.w};
#pragma unroll
        for (int u = 0; u < 16; ++u) { r0 += q[u] > k0 ? 1 : 0; r1 += q[u] > k1 ? 1 : 0; }
      }
      if (r0) atomicAdd(&s_rank[e0], r0);
      if (r1 && e1 < c) atomicAdd(&s_rank[e1], r1);
    }
  }
  __syncthreads();
  TOPK_STAMP(4);
  int r = 0;
  if (t < c) {
    r = s_rank[t];
    const int q = atomicAdd(&s_slot[r], 1);           // arrival order inside the group of equal keys
    stage[r + q] = mine;
  }
  __syncthreads();
  if (t < c) {
    const int g = s_slot[r];
    int rr = r;
#pragma unroll 4
    for (int j = 0; j < g; ++j) rr += stage[r + j] > mine ? 1 : 0;      // g == 1: nothing is greater
    sel[rr] = mine;
  }
  __syncthreads();
}

// The 16-wave form over the register-held keys v[] (any input; the fall-back of block_topk_sorted).  A pass of the bisection =
// one v_cmp per value whose wave-wide result lands in an SGPR pair (ballot), s_bcnt1 + s_add on the scalar unit, one LDS word per
// wave, ONE barrier.  Leaves the kept entries (K <= count <= CAPW) in sel, unordered, and returns their number.
// Needs *s_cnt == 0 on entry (visible to all threads).
template <int MAXV>
__device__ __forceinline__ int topk_select_registers(const unsigned (&v)[MAXV], int n, int K, int CAPW, unsigned long long* sel, int* s_cnt) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __shared__ __attribute__((aligned(16))) int s_part[2][TOPK_THREADS / 64];
  int pass = 0;
#define TOPK_COUNT(dst, pred)                                              \
  {                                                                        \
    int c_ = 0;                                   /* wave-uniform */       \
    _Pragma("unroll") for (int k = 0; k < MAXV; ++k) {                     \
      const unsigned key = v[k];                                           \
      const int idx = t + k * TOPK_THREADS;                                \
      (void)idx;                                                           \
      c_ += __popcll(__ballot(pred));                                      \
    }                                                                      \
    if (lane == 0) s_part[pass & 1][wv] = c_;                              \
    __syncthreads();                                                       \
    const int4* pp_ = reinterpret_cast<const int4*>(s_part[pass & 1]);     \
    int tot_ = 0;                                                          \
    _Pragma("unroll") for (int w = 0; w < TOPK_THREADS / 256; ++w) { const int4 q_ = pp_[w]; tot_ += (q_.x + q_.y) + (q_.z + q_.w); } \
    ++pass;                                                                \
    dst = tot_;                                                            \
  }
  // invariant: count(key >= lo) >= K, count(key >= hi) < K   (hi = 2^32 as 64-bit).  The bisection stops as soon as SOME cut keeps
  // between K and CAPW entries (`key >= mid` selects a superset of the top-K that holds every entry of any tie group it touches).
  unsigned long long lo = 1ull, hi = 0x100000000ull;
  int n_ge = -1;
  bool window = false;
  while (hi - lo > 1ull) {
    const unsigned mid = (unsigned)(lo + ((hi - lo) >> 1));
    int c;
    TOPK_COUNT(c, key >= mid);
    if (c >= K && c <= CAPW) { lo = mid; n_ge = c; window = true; break; }      // block-uniform
    if (c > K) lo = mid; else hi = mid;
  }
  const unsigned kth = (unsigned)lo;
  int n_gt = 0;
  int idx_cut = 0x7fffffff;                           // among the keys == kth, keep indices <= idx_cut
  if (!window) {
    if (n_ge < 0) TOPK_COUNT(n_ge, key >= kth);
    if (n_ge > K) {                                   // ties at the cut (block-uniform): keep the lowest indices
      if (kth != 0xffffffffu) TOPK_COUNT(n_gt, key > kth);
      const int need = K - n_gt;
      int ilo = -1, ihi = n - 1;                      // count(idx <= ilo) < need <= count(idx <= ihi)
      while (ihi - ilo > 1) {
        const int mid = ilo + ((ihi - ilo) >> 1);
        int c;
        TOPK_COUNT(c, key == kth && idx <= mid);
        if (c >= need) ihi = mid; else ilo = mid;
      }
      idx_cut = ihi;
    }
  }
#undef TOPK_COUNT
  // compaction, order arbitrary: a wave reserves its range with ONE LDS atomic, lanes take positions from the ballot masks
#define TOPK_TAKEN(k) (v[k] > kth || (v[k] == kth && (window || (t + (k) * TOPK_THREADS) <= idx_cut)))
  {
    int wc = 0;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) wc += __popcll(__ballot(TOPK_TAKEN(k)));
    int base = 0;
    if (lane == 0 && wc) base = atomicAdd(s_cnt, wc);
    int run = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const bool tk = TOPK_TAKEN(k);
      const unsigned long long m = __ballot(tk);
      if (m == 0ull) continue;
      if (tk) {
        const int pos = run + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (pos < TOPK_MAXK) sel[pos] = ((unsigned long long)v[k] << 32) | (unsigned long long)(0xffffffffu - (unsigned)(t + k * TOPK_THREADS));
      }
      run += __popcll(m);
    }
  }
#undef TOPK_TAKEN
  __syncthreads();
  return min(*s_cnt, TOPK_MAXK);
}

// Selects the K largest of vals[0..n) (ties -> lower index first) and leaves them sorted descending in
// sel[0..K) as (ord_key << 32) | (0xffffffff - index).  All TOPK_THREADS threads must call; sel is LDS [TOPK_MAXK].
template <int MAXV>      // values per thread held in registers: n <= TOPK_THREADS * MAXV
__device__ __forceinline__ void block_topk_sorted(const float* __restrict__ vals, int n, int K, unsigned long long* sel) {
  static_assert(TOPK_THREADS == TOPK_MAXK && TOPK_CAND == 4 * TOPK_THREADS && TOPK_SAMPLE == 2 * TOPK_THREADS, "one entry per thread in the rank stage");
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __shared__ __attribute__((aligned(16))) unsigned s_ckey[TOPK_CAND];    // candidate keys;    later the rank stage's keys
  __shared__ __attribute__((aligned(16))) unsigned s_cidx[TOPK_CAND];    // candidate indices; first the sample, later the rank stage's entries
  __shared__ int s_rank[TOPK_MAXK], s_slot[TOPK_MAXK];
  __shared__ int s_cnt;
  __shared__ unsigned s_ctl[4];
  TOPK_STAMP(0);
  unsigned v[MAXV];
  unsigned smp0 = 0u, smp1 = 0u;                      // n > TOPK_CAND: two entries of an evenly strided sample of the input
  {
    float f[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) f[k] = vals[min(t + k * TOPK_THREADS, n - 1)];       // all loads in flight together
    if (n > TOPK_CAND) {
      const float a = vals[(long)(2 * t) * n / TOPK_SAMPLE], b = vals[(long)(2 * t + 1) * n / TOPK_SAMPLE];
      smp0 = ord_key(a); smp1 = ord_key(b);
      smp0 = smp0 < 1u ? 1u : smp0; smp1 = smp1 < 1u ? 1u : smp1;
    }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      unsigned key = ord_key(f[k]);
      key = key < 1u ? 1u : key;
      v[k] = (t + k * TOPK_THREADS) < n ? key : 0u;   // padding: below every real key (real keys are >= 1)
    }
  }
  TOPK_STAMP(1);
  s_rank[t] = 0;
  s_slot[t] = 0;
  if (t == 0) s_cnt = 0;
  const int CAPW = min(TOPK_MAXK, K + max(K >> 3, 32));
  int ncand = n;
  bool registers = false;                             // block-uniform: the 16-wave form over the registers decides
  if (n <= TOPK_CAND) {
#pragma unroll
    for (int k = 0; k < (MAXV < 4 ? MAXV : 4); ++k) { s_ckey[t + k * TOPK_THREADS] = v[k]; s_cidx[t + k * TOPK_THREADS] = (unsigned)(t + k * TOPK_THREADS); }
    if (MAXV < 4) for (int k = MAXV; k < 4; ++k) s_ckey[t + k * TOPK_THREADS] = 0u;
    __syncthreads();
  } else {
    unsigned* s_smp = s_cidx;
    s_smp[t] = smp0;
    s_smp[TOPK_THREADS + t] = smp1;
    *reinterpret_cast<uint4*>(s_ckey + t * 4) = make_uint4(0u, 0u, 0u, 0u);     // keys past the candidates read as padding
    __syncthreads();
    // sample ranks [r_lo, 1.5 r_lo] <-> about [1.3 K + 10 n / S, 2 K + 15 n / S] keys of the input (S = TOPK_SAMPLE entries)
    if (wv == 0) {
      const int r_lo = (int)(1.3f * (float)K * (float)TOPK_SAMPLE / (float)n) + 10, r_hi = r_lo + (r_lo >> 1);
      unsigned long long lo, hi;
      int cnt_hi;
      bool win;
      wave_search(s_smp, TOPK_SAMPLE / 1024, r_lo, r_hi, lane, lo, hi, cnt_hi, win);
      if (lane == 0) s_ctl[0] = (unsigned)lo;
    }
    __syncthreads();
    const unsigned T = s_ctl[0];
    {
      int wc = 0;
#pragma unroll
      for (int k = 0; k < MAXV; ++k) wc += __popcll(__ballot(v[k] >= T));
      int base = 0;
      if (lane == 0 && wc) base = atomicAdd(&s_cnt, wc);
      int run = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const bool tk = v[k] >= T;
        const unsigned long long m = __ballot(tk);
        if (m == 0ull) continue;
        if (tk) {
          const int pos = run + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (pos < TOPK_CAND) { s_ckey[pos] = v[k]; s_cidx[pos] = (unsigned)(t + k * TOPK_THREADS); }
        }
        run += __popcll(m);
      }
    }
    __syncthreads();
    ncand = s_cnt;
    registers = ncand < K || ncand > TOPK_CAND;
    __syncthreads();
    if (t == 0) s_cnt = 0;
  }
  TOPK_STAMP(2);
  TOPK_NOTE(7, ncand + (registers ? (1 << 30) : 0));
  int c;
  if (registers) {
    __syncthreads();                                  // s_cnt = 0
    c = topk_select_registers<MAXV>(v, n, K, CAPW, sel, &s_cnt);
  } else {
    const int nk256 = (ncand + 255) >> 8, nk1024 = (ncand + 1023) >> 10;     // keys past the candidates are 0 up to TOPK_CAND
    if (wv == 0) {
      // invariant: count(key >= lo) >= K > count(key >= hi) = cnt_hi
      unsigned long long lo, hi;
      int cnt_hi;
      bool win;
      wave_search(s_ckey, nk1024, K, CAPW, lane, lo, hi, cnt_hi, win);
      const unsigned kth = (unsigned)lo;
      unsigned idx_cut = 0xffffffffu;                 // among the keys == kth, keep indices <= idx_cut
      if (!win) {
        const int n_ge = wave_count_ge(s_ckey, nk1024, kth, lane);
        if (n_ge > K) {                               // ties at the cut: keep the lowest indices (rare: one wave, binary)
          const int need = K - cnt_hi;                // cnt_hi = count(key > kth)
          long ilo = -1, ihi = (long)n - 1;           // count(idx <= ilo) < need <= count(idx <= ihi)
          while (ihi - ilo > 1) {
            const long mid = ilo + ((ihi - ilo) >> 1);
            const int cc = wave_count_eq_idx(s_ckey, s_cidx, nk256, kth, (unsigned)mid, lane);
            if (cc >= need) ihi = mid; else ilo = mid;
          }
          idx_cut = (unsigned)ihi;
        }
      }
      if (lane == 0) { s_ctl[0] = kth; s_ctl[1] = win ? 1u : 0u; s_ctl[2] = idx_cut; }
    }
    __syncthreads();                                  // also: s_cnt = 0
    const unsigned kth = s_ctl[0], idx_cut = s_ctl[2];
    const bool window = s_ctl[1] != 0u;
    // the kept candidates -> sel, order arbitrary (<= 4 candidates per thread)
    unsigned ck[4], ci[4];
    bool tk[4];
    int wc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p_ = t + j * TOPK_THREADS;
      ck[j] = s_ckey[p_]; ci[j] = s_cidx[p_];
      tk[j] = p_ < ncand && (ck[j] > kth || (ck[j] == kth && (window || ci[j] <= idx_cut)));
      wc += __popcll(__ballot(tk[j]));
    }
    int base = 0;
    if (lane == 0 && wc) base = atomicAdd(&s_cnt, wc);
    int run = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long m = __ballot(tk[j]);
      if (tk[j]) {
        const int pos = run + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (pos < TOPK_MAXK) sel[pos] = ((unsigned long long)ck[j] << 32) | (unsigned long long)(0xffffffffu - ci[j]);
      }
      run += __popcll(m);
    }
    __syncthreads();
    c = min(s_cnt, TOPK_MAXK);
  }
  TOPK_STAMP(3);
  TOPK_NOTE(6, c);
  topk_rank_place(sel, c, s_ckey, s_rank, s_slot);
  TOPK_STAMP(5);
}

template <int MAXV>
__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ vals, int n, int K, long* __restrict__ idx_out,
                                                            float* __restrict__ val_out) {
  __shared__ unsigned long long sel[TOPK_MAXK];
  block_topk_sorted<MAXV>(vals, n, K, sel);
  for (int i = threadIdx.x; i < K; i += TOPK_THREADS) {
    const unsigned long long e = sel[i];
    idx_out[i] = (long)(0xffffffffu - (unsigned)(e & 0xffffffffull));
    if (val_out) val_out[i] = ord_val((unsigned)(e >> 32));
  }
}

extern "C" int far3d_topk(const float* vals, int n, int K, int64_t* idx_out, float* val_out, void* stream) {
  FAR3D_CHECK_ARG(vals && idx_out, "far3d_topk: null pointer argument");
  FAR3D_CHECK_ARG(n > 0 && n <= TOPK_THREADS * TOPK_MAXV && K > 0 && K <= n && K <= TOPK_MAXK,
                  "far3d_topk: need 0 < K <= min(n, %d), n <= %d (got n=%d K=%d)", TOPK_MAXK, TOPK_THREADS * TOPK_MAXV, n, K);
  if (n <= TOPK_THREADS * 4)      // the memory update ranks A ~ 1.5k scores: 4 registers per thread instead of 40
    hipLaunchKernelGGL(topk_kernel<4>, dim3(1), dim3(TOPK_THREADS), 0, (hipStream_t)stream, vals, n, K, (long*)idx_out, val_out);
  else if (n <= TOPK_THREADS * 16)
    hipLaunchKernelGGL(topk_kernel<16>, dim3(1), dim3(TOPK_THREADS), 0, (hipStream_t)stream, vals, n, K, (long*)idx_out, val_out);
  else
    hipLaunchKernelGGL(topk_kernel<TOPK_MAXV>, dim3(1), dim3(TOPK_THREADS), 0, (hipStream_t)stream, vals, n, K, (long*)idx_out, val_out);
  FAR3D_CHECK_LAUNCH("far3d_topk");
  return FAR3D_OK;
}

// ------------------------------------------------------------------------------------------ NMS-free decode
struct DecodeParams {
  const float* cls;      // (A, ncls) logits of the last decoder layer
  const float* box;      // (A, code) normalised box codes of the last layer (cx,cy,cz,log w,log l,log h,sin,cos[,vx,vy])
  float* boxes;          // (K, code-1): cx,cy,cz - h/2,w,l,h,rot[,vx,vy]
  float* scores;         // (K)
  long* labels;          // (K)
  unsigned char* keep;   // (K) 1 = centre inside post_center_range
  int A, ncls, code, K;
  float lo[3], hi[3];
};

// cand_idx: null = the K winners' positions are indices into cls (A * ncls <= TOPK_THREADS * TOPK_MAXV, one launch);
// otherwise cls points at candidate logits (the per-chunk winners of decode_chunk_kernel, chunk-major, each chunk sorted) and
// cand_idx maps a position to the original index.  Equal logits keep their original index order in that layout, so the tie rule
// (lowest index first) is the single-launch one.
template <int MAXV>
__global__ __launch_bounds__(TOPK_THREADS) void decode_topk_kernel(DecodeParams p, int n, const int* __restrict__ cand_idx) {
  __shared__ unsigned long long sel[TOPK_MAXK];
  // sigmoid is monotone: the top-K of sigmoid(logits) is the top-K of the logits
  block_topk_sorted<MAXV>(p.cls, n, p.K, sel);
  for (int i = threadIdx.x; i < p.K; i += TOPK_THREADS) {
    const unsigned long long e = sel[i];
    int idx = (int)(0xffffffffu - (unsigned)(e & 0xffffffffull));
    if (cand_idx) idx = cand_idx[idx];
    const float logit = ord_val((unsigned)(e >> 32));
    const int q = idx / p.ncls;
    p.labels[i] = (long)(idx - q * p.ncls);
    p.scores[i] = 1.f / (1.f + expf(-logit));
    const float* b = p.box + (long)q * p.code;
    float* o = p.boxes + (long)i * (p.code - 1);
    const float cx = b[0], cy = b[1], cz = b[2];
    const float w = expf(b[3]), l = expf(b[4]), h = expf(b[5]);
    o[0] = cx; o[1] = cy; o[2] = cz - h * 0.5f;
    o[3] = w; o[4] = l; o[5] = h;
    o[6] = atan2f(b[6], b[7]);
    for (int k = 8; k < p.code; ++k) o[k - 1] = b[k];
    p.keep[i] = (cx >= p.lo[0] && cy >= p.lo[1] && cz >= p.lo[2] && cx <= p.hi[0] && cy <= p.hi[1] && cz <= p.hi[2]) ? 1 : 0;
  }
}

// far3d_decode_topk_mem: the decode and the memory update's top-k (ref farhead.py:488-491) are both single-workgroup rankings of the last
// layer's outputs and do not depend on each other: workgroup 0 decodes, workgroup 1 ranks the max-class scores -- one launch, the two
// latency chains side by side on two CUs (in the frame each such launch costs ~8 us of cold start on top of its own time).  Same code
// paths as the two stand-alone kernels: identical results.
template <int MAXV>
__global__ __launch_bounds__(TOPK_THREADS) void decode_topk_mem_kernel(DecodeParams p, int n, const float* __restrict__ vals2, int n2, int K2,
                                                                       long* __restrict__ idx2) {
  __shared__ unsigned long long sel[TOPK_MAXK];
  if (blockIdx.x == 1) {
    block_topk_sorted<4>(vals2, n2, K2, sel);
    for (int i = threadIdx.x; i < K2; i += TOPK_THREADS) idx2[i] = (long)(0xffffffffu - (unsigned)(sel[i] & 0xffffffffull));
    return;
  }
  block_topk_sorted<MAXV>(p.cls, n, p.K, sel);
  for (int i = threadIdx.x; i < p.K; i += TOPK_THREADS) {
    const unsigned long long e = sel[i];
    const int idx = (int)(0xffffffffu - (unsigned)(e & 0xffffffffull));
    const float logit = ord_val((unsigned)(e >> 32));
    const int q = idx / p.ncls;
    p.labels[i] = (long)(idx - q * p.ncls);
    p.scores[i] = 1.f / (1.f + expf(-logit));
    const float* b = p.box + (long)q * p.code;
    float* o = p.boxes + (long)i * (p.code - 1);
    const float cx = b[0], cy = b[1], cz = b[2];
    const float w = expf(b[3]), l = expf(b[4]), h = expf(b[5]);
    o[0] = cx; o[1] = cy; o[2] = cz - h * 0.5f;
    o[3] = w; o[4] = l; o[5] = h;
    o[6] = atan2f(b[6], b[7]);
    for (int k = 8; k < p.code; ++k) o[k - 1] = b[k];
    p.keep[i] = (cx >= p.lo[0] && cy >= p.lo[1] && cz >= p.lo[2] && cx <= p.hi[0] && cy <= p.hi[1] && cz <= p.hi[2]) ? 1 : 0;
  }
}

// Large inputs (A * ncls > TOPK_THREADS * TOPK_MAXV: the reference's threshold proposal mode on a busy frame): workgroup c
// ranks chunk c of the logits and leaves its K best (value, original index) in the workspace; decode_topk_kernel then ranks
// the nchunks * K candidates.  Exact: the global top-K is a subset of the union of the per-chunk top-K.
#define DECODE_CHUNK (TOPK_THREADS * TOPK_MAXV)
__global__ __launch_bounds__(TOPK_THREADS) void decode_chunk_kernel(const float* __restrict__ cls, int n, int K, float* __restrict__ cand_val,
                                                                    int* __restrict__ cand_idx) {
  __shared__ unsigned long long sel[TOPK_MAXK];
  const int c = blockIdx.x, base = c * DECODE_CHUNK;
  const int cn = min(DECODE_CHUNK, n - base), Kc = min(K, cn);
  block_topk_sorted<TOPK_MAXV>(cls + base, cn, Kc, sel);
  for (int i = threadIdx.x; i < K; i += TOPK_THREADS) {
    float v = -INFINITY;      // a short last chunk pads with -inf pointing at its own first element (never beats a real candidate
    int idx = base;           // on value; on an all -inf tie the padding sits after the real entries of the same chunk)
    if (i < Kc) {
      const unsigned long long e = sel[i];
      v = ord_val((unsigned)(e >> 32));
      idx = base + (int)(0xffffffffu - (unsigned)(e & 0xffffffffull));
    }
    cand_val[c * K + i] = v;
    cand_idx[c * K + i] = idx;
  }
}

extern "C" int far3d_decode_topk(const float* cls_last, const float* box_last, int A, int num_classes, int code_size, int K,
                                 const float* post_center_range, float* boxes, float* scores, int64_t* labels,
                                 unsigned char* keep, void* workspace, long workspace_bytes, void* stream) {
  FAR3D_CHECK_ARG(cls_last && box_last && post_center_range && boxes && scores && labels && keep, "far3d_decode_topk: null pointer argument");
  const long n = (long)A * num_classes;
  FAR3D_CHECK_ARG(A > 0 && num_classes > 0 && code_size >= 8 && K > 0 && K <= TOPK_MAXK && n < (1L << 30) && K <= n,
                  "far3d_decode_topk: bad sizes A=%d ncls=%d code=%d K=%d (K <= %d)", A, num_classes, code_size, K, TOPK_MAXK);
  DecodeParams p;
  p.cls = cls_last; p.box = box_last; p.boxes = boxes; p.scores = scores; p.labels = (long*)labels; p.keep = keep;
  p.A = A; p.ncls = num_classes; p.code = code_size; p.K = K;
  for (int k = 0; k < 3; ++k) { p.lo[k] = post_center_range[k]; p.hi[k] = post_center_range[3 + k]; }
  hipStream_t st = (hipStream_t)stream;
  if (n <= TOPK_THREADS * 4) {
    hipLaunchKernelGGL(decode_topk_kernel<4>, dim3(1), dim3(TOPK_THREADS), 0, st, p, (int)n, (const int*)nullptr);
  } else if (n <= TOPK_THREADS * 16) {      // the benchmarked frame: 1544 queries x 10 classes
    hipLaunchKernelGGL(decode_topk_kernel<16>, dim3(1), dim3(TOPK_THREADS), 0, st, p, (int)n, (const int*)nullptr);
  } else if (n <= DECODE_CHUNK) {
    hipLaunchKernelGGL(decode_topk_kernel<TOPK_MAXV>, dim3(1), dim3(TOPK_THREADS), 0, st, p, (int)n, (const int*)nullptr);
  } else {
    const int nchunks = (int)((n + DECODE_CHUNK - 1) / DECODE_CHUNK);
    const long ncand = (long)nchunks * K;
    FAR3D_CHECK_ARG(ncand <= DECODE_CHUNK, "far3d_decode_topk: A*num_classes=%ld needs %d chunks x K=%d candidates > %d", n, nchunks, K, DECODE_CHUNK);
    FAR3D_CHECK_ARG(workspace && workspace_bytes >= ncand * 8 && ((uintptr_t)workspace % 8) == 0,
                    "far3d_decode_topk: A*num_classes=%ld > %d needs a workspace of FAR3D_DECODE_WS_BYTES = %ld bytes (got %ld)", n, DECODE_CHUNK,
                    ncand * 8, workspace ? workspace_bytes : 0L);
    float* cand_val = (float*)workspace;
    int* cand_idx = (int*)(cand_val + ncand);
    hipLaunchKernelGGL(decode_chunk_kernel, dim3(nchunks), dim3(TOPK_THREADS), 0, st, cls_last, (int)n, K, cand_val, cand_idx);
    p.cls = cand_val;
    if (ncand <= TOPK_THREADS * 4)
      hipLaunchKernelGGL(decode_topk_kernel<4>, dim3(1), dim3(TOPK_THREADS), 0, st, p, (int)ncand, (const int*)cand_idx);
    else
      hipLaunchKernelGGL(decode_topk_kernel<TOPK_MAXV>, dim3(1), dim3(TOPK_THREADS), 0, st, p, (int)ncand, (const int*)cand_idx);
  }
  FAR3D_CHECK_LAUNCH("far3d_decode_topk");
  return FAR3D_OK;
}

extern "C" int far3d_decode_topk_mem(const float* cls_last, const float* box_last, int A, int num_classes, int code_size, int K,
                                     const float* post_center_range, float* boxes, float* scores, int64_t* labels,
                                     unsigned char* keep, void* workspace, long workspace_bytes, const float* mem_scores, int mem_n,
                                     int mem_K, int64_t* mem_idx_out, void* stream) {
  FAR3D_CHECK_ARG(mem_scores && mem_idx_out, "far3d_decode_topk_mem: null pointer argument");
  FAR3D_CHECK_ARG(mem_n > 0 && mem_n <= TOPK_THREADS * TOPK_MAXV && mem_K > 0 && mem_K <= mem_n && mem_K <= TOPK_MAXK,
                  "far3d_decode_topk_mem: need 0 < mem_K <= min(mem_n, %d), mem_n <= %d (got n=%d K=%d)", TOPK_MAXK, TOPK_THREADS * TOPK_MAXV, mem_n, mem_K);
  const long n = (long)A * num_classes;
  if (n > DECODE_CHUNK || n <= 0 || mem_n > TOPK_THREADS * 4) {      // shapes outside the fused launch: the two stand-alone calls
    if (const int rc = far3d_topk(mem_scores, mem_n, mem_K, mem_idx_out, nullptr, stream)) return rc;
    return far3d_decode_topk(cls_last, box_last, A, num_classes, code_size, K, post_center_range, boxes, scores, labels, keep, workspace,
                             workspace_bytes, stream);
  }
  FAR3D_CHECK_ARG(cls_last && box_last && post_center_range && boxes && scores && labels && keep, "far3d_decode_topk_mem: null pointer argument");
  FAR3D_CHECK_ARG(A > 0 && num_classes > 0 && code_size >= 8 && K > 0 && K <= TOPK_MAXK && K <= n,
                  "far3d_decode_topk_mem: bad sizes A=%d ncls=%d code=%d K=%d (K <= %d)", A, num_classes, code_size, K, TOPK_MAXK);
  DecodeParams p;
  p.cls = cls_last; p.box = box_last; p.boxes = boxes; p.scores = scores; p.labels = (long*)labels; p.keep = keep;
  p.A = A; p.ncls = num_classes; p.code = code_size; p.K = K;
  for (int k = 0; k < 3; ++k) { p.lo[k] = post_center_range[k]; p.hi[k] = post_center_range[3 + k]; }
  hipStream_t st = (hipStream_t)stream;
  if (n <= TOPK_THREADS * 4)
    hipLaunchKernelGGL(decode_topk_mem_kernel<4>, dim3(2), dim3(TOPK_THREADS), 0, st, p, (int)n, mem_scores, mem_n, mem_K, (long*)mem_idx_out);
  else if (n <= TOPK_THREADS * 16)
    hipLaunchKernelGGL(decode_topk_mem_kernel<16>, dim3(2), dim3(TOPK_THREADS), 0, st, p, (int)n, mem_scores, mem_n, mem_K, (long*)mem_idx_out);
  else
    hipLaunchKernelGGL(decode_topk_mem_kernel<TOPK_MAXV>, dim3(2), dim3(TOPK_THREADS), 0, st, p, (int)n, mem_scores, mem_n, mem_K, (long*)mem_idx_out);
  FAR3D_CHECK_LAUNCH("far3d_decode_topk_mem");
  return FAR3D_OK;
}

// ------------------------------------------------------------------------------------------ per-frame camera calibration
// one thread per camera: 4x4 inverse by Gauss-Jordan with partial pivoting in f64 (rounded once to f32)
__global__ __launch_bounds__(64) void camera_prep_kernel(const float* __restrict__ l2i, const float* __restrict__ intr,
                                                         const float* __restrict__ extr, float* __restrict__ i2l,
                                                         float* __restrict__ c14, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* m = l2i + n * 16;
  if (i2l) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { a[i][j] = (double)m[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
      int piv = c;
      double best = fabs(a[c][c]);
      for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
      if (piv != c) for (int j = 0; j < 8; ++j) { const double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
      const double inv = 1.0 / a[c][c];          // singular input -> inf/nan, like torch.linalg.inv's garbage-in behaviour
      for (int j = 0; j < 8; ++j) a[c][j] *= inv;
      for (int r = 0; r < 4; ++r) {
        if (r == c) continue;
        const double f = a[r][c];
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
      }
    }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) i2l[n * 16 + i * 4 + j] = (float)a[i][4 + j];
  }
  if (c14) {            // [fx/1e3, fy/1e3, extrinsics[:3,:4] flat]  (farhead.py:553-556)
    c14[n * 14 + 0] = intr[n * 16 + 0] / 1e3f;
    c14[n * 14 + 1] = intr[n * 16 + 5] / 1e3f;
    for (int k = 0; k < 12; ++k) c14[n * 14 + 2 + k] = extr[n * 16 + k];
  }
}

extern "C" int far3d_camera_prep(const float* lidar2img, const float* intrinsics, const float* extrinsics, float* img2lidar,
                                 float* c14, int N, void* stream) {
  FAR3D_CHECK_ARG(lidar2img && N > 0 && (!c14 || (intrinsics && extrinsics)), "far3d_camera_prep: bad arguments");
  hipLaunchKernelGGL(camera_prep_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, lidar2img, intrinsics, extrinsics,
                     img2lidar, c14, N);
  FAR3D_CHECK_LAUNCH("far3d_camera_prep");
  return FAR3D_OK;
}

// ------------------------------------------------------------------------------------------ aggregation query order
// key[a] = (camera whose image centre the reference point projects closest to) * 64 + 8x8 image cell; perm = queries sorted by
// key (counting sort in LDS; the order inside one key is arbitrary -- perm only decides which workgroup handles which query).
#define AGGO_MAX_BINS 1024
__global__ __launch_bounds__(1024) void agg_order_kernel(const float* __restrict__ ref, const float* __restrict__ l2i,
                                                         int* __restrict__ perm, int A, int N, float lo0, float lo1, float lo2,
                                                         float sp0, float sp1, float sp2, float pad_h, float pad_w,
                                                         const int* __restrict__ hole_count, int hole_start, int hole_end, int row_base,
                                                         const float* __restrict__ Vc, float* __restrict__ tables, int J, int split_extra,
                                                         int* __restrict__ inv, float4* __restrict__ qbase) {
  __shared__ int hist[AGGO_MAX_BINS];
  __shared__ int wsum[16];
  __shared__ int hsum[16];
  if (blockIdx.x > 0) {      // blocks 1.. : the aggregation softmax factors of decoder layer blockIdx.x - 1 (csrc/agg_tables.hpp)
    agg_tables_body(Vc + (long)(blockIdx.x - 1) * N * J, tables + (long)(blockIdx.x - 1) * (2 + N) * J, N, J, threadIdx.x, 1024);
    return;
  }
  const int t = threadIdx.x, nb = N * 64;
  // query rows [hole_lo, hole_end) hold no query (fixed-capacity proposal mode): they still get a slot (every row of `out` is
  // written) but their entry is ~a (negative), which far3d_aggregate_forward answers with a zero row and no work
  const int hole_lo = hole_count ? hole_start + min(max(*hole_count, 0), hole_end - hole_start) : hole_end;
  for (int i = t; i < AGGO_MAX_BINS; i += 1024) hist[i] = 0;
  __syncthreads();
  constexpr int MAXQ = 8;   // A <= 8192
  int key[MAXQ], rank[MAXQ], cam2[MAXQ];      // cam2: the camera the reference point projects second closest to (sorted mode's hint)
  unsigned heavy = 0u;      // bit k: query t + 1024 k projects into two or more cameras (twice the items: far3d_aggregate_forward variant 9)
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int a = t + k * 1024;
    key[k] = -1;
    if (a < A) {
      const float* rp = ref + (long)(row_base + a) * 3;      // rows [row_base, row_base + A) of the caller's reference points
      const float X = agg_base_metre(rp[0], sp0, lo0), Y = agg_base_metre(rp[1], sp1, lo1), Z = agg_base_metre(rp[2], sp2, lo2);
      float best = 3.0e9f, best2 = 3.0e9f, bu = 0.f, bv = 0.f;
      int cam = 0, camb = 0, nvis = 0;
      for (int n = 0; n < N; ++n) {
        const float* m = l2i + n * 16;
        const float x = m[0] * X + m[1] * Y + m[2] * Z + m[3], y = m[4] * X + m[5] * Y + m[6] * Z + m[7],
                    z = m[8] * X + m[9] * Y + m[10] * Z + m[11];
        const float zc = fmaxf(z, 1e-5f);
        const float u = x / zc / pad_w - 0.5f, v = y / zc / pad_h - 0.5f;
        const float cost = z > 1e-5f ? u * u + v * v : 1.0e9f;
        if (cost < best) { best2 = best; camb = cam; best = cost; cam = n; bu = u; bv = v; }
        else if (cost < best2) { best2 = cost; camb = n; }
        nvis += (z > 1e-5f && fabsf(u) < 0.55f && fabsf(v) < 0.55f) ? 1 : 0;      // inside the image, 5 % margin for the key-point offsets
      }
      const int arow = row_base + a;
      if (nvis >= 2 && !(arow >= hole_lo && arow < hole_end)) heavy |= 1u << k;
      const int ub = (int)(fminf(fmaxf(bu + 0.5f, 0.f), 0.999f) * 8.f), vb = (int)(fminf(fmaxf(bv + 0.5f, 0.f), 0.999f) * 8.f);
      key[k] = (cam * 8 + vb) * 8 + ub;
      cam2[k] = camb == cam ? (cam + 1 < N ? cam + 1 : 0) : camb;
      rank[k] = atomicAdd(&hist[key[k]], 1);
    }
  }
  __syncthreads();
  // exclusive prefix over the nb <= 1024 bins: thread t owns bin t
  int c = t < nb ? hist[t] : 0, inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o);
    if ((t & 63) >= o) inc += up;
  }
  if ((t & 63) == 63) wsum[t >> 6] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
  __syncthreads();
  if (t < AGGO_MAX_BINS) hist[t] = base + inc - c;
  __syncthreads();
  // sibling entries (variant 9): the first split_extra heavy queries IN ROW ORDER are split -- a deterministic choice (an atomic counter would
  // pick a different subset from run to run when there are more heavy queries than slots, and a split query sums in another order) --
  // query with heavy-rank e gets sibling slot e: perm[A + e] = row | sibling | split, and its main entry carries the split flag
  unsigned splitm = 0u;
  if (split_extra > 0) {
    int before = 0;                                          // heavy queries in the chunks already scanned
    for (int k = 0; k < MAXQ; ++k) {
      if (k * 1024 >= A) break;                              // block-uniform
      const bool h = (heavy >> k) & 1u;
      const unsigned long long bm = __ballot(h);
      const int inwave = __popcll(bm & ((1ull << (t & 63)) - 1ull));
      if ((t & 63) == 0) hsum[t >> 6] = __popcll(bm);
      __syncthreads();
      int wb = 0, tot = 0;
      for (int w = 0; w < 16; ++w) { wb += w < (t >> 6) ? hsum[w] : 0; tot += hsum[w]; }
      __syncthreads();
      const int e = before + wb + inwave;
      if (h && e < split_extra) {
        splitm |= 1u << k;
        perm[A + e] = (row_base + t + k * 1024) | (1 << 30) | (1 << 29);
      }
      before += tot;
    }
    for (int e = before + t; e < split_extra; e += 1024) perm[A + e] = 0x7fffffff;      // unused sibling slots
  }
#pragma unroll
  for (int k = 0; k < MAXQ; ++k)
    if (key[k] >= 0) {
      const int a = row_base + t + k * 1024;                 // absolute row
      perm[hist[key[k]] + rank[k]] = (a >= hole_lo && a < hole_end) ? ~a : (((splitm >> k) & 1u) ? (a | (1 << 29)) : a);
    }
  // launch-order operands of far3d_aggregate_forward's sorted mode: inv[i] = the slot (workgroup) of row row_base + i -- the producers
  // of the per-layer logits / offsets store row i at row inv[i] -- and qbase[slot] = (the reference point in metres, the camera hint:
  // cam0 | cam1 << 8, the two cameras it projects closest to), so that the aggregation kernel's first loads depend on its block index only
  if (inv || qbase) {
#pragma unroll
    for (int k = 0; k < MAXQ; ++k)
      if (key[k] >= 0) {
        const int i = t + k * 1024, slot = hist[key[k]] + rank[k];
        if (inv) inv[i] = slot;
        if (qbase) {
          const float* rp = ref + (long)(row_base + i) * 3;
          qbase[slot] = make_float4(agg_base_metre(rp[0], sp0, lo0), agg_base_metre(rp[1], sp1, lo1), agg_base_metre(rp[2], sp2, lo2),
                                    __int_as_float((key[k] >> 6) | (cam2[k] << 8)));
        }
      }
  }
}

extern "C" int far3d_agg_order(const float* ref, const float* lidar2img, int32_t* perm, int A, int N, const float* pc_range,
                               float pad_h, float pad_w, const int32_t* hole_count, int hole_start, int hole_end, int row_base,
                               const float* Vc, float* tables, int layers, int J, int split_extra, int32_t* inv, float* qbase,
                               void* stream) {
  FAR3D_CHECK_ARG(ref && lidar2img && perm && pc_range && A >= 0 && N > 0, "far3d_agg_order: bad arguments");
  FAR3D_CHECK_ARG(!qbase || ((uintptr_t)qbase % 16) == 0, "far3d_agg_order: qbase needs 16-byte alignment");
  FAR3D_CHECK_ARG(!tables || (Vc && layers > 0 && J > 0), "far3d_agg_order: tables need Vc, layers > 0 and J > 0");
  if (!tables) layers = 0;
  FAR3D_CHECK_ARG(A <= 8192 && N * 64 <= AGGO_MAX_BINS, "far3d_agg_order: A=%d (<= 8192) or N=%d (<= 16) too large", A, N);
  FAR3D_CHECK_ARG(row_base >= 0 && (!hole_count || (0 <= hole_start && hole_start <= hole_end)), "far3d_agg_order: bad hole [%d, %d) / row_base %d",
                  hole_start, hole_end, row_base);
  if (!hole_count) hole_start = hole_end = 0;
  FAR3D_CHECK_ARG(split_extra >= 0 && row_base + A < (1 << 29), "far3d_agg_order: bad split_extra %d / row range", split_extra);
  if (A == 0 && layers == 0 && split_extra == 0) return FAR3D_OK;
  hipLaunchKernelGGL(agg_order_kernel, dim3(1 + layers), dim3(1024), 0, (hipStream_t)stream, ref, lidar2img, perm, A, N, pc_range[0],
                     pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2], pad_h,
                     pad_w, (const int*)hole_count, hole_start, hole_end, row_base, Vc, tables, J, split_extra, (int*)inv,
                     reinterpret_cast<float4*>(qbase));
  FAR3D_CHECK_LAUNCH("far3d_agg_order");
  return FAR3D_OK;
}

// ------------------------------------------------------------------------------------------ nan_to_num
// torch.nan_to_num defaults: nan -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX; in place, optional bf16 copy for the next GEMM.
__global__ __launch_bounds__(256) void nan_to_num_kernel(float* __restrict__ x, bf16_t* __restrict__ xb, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<float4*>(x)[i];
    float* e = reinterpret_cast<float*>(&v);
    bool dirty = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = e[k];
      if (f != f) { e[k] = 0.f; dirty = true; }
      else if (f == INFINITY) { e[k] = 3.4028234663852886e38f; dirty = true; }
      else if (f == -INFINITY) { e[k] = -3.4028234663852886e38f; dirty = true; }
    }
    if (dirty) reinterpret_cast<float4*>(x)[i] = v;
    if (xb) reinterpret_cast<uint2*>(xb)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}

extern "C" int far3d_nan_to_num(float* x, void* bf16_copy, long n, void* stream) {
  FAR3D_CHECK_ARG(x && n >= 0 && (n % 4) == 0, "far3d_nan_to_num: bad arguments (n must be a multiple of 4)");
  if (n == 0) return FAR3D_OK;
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)bf16_copy, n4);
  FAR3D_CHECK_LAUNCH("far3d_nan_to_num");
  return FAR3D_OK;
}
