// Per-layer, per-frame factors of the aggregation softmax (consumed by aggregate_v8_kernel, csrc/sampling.hip).
// logit[n][j] = U[a][j] + V[n][j]  =>  exp(logit - m) = exp(U + mV - m') * exp(V[n] - mV) with mV[j] = max_n V[n][j]:
//   tab = [ mV (J) | EV (J) | eV[0..N) (N x J) ],   eV[n][j] = exp(V[n][j] - mV[j]) in (0, 1],   EV[j] = sum_n eV[n][j] >= 1.
// One layer per call; the calling block's threads stride over j.  Shared by far3d_agg_tables (its own launch) and
// far3d_agg_order (extra blocks of the per-frame ordering launch).
#pragma once
#include "common.hpp"

__device__ __forceinline__ void agg_tables_body(const float* __restrict__ v, float* __restrict__ o, int N, int J, int tid, int nthreads) {
  for (int j = tid; j < J; j += nthreads) {
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, v[n * J + j]);
    float s = 0.f;
    for (int n = 0; n < N; ++n) {
      const float e = __builtin_amdgcn_exp2f((v[n * J + j] - m) * 1.4426950408889634f);
      o[(2 + n) * J + j] = e;
      s += e;
    }
    o[j] = m;
    o[J + j] = s;
  }
}

// A query's reference point in metres (ref models/utils/detr3d_transformer.py:524-525): far3d_agg_order evaluates it once per frame for
// every slot of the aggregation kernel's sorted mode (`qbase`), the kernel's unsorted form evaluates it itself -- ONE fma, shared, so
// that both forms give the same bits.
__device__ __forceinline__ float agg_base_metre(float r, float span, float lo) { return __builtin_fmaf(r, span, lo); }
