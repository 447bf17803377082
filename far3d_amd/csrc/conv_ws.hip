// Instantiations of the persistent wave-specialised 3x3 convolution (conv_ws.hpp).  Tile ids of far3d_conv2d_nhwc:
//   400-419  pair-stored activations, split products (x_dt = FAR3D_DT_BF16_PAIR): the in-tolerance engine's backbone / FPN / 2D head
//   420-439  plain bf16
//   440-459  pair-stored, step hand-over through LDS counters instead of a barrier
// Epilogue: bias + activation + pair / bf16 store only (no residual, no second output, no channel sums): far3d_conv2d_nhwc refuses
// the tile for a layer that needs more.
#include "conv_ws.hpp"

#ifdef FAR3D_PROFILING
std::atomic<int> g_ws_ablate{0};
extern "C" int far3d_conv_ws_set_ablate(int mask) { g_ws_ablate.store(mask); return 0; }     // probes only (see conv_ws.hpp)
#endif

int far3d_conv_ws_launch(const IgemmParams& P, int tile, hipStream_t st) {
  switch (tile) {
    // (consumer grid WGM x WGN, tiles per consumer WM x WN, producers, pair, double-buffered fragments)
    case 400: return launch_conv3x3_ws<2, 4, 2, 2, 4, true, true>(P, st);    // 128 ch x 8 rows: 8 consumers of 64 ch x 2 rows + 4 producers
    case 401: return launch_conv3x3_ws<1, 8, 2, 1, 4, true, true>(P, st);    // 64 ch x 8 rows: 8 consumers of 64 ch x 1 row (Cout 64)
    case 402: return launch_conv3x3_ws<1, 8, 5, 1, 4, true, false>(P, st);   // 160 ch x 8 rows: 8 consumers of 160 ch x 1 row (stage 3)
    case 403: return launch_conv3x3_ws<2, 4, 3, 1, 4, true, true>(P, st);    // 192 ch x 4 rows: 8 consumers of 96 ch x 1 row (stage 4)
    case 404: return launch_conv3x3_ws<2, 2, 1, 2, 2, true, true>(P, st);    // 64 ch x 4 rows: 4 consumers of 32 ch x 2 rows + 2 producers (2 per CU)
    case 405: return launch_conv3x3_ws<2, 4, 2, 1, 4, true, true>(P, st);    // 128 ch x 4 rows: 8 consumers of 64 ch x 1 row
    case 406: return launch_conv3x3_ws<2, 4, 1, 2, 4, true, true>(P, st);    // 64 ch x 8 rows: 8 consumers of 32 ch x 2 rows
    case 407: return launch_conv3x3_ws<1, 4, 5, 1, 2, true, false>(P, st);   // 160 ch x 4 rows: 4 consumers of 160 ch x 1 row + 2 producers
    case 408: return launch_conv3x3_ws<1, 4, 3, 1, 2, true, true>(P, st);    // 96 ch x 4 rows: 4 consumers + 2 producers (2 per CU)
    case 409: return launch_conv3x3_ws<2, 4, 2, 2, 2, true, true>(P, st);    // 400 with 2 producers
    // deeper weight rings (the producers run NSW - 1 steps ahead)
    case 410: return launch_conv3x3_ws<2, 4, 2, 2, 4, true, true, 4>(P, st);   // 400 with 4 stages (154 KB)
    case 411: return launch_conv3x3_ws<2, 4, 2, 1, 4, true, true, 6>(P, st);   // 405 (128 ch x 4 rows) with 6 stages
    case 412: return launch_conv3x3_ws<2, 4, 1, 2, 4, true, true, 8>(P, st);   // 406 (64 ch x 8 rows) with 8 stages
    case 413: return launch_conv3x3_ws<1, 8, 2, 1, 4, true, true, 8>(P, st);   // 401 (64 ch x 8 rows, Cout 64) with 8 stages
    case 414: return launch_conv3x3_ws<1, 4, 5, 1, 2, true, false, 5>(P, st);  // 407 (160 ch x 4 rows) with 5 stages
    case 415: return launch_conv3x3_ws<2, 2, 1, 2, 2, true, true, 6>(P, st);   // 404 (64 ch x 4 rows, 2 per CU) with 6 stages
    case 416: return launch_conv3x3_ws<2, 4, 3, 1, 4, true, true, 4>(P, st);   // 403 (192 ch x 4 rows) with 4 stages
    case 417: return launch_conv3x3_ws<2, 4, 2, 1, 4, true, true, 4>(P, st);   // 405 with 4 stages
    case 418: return launch_conv3x3_ws<4, 2, 2, 2, 4, true, true, 3>(P, st);   // 256 ch x 4 rows: 8 consumers of 64 ch x 2 rows (148 KB)
    case 419: return launch_conv3x3_ws<4, 2, 2, 1, 4, true, true, 3>(P, st);   // 256 ch x 2 rows: 8 consumers of 64 ch x 1 row
    // hand-over through LDS counters instead of a workgroup barrier per step (FLAGS): consumer waves run free of each other.  Measured
    // SLOWER than the barrier form (profiles/r6/ws_ab_pair.txt: s2.c1 281 us against 212, s4.c1 59 against 38): the polls cost more than
    // the lockstep they remove.  Kept as tested tiles for the record.
    case 440: return launch_conv3x3_ws<2, 4, 2, 2, 4, true, true, 3, true>(P, st);   // 400
    case 444: return launch_conv3x3_ws<2, 2, 1, 2, 2, true, true, 3, true>(P, st);   // 404 (64 ch x 4 rows, 2 per CU)
    case 445: return launch_conv3x3_ws<2, 4, 2, 1, 4, true, true, 6, true>(P, st);   // 411 (128 ch x 4 rows, 6 stages)
    // one barrier per KERNEL ROW (3 taps) instead of per tap, ring of 2 or 3 rows (GRP = 3)
    case 450: return launch_conv3x3_ws<2, 4, 2, 1, 4, true, true, 6, false, 3>(P, st);   // 128 ch x 4 rows, ring of 2 rows (149 KB)
    case 451: return launch_conv3x3_ws<2, 4, 1, 2, 4, true, true, 6, false, 3>(P, st);   // 64 ch x 8 rows, ring of 2 rows
    case 452: return launch_conv3x3_ws<2, 4, 1, 2, 4, true, true, 9, false, 3>(P, st);   // 64 ch x 8 rows, ring of 3 rows (160 KB)
    case 453: return launch_conv3x3_ws<1, 8, 2, 1, 4, true, true, 6, false, 3>(P, st);   // 64 ch x 8 rows (Cout 64), ring of 2 rows
    case 454: return launch_conv3x3_ws<1, 8, 2, 1, 4, true, true, 9, false, 3>(P, st);   // 64 ch x 8 rows (Cout 64), ring of 3 rows
    case 455: return launch_conv3x3_ws<1, 7, 2, 1, 4, true, true, 9, false, 3>(P, st);   // 64 ch x 7 rows: stage 4's 40 rows = 6 x 7 -> 252 items on 256 CUs
    case 456: return launch_conv3x3_ws<2, 7, 1, 1, 2, true, true, 9, false, 3>(P, st);   // 64 ch x 7 rows, 14 consumers of 32 ch x 1 row + 2 producers
    case 457: return launch_conv3x3_ws<2, 2, 1, 2, 2, true, true, 6, false, 3>(P, st);   // 64 ch x 4 rows, 4 consumers + 2 producers
    case 458: return launch_conv3x3_ws<1, 4, 3, 1, 2, true, true, 6, false, 3>(P, st);   // 96 ch x 4 rows, 4 consumers + 2 producers
    case 459: return launch_conv3x3_ws<2, 4, 1, 1, 4, true, true, 9, false, 3>(P, st);   // 64 ch x 4 rows, 8 consumers of 32 ch x 1 row, ring of 3 rows
    case 420: return launch_conv3x3_ws<2, 4, 2, 2, 4, false, true>(P, st);   // plain bf16: 128 ch x 8 rows
    case 421: return launch_conv3x3_ws<1, 8, 2, 1, 4, false, true>(P, st);   // plain bf16: 64 ch x 8 rows
    case 422: return launch_conv3x3_ws<1, 8, 5, 1, 4, false, true>(P, st);   // plain bf16: 160 ch x 8 rows
    case 423: return launch_conv3x3_ws<2, 4, 3, 1, 4, false, true>(P, st);   // plain bf16: 192 ch x 4 rows
    default: break;
  }
  far3d_set_error("far3d_conv2d_nhwc: unknown wave-specialised tile %d", tile);
  return FAR3D_ERR_ARG;
}

// The wave-specialised persistent 1x1 GEMM on pair-stored maps (gemm1x1_ws_kernel), tile ids 460-477:
// (consumer grid WGM x WGN, tiles per consumer WM x WN, producers, ring stages, steps per hand-over)
int far3d_gemm_ws_launch(const IgemmParams& P, int tile, hipStream_t st) {
  switch (tile) {
    case 460: return launch_gemm1x1_ws<4, 2, 1, 2, 4, 4, 1>(P, st);   // 128 x 128: 8 consumers of 32 ch x 64 px + 4 producers, 4 stages (128 KB)
    case 461: return launch_gemm1x1_ws<4, 2, 1, 2, 4, 4, 2>(P, st);   // 460 with a hand-over every 2 steps
    case 462: return launch_gemm1x1_ws<2, 4, 2, 1, 4, 4, 2>(P, st);   // 128 x 128: 8 consumers of 64 ch x 32 px
    case 463: return launch_gemm1x1_ws<2, 2, 2, 2, 4, 4, 2>(P, st);   // 128 x 128: 4 consumers of 64 x 64 + 4 producers
    case 464: return launch_gemm1x1_ws<4, 2, 2, 2, 4, 3, 1>(P, st);   // 256 x 128: 8 consumers of 64 x 64, 3 stages of 48 KB
    case 465: return launch_gemm1x1_ws<4, 2, 1, 2, 2, 4, 2>(P, st);   // 461 with 2 producers
    case 466: return launch_gemm1x1_ws<2, 2, 1, 2, 4, 6, 3>(P, st);   // 64 x 128: 4 consumers of 32 ch x 64 px, 6 stages of 24 KB, hand-over every 3
    case 467: return launch_gemm1x1_ws<2, 4, 1, 1, 4, 6, 2>(P, st);   // 64 x 128: 8 consumers of 32 x 32, 6 stages
    case 468: return launch_gemm1x1_ws<2, 2, 1, 2, 2, 3, 1>(P, st);   // 64 x 128: 4 consumers + 2 producers, 3 stages (72 KB: 2 per CU)
    case 469: return launch_gemm1x1_ws<4, 2, 1, 2, 4, 2, 1>(P, st);   // 460 with 2 stages (64 KB: 2 per CU)
    case 470: return launch_gemm1x1_ws<2, 4, 2, 2, 4, 3, 1>(P, st);   // 128 x 256: 8 consumers of 64 x 64, 3 stages of 48 KB
    case 471: return launch_gemm1x1_ws<2, 2, 4, 2, 4, 3, 1>(P, st);   // 256 x 128: 4 consumers of 128 ch x 64 px
    case 473: return launch_gemm1x1_ws<4, 2, 2, 2, 2, 3, 1>(P, st);   // 464 with 2 producers
    case 474: return launch_gemm1x1_ws<3, 2, 2, 2, 4, 3, 1>(P, st);   // 192 x 128: 6 consumers of 64 x 64, 3 stages of 40 KB
    case 475: return launch_gemm1x1_ws<2, 2, 2, 2, 4, 4, 1>(P, st);   // 463 with a hand-over per step
    case 476: return launch_gemm1x1_ws<2, 4, 2, 2, 2, 3, 1>(P, st);   // 470 with 2 producers
    default: break;
  }
  far3d_set_error("far3d_conv2d_nhwc: unknown wave-specialised GEMM tile %d", tile);
  return FAR3D_ERR_ARG;
}
