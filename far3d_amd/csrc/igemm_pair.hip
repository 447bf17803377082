// Instantiations of the conv / GEMM kernels for pair-stored activations (FAR3D_DT_BF16_PAIR, common.hpp): the fast path of the
// "bf16x3" precision mode.  Separate translation unit so that it compiles in parallel with igemm.hip.
// Tile ids (far3d_conv2d_nhwc `tile` with x_dt = FAR3D_DT_BF16_PAIR), t = the bf16 id of the same workgroup shape:
//   100 + t  split products (3 MFMAs per product, NT = 3): 3x3/s1/p1 t in 50..67, 90..97; 1x1/s1 t in 70..81
//   200 + t  hi planes only (one bf16 product, NT = 1), a subset of the shapes
//   1..5     register-staged kernel (any kernel size / stride), split products
//   330, 331 3x3 / stride 2 / pad 1 on the LDS-patch kernel, split products (round 5)
#include "igemm_kernels.hpp"

int far3d_conv_pair_launch(const IgemmParams& P, int tile, hipStream_t st) {
  const bool k3 = P.KH == 3 && P.KW == 3 && P.stride == 1 && P.pad == 1 && P.Ho == P.H && P.Wo == P.W;
  const bool k1 = P.KH == 1 && P.KW == 1 && P.stride == 1 && P.pad == 0;
  if (tile == 0) tile = k3 ? 160 : (k1 ? 179 : 0);
  if (tile == 0) {   // strided / odd kernels: the heuristic of the register-staged kernel
    const long Npix = (long)P.N * P.Ho * P.Wo;
    const long big = ((Npix + 127) / 128) * ((P.Cout + 127) / 128);
    if (P.Cout <= 64) tile = (Npix >= 128 * 512) ? 2 : 3;
    else if (big >= 512) tile = 1;
    else if (((Npix + 63) / 64) * ((P.Cout + 127) / 128) >= 512) tile = 4;
    else tile = 3;
  }
  if (tile <= 5) {
    switch (tile) {
      case 1: launch_igemm<pair_t, split_t, 2, 2, 2, 2>(P, st); break;
      case 2: launch_igemm<pair_t, split_t, 2, 2, 1, 2>(P, st); break;
      case 3: launch_igemm<pair_t, split_t, 2, 2, 1, 1>(P, st); break;
      case 4: launch_igemm<pair_t, split_t, 2, 2, 2, 1>(P, st); break;
      case 5: launch_igemm<pair_t, split_t, 1, 4, 2, 2>(P, st); break;
      default: far3d_set_error("far3d_conv2d_nhwc: unknown tile %d", tile); return FAR3D_ERR_ARG;
    }
    return FAR3D_OK;
  }
  if (k1) {
    switch (tile) {
      case 170: return launch_gemm1x1_pipe<2, 2, 2, 2, 3, true>(P, st);   // 128 x 128, 4 waves
      case 171: return launch_gemm1x1_pipe<2, 4, 2, 1, 3, true>(P, st);   // 128 x 128, 8 waves
      case 172: return launch_gemm1x1_pipe<2, 4, 2, 2, 3, true>(P, st);   // 128 x 256, 8 waves
      case 173: return launch_gemm1x1_pipe<4, 2, 2, 2, 3, true>(P, st);   // 256 x 128, 8 waves
      case 174: return launch_gemm1x1_pipe<2, 2, 1, 2, 3, true>(P, st);   // 64 x 128, 4 waves
      case 175: return launch_gemm1x1_pipe<2, 4, 1, 1, 3, true>(P, st);   // 64 x 128, 8 waves
      case 176: return launch_gemm1x1_pipe<2, 2, 2, 1, 3, true>(P, st);   // 128 x 64, 4 waves
      case 177: return launch_gemm1x1_pipe<4, 4, 2, 1, 3, true>(P, st);   // 256 x 128, 16 waves
      case 178: return launch_gemm1x1_pipe<2, 4, 1, 2, 3, true>(P, st);   // 64 x 256, 8 waves
      case 179: return launch_gemm1x1_pipe<4, 2, 1, 2, 3, true>(P, st);   // 128 x 128, 8 waves (1x2 tiles per wave)
      case 180: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true>(P, st);   // 64 x 64, 4 waves
      case 181: return launch_gemm1x1_pipe<4, 4, 1, 1, 3, true>(P, st);   // 128 x 128, 16 waves
      // round 6: pixel tiles of 160 / 96 for the maps a 128-pixel grid leaves half empty (stage 5's 4 200 pixels x 1 024 channels are 264
      // workgroups of 128 x 128 = 1.03 per CU; 128 x 160 is 216 -- one round: 64 -> 58 us; FPN lateral 1: 33 -> 29 us).  192-channel tiles
      // for stage 4 (792 workgroups = 3.09 per CU) were measured too and are SLOWER than 128 x 128 (142-168 against 116-135 us,
      // profiles/r6/tune_pair_fill_tiles.log): two co-resident workgroups per CU already even that grid out.
      case 185: return launch_gemm1x1_pipe<2, 5, 2, 1, 3, true>(P, st);   // 128 x 160, 10 waves
      case 186: return launch_gemm1x1_pipe<4, 1, 1, 5, 3, true>(P, st);   // 128 x 160, 4 waves of 32 ch x 160 px
      case 187: return launch_gemm1x1_pipe<2, 3, 1, 1, 3, true>(P, st);   // 64 x 96, 6 waves
      case 188: return launch_gemm1x1_pipe<1, 3, 2, 1, 3, true>(P, st);   // 64 x 96, 3 waves
      case 279: return launch_gemm1x1_pipe<4, 2, 1, 2, 1, true>(P, st);   // hi only: 128 x 128, 8 waves
      case 280: return launch_gemm1x1_pipe<2, 2, 1, 1, 1, true>(P, st);   // hi only: 64 x 64, 4 waves
      default: break;
    }
  }
  if (k3) {
    switch (tile) {
      case 150: return launch_conv3x3_pipe<2, 2, 1, 2, 2, 1, 3, true>(P, st);   // 64 x 4 rows
      case 152: return launch_conv3x3_pipe<1, 4, 1, 1, 2, 1, 3, true>(P, st);   // 32 x 4 rows
      case 153: return launch_conv3x3_pipe<2, 2, 2, 2, 2, 1, 3, true>(P, st);   // 128 x 4 rows
      case 154: return launch_conv3x3_pipe<1, 4, 1, 2, 2, 1, 3, true>(P, st);   // 32 x 8 rows
      case 155: return launch_conv3x3_pipe<2, 2, 1, 1, 2, 1, 3, true>(P, st);   // 64 x 2 rows
      case 157: return launch_conv3x3_pipe<2, 2, 1, 4, 2, 1, 3, true>(P, st);   // 64 x 8 rows, 4 waves
      case 159: return launch_conv3x3_pipe<1, 4, 3, 1, 2, 1, 3, true>(P, st);   // 96 x 4 rows
      case 160: return launch_conv3x3_pipe<2, 4, 1, 2, 2, 1, 3, true>(P, st);   // 64 x 8 rows, 8 waves
      case 161: return launch_conv3x3_pipe<2, 4, 1, 1, 2, 1, 3, true>(P, st);   // 64 x 4 rows, 8 waves
      case 162: return launch_conv3x3_pipe<4, 2, 1, 2, 2, 1, 3, true>(P, st);   // 128 x 4 rows, 8 waves
      case 163: return launch_conv3x3_pipe<2, 8, 1, 1, 2, 1, 3, true>(P, st);   // 64 x 8 rows, 16 waves
      case 164: return launch_conv3x3_pipe<1, 8, 2, 1, 2, 1, 3, true>(P, st);   // 64 x 8 rows, 8 waves of 64 ch x 1 row
      case 165: return launch_conv3x3_pipe<1, 8, 1, 1, 2, 1, 3, true>(P, st);   // 32 x 8 rows, 8 waves
      case 166: return launch_conv3x3_pipe<4, 4, 1, 1, 2, 1, 3, true>(P, st);   // 128 x 4 rows, 16 waves
      case 167: return launch_conv3x3_pipe<4, 2, 1, 1, 2, 1, 3, true>(P, st);   // 128 x 2 rows, 8 waves
      case 168: return launch_conv3x3_pipe<2, 4, 2, 1, 2, 1, 3, true>(P, st);   // 128 x 4 rows, 8 waves of 64 ch x 1 row
      // 7 rows: stage 4's 40-row maps are 6 x 7 (252 workgroups of 64 channels on 256 CUs) instead of 5 x 8 (210)
      case 169: return launch_conv3x3_pipe<2, 7, 1, 1, 2, 1, 3, true>(P, st);   // 64 x 7 rows, 14 waves
      case 190: return launch_conv3x3_pipe<1, 7, 1, 1, 2, 1, 3, true>(P, st);   // 32 x 7 rows, 7 waves
      case 198: return launch_conv3x3_pipe<1, 7, 2, 1, 2, 1, 3, true>(P, st);   // 64 x 7 rows, 7 waves of 64 ch x 1 row
      // 3-deep weight ring
      case 191: return launch_conv3x3_pipe<2, 4, 1, 1, 3, 1, 3, true>(P, st);   // 64 x 4 rows, 8 waves
      case 192: return launch_conv3x3_pipe<1, 8, 1, 1, 3, 1, 3, true>(P, st);   // 32 x 8 rows, 8 waves
      case 193: return launch_conv3x3_pipe<1, 4, 1, 1, 3, 1, 3, true>(P, st);   // 32 x 4 rows, 4 waves
      case 197: return launch_conv3x3_pipe<2, 2, 1, 1, 3, 1, 3, true>(P, st);   // 64 x 2 rows, 4 waves
      // hi planes only
      case 260: return launch_conv3x3_pipe<2, 4, 1, 2, 2, 1, 1, true>(P, st);   // 64 x 8 rows, 8 waves
      case 265: return launch_conv3x3_pipe<1, 8, 1, 1, 2, 1, 1, true>(P, st);   // 32 x 8 rows, 8 waves
      case 252: return launch_conv3x3_pipe<1, 4, 1, 1, 2, 1, 1, true>(P, st);   // 32 x 4 rows, 4 waves
      default: break;
    }
  }
  const bool k3s2 = P.KH == 3 && P.KW == 3 && P.stride == 2 && P.pad == 1 && P.Ho == (P.H - 1) / 2 + 1 && P.Wo == (P.W - 1) / 2 + 1;
  if (k3s2) {      // 3x3 / stride 2 / pad 1 on the LDS-patch kernel (de-interleaved patch rows), split products: the LDS holds 2 output rows
    switch (tile) {
      case 330: return launch_conv3x3_pipe<2, 2, 1, 1, 2, 1, 3, true, 2>(P, st);   // 64 x 2 rows, 4 waves
      case 331: return launch_conv3x3_pipe<1, 2, 1, 1, 2, 1, 3, true, 2>(P, st);   // 32 x 2 rows, 2 waves
      default: break;
    }
  }
  far3d_set_error("far3d_conv2d_nhwc: tile %d is not available for pair-stored activations with k=%d stride=%d", tile, P.KH, P.stride);
  return FAR3D_ERR_ARG;
}

// fp32 activation rows x pre-split weights on the pipelined GEMM kernel (F32B: the rows are handed over as if pair-stored -- P.ldx and
// P.x_img_stride already doubled by the caller; 1x1 / stride 1, Cin % 32 == 0).  Tile ids 479-481 = the pair ids 179-181 + 300.
int far3d_conv_f32rows_launch(const IgemmParams& P, int tile, hipStream_t st) {
  switch (tile) {
    case 479: return launch_gemm1x1_pipe<4, 2, 1, 2, 3, true, 2, true>(P, st);   // 128 x 128, 8 waves (1x2 tiles per wave)
    case 480: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true, 2, true>(P, st);   // 64 x 64, 4 waves
    case 481: return launch_gemm1x1_pipe<4, 4, 1, 1, 3, true, 2, true>(P, st);   // 128 x 128, 16 waves
    default: break;
  }
  far3d_set_error("far3d_conv2d_nhwc: tile %d is not available for fp32 activation rows with split weights (479-481)", tile);
  return FAR3D_ERR_ARG;
}

// fp32 activation rows x fp32 weight rows, EXACT fp32 MFMA on the pipelined GEMM kernel (F32X; 1x1 / stride 1, Cin % 32 == 0; P.ldx and
// P.x_img_stride doubled by the caller: the kernel counts bf16-sized units).  Tile ids 482-494.
int far3d_conv_f32x_launch(const IgemmParams& P, int tile, hipStream_t st) {
  switch (tile) {
    case 482: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true, 2, true, true>(P, st);   // 64 x 64, 4 waves, 2 stages
    case 483: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true, 4, true, true>(P, st);   // 64 x 64, 4 waves, 4 stages (3 steps in flight)
    case 484: return launch_gemm1x1_pipe<4, 2, 1, 2, 3, true, 3, true, true>(P, st);   // 128 x 128, 8 waves, 3 stages
    case 485: return launch_gemm1x1_pipe<4, 4, 1, 1, 3, true, 2, true, true>(P, st);   // 128 x 128, 16 waves, 2 stages
    case 486: return launch_gemm1x1_pipe<2, 4, 1, 1, 3, true, 2, true, true>(P, st);   // 64 x 128, 8 waves, 2 stages
    // K groups inside the workgroup (igemm_kernels.hpp, KS): (channels x rows) tile, waves = groups x waves per group
    case 487: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true, 2, true, true, 2>(P, st);   // 64 x 64, 2 groups x 4 waves
    case 488: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true, 2, true, true, 4>(P, st);   // 64 x 64, 4 groups x 4 waves
    case 489: return launch_gemm1x1_pipe<1, 2, 1, 1, 3, true, 2, true, true, 4>(P, st);   // 32 x 64, 4 groups x 2 waves
    case 490: return launch_gemm1x1_pipe<2, 1, 1, 1, 3, true, 2, true, true, 4>(P, st);   // 64 x 32, 4 groups x 2 waves
    case 491: return launch_gemm1x1_pipe<1, 1, 1, 1, 3, true, 2, true, true, 4>(P, st);   // 32 x 32, 4 groups x 1 wave
    case 492: return launch_gemm1x1_pipe<1, 1, 1, 1, 3, true, 2, true, true, 8>(P, st);   // 32 x 32, 8 groups x 1 wave
    case 493: return launch_gemm1x1_pipe<1, 2, 1, 1, 3, true, 2, true, true, 2>(P, st);   // 32 x 64, 2 groups x 2 waves
    case 494: return launch_gemm1x1_pipe<2, 2, 1, 1, 3, true, 3, true, true, 2>(P, st);   // 64 x 64, 2 groups x 4 waves, 3 stages
    default: break;
  }
  far3d_set_error("far3d_conv2d_nhwc: tile %d is not available for exact-fp32 rows on the pipelined kernel (482-494)", tile);
  return FAR3D_ERR_ARG;
}
