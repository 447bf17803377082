// far3d_conv2d_nhwc: argument checks and tile dispatch; the kernels live in igemm_kernels.hpp (shared with igemm_pair.hip,
// which holds the instantiations for pair-stored activations so that the two halves compile in parallel).
#include "igemm_kernels.hpp"

int far3d_conv_pair_launch(const IgemmParams& P, int tile, hipStream_t st);   // igemm_pair.hip
int far3d_conv_f32rows_launch(const IgemmParams& P, int tile, hipStream_t st);   // igemm_pair.hip
int far3d_gemm_ws_launch(const IgemmParams& P, int tile, hipStream_t st);        // conv_ws.hip (persistent wave-specialised 1x1 GEMM, tiles 460-477)
int far3d_conv_f32x_launch(const IgemmParams& P, int tile, hipStream_t st);     // igemm_pair.hip (exact fp32 on the pipelined kernel, tiles 482-494)
int far3d_conv_ws_launch(const IgemmParams& P, int tile, hipStream_t st);        // conv_ws.hip (persistent wave-specialised 3x3, tiles 400-459)

#ifdef FAR3D_PROFILING
// tools/conv_phase_times.py: where the per-workgroup stamps of the pipelined conv / GEMM kernels go (8 x uint64 per workgroup)
static unsigned long long* g_conv_ts = nullptr;
extern "C" int far3d_prof_set_conv_timestamps(void* buf) { g_conv_ts = (unsigned long long*)buf; return 0; }
#endif

// See include/far3d_hip.h for the argument contract.
extern "C" int far3d_conv2d_nhwc(const void* x, int x_dt, const void* w, int w_dt, const float* bias, void* y,
                                 int y_dt, int N, int H, int W, int Cin, int ldx, long x_img_stride, int Ho,
                                 int Wo, int Cout, int ldy, long y_img_stride, int KH, int KW, int stride,
                                 int pad, int act, const void* res, int res_dt, int ldr, long res_img_stride,
                                 int Hr, int Wr, void* y2, int y2_dt, int ldy2, long y2_img_stride,
                                 const float* y2_scale, const float* y2_shift, long long* chan_sums, int tile, void* stream) {
  FAR3D_CHECK_ARG(x && w && y, "far3d_conv2d_nhwc: null x/w/y");
  FAR3D_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Ho > 0 && Wo > 0 && Cout > 0,
                  "far3d_conv2d_nhwc: bad sizes N=%d H=%d W=%d Cin=%d Ho=%d Wo=%d Cout=%d", N, H, W, Cin, Ho, Wo, Cout);
  FAR3D_CHECK_ARG(KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "far3d_conv2d_nhwc: bad kernel geometry");
  FAR3D_CHECK_ARG((Ho - 1) * stride - pad + KH - 1 < H + pad + stride && (Wo - 1) * stride - pad + KW - 1 < W + pad + stride,
                  "far3d_conv2d_nhwc: output size %dx%d inconsistent with input %dx%d k=%d s=%d p=%d", Ho, Wo, H, W, KH, stride, pad);
  const bool pair_in = x_dt == FAR3D_DT_BF16_PAIR, pair_out = y_dt == FAR3D_DT_BF16_PAIR;
  FAR3D_CHECK_ARG(ldx >= Cin * (pair_in ? 2 : 1) && ldy >= Cout * (pair_out ? 2 : 1), "far3d_conv2d_nhwc: pixel strides smaller than channel counts");
  FAR3D_CHECK_ARG((x_dt == FAR3D_DT_F32 || x_dt == FAR3D_DT_BF16 || pair_in) &&
                  (w_dt == FAR3D_DT_F32 || w_dt == FAR3D_DT_BF16 || w_dt == FAR3D_DT_F32_BF16X3) &&
                  (y_dt == FAR3D_DT_F32 || y_dt == FAR3D_DT_BF16 || pair_out), "far3d_conv2d_nhwc: unsupported dtype");
  FAR3D_CHECK_ARG(pair_in ? (w_dt == FAR3D_DT_F32_BF16X3 && y_dt != FAR3D_DT_BF16 && Cin % 32 == 0) : !pair_out,
                  "far3d_conv2d_nhwc: pair-stored activations need split weights (w_dt 2), Cin %% 32 == 0 and a pair or f32 output; "
                  "pair outputs come from pair inputs");
  FAR3D_CHECK_ARG(!pair_out || Cout % 32 == 0, "far3d_conv2d_nhwc: pair-stored output needs Cout %% 32 == 0 (got %d)", Cout);
  FAR3D_CHECK_ARG(!(res && res_dt == FAR3D_DT_BF16_PAIR) || Cout % 32 == 0, "far3d_conv2d_nhwc: pair-stored residual needs Cout %% 32 == 0");
  FAR3D_CHECK_ARG(!(y2 && y2_dt == FAR3D_DT_BF16_PAIR), "far3d_conv2d_nhwc: the second output is f32 or bf16");
  FAR3D_CHECK_ARG(!(x_dt == FAR3D_DT_BF16 && w_dt != FAR3D_DT_BF16),
                  "far3d_conv2d_nhwc: bf16 activations with fp32 weights is not a supported combination");
  FAR3D_CHECK_ARG(act >= 0 && act <= 2, "far3d_conv2d_nhwc: unknown activation %d", act);
  FAR3D_CHECK_ARG(!y2 || (y2_scale && y2_shift), "far3d_conv2d_nhwc: y2 needs scale and shift");
  IgemmParams P;
  memset(&P, 0, sizeof(P));
  P.x = x; P.w = w; P.bias = bias; P.y = y; P.y2 = y2; P.y2_scale = y2_scale; P.y2_shift = y2_shift; P.res = res;
  P.x_img_stride = x_img_stride; P.y_img_stride = y_img_stride; P.y2_img_stride = y2_img_stride;
  P.res_img_stride = res_img_stride;
  P.N = N; P.H = H; P.W = W; P.Cin = Cin; P.ldx = ldx; P.Ho = Ho; P.Wo = Wo; P.Cout = Cout; P.ldy = ldy;
  P.KH = KH; P.KW = KW; P.stride = stride; P.pad = pad;
  P.cin_pad = (Cin + 31) / 32 * 32;
  P.nsteps = KH * KW * P.cin_pad / 32;
  P.act = act; P.y_dt = y_dt; P.y2_dt = y2_dt; P.ldy2 = ldy2;
  P.res_dt = res_dt; P.ldr = ldr; P.Hr = res ? Hr : Ho; P.Wr = res ? Wr : Wo;
  const int xe = x_dt == FAR3D_DT_F32 ? 4 : 2;
  const int ve = w_dt == FAR3D_DT_BF16 ? 8 : 4;  // elements per staged 16-byte chunk of the compute type
  auto aligned = [](const void* p, long a) { return ((uintptr_t)p % a) == 0; };
  P.x_vec = aligned(x, 16) && (ldx % ve == 0) && (x_img_stride % ve == 0) && (long)ve * xe % 16 == 0;
  P.y_vec = aligned(y, 16) && (ldy % 4 == 0) && (y_img_stride % 4 == 0);
  P.y2_vec = y2 && aligned(y2, 16) && (ldy2 % 4 == 0) && (y2_img_stride % 4 == 0);
  P.chan_sums = chan_sums; P.sums_hw = Ho * Wo;
#ifdef FAR3D_PROFILING
  P.prof = g_conv_ts;
#endif
  P.y_rows16 = (y_dt == FAR3D_DT_BF16 || pair_out) && !res && !y2 && aligned(y, 16) && (ldy % 8 == 0) && (y_img_stride % 8 == 0) && (Cout % 8 == 0);
  hipStream_t st = (hipStream_t)stream;
  if (chan_sums) {       // only the pipelined GEMM kernels accumulate them; anything else is an error, not a silent fallback
    const bool gemm_tile = pair_in ? (tile == 0 || (tile >= 170 && tile <= 181) || (tile >= 185 && tile <= 188) || tile == 279 || tile == 280 || (tile >= 460 && tile < 478))
                                   : ((tile >= 70 && tile <= 89) || (tile >= 110 && tile <= 117) || (tile >= 120 && tile <= 129) || (tile >= 140 && tile <= 145));
    FAR3D_CHECK_ARG(KH == 1 && KW == 1 && stride == 1 && pad == 0 && gemm_tile && (pair_in || (x_dt == FAR3D_DT_BF16 && Cin % 32 == 0 && P.x_vec)),
                    "far3d_conv2d_nhwc: channel sums need a 1x1 / stride 1 layer on a pipelined GEMM tile (bf16: 70-89, 110-117, 120-129, 140-145; pair: 170-181, 185-188, 279, 280, 460-477); got k=%d tile=%d", KH, tile);
  }
  if (tile >= 400 && tile < 460) {     // persistent wave-specialised 3x3 kernel: a refusal is an error of the call, never a silent fallback
    const bool pair_tile = tile < 420 || tile >= 440;
    FAR3D_CHECK_ARG(KH == 3 && KW == 3 && stride == 1 && pad == 1 && Ho == H && Wo == W && !res && !y2 && !chan_sums && Cin % 32 == 0 && Cout % 32 == 0 &&
                    pair_tile == pair_in && (pair_in ? pair_out : (x_dt == FAR3D_DT_BF16 && w_dt == FAR3D_DT_BF16 && y_dt == FAR3D_DT_BF16)) &&
                    aligned(x, 16) && aligned(y, 16) && ldx % 8 == 0 && ldy % 8 == 0 && x_img_stride % 8 == 0 && y_img_stride % 8 == 0 &&
                    (long)N * H * W < (1L << 31) - 4096,
                    "far3d_conv2d_nhwc: tile %d (wave-specialised 3x3) needs a 3x3 / stride 1 / pad 1 layer, Cin and Cout multiples of 32, %s in and out, "
                    "16-byte aligned rows and no residual / second output / channel sums", tile, pair_tile ? "pair-stored" : "bf16");
    const int rc = far3d_conv_ws_launch(P, tile, st);
    if (rc != FAR3D_OK) return rc;
    FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
    return FAR3D_OK;
  }
  if (tile >= 460 && tile < 478) {     // persistent wave-specialised 1x1 GEMM on pair-stored maps: a refusal is an error of the call
    FAR3D_CHECK_ARG(KH == 1 && KW == 1 && stride == 1 && pad == 0 && !res && !y2 && Cin % 32 == 0 && Cout % 32 == 0 && pair_in && pair_out &&
                    aligned(x, 16) && aligned(y, 16) && ldx % 8 == 0 && ldy % 8 == 0 && x_img_stride % 8 == 0 && y_img_stride % 8 == 0 &&
                    (long)N * H * W < (1L << 31) - 4096 && ((long)(N - 1) * x_img_stride + (long)H * W * ldx) * 2 < 0x7fffffffL,
                    "far3d_conv2d_nhwc: tile %d (wave-specialised GEMM) needs a 1x1 / stride 1 layer on pair-stored maps, Cin and Cout multiples of 32, "
                    "16-byte aligned rows, an input map below 2 GB and no residual / second output", tile);
    const int rc = far3d_gemm_ws_launch(P, tile, st);
    if (rc != FAR3D_OK) return rc;
    FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
    return FAR3D_OK;
  }
  if (pair_in) {
    FAR3D_CHECK_ARG(aligned(x, 16) && ldx % 8 == 0 && x_img_stride % 8 == 0 && (!pair_out || (aligned(y, 8) && ldy % 4 == 0 && y_img_stride % 4 == 0)),
                    "far3d_conv2d_nhwc: pair-stored tensors must be 16-byte aligned with pixel strides that are multiples of 8 elements");
    const int rc = far3d_conv_pair_launch(P, tile, st);
    if (rc != FAR3D_OK) return rc;
    FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
    return FAR3D_OK;
  }
  const long Npix = (long)N * Ho * Wo;
  FAR3D_CHECK_ARG(Npix < (1L << 31) - 4096 && (long)N * H * W < (1L << 31) - 4096, "far3d_conv2d_nhwc: %ld pixels: the kernels index pixels with 32 bits", Npix);
  // fp32 activation rows x pre-split weights, 1x1: the pipelined GEMM kernel with in-register hi / lo split of the rows (tiles 479-481;
  // auto).  32 floats = the 128 bytes of a pair-stored 32-channel block: the rows go in as pair rows of twice the stride.
  if (x_dt == FAR3D_DT_F32 && w_dt == FAR3D_DT_F32_BF16X3 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && Cin % 32 == 0 && !chan_sums &&
      aligned(x, 16) && ldx % 4 == 0 && x_img_stride % 4 == 0 && ((long)(N - 1) * x_img_stride + (long)Ho * Wo * ldx) * 4 < 0x7fffffffL &&
      (tile == 0 || (tile >= 479 && tile <= 481))) {
    IgemmParams Q = P;
    Q.ldx = 2 * ldx; Q.x_img_stride = 2 * x_img_stride;
    // auto: the 64 x 64 tile (measured 7.3-10 us on the decoder's GEMMs against 11-18 for the 128 x 128 ones and 12-19 for the staged
    // exact-fp32 kernel, profiles/r5/fp32_rows_gemm.txt); the 8-wave 128 x 128 tile only when even that one fills the chip 8 times over
    const int t = tile ? tile : ((((Npix + 127) / 128) * ((Cout + 127) / 128) >= 2048) ? 479 : 480);
    const int rc = far3d_conv_f32rows_launch(Q, t, st);
    if (rc != FAR3D_OK) return rc;
    FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
    return FAR3D_OK;
  }
  // fp32 rows x fp32 weights, 1x1, explicit tiles 482-494: EXACT fp32 MFMA on the pipelined LDS-DMA kernel (far3d_amd.ops.linear picks them
  // for the decoder-sized GEMMs; tile 0 keeps the register-staged kernel, which takes every shape)
  if (tile >= 482 && tile <= 494) {
    FAR3D_CHECK_ARG(x_dt == FAR3D_DT_F32 && w_dt == FAR3D_DT_F32 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && Cin % 32 == 0 && !chan_sums &&
                    aligned(x, 16) && ldx % 4 == 0 && x_img_stride % 4 == 0 && ((long)(N - 1) * x_img_stride + (long)Ho * Wo * ldx) * 4 < 0x7fffffffL,
                    "far3d_conv2d_nhwc: tile %d (exact fp32 on the pipelined kernel) needs fp32 rows and fp32 weights, a 1x1 / stride 1 layer, Cin %% 32 == 0 "
                    "and 16-byte aligned rows", tile);
    IgemmParams Q = P;
    Q.ldx = 2 * ldx; Q.x_img_stride = 2 * x_img_stride;
    const int rc = far3d_conv_f32x_launch(Q, tile, st);
    if (rc != FAR3D_OK) return rc;
    FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
    return FAR3D_OK;
  }
  // tile: 0 = auto.  (channels x pixels per workgroup) 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 128x64, 5 = 64x256
  if (tile == 0) {   // fallback heuristic; far3d_amd/data/tuning_mi355x.json holds measured per-shape winners
    const long big = ((Npix + 127) / 128) * ((Cout + 127) / 128);
    if (Cout <= 64) tile = (Npix >= 128 * 512) ? 2 : 3;
    else if (big >= 512) tile = 1;
    else if (((Npix + 63) / 64) * ((Cout + 127) / 128) >= 512) tile = 4;
    else tile = 3;
  }
#define LAUNCH(TIN, TCC)                                                        \
  switch (tile) {                                                               \
    case 1: launch_igemm<TIN, TCC, 2, 2, 2, 2>(P, st); break;                   \
    case 2: launch_igemm<TIN, TCC, 2, 2, 1, 2>(P, st); break;                   \
    case 3: launch_igemm<TIN, TCC, 2, 2, 1, 1>(P, st); break;                   \
    case 4: launch_igemm<TIN, TCC, 2, 2, 2, 1>(P, st); break;                   \
    case 5: launch_igemm<TIN, TCC, 1, 4, 2, 2>(P, st); break;                   \
    default: far3d_set_error("far3d_conv2d_nhwc: unknown tile %d", tile); return FAR3D_ERR_ARG; \
  }
  if (x_dt == FAR3D_DT_BF16 && w_dt == FAR3D_DT_BF16 && (Cin % 32) == 0 && P.x_vec) {
    // global_load_lds ring kernel (any kernel size / stride): (channels x pixels, ring depth) 1 128x128/3  2 64x128/4  3 64x64/4
    // 4 128x64/4; 5 falls back to the register-staged 64x256
    bool done = true;
    int rc = 0;
    switch (tile) {
      case 1: rc = launch_igemm_dma<2, 2, 2, 2, 3>(P, st); break;
      case 2: rc = launch_igemm_dma<2, 2, 1, 2, 4>(P, st); break;
      case 3: rc = launch_igemm_dma<2, 2, 1, 1, 4>(P, st); break;
      case 4: rc = launch_igemm_dma<2, 2, 2, 1, 4>(P, st); break;
      // several 32-channel K chunks per barrier step
      case 18: rc = launch_igemm_dma<2, 2, 1, 1, 3, 3>(P, st); break;   // 64x64, 3 chunks/step
      // 2-deep rings: less LDS -> more resident workgroups per CU
      case 43: rc = launch_igemm_dma<2, 2, 2, 2, 2>(P, st); break;      // 128x128
      case 46: rc = launch_igemm_dma<2, 2, 2, 1, 2>(P, st); break;      // 128x64
      case 48: rc = launch_igemm_dma<2, 2, 1, 1, 2>(P, st); break;      // 64x64
      default: done = false;
    }
    if (!done && KH == 1 && KW == 1 && stride == 1 && pad == 0) {   // pipelined GEMM kernel (channels x pixels, waves)
      done = true;
      switch (tile) {
        case 70: rc = launch_gemm1x1_pipe<2, 2, 2, 2>(P, st); break;   // 128 x 128, 4 waves
        case 71: rc = launch_gemm1x1_pipe<2, 4, 2, 1>(P, st); break;   // 128 x 128, 8 waves
        case 72: rc = launch_gemm1x1_pipe<2, 4, 2, 2>(P, st); break;   // 128 x 256, 8 waves
        case 73: rc = launch_gemm1x1_pipe<4, 2, 2, 2>(P, st); break;   // 256 x 128, 8 waves
        case 74: rc = launch_gemm1x1_pipe<2, 2, 1, 2>(P, st); break;   // 64 x 128, 4 waves
        case 75: rc = launch_gemm1x1_pipe<2, 4, 1, 1>(P, st); break;   // 64 x 128, 8 waves
        case 76: rc = launch_gemm1x1_pipe<2, 2, 2, 1>(P, st); break;   // 128 x 64, 4 waves
        case 77: rc = launch_gemm1x1_pipe<4, 4, 2, 1>(P, st); break;   // 256 x 128, 16 waves
        case 78: rc = launch_gemm1x1_pipe<2, 4, 1, 2>(P, st); break;   // 64 x 256, 8 waves
        case 79: rc = launch_gemm1x1_pipe<4, 2, 1, 2>(P, st); break;   // 128 x 128, 8 waves (1x2 tiles per wave)
        case 80: rc = launch_gemm1x1_pipe<2, 2, 1, 1>(P, st); break;   // 64 x 64, 4 waves
        case 81: rc = launch_gemm1x1_pipe<4, 4, 1, 1>(P, st); break;   // 128 x 128, 16 waves
        // deeper LDS rings (NS - 1 steps of 64 channels in flight, counted vmcnt)
        case 82: rc = launch_gemm1x1_pipe<4, 2, 1, 2, 1, false, 3>(P, st); break;   // 128 x 128, 8 waves, 3 stages
        case 83: rc = launch_gemm1x1_pipe<2, 2, 1, 2, 1, false, 3>(P, st); break;   // 64 x 128, 4 waves, 3 stages
        case 84: rc = launch_gemm1x1_pipe<2, 2, 2, 2, 1, false, 3>(P, st); break;   // 128 x 128, 4 waves, 3 stages
        case 85: rc = launch_gemm1x1_pipe<4, 2, 2, 2, 1, false, 3>(P, st); break;   // 256 x 128, 8 waves, 3 stages
        case 86: rc = launch_gemm1x1_pipe<2, 2, 1, 2, 1, false, 4>(P, st); break;   // 64 x 128, 4 waves, 4 stages
        case 87: rc = launch_gemm1x1_pipe<2, 2, 1, 1, 1, false, 4>(P, st); break;   // 64 x 64, 4 waves, 4 stages
        case 88: rc = launch_gemm1x1_pipe<2, 2, 1, 1, 1, false, 3>(P, st); break;   // 64 x 64, 4 waves, 3 stages
        case 89: rc = launch_gemm1x1_pipe<2, 2, 2, 1, 1, false, 3>(P, st); break;   // 128 x 64, 4 waves, 3 stages
        // 256 x 256 tiles: half the L2 -> LDS bytes per MFMA of the 128 x 128 tiles (the GEMMs are fill-bound)
        case 110: rc = launch_gemm1x1_pipe<4, 2, 2, 4>(P, st); break;   // 8 waves of 64 ch x 128 px
        case 111: rc = launch_gemm1x1_pipe<2, 4, 4, 2>(P, st); break;   // 8 waves of 128 ch x 64 px
        case 112: rc = launch_gemm1x1_pipe<4, 4, 2, 2>(P, st); break;   // 16 waves of 64 x 64
        case 113: rc = launch_gemm1x1_pipe<2, 2, 4, 4>(P, st); break;   // 4 waves of 128 x 128
        case 114: rc = launch_gemm1x1_pipe<2, 4, 2, 4>(P, st); break;   // 128 ch x 512 px, 8 waves of 64 x 128
        case 115: rc = launch_gemm1x1_pipe<4, 2, 4, 2>(P, st); break;   // 512 ch x 128 px
        case 116: rc = launch_gemm1x1_pipe<2, 4, 4, 1>(P, st); break;   // 256 ch x 128 px, 8 waves of 128 x 32
        case 117: rc = launch_gemm1x1_pipe<4, 2, 1, 4>(P, st); break;   // 128 ch x 256 px, 8 waves of 32 x 128
        // full-line DMA pieces (8 rows x 128 B per instruction, 128-byte LDS rows)
        case 120: rc = launch_gemm1x1_wide<4, 2, 1, 2>(P, st); break;   // 128 x 128, 8 waves
        case 121: rc = launch_gemm1x1_wide<2, 2, 2, 2>(P, st); break;   // 128 x 128, 4 waves
        case 122: rc = launch_gemm1x1_wide<4, 2, 2, 2>(P, st); break;   // 256 x 128, 8 waves
        case 123: rc = launch_gemm1x1_wide<4, 2, 2, 4>(P, st); break;   // 256 x 256, 8 waves
        case 124: rc = launch_gemm1x1_wide<2, 2, 1, 2>(P, st); break;   // 64 x 128, 4 waves
        case 125: rc = launch_gemm1x1_wide<2, 2, 1, 1>(P, st); break;   // 64 x 64, 4 waves
        case 126: rc = launch_gemm1x1_wide<2, 4, 2, 2>(P, st); break;   // 128 x 256, 8 waves
        case 127: rc = launch_gemm1x1_wide<2, 4, 1, 2>(P, st); break;   // 64 x 256, 8 waves
        case 128: rc = launch_gemm1x1_wide<4, 4, 1, 1>(P, st); break;   // 128 x 128, 16 waves
        case 129: rc = launch_gemm1x1_wide<4, 2, 1, 2, 3>(P, st); break;   // 128 x 128, 8 waves, 3 stages
        // split rings, wave-specialised DMA issue (round 5): weight ring NSA deep, activation ring NSB deep (channels x pixels)
        case 140: rc = launch_gemm1x1_split<4, 2, 2, 4, 2, 3>(P, st); break;   // 256 x 256, 8 waves of 64 ch x 128 px, rings 2 + 3 (160 KiB)
        case 141: rc = launch_gemm1x1_split<2, 2, 4, 4, 2, 3>(P, st); break;   // 256 x 256, 4 waves of 128 ch x 128 px (256 accumulator registers), rings 2 + 3
        case 142: rc = launch_gemm1x1_split<2, 4, 2, 2, 2, 4>(P, st); break;   // 128 x 256, 8 waves of 64 x 64, rings 2 + 4 (160 KiB)
        case 143: rc = launch_gemm1x1_split<4, 2, 2, 2, 2, 4>(P, st); break;   // 256 x 128, 8 waves of 64 x 64, rings 2 + 4 (128 KiB)
        case 144: rc = launch_gemm1x1_split<4, 2, 2, 4, 2, 2>(P, st); break;   // 256 x 256, rings 2 + 2: tile 123 with the specialised issue (control)
        case 145: rc = launch_gemm1x1_split<4, 4, 2, 2, 2, 3>(P, st); break;   // 256 x 256, 16 waves of 64 x 64, rings 2 + 3
        default: done = false;
      }
    }
    // pipelined LDS-patch 3x3 kernel: (channels x rows-of-32-pixels, waves)
    if (!done && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Ho == H && Wo == W) {
      done = true;
      switch (tile) {
        // software-pipelined kernel (register double-buffered fragments, immediate-offset LDS addressing)
        case 50: rc = launch_conv3x3_pipe<2, 2, 1, 2>(P, st); break;   // 64 x 4 rows
        case 51: rc = launch_conv3x3_pipe<1, 4, 2, 1>(P, st); break;   // 64 x 4 rows (1x4 waves)
        case 52: rc = launch_conv3x3_pipe<1, 4, 1, 1>(P, st); break;   // 32 x 4 rows
        case 53: rc = launch_conv3x3_pipe<2, 2, 2, 2>(P, st); break;   // 128 x 4 rows
        case 54: rc = launch_conv3x3_pipe<1, 4, 1, 2>(P, st); break;   // 32 x 8 rows
        case 55: rc = launch_conv3x3_pipe<2, 2, 1, 1>(P, st); break;   // 64 x 2 rows
        case 57: rc = launch_conv3x3_pipe<2, 2, 1, 4>(P, st); break;   // 64 x 8 rows
        case 58: rc = launch_conv3x3_pipe<2, 2, 2, 1>(P, st); break;   // 128 x 2 rows
        case 59: rc = launch_conv3x3_pipe<1, 4, 3, 1>(P, st); break;   // 96 x 4 rows
        // 8 / 16 waves per workgroup sharing one patch + weight slab
        case 60: rc = launch_conv3x3_pipe<2, 4, 1, 2>(P, st); break;   // 64 x 8 rows, 8 waves
        case 61: rc = launch_conv3x3_pipe<2, 4, 1, 1>(P, st); break;   // 64 x 4 rows, 8 waves
        case 62: rc = launch_conv3x3_pipe<4, 2, 1, 2>(P, st); break;   // 128 x 4 rows, 8 waves
        case 63: rc = launch_conv3x3_pipe<2, 8, 1, 1>(P, st); break;   // 64 x 8 rows, 16 waves
        case 64: rc = launch_conv3x3_pipe<1, 8, 2, 1>(P, st); break;   // 64 x 8 rows, 8 waves of 64 ch x 1 row
        case 65: rc = launch_conv3x3_pipe<1, 8, 1, 1>(P, st); break;   // 32 x 8 rows, 8 waves
        case 66: rc = launch_conv3x3_pipe<4, 4, 1, 1>(P, st); break;   // 128 x 4 rows, 16 waves
        case 67: rc = launch_conv3x3_pipe<4, 2, 1, 1>(P, st); break;   // 128 x 2 rows, 8 waves
        // 3-deep weight ring (kernel rows prefetched two steps ahead)
        case 90: rc = launch_conv3x3_pipe<2, 4, 1, 2, 3>(P, st); break;   // 64 x 8 rows, 8 waves
        case 91: rc = launch_conv3x3_pipe<2, 4, 1, 1, 3>(P, st); break;   // 64 x 4 rows, 8 waves
        case 92: rc = launch_conv3x3_pipe<1, 8, 1, 1, 3>(P, st); break;   // 32 x 8 rows, 8 waves
        case 93: rc = launch_conv3x3_pipe<1, 4, 1, 1, 3>(P, st); break;   // 32 x 4 rows, 4 waves
        case 94: rc = launch_conv3x3_pipe<2, 2, 1, 2, 3>(P, st); break;   // 64 x 4 rows, 4 waves
        case 95: rc = launch_conv3x3_pipe<2, 8, 1, 1, 3>(P, st); break;   // 64 x 8 rows, 16 waves
        case 96: rc = launch_conv3x3_pipe<1, 8, 2, 1, 3>(P, st); break;   // 64 x 8 rows, 8 waves of 64 ch x 1 row
        case 97: rc = launch_conv3x3_pipe<2, 2, 1, 1, 3>(P, st); break;   // 64 x 2 rows, 4 waves
        // whole-chunk steps (9 taps per barrier) for the layers with a single workgroup per CU
        case 100: rc = launch_conv3x3_pipe<2, 4, 1, 2, 2, 3>(P, st); break;   // 64 x 8 rows, 8 waves
        case 101: rc = launch_conv3x3_pipe<1, 8, 1, 1, 2, 3>(P, st); break;   // 32 x 8 rows, 8 waves
        case 102: rc = launch_conv3x3_pipe<1, 4, 1, 1, 2, 3>(P, st); break;   // 32 x 4 rows, 4 waves
        case 103: rc = launch_conv3x3_pipe<2, 4, 1, 1, 2, 3>(P, st); break;   // 64 x 4 rows, 8 waves
        // whole-chunk steps, deeper rings (NSW - 1 chunks in flight, counted vmcnt).  Measured in round 4 and NOT faster anywhere
        // (profiles/r4/tune_bf16_3x3_deep_rings.log: stage-4 c1 21 -> 22 us, stage-5 c1 9.2 -> 10 us): the K-short layers are not
        // waiting for their DMA round trips; kept as tested tiles
        case 104: rc = launch_conv3x3_pipe<2, 4, 1, 1, 3, 3>(P, st); break;   // 64 x 4 rows, 8 waves, 3 chunks (150 KB)
        case 105: rc = launch_conv3x3_pipe<1, 4, 1, 1, 3, 3>(P, st); break;   // 32 x 4 rows, 4 waves, 3 chunks (94 KB)
        case 106: rc = launch_conv3x3_pipe<1, 4, 1, 1, 4, 3>(P, st); break;   // 32 x 4 rows, 4 waves, 4 chunks (126 KB)
        // fat tiles (round 5): 2x2 .. 2x4 / 4x2 / 5x1 MFMA tiles per wave -- half to a third of the L2 -> LDS bytes and of the
        // fragment reads per MFMA of the 1x2 tiles above, workgroups that live 4-8x longer (per-workgroup set-up, first fill and
        // epilogue amortised); for the layers with >= 4 rounds of workgroups (stem2, stage 2, stage 3, FPN / 2D-head level 0)
        case 130: rc = launch_conv3x3_pipe<2, 4, 2, 2>(P, st); break;      // 128 x 8 rows, 8 waves of 64 ch x 2 rows
        case 131: rc = launch_conv3x3_pipe<2, 4, 2, 4>(P, st); break;      // 128 x 16 rows, 8 waves of 64 ch x 4 rows
        case 132: rc = launch_conv3x3_pipe<1, 8, 4, 1>(P, st); break;      // 128 x 8 rows, 8 waves of 128 ch x 1 row
        case 133: rc = launch_conv3x3_pipe<2, 4, 1, 4>(P, st); break;      // 64 x 16 rows, 8 waves of 32 ch x 4 rows
        case 134: rc = launch_conv3x3_pipe<1, 8, 5, 1>(P, st); break;      // 160 x 8 rows, 8 waves of 160 ch x 1 row (stage 3: all channels)
        case 135: rc = launch_conv3x3_pipe<1, 8, 5, 1, 3>(P, st); break;   // 134 with a 3-deep weight ring
        case 136: rc = launch_conv3x3_pipe<2, 2, 2, 4>(P, st); break;      // 128 x 8 rows, 4 waves of 64 ch x 4 rows
        case 137: rc = launch_conv3x3_pipe<2, 4, 2, 2, 3>(P, st); break;   // 130 with a 3-deep weight ring
        case 138: rc = launch_conv3x3_pipe<1, 8, 2, 2>(P, st); break;      // 64 x 16 rows, 8 waves of 64 ch x 2 rows (stem2: Cout 64)
        case 139: rc = launch_conv3x3_pipe<1, 8, 3, 1>(P, st); break;      // 96 x 8 rows, 8 waves of 96 ch x 1 row
        default: done = false;
      }
    }
    // 3x3 / stride 2 / pad 1 on the LDS-patch kernel (round 5; de-interleaved patch rows): (channels x output rows of 32 pixels, waves)
    if (!done && KH == 3 && KW == 3 && stride == 2 && pad == 1 && Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1) {
      done = true;
      switch (tile) {
        case 30: rc = launch_conv3x3_pipe<4, 2, 1, 2, 2, 1, 1, false, 2>(P, st); break;   // 128 x 4 rows, 8 waves
        case 31: rc = launch_conv3x3_pipe<2, 2, 1, 1, 2, 1, 1, false, 2>(P, st); break;   // 64 x 2 rows, 4 waves
        case 32: rc = launch_conv3x3_pipe<4, 2, 1, 1, 2, 1, 1, false, 2>(P, st); break;   // 128 x 2 rows, 8 waves
        case 33: rc = launch_conv3x3_pipe<2, 4, 1, 1, 2, 1, 1, false, 2>(P, st); break;   // 64 x 4 rows, 8 waves
        case 34: rc = launch_conv3x3_pipe<2, 2, 1, 2, 2, 1, 1, false, 2>(P, st); break;   // 64 x 4 rows, 4 waves
        case 35: rc = launch_conv3x3_pipe<4, 2, 1, 2, 3, 1, 1, false, 2>(P, st); break;   // 30 with a 3-deep weight ring
        default: done = false;
      }
    }
    if (done) {
      if (rc) return rc;
      FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
      return FAR3D_OK;
    }
  }
  if (tile > 5) { far3d_set_error("far3d_conv2d_nhwc: tile %d needs the bf16 LDS-DMA path (Cin %% 32 == 0, aligned; 20+: 3x3 s1 p1)", tile); return FAR3D_ERR_ARG; }
  if (x_dt == FAR3D_DT_F32 && w_dt == FAR3D_DT_F32) { LAUNCH(float, float) }
  else if (x_dt == FAR3D_DT_F32 && w_dt == FAR3D_DT_F32_BF16X3) { LAUNCH(float, split_t) }
  else if (x_dt == FAR3D_DT_F32 && w_dt == FAR3D_DT_BF16) { LAUNCH(float, bf16_t) }
  else { LAUNCH(bf16_t, bf16_t) }
#undef LAUNCH
  FAR3D_CHECK_LAUNCH("far3d_conv2d_nhwc");
  return FAR3D_OK;
}
