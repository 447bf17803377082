/* far3d_hip.h -- C ABI of libfar3d_hip.so: the MI355X (gfx950) kernels of the Far3D inference hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every `const void*` / `float*` tensor argument is a DEVICE pointer
 *     to a contiguous buffer unless the parameter comment says "host";
 *   - `stream` is a hipStream_t (NULL = default stream); kernels are enqueued, never synchronised;
 *   - return 0 on success, negative FAR3D_ERR_* otherwise; far3d_last_error() gives the reason
 *     (thread-local, valid until the next failing call on the same thread);
 *   - dtype codes: FAR3D_DT_F32 = 0, FAR3D_DT_BF16 = 1 (raw bfloat16 bits).
 * Citations `ref:` are file:line under the reference checkout (projects/mmdet3d_plugin/...).
 */
#ifndef FAR3D_HIP_H
#define FAR3D_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAR3D_DT_F32 0
#define FAR3D_DT_BF16 1
/* weight-dtype code of far3d_conv2d_nhwc only: fp32 activations, products computed as a two-term bf16 split
 * (x = hi + lo, both round-to-nearest-even bf16; hi*hi' + hi*lo' + lo*hi' on the bf16 MFMA, fp32 accumulation): ~1e-5 relative
 * operand error at ~5x the fp32 matrix rate.  The "bf16x3" precision mode of the engine.  The weight buffer has the byte size
 * and row stride of the fp32 layout, but holds the split: every 32-channel block of a row (128 B) is 32 hi bf16 followed by
 * 32 lo bf16 (hi = bf16(w), lo = bf16(w - hi)). */
#define FAR3D_DT_F32_BF16X3 2
/* activation storage code ("pair"): a map of C logical channels (C % 32 == 0) holds fp32 values as 2C bf16 per pixel; every
 * 32-channel block is 32 hi = bf16(x) followed by 32 lo = bf16(x - hi) (hi + lo is exact in fp32 and carries 16 significant bits
 * at fp32's byte size).  Pointers, pixel strides and image strides of pair tensors count bf16 elements of the 2C-wide rows; channel
 * slices start at multiples of 32 logical channels.  Accepted by far3d_conv2d_nhwc (x / y / res, with w_dt = FAR3D_DT_F32_BF16X3:
 * the 64-byte hi / lo runs are what the LDS-DMA kernels stream, each product = 3 bf16 MFMAs), far3d_ese_nhwc,
 * far3d_groupnorm_nhwc, far3d_maxpool3x3s2_nhwc (dt) and far3d_stem_im2col (out_dt).  The "bf16x3" engine mode stores every
 * conv-stage activation this way. */
#define FAR3D_DT_BF16_PAIR 3
/* upper bound on the per-image partial-sum workgroups of far3d_ese_nhwc / far3d_groupnorm_nhwc (sizes their scratch) */
#define FAR3D_SUMS_MAX_PARTS 32
/* fraction bits of the fixed-point channel sums of far3d_conv2d_nhwc (chan_sums) */
#define FAR3D_SUMS_FRAC_BITS 18
/* partial sums [N][PARTS][C][2] + gates / group statistics [N][C] */
#define FAR3D_SUMS_SCRATCH_FLOATS(N, C) ((long)(N) * (C) * (2 * FAR3D_SUMS_MAX_PARTS + 1))

#define FAR3D_OK 0
#define FAR3D_ERR_ARG (-1)
#define FAR3D_ERR_LAUNCH (-2)
#define FAR3D_ERR_UNSUPPORTED (-3)

const char* far3d_last_error(void);
int far3d_abi_version(void);
int far3d_device_count(void);
int far3d_device_arch(int dev, char* buf, int buflen);

/* Multi-scale deformable attention, forward.
 * Replaces: mmcv-full 1.6.2 `ms_deform_attn_forward` as reached through
 *   MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
 *                                          sampling_locations, attention_weights, im2col_step)
 *   ref: models/utils/detr3d_transformer.py:561-563 (and models/utils/sparse_blocks.py:334-336).
 * value (bs,S,H,Dh) f32|bf16; spatial_shapes (L,2) int64 (h,w) DEVICE; level_start_index (L) int64 DEVICE;
 * sampling_loc (bs,Q,H,L,P,2) f32, (x,y) in [0,1]; attn_weight (bs,Q,H,L,P) f32; out (bs,Q,H*Dh) f32.
 * Semantics: out[b,q,h,:] = sum_l sum_p w * bilinear(value_l[b,:,h,:], x*W_l-0.5, y*H_l-0.5), zeros outside
 * (== grid_sample(align_corners=False, padding_mode='zeros')).  Dh must be a multiple of 4. */
int far3d_msda_forward(const void* value, int value_dtype, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* sampling_loc,
                       const float* attn_weight, float* out, int bs, int S, int H, int Dh, int L, int Q,
                       int P, void* stream);

/* Fused perspective-aware aggregation (3D deformable cross-attention sampling), forward, one sample.
 * Replaces, in one launch: DeformableFeatureAggregationCuda.feature_sampling
 *   ref: models/utils/detr3d_transformer.py:544-569 (projection :547-552, replication :555, MSDA :561,
 *   cross-camera sum :565-569), the key-point construction :524-525, and the softmax/permute of
 *   _get_weights :540-542.  Operator shape family follows the reference's own (unused) fused op
 *   ref: models/utils/deformable_aggregation.py:17-29.
 * feat (N,S,C=256) f32|bf16 token-major value maps; ref (A,3) f32 normalised reference points;
 * offsets (A,P,3) f32 = learnable_fc(x); lidar2img (N,4,4) f32 row-major;
 * U (A,L*P*G) f32 and Vc (N,L*P*G) f32 with logits[a,n,(l*P+p)*G+g] = U[a,.] + Vc[n,.]
 *   ( = weights_fc((x+pos)[a] + cam_embed[n]) split by linearity; bias lives in Vc );
 * level_hw (L,2) int32 HOST (h,w); level_start (L) int32 HOST; pc_range 6 floats HOST;
 * ldU / ldOffs: row strides in floats of U and offsets (0 = dense; U rows must stay 16-byte aligned) so both can be column
 * blocks of one merged-GEMM output.
 * pad_h/pad_w = img_metas['pad_shape'] (ref :551-552).  out (A,C) of dtype out_dt (f32 | bf16).  perm: optional (A) int32 DEVICE permutation giving
 * the order in which queries are assigned to workgroups (camera-sorted order keeps one XCD's L2 on 1-2 cameras); it never
 * changes results (row a of `out` is always query a); an entry ~a (negative) means "row a holds no query": a zero row is
 * written and nothing else is done (far3d_agg_order produces such entries).  With perm, A counts the ENTRIES of perm (workgroups);
 * the rows they name may lie anywhere in ref / offsets / U / out (a subset of a larger query set).  Requires C=256, G=8, L<=4, N<=16,
 * N*P<=256, N*P*L<=384.
 * cam_tables: the per-layer softmax factors of Vc from far3d_agg_tables / far3d_agg_order ((2+N)*L*P*G floats), or NULL.
 * variant: 0 = default: kernel 8 when cam_tables is given and N<=8, P<=16, value maps < 4 GiB, else kernel 7; 9 = kernel 8 with sibling
 * workgroups for heavy queries (below).
 *   8: two waves per query with specialised front ends (wave 0 projects and deals, wave 1 does the softmax statistics; two
 *      barriers), softmax factored into a query part (8 exp per lane) and the per-frame camera tables, (camera, level) items dealt to the waves by estimated work, bilinear taps merged per token row through the tent
 *      form of the interpolation weights (no atomics), padded LDS row lists, 16-byte row gathers from 32-bit byte offsets;
 *   7: round 2/3's kernel (two waves, levels split by parity, full 7-camera softmax from Vc, three barriers); 11 = 7 with its
 *      cross-lane reductions on the VALU and packed FMAs; 3 = round-1 kernel (workgroup per query, one gather per sample corner
 *      pair) -- 3, 7, 11 need Vc and are kept for A/B measurements; 12 = kernel 8 with round 4's greedy dealing of the items (kernel 8
 *      deals them by a contiguous split at the midpoints of the running work estimate since round 6; A/B).  All give the same result up
 *      to fp32 rounding. */
int far3d_aggregate_forward(const void* feat, int feat_dtype, const float* ref, const float* offsets,
                            const float* lidar2img, const float* U, const float* Vc, const float* cam_tables,
                            const int32_t* perm, void* out, int out_dt, int A,
                            int N, int S, int C, int G, int P, int L, const int32_t* level_hw,
                            const int32_t* level_start, const float* pc_range, float pad_h, float pad_w,
                            int ldU, int ldOffs, int variant, float* split_partials, int32_t* split_tickets, int split_extra,
                            const float* qbase, void* stream);
/* qbase != NULL (round 6, kernel 8 only, needs perm): SORTED mode.  U and offsets hold the row of the query of perm entry e AT ROW e
 * (launch order: their producers stored through far3d_agg_order's inv), qbase is far3d_agg_order's (A, 4) table and `ref` is not
 * read: every operand load of a workgroup depends on its block index only, perm[e] is needed for the hole test and the output row
 * alone.  out row a is query a as always, and the result is BIT-IDENTICAL to the unsorted call (same arithmetic in the same order).
 * Needs L * P <= 52 (the two hinted cameras' weight rows live in LDS).
 * The unsorted call costs a dependent round trip (perm[e] -> ref / offsets / U rows) before the first useful instruction. */
/* variant 13 (round 6, A/B measurement only): the sorted-mode kernel as TWO launches -- a list-build launch that writes every wave's row
 * list (offset + 8 weights per entry, <= 1024 entries per wave) and the softmax denominators to a global workspace, and a pure gather
 * launch (one workgroup per query) that reads them: VERDICT r5 item 3 (c).  Same entries in the same order (bit-identical on bf16 rows, 1 ulp on fp32 rows); measured 1.5 - 2.1x SLOWER than the fused launch.  Needs qbase,
 * perm, P = 13; split_partials = FAR3D_AGG_LISTS_FLOATS(A) floats (16-byte aligned), split_tickets = 2 * 8 * ceil(A / 8) int32. */
#define FAR3D_AGG_LISTS_FLOATS(A) ((long)(((A) + 7) / 8 * 8) * (2L * 1024 * 9 + 8))
/* variant 9 (round 5) = kernel 8 + SIBLING workgroups for heavy queries: the launch ends with its slowest wave, and the slowest waves
 * belong to the queries two cameras see.  far3d_agg_order(split_extra) marks those queries (flag 1 << 29 on their perm entry) and appends
 * one sibling entry per marked query (perm[A + e] = row | 1 << 30 | 1 << 29; unused slots 0x7fffffff); the launch runs split_extra extra
 * workgroups for them.  The two workgroups of a marked query deal its items into four shares instead of two, each publishes its
 * unnormalised 256-float sum in split_partials ([rows][2][256] f32) and draws a ticket from split_tickets ([rows] int32, ZERO at rest,
 * returned to zero); the second arrival adds the partials in part order and writes the row -- deterministic, run to run and graph vs
 * eager.  Unmarked queries take kernel 8's path unchanged.  The three split_* arguments are ignored by the other variants. */

/* Softmax factors of the aggregation logits' camera part, for `layers` decoder layers in one launch (block = layer):
 * Vc (layers, N, J) f32 (far3d_cam_embed_chain's output, J = L*P*G, J % 4 == 0) -> tables (layers, 2+N, J) f32 =
 * [ mV = max_n Vc | EV = sum_n eV[n] | eV[n] = exp(Vc[n] - mV) ].  softmax_{n,j}(U[a][j] + Vc[n][j]) (ref models/utils/
 * detr3d_transformer.py:539-540) = exp(U[a][j] + mV[j] - m_a) * eV[n][j] / sum_j exp(U[a][j] + mV[j] - m_a) EV[j]: the camera
 * factors depend on the layer and the frame only, the per-query part shrinks to 8 exp per lane. */
int far3d_agg_tables(const float* Vc, float* tables, int layers, int N, int J, void* stream);

/* Implicit-GEMM convolution / linear layer on the matrix cores (bf16 MFMA, or exact-fp32 MFMA when w_dt = F32).
 * Replaces the torch/cuDNN convolutions and nn.Linear GEMMs of the path:
 *   VoVNet conv+BN(eval)+ReLU  ref: models/backbones/vovnet.py:114-139,218-238 (BN folded into w/bias by the host),
 *   mmdet FPN laterals/outputs (ref cfg projects/configs/far3d.py:50-57; in-tree cousin models/necks/cp_fpn.py:156-208),
 *   YOLOX towers conv+BN+Swish ref: models/dense_heads/yolox_head.py:197-258, depth head convs
 *   ref: models/depth_predictor/depth_predictor.py:41-86, and every Linear of FarHead / the decoder
 *   (ref: models/dense_heads/farhead.py:228-282, models/utils/detr3d_transformer.py:503-512,525-540).
 * x: NHWC activations (x_dt), pixel stride ldx elements, image stride x_img_stride elements; the pointer is
 *    pre-offset to the first input channel (channel slices of a wider buffer are fine).
 * w: packed weights (w_dt) [R][KH*KW][ceil(Cin/32)*32] with R >= Cout + 256 rows (zero rows past Cout: channel tiles of up to
 *    256 rows over-read), zero padded in Cin; bias [R] f32 (16-byte aligned) or NULL.
 * y: NHWC output (y_dt), pixel stride ldy, image stride y_img_stride.  v = act(conv + bias) + res.
 * act: 0 none, 1 ReLU, 2 Swish.  res (optional, res_dt): NHWC Hr x Wr map added with nearest-neighbour upsampling
 *    (FPN top-down path); Hr=Ho, Wr=Wo gives a plain residual.
 * y2 (optional, y2_dt): second output y2 = y2_scale[n][m] * v + y2_shift[n][m] (FarHead's camera-aware MLN,
 *    ref: models/utils/misc.py:182-190, models/dense_heads/farhead.py:553-563) so that the FPN output conv writes
 *    the modulated token-major value maps directly.
 * tile: 0 auto (host callers pass the measured choice of far3d_amd/data/tuning_mi355x.json).  Any dtype: 1 128x128, 2 64x128,
 *    3 64x64, 4 128x64, 5 64x256 (channels x pixels).  bf16 with Cin % 32 == 0 only: 18, 43, 46, 48 (LDS-DMA ring variants, any
 *    kernel size / stride); 50-67, 90-97, 100-103 pipelined 3x3/s1/p1 kernel (channels x rows of 32 pixels, 4/8/16 waves, 2- or 3-deep
 *    weight ring, one kernel row or -- 100-103 -- all 9 taps per barrier step); 70-81 pipelined 1x1/s1 GEMM kernel.
 *    x_dt = FAR3D_DT_BF16_PAIR (w_dt must be FAR3D_DT_F32_BF16X3; y_dt pair or f32; res_dt any): 1-5 register-staged kernel (any
 *    kernel size / stride); 150-168, 191-197 the pipelined 3x3 shapes 50-68 / 91-97 with split products (169 / 190 / 198: 7-row
 *    forms); 170-181 the pipelined 1x1 shapes (185 / 186 128 x 160, 187 / 188 64 x 96 for the small maps); 252, 260, 265, 279, 280: the hi halves only (ONE bf16 product per term: a single-bf16 layer inside a pair-stored
 *    network).  An id the layer cannot use is an error, not a silent fallback.
 *    bf16, 1x1/s1 only: 82-89 the GEMM tiles with 3- / 4-deep LDS rings, 110-117 256 x 256 (and other large) tiles, 120-129 the
 *    GEMM with full-line LDS-DMA pieces (8 rows x 128 bytes per piece instead of 16 x 64), 140-145 split weight / activation rings;
 *    30-35 the 3x3 / stride 2 / pad 1 layers on the LDS-patch kernel (pair: 330, 331).
 *    x_dt = FAR3D_DT_F32 with w_dt = FAR3D_DT_F32_BF16X3, 1x1/s1, Cin % 32 == 0, x 16-byte aligned with strides that are multiples
 *    of 4 floats: 479-481 (auto) the pipelined GEMM kernel on fp32 rows -- 32 floats are the 128 bytes of a pair-stored block, so
 *    the LDS-DMA pattern is the pair kernel's and the hi / lo split of the rows happens in registers (same three products).
 *    x_dt = w_dt = FAR3D_DT_F32, 1x1/s1, Cin % 32 == 0, same alignment: 482-494 the pipelined GEMM kernel with the EXACT fp32 MFMA
 *    (v_mfma_f32_32x32x2_f32) on fp32 rows of both operands -- exact products, fp32 accumulation, another summation order than the
 *    register-staged kernel (tiles 1-5); 482 / 483 64 x 64 (2 / 4 LDS stages), 484 / 485 128 x 128, 486 64 x 128; 487-494 split K
 *    between 2-8 wave groups INSIDE the workgroup (each group its own LDS ring, partial tiles added in group order through LDS):
 *    487 / 488 / 494 64 x 64 (2 / 4 / 2 groups), 489 / 493 32 x 64 (4 / 2), 490 64 x 32 (4), 491 / 492 32 x 32 (4 / 8).  The bits
 *    of an output row depend on the tile and on Cin only, never on the number of rows in the call.
 *    3x3/s1/p1, Cin % 32 == 0, Cout % 32 == 0, same storage in and out, 16-byte aligned rows, NO res / y2 / chan_sums: the PERSISTENT
 *    wave-specialised kernel (csrc/conv_ws.hpp: producer waves issue every LDS-DMA, consumer waves only read LDS and run MFMAs, one
 *    workgroup per CU walks several tiles, 16-byte stores straight from the MFMA registers) -- pair storage 400-419 (one hand-over
 *    per tap), 450-459 (one per kernel row), 440 / 444 / 445 (LDS counters instead of the barrier; measured slower), bf16 420-423.
 *    Results are bit-identical to the pipelined 3x3 kernel (same products in the same order).
 *    1x1/s1 on pair-stored maps (x and y pair storage, Cin % 32 == 0, Cout % 32 == 0, 16-byte aligned rows, input below 2 GB, NO res /
 *    y2): 460-476 the persistent wave-specialised GEMM (gemm1x1_ws_kernel: the producers stream weights AND activation rows through one
 *    LDS ring; 460-463, 465, 469 128 x 128, 464 / 471 / 473 256 x 128, 470 / 476 128 x 256, 474 192 x 128, 466-468 64 x 128), bit-identical to
 *    the pipelined GEMM; chan_sums allowed when Ho*Wo >= the tile's pixels and Cin >= 64 x the tile's steps per hand-over.
 * chan_sums (optional, DEVICE int64 [N][Cout]; 1x1/s1 layers on a pipelined GEMM tile with a bf16 or pair output, Ho*Wo >= the
 *    tile's pixel count): every STORED output element v (for a pair output: its hi and its lo half) ADDS
 *    rint(v * 2^FAR3D_SUMS_FRAC_BITS) to chan_sums[n][channel] -- the global average pool of VoVNet's eSE block (ref
 *    models/backbones/vovnet.py:173-185) comes out of the concat convolution's epilogue instead of a second pass over its output.
 *    Integer sums are associative: the result does not depend on tile shapes, workgroup order or the number of images in the launch
 *    (bit-identical run to run, hipGraph vs eager, camera-sharded vs single rank) although they are accumulated with atomics.
 *    The caller provides zeros; far3d_ese_nhwc(chan_sums) consumes them and returns them to zero. */
int far3d_conv2d_nhwc(const void* x, int x_dt, const void* w, int w_dt, const float* bias, void* y, int y_dt,
                      int N, int H, int W, int Cin, int ldx, long x_img_stride, int Ho, int Wo, int Cout, int ldy,
                      long y_img_stride, int KH, int KW, int stride, int pad, int act, const void* res, int res_dt,
                      int ldr, long res_img_stride, int Hr, int Wr, void* y2, int y2_dt, int ldy2,
                      long y2_img_stride, const float* y2_scale, const float* y2_shift, long long* chan_sums, int tile,
                      void* stream);

/* Multi-head self-attention core: out = softmax(q k^T * scale) v per head (flash-style, no score tensor in HBM).
 * Replaces the bmm/softmax/bmm inside torch.nn.MultiheadAttention as wrapped by mmcv's MultiheadAttention
 *   (ref cfg projects/configs/far3d.py:112-116; call site models/utils/detr3d_transformer.py:385-394; in-tree
 *   statement of the wrapper models/utils/petr_transformer.py:286-326).  The in/out projections are
 *   far3d_conv2d_nhwc calls.
 * q (Aq, heads*32), k/v (Nk, heads*32) of `dtype` (F32 -> exact fp32 MFMA fed from registers, BF16 -> bf16 MFMA), row strides
 * ldq/ldk/ldv elements; out (Aq, heads*32) of dtype out_dt (f32 | bf16), row stride ldo.  head_dim must be 32.
 * The one mask of the inference path -- the "query hole" of the fixed-capacity proposal mode (see FAR3D "hole" below): with
 * hole_count != NULL (DEVICE int32) the keys [hole_start + *hole_count, hole_end) are excluded from every softmax; their K / V
 * rows must hold finite numbers.  hole_count NULL: no mask. */
int far3d_attention_forward(const void* q, const void* k, const void* v, int dtype, void* out, int out_dt, int Aq, int Nk,
                            int heads, int head_dim, int ldq, int ldk, int ldv, int ldo, float scale,
                            const int32_t* hole_count, int hole_start, int hole_end, void* stream);

/* A/B switch for the fp32 instantiation far3d_attention_forward launches (process-wide; tools/probe/attn_f32_ab.py): 0 = the default
 * (register-fed kernel, 64-query workgroups x 4 key parts), 14 / 118 other shapes of it, 4 / 2 the LDS-staged kernel of earlier rounds.
 * Returns the previous value; variant < 0 only reads.  Every variant computes the same products in fp32 (the summation order and
 * the exp differ at the last bit). */
int far3d_attention_f32_variant(int variant);

/* y = act(LayerNorm_C(x) * gamma + beta); optional y2 = y + add (next GEMM's "query + query_pos").
 * Replaces nn.LayerNorm at ref models/utils/detr3d_transformer.py:304-307,398-400,506-512 and
 * models/dense_heads/farhead.py:230-239,274-277.  x,y,add,y2: f32 rows with strides ldx,ldy,lda,ldy2 (multiples of 4);
 * gamma/beta may be NULL (no affine).  act: 0 none, 1 ReLU.  C multiple of 4, <= 1024.  y2 has dtype y2_dt (f32 | bf16);
 * yb (optional) receives a copy of y in dtype yb_dt (f32 | bf16; row stride ldyb) -- the next GEMM's operands without a
 * conversion pass; y2 and yb may be the two halves of one [y+add | y] row (merged-GEMM input). */
int far3d_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, int ldx,
                    int ldy, float eps, int act, const float* add, int lda, void* y2, int ldy2, int y2_dt, void* yb, int ldyb,
                    int yb_dt, void* stream);
/* far3d_layernorm with a ROW MAP for the two GEMM-operand outputs: row i of y2 / yb is stored at row out_rows[i] (y itself stays in
 * place).  The exact-fp32 decoder uses it to hand the aggregation's logit / offset GEMM its operand rows in the aggregation kernel's
 * launch order (far3d_agg_order's inv), so that the GEMM's output is in that order with no change to the GEMM.  out_rows (rows) int32
 * DEVICE, a permutation of [0, rows) (or any injective map into the caller's buffers). */
int far3d_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, int ldx,
                         int ldy, float eps, int act, const float* add, int lda, void* y2, int ldy2, int y2_dt, void* yb, int ldyb,
                         int yb_dt, const int32_t* out_rows, void* stream);

/* ROW-RESIDENT CHAINS of a decoder layer (bf16 decoder, embed dims 256, FFN hidden 1024): the row-local work between the
 * attention core and the aggregation kernel, and between the aggregation kernel and the next layer's attention core, in one
 * launch each -- a workgroup owns 16 query rows, keeps them in LDS across projections / residuals / LayerNorms and streams the
 * weights from L2 in MFMA fragment order.  They replace sequences of far3d_conv2d_nhwc + far3d_layernorm calls (the default
 * path of far3d_amd.engine, which stays available) and compute the same arithmetic: bf16 operands, fp32 accumulation, epilogue
 * order bias -> activation -> residual, far3d_layernorm's reduction tree; only the K order of the fp32 accumulation differs.
 * Rows never interact: a row's result does not depend on M or on the rows launched with it.
 * Packed weights: a (N, K) bf16 row-major weight (K % 256 == 0, rows zero-padded to a multiple of 16) re-ordered to
 * [N / 16][K / 32][64 lanes][8]: element j of lane l of (tile t, step s) is W[16 t + (l & 15)][32 s + 8 (l >> 4) + j]
 * (far3d_amd.ops.pack_rowchain).  Biases are f32 and cover the padded rows.  All pointers 16-byte aligned, row strides in
 * elements (bf16 rows: multiples of 8, f32 rows: multiples of 4).
 *
 * far3d_rowchain_attn_out: x1 = LN0(att W_out^T + b_out + x);  ul = [x1 + qpos | x1] W_wl^T + b_wl.
 *   Replaces the self-attention output projection + residual + norm (ref models/utils/detr3d_transformer.py:385-400) and the
 *   attention-weight / key-point-offset linears of the aggregation (:522-531; merged into one (n_wl, 512) weight whose first
 *   256 input columns see query + query_pos and whose last 256 see the query, as in the unfused path).
 *   att (M,256) bf16; x, qpos (M,256) f32; x1 (M,256) f32 out; ul (M, n_wl) f32 out, 448 < n_wl <= 464. */
int far3d_rowchain_attn_out(const void* att, int ld_att, const float* x, int ldx, const float* qpos, int ldq,
                            const void* w_out, const float* b_out, const float* g0, const float* be0,
                            const void* w_wl, const float* b_wl, int n_wl, float* x1, int ldx1, float* ul, int ldu,
                            const int32_t* ul_rows, int M, float eps, void* stream);
/*   ul_rows: optional (M) int32 DEVICE -- row i of the logits / offsets is stored at row ul_rows[i] of `ul` (far3d_agg_order's inv:
 *   the aggregation kernel's launch order); NULL = row i. */

/* far3d_rowchain_ffn: x2 = LN1(agg W_o^T + b_o + x1);  out = LN2(relu(x2 W_1^T + b_1) W_2^T + b_2 + x2);
 *   with w_qkv: qkv = [out + qpos | out] W_qkv^T + b_qkv (bf16; the NEXT layer's merged q / k / v in-projection).
 *   Replaces the aggregation's output projection + residual (ref detr3d_transformer.py:566-569), norm, FFN, norm (:398-422)
 *   and the next layer's in-projection (:378-384).  agg (M,256) bf16; x1, qpos (M,256) f32; out (M,256) f32;
 *   qkv (M,768) bf16 or NULL with w_qkv NULL; xop (optional, M x 512 bf16): [out + qpos | out], the operand the unfused
 *   in-projection reads. */
int far3d_rowchain_ffn(const void* agg, int ld_agg, const float* x1, int ldx1, const float* qpos, int ldq,
                       const void* w_o, const float* b_o, const float* g1, const float* be1,
                       const void* w_1, const float* b_1, const void* w_2, const float* b_2,
                       const float* g2, const float* be2, const void* w_qkv, const float* b_qkv,
                       float* out, int ldo, void* qkv, int ldqkv, void* xop, int ldxop, int M, float eps, void* stream);

/* far3d_rowchain_qkv: qkv = [x + qpos | x] W_qkv^T + b_qkv (bf16), the in-projection alone -- the operand is built and the GEMM
 *   run by the code of far3d_rowchain_ffn's tail, so for the same x (that kernel's `out`) and qpos the result is BIT-IDENTICAL
 *   to the tail's.  For the query-sharded decoder, where a layer's output rows exist on every rank only after the exchange.
 *   x, qpos (M,256) f32; qkv (M,768) bf16. */
int far3d_rowchain_qkv(const float* x, int ldx, const float* qpos, int ldq, const void* w_qkv, const float* b_qkv,
                       void* qkv, int ldqkv, int M, void* stream);

/* far3d_rowchain_branches: the shared classification and regression branches over the decoder outputs of all layers,
 *   cls = L2(relu(LN(L1(relu(LN(L0(h)))))));  reg = L2'(relu(L1'(relu(L0'(h)))))   (L: 256 -> 256 linears, the last ones -> n_cls / n_reg)
 *   Replaces 6 far3d_conv2d_nhwc + 2 far3d_layernorm launches (ref models/dense_heads/farhead.py:230-245, applied at :646-664).
 *   h (M,256) bf16; cls (M, n_cls) f32, reg (M, n_reg) f32, n_cls, n_reg <= 32; weights packed as above (the last linears as TWO
 *   16-row tiles, zero rows past their outputs), biases covering 32 rows. */
int far3d_rowchain_branches(const void* h, int ldh, const void* w_c0, const float* b_c0, const float* g_c0, const float* be_c0,
                            const void* w_c1, const float* b_c1, const float* g_c1, const float* be_c1,
                            const void* w_c2, const float* b_c2, int n_cls, const void* w_r0, const float* b_r0,
                            const void* w_r1, const float* b_r1, const void* w_r2, const float* b_r2, int n_reg,
                            float* cls, int ld_cls, float* reg, int ld_reg, int M, float eps, void* stream);

/* VoVNet eSE block on NHWC maps: y = x * hsigmoid(fc(mean_hw(x))) (+ identity).
 * Replaces eSEModule + the identity add, ref models/backbones/vovnet.py:173-185,232-236.
 * x,identity,y: NHWC `dt` with (pixel stride, image stride) pairs; fcw (C,C) f32 row-major, fcb (C) f32;
 * scratch: device workspace of FAR3D_SUMS_SCRATCH_FLOATS(N, C) floats; never needs zeroing.  The average pool writes
 * per-workgroup partial channel sums with plain stores and the gate kernel adds them in index order, so the sums are
 * deterministic: no atomics, bit-identical run to run and hipGraph vs eager.  C a multiple of 4, <= 1024.
 * chan_sums (optional, bf16 / pair maps): the fixed-point channel sums [N][C] that far3d_conv2d_nhwc accumulated while it wrote x;
 * the pooling pass is skipped, the gate is computed from them and the apply kernel returns them to zero for the next producer. */
int far3d_ese_nhwc(const void* x, int dt, const float* fcw, const float* fcb, const void* identity, void* y,
                   float* scratch, int N, int HW, int C, int ldx, long x_img_stride, int ldi, long i_img_stride,
                   int ldy, long y_img_stride, long long* chan_sums, void* stream);

/* The same block for maps whose channel sums came out of the concat convolution (chan_sums REQUIRED; dt bf16 or pair), with the stage-end
 * pooling of ref models/backbones/vovnet.py:249-250 fused into the apply pass: the gate launch of far3d_ese_nhwc, then ONE streaming launch
 * that writes y = x * gate (+ identity) in 16-byte pieces and, pooled != NULL, MaxPool2d(3, 2, ceil_mode=True)(y) ((Hp, Wp) its ceil-mode
 * size) computed from x with every tap rounded to the storage format first -- bit-identical to far3d_maxpool3x3s2_nhwc of the stored y.
 * y == NULL: only the pooled map is written (the caller needs the stage output at half resolution only).
 * gate: N*C floats of workspace (16-byte aligned); chan_sums come back zeroed.
 * C % 8 == 0 (pair: % 32), <= 1024; every pointer / stride a multiple of 16 bytes. */
int far3d_ese_fused_nhwc(const void* x, int dt, const float* fcw, const float* fcb, const void* identity, void* y, void* pooled,
                         float* gate, int N, int H, int W, int C, int ldx, long x_img_stride, int ldi,
                         long i_img_stride, int ldy, long y_img_stride, int Hp, int Wp, int ldp, long p_img_stride,
                         long long* chan_sums, void* stream);

/* GroupNorm(groups, C) (+ReLU) on dense NHWC maps.  Replaces nn.GroupNorm(32, 256)+ReLU of the depth head,
 * ref models/depth_predictor/depth_predictor.py:43-45.  scratch: FAR3D_SUMS_SCRATCH_FLOATS(N, C) floats (no zeroing needed);
 * statistics are reduced deterministically from per-workgroup partial sums. */
int far3d_groupnorm_nhwc(const void* x, int dt, const float* gamma, const float* beta, void* y, float* scratch, int N,
                         int HW, int C, int groups, float eps, int relu, void* stream);

/* MaxPool2d(kernel 3, stride 2, ceil_mode=True), dense NHWC input, output with pixel stride ldy / image stride
 * y_img_stride (so it can land in the first C channels of the next OSA concat buffer)
 * (ref models/backbones/vovnet.py:249-250). */
int far3d_maxpool3x3s2_nhwc(const void* x, int dt, void* y, int N, int H, int W, int C, int Ho, int Wo, int ldy,
                            long y_img_stride, void* stream);

/* NCHW fp32 image (N,3,H,W) -> NHWC (N,Ho,Wo,32) im2col of the stride-2 3x3 stem conv: channel = (ky*3+kx)*3 + c for the
 * 27 taps, 5 zero channels; Ho = (H-1)/2+1.  The first VoVNet conv (ref models/backbones/vovnet.py:306-311) then runs as
 * a K=32 far3d_conv2d_nhwc 1x1. */
int far3d_stem_im2col(const float* img, void* out, int out_dt, int N, int H, int W, void* stream);

/* The same convolution in ONE launch for bf16 engines: y = act(conv3x3 / stride 2 / pad 1 (img) + bias) with the folded-BN weights of
 * VoVNet's first stem layer, read straight from the NCHW fp32 image -- the im2col map is never written.  w: bf16 [>= 64][32] rows of the
 * layer packed like far3d_stem_im2col's consumer (k = (ky*3 + kx)*3 + c, 5 zero columns), bias [64] f32 or NULL; y: bf16 NHWC
 * (N, Ho, Wo, 64 channels) with pixel stride ldy, Ho = (H-1)/2+1, Wo = (W-1)/2+1; act 0 none / 1 ReLU.  Same products in the same order as
 * far3d_stem_im2col + far3d_conv2d_nhwc (bit-identical output).  Replaces ref models/backbones/vovnet.py:306-311 (stem_1). */
int far3d_stem_conv(const float* img, const void* w, const float* bias, void* y, int N, int H, int W, int ldy, long y_img_stride,
                    int act, void* stream);

/* 2D proposal scoring + ordered fixed-capacity selection (no host sync).
 * Replaces YOLOXHeadCustom.get_bboxes' score / 3x3-peak / threshold steps, ref models/dense_heads/yolox_head.py:426-438,
 * and its boolean-mask indexing :452-467.  cls[l] (N,h,w,ncls) f32 logits, reg[l] (N,h,w,nreg) f32 with channels
 * (dx,dy,log w,log h,objectness) -- HOST arrays of L device pointers.  level_hw (L,2), strides (L) int32 HOST.
 * scratch_sw, weights: (N,S) f32 device; weights receives sample_weight*(sample_weight==maxpool3x3).
 * mode 0: keep weights > thr (first `cap` in flat-index order); mode 1: the cap largest per camera (ties -> lower index).
 * sel_idx (N,cap) int32 ascending flat indices (camera-local, level-major), sel_cnt (N) int32. */
int far3d_proposal_select(const float* const* cls, const float* const* reg, int ncls, int nreg, int N, int L,
                          const int32_t* level_hw, const int32_t* strides, float* scratch_sw, float* weights, int* sel_idx,
                          int* sel_cnt, int cap, float thr, int mode, void* stream);

/* Adaptive-query construction from the selected 2D peaks: 2D box decode (ref yolox_head.py:491-501 + xyxy->cxcywh :457),
 * depth bin argmax at round(centre/stride) (ref models/dense_heads/farhead.py:736-747), LID un-binning (:521-527),
 * un-projection with img2lidar = inverse(lidar2img) and pc_range normalisation (:792-811), context = value-map token ||
 * log-odds(score) - log-odds(thr) (:576-581,773-784; the score is clamped to >= 1e-6 first so that zero-weight padding rows
 * of the static top-K mode stay finite).  Camera n's rows start at sum(sel_cnt[:n]) (camera-major, like the reference).
 * Outputs: ref2d (M,3), ctx (M,C+1), box2d (M,4 cxcywh), score (M).
 * FIXED-CAPACITY MODE (rows_total > 0; the reference's data-dependent M = sum(sel_cnt), ref yolox_head.py:429-458, farhead.py:
 * 576-581, without a host sync and with static shapes -- "cap + count"): the outputs have rows_total rows; rows [0, M) are
 * the proposals in the reference's order, rows [M, rows_total) are zero-filled and form the HOLE that the consumers
 * mask: *m_out = min(M, rows_total) (DEVICE int32; the `hole_count` of far3d_attention_forward / far3d_head_finalize /
 * far3d_agg_order), *overflow_out = 1 when proposals were dropped (M > rows_total) or a camera filled its `cap` (peaks may have
 * been lost in far3d_proposal_select), else 0 -- the host checks it after the fact instead of syncing on M.
 * rows_total = 0: legacy behaviour (rows past M untouched; m_out / overflow_out must be NULL). */
int far3d_proposal_gather(const float* const* reg, int nreg, int N, int L, const int32_t* level_hw, const int32_t* strides,
                          const int* sel_idx, const int* sel_cnt, int cap, const float* weights,
                          const float* depth_logit, int hd, int wd, int nd, int depth_stride, float depth_min,
                          float depth_max, int depth_bins, const float* img2lidar, const void* feat, int feat_dt, int C,
                          const float* pc_range, float score_thr, float* ref2d, float* ctx, float* box2d, float* score,
                          int rows_total, int32_t* m_out, int32_t* overflow_out, void* stream);

/* Blocks of rows -> one compact run (camera-sharded fixed-capacity mode): src (nblocks, rows_per_block, D) f32 of which the first
 * counts[b] rows of block b are valid; dst (dst_rows, D): rows [0, M) the valid rows in block order, the rest zero,
 * *m_out = M = min(sum counts, dst_rows); *overflow_out is OR-ed with (sum counts > dst_rows) (initialise it). */
int far3d_compact_rows(const float* src, const int32_t* counts, int nblocks, int rows_per_block, int D, float* dst, int dst_rows,
                       int32_t* m_out, int32_t* overflow_out, void* stream);

/* y[r] = gamma[r] * LN_noaffine(x[r]) + beta[r] (+ add[r]);  C = 256.  ldg / lda = 0 broadcast one row.  do_ln = 0 skips
 * the normalisation (MLN with use_ln=False).  Replaces MLN.forward, ref models/utils/misc.py:182-190, at
 * models/dense_heads/farhead.py:292-303. */
int far3d_row_affine_ln(const float* x, const float* gamma, const float* beta, const float* add, float* y, int rows, int C,
                        int ldx, int ldg, int lda, int ldy, float eps, int do_ln, void* stream);

/* --- fused FarHead bookkeeping kernels (far3d_amd/csrc/glue.hip); each replaces a chain of tiny tensor ops ------------- */

/* pos2posemb3d, ref models/utils/positional_encoding.py:13-25.  pos (R,3) f32, dim_t128 (128) f32 = 1e4^(2*floor(i/2)/128),
 * out (R,384) f32 = cat(emb(y), emb(x), emb(z)). */
int far3d_posemb3d(const float* pos, const float* dim_t128, float* out, int R, void* stream);

/* Streaming-memory pre-update + temporal codes, ref models/dense_heads/farhead.py:453-477,287,297-303.
 * state: emb (L,E) ref (L,3) ts (L) f64 pose (L,4,4) velo (L,2); ego_pose_inv (4,4); timestamp (1) f64 DEVICE;
 * pseudo_ref (P,3) normalised; prev_exists 0/1; fresh != 0: the state is the all-zero first-frame memory (no ego warp).
 * out: the pre-updated memory m_*, temp_ref (L,3) normalised, nerf (L,180) f32, tpos (L,256) f32 (pos2posemb1d in f64). */
int far3d_memory_prepare(const float* emb, const float* ref, const double* ts, const float* pose, const float* velo,
                         const float* ego_pose_inv, const double* timestamp, const float* pseudo_ref, const float* dim_t256,
                         float prev_exists, int fresh, const float* pc_range, int L, int E, int P, float* m_emb, float* m_ref,
                         double* m_ts, float* m_pose, float* m_velo, float* temp_ref, float* nerf, float* tpos, void* stream);

/* Box-code finalisation of the shared reg branch + memory scores, ref models/dense_heads/farhead.py:649-664,490.
 * reg (layers*A, code) f32, ref (A,3), cls_all (layers,A,ncls) logits (may be NULL with score NULL); score (A) from the last
 * layer.  hole_count != NULL (DEVICE int32): the query rows [hole_start + *hole_count, hole_end) hold no query (fixed-capacity
 * proposal mode): their score and their logits in EVERY layer are set to -inf, so neither the memory top-k nor the decode can pick
 * them.  hole_count NULL: cls_all is only read. */
int far3d_head_finalize(const float* reg, const float* ref, float* cls_all, float* box, float* score, int layers, int A,
                        int code_size, int num_classes, const float* pc_range, const int32_t* hole_count, int hole_start, int hole_end,
                        void* stream);

/* Streaming-memory post-update, ref models/dense_heads/farhead.py:479-508: the K top-scoring queries (topk_idx, int64
 * DEVICE) are pushed in front, the queue is truncated to L and warped by ego_pose; writes the persistent state in place. */
int far3d_memory_post_update(const float* m_emb, const float* m_ref, const double* m_ts, const float* m_pose, const float* m_velo,
                             const int64_t* topk_idx, const float* dec_last, const float* box_last, const float* ego_pose,
                             const double* timestamp, int L, int E, int K, int code_size, float* emb, float* ref, double* ts,
                             float* pose, float* velo, void* stream);

/* out_sum = a + b (sum_dt), out_a = a (a_dt, optional); a, b dense (rows, C) f32, outputs with row strides ld_sum / ld_a elements
 * (multiples of 4): the decoder's [query + query_pos | query] merged-GEMM operand in one pass. */
int far3d_add_cast(const float* a, const float* b, void* out_sum, int sum_dt, void* out_a, int a_dt, int rows, int C, long ld_sum,
                   long ld_a, void* stream);

/* Query order for far3d_aggregate_forward's `perm`: queries sorted by (nearest camera, 8x8 image cell) of their reference
 * point, one single-workgroup launch (keys + LDS counting sort).  A <= 8192, N <= 16.  Scheduling only -- except that with
 * hole_count != NULL the rows [hole_start + *hole_count, hole_end) (no query there: fixed-capacity proposal mode) are entered as
 * ~a (negative): far3d_aggregate_forward writes a zero row for such an entry and does no work.
 * row_base: order the A rows [row_base, row_base + A) of `ref` (a rank's share of the queries in the query-sharded decoder); the
 * entries of perm are ABSOLUTE row indices (the hole is given in absolute rows too), so far3d_aggregate_forward is then called
 * with the full-size ref / offsets / U / out buffers, A = the share's row count and this perm.
 * tables != NULL: the same launch also computes far3d_agg_tables(Vc, tables, layers, N, J) in `layers` extra workgroups (both are
 * per-frame preparations of the aggregation; one launch instead of two). */
int far3d_agg_order(const float* ref, const float* lidar2img, int32_t* perm, int A, int N, const float* pc_range, float pad_h,
                    float pad_w, const int32_t* hole_count, int hole_start, int hole_end, int row_base,
                    const float* Vc, float* tables, int layers, int J, int split_extra, int32_t* inv, float* qbase, void* stream);
/* inv / qbase (both optional, round 6): the operands of far3d_aggregate_forward's SORTED mode.  inv (A) int32: inv[i] = the perm slot
 * (= workgroup) that handles row row_base + i, i.e. perm[inv[i]] names row row_base + i; the producers of the per-layer logits /
 * key-point offsets store their row i at row inv[i] (far3d_rowchain_attn_out's ul_rows, far3d_layernorm_rows).  qbase (A, 4) f32,
 * 16-byte aligned: qbase[slot] = (X, Y, Z, hint) -- the reference point of the query in that slot in metres (ref
 * models/utils/detr3d_transformer.py:524-525) and, as an int32 bit pattern, hint = cam0 | cam1 << 8: the two cameras the reference point
 * projects closest to.  The aggregation kernel forms the softmax weights against those two cameras ahead of its item loop (an item of
 * any other camera forms its own: the hint never changes a result).
 * split_extra > 0 (far3d_aggregate_forward variant 9): perm has A + split_extra entries; the first split_extra queries IN ROW ORDER whose
 * reference point projects into two or more cameras are marked and get a sibling entry behind the A main ones (see above). */

/* Descending top-K of n <= 40960 floats (ties -> lower index), one workgroup: idx_out (K) int64, val_out (K) f32 or NULL.
 * Replaces torch.topk in post_update_memory, ref models/dense_heads/farhead.py:488-491 (K <= 1024). */
int far3d_topk(const float* vals, int n, int K, int64_t* idx_out, float* val_out, void* stream);

/* NMS-free box decode of the last decoder layer in ONE launch.  Replaces NMSFreeCoder.decode_single
 * (ref core/bbox/coders/nms_free_coder.py:39-91), denormalize_bbox (ref core/bbox/util.py:25-52) and the z shift of
 * FarHead.get_bboxes (ref models/dense_heads/farhead.py:1236-1238).  cls_last (A,num_classes) logits, box_last (A,code_size)
 * = (cx,cy,cz,log w,log l,log h,sin,cos[,vx,vy]); post_center_range 6 floats HOST.  Outputs, sorted by descending score:
 * boxes (K,code_size-1) = (cx,cy,cz-h/2,w,l,h,atan2(sin,cos)[,vx,vy]), scores (K) = sigmoid, labels (K) int64,
 * keep (K) uint8 = centre (before the z shift) inside post_center_range (the reference drops the others).
 * K <= 1024.  A*num_classes <= 40960: one launch, workspace may be NULL.  Larger inputs (the reference's threshold proposal mode
 * on a busy frame): every 40960-logit chunk is ranked by its own workgroup and a second launch ranks the chunks' K best; that needs
 * `workspace` = FAR3D_DECODE_WS_BYTES(A*num_classes, K) bytes of 8-byte aligned DEVICE memory (contents irrelevant) and
 * ceil(A*num_classes / 40960) * K <= 40960.  Ties: the lower flat index (query-major, then class) first, in both paths. */
#define FAR3D_DECODE_WS_BYTES(n, K) ((((long)(n) + 40959) / 40960) * (long)(K) * 8)
int far3d_decode_topk(const float* cls_last, const float* box_last, int A, int num_classes, int code_size, int K,
                      const float* post_center_range, float* boxes, float* scores, int64_t* labels, unsigned char* keep,
                      void* workspace, long workspace_bytes, void* stream);
/* far3d_decode_topk + far3d_topk(mem_scores, mem_n, mem_K, mem_idx_out) in ONE launch of two workgroups (round 6): the decode and the
 * memory update's top-k (ref models/dense_heads/farhead.py:488-491) both rank outputs of the last layer and do not depend on each
 * other.  Same results as the two calls.  Shapes outside the fused launch (A*num_classes > 40960 or mem_n > 4096) run as the two calls. */
int far3d_decode_topk_mem(const float* cls_last, const float* box_last, int A, int num_classes, int code_size, int K,
                          const float* post_center_range, float* boxes, float* scores, int64_t* labels, unsigned char* keep,
                          void* workspace, long workspace_bytes, const float* mem_scores, int mem_n, int mem_K,
                          int64_t* mem_idx_out, void* stream);

/* Per-frame camera calibration in one launch: img2lidar (N,4,4) = inverse(lidar2img) (ref models/dense_heads/farhead.py:798;
 * Gauss-Jordan with partial pivoting in f64) and c14 (N,14) = [fx/1e3, fy/1e3, extrinsics[:3,:4]] (ref farhead.py:553-556).
 * Either output may be NULL. */
int far3d_camera_prep(const float* lidar2img, const float* intrinsics, const float* extrinsics, float* img2lidar, float* c14,
                      int N, void* stream);

/* torch.nan_to_num (nan -> 0, +-inf -> +-FLT_MAX) in place on n floats (multiple of 4), ref models/dense_heads/farhead.py:646;
 * bf16_copy (optional) receives the sanitised values as bf16. */
int far3d_nan_to_num(float* x, void* bf16_copy, long n, void* stream);

/* Camera embedding chain of ALL decoder layers in one launch (ref models/utils/detr3d_transformer.py:497-505, 531-538):
 * out[l][n][:] = W3[l] LN(ReLU(W2[l] ReLU(W0[l] l2i[n] + b0[l]) + b2[l])) + b3[l]   (cam_embed -> camera term of weights_fc).
 * l2i: row n holds lidar2img[n][:3,:] flattened in its first 12 floats, row stride ld_l2i (16 = the (N,4,4) tensor in place); weights fp32, TRANSPOSED to [in][out]: w0t [L][12][Hd], w2t [L][Hd][256],
 * w3t [L][256][J]; b0 [L][Hd], b2/ln_g/ln_b [L][256], b3 [L][J]; out [L][N][J].  Embedding width 256, hidden Hd <= 256. */
int far3d_cam_embed_chain(const float* l2i, const float* w0t, const float* b0, const float* w2t, const float* b2,
                          const float* ln_g, const float* ln_b, const float* w3t, const float* b3, float* out, int N, int L,
                          int J, int Hd, float eps, int ld_l2i, void* stream);

/* --- device image pre-processing (far3d_amd/csrc/preproc.hip) ------------------------------------
 * Pillow-exact 8-bit separable resampling (Image.resize as used by AV2ResizeCropFlipRotImageV2._img_transform,
 * ref datasets/pipelines/custom_pipeline.py:277-311) restricted to the crop window, then crop / flip / normalise / pad / HWC->CHW
 * (ref datasets/pipelines/transform_3d.py:89-101, custom_pipeline.py:358-378).  bounds (out_size,2) and coeffs (out_size,ksize)
 * int32 DEVICE: per output position the first input sample, the sample count and the 22-bit fixed-point filter weights, built on
 * the host exactly like Pillow's precompute_coeffs + normalize_coeffs_8bpc (far3d_amd/data_pipeline/resample.py).
 *
 * far3d_image_resample_h: horizontal pass.  src u8 HWC (in_h,in_w,3), row pitch src_pitch bytes; dst u8 (rows,outw,3) = output
 *   columns [x0, x0+outw) of input rows [row0, row0+rows).
 * far3d_image_resample_v: vertical pass over that intermediate for output rows [y0, y0+outh), optional horizontal flip, then
 *   mode 0: u8 HWC (outh,outw,3) (an intermediate image: the portrait camera is resized twice, custom_pipeline.py:71-92), or
 *   mode 1: out (3,pad_h,pad_w) f32|bf16 = (pixel - mean[c]) * stdinv[c] inside the crop, 0 in the padding; mean / stdinv 3 floats
 *   HOST; to_rgb swaps channels 0 and 2 first. */
int far3d_image_resample_h(const unsigned char* src, long src_pitch, int in_h, int in_w, unsigned char* dst, const int32_t* bounds,
                           const int32_t* coeffs, int ksize, int row0, int rows, int x0, int outw, void* stream);
int far3d_image_resample_v(const unsigned char* tmp, int row0, int rows, int outw, const int32_t* bounds, const int32_t* coeffs,
                           int ksize, int y0, int outh, int flip, int mode, void* out, int out_dt, int pad_h, int pad_w,
                           const float* mean, const float* stdinv, int to_rgb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FAR3D_HIP_H */
