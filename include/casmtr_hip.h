/*
 * casmtr_hip.h -- C ABI of libcasmtr_hip.so: CasMTR's cascaded-matching hot path on MI355X (gfx950).
 *
 * Plain pointers and sizes only (no torch types).  Every pointer is a DEVICE pointer to a dense row-major
 * tensor; fp32 values, int64 indices (what the reference's extensions take: packed_accessor32<long,...>,
 * score_computation_kernal.cu:24-26).  `stream` is a hipStream_t.  Return value: 0 on success, a hipError_t
 * code, or CASMTR_ERR_UNSUPPORTED (1001) when a shape is outside what the kernel is built for -- callers must
 * surface that as an error, never route around it on the CPU.  All launches are asynchronous on `stream`.
 *
 * Citations are relative to the reference checkout (ewrfcas/CasMTR).
 */
#ifndef CASMTR_HIP_H
#define CASMTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* casmtr_stream_t; /* hipStream_t */

#define CASMTR_ABI_VERSION 8
int casmtr_abi_version(void);
/* Test hook (round 6): the kernels with a dynamic item schedule claim work from per-XCD counters that every launch leaves zeroed;
 * synchronises the device and returns the number of non-zero counter words (0 = consistent; < 0: HIP error).                     */
int casmtr_debug_work_counters_nonzero(void);

/* ------------------------------------------------------------------------------------------------------------
 * Drop-in primitives: one entry point per pybind function of the reference's three extensions.
 * ---------------------------------------------------------------------------------------------------------- */

/* score_computation_cuda.score_forward      (score_computation.cpp:11-20,36 ; kernel score_computation_kernal.cu:21-92)
 *   q [B,N1,4,H,D], key [B,N2,H,D], idx [B,N1,K,H] -> out [B,N1,4,K,H] = sum_d q*key[idx]                      */
int casmtr_qta_score_fwd(const float* q, const float* key, const int64_t* idx, float* out,
                         int B, int N1, int N2, int K, int H, int D, casmtr_stream_t stream);
/* score_computation_cuda.score_backward     (score_computation.cpp:22-33,37 ; kernel :94-184)
 *   grad [B,N1,4,K,H] -> dq [B,N1,4,H,D] (overwritten), dkey [B,N2,H,D] (zeroed, then accumulated)             */
int casmtr_qta_score_bwd(const float* grad, const float* q, const float* key, const int64_t* idx,
                         float* dq, float* dkey, int B, int N1, int N2, int K, int H, int D, casmtr_stream_t stream);

/* value_aggregation_cuda.value_aggregation_forward  (value_aggregation.cpp:9-31,63 ; kernel value_aggregation_kernel.cu:21-53)
 *   score [B,N,K,H], value [B,M,H,D], idx [B,N,K,H] -> out [B,N,H,D] (caller-allocated, overwritten)           */
int casmtr_qta_value_agg_fwd(const float* score, const float* value, const int64_t* idx, float* out,
                             int B, int N, int K, int H, int M, int D, casmtr_stream_t stream);
/* value_aggregation_cuda.value_aggregation_backward (value_aggregation.cpp:33-60,64 ; kernel :55-86)
 *   grad_score [B,N,K,H] overwritten; grad_value [B,M,H,D] ACCUMULATED into (caller zero-initialises it)       */
int casmtr_qta_value_agg_bwd(const float* grad_out, const float* score, const float* value, const int64_t* idx,
                             float* grad_score, float* grad_value, int B, int N, int K, int H, int M, int D,
                             casmtr_stream_t stream);

/* fast_score_computation.score_forward      (score_cuda/src/score_computation.cpp:8-17,30 ; kernel score_computation_kernel.cu:22-65)
 *   q [B,N1,C], key [B,N2,C], idx [B,N1,K] -> out [B,N1,K]                                                     */
int casmtr_window_score_fwd(const float* q, const float* key, const int64_t* idx, float* out,
                            int B, int N1, int N2, int K, int C, casmtr_stream_t stream);
/* fast_score_computation.score_backward     (score_cuda/src/score_computation.cpp:19-27,31 ; kernel :67-123)    */
int casmtr_window_score_bwd(const float* grad, const float* q, const float* key, const int64_t* idx,
                            float* dq, float* dkey, int B, int N1, int N2, int K, int C, casmtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused per-level kernels (what QTAttB / CascadeQTAttB / CoarseMatching / CascadeMatching run).
 * Token layout everywhere: [B, h*w, H, D] == [B, h*w, C], raster order of the h x w grid.
 * ---------------------------------------------------------------------------------------------------------- */

/* [B,C,h,w] -> [B,h*w,C]   (the `rearrange(x, "b c h w -> b (h w) c")...contiguous()` at
 * cuda_imp/.../modules/quadtree_attention.py:165-167,185-186,413-414)                                           */
int casmtr_nchw_to_tokens(const float* x, float* out, int B, int C, int HW, casmtr_stream_t stream);
/* the same for up to 9 tensors in ONE launch (a QTAttB call converts 3 pyramid levels x q,k,v); src/dst/C/HW are HOST
 * arrays of length n, every tensor has the same B                                                                */
int casmtr_nchw_to_tokens_multi(const float* const* src, float* const* dst, const int* C, const int* HW, int n, int B,
                                casmtr_stream_t stream);

/* QTAttB.process_coarse_level (modules/quadtree_attention.py:161-178): dense QK^T (fp32 MFMA) -> softmax over S
 * -> top-k -> A.V.   q [B,L,H,D], k/v [B,S,H,D].
 *   logits_ws : workspace of casmtr_qta_coarse_level_ws_floats_k(B,L,S,H,topk) floats: 1 (unused) when the register-tile kernel
 *               serves the shape (coarse_tile_kernel: S <= 1024, topk <= 60 -- the default for every shipped config); otherwise, and
 *               with CASMTR_COARSE_KERNEL=three, the three-kernel path's [B,H,L,S_pad] logits (S_pad = round_up(S,64)) + [B,H,L,2]
 *               row maxima / sums
 *   message [B,L,H,D]; acc_out [B,L,H,D] = message * w_level (NULL to skip); topk_score/topk_idx [B,L,topk,H]  */
int casmtr_qta_coarse_level_fwd(const float* q, const float* k, const float* v, float temp, int topk, float w_level,
                                float* logits_ws, float* message, float* acc_out, float* topk_score,
                                int64_t* topk_idx, int B, int L, int S, int H, int D, casmtr_stream_t stream);
size_t casmtr_qta_coarse_level_ws_floats(int B, int L, int S, int H);

/* QTAttB.process_fine_level + its share of the message merge (modules/quadtree_attention.py:180-229,262-284).
 *   q [B,h0*w0,H,D]; key/value [B,h1*w1,H,D]; prev_idx [B,(h0/2)*(w0/2),Kp,H] absolute key index on the
 *   (h1/2)x(w1/2) grid (the previous level's topk_idx); candidates = 4 children per entry, parent-major.
 *   message (nullable) [B,L,H,D]; acc_out (nullable) = acc_in[parent] + message*w_level; acc_in nullable (0).
 *   topk == 0 skips selection (finest level: the reference computes and discards it, :219-227).
 *   Outputs are in raster order of the h0 x w0 grid (i.e. after the un-quad rearrange :226-227).               */
int casmtr_qta_fine_level_fwd(const float* q, const float* key, const float* value, const int64_t* prev_idx,
                              float temp, int topk, float w_level, const float* acc_in, float* message,
                              float* acc_out, float* topk_score, int64_t* topk_idx,
                              int B, int h0, int w0, int h1, int w1, int H, int D, int Kp, casmtr_stream_t stream);

/* ---- the same level on QUAD-MAJOR operands (round 3; what QTAttB's fused path runs by default) ----------------------------------
 * Quad-major per head: x_qm[b][hd][Q][c][d] with Q = (r/2)*(w/2) + (col/2) the quad of token (r, col) of an h x w grid (h, w even),
 * c = (r&1)*2 + (col&1) its child slot (the reference's "(t1 t2)" order, :188-199), d < 32.  The 4 children of a parent selected at the
 * previous level are one contiguous 512-byte run, a (pair, head) slice is contiguous.
 *
 * casmtr_nchw_to_quads_multi: src_i [B, C_i, h_i, w_i] (the module's NCHW pyramids) -> dst_i [B, C_i/32, (h_i/2)*(w_i/2), 4, 32];
 *   the `rearrange(...)` calls of :165-167,185-189 folded into one layout pass.  src/dst/C/h/w: HOST arrays of length n <= 18 (both directions of a layer in one launch);
 *   tokens (nullable HOST array): tokens[i] != 0 converts tensor i to plain token-major [B, h_i*w_i, C_i] instead (the coarsest
 *   level's operands), so that ONE launch serves a whole QTAttB call.                                                                */
int casmtr_nchw_to_quads_multi(const float* const* src, float* const* dst, const int* C, const int* h, const int* w, const int* tokens,
                               int n, int B, casmtr_stream_t stream);
/* token-major [B, h*w, C] -> quad-major (the callers' / tests' entry into the layout)                                                 */
int casmtr_tokens_to_quads(const float* x, float* out, int B, int C, int h, int w, casmtr_stream_t stream);
/* [B,L,K,H] int64 (the reference's topk_idx layout) -> compact per-head table [B,H,L,K] int32                                          */
int casmtr_topk_idx_to_tab(const int64_t* idx, int32_t* tab, int B, int L, int K, int H, casmtr_stream_t stream);

/* QTAttB.process_fine_level on quad-major q [B,H,Lq0,4,32], key/value [B,H,Lq1,4,32] (Lq = quads of the h x w grid);
 *   parents [B,H,Lq0,Kp] int32: the previous level's top-k (absolute index on ITS key grid = quad index on this level's key grid);
 *   acc_in nullable [B,Lq0,H*32]; message / acc_out nullable [B,h0*w0,H*32] TOKEN-major raster (what the module returns);
 *   topk_tab nullable [B,H,h0*w0,topk] int32 (this level's top-k as the next level's `parents`); topk_score / topk_idx nullable
 *   [B,h0*w0,topk,H] (the reference's tensors, :219-227, on request).  Kp <= 32, topk <= 16, H in {1,2,4,8}; anything else is
 *   CASMTR_ERR_UNSUPPORTED and the caller uses the token-major entry point above.  Same results as casmtr_qta_fine_level_fwd.        */
int casmtr_qta_fine_level_quad_fwd(const float* q, const float* key, const float* value, const int32_t* parents, float temp,
                                   int topk, float w_level, const float* acc_in, float* message, float* acc_out,
                                   int32_t* topk_tab, float* topk_score, int64_t* topk_idx, int B, int h0, int w0, int h1,
                                   int w1, int H, int D, int Kp, casmtr_stream_t stream);
/* casmtr_qta_coarse_level_fwd with one more output: topk_tab nullable [B,H,L,topk] int32 (the finer level's `parents`).  Default
 * kernel (round 4, csrc/coarse_tile.hip; S <= 1024, topk <= 60): one launch, no workspace -- logits_ws may then be a 1-float dummy
 * (casmtr_qta_coarse_level_ws_floats_k says so) and topk_score / topk_idx may be NULL (the reference's int64 lists :170-175 are only
 * written on request).  Other shapes, or CASMTR_COARSE_KERNEL=three: the three-kernel path, which needs the workspace and both lists
 * (CASMTR_ERR_UNSUPPORTED otherwise).                                                                                                  */
size_t casmtr_qta_coarse_level_ws_floats_k(int B, int L, int S, int H, int topk);
int casmtr_qta_coarse_level_tab_fwd(const float* q, const float* k, const float* v, float temp, int topk, float w_level,
                                    float* logits_ws, float* message, float* acc_out, float* topk_score,
                                    int64_t* topk_idx, int32_t* topk_tab, int B, int L, int S, int H, int D, casmtr_stream_t stream);

/* CascadeQTAttB.forward (modules/quadtree_attention.py:400-452).
 *   q [B,h0*w0,C]; key/value [B,h1*w1,C]; topk_pos [B,(h0/2)*(w0/2),KW,2] (row,col) on the (h1/2)x(w1/2) grid;
 *   rel_pos nullable [B,nhead,h0*w0,4*KW]; message [B,h0*w0,C]; up_idx nullable [B,h0*w0,4*KW] int64.          */
int casmtr_cascade_attn_fwd(const float* q, const float* key, const float* value, const int64_t* topk_pos,
                            const float* rel_pos, float temp, int dilated, float* message, int64_t* up_idx,
                            int B, int h0, int w0, int h1, int w1, int nhead, int D, int KW, casmtr_stream_t stream);

/* CascadeQTAttB.forward on quad-major operands (see casmtr_nchw_to_quads_multi): q [B,H,Lq0,4,32], key/value [B,H,Lq1,4,32], topk_pos as above
 * with KW == 25 (5 x 5 windows) and dilation 1, rel_pos nullable [B,nhead,h0*w0,100]; message [B,h0*w0,nhead*32] token-major.  Horizontally
 * adjacent query quads whose windows are the same 5 x 5 block or one column apart share one gathered 5 x 6 box.  Window positions outside the
 * (h1/2) x (w1/2) grid are clamped per coordinate (the reference clamps the flattened child index, :429; get_window_warp_idx never
 * produces such positions).  Same results as casmtr_cascade_attn_fwd within the softmax tolerance; no up_idx output (callers that
 * need the explicit list use casmtr_cascade_attn_fwd or casmtr_window_expand_idx).                                                       */
int casmtr_cascade_attn_quad_fwd(const float* q, const float* key, const float* value, const int64_t* topk_pos,
                                 const float* rel_pos, float temp, float* message, int B, int h0, int w0, int h1, int w1,
                                 int nhead, int D, int KW, casmtr_stream_t stream);

/* CascadeFeatureTransformer.get_window_warp_idx, 'window' propagation (src/model/modules/transformer.py:416-440):
 *   idx [B,N] on an HxW grid -> out [B,N,ws*ws,2] (row,col) of the ws x ws window shifted inside the grid.      */
int casmtr_window_warp_idx(const int64_t* idx, int64_t* out, int B, int N, int H, int W, int ws,
                           casmtr_stream_t stream);

/* CoarseMatching.forward + get_coarse_match (src/model/functions/coarse_matching.py:40-153).
 *   feat0 [B,L,C], feat1 [B,S,C]; mask0 [B,L] / mask1 [B,S] uint8 or NULL; valid_hw [B,4] int32 (h0,w0,h1,w1) or NULL.
 *   sim_ws   : [B,L,S] floats -- receives the similarity matrix, and conf_matrix in place when want_conf != 0
 *   stats_ws : casmtr_dual_softmax_ws_bytes(...) bytes of scratch
 *   next_idx01/next_conf01 [B,L]; next_idx10/next_conf10 [B,S];
 *   b_ids/i_ids/j_ids [B*L] int64, mconf [B*L] float (capacity), n_matches: device int64 (count)                */
int casmtr_dual_softmax_fwd(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                            float temperature, int recip, float thr, int border_rm, const int32_t* valid_hw,
                            int h0c, int w0c, int h1c, int w1c, int want_conf, float* sim_ws, void* stats_ws,
                            int64_t* next_idx01, float* next_conf01, int64_t* next_idx10, float* next_conf10,
                            int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches,
                            int B, int L, int S, int C, casmtr_stream_t stream);
size_t casmtr_dual_softmax_ws_bytes(int B, int L, int S);

/* Same contract and results (indices identical, probabilities within the softmax tolerance); stats_ws sized by
 * casmtr_dual_softmax_split_ws_bytes.  The similarity matrix is computed on the f16 matrix pipe as three products of a two-term
 * f16 split of the row-normalised operands (error < 2^-15 |a||b|/(C T)); every entry within twice that bound of its row / column
 * maximum is re-evaluated with the exact fp32 fmaf chain before an index is decided, and inputs with more than 8 such entries in a
 * row or column (duplicated / all-zero feature rows) run the exact passes of casmtr_dual_softmax_fwd behind a device-side flag. */
int casmtr_dual_softmax_split_fwd(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                                  float temperature, int recip, float thr, int border_rm, const int32_t* valid_hw,
                                  int h0c, int w0c, int h1c, int w1c, int want_conf, float* sim_ws, void* stats_ws,
                                  int64_t* next_idx01, float* next_conf01, int64_t* next_idx10, float* next_conf10,
                                  int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches,
                                  int B, int L, int S, int C, casmtr_stream_t stream);
size_t casmtr_dual_softmax_split_ws_bytes(int B, int L, int S, int C);

/* CascadeMatching.forward, one direction per call (src/model/functions/cascade_matching.py:63-161).
 *   feat_q [B,N,C], feat_k [B,M,C], idx [B,N,K]; mask_q [B,N] / mask_k [B,M] uint8 or NULL;
 *   (h,w): the query grid, used only to let the 4 children of a quad share their (identical) window rows;
 *   conf nullable [B,N,K]; next_conf [B,N]; next_idx [B,N] (= idx[argmax])                                      */
int casmtr_window_match_fwd(const float* feat_q, const float* feat_k, const int64_t* idx, const uint8_t* mask_q,
                            const uint8_t* mask_k, float temperature, int recip, float* conf, float* next_conf,
                            int64_t* next_idx, int B, int N, int M, int K, int C, int h, int w,
                            casmtr_stream_t stream);

/* The same on IMPLICIT windows: instead of idx [B,N,4*KW] the kernel takes the topk_pos [B,(h0/2)*(w0/2),KW,2] (row,col on the
 * (h1/2)x(w1/2) grid) that CascadeQTAttB expands into it (cuda_imp/.../modules/quadtree_attention.py:419-450: 4 children per
 * position, parent-major, offsets (0,0),(0,d),(d,0),(d,d), clamped to [0,h1*w1-1]; every child of a query quad gets the same
 * list).  Results are identical to casmtr_window_match_fwd on the expanded tensor; 4*KW <= 128, h0 and w0 even.            */
int casmtr_window_match_pos_fwd(const float* feat_q, const float* feat_k, const int64_t* topk_pos, const uint8_t* mask_q,
                                const uint8_t* mask_k, float temperature, int recip, int dilated, float* conf,
                                float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW, int C,
                                casmtr_stream_t stream);
/* ... and the expansion itself, for callers that want the explicit tensor (data['stage_4c']['idx_c01'], training
 * supervision): up_idx [B,h0*w0,4*KW] int64 == CascadeQTAttB's second return value.                                        */
int casmtr_window_expand_idx(const int64_t* topk_pos, int64_t* up_idx, int B, int h0, int w0, int h1, int w1, int KW,
                             int dilated, casmtr_stream_t stream);

/* CascadeMatching.get_coarse_match, inference branch (cascade_matching.py:170-261,317-331) with
 * PostProcess.apply for method None / 'maxpool_nms' (post_processing.py:41-44,111-121) and
 * mask_window_border[_with_padding] (cascade_functions.py:120-172).
 *   nms_window 0 = no NMS; pre_conf{0,1} nullable [B,hp*wp]; ws: casmtr_nms_select_ws_bytes(...) bytes scratch;
 *   extra_keep nullable [B,H0*W0] uint8: a further per-token keep mask AND-ed in where PostProcess.apply's mask is formed
 *   (post methods other than None / 'maxpool_nms', post_processing.py:76-110, and the rt / rd filters, cascade_matching.py:193-226);
 *   outputs in (b,i) order with capacity B*H0*W0, count in *n_matches (device).                                 */
int casmtr_nms_select_fwd(const float* next_conf01, const int64_t* next_idx01, const int64_t* next_idx10,
                          int nms_window, float test_thr, const float* pre_conf0, int hp0, int wp0, float pre_thr0,
                          const float* pre_conf1, int hp1, int wp1, float pre_thr1, int border_rm,
                          const int32_t* valid_hw, int double_check, void* ws, int64_t* b_ids,
                          int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches,
                          int B, int H0, int W0, int H1, int W1, const uint8_t* extra_keep, casmtr_stream_t stream);
size_t casmtr_nms_select_ws_bytes(int B, int H0, int W0);

/* ------------------------------------------------------------------------------------------------------------
 * Callers of the attention kernels, token-major: QuadtreeAttention /
 * CascadeQuadtreeAttention (src/model/modules/quadtree_attention.py:9-100,103-176) without the NCHW round trip.
 * ---------------------------------------------------------------------------------------------------------- */

/* y_p[M,N] = x_p[M,K] . w_p[N,K]^T (+ bias_p[N]) for p < nprob <= 4 problems of one shape in ONE launch: the 1x1
 * q_proj / k_proj / v_proj convolutions (quadtree_attention.py:31-33,79-81,158-160; weight [N,K,1,1] viewed as [N,K])
 * and the output nn.Linear (:44,98,168).  fp32 MFMA, k-ascending fmaf chain, bias added after the chain.
 * x/w/bias/y are HOST arrays of device pointers; bias (or any bias[p]) may be NULL.  K % 32 == 0.                */
int casmtr_linear_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                      int nprob, int M, int N, int K, casmtr_stream_t stream);

/* The same projections with y_p written quad-major per head, [B][N/32][(h/2)*(w/2)][4][32] (the M = B*h*w rows of x_p are the
 * tokens of h x w grids, row-major): the operand layout of casmtr_qta_fine_level_quad_fwd / casmtr_cascade_attn_quad_fwd, i.e. the
 * reference's "b c (h t1) (w t2) -> b (h w) (t1 t2) c" (cuda_imp/.../modules/quadtree_attention.py:188-189) applied by the producer.
 * Same arithmetic and values as casmtr_linear_fwd followed by casmtr_tokens_to_quads.  h, w even, N % 32 == 0.  (round 5)      */
int casmtr_linear_quads_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                            int nprob, int M, int N, int K, int h, int w_, casmtr_stream_t stream);

/* The same projections, fp32-accurate, on the f16 matrix pipe (round 6; opt-in -- the default above is the exact chain).  Every operand
 * row is scaled by a power of two and split into two f16 terms (a = hi + lo + r, |r| <= 2^-22 |a|); three v_mfma_f32_32x32x16_f16
 * products with fp32 accumulation give x.w to |y_split - y_chain| <= 2^-15 |x_m| |w_n| (csrc/callers.hip; measured ~1e-7 |x||w|).  The
 * reference's own conv / linear (src/model/modules/quadtree_attention.py:31-33,44,79-81,98) is a BLAS call with unspecified
 * accumulation order, so this path's contract is that tolerance.
 *   casmtr_linear_split_prep: weight w [N,K] -> `prep` (casmtr_linear_split_prep_bytes(N, K) bytes: the GEMM's f16 tile image + per-row
 *     factors); once per weight tensor.  N % 128 == 0, K % 32 == 0, K <= 256, else CASMTR_ERR_UNSUPPORTED (bytes: 0).
 *   casmtr_linear_split_fwd: y_p = x_p . w_p^T (+ bias_p) from prepared weights; h, w_ != 0: y_p quad-major as casmtr_linear_quads_fwd,
 *     0: token-major [M,N].                                                                                                          */
size_t casmtr_linear_split_prep_bytes(int N, int K);
int casmtr_linear_split_prep(const float* w, void* prep, int N, int K, casmtr_stream_t stream);
int casmtr_linear_split_fwd(const float* const* x, const void* const* wprep, const float* const* bias, float* const* y,
                            int nprob, int M, int N, int K, int h, int w_, casmtr_stream_t stream);

/* The q / k / v projections of QuadtreeAttention.forward TOGETHER WITH their pyramid (src/model/modules/quadtree_attention.py:78-88:
 * conv1x1, then F.avg_pool2d(kernel 2, stride 2) per further level) in one launch (csrc/linear_pc.hip, round 6): x_p [B, h*w, K]
 * token-major -> y0_p quad-major as casmtr_linear_split_fwd(h, w_) writes it; y1_p = avg_pool2d(y0_p): quad-major
 * [B][N/32][(h/4)*(w/4)][4][32] (h % 4 == w_ % 4 == 0), or with y1_tokens token-major [B][(h/2)*(w/2)][N] (the coarsest level of a
 * two-level pyramid); y2_p = avg_pool2d(y1_p) token-major [B][(h/4)*(w/4)][N] (needs y1 quad-major).  y1 / y2: NULL, or arrays whose
 * NULL entries skip that problem's level.  Values: bit for bit those of casmtr_linear_split_fwd + casmtr_quad_pool_fwd (+ ..._fwd
 * with to_tokens) -- the pooled levels are summed in registers from the projected values in the same order -- without reading a
 * projected level back.  K = 128 and N % 128 == 0, or K = 256 and N % 256 == 0; nprob <= 8, nprob * N <= 2048; h, w_ even; else CASMTR_ERR_UNSUPPORTED.                        */
int casmtr_linear_split_pyramid_fwd(const float* const* x, const void* const* wprep, const float* const* bias, float* const* y0,
                                    float* const* y1, float* const* y2, int y1_tokens, int nprob, int B, int h, int w_, int N, int K,
                                    casmtr_stream_t stream);

/* F.avg_pool2d(kernel_size=2, stride=2) of the pyramid loop (quadtree_attention.py:82-90) on QUAD-major tensors: src_i
 * [B][C/32][(h/2)*(w/2)][4][32] (h x w tokens) -> the pooled (h/2 x w/2) level, quad-major again ([B][C/32][(h/4)*(w/4)][4][32];
 * h % 4 == w % 4 == 0) or, with to_tokens, token-major [B][(h/2)*(w/2)][C] (the coarsest level's layout).  A pooled token's window
 * is its quad's four children, summed in (row, col) order, then * 0.25: the values of casmtr_token_pool_fwd.  n <= 4.  (round 5) */
int casmtr_quad_pool_fwd(const float* const* src, float* const* dst, int n, int B, int h, int w, int C, int to_tokens,
                         casmtr_stream_t stream);

/* F.avg_pool2d(kernel_size=2, stride=2) of the pyramid loop (quadtree_attention.py:82-90) on token-major tensors:
 * src_i [B,H,W,C] -> dst_i [B,H/2,W/2,C] for i < n <= 4 tensors in one launch (odd H/W: last row/column dropped,
 * as torch does).  Window sum in (row, col) order, then * 0.25.  C % 4 == 0.                                      */
int casmtr_token_pool_fwd(const float* const* src, float* const* dst, int n, int B, int H, int W, int C,
                          casmtr_stream_t stream);

/* Depth-wise 3x3 convolution (stride 1, zero padding 1) on token-major data: x, y [B,H*W,C], w [C,1,3,3] viewed as [C,9],
 * bias [C] or NULL.  Replaces DWConv (transformer.py:52-63: transpose to NCHW, nn.Conv2d(groups=C), transpose back) inside
 * Mlp (:65-94, fc1 -> ReLU -> DWConv -> GELU -> fc2) and PosCNN (gvt.py:397-411, x + DWConv(x)):
 *   acc = bias[c]; for ky,kx row-major: acc = fmaf(in(y+ky-1, x+kx-1, c), w[c][3ky+kx], acc), taps outside the grid skipped;
 *   CASMTR_DW_PRE_RELU: in() = max(x, 0);  CASMTR_DW_POST_GELU: erf-form GELU of acc;  CASMTR_DW_ADD_INPUT: + x[y,x,c] last.
 * C % 4 == 0, y != x.                                                                                                  */
#define CASMTR_DW_PRE_RELU 1
#define CASMTR_DW_POST_GELU 2
#define CASMTR_DW_ADD_INPUT 4
int casmtr_dwconv3x3_tokens_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                int flags, casmtr_stream_t stream);

/* nn.LayerNorm over the last axis of x [rows, C] (norm1 / norm2 of QuadtreeBlock and CascadeQuadtreeBlock,
 * transformer.py:141-196, 305-345), optionally + residual [rows, C] after the affine map (x + norm(...) patterns):
 *   mean = (sum x) / C; var = (sum (x - mean)^2) / C; y = (x - mean) * (1 / sqrt(var + eps)) * gamma + beta (+ residual).
 * C % 4 == 0, C <= 1024; y may alias x or residual.                                                                   */
int casmtr_layer_norm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                          long long rows, int C, float eps, casmtr_stream_t stream);

/* Window self-attention of the local blocks (GroupAttention.forward_mask, cascade_attention.py:124-157 / gvt.py:102-133):
 * qkv [B,H*W,3,nhead,head_dim] = the fused projection of the UN-padded tokens, out [B,H*W,nhead*head_dim].  Tokens are grouped
 * in ws x ws windows anchored at (0,0); windows cut by the bottom / right edge hold fewer tokens (the reference pads them and
 * masks the padded keys with -1000, i.e. exp() == 0 exactly: same result for every real token).
 *   s_j = (chain_d fmaf(q[d], k_j[d])) * scale; p_j = exp(s_j - max s); o = (sum_j p_j v_j) * (1 / sum_j p_j), j in window raster order.
 * head_dim == 32 and ws == 7 (every shipped config), otherwise CASMTR_ERR_UNSUPPORTED.                                      */
int casmtr_window_attn_fwd(const float* qkv, float* out, int B, int H, int W, int nhead, int head_dim, int ws, float scale,
                           casmtr_stream_t stream);

/* POLA neighbourhood self-attention of the indoor model's local blocks (NeighborWindowAttention + the unfold plumbing of
 * POLATransBlock, src/model/modules/POLAttention.py:70-172, 280-320): the queries of a ws x ws window attend to the 3 x 3 windows
 * around it; bias_table [(4 ws - 1)^2, nhead] is relative_position_bias_table, indexed by (qy - ky + 3 ws - 1) * (4 ws - 1) +
 * (qx - kx + 3 ws - 1) with key coordinates inside the 3 ws x 3 ws neighbourhood.  q, k0, v0, out [B,H*W,nhead*head_dim]: projections
 * of the un-padded tokens, k0 / v0 WITHOUT bias (a key bias shifts every logit of a row equally; the caller adds the value bias to
 * the output).  Key positions outside the map are the reference's zero padding: logit = bias, value 0.
 *   s_j = (chain_d fmaf(q[d], k0_j[d])) * scale + bias[rel]; o = (sum_j exp(s_j - max) v0_j) / (sum_j exp(s_j - max)), running max / sum
 * over the nine windows.  head_dim == 32 and ws == 7, otherwise CASMTR_ERR_UNSUPPORTED.                                        */
int casmtr_pola_attn_fwd(const float* q, const float* k0, const float* v0, const float* bias_table, float* out, int B, int H, int W,
                         int nhead, int head_dim, int ws, float scale, casmtr_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart): per-kernel launch durations from HIP events recorded on the
 * launch stream.  Off by default.  casmtr_prof_enable(1) starts a fresh collection; casmtr_prof_read() waits for
 * the recorded events and returns the summed duration (ms) and the number of launches of kernel `id`.
 * Each timed launch costs two event records on the stream (~3 us of serialisation each on MI355X).
 * ---------------------------------------------------------------------------------------------------------- */
enum {
    CASMTR_PROF_DS_GEMM = 0, CASMTR_PROF_DS_REDUCE, CASMTR_PROF_DS_CONF, CASMTR_PROF_DS_SELECT,
    CASMTR_PROF_COARSE_LOGITS, CASMTR_PROF_COARSE_ROW, CASMTR_PROF_COARSE_AV, CASMTR_PROF_QTA_FINE,
    CASMTR_PROF_CASCADE_ATTN, CASMTR_PROF_WINDOW_MATCH, CASMTR_PROF_NMS_SELECT, CASMTR_PROF_LAYOUT,
    CASMTR_PROF_WINDOW_WARP, CASMTR_PROF_LINEAR, CASMTR_PROF_TOKEN_POOL, CASMTR_PROF_COARSE_FUSED,
    CASMTR_PROF_GLUE, CASMTR_PROF_QTA_FINE2, CASMTR_PROF_DS_SPLIT, CASMTR_PROF_DS_FIX, CASMTR_PROF_DS_GEMM_EDGE, CASMTR_PROF_COUNT
};
void casmtr_prof_enable(int on);
/* timing experiments only: phase-elimination switches of the LDS-DMA kernels (1: no row transfers, 2: no arithmetic).
 * Any non-zero value makes their results meaningless; 0 (default) is the product behaviour.                              */
void casmtr_debug_set(int flags);
/* fresh collection that times ONLY kernel `id` (two event records per launch of that kernel, none for the others) */
int casmtr_prof_enable_only(int id);
/* create `pairs` event pairs now, so that no hipEventCreate falls into a timed region (bench.py: launches per step x timed steps) */
int casmtr_prof_reserve(int pairs);
int casmtr_prof_read(int id, double* total_ms, int* count);
/* the individual launch durations (ms) of scope `id`, in launch order: writes min(count, cap) values, returns the count (-1: error) */
int casmtr_prof_read_all(int id, double* ms_out, int cap);
/* name of scope `id` (a stage of the path) / the kernel symbol(s) that ran under it since it was last timed ("" if none) */
const char* casmtr_prof_name(int id);
const char* casmtr_prof_symbol(int id);

#ifdef __cplusplus
}
#endif
#endif /* CASMTR_HIP_H */
