/* libsrec_hip.so - C ABI of the MI355X (gfx950) session-recommendation training hot path.
 *
 * The reference (SpaceLearner/SessionRec-pytorch) has no FFI: its hot path is Python calling
 * DGL / PyTorch CUDA ops.  Each entry point below names the reference call site(s) whose
 * arithmetic it replaces; the host-side mirror (sessionrec-pytorch_amd/) binds them with ctypes
 * (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (outputs pre-allocated), fp32 /
 *     int32 unless noted; the library allocates nothing and never synchronises;
 *   - `stream` is a hipStream_t (pass the framework's current stream; NULL = default stream);
 *   - row-major matrices with an explicit leading dimension `ld_*` (floats); d, ld multiples of 4,
 *     base pointers 16-byte aligned (float4 / 1 KiB-per-wave coalesced rows);
 *   - `dyn*` (nullable) points at an int32 in device memory holding the LIVE extent of a
 *     capacity-padded dimension, so one captured hipGraph serves batches of any size:
 *     effective n = min(n_cap, *dyn); rows in [n, n_cap) are written as zeros by producers;
 *   - return 0 on success, a hipError_t value or SREC_BAD_ARG (1001) otherwise.
 */
#ifndef SREC_H
#define SREC_H
#include "srec_hg.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SREC_BAD_ARG 1001

/* ---- dense linear algebra on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) ----------------
 * C[m,n] = alpha * sum_k A(m,k) B(n,k) + beta*C[m,n] + bias[n];  A(m,k)=A[m*a_rs+k*a_cs], same for B.
 * dyn_mode: 0 none, 1 clamps M, 2 clamps K.  ws (nullable): ws_floats of scratch for split-K slabs.
 * Replaces nn.Linear fwd/bwd: srgnn.py:66-68,124  lessr.py:16-17,59-62,94-100,164  msgifsr.py:114-116,202
 * gatconv.py:157,282-283; GRU input/hidden projections srgnn.py:15, msgifsr.py:25. */
int srec_gemm_f32(const float* A, int a_rs, int a_cs, const float* B, int b_rs, int b_cs, float* C, int ldc,
                  const float* bias, int M, int N, int K, const int* dyn, int dyn_mode, float alpha, float beta,
                  float* ws, long ws_floats, void* stream);

/* bf16-operand variant (v_mfma_f32_32x32x16_bf16, fp32 accumulate): A [M,K], B [N,K] fp32 in HBM, rounded to
 * bf16 while staged into LDS; K % 32 == 0; dyn (nullable) clamps M.  The reduced-precision path BASELINE
 * config C3 names; same call sites as srec_gemm_f32 (forward and backward-data products). */
int srec_gemm_bf16_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, int M,
                      int N, int K, const int* dyn, float alpha, float beta, float* ws, long ws_floats, void* stream);

/* bf16-operand weight-gradient product: C[N,K] = alpha * A[Mred,N]^T B[Mred,K] + beta*C (dW = dY^T X of every
 * nn.Linear backward: autograd of F.linear at e.g. msgifsr.py:77-79, gatconv.py:166-175); operands are transposed
 * to reduction-contiguous bf16 while staged into LDS; dyn (nullable) clamps the reduction rows. N % 4 == K % 4 == 0. */
int srec_gemm_bf16_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int Mred, int N, int K,
                      const int* dyn, float alpha, float beta, float* ws, long ws_floats, void* stream);

/* ---- fused full-catalog scoring + softmax-CE (score_ce.hip) -----------------------------------------
 * z[b,v] = cs[v] * <sr_b, E_v> (cs NULL -> 1).  Replaces sr @ E^T, log(softmax), nll_loss:
 * srgnn.py:145-147  niser.py:149-156  lessr.py:182-183  msgifsr.py:276-309,321  train.py:99 (+ backward). */
int srec_ce_plan(int B, int V, int d, int* n_item_tiles, int* n_ranges);
/* ws_stats: 2*n_item_tiles*B floats.  Outputs lab_logit[B], lse[B], lossvec[B], loss[1]. */
int srec_score_ce_fwd(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs, const int* labels,
                      int B, int V, int d, const int* dynB, float* ws_stats, float* lab_logit, float* lse,
                      float* lossvec, float* loss, void* stream);
/* ws_dsr: n_ranges*B*d floats.  gscale (nullable): upstream d loss (with ga / gc given it multiplies them: no 1 / B).  Outputs dE[V,d] (every row), dsr[B,d].
 * parts: bit0 = dE kernel, bit1 = d sr kernels (3 = both), bit2 = accumulate into dE.  ga/gc (nullable): per-session coefficients
 * dS[b,v] = (ga[b] softmax[b,v] - gc[b] [v==label_b]) cs[v] for losses built from (lse_b, z[b,label_b]), e.g. the
 * order-fusion mixture of msgifsr.py:311-317. */
int srec_score_ce_bwd(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs, const int* labels,
                      const float* lse, const float* gscale, const float* ga, const float* gc, int B, int V, int d,
                      const int* dynB, float* dE, int ld_de, float* ws_dsr, float* dsr, int parts, void* stream);
/* log-probabilities (B,V): the (B,num_items) tensor every reference model's forward() returns. */
int srec_score_logp(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs, const float* lse, int B,
                    int V, int d, const int* dynB, float* logp, long ld_logp, void* stream);

/* bf16-operand variant of the fused scoring (score_ce_bf16.hip; BASELINE config C3 "bf16"): same call sites as
 * srec_score_ce_fwd/bwd.  srec_bf16_prepare rounds rows [R,d] fp32 (RNE) into a row-major copy dst16 [Rp, d_pad] (zero for
 * rows >= live R / columns >= d; Rp % 128 == 0; d <= 256, d % 4 == 0; d_pad from srec_ce_plan_bf16) and, when dstT16 != NULL,
 * a transposed copy [d_pad, Rp] (no longer read by the scoring kernels: pass NULL); call it once per step for the table and
 * once per head for the session vectors.
 * ws_stats >= 2 * n_stat_slabs * B floats, ws_dsr >= n_ranges * B * d floats.  Soft-max statistics, exp and all
 * accumulation stay fp32; dE / dsr are fp32.  The backward runs both parts in one launch (parts bits as above;
 * bit3 = leave the d-sr partial slabs in ws_dsr unreduced); it reads only the row-major copies (ET16 is ignored and may be
 * NULL: the transposed MFMA fragments are taken from the row-major LDS image by ds_read_b64_tr_b16; ws_de: see
 * srec_ce_de_split below, NULL = no session split). */
int srec_bf16_prepare(const float* src, int ld, int R, const int* dynR, int d, void* dst16, void* dstT16, int Rp,
                      void* stream);
int srec_ce_plan_bf16(int B, int V, int d, int* n_stat_slabs, int* n_ranges, int* d_pad);
int srec_score_ce_fwd_bf16(const void* sr16, int Bp, const void* E16, int Vp, const float* cs, const int* labels,
                           int B, int V, int d, const int* dynB, float* ws_stats, float* lab_logit, float* lse,
                           float* lossvec, float* loss, void* stream);
int srec_score_ce_bwd_bf16(const void* sr16, float* ws_de, int Bp, const void* E16, const void* ET16, int Vp,
                           const float* cs, const int* labels, const float* lse, const float* gscale, const float* ga,
                           const float* gc, int B, int V, int d, const int* dynB, float* dE, int ld_de, float* ws_dsr,
                           float* dsr, int parts, void* stream);
/* Session split of the backward's item tiles (round 6; the step a rank of an N-GPU job runs scores N x 512 sessions against V / N
 * table rows: few item tiles, each streaming N x the sessions - train.py:94-101 sharded): srec_ce_de_split(B, V, d, &split): workgroups
 * per item tile at this shape (1 = none).  The caller may pass ws_de (nullable) >= split x V x d floats and the split in `parts`
 * bits 8 - 15: workgroup (tile, s) then writes slab s, and the slabs are summed in order into dE (added to it with parts bit 2)
 * by one more launch.  Needs ld_de == d == d_pad. */
int srec_ce_de_split(int B, int V, int d, int* split);

/* ---- embedding rows (rowops.hip) --------------------------------------------------------------------
 * gather: nn.Embedding lookup srgnn.py:133 niser.py:133 lessr.py:168 msgifsr.py:247.
 * scatter_add_sorted: its backward as a deterministic segmented sum (items[u] distinct, pos grouped by ptr). */
int srec_gather_rows(const float* src, int ld_src, const int* idx, float* out, int ld_out, int n_cap,
                     const int* dyn, int d, void* stream);
/* backward of the last-node pick x[last] (srgnn.py:140 niser.py:140 lessr.py:177; `last` non-negative and strictly ascending,
 * one entry per session): dst [nrows, d] = zeros with dst[idx[j]] = g[j] for the live j < *dyn - every row written, no
 * separate zero fill */
int srec_expand_rows_sorted(const float* g, int ld_g, const int* idx, int n_cap, const int* dyn, int nrows, int d,
                            float* dst, int ld_dst, void* stream);
int srec_scatter_add_sorted(const float* g, int ld_g, const int* items, const int* ptr, const int* pos, float* dst,
                            int ld_dst, int u_cap, const int* dyn, int d, int accumulate, void* stream);
/* the lookup with its feature dropout fused (msgifsr.py:247 `dropout(embedding(iid))`): out[r, c] *= mask(r * d + c), mask =
 * 1 / (1 - p) with probability 1 - p else 0, from a counter-based hash keyed by (seed, *counter, salt) - counter (nullable) is a
 * device int that is constant within a training step and changes between steps (the optimizer's step count).  The backward
 * re-derives the same mask while it sums the gradient rows (g must be contiguous: ld_g == d). */
int srec_gather_rows_drop(const float* src, int ld_src, const int* idx, float* out, int ld_out, int n_cap, const int* dyn,
                          int d, float p, int seed, const int* counter, int salt, void* stream);
int srec_scatter_add_sorted_drop(const float* g, int ld_g, const int* items, const int* ptr, const int* pos, float* dst,
                                 int ld_dst, int u_cap, const int* dyn, int d, int accumulate, float p, int seed,
                                 const int* counter, int salt, void* stream);
int srec_scatter_add_sorted_ex(const float* g, int ld_g, const int* items, const int* ptr, const int* pos, float* dst,
                               int ld_dst, int u_cap, const int* dyn, int d, int accumulate, float p, int seed,
                               const int* counter, int salt, const float* projW, int ld_w, float* radial, void* stream);
/* row-sharded lookup backward (the nn.Embedding gradient of srgnn.py:133 / msgifsr.py:247 when the table is sharded over the
 * ranks): dst[rel[q], :] += rows[q, :] for every request q < w ucap with rel[q] >= 0, the requests of ALL w <= 16 ranks in one launch
 * and - an item may be requested by several ranks - added per item in rank order (the bits of w rank-by-rank
 * srec_scatter_add_sorted calls).  ids [w ucap] = the requests' global item ids, ascending inside each rank's list of ucap (-1
 * padding behind them); rows [w ucap, d] dense.  radial (nullable) as in srec_scatter_add_sorted_ex. */
int srec_add_rows_ranks(const float* rows, int d, const int* rel, const int* ids, int w, int ucap, float* dst, int ld_dst,
                        const float* projW, int ld_w, float* radial, void* stream);
/* Embedding(max_norm) in-place renorm, idx distinct or NULL (= all rows): lessr.py:126 msgifsr.py:162 */
int srec_renorm_rows(float* W, int ld, const int* idx, int n_cap, const int* dyn, int d, float max_norm,
                     void* stream);
/* out[v] = scale / norm(E_v); eps_mode 0: max(norm,eps) (F.normalize), 1: norm+eps (niser.py:151) */
/* the renorm of every row (max_norm > 0; <= 0: none) + the bf16 operand copy dst16 [>= n, Dp] (Dp >= d, Dp % 4 == 0,
 * columns d .. Dp zero) of the table the bf16 scoring kernels read, in one pass (d <= 1024) */
int srec_renorm_rows_bf16(float* W, int ld, int n, int d, float max_norm, void* dst16, int Dp, void* stream);
int srec_row_invnorm(const float* W, int ld, int n, int d, int eps_mode, float eps, float scale, float* out,
                     void* stream);
/* row L2 normalisation of node / session features: niser.py:135,142,148  msgifsr.py:253,263,273 */
int srec_normalize_fwd(const float* X, int ld_x, float* Y, int ld_y, float* inv, int n_cap, const int* dyn, int d,
                       int eps_mode, float eps, void* stream);
/* the same + a bf16 copy dst16 [>= n_cap, Dp] of the normalised rows (zero rows past the live count; columns d .. Dp are
 * not written): the session-vector operand of srec_score_ce_*_bf16 without a conversion pass of its own */
int srec_normalize_fwd_bf16(const float* X, int ld_x, float* Y, int ld_y, float* inv, int n_cap, const int* dyn, int d,
                            int eps_mode, float eps, void* dst16, int Dp, void* stream);
int srec_normalize_bwd(const float* Y, int ld_y, const float* dY, int ld_dy, const float* inv, float* dX, int ld_dx,
                       int n_cap, const int* dyn, int d, void* stream);
/* the same for np <= 4 row blocks of different tensors normalised into ONE stacked matrix Y [sum n_p, d] (MSGIFSR: the
 * per-order features of msgifsr.py:253 feed the batched MSHGNN layer stacked) - one launch instead of np + a concatenation.
 * X / dX / dyn: HOST arrays of np device pointers (dyn entries nullable), ld / n: HOST int arrays. */
int srec_normalize_group_fwd(int np, const void* X, const int* ld, const int* n, const void* dyn, float* Y, int ld_y,
                             float* inv, int d, int eps_mode, float eps, void* stream);
int srec_normalize_group_bwd(int np, const void* dX, const int* ld, const int* n, const void* dyn, const float* Y, int ld_y,
                             const float* dY, int ld_dy, const float* inv, int d, void* stream);
/* out [n, da + db] = [a | b]: the feature-axis concatenation in front of fc_sr (srgnn.py:143, msgifsr.py:270-272) */
int srec_cat_cols(const float* a, int lda, int da, const float* b, int ldb, int db, int n, float* out, void* stream);
/* chain rule of the catalog-row normalisation on the dense dE: G_v -= e_v <e_v, G_v> */
int srec_rownorm_project(const float* W, int ld_w, const float* cs, float inv_scale, float* G, int ld_g, int n, int d,
                         void* stream);
/* deferred form of the projection: the scoring backward leaves dE unprojected, later additions to it (lookup gradients)
 * record their radial part radial[v] += <W_v, l_v> (srec_scatter_add_sorted_ex), and either this call or the optimizer's row pass
 * (srec_adam_rows_proj) applies G_v -= W_v (<W_v, G_v> - radial[v]) inv_v^2 and clears radial */
int srec_rownorm_project_radial(const float* W, int ld_w, const float* cs, float inv_scale, float* G, int ld_g, int n, int d,
                                float* radial, void* stream);
/* out[c] (+)= sum_r w[r, c/D] * X[r,c] (wgt NULL -> plain column sums: bias / fc_e / attention-vector
 * gradients).  Two deterministic stages; ws = 128*ncol floats of scratch (16-byte aligned). */
int srec_col_sum(const float* X, int ld, const float* wgt, int H, int D, int n_cap, const int* dyn, int ncol,
                 float* out, int accumulate, float* ws, void* stream);

/* Static per-session budgets of the one-wavefront-per-session / per-destination kernels (LDS arrays): nodes of one session in
 * a read-out (srec_seg_attn_*), node degree per relation (srec_gat_*, srec_hg_*), in-degree in LESSR's shortcut graph
 * (srec_sgat_*).  HOST pointers (nullable).  The reference has no such limit (DGL's degree buckets grow with the data,
 * collate.py:87-217 builds any session length); the host mirror checks every batch against these when it is collated and
 * raises instead of truncating (ops.check_limits). */
int srec_limits(int* max_session_nodes, int* max_degree, int* max_degree_sgat);

/* Batch intake of a replayed step (graph.GraphedTrainStep._stage; replaces the `x.to(device)` of the reference's
 * prepare_batch, /root/reference/src/utils/train.py:26-30, for capacity-padded batches): dst [n] (device) = src [n] int32 words
 * read by a kernel from page-locked HOST memory (or device memory); both 16-byte aligned.  Ordered on `stream`. */
int srec_copy_words(const int* src, int* dst, long n, void* stream);
/* the same INSIDE a captured step: the source of this replay is looked up in a mailbox of M entries {address lo, address hi,
 * words, expected counter} in page-locked host memory at index *counter % M (counter: the optimizer's device-side step count),
 * so no copy command precedes the graph launch; *err (device) = 1 when the entry's expected counter differs. */
int srec_copy_words_mailbox(const int* mailbox, int M, const int* counter, int* dst, long cap, int* err, void* stream);

/* ---- per-session kernels (segops.hip): one wavefront per session ------------------------------------
 * attention readout core: srgnn.py:79-86 niser.py:77-84 lessr.py:106-113 msgifsr.py:139-146 */
int srec_seg_attn_fwd(const float* U, int ld_u, const float* Vq, int ld_v, const float* we, const float* X, int ld_x,
                      const int* seg, int B, const int* dynB, int h, int D, float* alpha, float* out, int ld_out, void* stream);
int srec_seg_attn_bwd(const float* dout, int ld_do, const float* X, int ld_x, const float* alpha, const float* U,
                      int ld_u, const float* Vq, int ld_v, const float* we, const int* seg, int B, const int* dynB,
                      int h, int D, int n_cap, float* dX, int ld_dx, float* dU, int ld_du, float* dVq, int ld_dv,
                      float* dwe_part, int ld_dw, void* stream);
/* (bu, nullable: bias of the U product added inside, sigmoid(U + bu + Vq[b]) - msgifsr.py:114-115 puts the bias on fc_u;
 * out_hi / out_lo [B, ld16] and dU16 [2][n_cap, h] / dV16, dW16 [2][B, h], nullable: bf16 hi / lo splits of the outputs, the
 * operand copies of the 3-term split products of the read-out head - csrc/split16.hip; dU may be NULL when dU16 is given) */
/* out_i = h_i + mean_{session(i)} f  (msgifsr.py:86-89) */
int srec_seg_mean_add_fwd(const float* H, int ld_h, const float* F, int ld_f, const int* seg, int B, const int* dynB,
                          int D, float* out, int ld_out, void* stream);
int srec_seg_mean_add_bwd(const float* dout, int ld_do, const int* seg, int B, const int* dynB, int D, float* dF,
                          int ld_df, void* stream);

/* ---- MSGIFSR message passing (gat.hip): multi-head GAT over one relation, layout [N,H,D] ------------
 * gatconv.py:267-311 via msgifsr.py:58-64,74-89.  el/er [N,H]; A, DP [E,H] indexed by edge id;
 * in_ptr/in_idx = in-edge CSR of the destinations, out_ptr/out_idx = out-edge CSR of the sources. */
int srec_head_dot(const float* X, int ld, const float* a, int n_cap, const int* dyn, int H, int D, float* out,
                  void* stream);
int srec_gat_agg_fwd(const float* Fs, int ld_s, const float* el, const float* er, const int* in_ptr,
                     const int* in_idx, const int* esrc, int nd_cap, const int* dyn_nd, int H, int D, float slope,
                     float* A, float* rst, int ld_r, void* stream);
int srec_gat_bwd_dst(const float* dR, int ld_r, const float* Fs, int ld_s, const float* el, const float* er,
                     const float* A, const int* in_ptr, const int* in_idx, const int* esrc, int nd_cap,
                     const int* dyn_nd, int H, int D, float slope, float* DP, float* der, void* stream);
int srec_gat_bwd_src(const float* dR, int ld_r, const float* A, const float* DP, const float* attn_l,
                     const int* out_ptr, const int* out_idx, const int* edst, int ns_cap, const int* dyn_ns, int H,
                     int D, float* dFs, int ld_s, float* del, const float* der, const float* attn_r, void* stream);
int srec_head_outer(const float* wgt, const float* a, int n_cap, const int* dyn, int H, int D, float* out, int ld,
                    void* stream);
/* h_v = max_head(sum_i R_i + bias + nres*x): msgifsr.py:78-85 (+ identity residual / bias of gatconv.py:306-311).
 * rsts is a HOST array of n_rst (<= 8) device pointers. */
int srec_head_combine_fwd(const float* const* rsts, int n_rst, int ld_r, const float* x, int ld_x, const float* bias,
                          float nres, int n_cap, const int* dyn, int H, int D, float* out, int ld_o,
                          unsigned char* arg, void* stream);
int srec_head_combine_bwd(const float* dout, int ld_o, const unsigned char* arg, int n_cap, const int* dyn, int H,
                          int D, float* dR, int ld_r, void* stream);

/* ---- GRU gate math (gru.hip): msgifsr.py:25,42 (k-gram GRU), srgnn.py:15,45 (GRUCell) -----------------
 * GI/GH = input/hidden projections [n,3d] (r,z,n); GH NULL => h_prev = 0 and gh = bhh. */
int srec_gru_pointwise_fwd(const float* GI, int ld_gi, const float* GH, int ld_gh, const float* bhh, const float* Hp,
                           int ld_hp, int n_cap, const int* dyn, int d, float* Hn, int ld_hn, float* gates,
                           void* stream);
int srec_gru_pointwise_bwd(const float* dHn, int ld_dh, const float* gates, const float* GH, int ld_gh,
                           const float* bhh, const float* Hp, int ld_hp, int n_cap, const int* dyn, int d, float* dGI,
                           int ld_dgi, float* dGH, int ld_dgh, float* dHp, int ld_dhp, void* stream);
/* out = 0.5*mean_t X[n,t,:] + 0.5*Hl  (msgifsr.py:37,45), X contiguous [n,k,d] */
int srec_gram_combine_fwd(const float* X, const float* Hl, int ld_h, int n_cap, const int* dyn, int k, int d,
                          float* out, int ld_o, void* stream);
int srec_gram_combine_bwd(const float* dout, int ld_o, int n_cap, const int* dyn, int k, int d, float* dX, float* dHl,
                          int ld_h, void* stream);

/* SRGNN weighted-mean neighbour aggregation (srgnn.py:21-29,36-41): coef[e] = w_e / sum of w into the same
 * endpoint; out[v] = sum_{e in list(v)} coef[e] X[other[e]].  (ptr, idx) = in-edge CSR with other = esrc for
 * the graph, out-edge CSR with other = edst for the reversed graph; the backward is the same kernel on the
 * opposite CSR. */
int srec_edge_coef(const int* ptr, const int* idx, const int* ew, int n_cap, const int* dyn, float* coef,
                   void* stream);
int srec_edge_agg(const float* X, int ld_x, const int* ptr, const int* idx, const int* other, const float* coef,
                  int n_cap, const int* dyn, int D, float* out, int ld_o, void* stream);

/* ---- LESSR (bn.hip, gruseq.hip, sgat.hip) -------------------------------------------------------------
 * BatchNorm1d lessr.py:12,32,56,66,90,105,162,179 (ws = 64*D floats); PReLU lessr.py:140,149,159 (ws = 32*D).
 * srec_bn_fwd_train: training mode in two launches - batch statistics over the live rows, y, mean / biased var [D] for the
 * backward, and nn.BatchNorm1d's buffer updates (running_mean / running_var with the unbiased variance, nbt =
 * num_batches_tracked (int64) += 1; all three nullable).  srec_bn_apply_fwd: eval mode (given statistics). */
int srec_bn_fwd_train(const float* X, int ld_x, int n_cap, const int* dyn, int D, const float* gamma, const float* beta,
                      float eps, float momentum, float* rmean, float* rvar, long long* nbt, float* mean, float* var,
                      float* Y, int ld_y, float* ws, void* stream);
int srec_bn_apply_fwd(const float* X, int ld_x, const float* mean, const float* var, float eps, const float* gamma,
                      const float* beta, int n_cap, const int* dyn, int D, float* Y, int ld_y, void* stream);
int srec_bn_bwd(const float* dY, int ld_dy, const float* X, int ld_x, const float* mean, const float* var, float eps,
                const float* gamma, int training, int n_cap, const int* dyn, int D, float* dX, int ld_dx,
                float* dgamma, float* dbeta, float* ws, void* stream);
int srec_prelu_fwd(const float* X, int ld_x, const float* a, int n_cap, const int* dyn, int D, float* Y, int ld_y,
                   void* stream);
/* da NULL: the 32 chunk partials of d a stay in ws [32][D] for the caller to sum (srec_sum_slabs_multi, a 'tall' task) */
int srec_prelu_bwd(const float* dY, int ld_dy, const float* X, int ld_x, const float* a, int n_cap, const int* dyn,
                   int D, float* dX, int ld_dx, float* da, float* ws, void* stream);
/* EOPA: per node GRU over in-neighbours in edge-id order (lessr.py:20-27,35).  GI [Nsrc,3D] = ft W_ih^T + b_ih;
 * Whh [3D,D] as stored, WhhT [D,3D] k-major copy (read only when D > 32, else nullable: W_hh is held in registers);
 * gates [E,3D], Hprev/ghn [E,D], dGIe/dGHe [E,3D] by edge id. */
int srec_gru_seq_fwd(const float* GI, int ld_gi, const float* Whh, const float* WhhT, const float* bhh,
                     const int* in_ptr, const int* in_idx, const int* esrc, int n_cap, const int* dyn, int D,
                     float* neigh, int ld_n, float* gates, float* Hprev, float* ghn, void* stream);
int srec_gru_seq_bwd(const float* dneigh, int ld_dn, const float* Whh, const float* gates, const float* Hprev,
                     const float* ghn, const int* in_ptr, const int* in_idx, int n_cap, const int* dyn, int D,
                     float* dGIe, float* dGHe, void* stream);
/* SGAT: e = fc_e(sigmoid(q_u + k_v)), softmax over in-edges, sum a v_u (lessr.py:68-74).  A [E], dQe [E,Hh]. */
int srec_sgat_fwd(const float* Q, int ld_q, const float* K, int ld_k, const float* we, const float* Vf, int ld_v,
                  const int* in_ptr, const int* in_idx, const int* esrc, int n_cap, const int* dyn, int Hh, int Do,
                  float* A, float* out, int ld_o, void* stream);
int srec_sgat_bwd_dst(const float* dout, int ld_o, const float* Q, int ld_q, const float* K, int ld_k,
                      const float* we, const float* Vf, int ld_v, const float* A, const int* in_ptr,
                      const int* in_idx, const int* esrc, int n_cap, const int* dyn, int Hh, int Do, float* dQe,
                      float* dK, int ld_dk, float* dwe_part, int ld_dw, void* stream);
int srec_sgat_bwd_src(const float* dout, int ld_o, const float* A, const float* dQe, const int* out_ptr,
                      const int* out_idx, const int* edst, int n_cap, const int* dyn, int Hh, int Do, float* dQ,
                      int ld_dq, float* dVf, int ld_dv, void* stream);

/* ---- optimizer (adam.hip): torch.optim.Adam + coupled L2, train.py:70-75,101 ------------------------
 * hyper (device, 8 floats) = {lr/(1-b1^t), beta1, beta2, eps, weight_decay, 1-beta1, 1-beta2, sqrt(1-b2^t)} */
/* row-sharded item table: global ids (int64; -1 = padding) -> local row of the shard [lo, lo + n_loc) or -1 */
int srec_localize_idx(const long long* idx, long n, long lo, int n_loc, int* out, void* stream);
/* int32 ids (padded batches); zero (nullable): n - zero_from floats cleared by the same launch (the label-logit array of the sharded
 * forward: idx = (requested ids | labels) of one exchange, the labels from zero_from on) */
int srec_localize_idx32(const int* idx, long n, long lo, int n_loc, int* out, float* zero, long zero_from, void* stream);

/* inv[p] = u for the positions p = pos[ptr[u] .. ptr[u+1]) of item u < U (uniq_ptr / uniq_pos of a FlatBatch), -1 elsewhere */
int srec_inverse_index(const int* ptr, const int* pos, int U, int n, int* inv, void* stream);
/* row-sharded scoring: st [w, 2, B] = per-shard (log-sum-exp, label logit) gathered from the w ranks -> global lse [B],
 * label logit [B] and loss = mean over the LIVE sessions of (lse - lab).  lab_all (nullable, int64 [B]): the gathered
 * global labels, < 0 marking the capacity padding of a rank's batch (left out of the mean); gw (nullable, [B]):
 * d loss / d (lse_b - lab_b) = 1 / n_live on live sessions, 0 on padding - the ga / gc of srec_score_ce_bwd*. */
int srec_merge_stats(const float* st, int w, int B, const long long* lab_all, float* lse, float* lab, float* loss,
                     float* gw, void* stream);
int srec_merge_stats32(const float* st, int w, int B, const int* lab_all, float* lse, float* lab, float* loss,
                       float* gw, void* stream);        /* int32 labels */

int srec_adam_flat(float* p, const float* g, float* m, float* v, long n, const float* hyper, int use_wd,
                   void* stream);
/* step scalars on the DEVICE: *counter += 1, then hyper[8] = {lr/(1-b1^t), b1, b2, eps, wd, 1-b1, 1-b2, sqrt(1-b2^t)}
 * in double from cfg = double[5] {lr, beta1, beta2, eps, weight_decay} (torch.optim.Adam's host arithmetic,
 * train.py:70-75).  Makes a captured / pipelined optimizer step independent of host timing. */
int srec_adam_hyper(int* counter, const void* cfg, float* hyper, void* stream);
/* ... for n <= 16 (counter, cfg, hyper) slots in one launch; the three arguments are HOST arrays of n device pointers.
 * Loss tap (tap_ring nullable): the slot whose counter is tap_counter stores *tap_src - the step's loss, train.py:99-104
 * reads it after every step - into tap_ring[(steps taken before this one) % tap_n], so that a replayed step's loss survives
 * the next replay without a copy command between two graph launches.  skip (nullable, device int32: the err flag of
 * srec_copy_words_mailbox): non-zero = this step ran on a stale batch - the counters advance, the scalars written make every
 * Adam kernel of the step the identity (parameters and moments keep their bits), the loss slot reads NaN. */
int srec_adam_hyper_multi(int n, const void* counter, const void* cfg, const void* hyper, const int* tap_counter,
                          const float* tap_src, float* tap_ring, int tap_n, const int* skip, void* stream);
/* one launch for many small tensors (48 per launch): desc = HOST srec_adam_multi_desc below; the pointers travel by
 * value in the kernel arguments (nothing staged in device memory; a captured hipGraph bakes them into the node) */
typedef struct srec_adam_multi_desc {
    int nt;                    /* number of tensors */
    const int* use_wd;         /* [nt] apply the group's weight decay (0 for bias / batch_norm / activation, train.py:18) */
    const long* numel;         /* [nt] */
    float* const* p;           /* [nt] parameters ... */
    const float* const* g;     /* ... gradients ... */
    float* const* m;           /* ... exp_avg ... */
    float* const* v;           /* ... exp_avg_sq (device pointers, 4-B aligned; float4 path when all four are 16-B aligned) */
} srec_adam_multi_desc;
int srec_adam_multi(const void* desc, const float* hyper, void* stream);
/* item table, one wavefront per row.  max_norm > 0: Embedding(max_norm) renorm of the updated row (lessr.py:126,
 * msgifsr.py:162) - written back when renorm_write != 0, otherwise W keeps the plain Adam result (the reference
 * renormalises at the start of the NEXT forward) and only cs_out (nullable; = cs_scale / norm of the row as that forward
 * will see it: niser.py:151, msgifsr.py:279) reflects it.  dst16 (nullable; d <= 1024): bf16 copy [n, Dp] of the rows as
 * written - with renorm_write = 1 this pass leaves the table exactly as the next forward's stand-alone renorm + operand-copy
 * pass (srec_renorm_rows_bf16) would, which then need not run. */
int srec_adam_rows(float* W, const float* G, float* M, float* V, int n, int d, int ld, const float* hyper,
                   int use_wd, float max_norm, int renorm_write, float* cs_out, float cs_scale, int eps_mode,
                   float cs_eps, void* dst16, int Dp, void* stream);
int srec_adam_rows_proj(float* W, const float* G, float* M, float* V, int n, int d, int ld, const float* hyper, int use_wd,
                        float max_norm, int renorm_write, float* cs_out, float cs_scale, int eps_mode, float cs_eps,
                        const float* proj_cs, float proj_inv_scale, float* radial, void* dst16, int Dp, void* stream);

/* ---- MSGIFSR MSHGNN layer, all relations of both HeteroGraphConvs in one batched pass (hgat.hip) ------------------
 * Replaces msgifsr.py:70-89 (conv1(g) + conv2(reverse g), relation sum, head max, + session mean) around the fc GEMMs
 * of the GAT modules; `desc` points to a host srec_hg_desc (srec_hg.h).
 *   fwd: the caller has filled P[m] = x[rows of m] fc_m^T (p16 bit 3: and run srec_hg_fold); writes out[NT, D] = max_h(sum_rel rst + bias + n_rel x) +
 *        session mean of x, arg[NT, D] (winning head), and the saved A / eL / eR.
 *   bwd: g = d out; writes dx = n_rel g + session-mean term (the caller then accumulates dP[m] fc_m into it), dP[m],
 *        d_attn_l / d_attn_r / d_bias of every module.  ws: srec_hg_ws_floats() floats of scratch. */
int srec_hg_ws_floats(const void* desc, long* n_floats);
int srec_hg_fwd(const void* desc, const float* x, int ld_x, float* out, int ld_out, unsigned char* arg, void* stream);
/* the weight-only part of srec_hg_fwd on its own: V[m] (attention vectors folded into fc_m, gatconv.py:285-292 as a product over x)
 * and the per-type bias sums of desc.  srec_hg_fwd runs it itself unless bit 3 of desc.p16 says it was done since the weights
 * last changed - by this call or by the step's prologue launch: */
int srec_hg_fold(const void* desc, void* stream);
/* desc: HOST srec_step_prep_desc (srec_hg.h): fold + bf16 weight copies + GRU / head fragment copies + mailbox intake, one launch
 * in front of a step (each of them a ~5 us graph node of its own otherwise; utils/train.py:94-101 is the step) */
int srec_step_prep(const void* desc, void* stream);
/* feature-dropout glue of a layer call in one pass each (GATConv feat_drop, gatconv.py:268-283): masks from the counter-based
 * hash of srec_gather_rows_drop (one mask per conv) and cnt [2, rows] (relation instances of the conv into each row): ms = mask / (1-p),
 * xc = x * ms (the convs' dropped inputs), rm = cnt0 ms0 + cnt1 ms1, xres = x * rm (summed identity residuals);
 * backward: dx += (sum_s t[0][s]) * ms0 + (sum_s t[1][s]) * ms1 (the convs' masked data gradients, t [2, S, n]: S partial sums
 * per conv, one per GAT module projecting the row's node type). */
/* ... and, in the same launch, the attention-dropout multipliers (gatconv.py:300) mk [na] = 0 or 1 / (1 - pa) (pa = 0: none) */
int srec_hg_drop_prep(const float* x, const float* cnt, int rows, int D, float p, int seed, const int* counter, int salt,
                      float* ms, float* xc, float* rm, float* xres, float pa, long na, float* mk, void* stream);
/* ... and, in the same pass, xc16 [2, rows, D] = bf16(xc): the operand copy the bf16 projection GEMM reads */
int srec_hg_drop_prep16(const float* x, const float* cnt, int rows, int D, float p, int seed, const int* counter, int salt,
                        float* ms, float* xc, float* rm, float* xres, float pa, long na, float* mk, void* xc16, void* stream);
/* (ms of srec_hg_drop_prep / _prep16 is nullable: srec_hg_drop_merge with ms = NULL recomputes the masks from p / seed / counter /
 * salt, the mask tensor is then neither written nor read) */
int srec_hg_drop_merge(const float* t, int S, const float* ms, long n, float* dx, float p, int seed, const int* counter,
                       int salt, void* stream);
/* feature-dropout calls: desc.p16 bit 2 makes srec_hg_bwd leave dx alone; after the caller's backward-data GEMMs
 * srec_hg_pre_merge writes the whole d x in one pass (the pre-fill of srec_hg_bwd + srec_hg_drop_merge: one kernel, one pass
 * over dx less); t [2, S, NT, D] as for srec_hg_drop_merge, masks recomputed from desc.rm_* */
int srec_hg_pre_merge(const void* desc, const float* g, int ld_g, const float* t, int S, float* dx, int ld_dx, void* stream);
int srec_hg_bwd(const void* desc, const float* x, int ld_x, const float* g, int ld_g, const unsigned char* arg, float* dx,
                int ld_dx, float* ws, void* stream);

/* grouped, K-segmented bf16-operand GEMM (gemm_group_bf16.hip): up to 8 problems C_p [M,N] (+)= sum_s opA(A_ps) opB(B_ps)
 * in one launch.  desc: host srec_gemm_group (srec_hg.h).  mode 0: A [M,K], B [N,K] (nn.Linear forward); 1: A [M,K],
 * B [K,N] reduction-major, segments summed (backward-data over several modules); 2: A [K,M], B [K,N] both
 * reduction-major (weight gradient; dyn clamps the reduction).  dyn clamps the output rows in modes 0 / 1. */
int srec_gemm_group_bf16(const void* desc, int mode, void* stream);

/* grouped exact-fp32 GEMM (csrc/gemm.hip): desc = srec_gemm_f32_group (srec_hg.h), every problem as srec_gemm_f32.  One
 * launch for all tiles of all problems (+ one reduce launch when a long-K problem was k-split into slabs of ws).
 * Replaces the chain of cuBLAS calls of the attention read-out / session-vector head (msgifsr.py:127-146,272-279;
 * srgnn.py:73-88,123-127) and their backward. */
int srec_gemm_f32_group_run(const void* desc, float* ws, long ws_floats, void* stream);
/* ... with the slab sums of the plain split problems (alpha 1, beta 0, no bias, no row clamp: weight gradients) left
 * to the caller where the caller allows it (slab_n[p] != 0 ON ENTRY): slab_off[p] = float offset of problem p's slabs in ws (-1:
 * finished by the call), slab_n[p] = their count, each M N floats (dense), to be summed into C[p] (srec_sum_slabs_multi_ld when ldc != N).  ws must be
 * private to the call until that sum has run. */
int srec_gemm_f32_group_run_defer(const void* desc, float* ws, long ws_floats, long* slab_off, int* slab_n, void* stream);
/* bf16-in-HBM grouped GEMMs (gemm16.hip): the GAT fc projections and their backward (gatconv.py:166-175,282-283) with every
 * operand already stored as bf16 - LDS-DMA staging, no conversion pass.  desc: host srec_gemm16_group (srec_hg.h, up to 16
 * problems per launch).
 *   srec_gemm16_nt: C_p [M, N] (+)= sum_s A_ps [M, K] B_ps [N, K]^T; K % 32 == 0, lda / ldb % 8 == 0; c16 = bf16 output;
 *                   dyn clamps the output rows (rows past it: zeros when beta == 0 or c16).
 *   srec_gemm16_tn: C_p [M, N] = beta C_p + sum_s sum_{m < min(K, *dyn)} A_ps [m, M] B_ps [m, N]  (weight gradients: the
 *                   reduction runs over rows; both operands row-major as stored, M / N % 8 == 0), fp32 output.
 *   srec_rows_bf16: dst16 [n, d] = bf16(src [n, d]), zero rows past *dyn.
 *   srec_weights_bf16: n <= 8 matrices W_i [R_i, C_i] fp32 -> bf16 copy and transposed bf16 copy [C_i, R_i] in one launch;
 *                   W / W16 / WT16 are HOST arrays of n device pointers (WT16 entries may be NULL), R / Cc HOST int arrays. */
int srec_gemm16_nt(const void* desc, void* stream);
int srec_gemm16_tn(const void* desc, void* stream);
int srec_rows_bf16(const float* src, int ld, int n, const int* dyn, int d, void* dst16, void* stream);
/* out [C, n] = sum_r part [C, R, n] in fixed order (the row-split weight-gradient products of one module), n % 4 == 0 */
int srec_sum_slabs(const float* part, int C, int R, long n, float* out, void* stream);
int srec_weights_bf16(int n, const void* W, const void* W16, const void* WT16, const int* R, const int* Cc, void* stream);
/* MSGIFSR after its last MSHGNN layer in ONE launch each way (csrc/rowops.hip): F.normalize of every node row
 * (msgifsr.py:260-263), the per-session concatenation of all orders' nodes (msgifsr.py:135, perm = cat_perm of the FlatBatch)
 * and the last-node picks (filter_nodes, msgifsr.py:264): allf [n_cap, D], invr [n_cap] = 1 / norm, pout_k [B, D].
 * pick / pout / ld_p / g_pick / ld_gp / row0 / ncap / dyn_n are HOST arrays.  The backward writes EVERY row of dx [n_rows, D] (capacity padding as zeros). */
int srec_norm_perm_pick_fwd(const float* x, int ld_x, const int* perm, int n_cap, const int* dyn_t, int D, int eps_mode,
                            float eps, float* allf, float* invr, int npick, int B, const int* dyn_b, const void* pick,
                            const void* pout, const int* ld_p, void* stream);
int srec_norm_perm_pick_bwd(const float* allf, const float* invr, const float* g_allf, int ld_g, const int* perm, int n_cap,
                            const int* cat_seg, int B, const int* dyn_b, int D, int npick, const void* pick, const void* g_pick,
                            const int* ld_gp, float* dx, int ld_dx, int nt, const int* row0, const int* ncap, const void* dyn_n,
                            int n_rows, void* stream);
/* k-gram GRU of the SemanticExpander, all orders per launch (grux.hip; msgifsr.py:25,32-45): desc = host
 * srec_gru_step_desc (srec_hg.h).  d % 4 == 0 and 256 % (d / 4) == 0. */
int srec_gru_step_fwd(const void* desc, void* stream);
int srec_gru_step_bwd(const void* desc, void* stream);
/* forward of ALL time steps in one launch (csrc/gruf.hip; d = 128 / 256): desc = HOST srec_gru_fused_desc (srec_hg.h);
 * srec_gru_wfrag: n <= 8 GRU weights W_i [3 d, d] fp32 -> fragment-major bf16 copies dst_i [3 d d] (W, dst: HOST arrays of
 * device pointers), the B operands the fused forward streams */
int srec_gru_fused_fwd(const void* desc, void* stream);
int srec_gru_wfrag(int n, const void* W, const void* dst, int d, void* stream);
/* backward of all time steps in one launch (csrc/grufb.hip): desc = HOST srec_gru_fused_bwd_desc; srec_gru_wfrag_t: the
 * fragment-major weight copies for its backward-data products (B operand = W [3 d, d] itself, reduction over its rows) */
int srec_gru_fused_bwd(const void* desc, void* stream);
/* nodes per workgroup (16 / 32) both use for np problems of n[p] nodes; bias_part of the backward holds one row [6 d] per
 * workgroup: sum_p ceil(n[p] / nodes) rows */
int srec_gru_fused_nodes(int np, const int* n, int d, int* nodes);
/* mixed launches (round 6): with 16-node workgroups and every k[p] <= 4, the problems - in LAUNCH order: longest k first, as both
 * launchers sort them - whose workgroups hold 32 nodes instead, so that one launch is at most one workgroup per CU (bit p of
 * *mask; the shortest problems first, never those of the longest order - their 16-node kernel batches the input projections; 0 = none).  A workgroup of the fused kernels owns a CU and lives as long as its weight
 * stream whatever its node count: the bench batches (250 - 280 live 16-node tiles on 256 CUs) ran a second round on ~45 % of the
 * steps.  bias_part keeps one row per 16 nodes: a 32-node workgroup writes row 2 t and zeroes row 2 t + 1. */
int srec_gru_fused_wide(int np, const int* n, const int* k, int* mask);
/* waves per workgroup (4 / 8) of both and of the fragment-major weight copies for hidden size d */
int srec_gru_fused_waves(int d, int* waves);
int srec_gru_wfrag_t(int n, const void* W, const void* dst, int d, void* stream);
/* both copies of the same weights in one launch */
int srec_gru_wfrag_both(int n, const void* W, const void* dst_fwd, const void* dst_bwd, int d, void* stream);
/* out[p] [ncol] = column sums of part[p] [rows[p], ncol] for np <= 4 problems in one launch (the GRU bias gradients from the
 * per-block partial rows of srec_gru_step_bwd); part / out: HOST arrays of np device pointers, rows: HOST int array */
int srec_gru_bias_final(int np, const void* part, const int* rows, int ncol, const void* out, void* stream);
/* np <= 32 outputs out_i [n_i] = sum_r part_i [R_i, n_i] in one launch (row-split weight gradients); HOST arrays.  tall
 * (nullable HOST array): tall_i != 0 = few columns summed over hundreds of rows (the GRU bias partials of srec_gru_fused_bwd
 * / srec_gru_step_bwd, what srec_gru_bias_final does as a launch of its own).  Column tasks: n_i % 4 == 0, both pointers
 * 16-byte aligned; tall tasks are scalar (any n_i, any alignment).  With R_i = 1 a task is a copy: the row-sharded path fills
 * its gradient-bucket arenas with it (dist.VocabParallel._bucket_fill: gradients that did not land in their slot, zeros for
 * parameters without a gradient on this rank, the flag tail) - no torch.cat in the rank step. */
int srec_sum_slabs_multi(int np, const void* part, const int* R, const long* n, const void* out, const int* tall,
                         void* stream);
/* ... with strided destinations: w_i > 0 = out_i is a block of w_i columns in rows of stride ld_i floats (a column slice of a
 * weight gradient: the concat-free linear layers); w, ld: HOST int arrays, both or neither nullable */
int srec_sum_slabs_multi_ld(int np, const void* part, const int* R, const long* n, const void* out, const int* tall,
                            const int* w, const int* ld, void* stream);
/* ... and with the optimizer's step-scalar role (srec_adam_hyper_multi, arguments as there) in one more workgroup of the same launch:
 * the end-of-backward sums of a captured step and the Adam scalars need nothing of each other (one ~6 us graph node less) */
int srec_sum_slabs_multi_hyper(int np, const void* part, const int* R, const long* n, const void* out, const int* tall,
                               const int* w, const int* ld, int nh, const void* counter, const void* cfg, const void* hyper,
                               const int* tap_counter, const float* tap_src, float* tap_ring, int tap_n, const int* skip,
                               void* stream);

/* ---- evaluation: K best items per session without the (B, V) score matrix (topk.hip) -----------------------------
 * Replaces `logits = model(...); logits.topk(20)` of train.py:36-55 for models whose score is one soft-max
 * (ranking by z[b,v] = cs[v] * <sr_b, E_v>): out_val [B,K] descending, out_idx [B,K] int32 item ids, ties towards
 * the lower id.  K <= 32, d % 4 == 0; ws: srec_score_topk_ws() bytes. */
int srec_score_topk_ws(int B, int V, int K, long* bytes);
int srec_score_topk(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs, int B, int V, int d, int K,
                    float* out_val, int* out_idx, void* ws, void* stream);

/* ---- fused read-out head (headf.hip): msgifsr.py:124-155 (AttnReadout.forward) + :269-273 (fc_sr, F.normalize) for all live
 * orders in ONE launch, a group of SREC_HEAD_SESSIONS sessions per workgroup; replaces the {U, Vq} GEMM / srec_seg_attn_fwd /
 * {s} GEMM / split-K sum / srec_normalize_fwd chain of the grouped head in bf16 mode (d = 128 / 256).  desc: HOST
 * srec_head_desc (srec_hg.h).  srec_head_wfrag: n <= SREC_HEAD_MAXW fp32 matrices W_i [rows_i, cols_i] (HOST arrays of
 * device pointers / ints) -> hi / lo fragment-major bf16 operand copies dst_i [2 rows_i cols_i] of W_i (trans_i = 0) or W_i^T
 * (trans_i = 1), once per optimizer step. */
int srec_head_wfrag(int n, const void* W, const void* dst, const int* rows, const int* cols, const int* trans, void* stream);
int srec_head_fwd(const void* desc, void* stream);
/* per-session half of the head's backward (normalise-backward, d cat product, attention read-out backward); desc: HOST
 * srec_head_bwd_desc (srec_hg.h) */
int srec_head_bwd(const void* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
