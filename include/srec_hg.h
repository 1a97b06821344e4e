/* Layer descriptor of the batched MSHGNN (heterogeneous multi-head GAT) pass - see srec_hg_fwd / srec_hg_bwd in
 * srec.h and csrc/hgat.hip.  Plain C, host memory, read only during the call; every pointer inside is a DEVICE
 * pointer owned by the caller.  Mirrors msgifsr.py:47-91 (conv1 + conv2 over the reversed graph) with
 * gatconv.py:136-319 per relation.
 *
 *   node types   t < n_types : rows [row0[t], row0[t]+ncap[t]) of the stacked feature matrix x [NT, D]; live prefix
 *                              *dyn_n[t] (NULL = ncap); seg[t] = [B+1] offsets of each session's nodes of that type
 *   modules      m < n_mods  : one GATConv (conv, edge-type name): projection P[m] = x[rows of m] fc_m^T  [rows_m, H*D]
 *                              (computed by the caller's GEMM), attn_l / attn_r [H*D], bias [H*D]; gradients d*
 *   blocks       b < n_blocks: rows of type blk_type[b] inside module blk_mod[b]'s projection, starting at row
 *                              blk_row[b] of P[m] / dP[m]; eL/eR/wL/wR are [ncap, H] scratch per block
 *   instances    i < n_inst  : relation instance (conv, relation) of module inst_mod[i]: source block inst_sblk[i],
 *                              destination block inst_dblk[i] (its type receives the messages); CSR by destination
 *                              (in_ptr [Nd+1], in_idx -> edge id, esrc[e] source node) and by source (out_ptr, out_idx,
 *                              edst[e]), node ids local to the type; A / DP [E, H], der [Nd, H] scratch
 */
#ifndef SREC_HG_H
#define SREC_HG_H

#define SREC_HG_MAXT 4
#define SREC_HG_MAXM 8
#define SREC_HG_MAXB 16
#define SREC_HG_MAXI 16

typedef struct {
    int H, D, n_types, n_mods, n_blocks, n_inst, B;
    float slope;
    int p16;                       /* bit 0: P / dP hold bf16 (2-byte) elements (0: fp32); bit 1 (srec_hg_bwd): leave the dP rows past
                                      the live count unwritten - the caller's readers of dP stop at *dyn_n (srec_gemm16_group.dyn);
                                      bit 2 (srec_hg_bwd): do not write dx - the caller finishes it with srec_hg_pre_merge;
                                      bit 3 (srec_hg_fwd): V / the bias sums in Z are current (srec_hg_fold or srec_step_prep ran) */
    const int* dynB;
    /* node types */
    int row0[SREC_HG_MAXT], ncap[SREC_HG_MAXT];
    const int* dyn_n[SREC_HG_MAXT];
    const int* seg[SREC_HG_MAXT];
    /* modules */
    const void* P[SREC_HG_MAXM];
    void* dP[SREC_HG_MAXM];
    const float* W[SREC_HG_MAXM];      /* fc.weight [H*D, D] fp32 */
    float* V[SREC_HG_MAXM];            /* scratch [2][D][H]: attention vectors folded into W (forward) */
    float* Z[SREC_HG_MAXM];            /* scratch [2][H][D]: x^T wL / x^T wR (backward); during the forward Z[t][0 .. H D) holds the summed bias rows of node type t */
    const float* attn_l[SREC_HG_MAXM];
    const float* attn_r[SREC_HG_MAXM];
    const float* bias[SREC_HG_MAXM];
    float* d_attn_l[SREC_HG_MAXM];
    float* d_attn_r[SREC_HG_MAXM];
    float* d_bias[SREC_HG_MAXM];
    /* dropout (all NULL when off): xin[m] = the feature-dropped input matrix module m projects ([NT, D], same row
     * layout as x); xres [NT, D] = residual rows already summed over the instances; rm [NT, D] = their per-element
     * scale (the d x factor of g); Mk[i] [E, H] = attention-dropout mask of instance i, 0 or 1/(1-p) */
    const float* xin[SREC_HG_MAXM];
    const float* xres;
    const float* rm;
    /* projection blocks */
    int blk_mod[SREC_HG_MAXB], blk_type[SREC_HG_MAXB], blk_row[SREC_HG_MAXB];
    float* eL[SREC_HG_MAXB];
    float* eR[SREC_HG_MAXB];
    float* wL[SREC_HG_MAXB];
    float* wR[SREC_HG_MAXB];
    /* relation instances */
    int inst_mod[SREC_HG_MAXI], inst_sblk[SREC_HG_MAXI], inst_dblk[SREC_HG_MAXI];
    const int* in_ptr[SREC_HG_MAXI];
    const int* in_idx[SREC_HG_MAXI];
    const int* esrc[SREC_HG_MAXI];
    const int* out_ptr[SREC_HG_MAXI];
    const int* out_idx[SREC_HG_MAXI];
    const int* edst[SREC_HG_MAXI];
    float* A[SREC_HG_MAXI];
    float* DP[SREC_HG_MAXI];
    float* der[SREC_HG_MAXI];
    const float* Mk[SREC_HG_MAXI];
    /* per (node type, session) mean of the layer's input rows [B, D] and the session of every stacked row [sum ncap] - scratch
     * written by srec_hg_fwd (msgifsr.py:86-89 segment_mean + broadcast), read again by srec_hg_bwd */
    float* smean[SREC_HG_MAXT];
    int* sess;
    /* feature dropout without a stored rm (rm == NULL, rm_cnt != NULL): srec_hg_bwd recomputes rm[row, c] = cnt[0][row] m0 +
     * cnt[1][row] m1 from the masks' hash - cnt [2, NT] and (p, seed, counter, salt) as given to srec_hg_drop_prep */
    const float* rm_cnt;
    const int* rm_counter;
    float rm_p;
    int rm_seed, rm_salt;
} srec_hg_desc;

/* problem table of srec_gemm_group_bf16 (srec.h) */
#define SREC_GG_MAXP 8
#define SREC_GG_MAXS 4
typedef struct {
    int np, lda, ldb, ldc;
    float beta;
    int a16, c16;                  /* A operands / C outputs are bf16 (2-byte) instead of fp32 */
    int M[SREC_GG_MAXP], N[SREC_GG_MAXP], K[SREC_GG_MAXP], nseg[SREC_GG_MAXP];
    const void* A[SREC_GG_MAXP][SREC_GG_MAXS];
    const float* B[SREC_GG_MAXP][SREC_GG_MAXS];
    void* C[SREC_GG_MAXP];
    const int* dyn[SREC_GG_MAXP];
} srec_gemm_group;


/* problem table of srec_gemm16_nt / srec_gemm16_tn (srec.h): up to 16 problems, every operand bf16 in device memory */
#define SREC_G16_MAXP 16
#define SREC_G16_MAXS 4
typedef struct {
    int np, lda, ldb, ldc;
    float beta;
    int c16;                       /* bit 0: C outputs are bf16 (nt only); bit 1: leave output rows past *dyn unwritten */
    int M[SREC_G16_MAXP], N[SREC_G16_MAXP], K[SREC_G16_MAXP], nseg[SREC_G16_MAXP];
    const void* A[SREC_G16_MAXP][SREC_G16_MAXS];
    const void* B[SREC_G16_MAXP][SREC_G16_MAXS];
    void* C[SREC_G16_MAXP];
    const int* dyn[SREC_G16_MAXP];
    int koff[SREC_G16_MAXP];       /* tn only: this problem reduces rows koff .. koff + K of a longer operand whose live
                                      row count is *dyn: live rows here = clamp(*dyn - koff, 0, K) (row-split products) */
    int nsplit[SREC_G16_MAXP];     /* tn only (0 / 1 = off): split the reduction rows into nsplit pieces (64-row multiples), piece
                                      s writing its own slab C + s * M * ldc (caller sums the slabs: srec_sum_slabs_multi) */
    int lda_p[SREC_G16_MAXP], ldb_p[SREC_G16_MAXP], ldc_p[SREC_G16_MAXP];   /* per-problem leading dimensions; 0 = the group's */
    int mhint[SREC_G16_MAXP];      /* nt: expected live rows (*dyn) of a capacity-padded problem, 0 = unknown: tile shape and tile
                                      order are chosen for the live work, not for the capacity */
} srec_gemm16_group;

/* problem table of srec_gemm_f32_group_run (srec.h, csrc/gemm.hip): up to 16 independent exact-fp32 products, each with
 * the operand conventions of srec_gemm_f32 (C = alpha A B^T + beta C + bias; per operand one unit stride; dyn_mode 1 clamps
 * M, 2 clamps K).  nsplit and ws are chosen / filled by the launcher.  split3 != 0: the products run on the bf16 matrix
 * pipe as three terms of a hi / lo split of both operands (hi hi + lo hi + hi lo, split in registers on the way from LDS to
 * the MFMA): results to ~2^-17 relative of the fp32 product at 1/5 of the matrix-pipe time - the bf16 mode's read-out head
 * backward; 0: v_mfma_f32_32x32x2_f32, exact. */
#define SREC_GEMM32_MAXP 16
typedef struct {
    int np;
    const float* A[SREC_GEMM32_MAXP]; int a_rs[SREC_GEMM32_MAXP], a_cs[SREC_GEMM32_MAXP];
    const float* B[SREC_GEMM32_MAXP]; int b_rs[SREC_GEMM32_MAXP], b_cs[SREC_GEMM32_MAXP];
    float* C[SREC_GEMM32_MAXP]; int ldc[SREC_GEMM32_MAXP];
    const float* bias[SREC_GEMM32_MAXP];
    int M[SREC_GEMM32_MAXP], N[SREC_GEMM32_MAXP], K[SREC_GEMM32_MAXP];
    const int* dyn[SREC_GEMM32_MAXP]; int dyn_mode[SREC_GEMM32_MAXP];
    float alpha[SREC_GEMM32_MAXP], beta[SREC_GEMM32_MAXP];
    int nsplit[SREC_GEMM32_MAXP];
    float* ws;
    int split3;
} srec_gemm_f32_group;

/* one time step of the k-gram GRU (msgifsr.py:25,32-45) for up to 4 orders at once: srec_gru_step_fwd / _bwd (srec.h,
 * csrc/grux.hip).  Problem p = order k[p] with n[p] nodes (live prefix *dyn[p]), hidden size d, at time step t[p].
 * x rows are node-major: row (node * k + t). */
#define SREC_GRU_MAXP 4
typedef struct {
    int np, d;
    int n[SREC_GRU_MAXP], k[SREC_GRU_MAXP], t[SREC_GRU_MAXP];
    const int* dyn[SREC_GRU_MAXP];
    /* forward: GI [n k, 3 d] = x W_ih^T (no bias), GH [n, 3 d] = h_{t-1} W_hh^T (no bias; NULL at t = 0), Hp = h_{t-1};
     * writes Hn [n, d] (+ bf16 copy Hn16, nullable), gates [n, 4 d] = r, z, n, gh_n + b_hh_n; when out != NULL and
     * t == k - 1 also out [n, d] = 0.5 mean_t X[n, t, :] + 0.5 Hn (X [n, k, d]) */
    const float* GI[SREC_GRU_MAXP];
    const float* GH[SREC_GRU_MAXP];
    const float* bih[SREC_GRU_MAXP];
    const float* bhh[SREC_GRU_MAXP];
    const float* Hp[SREC_GRU_MAXP];
    float* Hn[SREC_GRU_MAXP];
    void* Hn16[SREC_GRU_MAXP];
    float* gates[SREC_GRU_MAXP];
    const float* X[SREC_GRU_MAXP];
    float* out[SREC_GRU_MAXP];
    /* backward: d h_t comes from dH [n, d], or (t == k - 1 and dout != NULL) as 0.5 dout with the mean term 0.5 dout / k
     * written to all k rows of dX [n k, d]; writes dGI16 rows (node k + t) [n k, 3 d] and dGH16 [n, 3 d] (bf16, nullable),
     * dHp [n, d] = d h_t * z (nullable: the direct term of d h_{t-1}), and one row of column sums (d gi | d gh) [6 d] per
     * max(8, 1024 / d)-node block into bias_part at row part_row0 + block */
    const float* dH[SREC_GRU_MAXP];
    const float* dout[SREC_GRU_MAXP];
    void* dGI16[SREC_GRU_MAXP];
    void* dGH16[SREC_GRU_MAXP];
    float* dHp[SREC_GRU_MAXP];
    float* dX[SREC_GRU_MAXP];
    float* bias_part[SREC_GRU_MAXP];
    int part_row0[SREC_GRU_MAXP];
} srec_gru_step_desc;

/* the whole k-gram GRU forward (all time steps, up to 4 orders) in one launch: srec_gru_fused_fwd (srec.h, csrc/gruf.hip;
 * msgifsr.py:25,32-45), d = 128 or 256.  Problem p = order k[p] with n[p] nodes (live prefix *dyn[p]); x rows node-major
 * (node * k + t).  Reads X [n k, d] fp32 and the fragment-major bf16 weights of srec_gru_wfrag; writes X16 [n k, d] (bf16
 * copy of X), H [k, n, d] fp32 hidden states, H16 [k - 1, n, d] their bf16 copies (steps 0 .. k - 2), gates [k, n, 4 d] fp16 =
 * r, z, n, gh_n + b_hh_n (live rows), out [n, d] = 0.5 mean_t X[n, t, :] + 0.5 h_{k-1}; rows past the live prefix: H, H16,
 * out = 0. */
typedef struct {
    int np, d;
    int n[SREC_GRU_MAXP], k[SREC_GRU_MAXP];
    const int* dyn[SREC_GRU_MAXP];
    const float* X[SREC_GRU_MAXP];
    void* X16[SREC_GRU_MAXP];
    const void* Wih_f[SREC_GRU_MAXP];
    const void* Whh_f[SREC_GRU_MAXP];
    const float* bih[SREC_GRU_MAXP];
    const float* bhh[SREC_GRU_MAXP];
    float* H[SREC_GRU_MAXP];
    void* H16[SREC_GRU_MAXP];
    void* gates[SREC_GRU_MAXP];              /* fp16 */
    float* out[SREC_GRU_MAXP];
} srec_gru_fused_desc;

/* the whole k-gram GRU backward except the weight gradients, in one launch: srec_gru_fused_bwd (srec.h, csrc/grufb.hip), d = 128
 * or 256; problems as srec_gru_fused_desc.  Reads the saved gates [k, n, 4 d] (fp16) and H [k, n, d], dout [n, d] (gradient of the
 * expander output) and the fragment-major weights of srec_gru_wfrag_t; writes dGI16 [n k, 3 d] (row node k + t) and dGH16
 * [k - 1, n, 3 d] (slot t - 1), bf16 operands of the weight-gradient GEMMs, dX [n k, d] = 0.5 dout / k + d(gi) W_ih, and one
 * row [6 d] = column sums (d gi | d gh) per 32-node workgroup into bias_part at row part_row0 + workgroup.  Rows past the live
 * prefix: zeros. */
typedef struct {
    int np, d;
    int n[SREC_GRU_MAXP], k[SREC_GRU_MAXP];
    const int* dyn[SREC_GRU_MAXP];
    const void* gates[SREC_GRU_MAXP];        /* fp16, as srec_gru_fused_fwd saved them */
    const float* H[SREC_GRU_MAXP];
    const float* dout[SREC_GRU_MAXP];
    const void* Wih_f[SREC_GRU_MAXP];
    const void* Whh_f[SREC_GRU_MAXP];
    void* dGI16[SREC_GRU_MAXP];
    void* dGH16[SREC_GRU_MAXP];
    float* dX[SREC_GRU_MAXP];
    float* bias_part[SREC_GRU_MAXP];
    int part_row0[SREC_GRU_MAXP];
} srec_gru_fused_bwd_desc;

/* ---- fused read-out head (csrc/headf.hip): AttnReadout + fc_sr + F.normalize of MSGIFSR for a group of sessions per
 * workgroup (/root/reference/src/models/msgifsr.py:124-155 AttnReadout.forward, :269-273 fc_sr / normalize), up to
 * SREC_HEAD_MAXH live orders ("heads") per launch, d = hidden = output = 128 or 256, bf16 mode (3-term hi / lo split products:
 * fp32-grade results).  Per head h:
 *   cat[h]   [B, 2 d] fp32: left half = the query rows v_b (in), right half = the read-out rows g_b (out)
 *   Wu_f / Wv_f / Wsr_f: hi / lo fragment-major copies of fc_u [d, d], fc_v [d, d], fc_sr [d, 2 d] (srec_head_wfrag, trans = 0)
 *   bu (nullable) [d], we [d]: fc_u bias, fc_e weight
 *   alpha [NT] soft-max weights (out), U (nullable) [NT, d] = x Wu^T + bu (out, only for the grouped backward),
 *   Vq (nullable) [B, d] (out), y [B, d] = normalised session vectors (out), inv [B] = 1 / max(|s|, eps) resp. 1 / (|s| + eps)
 *   (out), y16 (nullable) [B, ld16] bf16 copy of y (the scoring kernels' operand)
 * X = allf [NT, d] (row stride ld_x; NT = row capacity), seg [B + 1] = first row of every session (seg[live B] = live rows),
 * dynB (nullable) = live sessions; sessions past it get zero rows.  A workgroup owns the sessions that START in its window of
 * SREC_HEAD_WINDOW rows, SREC_HEAD_SESSIONS of them per pass, their rows in chunks of SREC_HEAD_ROWS.  A session may have at most SREC_MAX_SESSION_NODES rows (srec_limits). */
#define SREC_HEAD_SESSIONS 16
#define SREC_HEAD_ROWS 64
#define SREC_HEAD_WINDOW 32
#define SREC_HEAD_MAXH 4
#define SREC_HEAD_MAXW 16
typedef struct {
    int nh, d, B, NT, ld_x, ld16, eps_mode;
    float eps;
    const float* X;
    const int* seg;
    const int* dynB;
    float* cat[SREC_HEAD_MAXH];
    const void* Wu_f[SREC_HEAD_MAXH];
    const void* Wv_f[SREC_HEAD_MAXH];
    const void* Wsr_f[SREC_HEAD_MAXH];
    const float* bu[SREC_HEAD_MAXH];
    const float* we[SREC_HEAD_MAXH];
    float* alpha[SREC_HEAD_MAXH];
    float* U[SREC_HEAD_MAXH];
    float* Vq[SREC_HEAD_MAXH];
    float* y[SREC_HEAD_MAXH];
    float* inv[SREC_HEAD_MAXH];
    void* y16[SREC_HEAD_MAXH];
} srec_head_desc;

/* the per-session half of the head's backward in one launch (srec_head_bwd, csrc/headf.hip): normalise-backward, d cat = g_s Wsr
 * and the attention read-out backward (msgifsr.py:139-146 under autograd).  Per head: gy [B, d] (row stride ld_gy) = gradient of
 * the normalised session vectors, y / inv / alpha / U / Vq as srec_head_fwd wrote them, WsrT_f = hi / lo fragment-major copy of
 * fc_sr^T [2 d, d] (srec_head_wfrag, trans = 1), we [d].  Out: gs [B, d] = gradient of s, gcat [B, 2 d] = gs Wsr (left half:
 * direct part of d v, right half: d read-out), dX [NT, d] = alpha_i * d read-out of the row's session, dU [NT, d], dVq [B, d],
 * dwp [B, d] = per-session sums of d e sigma (d fc_e = their column sum); rows / sessions past the live counts: zeros. */
typedef struct {
    int nh, d, B, NT, ld_x, ld_gy;
    const float* X;
    const int* seg;
    const int* dynB;
    const float* gy[SREC_HEAD_MAXH];
    const float* y[SREC_HEAD_MAXH];
    const float* inv[SREC_HEAD_MAXH];
    const void* WsrT_f[SREC_HEAD_MAXH];
    const float* alpha[SREC_HEAD_MAXH];
    const float* U[SREC_HEAD_MAXH];
    const float* Vq[SREC_HEAD_MAXH];
    const float* we[SREC_HEAD_MAXH];
    float* gs[SREC_HEAD_MAXH];
    float* gcat[SREC_HEAD_MAXH];
    float* dX[SREC_HEAD_MAXH];
    float* dU[SREC_HEAD_MAXH];
    float* dVq[SREC_HEAD_MAXH];
    float* dwp[SREC_HEAD_MAXH];
} srec_head_bwd_desc;

/* the prologue of a step in ONE launch (srec_step_prep, csrc/prep.hip): the operand copies of the weights the optimizer just wrote
 * and the intake of the batch - the work of srec_hg_fold, srec_weights_bf16, srec_gru_wfrag_both, srec_head_wfrag and
 * srec_copy_words_mailbox (srec.h; arguments as documented there, HOST arrays of n entries) as workgroup ranges of one kernel.
 * None of them reads the batch or another one's output.  A role is absent with hg = NULL / n = 0 / box_cap = 0. */
typedef struct {
    const void* hg;                    /* HOST srec_hg_desc whose fold is wanted (the caller then sets bit 3 of ITS p16 for srec_hg_fwd) */
    int n_w16;                         /* srec_weights_bf16 */
    const void* w16_W;
    const void* w16_out;
    const void* w16_T;
    const int* w16_R;
    const int* w16_C;
    int n_gru, gru_d;                  /* srec_gru_wfrag_both */
    const void* gru_W;
    const void* gru_fwd;
    const void* gru_bwd;
    int n_head;                        /* srec_head_wfrag */
    const void* head_W;
    const void* head_out;
    const int* head_rows;
    const int* head_cols;
    const int* head_trans;
    const int* mailbox;                /* srec_copy_words_mailbox */
    int M;
    const int* counter;
    int* box_dst;
    long box_cap;
    int* box_err;
} srec_step_prep_desc;

#endif
