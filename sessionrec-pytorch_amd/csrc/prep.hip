// The step's PROLOGUE launch: what a training / scoring step does before the first row of its batch is read - the operand
// copies of the weights the optimizer just wrote and the intake of the batch itself - as ROLES of one kernel.
//
// Every one of these was a launch of its own in front of the kernel that reads its result (weights_bf16 ahead of the GAT
// projections, gru_wfrag_both ahead of the k-gram GRU, head_wfrag ahead of the read-out head, hg_fold ahead of the attention
// logits, the mailbox copy ahead of the gather): 5 nodes of a captured step, 4.8 - 7.7 us each = 31.3 us, most of it the
// ~4.8 us a graph node costs whatever it does (graph branches serialise on this ROCm build, tools/graph_branch_probe.py).
// None of them depends on the batch or on each other - they depend on the weights only - so srec_step_prep runs them side
// by side, workgroup ranges of ONE launch: 15.5 us in the step (median of 60 replays; bound by the W16 role's 16 MB in +
// 16 MB out: 7.4 us warm / 11.2 us from HBM on its own, tools/prep_timing.py), step busy time 770 -> 758 us
// (profiles/r06_notes.md 7).  The single entry points below are the same kernel with one role filled.
//
//   role FOLD   attention vectors folded into the fc weights + per-type bias sums of an MSHGNN layer   (msgifsr.py:47-91,
//               gatconv.py:285-292; was hgat.hip)
//   role W16    bf16 / transposed bf16 copies of the GAT fc weights for the gemm16 products            (gatconv.py:282-283)
//   role GRU    fragment-major bf16 copies of the k-gram GRU weights, forward and backward layouts     (msgifsr.py:25,32-45)
//   role HEAD   hi / lo fragment-major copies of the read-out head's weights                           (msgifsr.py:124-155)
//   role BOX    batch intake through the in-graph mailbox                                              (utils/train.py:94-96)
#include "common.h"
#include "../../include/srec_hg.h"

extern "C" int srec_gru_fused_waves(int d, int* nw);      // gruf.hip: waves per workgroup of the fused GRU kernels at width d

namespace {

constexpr int MAXT = SREC_HG_MAXT, MAXM = SREC_HG_MAXM;
constexpr int GRU_MAXW = 2 * SREC_GRU_MAXP;
constexpr int HEAD_NW = 4;                 // waves per workgroup of headf.hip: wave w owns the columns [w d/4, (w+1) d/4)

// ------------------------------------------------------------------------------------------------ role FOLD
// el[n,h] = <P[n,h,:], a_l[h,:]> with P = x W^T  ==  x[n,:] . V_l[:,h],  V_l[c,h] = sum_j W[hD+j, c] a_l[hD+j].
// Folding the attention vectors into the fc weights first (2 MB of W per module, once) turns the logits into a
// [N, D] x [D, 2H] product over x instead of a pass over the 8x larger projections.  V[m] layout: [2][D][H].
struct FoldArgs {
    const float* W[MAXM]; const float* al[MAXM]; const float* ar[MAXM];
    float* V[MAXM];
    int H, D, blocks;
    // per node type t < nt: bsum[t][h D + c] = sum of the bias vectors of the relation instances into t (the workgroups
    // (m, h) with m < nt compute it on the side: hg_agg then reads ONE bias row per (node, head) instead of one per instance)
    int nt, tn[MAXT];
    const float* tb[MAXT][8];
    float* bsum[MAXT];
};

// block = (module, head, 64-column slice); 256 threads = 16 row groups x 16 column quads: a thread reads 16 bytes of 16 rows of
// W (one-column threads issued 64 four-byte loads per wave for the same bytes: the kernel was bound by their issue, 9.5 us for
// 16 MB), the attention vectors of the head wait in LDS; the 16 row groups are summed through LDS in row-group order.
// (One 1024-thread block per (module, head) - 64 workgroups - kept 3/4 of the CUs idle: 12 us.)
constexpr int FOLD_COLS = 64, MAXD_FOLD = 256;        // (D <= 256, D % 4 == 0: checked by fold_fill)
constexpr int FOLD_LDS = (2 * 16 * FOLD_COLS + 2 * MAXD_FOLD) * 4;
__device__ __forceinline__ void fold_role(const FoldArgs& a, const int bx, unsigned char* smem) {
    float (*red)[16][FOLD_COLS] = reinterpret_cast<float (*)[16][FOLD_COLS]>(smem);
    float (*av)[MAXD_FOLD] = reinterpret_cast<float (*)[MAXD_FOLD]>(smem + 2 * 16 * FOLD_COLS * 4);
    const int H = a.H, D = a.D;
    const int ncq = (D + FOLD_COLS - 1) / FOLD_COLS;
    const int cq = bx % ncq, mh = bx / ncq;
    const int m = mh / H, h = mh % H;
    for (int j = threadIdx.x; j < D; j += 256) { av[0][j] = a.al[m][h * D + j]; av[1][j] = a.ar[m][h * D + j]; }
    __syncthreads();
    const int c4 = threadIdx.x & 15, jg = threadIdx.x >> 4;
    const int c = cq * FOLD_COLS + 4 * c4;
    float4 sl = make_float4(0.f, 0.f, 0.f, 0.f), sr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) {
        const float* W = a.W[m] + (size_t)h * D * D + c;
#pragma unroll 8
        for (int j = jg; j < D; j += 16) {
            const float4 w = *reinterpret_cast<const float4*>(W + (size_t)j * D);
            const float l = av[0][j], r = av[1][j];
            sl.x += w.x * l; sl.y += w.y * l; sl.z += w.z * l; sl.w += w.w * l;
            sr.x += w.x * r; sr.y += w.y * r; sr.z += w.z * r; sr.w += w.w * r;
        }
    }
    *reinterpret_cast<float4*>(&red[0][jg][4 * c4]) = sl;
    *reinterpret_cast<float4*>(&red[1][jg][4 * c4]) = sr;
    __syncthreads();
    const int cl = threadIdx.x & (FOLD_COLS - 1), part = threadIdx.x / FOLD_COLS, cc = cq * FOLD_COLS + cl;
    if (part < 2 && cc < D) {                            // threads 0 .. 63: V_l, 64 .. 127: V_r
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[part][g][cl];
        a.V[m][(size_t)(part * D + cc) * H + h] = t;
    }
    if (part == 2 && cc < D) {
        // node types are dealt to the module workgroups round-robin: a tiny batch may have fewer live modules than types
        const int nmods = a.blocks / (H * ncq);
        for (int t = m; t < a.nt; t += nmods) {
            if (a.bsum[t] == nullptr) continue;
            float b = 0.f;
            for (int q = 0; q < a.tn[t]; ++q) b += a.tb[t][q][h * D + cc];  // instance order = the order hg_agg used
            a.bsum[t][h * D + cc] = b;
        }
    }
}

// the fold of one layer call from its descriptor (the fields srec_hg_fwd documents as the fold's: W, attn_l, attn_r, bias, V, Z,
// the instance topology); -> workgroups in f.blocks
int fold_fill(const srec_hg_desc* d, FoldArgs& f) {
    f = FoldArgs{};
    if (d == nullptr || d->H <= 0 || d->D <= 0 || (d->D & 3) || d->D > MAXD_FOLD || d->n_mods < 0 || d->n_mods > MAXM ||
        d->n_types <= 0 || d->n_types > MAXT || d->n_inst < 0 || d->n_inst > SREC_HG_MAXI)
        return SREC_BAD_ARG;
    if (d->n_mods == 0) return 0;
    f.H = d->H; f.D = d->D;
    for (int m = 0; m < d->n_mods; ++m) {
        f.W[m] = d->W[m]; f.al[m] = d->attn_l[m]; f.ar[m] = d->attn_r[m]; f.V[m] = d->V[m];
        if (f.W[m] == nullptr || f.al[m] == nullptr || f.ar[m] == nullptr || f.V[m] == nullptr) return SREC_BAD_ARG;
    }
    // bias sums per node type, kept in the first H D floats of the backward's Z scratch (free during the forward)
    f.nt = d->n_types;
    for (int t = 0; t < d->n_types; ++t) { f.tn[t] = 0; f.bsum[t] = d->Z[t]; }
    for (int i = 0; i < d->n_inst; ++i) {
        const int b = d->inst_dblk[i], m = d->inst_mod[i];
        if (b < 0 || b >= SREC_HG_MAXB || m < 0 || m >= d->n_mods) return SREC_BAD_ARG;
        const int t = d->blk_type[b];
        if (t < 0 || t >= d->n_types || f.tn[t] >= 8 || d->bias[m] == nullptr) return SREC_BAD_ARG;
        f.tb[t][f.tn[t]++] = d->bias[m];
    }
    // (the caller provides Z slots for max(n_mods, n_types): a batch of very short sessions has fewer live modules than types)
    for (int t = 0; t < d->n_types; ++t)
        if (f.tn[t] > 0 && f.bsum[t] == nullptr) return SREC_BAD_ARG;
    f.blocks = d->n_mods * d->H * cdiv(d->D, FOLD_COLS);
    return 0;
}

// ------------------------------------------------------------------------------------------------ role W16
// fc weights of up to 8 modules: W [R, Cc] fp32 -> W16 [R, Cc] and WT16 [Cc, R] bf16 (64 x 64 tiles through LDS)
struct W16Args {
    const float* W[8];
    unsigned short* W16[8];
    unsigned short* WT16[8];
    int R[8], Cc[8], start[9];
    int n;
};
constexpr int W16_LDS = 64 * 68 * 2;
__device__ __forceinline__ void w16_role(const W16Args& a, const int bx, unsigned char* smem) {
    unsigned short (*tile)[68] = reinterpret_cast<unsigned short (*)[68]>(smem);
    int t = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.n && bx >= a.start[i]) t = i;
    const int R = a.R[t], Cc = a.Cc[t];
    const int tc = (Cc + 63) / 64, b = bx - a.start[t];
    const int r0 = (b / tc) * 64, c0 = (b % tc) * 64;
    const float* __restrict__ W = a.W[t];
    if (((R | Cc) & 3) == 0) {
        // 4 columns per thread: float4 in, 8-byte bf16 stores in both layouts (2-byte stores ran at a third of this rate)
        const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
        for (int rr = y; rr < 64; rr += 16) {
            const int r = r0 + rr, c = c0 + 4 * x;
            uint2 v = make_uint2(0u, 0u);
            if (r < R && c < Cc) {
                const float4 f = *reinterpret_cast<const float4*>(W + (size_t)r * Cc + c);
                v = make_uint2(srec_pack_bf16(f.x, f.y), srec_pack_bf16(f.z, f.w));
                *reinterpret_cast<uint2*>(a.W16[t] + (size_t)r * Cc + c) = v;
            }
            *reinterpret_cast<uint2*>(&tile[rr][4 * x]) = v;
        }
        if (a.WT16[t] == nullptr) return;
        __syncthreads();
        for (int cc = y; cc < 64; cc += 16) {
            const int c = c0 + cc, r = r0 + 4 * x;
            if (c < Cc && r < R) {
                const unsigned lo = tile[4 * x][cc] | ((unsigned)tile[4 * x + 1][cc] << 16);
                const unsigned hi = tile[4 * x + 2][cc] | ((unsigned)tile[4 * x + 3][cc] << 16);
                *reinterpret_cast<uint2*>(a.WT16[t] + (size_t)c * R + r) = make_uint2(lo, hi);
            }
        }
        return;
    }
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    for (int rr = y; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + x;
        unsigned short v = 0;
        if (r < R && c < Cc) {
            v = srec_f2bf(W[(size_t)r * Cc + c]);
            a.W16[t][(size_t)r * Cc + c] = v;
        }
        tile[rr][x] = v;
    }
    __syncthreads();
    if (a.WT16[t] != nullptr)
        for (int cc = y; cc < 64; cc += 4) {
            const int c = c0 + cc, r = r0 + x;
            if (c < Cc && r < R) a.WT16[t][(size_t)c * R + r] = tile[x][cc];
        }
}

int w16_fill(int n, const void* W, const void* W16, const void* WT16, const int* R, const int* Cc, W16Args& a) {
    a = W16Args{};
    if (n <= 0) return 0;
    if (n > 8 || W == nullptr || W16 == nullptr || WT16 == nullptr || R == nullptr || Cc == nullptr) return SREC_BAD_ARG;
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i];
        a.W16[i] = ((unsigned short* const*)W16)[i];
        a.WT16[i] = ((unsigned short* const*)WT16)[i];
        a.R[i] = R[i]; a.Cc[i] = Cc[i];
        if (a.W[i] == nullptr || a.W16[i] == nullptr || R[i] <= 0 || Cc[i] <= 0) return SREC_BAD_ARG;
        a.start[i] = blocks;
        blocks += cdiv(R[i], 64) * cdiv(Cc[i], 64);
    }
    a.start[n] = blocks;
    return 0;
}

// ------------------------------------------------------------------------------------------------ role GRU
struct GruArgs {
    int d, jb, n, nbx;
    const float* W[GRU_MAXW];
    unsigned short* dstf[GRU_MAXW];
    unsigned short* dstb[GRU_MAXW];
};

// both fragment-major copies of a GRU weight [3 d, d]: z = 0 the forward layout (gruf.hip: fragment ((w KS + s) 3 JB + g JB + j),
// lane l <- W[g d + w d/4 + 32 j + (l & 31)][16 s + 8 (l >> 5) .. + 7]), z = 1 the backward-data layout (grufb.hip: fragment
// ((w KS3 + s) JB + j), lane l <- W[16 s + 8 (l >> 5) .. + 7][w 32 JB + 32 j + (l & 31)], the reduction runs over the 3 d gate rows)
__device__ __forceinline__ void gru_role(const GruArgs& a, const int bx, const int by, const int bz) {
    const int d = a.d, JB = a.jb;
    const int idx = bx * 256 + threadIdx.x;
    if (idx >= 3 * d * d / 8) return;
    const int lane = idx & 63, frag = idx >> 6;
    float v[8];
    unsigned short* dst;
    if (bz == 0) {
        const int KS = d / 16, NF = 3 * JB;
        const int f = frag % NF, ws = frag / NF, s = ws % KS, w = ws / KS, g = f / JB, j = f % JB;
        const int nrow = g * d + w * 32 * JB + 32 * j + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
        const float* src = a.W[by] + (size_t)nrow * d + kk;
        const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        dst = a.dstf[by];
    } else {
        const int KS3 = 3 * d / 16;
        const int j = frag % JB, ws = frag / JB, s = ws % KS3, w = ws / KS3;
        const int col = w * 32 * JB + 32 * j + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
        const float* src = a.W[by] + (size_t)kk * d + col;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(size_t)e * d];
        dst = a.dstb[by];
    }
    uint4 o;
    o.x = srec_pack_bf16(v[0], v[1]); o.y = srec_pack_bf16(v[2], v[3]);
    o.z = srec_pack_bf16(v[4], v[5]); o.w = srec_pack_bf16(v[6], v[7]);
    *reinterpret_cast<uint4*>(dst + (size_t)idx * 8) = o;
}

int gru_fill(int n, const void* W, const void* dst_fwd, const void* dst_bwd, int d, GruArgs& a) {
    a = GruArgs{};
    if (n <= 0) return 0;
    if (n > GRU_MAXW || W == nullptr || dst_fwd == nullptr || dst_bwd == nullptr || (d != 128 && d != 256)) return SREC_BAD_ARG;
    int nw = 4;
    if (int rc = srec_gru_fused_waves(d, &nw)) return rc;
    a.d = d; a.jb = d / (32 * nw); a.n = n; a.nbx = (3 * d * d / 8 + 255) / 256;
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i];
        a.dstf[i] = ((unsigned short* const*)dst_fwd)[i]; a.dstb[i] = ((unsigned short* const*)dst_bwd)[i];
        if (a.W[i] == nullptr || a.dstf[i] == nullptr || a.dstb[i] == nullptr) return SREC_BAD_ARG;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ role HEAD
struct HeadWfArgs {
    int n, nbx;
    const float* W[SREC_HEAD_MAXW];
    unsigned short* dst[SREC_HEAD_MAXW];
    int rows[SREC_HEAD_MAXW], cols[SREC_HEAD_MAXW], trans[SREC_HEAD_MAXW];
};

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = srec_pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = srec_pack_bf16(a - ah, b - bh);
}

// hi / lo fragment-major copy of an operand matrix M [N, K] (trans = 0: M = W [rows = N, cols = K] as stored; trans = 1:
// M = W^T of the stored W [rows = K, cols = N]): fragment (((w KS + s) 2 + t) JB + j), JB = N / 128, KS = K / 16, holds for lane l
// the 8 bf16 of t(M[w N/4 + 32 j + (l & 31)][16 s + 8 (l >> 5) .. + 7]), t = hi / lo
__device__ __forceinline__ void head_role(const HeadWfArgs& a, const int bx, const int m) {
    const int N = a.trans[m] ? a.cols[m] : a.rows[m], K = a.trans[m] ? a.rows[m] : a.cols[m];
    const int JB = N / (32 * HEAD_NW), KS = K / 16;
    const int idx = bx * 256 + threadIdx.x;              // (w, s, j, lane)
    if (idx >= N * K / 8) return;
    const int lane = idx & 63, f = idx >> 6;
    const int j = f % JB, ws = f / JB, s = ws % KS, w = ws / KS;
    const int nrow = w * 32 * JB + 32 * j + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
    float v[8];
    const float* W = a.W[m];
    if (!a.trans[m]) {
        const float4 v0 = *reinterpret_cast<const float4*>(W + (size_t)nrow * K + kk), v1 = *reinterpret_cast<const float4*>(W + (size_t)nrow * K + kk + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = W[(size_t)(kk + i) * N + nrow];
    }
    uint4 h, l;
    split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y); split2(v[4], v[5], h.z, l.z); split2(v[6], v[7], h.w, l.w);
    unsigned short* dst = a.dst[m] + ((((size_t)(w * KS + s) * 2) * JB + j) * 64 + lane) * 8;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + (size_t)JB * 512) = l;
}

int head_fill(int n, const void* W, const void* dst, const int* rows, const int* cols, const int* trans, HeadWfArgs& a) {
    a = HeadWfArgs{};
    if (n <= 0) return 0;
    if (n > SREC_HEAD_MAXW || W == nullptr || dst == nullptr || rows == nullptr || cols == nullptr || trans == nullptr) return SREC_BAD_ARG;
    a.n = n;
    int maxe = 0;
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i]; a.dst[i] = ((unsigned short* const*)dst)[i];
        a.rows[i] = rows[i]; a.cols[i] = cols[i]; a.trans[i] = trans[i];
        const int N = trans[i] ? cols[i] : rows[i], K = trans[i] ? rows[i] : cols[i];
        if (a.W[i] == nullptr || a.dst[i] == nullptr || N <= 0 || K <= 0 || (N % 128) || (K % 16)) return SREC_BAD_ARG;
        maxe = max(maxe, N * K / 8);
    }
    a.nbx = (maxe + 255) / 256;
    return 0;
}

// ------------------------------------------------------------------------------------------------ role BOX
// batch intake INSIDE a captured step: where this replay's batch lives is looked up in a mailbox the host fills ahead of the
// launch - entry (*counter % M) = {source address (2 words), words to copy, the counter value the host expects} in page-locked
// host memory - so a replayed step needs no copy command in front of its graph launch (the copy command + the gap behind it
// cost ~12 us per step, profiles/r03_notes.md "gap legs").  A mismatch of the expected counter raises *err (checked by the host).
struct BoxArgs {
    const int4* mailbox; int M; const int* counter;
    int4* dst; long cap4; int* err;
};
__device__ __forceinline__ void box_role(const BoxArgs& a, const int bx) {
    const int ctr = *a.counter;
    const int4 e = a.mailbox[ctr % a.M];
    const int4* src = reinterpret_cast<const int4*>(((unsigned long long)(unsigned)e.y << 32) | (unsigned long long)(unsigned)e.x);
    const long n4 = min((long)e.z / 4, a.cap4);
    if (e.w != ctr) {
        if (bx == 0 && threadIdx.x == 0) *a.err = 1;
        return;
    }
    const long i = (long)bx * 256 + threadIdx.x;
    if (i < n4) a.dst[i] = src[i];
}

int box_fill(const int* mailbox, int M, const int* counter, int* dst, long cap, int* err, BoxArgs& a) {
    a = BoxArgs{};
    if (cap <= 0) return 0;
    if (mailbox == nullptr || counter == nullptr || dst == nullptr || err == nullptr || M <= 0 || ((uintptr_t)dst & 15) ||
        ((uintptr_t)mailbox & 15))
        return SREC_BAD_ARG;
    a.mailbox = (const int4*)mailbox; a.M = M; a.counter = counter; a.dst = (int4*)dst; a.cap4 = cap / 4; a.err = err;
    return 0;
}

// ------------------------------------------------------------------------------------------------ the launch
// workgroup ranges, longest role first: [0, e_fold) FOLD, [.., e_w16) W16, [.., e_gru) GRU, [.., e_head) HEAD, [.., e_box) BOX
struct PrepArgs {
    int e_fold, e_w16, e_gru, e_head, e_box;
    FoldArgs f;
    W16Args w;
    GruArgs g;
    HeadWfArgs h;
    BoxArgs b;
};
static_assert(sizeof(PrepArgs) <= 4096, "kernel arguments are passed by value");

constexpr int PREP_LDS = FOLD_LDS > W16_LDS ? FOLD_LDS : W16_LDS;
__global__ __launch_bounds__(256) void step_prep_kernel(const PrepArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PREP_LDS];
    const int b = blockIdx.x;
    if (b < a.e_fold) {
        fold_role(a.f, b, smem);
    } else if (b < a.e_w16) {
        w16_role(a.w, b - a.e_fold, smem);
    } else if (b < a.e_gru) {
        const int r = b - a.e_w16;
        gru_role(a.g, r % a.g.nbx, (r / a.g.nbx) % a.g.n, r / (a.g.nbx * a.g.n));
    } else if (b < a.e_head) {
        const int r = b - a.e_gru;
        head_role(a.h, r % a.h.nbx, r / a.h.nbx);
    } else {
        box_role(a.b, b - a.e_head);
    }
}

int launch(PrepArgs& a, void* stream) {
    a.e_fold = a.f.blocks;
    a.e_w16 = a.e_fold + (a.w.n > 0 ? a.w.start[a.w.n] : 0);
    a.e_gru = a.e_w16 + a.g.nbx * a.g.n * 2;
    a.e_head = a.e_gru + a.h.nbx * a.h.n;
    a.e_box = a.e_head + (int)((a.b.cap4 + 255) / 256);
    if (a.e_box <= 0) return 0;
    hipLaunchKernelGGL(step_prep_kernel, dim3((unsigned)a.e_box), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ single entry points
// n <= 8 weight matrices W_i [R_i, C_i] fp32 (contiguous) -> bf16 copy W16_i and transposed bf16 copy WT16_i [C_i, R_i]
// (WT16 entries may be NULL).  W / W16 / WT16 / R / Cc are HOST arrays of n entries.
extern "C" int srec_weights_bf16(int n, const void* W, const void* W16, const void* WT16, const int* R, const int* Cc,
                                 void* stream) {
    PrepArgs a{};
    if (int rc = w16_fill(n, W, W16, WT16, R, Cc, a.w)) return rc;
    return launch(a, stream);
}

// srec_gru_wfrag and srec_gru_wfrag_t of the same n <= 8 weights in ONE launch (dst_fwd, dst_bwd: HOST arrays of device pointers)
extern "C" int srec_gru_wfrag_both(int n, const void* W, const void* dst_fwd, const void* dst_bwd, int d, void* stream) {
    PrepArgs a{};
    if (int rc = gru_fill(n, W, dst_fwd, dst_bwd, d, a.g)) return rc;
    return launch(a, stream);
}

// n <= SREC_HEAD_MAXW matrices W_i [rows_i, cols_i] fp32 row-major (HOST arrays) -> hi / lo fragment-major bf16 copies dst_i
// [2 rows_i cols_i] of W_i (trans_i = 0) or W_i^T (trans_i = 1) as the A operands of srec_head_fwd; operand rows % 128 == 0,
// operand columns % 16 == 0.  One launch.
extern "C" int srec_head_wfrag(int n, const void* W, const void* dst, const int* rows, const int* cols, const int* trans,
                               void* stream) {
    PrepArgs a{};
    if (int rc = head_fill(n, W, dst, rows, cols, trans, a.h)) return rc;
    return launch(a, stream);
}

// dst [<= cap words] (device) = the words mailbox[*counter % M] points at (page-locked host or device memory, 16-byte
// aligned, a multiple of 4 words); mailbox: M entries of 4 int32 in page-locked host memory (address lo, address hi, words,
// expected counter); err: device int32, set to 1 when the entry does not carry the counter's value
extern "C" int srec_copy_words_mailbox(const int* mailbox, int M, const int* counter, int* dst, long cap, int* err, void* stream) {
    PrepArgs a{};
    if (int rc = box_fill(mailbox, M, counter, dst, cap, err, a.b)) return rc;
    return launch(a, stream);
}

// the fold of an MSHGNN layer call on its own: V[m] and the per-type bias sums (Z[t][0 .. H D)) of desc (HOST srec_hg_desc);
// srec_hg_fwd runs it itself unless desc.p16 bit 3 says it was done (here or by srec_step_prep) since the weights last changed
extern "C" int srec_hg_fold(const void* desc, void* stream) {
    PrepArgs a{};
    if (int rc = fold_fill((const srec_hg_desc*)desc, a.f)) return rc;
    return launch(a, stream);
}

// every role that is present in ONE launch: desc = HOST srec_step_prep_desc (srec_hg.h); absent roles have n = 0 / NULL / cap = 0
extern "C" int srec_step_prep(const void* desc, void* stream) {
    const srec_step_prep_desc* q = (const srec_step_prep_desc*)desc;
    if (q == nullptr) return SREC_BAD_ARG;
    PrepArgs a{};
    if (q->hg != nullptr)
        if (int rc = fold_fill((const srec_hg_desc*)q->hg, a.f)) return rc;
    if (int rc = w16_fill(q->n_w16, q->w16_W, q->w16_out, q->w16_T, q->w16_R, q->w16_C, a.w)) return rc;
    if (int rc = gru_fill(q->n_gru, q->gru_W, q->gru_fwd, q->gru_bwd, q->gru_d, a.g)) return rc;
    if (int rc = head_fill(q->n_head, q->head_W, q->head_out, q->head_rows, q->head_cols, q->head_trans, a.h)) return rc;
    if (int rc = box_fill(q->mailbox, q->M, q->counter, q->box_dst, q->box_cap, q->box_err, a.b)) return rc;
    return launch(a, stream);
}
