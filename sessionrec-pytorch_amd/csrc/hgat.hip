// MSGIFSR's MSHGNN layer as ONE batched pass over all relations of both HeteroGraphConvs
// (msgifsr.py:47-91 = conv1(g) + conv2(reverse g), sum over relations, max over heads, + session mean;
// gatconv.py:267-311 per relation).  The per-relation formulation in gat.hip launches ~18 kernels per relation
// instance (14 instances at order 3) and materialises a [N, H*D] result per instance; here the whole layer is
//   forward : srec_hg_fwd  = 2 launches  (attention logits of every projection block; fused edge-softmax +
//             aggregation over EVERY relation into a destination + bias + residual + head-max + session mean)
//   backward: srec_hg_bwd  = 5 launches  (d x pre-fill; per-destination score gradients; per-source projection
//             gradients; two-stage column sums for attn_l / attn_r / bias of every module)
// around the fc GEMMs of the 8 GAT modules.  Nothing of size [N, H*D] is written except the projection
// gradients the backward GEMMs consume.  All reductions are gather-style and ordered: deterministic, no atomics.
//
// Layout: node features of all types stacked [NT, D]; type t owns rows [row0[t], row0[t] + ncap[t]) with a live
// prefix dyn_n[t].  Module m's projection P[m] = x[rows of m] W_m^T is [rows_m, H*D]; a projection BLOCK is the
// row range of one type inside one module's projection (the shared 'inter' module projects every type at once).
// A relation INSTANCE (conv, relation) reads its source block and writes into its destination type.
// Grid geometry (H = 8, D % 8 == 0; round 3): ONE WAVEFRONT PER NODE with all heads in the lane for the fused aggregation
// (hg_agg_node_kernel) and the score gradients (hg_bwd_dst_node_kernel: all relation instances of the node side by side),
// one wavefront per (projection block, node) for the projection gradients (hg_bwd_src_kernel), the attention logits on the
// fp32 matrix pipe (hg_dots_kernel).  These kernels are bound by memory latency x occupancy (a dependent load costs 1.2 - 1.7 us
// behind the cold per-XCD L2s of a fresh kernel), so they are built for few dependent hops and many resident waves.  The
// round-2 shapes - an 8-wave workgroup per destination node (wave = head, head-max through 8 KB of LDS), a wavefront per
// (instance, destination) - remain as the fallbacks for other H / D.
#include "common.h"
#include "../../include/srec_hg.h"

extern "C" int srec_hg_fold(const void* desc, void* stream);      // prep.hip

namespace {

constexpr int WPB = 4;
constexpr int MAXDEG = SREC_MAX_DEGREE;
constexpr int MAXH = 8;
constexpr int NCHUNK_C = 32;  // row chunks of the column-thread sums (hg_colsum_cols_kernel)
constexpr int SEGCAP = 1025;   // session offsets staged in LDS by hg_agg / hg_pre (batches of up to 1024 sessions)
constexpr int NCHUNK = 16;
constexpr int MAXT = SREC_HG_MAXT, MAXM = SREC_HG_MAXM, MAXB = SREC_HG_MAXB, MAXI = SREC_HG_MAXI;

// projection element type: fp32, or bf16 (unsigned short) when the GEMMs run on bf16 operands anyway - halves the
// traffic of every pass over the [N, H*D] projections and their gradients
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const unsigned short* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const unsigned short* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(unsigned short* p, float4 v) {
    const f32x4_t f = {v.x, v.y, v.z, v.w};
    const bf16x4_t b = __builtin_convertvector(f, bf16x4_t);      // v_cvt_pk_bf16_f32 (RNE)
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, b);
}

template <int N>
__device__ __forceinline__ int find_range(const int (&start)[N], int n, int x) {
    int b = 0;
#pragma unroll
    for (int i = 1; i < N; ++i)
        if (i < n && x >= start[i]) b = i;
    return b;
}

// ------------------------------------------------------------------------------------------------ logits
// el[n,h] = <P[n,h,:], a_l[h,:]> with P = x W^T  ==  x[n,:] . V_l[:,h],  V_l[c,h] = sum_j W[hD+j, c] a_l[hD+j]: the attention
// vectors are folded into the fc weights first (V[m] [2][D][H] and the per-type bias sums: srec_hg_fold, a role of the step's
// prologue launch, prep.hip), which turns the logits into a [N, D] x [D, 2H] product over x instead of a pass over the 8x
// larger projections.
struct DotsArgs {
    const float* V[MAXB];
    float* eL[MAXB]; float* eR[MAXB];
    const int* dyn[MAXB];
    int ncap[MAXB], row0[MAXB];
    int start[MAXB + 1];     // first thread block of each projection block
    int nb, H, D;
    const float* x[MAXB]; int ld_x;          // the module's (possibly feature-dropped) input rows
    // workgroups behind the projection blocks: (node type t, session b) - the mean of the session's INPUT rows of that type
    // (msgifsr.py:86-89) and the session of every node, computed once per session here instead of once per NODE inside hg_agg
    // (where the session search + the row loop were 8.6 k of a workgroup's 20 k cycles, tools/hg_timing.py)
    int nt, B; const int* dynB;
    const int* seg[MAXT]; int trow0[MAXT];
    const float* xm;                         // layer input rows (not dropped)
    float* smean[MAXT]; int* sess;
};

// workgroup = 64 nodes of a projection block x 16 outputs (l/r x head), one 16-node tile per wave on the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation): the block's folded vectors V sit in LDS, transposed to
// [output][D + 4] so a lane reads 4 consecutive k of its output as one conflict-free 16-byte read; a lane (node l & 15, k-slot
// l >> 4) reads 16 bytes of its node row per 16-k block - both operands use the k-order (16 j + 4 (l >> 4) + t), which a
// reduction does not care about.  A [N, D] x [D, 16] product: 16 row loads + 16 LDS reads + 64 MFMAs per wave and 16 nodes
// (round 2's thread-per-output loop: 64 + 64 + 256 FMA per thread, 16 nodes per workgroup and a 16-KB V staging each: 25 us).
#ifndef SREC_DOTS_NODES
#define SREC_DOTS_NODES 64
#endif
constexpr int DOTS_NODES = SREC_DOTS_NODES;
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hg_dots_kernel(DotsArgs a) {
    extern __shared__ float vt[];                              // [16][D + 4] + 16 (a k tail past D reads finite values: its A side is 0)
    if ((int)blockIdx.x >= a.start[a.nb]) {
        const int e = (int)blockIdx.x - a.start[a.nb], t = e / a.B, sb = e - t * a.B;
        if (t >= a.nt || sb >= dyn_count(a.dynB, a.B)) return;
        const int s0 = a.seg[t][sb], s1 = a.seg[t][sb + 1];
        for (int j = s0 + (int)threadIdx.x; j < s1; j += 256) a.sess[a.trow0[t] + j] = sb;
        for (int c = threadIdx.x; c < a.D; c += 256) {
            const float* xp = a.xm + (size_t)a.trow0[t] * a.ld_x + c;
            float m = 0.f;
            int j = s0;
            for (; j + 3 < s1; j += 4) {
                const float x0 = xp[(size_t)j * a.ld_x], x1 = xp[(size_t)(j + 1) * a.ld_x], x2 = xp[(size_t)(j + 2) * a.ld_x],
                            x3 = xp[(size_t)(j + 3) * a.ld_x];
                m += x0; m += x1; m += x2; m += x3;                          // row order, as one by one
            }
            for (; j < s1; ++j) m += xp[(size_t)j * a.ld_x];
            a.smean[t][(size_t)sb * a.D + c] = m / (float)(s1 - s0 > 0 ? s1 - s0 : 1);
        }
        return;
    }
    const int b = find_range(a.start, a.nb, (int)blockIdx.x);
    const int H = a.H, D = a.D, LDV = D + 4;
    const int Dp = (D + 15) & ~15;                             // k padded to whole 16-blocks (zeros)
    for (int i = threadIdx.x; i < 16 * LDV + 16; i += 256) vt[i] = 0.f;    // heads >= H: zero rows; finite tail
    __syncthreads();
    if ((H & 3) == 0) {                                        // 16-byte loads: 4 heads of one (l/r, column)
        for (int i = threadIdx.x * 4; i < 2 * D * H; i += 1024) {
            const int lr = i / (D * H), c = (i / H) % D, h = i % H;
            const float4 q = *reinterpret_cast<const float4*>(a.V[b] + i);
            float* d0 = vt + (lr * 8 + h) * LDV + c;
            d0[0] = q.x; d0[LDV] = q.y; d0[2 * LDV] = q.z; d0[3 * LDV] = q.w;
        }
    } else {
        for (int i = threadIdx.x; i < 2 * D * H; i += 256) {
            const int lr = i / (D * H), c = (i / H) % D, h = i % H;
            vt[(lr * 8 + h) * LDV + c] = a.V[b][i];
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nl = dyn_count(a.dyn[b], a.ncap[b]);
    for (int tl = 0; tl < DOTS_NODES / 64; ++tl) {                // (the staged vectors serve DOTS_NODES nodes)
    const int n0 = ((int)blockIdx.x - a.start[b]) * DOTS_NODES + tl * 64 + wave * 16;
    if (n0 >= a.ncap[b]) return;
    const int r = lane & 15, kq = lane >> 4;
    const bool rlive = n0 + r < nl;
    const float* xr = a.x[b] + (size_t)(a.row0[b] + n0 + r) * a.ld_x + 4 * kq;
    const float* v = vt + r * LDV + 4 * kq;                    // B operand: output column r of this lane
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Dp; k0 += 64) {                      // 4 row loads in flight (one by one the loop is a latency chain)
        float4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 16 * u;
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rlive && kk + 4 * kq < D) xv[u] = *reinterpret_cast<const float4*>(xr + kk);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 16 * u;
            if (kk < Dp) {                                         // (uniform)
                const float4 vv = *reinterpret_cast<const float4*>(v + kk);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].x, vv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].y, vv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].z, vv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].w, vv.w, acc, 0, 0, 0);
            }
        }
    }
    // acc[q] = result (node n0 + 4 (lane >> 4) + q, output lane & 15)
    const int o = lane & 15, lr = o >> 3, h = o & 7;
    float* out = lr ? a.eR[b] : a.eL[b];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + 4 * (lane >> 4) + q;
        if (h < H && n < a.ncap[b]) out[(size_t)n * H + h] = acc[q];
    }
    }
}

// ------------------------------------------------------------------------------------------------ forward
struct AggArgs {
    // node types
    const int* seg[MAXT]; const int* dyn_n[MAXT];
    int row0[MAXT + 1], ncap[MAXT], ninst[MAXT], inst[MAXT][8];
    int nt, B;
    const int* dynB;
    // instances
    const float* bsum[MAXT];            // per node type: the summed bias rows of its instances (srec_hg_fold, prep.hip)
    const void* Ps[MAXI]; const float* eLs[MAXI]; const float* eRd[MAXI]; const float* bias[MAXI];
    const int* in_ptr[MAXI]; const int* in_idx[MAXI]; const int* esrc[MAXI];
    float* A[MAXI];
    const float* Mk[MAXI];              // attention dropout: 0 or 1/(1-p) per (edge, head); NULL = none
    const float* x; int ld_x;
    const float* smean[MAXT]; const int* sess;   // per (type, session) mean of the input rows, session of every stacked row (hg_dots)
    const float* xres;                  // residual rows already summed over the instances (feature dropout); NULL -> n_inst * x
    float* out; int ld_out;
    unsigned char* arg;
    int H, D;
    float slope;
};

#ifdef SREC_HG_TIMING   // development probe (tools/hg_timing.py)
__device__ unsigned long long g_hg_blk[16384][2];
__device__ unsigned long long g_hg_tim[16];
#define HGT(i) do { __builtin_amdgcn_sched_barrier(0); hgt[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define HGT_BEGIN() unsigned long long hgt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_hg_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime(); HGT(0)
#define HGT_W(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); HGT(i); } while (0)
#define HGT_END() do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_hg_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0 && blockIdx.x == 100) for (int i_ = 0; i_ < 7; ++i_) g_hg_tim[i_] = hgt[i_] - hgt[0]; } while (0)
#else
#define HGT(i)
#define HGT_BEGIN()
#define HGT_W(i)
#define HGT_END()
#endif

template <typename T>
__global__ __launch_bounds__(512) void hg_agg_kernel(AggArgs a) {
#ifdef SREC_HG_TIMING
    unsigned long long hgt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_hg_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
    HGT(0);
#endif
    __shared__ float sc[MAXH][MAXDEG];
    __shared__ int su[MAXH][MAXDEG];
    __shared__ float comb[MAXH][256];
    const int row = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int H = a.H, D = a.D, HD = H * D;
    const int t = find_range(a.row0, a.nt, row);
    const int v = row - a.row0[t];
    const bool live = v < dyn_count(a.dyn_n[t], a.ncap[t]);
    const int c = lane * 4;
    // the session of this node (its mean row is added at the very end): one early load, hidden behind the edge chains
    const int sb = live ? a.sess[row] : 0;
    HGT(1);
    if (w < H) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // Fast path (the usual case: a handful of in-edges per relation): lane = (instance slot q = lane >> 3, edge j =
        // lane & 7), so the dependent chains in_ptr -> in_idx -> esrc -> logit of ALL instances run side by side instead of
        // one instance after the other, the soft-max is a reduction over 8-lane groups, and the (instance, edge) pairs are
        // then aggregated with four projection-row loads in flight.  A destination with more than 8 in-edges in some
        // relation takes the general loop below.
        bool fast = false;
        if (live && a.ninst[t] <= 8) {
            const int q = lane >> 3, j = lane & 7;
            int i = -1, beg = 0, deg = 0;
            if (q < a.ninst[t]) {
                i = a.inst[t][q];
                beg = a.in_ptr[i][v];
                deg = a.in_ptr[i][v + 1] - beg;
            }
            fast = __ballot(deg > 8) == 0ull;
            if (fast) {
                const bool valid = i >= 0 && j < deg;
                int e = 0, src = 0;
                float sv = -INFINITY;
                if (valid) {
                    e = a.in_idx[i][beg + j];
                    src = a.esrc[i][e];
                    sv = a.eLs[i][(size_t)src * H + w] + a.eRd[i][(size_t)v * H + w];
                    sv = sv > 0.f ? sv : a.slope * sv;
                }
                float m = sv;
                m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
                const float ex = valid ? expf(sv - m) : 0.f;
                float z = ex;
                z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64); z += __shfl_xor(z, 4, 64);
                float pm = valid ? ex / z : 0.f;
                if (valid) {
                    a.A[i][(size_t)e * H + w] = pm;                        // soft-max value (the backward needs it)
                    const float* mk = a.Mk[i];
                    if (mk != nullptr) pm *= mk[(size_t)e * H + w];
                }
                HGT(2);
                unsigned long long mask = __ballot(valid);
                const bool cok = c < D;                                    // every lane stays in the loop: it feeds the shuffles
                while (mask != 0ull) {
                    int L[4]; float pp[4]; const T* rp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        L[u] = mask != 0ull ? __builtin_ctzll(mask) : -1;
                        if (mask != 0ull) mask &= mask - 1ull;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int l = L[u] >= 0 ? L[u] : 0;
                        const int ii = __builtin_amdgcn_readfirstlane(__shfl(i, l, 64));
                        const int ss = __shfl(src, l, 64);
                        pp[u] = L[u] >= 0 ? __shfl(pm, l, 64) : 0.f;
                        rp[u] = static_cast<const T*>(a.Ps[ii >= 0 ? ii : 0]) + (size_t)ss * HD + w * D + c;
                    }
                    float4 f[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) f[u] = (L[u] >= 0 && cok) ? ld4(rp[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc.x += pp[u] * f[u].x; acc.y += pp[u] * f[u].y; acc.z += pp[u] * f[u].z; acc.w += pp[u] * f[u].w; }
                }
                HGT(3);
                if (c < D && a.ninst[t] > 0) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.bsum[t] + w * D + c);
                    acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
                }
            }
        }
        if (live) {
            for (int q = 0; q < (fast ? 0 : a.ninst[t]); ++q) {
                const int i = a.inst[t][q];
                const int* ip = a.in_ptr[i];
                const int beg = ip[v], deg = min(ip[v + 1] - beg, MAXDEG);
                const int* idx = a.in_idx[i] + beg;
                for (int j = lane; j < deg; j += 64) su[w][j] = a.esrc[i][idx[j]];
                __builtin_amdgcn_wave_barrier();
                const float erv = a.eRd[i][(size_t)v * H + w];
                float m = -INFINITY;
                for (int j = lane; j < deg; j += 64) {
                    float s = a.eLs[i][(size_t)su[w][j] * H + w] + erv;
                    s = s > 0.f ? s : a.slope * s;
                    sc[w][j] = s;
                    m = fmaxf(m, s);
                }
                m = wave_max(m);
                float z = 0.f;
                for (int j = lane; j < deg; j += 64) z += expf(sc[w][j] - m);
                z = wave_sum(z);
                const float iz = deg > 0 ? 1.f / z : 0.f;
                const float* mk = a.Mk[i];
                for (int j = lane; j < deg; j += 64) {
                    const float p = expf(sc[w][j] - m) * iz;
                    a.A[i][(size_t)idx[j] * H + w] = p;                       // soft-max value (the backward needs it)
                    sc[w][j] = mk != nullptr ? p * mk[(size_t)idx[j] * H + w] : p;   // what the aggregation uses
                }
                __builtin_amdgcn_wave_barrier();
                if (c < D) {
                    const T* ps = static_cast<const T*>(a.Ps[i]) + w * D + c;
                    for (int j = 0; j < deg; ++j) {
                        const float p = sc[w][j];
                        const float4 f = ld4(ps + (size_t)su[w][j] * HD);
                        acc.x += p * f.x; acc.y += p * f.y; acc.z += p * f.z; acc.w += p * f.w;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!fast && c < D && a.ninst[t] > 0) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bsum[t] + w * D + c);
                acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
            }
            if (c < D) {
                if (a.xres != nullptr) {
                    const float4 xv = *reinterpret_cast<const float4*>(a.xres + (size_t)row * a.ld_x + c);
                    acc.x += xv.x; acc.y += xv.y; acc.z += xv.z; acc.w += xv.w;
                } else {
                    const float nres = (float)a.ninst[t];
                    const float4 xv = *reinterpret_cast<const float4*>(a.x + (size_t)row * a.ld_x + c);
                    acc.x += nres * xv.x; acc.y += nres * xv.y; acc.z += nres * xv.z; acc.w += nres * xv.w;
                }
            }
        }
        if (c < D) *reinterpret_cast<float4*>(&comb[w][c]) = acc;
    }
    HGT(4);
    __syncthreads();
    HGT(5);
    if (tid < D) {
        float best = 0.f;
        int bi = 0;
        if (live) {
            best = -INFINITY;
            for (int h = 0; h < H; ++h)
                if (comb[h][tid] > best) { best = comb[h][tid]; bi = h; }
            best += a.smean[t][(size_t)sb * D + tid];      // + mean of the session's input features (nodes of this type)
        }
        a.out[(size_t)row * a.ld_out + tid] = best;
        a.arg[(size_t)row * D + tid] = (unsigned char)bi;
    }
#ifdef SREC_HG_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HGT(6);
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_hg_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 100) for (int i = 0; i < 7; ++i) g_hg_tim[i] = hgt[i] - hgt[0];
#endif
}
#ifdef SREC_HG_TIMING
extern "C" int srec_hg_timing(unsigned long long* tim16, unsigned long long* blk) {
    if (hipMemcpyFromSymbol(tim16, HIP_SYMBOL(g_hg_tim), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(blk, HIP_SYMBOL(g_hg_blk), sizeof(unsigned long long) * 32768) == hipSuccess ? 0 : 1;
}
#endif

// One WAVEFRONT per destination node, all heads in the lane (H = 8, D % 8 == 0): lane = (head parity = lane >> 5, columns
// 8 (lane & 31) .. + 7), i.e. a lane accumulates heads parity, parity + 2, + 4, + 6 of its 8 columns (32 registers) and an edge
// costs FOUR 16-byte row loads per lane.  Against the 8-wave workgroup above (wave = head) the dependent chain in_ptr -> in_idx ->
// esrc -> logits -> soft-max runs ONCE per node instead of once per head-wave (lane = (instance, edge) as in its fast path, the
// 8 logits of an edge are two 16-byte loads), the head-max is in registers (one exchange between the wave's halves, no LDS, no
// barrier), bias / residual / session mean are read once.  Same summation order per output element as hg_agg_kernel:
// bit-identical results (tests/test_ops_gpu.py).
template <typename T> struct Row8;
template <> struct Row8<unsigned short> {
    uint4 v;
    __device__ __forceinline__ void load(const unsigned short* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { v = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
};
template <> struct Row8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void zero() { a = b = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};
__device__ __forceinline__ float rdlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <typename T, int AGG_EIF, int WV>   // AGG_EIF: edges (x 4 row loads of 16 bytes) in flight per wave; WV: waves per SIMD
__global__ __launch_bounds__(256, WV) void hg_agg_node_kernel(AggArgs a) {
    __shared__ float sc[WPB][MAXH][MAXDEG];                        // general path only (a relation with > 8 in-edges)
    __shared__ int su[WPB][MAXDEG];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * WPB + w;
    if (row >= a.row0[a.nt]) return;
    constexpr int H = MAXH;
    const int D = a.D, HD = H * D;
    const int t = find_range(a.row0, a.nt, row);
    const int v = row - a.row0[t];
    const bool live = v < dyn_count(a.dyn_n[t], a.ncap[t]);
    const int half = lane >> 5, c = (lane & 31) * 8;
    const bool cok = c < D;
    const int sb = live ? a.sess[row] : 0;                          // early: hidden behind the edge chains
    float acc[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[k][x] = 0.f;
    if (live) {
        const int q = lane >> 3, j = lane & 7;
        int i = -1, beg = 0, deg = 0;
        if (q < a.ninst[t]) {
            i = a.inst[t][q];
            beg = a.in_ptr[i][v];
            deg = a.in_ptr[i][v + 1] - beg;
        }
        const bool fast = a.ninst[t] <= 8 && __ballot(deg > 8) == 0ull;
        if (fast) {
            const bool valid = i >= 0 && j < deg;
            int e = 0, src = 0;
            float pm[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) pm[h] = -INFINITY;
            if (valid) {
                e = a.in_idx[i][beg + j];
                src = a.esrc[i][e];
                float el[8], er[8];
                ld8f(a.eLs[i] + (size_t)src * H, el);
                ld8f(a.eRd[i] + (size_t)v * H, er);
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const float sv = el[h] + er[h];
                    pm[h] = sv > 0.f ? sv : a.slope * sv;
                }
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float sv = pm[h];
                float m = sv;
                m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
                const float ex = valid ? expf(sv - m) : 0.f;
                float z = ex;
                z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64); z += __shfl_xor(z, 4, 64);
                pm[h] = valid ? ex / z : 0.f;
            }
            if (valid) {
                float* ap = a.A[i] + (size_t)e * H;                  // soft-max values (the backward needs them)
                *reinterpret_cast<float4*>(ap) = make_float4(pm[0], pm[1], pm[2], pm[3]);
                *reinterpret_cast<float4*>(ap + 4) = make_float4(pm[4], pm[5], pm[6], pm[7]);
                if (a.Mk[i] != nullptr) {
                    float mk[8];
                    ld8f(a.Mk[i] + (size_t)e * H, mk);
#pragma unroll
                    for (int h = 0; h < 8; ++h) pm[h] *= mk[h];
                }
            }
            unsigned long long mask = __ballot(valid);
            while (mask != 0ull) {
                Row8<T> f[AGG_EIF][4];
                float pp[AGG_EIF][4];
#pragma unroll
                for (int u = 0; u < AGG_EIF; ++u) {
                    const bool on = mask != 0ull;
                    const int l = on ? __builtin_ctzll(mask) : 0;
                    if (on) mask &= mask - 1ull;
                    const int ii = __builtin_amdgcn_readlane(i, l), ss = __builtin_amdgcn_readlane(src, l);
                    const T* rp = static_cast<const T*>(a.Ps[ii >= 0 ? ii : 0]) + (size_t)ss * HD + half * D + c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float p0 = rdlane(pm[2 * k], l), p1 = rdlane(pm[2 * k + 1], l);
                        pp[u][k] = on ? (half ? p1 : p0) : 0.f;
                        if (on && cok) f[u][k].load(rp + 2 * k * D); else f[u][k].zero();
                    }
                }
#pragma unroll
                for (int u = 0; u < AGG_EIF; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float fv[8];
                        f[u][k].get(fv);
#pragma unroll
                        for (int x = 0; x < 8; ++x) acc[k][x] += pp[u][k] * fv[x];
                    }
            }
        } else {
            for (int qq = 0; qq < a.ninst[t]; ++qq) {
                const int ig = a.inst[t][qq];
                const int* ip = a.in_ptr[ig];
                const int bg = ip[v], dg = min(ip[v + 1] - bg, MAXDEG);
                const int* idx = a.in_idx[ig] + bg;
                for (int jj = lane; jj < dg; jj += 64) su[w][jj] = a.esrc[ig][idx[jj]];
                __builtin_amdgcn_wave_barrier();
                const float* mk = a.Mk[ig];
                for (int h = 0; h < H; ++h) {
                    const float erv = a.eRd[ig][(size_t)v * H + h];
                    float m = -INFINITY;
                    for (int jj = lane; jj < dg; jj += 64) {
                        float s = a.eLs[ig][(size_t)su[w][jj] * H + h] + erv;
                        s = s > 0.f ? s : a.slope * s;
                        sc[w][h][jj] = s;
                        m = fmaxf(m, s);
                    }
                    m = wave_max(m);
                    float z = 0.f;
                    for (int jj = lane; jj < dg; jj += 64) z += expf(sc[w][h][jj] - m);
                    z = wave_sum(z);
                    const float iz = dg > 0 ? 1.f / z : 0.f;
                    for (int jj = lane; jj < dg; jj += 64) {
                        const float p = expf(sc[w][h][jj] - m) * iz;
                        a.A[ig][(size_t)idx[jj] * H + h] = p;
                        sc[w][h][jj] = mk != nullptr ? p * mk[(size_t)idx[jj] * H + h] : p;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (cok) {
                    const T* ps = static_cast<const T*>(a.Ps[ig]) + half * D + c;
                    for (int jj = 0; jj < dg; ++jj) {
                        const T* rp = ps + (size_t)su[w][jj] * HD;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            Row8<T> f;
                            f.load(rp + 2 * k * D);
                            float fv[8];
                            f.get(fv);
                            const float p = sc[w][2 * k + half][jj];
#pragma unroll
                            for (int x = 0; x < 8; ++x) acc[k][x] += p * fv[x];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (cok) {
            if (a.ninst[t] > 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float bv[8];
                    ld8f(a.bsum[t] + (2 * k + half) * D + c, bv);
#pragma unroll
                    for (int x = 0; x < 8; ++x) acc[k][x] += bv[x];
                }
            }
            float xv[8];
            if (a.xres != nullptr) {
                ld8f(a.xres + (size_t)row * a.ld_x + c, xv);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int x = 0; x < 8; ++x) acc[k][x] += xv[x];
            } else {
                const float nres = (float)a.ninst[t];
                ld8f(a.x + (size_t)row * a.ld_x + c, xv);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int x = 0; x < 8; ++x) acc[k][x] += nres * xv[x];
            }
        }
    }
    // head-max: this lane's four heads (ascending, first maximum wins as in a scan over h = 0 .. 7), then the other parity's
    float best[8];
    int bi[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        float b = -INFINITY;
        int bh = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (acc[k][x] > b) { b = acc[k][x]; bh = 2 * k + half; }
        const float ob = __shfl_xor(b, 32, 64);
        const int oh = __shfl_xor(bh, 32, 64);
        // the scan picks the lowest head among equal maxima; a lane half that saw only NaN / -inf keeps (-inf, 0)
        if (ob > b || (ob == b && oh < bh)) { b = ob; bh = oh; }
        best[x] = b; bi[x] = bh;
    }
    if (half == 0 && cok) {
        float o[8];
        unsigned lo = 0u, hi = 0u;
        if (live) {
            float sm[8];
            ld8f(a.smean[t] + (size_t)sb * D + c, sm);
#pragma unroll
            for (int x = 0; x < 8; ++x) o[x] = best[x] + sm[x];   // + mean of the session's input features (nodes of this type)
#pragma unroll
            for (int x = 0; x < 4; ++x) { lo |= (unsigned)bi[x] << (8 * x); hi |= (unsigned)bi[4 + x] << (8 * x); }
        } else {
#pragma unroll
            for (int x = 0; x < 8; ++x) o[x] = 0.f;
        }
        float* op = a.out + (size_t)row * a.ld_out + c;
        *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
        *reinterpret_cast<uint2*>(a.arg + (size_t)row * D + c) = make_uint2(lo, hi);
    }
}

// ------------------------------------------------------------------------------------------------ backward
struct PreArgs {
    const int* seg[MAXT]; const int* dyn_n[MAXT];
    int row0[MAXT + 1], ncap[MAXT], ninst[MAXT];
    int nt, B, D;
    const int* dynB;
    const float* g; int ld_g;
    const float* rm;                    // [NT, D] per-element residual scale (feature dropout); NULL -> n_inst, or recomputed:
    const float* rm_cnt; float rm_p; srec_rng rm_rng;    // rm_cnt [2, NT] != NULL: rm = cnt0 m0 + cnt1 m1 from the masks' hash
    float* dx; int ld_dx;
    const int* sess;                    // session of every stacked row (written by the forward's hg_dots launch)
};

// dx[row,:] = nres * g[row,:] + (1/n_session) * sum_{rows of the session} g   (residual + session-mean terms)
// MERGE: ... + (sum_s t[0][s]) * ms0 + (sum_s t[1][s]) * ms1, the two convs' masked data gradients (S partial sums each, t [2, S,
// NT, D] contiguous; masks recomputed from the hash of hg_drop_prep as for the residual scale) - the layer's whole d x in ONE
// pass at the end of its backward instead of a pre-fill here and a read-modify-write in hg_drop_merge (srec_hg_pre_merge)
template <bool MERGE>
__global__ void hg_pre_kernel(PreArgs a, const float* __restrict__ tm = nullptr, int S = 0) {
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.row0[a.nt]) return;
    const int t = find_range(a.row0, a.nt, row);
    const int v = row - a.row0[t];
    const bool live = v < dyn_count(a.dyn_n[t], a.ncap[t]);
    const int c = lane * 4;
    if (c >= a.D) return;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const int* seg = a.seg[t];
        const int lo = a.sess[row];                        // (one load instead of a 5-round-trip search of the offsets)
        const int s0 = seg[lo], s1 = seg[lo + 1];
        int j = s0;
        for (; j + 3 < s1; j += 4) {                       // four rows in flight, added in row order
            float4 gv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = *reinterpret_cast<const float4*>(a.g + (size_t)(a.row0[t] + j + e) * a.ld_g + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o.x += gv[e].x; o.y += gv[e].y; o.z += gv[e].z; o.w += gv[e].w; }
        }
        for (; j < s1; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(a.g + (size_t)(a.row0[t] + j) * a.ld_g + c);
            o.x += gv.x; o.y += gv.y; o.z += gv.z; o.w += gv.w;
        }
        const float inv = 1.f / (float)(s1 - s0 > 0 ? s1 - s0 : 1), nres = (float)a.ninst[t];
        const float4 gv = *reinterpret_cast<const float4*>(a.g + (size_t)row * a.ld_g + c);
        float4 rs = make_float4(nres, nres, nres, nres);
        if (a.rm != nullptr) {
            rs = *reinterpret_cast<const float4*>(a.rm + (size_t)row * a.D + c);
        } else if (a.rm_cnt != nullptr) {
            const int NT = a.row0[a.nt];
            const float c0 = a.rm_cnt[row], c1 = a.rm_cnt[NT + row];
            const unsigned key = srec_rng_key(a.rm_rng);
            const float p = a.rm_p, sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
            const unsigned i0 = (unsigned)row * (unsigned)a.D + (unsigned)c, i1 = (unsigned)NT * (unsigned)a.D + i0;
            rs.x = c0 * srec_keep(key, i0, p, sc) + c1 * srec_keep(key, i1, p, sc);
            rs.y = c0 * srec_keep(key, i0 + 1, p, sc) + c1 * srec_keep(key, i1 + 1, p, sc);
            rs.z = c0 * srec_keep(key, i0 + 2, p, sc) + c1 * srec_keep(key, i1 + 2, p, sc);
            rs.w = c0 * srec_keep(key, i0 + 3, p, sc) + c1 * srec_keep(key, i1 + 3, p, sc);
        }
        o.x = o.x * inv + rs.x * gv.x; o.y = o.y * inv + rs.y * gv.y;
        o.z = o.z * inv + rs.z * gv.z; o.w = o.w * inv + rs.w * gv.w;
    }
    if (MERGE) {
        const int NT = a.row0[a.nt];
        const size_t n = (size_t)NT * a.D, i = (size_t)row * a.D + c;
        float4 s0 = *reinterpret_cast<const float4*>(tm + i), s1 = *reinterpret_cast<const float4*>(tm + (size_t)S * n + i);
        for (int s = 1; s < S; ++s) {
            const float4 a2 = *reinterpret_cast<const float4*>(tm + (size_t)s * n + i);
            const float4 b2 = *reinterpret_cast<const float4*>(tm + (size_t)(S + s) * n + i);
            s0.x += a2.x; s0.y += a2.y; s0.z += a2.z; s0.w += a2.w;
            s1.x += b2.x; s1.y += b2.y; s1.z += b2.z; s1.w += b2.w;
        }
        const unsigned key = srec_rng_key(a.rm_rng);
        const float p = a.rm_p, sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
        const unsigned i0 = (unsigned)i, i1 = (unsigned)(n + i);
        o.x += s0.x * srec_keep(key, i0, p, sc) + s1.x * srec_keep(key, i1, p, sc);
        o.y += s0.y * srec_keep(key, i0 + 1, p, sc) + s1.y * srec_keep(key, i1 + 1, p, sc);
        o.z += s0.z * srec_keep(key, i0 + 2, p, sc) + s1.z * srec_keep(key, i1 + 2, p, sc);
        o.w += s0.w * srec_keep(key, i0 + 3, p, sc) + s1.w * srec_keep(key, i1 + 3, p, sc);
    }
    *reinterpret_cast<float4*>(a.dx + (size_t)row * a.ld_dx + c) = o;
}

__device__ __forceinline__ float4 masked_grad(const float* g, int ld_g, const unsigned char* arg, int D, int row, int h,
                                              int c) {
    const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)row * ld_g + c);
    const uchar4 bi = *reinterpret_cast<const uchar4*>(arg + (size_t)row * D + c);
    return make_float4(bi.x == h ? gv.x : 0.f, bi.y == h ? gv.y : 0.f, bi.z == h ? gv.z : 0.f, bi.w == h ? gv.w : 0.f);
}

struct DstArgs {
    const void* Ps[MAXI]; const float* eLs[MAXI]; const float* eRd[MAXI]; const float* A[MAXI];
    const int* in_ptr[MAXI]; const int* in_idx[MAXI]; const int* esrc[MAXI];
    float* DP[MAXI]; float* der[MAXI];
    const float* Mk[MAXI];
    const int* dyn_d[MAXI];
    int ncap_d[MAXI], row0_d[MAXI];
    int start[MAXI + 1];
    int ni, H, D;
    float slope;
    const float* g; int ld_g;
    const unsigned char* arg;
    // node types (hg_bwd_dst_node_kernel: one wavefront per destination NODE over all instances into its type)
    const int* dyn_t[MAXT];
    int row0_t[MAXT + 1], ncap_t[MAXT], ninst_t[MAXT], inst_t[MAXT][8];
    int nt;
};

// per (instance, destination), ALL heads in one wavefront: d(pre-activation score) of every in-edge -> DP[e,h];
// der[v,h] = their sum.  The gradient of every relation result into a destination is the same masked tensor
// g[v,c] * [arg[v,c] == h]: each column c belongs to exactly ONE head, so an edge needs only the D gathered elements
// P[u, arg[v,c], c] (not H*D) - 8x less gather traffic than a wave per head.  Lane layout of the small per-edge
// arrays: head = lane & 7, edge slot = lane >> 3.
template <typename T>
__global__ void hg_bwd_dst_kernel(DstArgs a) {
    __shared__ float da[WPB][MAXDEG][MAXH];
    __shared__ int su[WPB][MAXDEG];
    const int i = find_range(a.start, a.ni, (int)blockIdx.x);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = ((int)blockIdx.x - a.start[i]) * WPB + w;
    const int H = a.H, D = a.D, HD = H * D;
    if (v >= a.ncap_d[i]) return;
    const bool live = v < dyn_count(a.dyn_d[i], a.ncap_d[i]);
    const int hl = lane & 7, jl = lane >> 3;
    if (!live) {
        if (lane < H) a.der[i][(size_t)v * H + lane] = 0.f;
        return;
    }
    const int beg = a.in_ptr[i][v];
    const int deg = min(a.in_ptr[i][v + 1] - beg, MAXDEG);
    const int* idx = a.in_idx[i] + beg;
    for (int j = lane; j < deg; j += 64) su[w][j] = a.esrc[i][idx[j]];
    __builtin_amdgcn_wave_barrier();
    const int c = lane * 4;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 a4 = make_uchar4(0, 0, 0, 0);
    if (c < D) {
        const size_t row = (size_t)(a.row0_d[i] + v);
        g4 = *reinterpret_cast<const float4*>(a.g + row * a.ld_g + c);
        a4 = *reinterpret_cast<const uchar4*>(a.arg + row * D + c);
    }
    const T* P = static_cast<const T*>(a.Ps[i]);
    for (int j = 0; j < deg; ++j) {
        float ph[MAXH];
#pragma unroll
        for (int h = 0; h < MAXH; ++h) ph[h] = 0.f;
        if (c < D) {
            const T* pr = P + (size_t)su[w][j] * HD + c;
            const float v0 = g4.x * ld1(pr + a4.x * D), v1 = g4.y * ld1(pr + a4.y * D + 1);
            const float v2 = g4.z * ld1(pr + a4.z * D + 2), v3 = g4.w * ld1(pr + a4.w * D + 3);
#pragma unroll
            for (int h = 0; h < MAXH; ++h)
                ph[h] = (a4.x == h ? v0 : 0.f) + (a4.y == h ? v1 : 0.f) + (a4.z == h ? v2 : 0.f) + (a4.w == h ? v3 : 0.f);
        }
        // 8 wave sums in 10 shuffles instead of 48: a reduce-scatter over lane bits 0-2 (each step a lane keeps the half
        // of its values that matches its bit and adds the partner's copy of that half), then a plain reduction over bits
        // 3-5.  Lane l ends up with the total of head hb = bit-reversal of (l & 7).
        {
            const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
            float q4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float keep = b0 ? ph[k + 4] : ph[k], send = b0 ? ph[k] : ph[k + 4];
                q4[k] = keep + __shfl_xor(send, 1, 64);
            }
            float q2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float keep = b1 ? q4[k + 2] : q4[k], send = b1 ? q4[k] : q4[k + 2];
                q2[k] = keep + __shfl_xor(send, 2, 64);
            }
            float t = (b2 ? q2[1] : q2[0]) + __shfl_xor(b2 ? q2[0] : q2[1], 4, 64);
            t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            const int hb = (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);
            if (lane < MAXH) {
                if (a.Mk[i] != nullptr && hb < H) t *= a.Mk[i][(size_t)idx[j] * H + hb];   // d a = d a_dropped * mask
                da[w][j][hb] = t;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (hl >= H) return;
    float tsum = 0.f;
    for (int j = jl; j < deg; j += 8) tsum += a.A[i][(size_t)idx[j] * H + hl] * da[w][j][hl];
    tsum += __shfl_xor(tsum, 8, 64); tsum += __shfl_xor(tsum, 16, 64); tsum += __shfl_xor(tsum, 32, 64);
    const float erv = a.eRd[i][(size_t)v * H + hl];
    float dsum = 0.f;
    for (int j = jl; j < deg; j += 8) {
        const int e = idx[j];
        const float p = a.A[i][(size_t)e * H + hl];
        const float pre = a.eLs[i][(size_t)su[w][j] * H + hl] + erv;
        const float dp = p * (da[w][j][hl] - tsum) * (pre > 0.f ? 1.f : a.slope);
        a.DP[i][(size_t)e * H + hl] = dp;
        dsum += dp;
    }
    dsum += __shfl_xor(dsum, 8, 64); dsum += __shfl_xor(dsum, 16, 64); dsum += __shfl_xor(dsum, 32, 64);
    if (jl == 0) a.der[i][(size_t)v * H + hl] = dsum;
}

// the 8 per-head sums of ph over the wave in 10 shuffles (see hg_bwd_dst_kernel); lane l < 8 ends with head *hb
__device__ __forceinline__ float head_sums8(const float (&ph)[MAXH], int lane, int* hb) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float q4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = b0 ? ph[k + 4] : ph[k], send = b0 ? ph[k] : ph[k + 4];
        q4[k] = keep + __shfl_xor(send, 1, 64);
    }
    float q2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = b1 ? q4[k + 2] : q4[k], send = b1 ? q4[k] : q4[k + 2];
        q2[k] = keep + __shfl_xor(send, 2, 64);
    }
    float t = (b2 ? q2[1] : q2[0]) + __shfl_xor(b2 ? q2[0] : q2[1], 4, 64);
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    *hb = (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);
    return t;
}

// One wavefront per destination NODE, all relation instances into its type side by side (H = 8; lane = (instance slot
// lane >> 3, in-edge lane & 7) for the per-edge arrays, as in hg_agg_node_kernel): 14 x fewer wavefronts than one per
// (instance, destination) - most of which only found an empty edge list -, the chains in_ptr -> in_idx -> esrc of all
// instances run in parallel, g / arg of the node are read once, the 8 values of an (edge, head) array are 16-byte accesses.
// Same arithmetic per edge and the same reduction trees as hg_bwd_dst_kernel: bit-identical DP / der.  A node with more than
// 8 in-edges in some relation takes the per-instance loop (the body of hg_bwd_dst_kernel).
template <typename T, int DIF, int WV>   // DIF: edges (x 4 row loads) in flight together; WV: waves per SIMD
__global__ __launch_bounds__(256, WV) void hg_bwd_dst_node_kernel(DstArgs a) {
    __shared__ float da[WPB][MAXDEG][MAXH];
    __shared__ int su[WPB][MAXDEG];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // w in an SGPR: row, type, node are wave-uniform
    HGT_BEGIN();
    const int row = blockIdx.x * WPB + w;
    if (row >= a.row0_t[a.nt]) return;
    const int t = find_range(a.row0_t, a.nt, row);
    const int v = row - a.row0_t[t];
    const int H = a.H, D = a.D, HD = H * D;
    const int ni = a.ninst_t[t];
    if (v >= dyn_count(a.dyn_t[t], a.ncap_t[t])) {
        for (int q = 0; q < ni; ++q)
            if (lane < H) a.der[a.inst_t[t][q]][(size_t)v * H + lane] = 0.f;
        return;
    }
    const int q = lane >> 3, j = lane & 7;
    int i = -1, beg = 0, deg = 0;
    if (q < ni && q < 8) {
        i = a.inst_t[t][q];
        beg = a.in_ptr[i][v];
        deg = a.in_ptr[i][v + 1] - beg;
    }
    const int c = lane * 4;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 a4 = make_uchar4(0, 0, 0, 0);
    if (c < D) {
        g4 = *reinterpret_cast<const float4*>(a.g + (size_t)row * a.ld_g + c);
        a4 = *reinterpret_cast<const uchar4*>(a.arg + (size_t)row * D + c);
    }
    const bool valid = i >= 0 && j < deg;
    HGT_W(1);
    int e = 0, src = 0;
    if (valid) {
        e = a.in_idx[i][beg + j];
        src = a.esrc[i][e];
    }
    HGT_W(2);
    // (host: this kernel is launched only when H == 8 and every type has <= 8 instances)
    if (__ballot(deg > 8) != 0ull) {                               // general path: one instance after the other
        for (int qq = 0; qq < ni; ++qq) {
            const int ig = a.inst_t[t][qq];
            const int bg = a.in_ptr[ig][v];
            const int dg = min(a.in_ptr[ig][v + 1] - bg, MAXDEG);
            const int* idx = a.in_idx[ig] + bg;
            for (int jj = lane; jj < dg; jj += 64) su[w][jj] = a.esrc[ig][idx[jj]];
            __builtin_amdgcn_wave_barrier();
            const T* P = static_cast<const T*>(a.Ps[ig]);
            for (int jj = 0; jj < dg; ++jj) {
                float ph[MAXH];
#pragma unroll
                for (int h = 0; h < MAXH; ++h) ph[h] = 0.f;
                if (c < D) {
                    const T* pr = P + (size_t)su[w][jj] * HD + c;
                    const float v0 = g4.x * ld1(pr + a4.x * D), v1 = g4.y * ld1(pr + a4.y * D + 1);
                    const float v2 = g4.z * ld1(pr + a4.z * D + 2), v3 = g4.w * ld1(pr + a4.w * D + 3);
#pragma unroll
                    for (int h = 0; h < MAXH; ++h)
                        ph[h] = (a4.x == h ? v0 : 0.f) + (a4.y == h ? v1 : 0.f) + (a4.z == h ? v2 : 0.f) + (a4.w == h ? v3 : 0.f);
                }
                int hb;
                float ts = head_sums8(ph, lane, &hb);
                if (lane < MAXH) {
                    if (a.Mk[ig] != nullptr && hb < H) ts *= a.Mk[ig][(size_t)idx[jj] * H + hb];
                    da[w][jj][hb] = ts;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int hl = lane & 7, jl = lane >> 3;
            if (hl < H) {
                float tsum = 0.f;
                for (int jj = jl; jj < dg; jj += 8) tsum += a.A[ig][(size_t)idx[jj] * H + hl] * da[w][jj][hl];
                tsum += __shfl_xor(tsum, 8, 64); tsum += __shfl_xor(tsum, 16, 64); tsum += __shfl_xor(tsum, 32, 64);
                const float erv = a.eRd[ig][(size_t)v * H + hl];
                float dsum = 0.f;
                for (int jj = jl; jj < dg; jj += 8) {
                    const int ee = idx[jj];
                    const float p = a.A[ig][(size_t)ee * H + hl];
                    const float pre = a.eLs[ig][(size_t)su[w][jj] * H + hl] + erv;
                    const float dp = p * (da[w][jj][hl] - tsum) * (pre > 0.f ? 1.f : a.slope);
                    a.DP[ig][(size_t)ee * H + hl] = dp;
                    dsum += dp;
                }
                dsum += __shfl_xor(dsum, 8, 64); dsum += __shfl_xor(dsum, 16, 64); dsum += __shfl_xor(dsum, 32, 64);
                if (jl == 0) a.der[ig][(size_t)v * H + hl] = dsum;
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // the per-edge arrays of this lane's (instance, edge): requested before the gather loop, used after it
    float pa[8], mk[8], el[8], er[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) { pa[h] = 0.f; mk[h] = 1.f; el[h] = 0.f; er[h] = 0.f; }
    if (valid) {
        ld8f(a.A[i] + (size_t)e * 8, pa);
        if (a.Mk[i] != nullptr) ld8f(a.Mk[i] + (size_t)e * 8, mk);
        ld8f(a.eLs[i] + (size_t)src * 8, el);
    }
    if (i >= 0 && j == 0) ld8f(a.eRd[i] + (size_t)v * 8, er);
    unsigned pos = 0u;                                               // bit h: pre-activation score of (edge, head h) > 0
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const float erv = __shfl(er[h], lane & ~7, 64);
        pos |= (el[h] + erv > 0.f ? 1u : 0u) << h;
    }
    // The masked dot products <g[v,:] [arg[v,:] == h], P[src,h,:]> of every in-edge.  The source's projection row is read WHOLE
    // in the layout of hg_agg_node_kernel (lane = head parity x 8 columns: four 16-byte loads, 32 cache lines per edge) and the
    // elements of the winning head are picked in registers: gathering the 256 wanted elements one by one (hg_bwd_dst_kernel)
    // touches the same 32 lines once per load instruction - 4 x the L1 traffic for 1/8 of the bytes.
    unsigned long long mask = __ballot(valid);
    const int half = lane >> 5, c8 = (lane & 31) * 8;
    const bool cok8 = c8 < D;
    float gsel[8];                                                   // g[v, c] where a head of this lane's parity won column c, else 0
    int kx[8];                                                       // ... and which of the lane's four heads it was
    {
        float gq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint2 aq = make_uint2(0xffffffffu, 0xffffffffu);
        if (cok8) {
            ld8f(a.g + (size_t)row * a.ld_g + c8, gq);
            aq = *reinterpret_cast<const uint2*>(a.arg + (size_t)row * D + c8);
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int hx = (int)(((x < 4 ? aq.x : aq.y) >> (8 * (x & 3))) & 0xffu);
            gsel[x] = (hx & 1) == half && hx < 8 ? gq[x] : 0.f;
            kx[x] = hx >> 1;
        }
    }
    while (mask != 0ull) {
        int ls[DIF];
        Row8<T> f[DIF][4];
#pragma unroll
        for (int u = 0; u < DIF; ++u) {
            const bool on = mask != 0ull;
            const int l = on ? __builtin_ctzll(mask) : 0;
            if (on) mask &= mask - 1ull;
            ls[u] = on ? l : -1;
            const int ii = __builtin_amdgcn_readlane(i, l), ss = __builtin_amdgcn_readlane(src, l);
            const T* rp = static_cast<const T*>(a.Ps[ii >= 0 ? ii : 0]) + (size_t)ss * HD + half * D + c8;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (on && cok8) f[u][k].load(rp + 2 * k * D); else f[u][k].zero();
        }
#pragma unroll
        for (int u = 0; u < DIF; ++u) {
            if (ls[u] < 0) break;                                    // wave-uniform
            float ph[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float fv[8];
                f[u][k].get(fv);
                float sacc = 0.f;
#pragma unroll
                for (int x = 0; x < 8; ++x) sacc += (kx[x] == k ? gsel[x] : 0.f) * fv[x];
                ph[k] = sacc;
            }
            // four sums over the 32 lanes of a half in 6 shuffles: reduce-scatter over lane bits 0-1, plain sums over bits 2-4
            const bool b0 = lane & 1, b1 = lane & 2;
            const float k0 = b0 ? ph[2] : ph[0], s0 = b0 ? ph[0] : ph[2];
            const float k1 = b0 ? ph[3] : ph[1], s1 = b0 ? ph[1] : ph[3];
            const float q0 = k0 + __shfl_xor(s0, 1, 64), q1 = k1 + __shfl_xor(s1, 1, 64);
            float ts = (b1 ? q1 : q0) + __shfl_xor(b1 ? q0 : q1, 2, 64);
            ts += __shfl_xor(ts, 4, 64); ts += __shfl_xor(ts, 8, 64); ts += __shfl_xor(ts, 16, 64);
            const int hb = 2 * ((b0 ? 2 : 0) + (b1 ? 1 : 0)) + half;
            if ((lane & 28) == 0) da[w][ls[u]][hb] = ts;             // slot = the (instance, edge) lane
        }
    }
    HGT_W(4);
    __builtin_amdgcn_wave_barrier();
    float dv[8];
    ld8f(&da[w][lane][0], dv);
    float dp[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const float d_a = valid ? dv[h] * mk[h] : 0.f;               // d a = d a_dropped * mask
        float tsum = valid ? pa[h] * d_a : 0.f;
        tsum += __shfl_xor(tsum, 1, 64); tsum += __shfl_xor(tsum, 2, 64); tsum += __shfl_xor(tsum, 4, 64);
        dp[h] = valid ? pa[h] * (d_a - tsum) * ((pos >> h) & 1u ? 1.f : a.slope) : 0.f;
    }
    if (valid) {
        float* o = a.DP[i] + (size_t)e * 8;
        *reinterpret_cast<float4*>(o) = make_float4(dp[0], dp[1], dp[2], dp[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(dp[4], dp[5], dp[6], dp[7]);
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        float dsum = dp[h];
        dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64); dsum += __shfl_xor(dsum, 4, 64);
        dp[h] = dsum;
    }
    if (i >= 0 && j == 0) {
        float* o = a.der[i] + (size_t)v * 8;
        *reinterpret_cast<float4*>(o) = make_float4(dp[0], dp[1], dp[2], dp[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(dp[4], dp[5], dp[6], dp[7]);
    }
    HGT_W(5);
    HGT_END();
}

struct SrcArgs {
    // projection blocks
    void* dP[MAXB]; float* wL[MAXB]; float* wR[MAXB];
    const float* al[MAXB]; const float* ar[MAXB];
    const int* dyn[MAXB];
    int ncap[MAXB], nsrc[MAXB], ndst[MAXB], src[MAXB][4], dst[MAXB][4];
    int start[MAXB + 1];
    // instances
    const float* A[MAXI]; const float* DP[MAXI]; const float* der[MAXI]; const float* Mk[MAXI];
    const int* out_ptr[MAXI]; const int* out_idx[MAXI]; const int* edst[MAXI];
    int row0_d[MAXI];
    int nb, H, D;
    const float* g; int ld_g;
    const unsigned char* arg;
    int skip_dead;                      // dP rows past the live count stay unwritten (their readers stop at the live rows)
};

// per (projection block, node u), ALL heads in one wavefront:
//   dP[u,h,:] = sum over the instances that read the block as SOURCE of
//                 sum_{e in out(u)} A[e,h] * dT[dst_e,h,:] + del[u,h] * a_l[h,:]   (del = sum of DP over out-edges)
//             + sum over the instances that use it as DESTINATION of der[u,h] * a_r[h,:];  wL / wR = summed del / der.
// dT[dst,h,c] = g[dst,c] * [arg[dst,c] == h]: the row g[dst,:] is read once per edge, not once per head.
// Workgroup = WPB_SRC nodes of ONE block: the block's attention vectors a_l | a_r (2 x H*D floats, 16 KB at H = 8, D = 256)
// are staged in LDS once per workgroup - read per wave from L2 they were 17 of the kernel's 58 us (knock-out runs, r03 notes).
constexpr int WPB_SRC = 4;
template <typename T, int WV, bool DER_FIRST>
__global__ __launch_bounds__(64 * WPB_SRC, WV) void hg_bwd_src_kernel(SrcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float alr[];       // [2][H*D]
    HGT_BEGIN();
    const int b = find_range(a.start, a.nb, (int)blockIdx.x);
    const int lane = threadIdx.x & 63;
    const int u = ((int)blockIdx.x - a.start[b]) * WPB_SRC + (threadIdx.x >> 6);
    const int H = a.H, D = a.D, HD = H * D;
    {
        const float* al = a.al[b]; const float* ar = a.ar[b];
        for (int i = threadIdx.x * 4; i < HD; i += 64 * WPB_SRC * 4) {
            *reinterpret_cast<float4*>(alr + i) = *reinterpret_cast<const float4*>(al + i);
            *reinterpret_cast<float4*>(alr + HD + i) = *reinterpret_cast<const float4*>(ar + i);
        }
    }
    const bool in_range = u < a.ncap[b];
    const bool live = in_range && u < dyn_count(a.dyn[b], a.ncap[b]);
    const int c = lane * 4, hl = lane & 7, jl = lane >> 3;
    float4 o[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) o[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    float wl = 0.f, wr = 0.f;                          // lane holds head hl = lane & 7
    if (DER_FIRST && live && hl < H)                   // (independent of the edge chains below: requested first)
        for (int q = 0; q < a.ndst[b]; ++q) wr += a.der[a.dst[b][q]][(size_t)u * H + hl];
    // Fast path (H = 8, <= 16 out-edges per instance): lane = (source-instance slot lane >> 4, out-edge lane & 15), so the
    // chains out_ptr -> out_idx -> edst and the per-edge arrays (DP, A, Mk) of ALL instances are fetched side by side, and the
    // destination rows g / arg are then read with every address already in registers.  The loop below it walks instance after
    // instance and edge after edge with three dependent loads each (out_idx -> edst -> g row): ~18 memory round trips for a
    // node with two source instances of two edges, the bulk of this kernel's 54 us.
    bool fast = false;
    if (H == MAXH) {
        const int sl = lane >> 4, j = lane & 15;
        int i = -1, beg = 0, deg = 0;
        if (live && sl < a.nsrc[b]) {
            i = a.src[b][sl];
            beg = a.out_ptr[i][u];
            deg = a.out_ptr[i][u + 1] - beg;
        }
        fast = __ballot(deg > 16) == 0ull;
        if (fast) {
            const bool valid = i >= 0 && j < deg;
            int drow = 0;
            float p8[8], d8[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) { p8[h] = 0.f; d8[h] = 0.f; }
            if (valid) {
                const int e = a.out_idx[i][beg + j];
                drow = a.row0_d[i] + a.edst[i][e];
                ld8f(a.DP[i] + (size_t)e * 8, d8);
                ld8f(a.A[i] + (size_t)e * 8, p8);
                if (a.Mk[i] != nullptr) {
                    float m8[8];
                    ld8f(a.Mk[i] + (size_t)e * 8, m8);
#pragma unroll
                    for (int h = 0; h < 8; ++h) p8[h] *= m8[h];
                }
            }
            HGT_W(1);
            {   // del[u, h] = sum of DP over all out-edges: 8 wave sums in 10 shuffles, then head h to the lanes with hl == h
                int hb;
                const float ts = head_sums8(d8, lane, &hb);
                const int rev = ((hl & 1) << 2) | (hl & 2) | ((hl & 4) >> 2);    // lane rev holds head hl
                wl = __shfl(ts, rev, 64);
            }
            unsigned long long mask = __ballot(valid);
            while (mask != 0ull) {
                constexpr int SIF = 2;                   // destination rows in flight
                float4 gv[SIF]; uchar4 bi[SIF]; int ls[SIF];
#pragma unroll
                for (int x = 0; x < SIF; ++x) {
                    const bool on = mask != 0ull;
                    const int l = on ? __builtin_ctzll(mask) : 0;
                    if (on) mask &= mask - 1ull;
                    ls[x] = on ? l : -1;
                    const size_t row = (size_t)__builtin_amdgcn_readlane(drow, l);
                    gv[x] = make_float4(0.f, 0.f, 0.f, 0.f); bi[x] = make_uchar4(255, 255, 255, 255);
                    if (on && c < D) {
                        gv[x] = *reinterpret_cast<const float4*>(a.g + row * a.ld_g + c);
                        bi[x] = *reinterpret_cast<const uchar4*>(a.arg + row * D + c);
                    }
                }
#pragma unroll
                for (int x = 0; x < SIF; ++x) {
                    if (ls[x] < 0) break;                // wave-uniform
#pragma unroll
                    for (int h = 0; h < MAXH; ++h) {
                        const float ph = rdlane(p8[h], ls[x]);
                        o[h].x += bi[x].x == h ? ph * gv[x].x : 0.f; o[h].y += bi[x].y == h ? ph * gv[x].y : 0.f;
                        o[h].z += bi[x].z == h ? ph * gv[x].z : 0.f; o[h].w += bi[x].w == h ? ph * gv[x].w : 0.f;
                    }
                }
            }
            HGT_W(2);
        }
    }
    if (live && !fast) {
        for (int q = 0; q < a.nsrc[b]; ++q) {
            const int i = a.src[b][q];
            const int beg = a.out_ptr[i][u], deg = a.out_ptr[i][u + 1] - beg;
            const int* idx = a.out_idx[i] + beg;
            float dl = 0.f;
            if (hl < H)
                for (int j = jl; j < deg; j += 8) dl += a.DP[i][(size_t)idx[j] * H + hl];
            dl += __shfl_xor(dl, 8, 64); dl += __shfl_xor(dl, 16, 64); dl += __shfl_xor(dl, 32, 64);
            wl += dl;
            if (c < D) {
                for (int j = 0; j < deg; ++j) {
                    const int e = idx[j];
                    const size_t row = (size_t)(a.row0_d[i] + a.edst[i][e]);
                    const float4 gv = *reinterpret_cast<const float4*>(a.g + row * a.ld_g + c);
                    const uchar4 bi = *reinterpret_cast<const uchar4*>(a.arg + row * D + c);
                    const float* ae = a.A[i] + (size_t)e * H;
                    const float* me = a.Mk[i] != nullptr ? a.Mk[i] + (size_t)e * H : nullptr;
                    float p[MAXH];
                    if (H == MAXH) {                   // the 8 soft-max values (and multipliers) of an edge: 16-byte loads
                        const float4 a0 = *reinterpret_cast<const float4*>(ae), a1 = *reinterpret_cast<const float4*>(ae + 4);
                        p[0] = a0.x; p[1] = a0.y; p[2] = a0.z; p[3] = a0.w; p[4] = a1.x; p[5] = a1.y; p[6] = a1.z; p[7] = a1.w;
                        if (me != nullptr) {
                            const float4 m0 = *reinterpret_cast<const float4*>(me), m1 = *reinterpret_cast<const float4*>(me + 4);
                            p[0] *= m0.x; p[1] *= m0.y; p[2] *= m0.z; p[3] *= m0.w; p[4] *= m1.x; p[5] *= m1.y; p[6] *= m1.z; p[7] *= m1.w;
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < MAXH; ++h) p[h] = h < H ? (me != nullptr ? ae[h] * me[h] : ae[h]) : 0.f;
                    }
#pragma unroll
                    for (int h = 0; h < MAXH; ++h) {
                        if (h < H) {
                            o[h].x += bi.x == h ? p[h] * gv.x : 0.f; o[h].y += bi.y == h ? p[h] * gv.y : 0.f;
                            o[h].z += bi.z == h ? p[h] * gv.z : 0.f; o[h].w += bi.w == h ? p[h] * gv.w : 0.f;
                        }
                    }
                }
            }
        }
    }
    if (!DER_FIRST && live && hl < H)
        for (int q = 0; q < a.ndst[b]; ++q) wr += a.der[a.dst[b][q]][(size_t)u * H + hl];
    HGT_W(3);
    __syncthreads();                                   // a_l | a_r are in LDS
    HGT_W(4);
    if (!in_range) return;
    if (!live && a.skip_dead) {                        // 29 % of the rows at the bench capacities: 36 MB of zeros not written
        if (lane < H) { a.wL[b][(size_t)u * H + lane] = 0.f; a.wR[b][(size_t)u * H + lane] = 0.f; }
        return;
    }
    T* dp = static_cast<T*>(a.dP[b]) + (size_t)u * HD + c;
#pragma unroll
    for (int h = 0; h < MAXH; ++h) {
        if (h < H) {
            const float wlh = __shfl(wl, h, 64), wrh = __shfl(wr, h, 64);
            if (c < D) {
                const float4 l4 = *reinterpret_cast<const float4*>(alr + h * D + c);
                const float4 r4 = *reinterpret_cast<const float4*>(alr + HD + h * D + c);
                float4 v = o[h];
                v.x += wlh * l4.x + wrh * r4.x; v.y += wlh * l4.y + wrh * r4.y;
                v.z += wlh * l4.z + wrh * r4.z; v.w += wlh * l4.w + wrh * r4.w;
                st4(dp + h * D, v);
            }
        }
    }
    if (lane < H) {
        a.wL[b][(size_t)u * H + lane] = wl;
        a.wR[b][(size_t)u * H + lane] = wr;
    }
    HGT_W(5);
    HGT_END();
}

// two-stage ordered reductions over the nodes:
//   job t < nt      : bias column sums  CS_t[h,c] = sum_v g[v,c] * [arg[v,c] == h]   over the rows of node type t
//   job nt + m      : Z[m][lr][c][h] = sum over the module's blocks of sum_u x[u,c] * w_lr[u,h]   (w = wL / wR)
// d attn_l[hD+j] = sum_u wL[u,h] P[u,h,j] = sum_c W[hD+j,c] Z[0][c][h]  (P = x W^T): the weighted column sums over
// the [N, H*D] projections become a reduction over x (8x smaller) plus a pass over W (hg_dattn_kernel).
struct ColArgs {
    const float* wL[MAXB]; const float* wR[MAXB];
    const int* dyn_b[MAXB];
    int ncap_b[MAXB], row0_b[MAXB];
    int mod_nb[MAXM], mod_blk[MAXM][4];
    const int* dyn_t[MAXT];
    int row0[MAXT + 1], ncap_t[MAXT];
    int nt, nm, H, D;
    const float* g; int ld_g;
    const unsigned char* arg;
    const float* xb[MAXB]; int ld_x;
    float* part;             // [nt + nm][NCHUNK][2 * H*D]
};

__global__ void hg_colsum_part_kernel(ColArgs a) {
    __shared__ float red[2][4][64];
    const int H = a.H, D = a.D, HD = H * D;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int job = blockIdx.z, chunk = blockIdx.y;
    float s0 = 0.f, s1 = 0.f;
    if (col < HD) {
        const int h = col / D, c = col % D;
        if (job < a.nt) {
            const int t = job;
            const int n = dyn_count(a.dyn_t[t], a.ncap_t[t]);
            const int per = (n + NCHUNK - 1) / NCHUNK, r0 = chunk * per, r1 = min(n, r0 + per);
            // independent loads 4 rows deep: an in-order wave otherwise pays one memory latency per row
            int r = r0 + rg;
            for (; r + 12 < r1; r += 16) {
                float gv[4]; unsigned char av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t row = (size_t)(a.row0[t] + r + 4 * e);
                    av[e] = a.arg[row * D + c]; gv[e] = a.g[row * a.ld_g + c];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s0 += av[e] == h ? gv[e] : 0.f;
            }
            for (; r < r1; r += 4) {
                const size_t row = (size_t)(a.row0[t] + r);
                if (a.arg[row * D + c] == h) s0 += a.g[row * a.ld_g + c];
            }
        } else {
            const int m = job - a.nt;
            for (int q = 0; q < a.mod_nb[m]; ++q) {
                const int b = a.mod_blk[m][q];
                const int n = dyn_count(a.dyn_b[b], a.ncap_b[b]);
                const int per = (n + NCHUNK - 1) / NCHUNK, r0 = chunk * per, r1 = min(n, r0 + per);
                int r = r0 + rg;
                for (; r + 12 < r1; r += 16) {
                    float xv[4], lv[4], rv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xv[e] = a.xb[b][(size_t)(a.row0_b[b] + r + 4 * e) * a.ld_x + c];
                        lv[e] = a.wL[b][(size_t)(r + 4 * e) * H + h];
                        rv[e] = a.wR[b][(size_t)(r + 4 * e) * H + h];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s0 += xv[e] * lv[e]; s1 += xv[e] * rv[e]; }
                }
                for (; r < r1; r += 4) {
                    const float xv = a.xb[b][(size_t)(a.row0_b[b] + r) * a.ld_x + c];
                    s0 += xv * a.wL[b][(size_t)r * H + h];
                    s1 += xv * a.wR[b][(size_t)r * H + h];
                }
            }
        }
    }
    red[0][rg][threadIdx.x & 63] = s0;
    red[1][rg][threadIdx.x & 63] = s1;
    __syncthreads();
    if (rg == 0 && col < HD) {
        const int l = threadIdx.x;
        float* p = a.part + ((size_t)job * NCHUNK + chunk) * 2 * HD;
        p[col] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        p[HD + col] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
    }
}

// H == 8, D <= 256, D % 4 == 0: a thread carries FOUR feature columns with all heads, one job per node type (bias sums) and per
// projection BLOCK (Z sums), 32 row chunks each.  The head-per-workgroup version above re-reads every x / g row once per head out
// of L2 (~310 MB for 39 MB of rows: 31 us, whatever its launch shape); with all heads on the column's thread a row is read
// once.  Thread = (4 columns, row lane of 4): a 16-row tile is 4 sixteen-byte loads per thread instead of 16 four-byte ones (the
// one-column-per-thread version was bound by the ISSUE of its loads: 68 vector-memory instructions per tile and workgroup for
// 1 k cycles of FMAs); the per-(row, head) factors of the tile go through LDS (broadcast reads); the four row lanes are summed
// through LDS at the end, 16 accumulators per pass.
__global__ __launch_bounds__(256) void hg_colsum_cols_kernel(ColArgs a) {
    __shared__ __attribute__((aligned(16))) float sw[16][16];
    __shared__ __attribute__((aligned(16))) float red4[4][64][4];
    const int D = a.D, HD = 8 * D;
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6, c = 4 * cg;
    const int job = blockIdx.y, chunk = blockIdx.x;
    const bool cok = c < D;
    float4 s0[8], s1[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) { s0[h] = make_float4(0.f, 0.f, 0.f, 0.f); s1[h] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (job < a.nt) {
        const int t = job;
        const int n = dyn_count(a.dyn_t[t], a.ncap_t[t]);
        const int per = (n + NCHUNK_C - 1) / NCHUNK_C, r0 = chunk * per, r1 = min(n, r0 + per);
        if (cok)
            for (int r = r0 + rl; r < r1; r += 16) {
                float4 gv[4]; unsigned av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t row = (size_t)(a.row0[t] + min(r + 4 * e, r1 - 1));
                    gv[e] = *reinterpret_cast<const float4*>(a.g + row * a.ld_g + c);
                    av[e] = r + 4 * e < r1 ? *reinterpret_cast<const unsigned*>(a.arg + row * D + c) : 0xffffffffu;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        s0[h].x += (av[e] & 0xffu) == (unsigned)h ? gv[e].x : 0.f;
                        s0[h].y += ((av[e] >> 8) & 0xffu) == (unsigned)h ? gv[e].y : 0.f;
                        s0[h].z += ((av[e] >> 16) & 0xffu) == (unsigned)h ? gv[e].z : 0.f;
                        s0[h].w += (av[e] >> 24) == (unsigned)h ? gv[e].w : 0.f;
                    }
            }
    } else {
        const int b = job - a.nt;
        const int n = dyn_count(a.dyn_b[b], a.ncap_b[b]);
        const int per = (n + NCHUNK_C - 1) / NCHUNK_C, r0 = chunk * per, r1 = min(n, r0 + per);
        const float* __restrict__ xb = a.xb[b] + (size_t)a.row0_b[b] * a.ld_x + (cok ? c : 0);
        const float* __restrict__ wL = a.wL[b];
        const float* __restrict__ wR = a.wR[b];
        for (int r = r0; r < r1; r += 16) {
            float4 xv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                xv[e] = (cok && r + rl + 4 * e < r1) ? *reinterpret_cast<const float4*>(xb + (size_t)(r + rl + 4 * e) * a.ld_x)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();                               // the previous tile's factors have been read
            {
                const int row = threadIdx.x >> 4, k = threadIdx.x & 15;
                float w = 0.f;
                if (r + row < r1) w = k < 8 ? wL[(size_t)(r + row) * 8 + k] : wR[(size_t)(r + row) * 8 + k - 8];
                sw[row][k] = w;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* f = &sw[rl + 4 * e][0];
                const float4 l0 = *reinterpret_cast<const float4*>(f), l1 = *reinterpret_cast<const float4*>(f + 4);
                const float4 q0 = *reinterpret_cast<const float4*>(f + 8), q1 = *reinterpret_cast<const float4*>(f + 12);
                const float lw[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                const float qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    s0[h].x += xv[e].x * lw[h]; s0[h].y += xv[e].y * lw[h]; s0[h].z += xv[e].z * lw[h]; s0[h].w += xv[e].w * lw[h];
                    s1[h].x += xv[e].x * qw[h]; s1[h].y += xv[e].y * qw[h]; s1[h].z += xv[e].z * qw[h]; s1[h].w += xv[e].w * qw[h];
                }
            }
        }
    }
    // the four row lanes of a column group -> one partial: 16 float4 accumulators, one LDS pass each (row lanes added in order)
    float* p = a.part + ((size_t)job * NCHUNK_C + chunk) * 2 * HD + c;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 v = k < 8 ? s0[k] : s1[k - 8];
        __syncthreads();
        *reinterpret_cast<float4*>(&red4[rl][cg][0]) = v;
        __syncthreads();
        if (rl == 0 && cok) {
            const float4 v1 = *reinterpret_cast<const float4*>(&red4[1][cg][0]), v2 = *reinterpret_cast<const float4*>(&red4[2][cg][0]),
                         v3 = *reinterpret_cast<const float4*>(&red4[3][cg][0]);
            const float4 o = make_float4(((v.x + v1.x) + v2.x) + v3.x, ((v.y + v1.y) + v2.y) + v3.y, ((v.z + v1.z) + v2.z) + v3.z,
                                         ((v.w + v1.w) + v2.w) + v3.w);
            *reinterpret_cast<float4*>(p + (k < 8 ? k * D : HD + (k - 8) * D)) = o;
        }
    }
}

// out o < nm: d_bias[m][h*D + c] = sum over the module's instances of CS_{dst type};  o >= nm: Z[m] ([2][H][D])
struct ColFinalArgs {
    float* out[2 * MAXM];
    int njob[2 * MAXM], jobs[2 * MAXM][8];
    int nm, H, D, nchunk;
    const float* part;
};

// s + p[0] + p[stride] + ... (n terms, added in that order): eight loads in flight instead of one dependent load per term
__device__ __forceinline__ float chunk_sum(const float* __restrict__ p, int n, size_t stride, float s) {
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[(size_t)(k + e) * stride];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; k < n; ++k) s += p[(size_t)k * stride];
    return s;
}

__global__ void hg_colsum_final_kernel(ColFinalArgs a) {
    const int HD = a.H * a.D, o = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // < 2*HD
    if (a.out[o] == nullptr) return;
    if (o < a.nm) {                                              // bias: first half of the slabs only
        if (idx >= HD) return;
        float s = 0.f;
        for (int q = 0; q < a.njob[o]; ++q) {
            s = chunk_sum(a.part + (size_t)a.jobs[o][q] * a.nchunk * 2 * HD + idx, a.nchunk, 2 * HD, s);
        }
        a.out[o][idx] = s;
    } else {
        if (idx >= 2 * HD) return;
        float s = 0.f;
        for (int q = 0; q < a.njob[o]; ++q) {                     // one job per module, or one per projection block of it
            s = chunk_sum(a.part + (size_t)a.jobs[o][q] * a.nchunk * 2 * HD + idx, a.nchunk, 2 * HD, s);
        }
        a.out[o][idx] = s;                                        // Z [lr][h][c] = the slab order (16-byte reads in hg_dattn)
    }
}

struct DattnArgs {
    const float* W[MAXM]; const float* Z[MAXM];
    float* dal[MAXM]; float* dar[MAXM];
    int H, D;
};

// one wavefront per fc row r = h*D + j of a module: d attn_l[r] = <W[r,:], Z[0][h,:]>, d attn_r[r] = <W[r,:], Z[1][h,:]>
__global__ void hg_dattn_kernel(DattnArgs a) {
    const int H = a.H, D = a.D, HD = H * D;
    const int gid = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int m = gid / HD, r = gid % HD, h = r / D;
    const float* w = a.W[m] + (size_t)r * D;
    const float* z = a.Z[m];
    float sl = 0.f, sr = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(w + c);
        const float4 zl = *reinterpret_cast<const float4*>(z + (size_t)h * D + c);
        const float4 zr = *reinterpret_cast<const float4*>(z + (size_t)HD + (size_t)h * D + c);
        sl += wv.x * zl.x + wv.y * zl.y + wv.z * zl.z + wv.w * zl.w;
        sr += wv.x * zr.x + wv.y * zr.y + wv.z * zr.z + wv.w * zr.w;
    }
    sl = wave_sum(sl);
    sr = wave_sum(sr);
    if (lane == 0) { a.dal[m][r] = sl; a.dar[m][r] = sr; }
}

bool bad_desc(const srec_hg_desc* d) {
    return d == nullptr || d->H <= 0 || d->H > MAXH || d->D <= 0 || d->D > 256 || (d->D & 3) || d->n_types <= 0 ||
           d->n_types > MAXT || d->n_blocks < 0 || d->n_blocks > MAXB || d->n_inst < 0 || d->n_inst > MAXI ||
           d->n_mods < 0 || d->n_mods > MAXM;
}

}  // namespace

extern "C" int srec_hg_ws_floats(const void* desc_, long* n_floats) {
    const srec_hg_desc* d = (const srec_hg_desc*)desc_;
    if (bad_desc(d) || n_floats == nullptr) return SREC_BAD_ARG;
    const long a = (long)(d->n_types + d->n_mods) * NCHUNK, b = (long)(d->n_types + d->n_blocks) * NCHUNK_C;
    *n_floats = (a > b ? a : b) * 2 * d->H * d->D;
    return 0;
}

extern "C" int srec_hg_fwd(const void* desc_, const float* x, int ld_x, float* out, int ld_out, unsigned char* arg,
                           void* stream) {
    const srec_hg_desc* d = (const srec_hg_desc*)desc_;
    if (bad_desc(d) || (ld_x & 3)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int H = d->H, D = d->D, HD = H * D;
    const size_t esz = (d->p16 & 1) ? 2 : 4;
    // (bit 3 of p16: V and the bias sums of this descriptor's weights were written since the optimizer last changed them - by
    // srec_hg_fold or, in a step that starts with srec_step_prep, by the prologue launch)
    if (d->n_mods > 0 && !(d->p16 & 8))
        if (int rc = srec_hg_fold(d, stream)) return rc;
    if (d->sess == nullptr || d->B <= 0) return SREC_BAD_ARG;
    for (int t = 0; t < d->n_types; ++t)
        if (d->smean[t] == nullptr) return SREC_BAD_ARG;
    {
        DotsArgs a{};
        a.nb = d->n_blocks; a.H = H; a.D = D; a.ld_x = ld_x;
        int blocks = 0;
        for (int b = 0; b < d->n_blocks; ++b) {
            const int m = d->blk_mod[b], t = d->blk_type[b];
            a.V[b] = d->V[m];
            a.x[b] = d->xin[m] != nullptr ? d->xin[m] : x;
            a.eL[b] = d->eL[b]; a.eR[b] = d->eR[b];
            a.dyn[b] = d->dyn_n[t]; a.ncap[b] = d->ncap[t]; a.row0[b] = d->row0[t];
            a.start[b] = blocks;
            blocks += cdiv(d->ncap[t], DOTS_NODES);
        }
        a.start[d->n_blocks] = blocks;
        // + one workgroup per (node type, session): session means and the node -> session map (see DotsArgs)
        a.nt = d->n_types; a.B = d->B; a.dynB = d->dynB; a.xm = x; a.sess = d->sess;
        for (int t = 0; t < d->n_types; ++t) { a.seg[t] = d->seg[t]; a.trow0[t] = d->row0[t]; a.smean[t] = d->smean[t]; }
        blocks += d->n_types * d->B;
        if (blocks > 0) hipLaunchKernelGGL(hg_dots_kernel, dim3(blocks), dim3(256), (size_t)(16 * (D + 4) + 16) * 4, st, a);
    }
    AggArgs g{};
    g.nt = d->n_types; g.B = d->B; g.dynB = d->dynB; g.H = H; g.D = D; g.slope = d->slope;
    g.x = x; g.ld_x = ld_x; g.out = out; g.ld_out = ld_out; g.arg = arg; g.xres = d->xres; g.sess = d->sess;
    for (int t = 0; t < d->n_types; ++t) g.smean[t] = d->smean[t];
    int rows = 0;
    for (int t = 0; t < d->n_types; ++t) {
        if (d->row0[t] != rows) return SREC_BAD_ARG;            // types must tile the stacked matrix
        g.seg[t] = d->seg[t]; g.dyn_n[t] = d->dyn_n[t]; g.row0[t] = d->row0[t]; g.ncap[t] = d->ncap[t];
        g.ninst[t] = 0;
        g.bsum[t] = d->Z[t];
        rows += d->ncap[t];
    }
    g.row0[d->n_types] = rows;
    for (int i = 0; i < d->n_inst; ++i) {
        const int sb = d->inst_sblk[i], db = d->inst_dblk[i], m = d->inst_mod[i], t = d->blk_type[db];
        if (g.ninst[t] >= 8) return SREC_BAD_ARG;
        g.inst[t][g.ninst[t]++] = i;
        g.Ps[i] = (const char*)d->P[m] + (size_t)d->blk_row[sb] * HD * esz;
        g.eLs[i] = d->eL[sb]; g.eRd[i] = d->eR[db]; g.bias[i] = d->bias[m];
        g.in_ptr[i] = d->in_ptr[i]; g.in_idx[i] = d->in_idx[i]; g.esrc[i] = d->esrc[i];
        g.A[i] = d->A[i]; g.Mk[i] = d->Mk[i];
    }
    if (rows > 0) {
        // wave-per-node kernel: 8 heads, 8-column lanes, 16-byte loads of the per-(node | edge, head) arrays
        auto al16 = [](const void* p) { return ((size_t)p & 15) == 0; };
        bool node = H == MAXH && (D & 7) == 0 && (ld_x & 3) == 0 && (ld_out & 3) == 0 && al16(out) && al16(arg) &&
                    al16(x) && al16(g.xres);
        for (int t = 0; node && t < d->n_types; ++t) node = al16(g.smean[t]) && (g.ninst[t] == 0 || al16(g.bsum[t]));
        for (int i = 0; node && i < d->n_inst; ++i)
            node = al16(g.Ps[i]) && al16(g.eLs[i]) && al16(g.eRd[i]) && al16(g.A[i]) && al16(g.Mk[i]);
        if (node) {
            // (same box, rocprof: 1 edge in flight at 5 waves per SIMD 31.6 us; 2 edges at 4: 32.5; anything that spills: 38 - 97)
            if (d->p16 & 1) hipLaunchKernelGGL((hg_agg_node_kernel<unsigned short, 1, 5>), dim3(cdiv(rows, WPB)), dim3(64 * WPB), 0, st, g);
            else hipLaunchKernelGGL((hg_agg_node_kernel<float, 2, 3>), dim3(cdiv(rows, WPB)), dim3(64 * WPB), 0, st, g);
        } else if (d->p16 & 1) hipLaunchKernelGGL(hg_agg_kernel<unsigned short>, dim3(rows), dim3(512), 0, st, g);
        else hipLaunchKernelGGL(hg_agg_kernel<float>, dim3(rows), dim3(512), 0, st, g);
    }
    SREC_LAUNCH_CHECK();
    return 0;
}

static int fill_pre(const srec_hg_desc* d, const float* g, int ld_g, float* dx, int ld_dx, PreArgs& a) {
    int ninst_t[MAXT] = {0, 0, 0, 0};
    for (int i = 0; i < d->n_inst; ++i) ninst_t[d->blk_type[d->inst_dblk[i]]]++;
    a.nt = d->n_types; a.B = d->B; a.D = d->D; a.dynB = d->dynB; a.g = g; a.ld_g = ld_g; a.dx = dx; a.ld_dx = ld_dx; a.rm = d->rm;
    a.rm_cnt = d->rm_cnt; a.rm_p = d->rm_p; a.rm_rng = srec_rng{(unsigned)d->rm_seed, d->rm_counter, (unsigned)d->rm_salt, d->rm_p};
    a.sess = d->sess;
    if (d->sess == nullptr) return SREC_BAD_ARG;
    int rows = 0;
    for (int t = 0; t < d->n_types; ++t) {
        a.seg[t] = d->seg[t]; a.dyn_n[t] = d->dyn_n[t]; a.row0[t] = d->row0[t]; a.ncap[t] = d->ncap[t];
        a.ninst[t] = ninst_t[t];
        rows += d->ncap[t];
    }
    a.row0[d->n_types] = rows;
    return 0;
}

// the layer's d x in one pass, AFTER the backward-data GEMMs of a feature-dropout call (desc.p16 bit 2 made srec_hg_bwd leave
// d x alone): dx [NT, D] = residual + session-mean terms of g (as srec_hg_bwd would have written) + the two convs' masked data
// gradients t [2, S, NT, D] (as srec_hg_drop_merge would have added); masks from desc.rm_p / rm_seed / rm_counter / rm_salt.
extern "C" int srec_hg_pre_merge(const void* desc_, const float* g, int ld_g, const float* t, int S, float* dx, int ld_dx,
                                 void* stream) {
    const srec_hg_desc* d = (const srec_hg_desc*)desc_;
    if (bad_desc(d) || (ld_g & 3) || (ld_dx & 3) || t == nullptr || S < 1 || d->rm_cnt == nullptr) return SREC_BAD_ARG;
    PreArgs a{};
    if (int rc = fill_pre(d, g, ld_g, dx, ld_dx, a)) return rc;
    const int rows = a.row0[d->n_types];
    if ((long)2 * rows * d->D > 0xffffffffL) return SREC_BAD_ARG;
    if (rows > 0) hipLaunchKernelGGL(hg_pre_kernel<true>, dim3(cdiv(rows, WPB)), dim3(256), 0, (hipStream_t)stream, a, t, S);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_hg_bwd(const void* desc_, const float* x, int ld_x, const float* g, int ld_g, const unsigned char* arg,
                           float* dx, int ld_dx, float* ws, void* stream) {
    const srec_hg_desc* d = (const srec_hg_desc*)desc_;
    if (bad_desc(d) || (ld_g & 3) || (ld_dx & 3) || (ld_x & 3) || ws == nullptr) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int H = d->H, D = d->D, HD = H * D;
    const size_t esz = (d->p16 & 1) ? 2 : 4;
    int ninst_t[MAXT] = {0, 0, 0, 0};
    for (int i = 0; i < d->n_inst; ++i) ninst_t[d->blk_type[d->inst_dblk[i]]]++;
    int rows = 0;
    for (int t = 0; t < d->n_types; ++t) rows += d->ncap[t];
    if (!(d->p16 & 4)) {                       // (bit 2: the caller finishes d x with srec_hg_pre_merge after its GEMMs)
        PreArgs a{};
        if (int rc = fill_pre(d, g, ld_g, dx, ld_dx, a)) return rc;
        if (rows > 0) hipLaunchKernelGGL(hg_pre_kernel<false>, dim3(cdiv(rows, WPB)), dim3(256), 0, st, a, (const float*)nullptr, 0);
    }
    if (d->n_inst > 0) {
        DstArgs a{};
        a.ni = d->n_inst; a.H = H; a.D = D; a.slope = d->slope; a.g = g; a.ld_g = ld_g; a.arg = arg;
        int blocks = 0;
        for (int i = 0; i < d->n_inst; ++i) {
            const int sb = d->inst_sblk[i], db = d->inst_dblk[i], m = d->inst_mod[i], t = d->blk_type[db];
            a.Ps[i] = (const char*)d->P[m] + (size_t)d->blk_row[sb] * HD * esz;
            a.eLs[i] = d->eL[sb]; a.eRd[i] = d->eR[db]; a.A[i] = d->A[i];
            a.in_ptr[i] = d->in_ptr[i]; a.in_idx[i] = d->in_idx[i]; a.esrc[i] = d->esrc[i];
            a.DP[i] = d->DP[i]; a.der[i] = d->der[i]; a.Mk[i] = d->Mk[i];
            a.dyn_d[i] = d->dyn_n[t]; a.ncap_d[i] = d->ncap[t]; a.row0_d[i] = d->row0[t];
            a.start[i] = blocks;
            blocks += cdiv(d->ncap[t], WPB);
        }
        a.start[d->n_inst] = blocks;
        // one wavefront per destination node over all its instances (H == 8, <= 8 instances per type, 16-byte aligned arrays)
        auto al16 = [](const void* p) { return ((size_t)p & 15) == 0; };
        bool node = H == MAXH && (D & 7) == 0 && al16(g) && al16(arg);   // 8-column lanes
        a.nt = d->n_types;
        int trows = 0;
        for (int t = 0; t < d->n_types; ++t) {
            a.dyn_t[t] = d->dyn_n[t]; a.row0_t[t] = d->row0[t]; a.ncap_t[t] = d->ncap[t]; a.ninst_t[t] = 0;
            trows += d->ncap[t];
        }
        a.row0_t[d->n_types] = trows;
        for (int i = 0; i < d->n_inst; ++i) {
            const int t = d->blk_type[d->inst_dblk[i]];
            if (a.ninst_t[t] >= 8) { node = false; continue; }
            a.inst_t[t][a.ninst_t[t]++] = i;
            node = node && al16(a.A[i]) && al16(a.Mk[i]) && al16(a.eLs[i]) && al16(a.eRd[i]) && al16(a.DP[i]) && al16(a.der[i]);
        }
        if (node && trows > 0) {
            // (same box, rocprof: 2 edges in flight at 4 waves per SIMD 28.2 us; 1 edge 30.2; 5 waves per SIMD spill: 32.7 / 36.1)
            if (d->p16 & 1) hipLaunchKernelGGL((hg_bwd_dst_node_kernel<unsigned short, 2, 4>), dim3(cdiv(trows, WPB)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((hg_bwd_dst_node_kernel<float, 1, 3>), dim3(cdiv(trows, WPB)), dim3(256), 0, st, a);
        } else if (blocks > 0) {
            if (d->p16 & 1) hipLaunchKernelGGL(hg_bwd_dst_kernel<unsigned short>, dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(hg_bwd_dst_kernel<float>, dim3(blocks), dim3(256), 0, st, a);
        }
    }
    if (d->n_blocks > 0) {
        SrcArgs a{};
        a.nb = d->n_blocks; a.H = H; a.D = D; a.g = g; a.ld_g = ld_g; a.arg = arg;
        a.skip_dead = (d->p16 & 2) ? 1 : 0;
        int blocks = 0;
        for (int b = 0; b < d->n_blocks; ++b) {
            const int m = d->blk_mod[b], t = d->blk_type[b];
            a.dP[b] = (char*)d->dP[m] + (size_t)d->blk_row[b] * HD * esz;
            a.wL[b] = d->wL[b]; a.wR[b] = d->wR[b]; a.al[b] = d->attn_l[m]; a.ar[b] = d->attn_r[m];
            a.dyn[b] = d->dyn_n[t]; a.ncap[b] = d->ncap[t];
            a.nsrc[b] = a.ndst[b] = 0;
            a.start[b] = blocks;
            blocks += cdiv(d->ncap[t], WPB_SRC);
        }
        a.start[d->n_blocks] = blocks;
        for (int i = 0; i < d->n_inst; ++i) {
            const int sb = d->inst_sblk[i], db = d->inst_dblk[i];
            if (a.nsrc[sb] >= 4 || a.ndst[db] >= 4) return SREC_BAD_ARG;
            a.src[sb][a.nsrc[sb]++] = i;
            a.dst[db][a.ndst[db]++] = i;
            a.A[i] = d->A[i]; a.DP[i] = d->DP[i]; a.der[i] = d->der[i]; a.Mk[i] = d->Mk[i];
            a.out_ptr[i] = d->out_ptr[i]; a.out_idx[i] = d->out_idx[i]; a.edst[i] = d->edst[i];
            a.row0_d[i] = d->row0[d->blk_type[db]];
        }
        if (blocks > 0) {
            const size_t lds = (size_t)2 * HD * sizeof(float);
            // (same box, rocprof: 6 waves per SIMD 47.9 us; 5: 50.3; 7: 49.4; 8 (spills): 58.6; der loads first: +1 - 2 us)
            if (d->p16 & 1) hipLaunchKernelGGL((hg_bwd_src_kernel<unsigned short, 6, false>), dim3(blocks), dim3(64 * WPB_SRC), lds, st, a);
            else hipLaunchKernelGGL((hg_bwd_src_kernel<float, 4, false>), dim3(blocks), dim3(64 * WPB_SRC), lds, st, a);
        }
    }
    {
        ColArgs a{};
        a.nt = d->n_types; a.nm = d->n_mods; a.H = H; a.D = D; a.g = g; a.ld_g = ld_g; a.arg = arg; a.part = ws;
        a.ld_x = ld_x;
        for (int b = 0; b < d->n_blocks; ++b) {
            const int m = d->blk_mod[b], t = d->blk_type[b];
            a.xb[b] = d->xin[m] != nullptr ? d->xin[m] : x;
            a.wL[b] = d->wL[b]; a.wR[b] = d->wR[b]; a.dyn_b[b] = d->dyn_n[t]; a.ncap_b[b] = d->ncap[t];
            a.row0_b[b] = d->row0[t];
            if (a.mod_nb[m] >= 4) return SREC_BAD_ARG;
            a.mod_blk[m][a.mod_nb[m]++] = b;
        }
        int r = 0;
        for (int t = 0; t < d->n_types; ++t) {
            a.dyn_t[t] = d->dyn_n[t]; a.row0[t] = d->row0[t]; a.ncap_t[t] = d->ncap[t];
            r += d->ncap[t];
        }
        a.row0[d->n_types] = r;
        const bool cols = H == 8 && D <= 256 && (D & 3) == 0 && (a.ld_x & 3) == 0 && (a.ld_g & 3) == 0;   // column-thread sums: jobs = node types + projection blocks
        ColFinalArgs f{};
        f.nm = d->n_mods; f.H = H; f.D = D; f.part = ws;
        if (cols) {
            hipLaunchKernelGGL(hg_colsum_cols_kernel, dim3(NCHUNK_C, d->n_types + d->n_blocks), dim3(256), 0, st, a);
            f.nchunk = NCHUNK_C;
        } else {
            hipLaunchKernelGGL(hg_colsum_part_kernel, dim3(cdiv(HD, 64), NCHUNK, d->n_types + d->n_mods), dim3(256), 0, st, a);
            f.nchunk = NCHUNK;
        }
        for (int m = 0; m < d->n_mods; ++m) {
            f.out[m] = d->d_bias[m];
            f.out[d->n_mods + m] = d->Z[m];
            if (cols) {
                for (int q = 0; q < a.mod_nb[m]; ++q) f.jobs[d->n_mods + m][f.njob[d->n_mods + m]++] = d->n_types + a.mod_blk[m][q];
            } else {
                f.njob[d->n_mods + m] = 1;
                f.jobs[d->n_mods + m][0] = d->n_types + m;
            }
        }
        for (int i = 0; i < d->n_inst; ++i) {
            const int m = d->inst_mod[i], t = d->blk_type[d->inst_dblk[i]];
            if (f.njob[m] >= 8) return SREC_BAD_ARG;
            f.jobs[m][f.njob[m]++] = t;
        }
        if (d->n_mods > 0) {
            hipLaunchKernelGGL(hg_colsum_final_kernel, dim3(cdiv(2 * HD, 256), 2 * d->n_mods), dim3(256), 0, st, f);
            DattnArgs q{};
            q.H = H; q.D = D;
            for (int m = 0; m < d->n_mods; ++m) { q.W[m] = d->W[m]; q.Z[m] = d->Z[m]; q.dal[m] = d->d_attn_l[m]; q.dar[m] = d->d_attn_r[m]; }
            hipLaunchKernelGGL(hg_dattn_kernel, dim3(d->n_mods * HD / WPB), dim3(256), 0, st, q);
        }
    }
    SREC_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ feature dropout glue
namespace {
// One pass for the feature dropout of a layer call (HGATLayer with drop = (p_feat, p_attn), ops.py): from the uniform draws
// u [2, n] (one per conv) and the rows x [n] (n = NT * D elements, 4 per thread):
//   ms[c] = (u[c] >= p) / (1 - p)      mask * scale of conv c (kept for the backward)
//   xc[c] = x * ms[c]                  the dropped inputs of the conv's GAT modules
//   rm    = cnt[0] * ms[0] + cnt[1] * ms[1]      per-element residual scale (cnt[c][row] = instances of conv c into the row)
//   xres  = x * rm
// blocks [0, nb1): feature masks of the two convs and the masked inputs; blocks [nb1, ...): attention-dropout multipliers
// mk [na] (all relation instances back to back).  Masks come from the counter-based hash of common.h.
__global__ void hg_drop_prep_kernel(const float* __restrict__ x, const float* __restrict__ cnt, long n, int D, long rows,
                                    float p, srec_rng rng, float* __restrict__ ms, float* __restrict__ xc,
                                    float* __restrict__ rm, float* __restrict__ xres, int nb1, float pa, long na,
                                    float* __restrict__ mk, unsigned short* __restrict__ xc16) {
    const unsigned key = srec_rng_key(rng);
    if ((int)blockIdx.x >= nb1) {
        const long j = ((long)((int)blockIdx.x - nb1) * blockDim.x + threadIdx.x) * 4;
        const float sa = 1.f / (1.f - pa);
        for (int e = 0; e < 4; ++e)
            if (j + e < na) mk[j + e] = srec_keep(key ^ 0xA5A5A5A5u, (unsigned)(j + e), pa, sa);
        return;
    }
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const long row = i / D;
    const float c0 = cnt[row], c1 = cnt[rows + row];
    const float4 xv = *reinterpret_cast<const float4*>(x + i);
    float4 m0, m1;
    const unsigned i0 = (unsigned)i, i1 = (unsigned)(n + i);
    m0.x = srec_keep(key, i0, p, sc); m0.y = srec_keep(key, i0 + 1, p, sc); m0.z = srec_keep(key, i0 + 2, p, sc); m0.w = srec_keep(key, i0 + 3, p, sc);
    m1.x = srec_keep(key, i1, p, sc); m1.y = srec_keep(key, i1 + 1, p, sc); m1.z = srec_keep(key, i1 + 2, p, sc); m1.w = srec_keep(key, i1 + 3, p, sc);
    if (ms != nullptr) {                        // (kept only on request: hg_drop_merge recomputes the masks from the hash)
        *reinterpret_cast<float4*>(ms + i) = m0;
        *reinterpret_cast<float4*>(ms + n + i) = m1;
    }
    const float4 x0 = make_float4(xv.x * m0.x, xv.y * m0.y, xv.z * m0.z, xv.w * m0.w);
    const float4 x1 = make_float4(xv.x * m1.x, xv.y * m1.y, xv.z * m1.z, xv.w * m1.w);
    *reinterpret_cast<float4*>(xc + i) = x0;
    *reinterpret_cast<float4*>(xc + n + i) = x1;
    if (xc16 != nullptr) {                       // the bf16 operand copy of the projections' GEMM, written here (no rows_bf16 pass)
        *reinterpret_cast<uint2*>(xc16 + i) = make_uint2(srec_pack_bf16(x0.x, x0.y), srec_pack_bf16(x0.z, x0.w));
        *reinterpret_cast<uint2*>(xc16 + n + i) = make_uint2(srec_pack_bf16(x1.x, x1.y), srec_pack_bf16(x1.z, x1.w));
    }
    const float4 r = make_float4(c0 * m0.x + c1 * m1.x, c0 * m0.y + c1 * m1.y, c0 * m0.z + c1 * m1.z, c0 * m0.w + c1 * m1.w);
    if (rm != nullptr) *reinterpret_cast<float4*>(rm + i) = r;    // (nullable: srec_hg_bwd recomputes it, srec_hg_desc.rm_cnt)
    *reinterpret_cast<float4*>(xres + i) = make_float4(xv.x * r.x, xv.y * r.y, xv.z * r.z, xv.w * r.w);
}

// dx += (sum_s t[0][s]) * ms[0] + (sum_s t[1][s]) * ms[1]   (the two convs' masked data gradients, each given as S
// partial sums - one per GAT module that projects the row's node type).  ms == NULL: the masks are recomputed from the
// counter-based hash of hg_drop_prep (same p / seed / counter / salt: the device counter only moves with the optimizer step)
// instead of being written there and read back here (2 x 11 MB at the bench shape).
__global__ void hg_drop_merge_kernel(const float* __restrict__ t, int S, const float* __restrict__ ms, long n,
                                     float* __restrict__ dx, float p, srec_rng rng) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a = *reinterpret_cast<const float4*>(t + i), b = *reinterpret_cast<const float4*>(t + (size_t)S * n + i);
    for (int s = 1; s < S; ++s) {
        const float4 a2 = *reinterpret_cast<const float4*>(t + (size_t)s * n + i);
        const float4 b2 = *reinterpret_cast<const float4*>(t + (size_t)(S + s) * n + i);
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
    }
    float4 m0, m1;
    if (ms != nullptr) {
        m0 = *reinterpret_cast<const float4*>(ms + i); m1 = *reinterpret_cast<const float4*>(ms + n + i);
    } else {
        const unsigned key = srec_rng_key(rng);
        const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
        const unsigned i0 = (unsigned)i, i1 = (unsigned)(n + i);
        m0.x = srec_keep(key, i0, p, sc); m0.y = srec_keep(key, i0 + 1, p, sc); m0.z = srec_keep(key, i0 + 2, p, sc); m0.w = srec_keep(key, i0 + 3, p, sc);
        m1.x = srec_keep(key, i1, p, sc); m1.y = srec_keep(key, i1 + 1, p, sc); m1.z = srec_keep(key, i1 + 2, p, sc); m1.w = srec_keep(key, i1 + 3, p, sc);
    }
    float4 d = *reinterpret_cast<float4*>(dx + i);
    d.x += a.x * m0.x + b.x * m1.x; d.y += a.y * m0.y + b.y * m1.y; d.z += a.z * m0.z + b.z * m1.z; d.w += a.w * m0.w + b.w * m1.w;
    *reinterpret_cast<float4*>(dx + i) = d;
}

}  // namespace

// x [rows, D] contiguous, cnt [2, rows]; outputs ms / xc [2, rows, D], rm / xres [rows, D] and (pa > 0) the attention-dropout
// multipliers mk [na] - all masks from the counter-based hash keyed by (seed, *counter, salt).
static int hg_drop_prep_run(const float* x, const float* cnt, int rows, int D, float p, int seed, const int* counter,
                            int salt, float* ms, float* xc, float* rm, float* xres, float pa, long na, float* mk,
                            unsigned short* xc16, void* stream) {
    if (rows <= 0) return 0;
    if (D <= 0 || (D & 3) || p < 0.f || p >= 1.f || pa < 0.f || pa >= 1.f) return SREC_BAD_ARG;
    const long n = (long)rows * D;
    if (2 * n > 0xffffffffL || na > 0xffffffffL) return SREC_BAD_ARG;
    const int nb1 = (int)((n / 4 + 255) / 256);
    const int nb2 = (pa > 0.f && na > 0 && mk != nullptr) ? (int)(((na + 3) / 4 + 255) / 256) : 0;
    hipLaunchKernelGGL(hg_drop_prep_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, (hipStream_t)stream, x, cnt, n, D,
                       (long)rows, p, srec_rng{(unsigned)seed, counter, (unsigned)salt, p}, ms, xc, rm, xres, nb1, pa, na, mk, xc16);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_hg_drop_prep(const float* x, const float* cnt, int rows, int D, float p, int seed, const int* counter,
                                 int salt, float* ms, float* xc, float* rm, float* xres, float pa, long na, float* mk,
                                 void* stream) {
    return hg_drop_prep_run(x, cnt, rows, D, p, seed, counter, salt, ms, xc, rm, xres, pa, na, mk, nullptr, stream);
}
// ... and xc16 [2, rows, D] = bf16(xc): the operand copy the bf16 projection GEMM reads (saves the srec_rows_bf16 pass)
extern "C" int srec_hg_drop_prep16(const float* x, const float* cnt, int rows, int D, float p, int seed, const int* counter,
                                   int salt, float* ms, float* xc, float* rm, float* xres, float pa, long na, float* mk,
                                   void* xc16, void* stream) {
    if (xc16 == nullptr) return SREC_BAD_ARG;
    return hg_drop_prep_run(x, cnt, rows, D, p, seed, counter, salt, ms, xc, rm, xres, pa, na, mk, (unsigned short*)xc16, stream);
}

// dx [n] += (sum_s t[0][s]) * ms[0] + (sum_s t[1][s]) * ms[1], t [2, S, n], ms [2, n] or NULL (then the masks of
// srec_hg_drop_prep with the same p / seed / counter / salt are recomputed); n % 4 == 0
extern "C" int srec_hg_drop_merge(const float* t, int S, const float* ms, long n, float* dx, float p, int seed,
                                  const int* counter, int salt, void* stream) {
    if (n <= 0) return 0;
    if ((n & 3) || S < 1 || p < 0.f || p >= 1.f || 2 * n > 0xffffffffL) return SREC_BAD_ARG;
    hipLaunchKernelGGL(hg_drop_merge_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, S,
                       ms, n, dx, p, srec_rng{(unsigned)seed, counter, (unsigned)salt, p});
    SREC_LAUNCH_CHECK();
    return 0;
}
