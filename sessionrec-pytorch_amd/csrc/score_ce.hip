// Fused full-catalog scoring + softmax cross-entropy, forward and backward ("flash-CE").
//
// Reference path replaced (logits (B,V) are materialised there, >= 4 passes fwd, more bwd):
//   logits = sr @ E^T ; log(softmax(logits)) ; nll_loss(mean)
//   srgnn.py:145-147, niser.py:149-156, lessr.py:182-183, msgifsr.py:276-309,321 + train.py:99
//
// z[b,v] = cs[v] * <sr_b, E_v>   (cs = scale / ||E_v|| for the cosine models, NULL -> 1)
// loss   = mean_b ( lse_b - z[b,label_b] ),  lse_b = log sum_v exp z[b,v]
// dS     = (softmax(z) - onehot) * cs[v] / B          (never written to HBM)
// dE_v   = sum_b dS[b,v] sr_b        d sr_b = sum_v dS[b,v] E_v
//
// One kernel template serves all four passes, flash-attention style with K == V:
// an OWNER tile X (64 rows x d) stays in LDS, 64-row chunks of the other operand Y
// are streamed through LDS (next chunk prefetched into registers while the matrix
// cores work), S = X Y^T on v_mfma_f32_32x32x2_f32 (exact fp32), P = f(S) goes to LDS and
// ACC += P Y runs on the same Y chunk.
//   MODE_FWD  X = items,    Y = sessions : per (item tile, session) online-softmax partials
//   MODE_DE   X = items,    Y = sessions : dE tile complete in registers -> one dense write
//   MODE_DSR  X = sessions, Y = items of one V-range : partial d sr, reduced by a 2nd kernel
//   MODE_LOGP X = sessions, Y = items    : log-probabilities for the nn.Module.forward API / eval
// With items as the MFMA row dimension a session's softmax statistics reduce over
// accumulator registers of ONE lane (+1 cross-half shuffle), not across lanes.
// LDS tiles are row-major with a 1-float pad: every operand read is a conflict-free
// ds_read_b32.  d <= 256 (NT = ceil(d/32) 32-wide output blocks, compile-time).
#include "common.h"
#include "score_common.h"

namespace {

enum { MODE_FWD = 0, MODE_DE = 1, MODE_DSR = 2, MODE_LOGP = 3 };

struct CEArgs {
    const float* sr; int ld_sr;
    const float* E; int ld_e;
    const float* cs;
    const int* labels;
    const float* lse;
    const float* gscale;
    const float* ga; const float* gc;   // optional per-session coefficients of the softmax / one-hot terms
    const int* dynB;
    int B, V, d;
    float* part_m; float* part_l; float* lab_logit;   // FWD
    float* dE; int ld_de; int acc_dE;                  // DE (acc_dE: add to the existing contents)
    float* part_dsr;                                   // DSR [R][B][d]
    float* logp; long ld_logp;                         // LOGP
    int chunks_per_range;
};

template <int NT>
struct Tile {
    static constexpr int DP = NT * 32;
    static constexpr int LD = DP + 1;
    static constexpr int NV = 2 * NT;   // float4 per thread per 64-row tile
};

// Tile staging.  Thread t owns the float4 column c = 4*(t % Q) of rows (t / Q) + p*(256/Q): all address
// arithmetic is loop-invariant (one base pointer + compile-time strides), loads are unconditional from a
// clamped row and zeroed afterwards, so a full tile costs 2*NT global_load_dwordx4 + 8*NT ds_write_b32 and
// almost no VALU (with one wave per SIMD every VALU cycle is a cycle the matrix pipe idles).
template <int NT>
__device__ __forceinline__ void gload_tile(float4 (&regs)[2 * NT], const float* __restrict__ src, int ld,
                                           int row0, int nrows, int d, int tid) {
    constexpr int Q = NT * 8, RS = 256 / Q;            // float4 per row, row stride between a thread's loads
    const int last = nrows - 1;
    if constexpr (256 % Q == 0) {
        const int r = tid / Q, c = (tid % Q) * 4;
        const int cc = c < d ? c : 0;
#pragma unroll
        for (int p = 0; p < 2 * NT; ++p) {
            const int row = row0 + r + p * RS;
            regs[p] = *reinterpret_cast<const float4*>(src + (size_t)min(row, last) * ld + cc);
        }
    } else {                                           // NT = 3, 6: rows do not divide the 256 threads evenly
#pragma unroll
        for (int p = 0; p < 2 * NT; ++p) {
            const int idx = tid + p * 256;
            const int row = row0 + idx / Q, c = (idx % Q) * 4;
            regs[p] = *reinterpret_cast<const float4*>(src + (size_t)min(row, last) * ld + (c < d ? c : 0));
        }
    }
}

template <int NT>
__device__ __forceinline__ void lstore_tile(float* __restrict__ tile, const float4 (&regs)[2 * NT], int row0,
                                            int nrows, int d, int tid) {
    constexpr int Q = NT * 8, RS = 256 / Q, LD = NT * 32 + 1;
    if constexpr (256 % Q == 0) {
        const int r = tid / Q, c = (tid % Q) * 4;
        const bool cok = c < d;
        float* dst = tile + r * LD + c;
#pragma unroll
        for (int p = 0; p < 2 * NT; ++p) {
            const bool ok = cok && (row0 + r + p * RS < nrows);
            const float4 v = regs[p];
            float* q = dst + p * RS * LD;
            q[0] = ok ? v.x : 0.f; q[1] = ok ? v.y : 0.f; q[2] = ok ? v.z : 0.f; q[3] = ok ? v.w : 0.f;
        }
    } else {
#pragma unroll
        for (int p = 0; p < 2 * NT; ++p) {
            const int idx = tid + p * 256;
            const int row = idx / Q, c = (idx % Q) * 4;
            const bool ok = c < d && (row0 + row < nrows);
            const float4 v = regs[p];
            float* q = tile + row * LD + c;
            q[0] = ok ? v.x : 0.f; q[1] = ok ? v.y : 0.f; q[2] = ok ? v.z : 0.f; q[3] = ok ? v.w : 0.f;
        }
    }
}

template <int NT, int MODE>
__global__ __launch_bounds__(256) void flash_ce_kernel(CEArgs a) {
    constexpr int DP = NT * 32, LD = DP + 1, PLD = 65;
    constexpr int NCB = (NT + 1) / 2;                      // output col blocks per wave
    constexpr bool ITEMS_X = (MODE == MODE_FWD || MODE == MODE_DE);
    constexpr bool HAS_ACC = (MODE == MODE_DE || MODE == MODE_DSR);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                    // [64][LD]
    float* Ys = Xs + 64 * LD;            // [64][LD]
    float* Ps = Ys + 64 * LD;            // [64][PLD]   (P tile, or FWD cross-wave scratch)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int si = wave >> 1, sj = wave & 1;
    const int Bd = dyn_count(a.dynB, a.B);
    const int d = a.d;

    // owner tile / streamed range
    const float* Xsrc; int ldx, nx, x0; const float* Ysrc; int ldy, ny, ybeg, yend;
    if (ITEMS_X) {
        Xsrc = a.E; ldx = a.ld_e; nx = a.V; x0 = blockIdx.x * 64;
        Ysrc = a.sr; ldy = a.ld_sr; ny = Bd; ybeg = 0; yend = Bd;
    } else {
        Xsrc = a.sr; ldx = a.ld_sr; nx = Bd; x0 = blockIdx.y * 64;
        Ysrc = a.E; ldy = a.ld_e; ny = a.V;
        ybeg = blockIdx.x * a.chunks_per_range * 64;
        yend = min(a.V, ybeg + a.chunks_per_range * 64);
    }
    const bool x_empty = (x0 >= nx);
    const int tile_id = ITEMS_X ? blockIdx.x : 0;

    float gs = 1.f, gm = 1.f;                  // gm: the upstream scalar on caller-given per-session coefficients ga / gc
    if (HAS_ACC) { gm = a.gscale != nullptr ? *a.gscale : 1.f; gs = gm / (float)(Bd > 0 ? Bd : 1); }

    float4 regs[2 * NT];
    gload_tile<NT>(regs, Xsrc, ldx, x0, nx > 0 ? nx : 1, d, tid);
    lstore_tile<NT>(Xs, regs, x0, nx, d, tid);

    // per-X-row quantities for this lane's 16 accumulator rows
    float xq[16];      // ITEMS_X: cs[item]   SESS_X: lse[session]
    int xlab[16];      // SESS_X: label[session]
    float xga[16], xgc[16];   // SESS_X: per-session gradient coefficients
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int xi = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ITEMS_X) {
            xq[r] = (a.cs != nullptr && xi < nx) ? a.cs[xi] : 1.f;
            xlab[r] = 0;
        } else {
            xq[r] = (MODE != MODE_FWD && xi < nx) ? a.lse[xi] : 0.f;
            xlab[r] = (MODE == MODE_DSR && xi < nx) ? a.labels[xi] : -1;
        }
        xga[r] = (MODE == MODE_DSR && a.ga != nullptr && xi < nx) ? a.ga[xi] * gm : gs;
        xgc[r] = (MODE == MODE_DSR && a.gc != nullptr && xi < nx) ? a.gc[xi] * gm : gs;
    }

    f32x16 acc[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    if (ybeg < yend && !x_empty) gload_tile<NT>(regs, Ysrc, ldy, ybeg, yend, d, tid);

    for (int y0 = ybeg; y0 < yend && !x_empty; y0 += 64) {
        lstore_tile<NT>(Ys, regs, y0, yend, d, tid);
        __syncthreads();                                                   // (A) Xs / Ys visible
        if (y0 + 64 < yend) gload_tile<NT>(regs, Ysrc, ldy, y0 + 64, yend, d, tid);

        // per-column quantities of this chunk: issued before the MFMAs so their L2 latency hides under them
        const int yj = y0 + sj * 32 + l31;           // global Y row of this lane's column
        const bool yvalid = yj < yend;
        float yq = 1.f; int ylab = -1;
        float yga = gs, ygc = gs;
        if (MODE == MODE_FWD) {
            ylab = yvalid ? a.labels[yj] : -1;
        } else if (MODE == MODE_DE) {
            yq = yvalid ? a.lse[yj] : 0.f;
            ylab = yvalid ? a.labels[yj] : -1;
            if (a.ga != nullptr) { yga = yvalid ? a.ga[yj] * gm : 0.f; ygc = yvalid ? a.gc[yj] * gm : 0.f; }
        } else {
            yq = (a.cs != nullptr && yvalid) ? a.cs[yj] : 1.f;
        }

        // ---- S = X Y^T for this wave's 32x32 sub-tile.  One wave per SIMD => nothing else hides the LDS
        // latency: operands are read one 8-step block AHEAD of the MFMAs that consume them (two register
        // sets), so the matrix pipe issues back to back.
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            const float* xa = Xs + (si * 32 + l31) * LD + half;
            const float* yb = Ys + (sj * 32 + l31) * LD + half;
            constexpr int UB = 8, NBLK = DP / 2 / UB;
            float a0[UB], b0[UB], a1[UB], b1[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) { a0[u] = xa[2 * u]; b0[u] = yb[2 * u]; }
            for (int blk = 0; blk < NBLK; blk += 2) {
                {
                    const float* xn = xa + 2 * UB * (blk + 1);
                    const float* yn = yb + 2 * UB * (blk + 1);
#pragma unroll
                    for (int u = 0; u < UB; ++u) { a1[u] = xn[2 * u]; b1[u] = yn[2 * u]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UB; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (blk + 2 < NBLK) {
                    const float* xn = xa + 2 * UB * (blk + 2);
                    const float* yn = yb + 2 * UB * (blk + 2);
#pragma unroll
                    for (int u = 0; u < UB; ++u) { a0[u] = xn[2 * u]; b0[u] = yn[2 * u]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UB; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (MODE == MODE_FWD) {
            // per-session (column) stats over this wave's 32 items
            const int lab = ylab;
            float m = -INFINITY;
            float z[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int item = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                z[r] = (item < nx) ? xq[r] * s[r] : -INFINITY;
                m = fmaxf(m, z[r]);
                if (yvalid && item == lab) a.lab_logit[yj] = z[r];
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float ms = (m == -INFINITY) ? 0.f : m;
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) l += expf(z[r] - ms);
            l += __shfl_xor(l, 32, 64);
            if (half == 0) {
                Ps[(si * 2 + 0) * 64 + sj * 32 + l31] = m;
                Ps[(si * 2 + 1) * 64 + sj * 32 + l31] = l;
            }
            __syncthreads();                                               // (B)
            if (tid < 64) {
                const int y = y0 + tid;
                if (y < yend) {
                    const float m0 = Ps[0 * 64 + tid], l0 = Ps[1 * 64 + tid];
                    const float m1 = Ps[2 * 64 + tid], l1 = Ps[3 * 64 + tid];
                    const float mm = fmaxf(m0, m1);
                    const float mms = (mm == -INFINITY) ? 0.f : mm;
                    const float ll = l0 * expf(m0 - mms) + l1 * expf(m1 - mms);
                    a.part_m[(size_t)tile_id * a.B + y] = mm;
                    a.part_l[(size_t)tile_id * a.B + y] = ll;
                }
            }
            __syncthreads();                                               // (C)
        } else if (MODE == MODE_LOGP) {
            const float csy = yq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sess = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (sess < nx && yvalid) a.logp[(size_t)sess * a.ld_logp + yj] = csy * s[r] - xq[r];
            }
            __syncthreads();                                               // (C)
        } else {
            // ---- P tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xl = si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int xi = x0 + xl;
                float p = 0.f;
                if (xi < nx && yvalid) {
                    if (ITEMS_X) {
                        const float zz = xq[r] * s[r];
                        p = (__expf(zz - yq) * yga - (ylab == xi ? ygc : 0.f)) * xq[r];
                    } else {
                        const float zz = yq * s[r];
                        p = (__expf(zz - xq[r]) * xga[r] - (xlab[r] == yj ? xgc[r] : 0.f)) * yq;
                    }
                }
                Ps[xl * PLD + sj * 32 + l31] = p;
            }
            __syncthreads();                                               // (B) P visible
            // ---- ACC += P Y : this wave owns row block si, col blocks sj, sj+2, ...  (operands of step
            // k2+1 are read while the MFMAs of step k2 run)
            {
                const float* pa = Ps + (si * 32 + l31) * PLD + half;
                const float* yb = Ys + half * LD + sj * 32 + l31;
                float av0, av1, bv0[NCB], bv1[NCB];
                av0 = pa[0];
#pragma unroll
                for (int c = 0; c < NCB; ++c) bv0[c] = (sj + 2 * c < NT) ? yb[c * 64] : 0.f;
                for (int k2 = 0; k2 < 32; k2 += 2) {
                    av1 = pa[2 * (k2 + 1)];
#pragma unroll
                    for (int c = 0; c < NCB; ++c) bv1[c] = (sj + 2 * c < NT) ? yb[(2 * (k2 + 1)) * LD + c * 64] : 0.f;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        if (sj + 2 * c < NT) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0[c], acc[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k2 + 2 < 32) {
                        av0 = pa[2 * (k2 + 2)];
#pragma unroll
                        for (int c = 0; c < NCB; ++c) bv0[c] = (sj + 2 * c < NT) ? yb[(2 * (k2 + 2)) * LD + c * 64] : 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        if (sj + 2 * c < NT) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1[c], acc[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                               // (C) before Ys/Ps reuse
        }
    }

    if (HAS_ACC) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int cb = sj + 2 * c;
            if (cb >= NT) continue;
            const int col = cb * 32 + l31;
            if (col >= d) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xi = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (MODE == MODE_DE) {
                    if (xi < a.V) {
                        float* q = a.dE + (size_t)xi * a.ld_de + col;
                        *q = a.acc_dE ? *q + acc[c][r] : acc[c][r];
                    }
                } else {
                    if (xi < a.B) a.part_dsr[((size_t)blockIdx.x * a.B + xi) * d + col] = (xi < nx) ? acc[c][r] : 0.f;
                }
            }
        }
    }
}

template <int NTV, int MODE>
int launch_nt(const CEArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * 64 * (NTV * 32 + 1) + 64 * 65) * sizeof(float);
    static std::atomic<unsigned long long> optin{0};   // > 64 KiB dynamic LDS needs the opt-in once per device
    if (int rc = srec_lds_optin((const void*)flash_ce_kernel<NTV, MODE>, (int)lds, optin)) return rc;
    hipLaunchKernelGGL((flash_ce_kernel<NTV, MODE>), grid, dim3(256), lds, st, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
int launch_mode(const CEArgs& a, dim3 grid, hipStream_t st) {
    switch ((a.d + 31) / 32) {
        case 1: return launch_nt<1, MODE>(a, grid, st);
        case 2: return launch_nt<2, MODE>(a, grid, st);
        case 3: return launch_nt<3, MODE>(a, grid, st);
        case 4: return launch_nt<4, MODE>(a, grid, st);
        case 5: case 6: return launch_nt<6, MODE>(a, grid, st);
        case 7: case 8: return launch_nt<8, MODE>(a, grid, st);
        default: return SREC_BAD_ARG;
    }
}

int pick_ranges(int B, int V) {
    const int sess_tiles = cdiv(B, 64);
    const int chunks = cdiv(V, 64);
    int R = cdiv(512, sess_tiles);            // ~2 workgroups per CU worth of blocks
    if (R > chunks) R = chunks;
    if (R < 1) R = 1;
    return R;
}

}  // namespace

extern "C" int srec_ce_plan(int B, int V, int d, int* n_item_tiles, int* n_ranges) {
    if (d <= 0 || d > 256 || (d & 3)) return SREC_BAD_ARG;
    *n_item_tiles = cdiv(V, 64);
    *n_ranges = pick_ranges(B, V);
    return 0;
}

// Forward: loss (mean CE), lse[B], lossvec[B].  ws_stats holds 2*n_item_tiles*B floats.
extern "C" int srec_score_ce_fwd(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs,
                                 const int* labels, int B, int V, int d, const int* dynB, float* ws_stats,
                                 float* lab_logit, float* lse, float* lossvec, float* loss, void* stream) {
    if (d <= 0 || d > 256 || (d & 3) || (ld_sr & 3) || (ld_e & 3)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nt = cdiv(V, 64);
    CEArgs a{};
    a.sr = sr; a.ld_sr = ld_sr; a.E = E; a.ld_e = ld_e; a.cs = cs; a.labels = labels; a.dynB = dynB;
    a.B = B; a.V = V; a.d = d;
    a.part_m = ws_stats; a.part_l = ws_stats + (size_t)nt * B; a.lab_logit = lab_logit;
    int rc = launch_mode<MODE_FWD>(a, dim3(nt), st);
    if (rc) return rc;
    hipLaunchKernelGGL(ce_reduce_stats_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, a.part_m, a.part_l, lab_logit, nt,
                       B, dynB, lse, lossvec);
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, lossvec, B, dynB, loss);
    SREC_LAUNCH_CHECK();
    return 0;
}

// Backward: dE[V,d] (dense, every row written) and dsr[B,d].  ws_dsr holds n_ranges*B*d floats.
// parts: bit0 = dE kernel, bit1 = d sr kernels (3 = both; single parts exist so bench.py can time one kernel),
// bit2 = accumulate into dE instead of overwriting it (several scoring heads on one table).
// ga/gc (nullable, both or none): per-session coefficients, dS[b,v] = (ga[b]*softmax[b,v] - gc[b]*[v==label_b]) * cs[v]
// (any loss that is a function of (lse_b, z[b,label_b]): mixtures of soft-maxes, weighted CE); default gscale/B.
extern "C" int srec_score_ce_bwd(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs,
                                 const int* labels, const float* lse, const float* gscale, const float* ga,
                                 const float* gc, int B, int V, int d, const int* dynB, float* dE, int ld_de,
                                 float* ws_dsr, float* dsr, int parts, void* stream) {
    if (d <= 0 || d > 256 || (d & 3) || (ld_sr & 3) || (ld_e & 3)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    CEArgs a{};
    a.sr = sr; a.ld_sr = ld_sr; a.E = E; a.ld_e = ld_e; a.cs = cs; a.labels = labels; a.lse = lse;
    a.gscale = gscale; a.ga = ga; a.gc = gc; a.dynB = dynB; a.B = B; a.V = V; a.d = d;
    a.dE = dE; a.ld_de = ld_de; a.part_dsr = ws_dsr; a.acc_dE = (parts & 4) ? 1 : 0;
    int rc = 0;
    if (parts & 1) rc = launch_mode<MODE_DE>(a, dim3(cdiv(V, 64)), st);
    if (rc) return rc;
    if (!(parts & 2)) return 0;
    const int R = pick_ranges(B, V);
    a.chunks_per_range = cdiv(cdiv(V, 64), R);
    rc = launch_mode<MODE_DSR>(a, dim3(R, cdiv(B, 64)), st);
    if (rc) return rc;
    const size_t n = (size_t)B * d;
    hipLaunchKernelGGL(dsr_reduce_kernel, dim3((unsigned)cdiv((int)(n / 4), 64)), dim3(256), 0, st, ws_dsr, R, n, dsr);
    SREC_LAUNCH_CHECK();
    return 0;
}

// log-probabilities (B,V) for the reference's forward() contract / evaluation.
extern "C" int srec_score_logp(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs,
                               const float* lse, int B, int V, int d, const int* dynB, float* logp, long ld_logp,
                               void* stream) {
    if (d <= 0 || d > 256 || (d & 3) || (ld_sr & 3) || (ld_e & 3)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    CEArgs a{};
    a.sr = sr; a.ld_sr = ld_sr; a.E = E; a.ld_e = ld_e; a.cs = cs; a.lse = lse; a.dynB = dynB;
    a.B = B; a.V = V; a.d = d; a.logp = logp; a.ld_logp = ld_logp;
    const int R = pick_ranges(B, V);
    a.chunks_per_range = cdiv(cdiv(V, 64), R);
    return launch_mode<MODE_LOGP>(a, dim3(R, cdiv(B, 64)), st);
}
