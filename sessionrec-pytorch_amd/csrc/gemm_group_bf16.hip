// Grouped, K-segmented bf16-operand GEMM: several independent problems in ONE launch, each
//   C_p [M_p, N_p] (+)= sum_s  opA(A_{p,s}) * opB(B_{p,s})        (fp32 in HBM, bf16 MFMA operands, fp32 accumulate)
// Used by the batched MSHGNN layer (ops.HGATLayer), whose 8 GAT modules otherwise cost 24 GEMM launches + 16 split-K
// reductions + 8 weight transposes per layer:
//   forward        P_m  = x[rows_m] W_m^T            A k-contiguous, B k-contiguous       (nn.Linear, gatconv.py:166-175)
//   backward-data  dx_t = sum_{m touching t} dP_m[rows_t] W_m   A k-contiguous, B reduction-major, one segment per module:
//                  the sum over modules is the K loop, so no split-K slabs, no beta chains, no transposed weight copies
//   weight grad    dW_m = dP_m^T x[rows_m]           A and B reduction-major (reduction = node rows, clamped by dyn)
// 64x64 or 128x128 tiles, 4 waves (2x2) of 1 or 2x2 v_mfma_f32_32x32x16_bf16 accumulators each; operands are rounded to bf16 while
// staged (k-contiguous: float4 along k; reduction-major: (m, m+1) pairs packed and transposed, see gemm_bf16.hip).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MAXP = 8, MAXS = 4;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return srec_pack_bf16(a, b); }

struct GArgs {
    const void* A[MAXP][MAXS];
    const float* B[MAXP][MAXS];
    void* C[MAXP];
    const int* dyn[MAXP];
    int M[MAXP], N[MAXP], K[MAXP], nseg[MAXP], start[MAXP + 1];
    int np, lda, ldb, ldc;
    float beta;
};

// AK / BKC: operand is k-contiguous ([rows, K] row-major); otherwise reduction-major ([K, rows] row-major).
// dyn clamps the output rows M when AK (rows >= live: zeroed if beta == 0, untouched otherwise), the reduction otherwise.
// TILE = 64 (one 32x32 accumulator per wave) or 128 (2x2 accumulators per wave: one LDS fragment read per MFMA instead
// of two, 4x the MFMA work behind every global load - these loops are latency bound, not MFMA bound).
// BKT = k-tile (32 / 64).  A16: the A operands are already bf16 in HBM; C16: C is stored as bf16.
template <bool AK, bool BKC, int TILE, int BKT, bool A16, bool C16>
__global__ __launch_bounds__(256) void gemm_group_bf16_kernel(GArgs g) {
    constexpr int LD = BKT + 8;                 // bf16 elements per LDS row: 16-B fragment reads hit distinct 4-bank slots
    constexpr int TM = TILE / 64;               // accumulators per wave and dimension
    constexpr int NLK = TILE * BKT / 1024;      // float4 loads per thread, k-contiguous fp32 operand
    constexpr int NL16 = TILE * BKT / 2048;     // 16-B loads per thread, k-contiguous bf16 operand
    constexpr int NLR = TILE * BKT / 2048;      // (row pair, 4 columns) items per thread, reduction-major operand
    constexpr int CB = TILE / 32;               // 32-column blocks of a reduction-major tile
    __shared__ __attribute__((aligned(16))) unsigned short As[2][TILE][LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][TILE][LD];
    // Tile order = launch order.  An XCD-aware remap (each XCD walking one contiguous run of the tile list, i.e. one
    // module) was measured: forward / data gradient unchanged, weight gradient 141 -> 216 us (modules of unequal
    // reduction length leave XCDs idle) - these kernels are bound by staging, not by L2 misses.
    const int bid = (int)blockIdx.x;
    int p = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i)
        if (i < g.np && bid >= g.start[i]) p = i;
    const int M = g.M[p], N = g.N[p];
    const int tn = (N + TILE - 1) / TILE, tile = bid - g.start[p];
    const int m0 = (tile / tn) * TILE, n0 = (tile % tn) * TILE;
    const int live = dyn_count(g.dyn[p], AK ? M : g.K[p]);
    const int Ml = AK ? live : M, Kr = AK ? g.K[p] : live;          // live output rows, reduction length per segment
    float* __restrict__ C = static_cast<float*>(g.C[p]);
    unsigned short* __restrict__ C16p = static_cast<unsigned short*>(g.C[p]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    if (m0 >= Ml) {
        if (g.beta == 0.f)
            for (int i = tid; i < TILE * TILE; i += 256) {
                const int r = m0 + i / TILE, c = n0 + i % TILE;
                if (r < M && c < N) { if (C16) C16p[(size_t)r * g.ldc + c] = 0; else C[(size_t)r * g.ldc + c] = 0.f; }
            }
        return;
    }
    f32x16 acc[TM][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (Kr + BKT - 1) / BKT, total = nkt * g.nseg[p];
    constexpr int NA = A16 ? 1 : (AK ? NLK : 2 * NLR), NB = BKC ? NLK : 2 * NLR;
    float4 ra[NA], rb[NB];
    uint4 ra16[(A16 && AK) ? NL16 : 1];                      // A16, k-contiguous: 8 bf16 per 16-B load
    uint2 rr16[(A16 && !AK) ? 2 * NLR : 1];                  // A16, reduction-major: 4 bf16 per 8-B load
    // k-contiguous fp32 item q: row = idx / (BKT/4), k = 4 * (idx % (BKT/4)), idx = tid + 256 q;  bf16: 8 per item
    auto kc_row = [&](int q) { return (tid + 256 * q) / (BKT / 4); };
    auto kc_k = [&](int q) { return ((tid + 256 * q) % (BKT / 4)) * 4; };
    auto k16_row = [&](int q) { return (tid + 256 * q) / (BKT / 8); };
    auto k16_k = [&](int q) { return ((tid + 256 * q) % (BKT / 8)) * 8; };
    // reduction-major item q: unit u = wave + 4 q -> (32-column block u % CB, 8-row-pair block u / CB); inside a wave
    // lane & 7 = 4-column group (8 lanes = 128 contiguous bytes of one fp32 row), lane >> 3 = row pair; the packed
    // ds_write_b32 are 2-way bank conflicted, the global loads fetch full 128-B lines (64 B for bf16 rows)
    auto rm_c4 = [&](int q) { return (((wave + 4 * q) % CB) * 8 + (tid & 7)) * 4; };
    auto rm_q2 = [&](int q) { return (((wave + 4 * q) / CB) * 8 + ((tid >> 3) & 7)) * 2; };
    auto gload = [&](int it) {
        const int s = it / nkt, k0 = (it % nkt) * BKT;
        const float* __restrict__ A = static_cast<const float*>(g.A[p][s]);
        const unsigned short* __restrict__ A16p = static_cast<const unsigned short*>(g.A[p][s]);
        const float* __restrict__ B = g.B[p][s];
        if (A16 && AK) {
#pragma unroll
            for (int q = 0; q < NL16; ++q)
                ra16[q] = *reinterpret_cast<const uint4*>(A16p + (size_t)min(m0 + k16_row(q), Ml - 1) * g.lda + min(k0 + k16_k(q), Kr - 8));
        } else if (A16) {
#pragma unroll
            for (int q = 0; q < NLR; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    rr16[2 * q + e] = *reinterpret_cast<const uint2*>(A16p + (size_t)min(k0 + rm_q2(q) + e, Kr - 1) * g.lda + min(m0 + rm_c4(q), M - 4));
        } else if (AK) {
#pragma unroll
            for (int q = 0; q < NLK; ++q)
                ra[q] = *reinterpret_cast<const float4*>(A + (size_t)min(m0 + kc_row(q), Ml - 1) * g.lda + min(k0 + kc_k(q), Kr - 4));
        } else {
#pragma unroll
            for (int q = 0; q < NLR; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    ra[2 * q + e] = *reinterpret_cast<const float4*>(A + (size_t)min(k0 + rm_q2(q) + e, Kr - 1) * g.lda + min(m0 + rm_c4(q), M - 4));
        }
        if (BKC) {
#pragma unroll
            for (int q = 0; q < NLK; ++q)
                rb[q] = *reinterpret_cast<const float4*>(B + (size_t)min(n0 + kc_row(q), N - 1) * g.ldb + min(k0 + kc_k(q), Kr - 4));
        } else {
#pragma unroll
            for (int q = 0; q < NLR; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    rb[2 * q + e] = *reinterpret_cast<const float4*>(B + (size_t)min(k0 + rm_q2(q) + e, Kr - 1) * g.ldb + min(n0 + rm_c4(q), N - 4));
        }
    };
    auto lstore = [&](int buf, int it) {
        const int k0 = (it % nkt) * BKT;
        if (A16 && AK) {
#pragma unroll
            for (int q = 0; q < NL16; ++q) {
                const bool ok = (k0 + k16_k(q) < Kr) && (m0 + k16_row(q) < Ml);
                *reinterpret_cast<uint4*>(&As[buf][k16_row(q)][k16_k(q)]) = ok ? ra16[q] : make_uint4(0u, 0u, 0u, 0u);
            }
        } else if (A16) {
#pragma unroll
            for (int q = 0; q < NLR; ++q) {
                const int q2 = rm_q2(q), c4 = rm_c4(q);
                const bool cok = m0 + c4 < M;
                const bool ok0 = cok && (k0 + q2 < Kr), ok1 = cok && (k0 + q2 + 1 < Kr);
                const uint2 r0 = ok0 ? rr16[2 * q] : make_uint2(0u, 0u), r1 = ok1 ? rr16[2 * q + 1] : make_uint2(0u, 0u);
                *reinterpret_cast<unsigned*>(&As[buf][c4 + 0][q2]) = (r0.x & 0xffffu) | (r1.x << 16);
                *reinterpret_cast<unsigned*>(&As[buf][c4 + 1][q2]) = (r0.x >> 16) | (r1.x & 0xffff0000u);
                *reinterpret_cast<unsigned*>(&As[buf][c4 + 2][q2]) = (r0.y & 0xffffu) | (r1.y << 16);
                *reinterpret_cast<unsigned*>(&As[buf][c4 + 3][q2]) = (r0.y >> 16) | (r1.y & 0xffff0000u);
            }
        } else if (AK) {
#pragma unroll
            for (int q = 0; q < NLK; ++q) {
                const bool ok = (k0 + kc_k(q) < Kr) && (m0 + kc_row(q) < Ml);
                uint2 v;
                v.x = ok ? pack_bf16(ra[q].x, ra[q].y) : 0u; v.y = ok ? pack_bf16(ra[q].z, ra[q].w) : 0u;
                *reinterpret_cast<uint2*>(&As[buf][kc_row(q)][kc_k(q)]) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NLR; ++q) {
                const int q2 = rm_q2(q), c4 = rm_c4(q);
                const bool cok = m0 + c4 < M;
                const bool ok0 = cok && (k0 + q2 < Kr), ok1 = cok && (k0 + q2 + 1 < Kr);
                const float a0[4] = {ra[2 * q].x, ra[2 * q].y, ra[2 * q].z, ra[2 * q].w};
                const float a1[4] = {ra[2 * q + 1].x, ra[2 * q + 1].y, ra[2 * q + 1].z, ra[2 * q + 1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<unsigned*>(&As[buf][c4 + j][q2]) = pack_bf16(ok0 ? a0[j] : 0.f, ok1 ? a1[j] : 0.f);
            }
        }
        if (BKC) {
#pragma unroll
            for (int q = 0; q < NLK; ++q) {
                const bool ok = (k0 + kc_k(q) < Kr) && (n0 + kc_row(q) < N);
                uint2 v;
                v.x = ok ? pack_bf16(rb[q].x, rb[q].y) : 0u; v.y = ok ? pack_bf16(rb[q].z, rb[q].w) : 0u;
                *reinterpret_cast<uint2*>(&Bs[buf][kc_row(q)][kc_k(q)]) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NLR; ++q) {
                const int q2 = rm_q2(q), c4 = rm_c4(q);
                const bool cok = n0 + c4 < N;
                const bool ok0 = cok && (k0 + q2 < Kr), ok1 = cok && (k0 + q2 + 1 < Kr);
                const float b0[4] = {rb[2 * q].x, rb[2 * q].y, rb[2 * q].z, rb[2 * q].w};
                const float b1[4] = {rb[2 * q + 1].x, rb[2 * q + 1].y, rb[2 * q + 1].z, rb[2 * q + 1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<unsigned*>(&Bs[buf][c4 + j][q2]) = pack_bf16(ok0 ? b0[j] : 0.f, ok1 ? b1[j] : 0.f);
            }
        }
    };
    if (total > 0) {
        gload(0);
        lstore(0, 0);
    }
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const int buf = it & 1;
        if (it + 1 < total) gload(it + 1);
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            bf16x8 a[TM], b[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][wm * (TILE / 2) + i * 32 + l31][ks * 16 + half * 8]);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][wn * (TILE / 2) + j * 32 + l31][ks * 16 + half * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < total) lstore(buf ^ 1, it + 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int col = n0 + wn * (TILE / 2) + j * 32 + l31;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (TILE / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) {
                    if (C16) {                                   // bf16 output (beta is ignored: forward projections)
                        C16p[(size_t)row * g.ldc + col] = srec_f2bf(row < Ml ? acc[i][j][r] : 0.f);
                    } else {
                        float* q = C + (size_t)row * g.ldc + col;
                        if (row < Ml) *q = g.beta != 0.f ? acc[i][j][r] + g.beta * *q : acc[i][j][r];
                        else if (g.beta == 0.f) *q = 0.f;
                    }
                }
            }
        }
}

}  // namespace

// desc (host): srec_gemm_group (srec_hg.h).  mode 0: A, B k-contiguous (NT); 1: A k-contiguous, B reduction-major
// (NN, segments summed); 2: both reduction-major (TN, dyn clamps the reduction).
extern "C" int srec_gemm_group_bf16(const void* desc_, int mode, void* stream) {
    struct Desc {
        int np, lda, ldb, ldc;
        float beta;
        int a16, c16;
        int M[MAXP], N[MAXP], K[MAXP], nseg[MAXP];
        const void* A[MAXP][MAXS];
        const float* B[MAXP][MAXS];
        void* C[MAXP];
        const int* dyn[MAXP];
    };
    const Desc* d = (const Desc*)desc_;
    if (d == nullptr || d->np <= 0 || d->np > MAXP || mode < 0 || mode > 2 || (d->lda & 3) || (d->ldb & 3)) return SREC_BAD_ARG;
    if ((d->a16 && (mode == 0 || (d->lda & 7))) || (d->c16 && mode != 0)) return SREC_BAD_ARG;
    GArgs g{};
    g.np = d->np; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.beta = d->beta;
    // 128x128 tiles for the forward when they still give every CU a workgroup.  The reductions stay on 64x64: their
    // outputs are small (a 128-tile grid of the weight gradients is one workgroup per CU and measured slower)
    long t128 = 0;
    for (int p = 0; p < d->np; ++p) t128 += (long)cdiv(d->M[p], 128) * cdiv(d->N[p], 128);
    const int tile = (mode == 0 && t128 >= 256) ? 128 : 64;
    int blocks = 0;
    for (int p = 0; p < d->np; ++p) {
        if (d->nseg[p] <= 0 || d->nseg[p] > MAXS || d->M[p] <= 0 || d->N[p] <= 0 || d->K[p] < 4) return SREC_BAD_ARG;
        if (mode != 2 && (d->K[p] & (d->a16 ? 7 : 3))) return SREC_BAD_ARG;   // k-contiguous 16-B reads
        if ((mode == 2 && (d->M[p] & 3)) || (mode != 0 && (d->N[p] & 3))) return SREC_BAD_ARG;
        g.M[p] = d->M[p]; g.N[p] = d->N[p]; g.K[p] = d->K[p]; g.nseg[p] = d->nseg[p]; g.C[p] = d->C[p]; g.dyn[p] = d->dyn[p];
        for (int s = 0; s < d->nseg[p]; ++s) {
            if (((uintptr_t)d->A[p][s] & 15) || ((uintptr_t)d->B[p][s] & 15)) return SREC_BAD_ARG;
            g.A[p][s] = d->A[p][s]; g.B[p][s] = d->B[p][s];
        }
        g.start[p] = blocks;
        blocks += cdiv(d->M[p], tile) * cdiv(d->N[p], tile);
    }
    g.start[d->np] = blocks;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gr(blocks), bl(256);
#define SREC_GG(AK, BKC, T, BKT, A16, C16) hipLaunchKernelGGL((gemm_group_bf16_kernel<AK, BKC, T, BKT, A16, C16>), gr, bl, 0, st, g)
    // k-tile 32 for the forward (K = D: a 64-deep 128x128 stage needs 74 KB of LDS and measured 88 -> 109 us), 128 for
    // the two reductions (half the barriers of 64: 126 -> 115 us, 141 -> 137 us)
    if (mode == 0) {
        if (tile == 128) { if (d->c16) SREC_GG(true, true, 128, 32, false, true); else SREC_GG(true, true, 128, 32, false, false); }
        else { if (d->c16) SREC_GG(true, true, 64, 32, false, true); else SREC_GG(true, true, 64, 32, false, false); }
    } else if (mode == 1) {
        if (d->a16) SREC_GG(true, false, 64, 128, true, false); else SREC_GG(true, false, 64, 128, false, false);
    } else {
        if (d->a16) SREC_GG(false, false, 64, 128, true, false); else SREC_GG(false, false, 64, 128, false, false);
    }
#undef SREC_GG
    SREC_LAUNCH_CHECK();
    return 0;
}
