// srec_gemm_bf16_nt: C[m,n] = alpha * sum_k A[m,k] B[n,k] + beta*C + bias[n] on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate, 16x the fp32-MFMA rate on gfx950).
//
// Operands stay fp32 in HBM (master weights / activations); they are rounded to bf16 (RNE) while being
// staged into LDS, so no bf16 shadow copies have to be kept coherent.  Both operands are k-contiguous
// ("NT": nn.Linear forward; the backward-data product runs as NT against a transposed weight copy).
// This is the reduced-precision path BASELINE config C3 names ("bf16"); the exact fp32 path is gemm.hip.
//
// 256 threads = 4 waves (2x2); block tile 128x128x32 (wave tile 64x64 = 2x2 MFMA tiles) for large outputs,
// 64x64x32 + split-K slabs for skinny outputs with long K (every backward-data product of a wide layer).  LDS rows are
// 32 bf16 + 8 pad = 80 B: every fragment read is one 16-B ds_read_b128 and the 16 lanes of a b128 group
// land on 16 distinct 4-bank slots (20*r mod 64 distinct for r = 0..15) -> conflict free.  Global loads are
// unconditional from clamped addresses and prefetched one k-tile ahead (double-buffered LDS).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32, LDS_LD = BK + 8;   // elements (bf16)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return srec_pack_bf16(a, b); }

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(const float* __restrict__ A, int lda,
                                                           const float* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc,
                                                           const float* __restrict__ bias, int M, int N, int K,
                                                           const int* __restrict__ dyn, float alpha, float beta,
                                                           float* __restrict__ part) {
    constexpr int TM = BM / 64, TN = BN / 64, LA = BM / 32, LB = BN / 32, WM = BM / 2, WN = BN / 2;
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM][LDS_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN][LDS_LD];
    const int Mfull = M;
    M = dyn_count(dyn, M);
    // split-K: blockIdx.z owns k-tiles [kt0, kt1); raw partial sums go to part[z][Mfull][N]
    const int nsplit = gridDim.z, nk_all = K / BK;
    const int kper = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = blockIdx.z * kper, kt1 = min(nk_all, kt0 + kper);
    if (nsplit > 1) {
        C = part + (size_t)blockIdx.z * Mfull * N;
        ldc = N; bias = nullptr; alpha = 1.f; beta = 0.f;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= M) {
        if (beta == 0.f)
            for (int i = tid; i < BM * BN; i += 256) {
                const int r = m0 + i / BN, c = n0 + i % BN;
                if (r < Mfull && c < N) C[(size_t)r * ldc + c] = 0.f;
            }
        return;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread t owns the float4 at k = 4*(t%8) of rows t/8 + 32p  (p = 0..3) of both tiles
    const int sr = tid >> 3, sk = (tid & 7) * 4;
    const int Mc = M - 1, Nc = N - 1;
    float4 ra[LA], rb[LB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < LA; ++p)
            ra[p] = *reinterpret_cast<const float4*>(A + (size_t)min(m0 + sr + 32 * p, Mc) * lda + k0 + sk);
#pragma unroll
        for (int p = 0; p < LB; ++p)
            rb[p] = *reinterpret_cast<const float4*>(B + (size_t)min(n0 + sr + 32 * p, Nc) * ldb + k0 + sk);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < LA; ++p) {
            const bool ok = m0 + sr + 32 * p < M;
            uint2 v;
            v.x = ok ? pack_bf16(ra[p].x, ra[p].y) : 0u; v.y = ok ? pack_bf16(ra[p].z, ra[p].w) : 0u;
            *reinterpret_cast<uint2*>(&As[buf][sr + 32 * p][sk]) = v;
        }
#pragma unroll
        for (int p = 0; p < LB; ++p) {
            const bool ok = n0 + sr + 32 * p < N;
            uint2 v;
            v.x = ok ? pack_bf16(rb[p].x, rb[p].y) : 0u; v.y = ok ? pack_bf16(rb[p].z, rb[p].w) : 0u;
            *reinterpret_cast<uint2*>(&Bs[buf][sr + 32 * p][sk]) = v;
        }
    };
    if (kt0 < kt1) {
        gload(kt0 * BK);
        lstore(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][wm * WM + i * 32 + l31][ks * 16 + half * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][wn * WN + j * 32 + l31][ks * 16 + half * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt1) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            const float bv = (bias != nullptr && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < Mfull && col < N) {
                    float* p = C + (size_t)row * ldc + col;
                    if (row < M) {
                        float v = alpha * acc[i][j][r] + bv;
                        if (beta != 0.f) v += beta * *p;
                        *p = v;
                    } else if (beta == 0.f) {
                        *p = 0.f;
                    }
                }
            }
        }
}

// C = alpha * sum_z part[z] + bias + beta * C   (rows >= live M zeroed when beta == 0)
__global__ void splitk_reduce_bf16_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ C, int ldc,
                                          const float* __restrict__ bias, int M, int N, const int* __restrict__ dyn,
                                          float alpha, float beta) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (size_t)M * N) return;
    const int row = (int)(i / N), col = (int)(i % N);
    const int Ml = dyn_count(dyn, M);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nsplit; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)z * M * N + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* c = C + (size_t)row * ldc + col;
    if (row >= Ml) {
        if (beta == 0.f) { c[0] = 0.f; c[1] = 0.f; c[2] = 0.f; c[3] = 0.f; }
        return;
    }
    float o[4] = {alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (bias != nullptr) o[j] += bias[col + j];
        if (beta != 0.f) o[j] += beta * c[j];
        c[j] = o[j];
    }
}

// C[n,k] = alpha * sum_m A[m,n] B[m,k] + beta*C : both operands REDUCTION-major (weight gradients dW = dY^T X).
// The MFMA fragments need 8 consecutive reduction indices per lane, so the 32 x 64 fp32 global tiles are transposed
// while they are staged: a thread loads the same 4 columns of two consecutive reduction rows, packs the (m, m+1)
// pairs to bf16x2 and writes four 4-B words into the [col][m] LDS tile.  Lane -> (column group, row pair) is
// chosen so a wave's 64 words hit 64 distinct banks (80-B rows: 16c + q mod 64) while each global row is still
// read in 64-B runs.  Split over the reduction (grid z) by LIVE row count; slabs reduced by splitk_reduce_bf16.
__global__ __launch_bounds__(256) void gemm_bf16_tn_kernel(const float* __restrict__ A, int lda,
                                                           const float* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc, int Mred, int N, int K,
                                                           const int* __restrict__ dyn, float alpha, float beta,
                                                           float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][64][LDS_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][64][LDS_LD];
    const int Ml = dyn_count(dyn, Mred);
    const int nsplit = gridDim.z, nk_all = (Ml + BK - 1) / BK;
    const int kper = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = blockIdx.z * kper, kt1 = min(nk_all, kt0 + kper);
    if (nsplit > 1) {
        C = part + (size_t)blockIdx.z * N * K;
        ldc = K; alpha = 1.f; beta = 0.f;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int c4 = ((tid & 3) + 4 * (tid >> 6)) * 4, q2 = ((tid >> 2) & 15) * 2;
    const int acol = min(n0 + c4, N - 4), bcol = min(k0 + c4, K - 4);
    const bool aok = n0 + c4 < N, bok = k0 + c4 < K;
    const int Mc = Ml - 1;
    float4 ra[2], rb[2];
    auto gload = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t row = (size_t)min(m0 + q2 + i, Mc);
            ra[i] = *reinterpret_cast<const float4*>(A + row * lda + acol);
            rb[i] = *reinterpret_cast<const float4*>(B + row * ldb + bcol);
        }
    };
    auto lstore = [&](int buf, int m0) {
        const bool ok0 = m0 + q2 < Ml, ok1 = m0 + q2 + 1 < Ml;
        const float a0[4] = {ra[0].x, ra[0].y, ra[0].z, ra[0].w}, a1[4] = {ra[1].x, ra[1].y, ra[1].z, ra[1].w};
        const float b0[4] = {rb[0].x, rb[0].y, rb[0].z, rb[0].w}, b1[4] = {rb[1].x, rb[1].y, rb[1].z, rb[1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<unsigned*>(&As[buf][c4 + j][q2]) =
                pack_bf16((aok && ok0) ? a0[j] : 0.f, (aok && ok1) ? a1[j] : 0.f);
            *reinterpret_cast<unsigned*>(&Bs[buf][c4 + j][q2]) =
                pack_bf16((bok && ok0) ? b0[j] : 0.f, (bok && ok1) ? b1[j] : 0.f);
        }
    };
    if (kt0 < kt1) {
        gload(kt0 * BK);
        lstore(0, kt0 * BK);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[buf][wm * 32 + l31][ks * 16 + half * 8]);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[buf][wn * 32 + l31][ks * 16 + half * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (kt + 1 < kt1) lstore(buf ^ 1, (kt + 1) * BK);
        __syncthreads();
    }
    const int col = k0 + wn * 32 + l31;
    if (col < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < N) {
                float* p = C + (size_t)row * ldc + col;
                float v = alpha * acc[r];
                if (beta != 0.f) v += beta * *p;
                *p = v;
            }
        }
    }
}

}  // namespace

// A [M,K] (lda), B [N,K] (ldb), both k-contiguous fp32; K % 32 == 0; dyn (nullable) clamps M.
// ws (nullable): ws_floats of scratch for split-K slabs (skinny outputs with long K).
extern "C" int srec_gemm_bf16_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                                 int M, int N, int K, const int* dyn, float alpha, float beta, float* ws,
                                 long ws_floats, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K % BK) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (tiles128 >= 256) {
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<128, 128>), dim3(cdiv(N, 128), cdiv(M, 128), 1), dim3(256), 0, st, A, lda,
                           B, ldb, C, ldc, bias, M, N, K, dyn, alpha, beta, nullptr);
        SREC_LAUNCH_CHECK();
        return 0;
    }
    const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
    int nsplit = 1;
    if (ws != nullptr && tiles64 < 384 && K >= 256 && (N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0) {
        nsplit = (int)(768 / tiles64);
        if (nsplit > K / 128) nsplit = K / 128;
        if (nsplit > 16) nsplit = 16;
        while (nsplit > 1 && (long)nsplit * M * N > ws_floats) --nsplit;
        if (nsplit < 1) nsplit = 1;
    }
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<64, 64>), dim3(cdiv(N, 64), cdiv(M, 64), nsplit), dim3(256), 0, st, A, lda, B,
                       ldb, C, ldc, bias, M, N, K, dyn, alpha, beta, ws);
    if (nsplit > 1)
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)(((size_t)M * N / 4 + 255) / 256)), dim3(256), 0, st,
                           ws, nsplit, C, ldc, bias, M, N, dyn, alpha, beta);
    SREC_LAUNCH_CHECK();
    return 0;
}

// C[N,K] = alpha * A[Mred,N]^T B[Mred,K] + beta*C; A, B fp32 row-major over the reduction rows; dyn (nullable)
// clamps the reduction.  N % 4 == K % 4 == 0.  ws: split-K slabs (ws_floats >= N*K for any split to happen).
extern "C" int srec_gemm_bf16_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int Mred, int N,
                                 int K, const int* dyn, float alpha, float beta, float* ws, long ws_floats,
                                 void* stream) {
    if (N <= 0 || K <= 0) return 0;
    if (Mred <= 0 || (N & 3) || (K & 3) || (lda & 3) || (ldb & 3) || (ldc & 3) || ((uintptr_t)A & 15) ||
        ((uintptr_t)B & 15) || ((uintptr_t)C & 15))
        return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long tiles = (long)cdiv(N, 64) * cdiv(K, 64);
    int nsplit = 1;
    if (ws != nullptr) {
        nsplit = (int)(1024 / tiles);
        if (nsplit > cdiv(Mred, 128)) nsplit = cdiv(Mred, 128);
        if (nsplit > 32) nsplit = 32;
        while (nsplit > 1 && (long)nsplit * N * K > ws_floats) --nsplit;
        if (nsplit < 1) nsplit = 1;
    }
    hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3(cdiv(K, 64), cdiv(N, 64), nsplit), dim3(256), 0, st, A, lda, B, ldb, C,
                       ldc, Mred, N, K, dyn, alpha, beta, ws);
    if (nsplit > 1)
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)(((size_t)N * K / 4 + 255) / 256)), dim3(256), 0, st,
                           ws, nsplit, C, ldc, nullptr, N, K, nullptr, alpha, beta);
    SREC_LAUNCH_CHECK();
    return 0;
}
