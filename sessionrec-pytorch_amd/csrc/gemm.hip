// srec_gemm_f32: exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[m,n] = alpha * sum_k A(m,k) * B(n,k) + beta * C[m,n] + bias[n]
//
// A(m,k) lives at A[m*a_rs + k*a_cs], B(n,k) at B[n*b_rs + k*b_cs]; for each
// operand exactly one of the two strides is 1.  That covers the three products a
// linear layer needs without any transposed copies:
//   Y  = X W^T  (nn.Linear forward)   A=X (k contiguous)  B=W  (k contiguous)
//   dX = dY W                          A=dY (k contiguous) B=W  (n contiguous)
//   dW = dY^T X                        A=dY (m contiguous) B=X  (n contiguous)
// Replaces the cuBLAS calls behind nn.Linear in the reference models
// (srgnn.py:66-68,124; lessr.py:16-17,59-62; msgifsr.py:114-116,202; gatconv.py:157).
//
// Tiling: 256 threads = 4 waves (2x2); block tile BMxBNx16, LDS tiles stored
// k-major ([k][m]) so each MFMA operand read is a conflict-free ds_read_b32 of 32
// consecutive floats per half-wave; global->register prefetch of the next k-tile
// overlaps the MFMAs of the current one (double-buffered LDS, one barrier/tile).
// A "dynamic extent" (node count living in device memory, so a captured hipGraph
// can be replayed for batches of different size) may clamp M or K.
// Skinny products (weight gradients: K = #nodes) are split along K into scratch slabs + a
// deterministic reduce so that >= 2 workgroups per CU are in flight.
#include <type_traits>
#include <utility>
#include "common.h"
#include "../../include/srec_hg.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 fp32 -> bf16 hi (round to nearest even) and bf16 lo = bf16(x - hi): x = hi + lo to ~2^-17 relative
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = srec_pack_bf16(v[2 * i], v[2 * i + 1]);
        const float a = __builtin_bit_cast(float, h[i] << 16), b = __builtin_bit_cast(float, h[i] & 0xffff0000u);
        l[i] = srec_pack_bf16(v[2 * i] - a, v[2 * i + 1] - b);
    }
    hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

template <int BM> struct TileK { static constexpr int value = BM >= 128 ? 16 : 32; };   // k-depth of one LDS tile: the
// global loads of tile t + 1 are issued before the MFMAs of tile t, so a tile has to hold about one load latency of
// matrix work (64 x 64 x 16 = 512 cycles per wave did not: the small-grid products ran at ~0.75 us per k-tile)

// One BM x BN output tile (bx, by) of split bz / nsplit.  As / Bs: the workgroup's double-buffered LDS tiles.
template <int BM, int BN, bool A_KC, bool B_KC, bool S3 = false, int BK = TileK<BM>::value>
__device__ __forceinline__ void gemm_f32_tile(
    const float* __restrict__ A, int a_rs, int a_cs, const float* __restrict__ B, int b_rs, int b_cs,
    float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N, int K,
    const int* __restrict__ dyn, int dyn_mode, float alpha, float beta, float* __restrict__ part,
    int bx, int by, int bz, int nsplit, float (*As)[BK][BM + 4], float (*Bs)[BK][BN + 4]) {
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 MFMA tiles per wave
    constexpr int LA = BM * BK / 4 / 256, LB = BN * BK / 4 / 256;   // float4 loads per thread
    // LDS layout per operand: reduction-major [k][row + 4 pad] when the operand's ROWS are contiguous in memory (float4 of
    // 4 rows stored as is, ds_read_b32 over 32 consecutive rows); row-major [row][BK] when k is contiguous (float4 of 4 k
    // stored as is, one ds_read_b128 per 4 MFMAs), 16-B pieces XOR-swizzled by the row so that the rows of one
    // ds_read_b128 lane group fall into distinct banks.  (The first version transposed k-contiguous float4s into the
    // reduction-major layout with 4 ds_write_b32 each, 4-way bank conflicted.)
    constexpr int PP = BK / 4, RPB = 64 / BK;          // pieces per row; rows per 256 B of LDS
    auto swz = [](int row) { return (row / RPB) & (PP - 1); };
    auto fa = [&](int buf) { return &As[buf][0][0]; };
    auto fb = [&](int buf) { return &Bs[buf][0][0]; };

    const int Mfull = M, Kfull = K;
    if (dyn_mode == 1) M = dyn_count(dyn, M);
    if (dyn_mode == 2) K = dyn_count(dyn, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;

    // split-K: split bz owns the k-range [kbeg, kend); raw partial sums go to part[z][Mfull][N]
    int kbeg = 0, kend = K;
    if (nsplit > 1) {
        const int kper = ((K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
        kbeg = bz * kper;
        kend = min(K, kbeg + kper);
        C = part + (size_t)bz * Mfull * N;
        ldc = N; bias = nullptr; alpha = 1.f; beta = 0.f;
    }

    if (m0 >= M) {                                    // tile fully in the padded (dynamic) region
        if (beta == 0.f) {
            for (int i = tid; i < BM * BN; i += 256) {
                int r = m0 + i / BN, c = n0 + i % BN;
                if (r < Mfull && c < N) C[(size_t)r * ldc + c] = 0.f;
            }
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

    // TWO register stages of prefetched k-tiles: tile kt + 1 waits in one while tile kt + 2 is in flight into the other, so a
    // tile's loads have two iterations to land (with one stage the 3-term-split products - ~100 matrix-pipe cycles per
    // k-tile - ran at one global-load latency per k-tile)
    float4 ra[2][LA], rb[2][LB];
    int va[2][LA], vb[2][LB];    // number of valid leading elements (0..4) of each prefetched float4
    // Loads are UNCONDITIONAL from clamped (always valid) addresses; the masking happens when the registers
    // are written to LDS, AFTER the MFMAs of the current tile.  (Predicated loads make hipcc wait vmcnt(0)
    // inside every exec-mask branch: 4 dependent L2/HBM round trips per k-tile instead of one in flight
    // under the matrix cores.)
    const int Mc = M > 0 ? M - 1 : 0, Nc = N - 1;
    auto nvalid = [](int first, int limit, bool ok) { return ok ? max(0, min(4, limit - first)) : 0; };
    auto gload = [&](int k0, auto stage) {
        constexpr int S = decltype(stage)::value;
#pragma unroll
        for (int p = 0; p < LA; ++p) {
            int idx = tid + p * 256;
            if (A_KC) {
                const int m = m0 + idx / (BK / 4), k = k0 + (idx % (BK / 4)) * 4;
                const int kc = min(k, Kfull - 4);
                ra[S][p] = *reinterpret_cast<const float4*>(A + (size_t)min(m, Mc) * a_rs + kc);
                va[S][p] = nvalid(k, K, m < M && kc == k);
            } else {
                const int k = k0 + idx / (BM / 4), m = m0 + (idx % (BM / 4)) * 4;
                const int mc = min(m, Mfull - 4);
                ra[S][p] = *reinterpret_cast<const float4*>(A + (size_t)min(k, Kfull - 1) * a_cs + mc);
                va[S][p] = nvalid(m, M, k < K && mc == m);
            }
        }
#pragma unroll
        for (int p = 0; p < LB; ++p) {
            int idx = tid + p * 256;
            if (B_KC) {
                const int n = n0 + idx / (BK / 4), k = k0 + (idx % (BK / 4)) * 4;
                const int kc = min(k, Kfull - 4);
                rb[S][p] = *reinterpret_cast<const float4*>(B + (size_t)min(n, Nc) * b_rs + kc);
                vb[S][p] = nvalid(k, K, n < N && kc == k);
            } else {
                const int k = k0 + idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
                const int nc = min(n, N - 4);
                rb[S][p] = *reinterpret_cast<const float4*>(B + (size_t)min(k, Kfull - 1) * b_cs + nc);
                vb[S][p] = nvalid(n, N, k < K && nc == n);
            }
        }
    };
    auto masked = [](float4 v, int nv) {
        v.x = nv > 0 ? v.x : 0.f; v.y = nv > 1 ? v.y : 0.f; v.z = nv > 2 ? v.z : 0.f; v.w = nv > 3 ? v.w : 0.f;
        return v;
    };
    auto lstore = [&](int buf, auto stage) {
        constexpr int S = decltype(stage)::value;
#pragma unroll
        for (int p = 0; p < LA; ++p) {
            int idx = tid + p * 256;
            const float4 v = masked(ra[S][p], va[S][p]);
            if (A_KC) {
                const int m = idx / PP, pc = idx % PP;
                *reinterpret_cast<float4*>(fa(buf) + m * BK + ((pc ^ swz(m)) << 2)) = v;
            } else {
                int k = idx / (BM / 4), m = (idx % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(&As[buf][k][m]) = v;
            }
        }
#pragma unroll
        for (int p = 0; p < LB; ++p) {
            int idx = tid + p * 256;
            const float4 v = masked(rb[S][p], vb[S][p]);
            if (B_KC) {
                const int n = idx / PP, pc = idx % PP;
                *reinterpret_cast<float4*>(fb(buf) + n * BK + ((pc ^ swz(n)) << 2)) = v;
            } else {
                int k = idx / (BN / 4), n = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(&Bs[buf][k][n]) = v;
            }
        }
    };

    K = kend;                                          // loads are bounded by this split's k-range
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    using St0 = std::integral_constant<int, 0>;
    using St1 = std::integral_constant<int, 1>;
    if (nk > 0) {
        gload(kbeg, St0{});
        lstore(0, St0{});
        if (nk > 1) gload(kbeg + BK, St1{});             // tile 1 -> stage 1, tile 2 -> stage 0 (free again)
        if (nk > 2) gload(kbeg + 2 * BK, St0{});
    }
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    auto ktile = [&](const int kt, auto nxt) {           // nxt: the register stage that holds tile kt + 1
        const int buf = kt & 1;
        // k-steps in groups of 8: MFMA e of group j takes k = 8 j + 4 half + e from BOTH operands (any pairing of the
        // tile's k values with (lane half, step) is a valid order of the sum) - a k-contiguous operand then needs ONE
        // ds_read_b128 per lane per 4 MFMAs.
        if constexpr (S3) {
            // two groups of 8 k-values = one 16-deep bf16 MFMA step: a lane's 4 + 4 values of an operand are its 8 k slots
            // (the same k <-> (lane half, slot) map on both operands), split into hi / lo in registers
#pragma unroll
            for (int j16 = 0; j16 < BK / 16; ++j16) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = wm * (BM / 2) + i * 32 + l31;
                    float v[8];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int j8 = 2 * j16 + g;
                        if (A_KC) {
                            const float4 q = *reinterpret_cast<const float4*>(fa(buf) + m * BK + (((2 * j8 + half) ^ swz(m)) << 2));
                            v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 * g + e] = As[buf][8 * j8 + 4 * half + e][m];
                        }
                    }
                    split8(v, ah[i], al[i]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = wn * (BN / 2) + j * 32 + l31;
                    float v[8];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int j8 = 2 * j16 + g;
                        if (B_KC) {
                            const float4 q = *reinterpret_cast<const float4*>(fb(buf) + n * BK + (((2 * j8 + half) ^ swz(n)) << 2));
                            v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 * g + e] = Bs[buf][8 * j8 + 4 * half + e][n];
                        }
                    }
                    split8(v, bh[j], bl[j]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
        for (int j8 = 0; j8 < BK / 8; ++j8) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = wm * (BM / 2) + i * 32 + l31;
                if (A_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(fa(buf) + m * BK + (((2 * j8 + half) ^ swz(m)) << 2));
                    a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][e] = As[buf][8 * j8 + 4 * half + e][m];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn * (BN / 2) + j * 32 + l31;
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(fb(buf) + n * BK + (((2 * j8 + half) ^ swz(n)) << 2));
                    b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[j][e] = Bs[buf][8 * j8 + 4 * half + e][n];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        }
        if (kt + 1 < nk) lstore(buf ^ 1, nxt);
        if (kt + 3 < nk) gload(kbeg + (kt + 3) * BK, nxt);
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(kt, St1{});
        if (kt + 1 < nk) ktile(kt + 1, St0{});
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) - ONE float of a row per lane.  Stored as
    // it comes that is 16 4-byte store (and, with beta, load) instructions per MFMA tile and wave, and a vector-memory
    // instruction costs a CU ~40 cycles whatever its width (csrc/gruf.hip, score_ce_bf16.hip): the tile goes through a per-wave
    // LDS patch (the operand tiles are free now) and leaves as 16-byte accesses, 4 per lane and MFMA tile.
    const bool vec = (N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 15) == 0);
    if (vec) {
        constexpr int PS = 36;                             // patch row stride (floats): 16-byte aligned rows, conflict-free
        float* patch = ((wave < 2) ? fa(0) : fb(0)) + (wave & 1) * (32 * PS);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PS + l31] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's patch writes have landed
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = q * 64 + lane, rl = idx >> 3, c4 = (idx & 7) * 4;
                    const int row = m0 + wm * (BM / 2) + i * 32 + rl, col = n0 + wn * (BN / 2) + j * 32 + c4;
                    if (row < Mfull && col < N) {
                        float4* p = reinterpret_cast<float4*>(C + (size_t)row * ldc + col);
                        if (row < M) {
                            float4 v = *reinterpret_cast<const float4*>(patch + rl * PS + c4);
                            v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
                            if (bias != nullptr) {
                                const float4 bq = *reinterpret_cast<const float4*>(bias + col);
                                v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                            }
                            if (beta != 0.f) {
                                const float4 o = *p;
                                v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
                            }
                            *p = v;
                        } else if (beta == 0.f) {
                            *p = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();           // (the next MFMA tile reuses the patch)
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + l31;
            const float bv = (bias != nullptr && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
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

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(
    const float* __restrict__ A, int a_rs, int a_cs, const float* __restrict__ B, int b_rs, int b_cs,
    float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N, int K,
    const int* __restrict__ dyn, int dyn_mode, float alpha, float beta, float* __restrict__ part) {
    constexpr int BK = TileK<BM>::value;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + 4];
    gemm_f32_tile<BM, BN, A_KC, B_KC>(A, a_rs, a_cs, B, b_rs, b_cs, C, ldc, bias, M, N, K, dyn, dyn_mode, alpha, beta, part,
                                      blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, As, Bs);
}

// ---- grouped launch: up to SREC_GEMM32_MAXP independent products (any mix of the three operand layouts) in ONE grid.
// The session-vector head of the models is a chain of small fp32 products (B x d x d; a few microseconds of matrix
// work each) next to one or two larger ones over all the nodes: launched one by one they cost a kernel node each plus a
// split-K reduce each; grouped, the small ones run in the shadow of the large ones.
struct GroupK {
    srec_gemm_f32_group g;
    int tile_end[SREC_GEMM32_MAXP];      // prefix sums of workgroups per problem
    int tiles_n[SREC_GEMM32_MAXP], tiles_mn[SREC_GEMM32_MAXP];
    long ws_off[SREC_GEMM32_MAXP];       // slab offsets (floats) of the split problems
    unsigned defer_mask;                 // bit p: the slab sum of problem p is left to the caller (srec_gemm_f32_group_run_defer)
};

template <bool S3>
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(GroupK k) {
    constexpr int BK = TileK<64>::value;
    __shared__ __attribute__((aligned(16))) float As[2][BK][64 + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][64 + 4];
    int p = 0;
    while (p + 1 < k.g.np && (int)blockIdx.x >= k.tile_end[p]) ++p;
    const int t = blockIdx.x - (p > 0 ? k.tile_end[p - 1] : 0);
    const int bz = t / k.tiles_mn[p], r = t % k.tiles_mn[p];
    const int by = r / k.tiles_n[p], bx = r % k.tiles_n[p];
    const srec_gemm_f32_group& g = k.g;
    const int nsplit = g.nsplit[p] > 1 ? g.nsplit[p] : 1;
    float* part = k.g.ws + k.ws_off[p];
#define SREC_TILE(AK, BK_)                                                                                             \
    gemm_f32_tile<64, 64, AK, BK_, S3>(g.A[p], g.a_rs[p], g.a_cs[p], g.B[p], g.b_rs[p], g.b_cs[p], g.C[p], g.ldc[p],         \
                                   g.bias[p], g.M[p], g.N[p], g.K[p], g.dyn[p], g.dyn_mode[p], g.alpha[p], g.beta[p], \
                                   part, bx, by, bz, nsplit, As, Bs)
    const bool akc = g.a_cs[p] == 1, bkc = g.b_cs[p] == 1;     // uniform per workgroup
    if (akc && bkc) SREC_TILE(true, true);
    else if (akc) SREC_TILE(true, false);
    else if (bkc) SREC_TILE(false, true);
    else SREC_TILE(false, false);
#undef SREC_TILE
}

// the split problems of a group, reduced in one launch: blockIdx.y = problem
__global__ void splitk_reduce_group_kernel(GroupK k) {
    const int p = blockIdx.y;
    const srec_gemm_f32_group& g = k.g;
    const int nsplit = g.nsplit[p];
    if (nsplit <= 1 || ((k.defer_mask >> p) & 1u)) return;
    const int M = g.M[p], N = g.N[p];
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (size_t)M * N) return;
    const float* part = g.ws + k.ws_off[p];
    const int row = (int)(i / N), col = (int)(i % N);
    const int Ml = g.dyn_mode[p] == 1 ? dyn_count(g.dyn[p], M) : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int z = 0; z < nsplit; ++z) {                       // (8 slab loads in flight: the launch is a latency chain)
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)z * M * N + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* c = g.C[p] + (size_t)row * g.ldc[p] + col;
    const float alpha = g.alpha[p], beta = g.beta[p];
    const float* bias = g.bias[p];
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

// C = alpha * sum_z part[z] + bias + beta * C   (rows >= live M are zeroed when beta == 0)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ C, int ldc,
                                     const float* __restrict__ bias, int M, int N, const int* __restrict__ dyn,
                                     int dyn_mode, float alpha, float beta) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (size_t)M * N) return;
    const int row = (int)(i / N), col = (int)(i % N);
    const int Ml = dyn_mode == 1 ? dyn_count(dyn, M) : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
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

template <int BM, int BN>
int launch(const float* A, int a_rs, int a_cs, const float* B, int b_rs, int b_cs, float* C, int ldc,
           const float* bias, int M, int N, int K, const int* dyn, int dyn_mode, float alpha, float beta,
           float* ws, int nsplit, hipStream_t st) {
    dim3 grid(cdiv(N, BN), cdiv(M, BM), nsplit);
    const bool akc = (a_cs == 1), bkc = (b_cs == 1);
#define SREC_GEMM_GO(AK, BK_)                                                                             \
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BK_>), grid, dim3(256), 0, st, A, a_rs, a_cs, B, b_rs, \
                       b_cs, C, ldc, bias, M, N, K, dyn, dyn_mode, alpha, beta, ws)
    if (akc && bkc) SREC_GEMM_GO(true, true);
    else if (akc && !bkc) SREC_GEMM_GO(true, false);
    else if (!akc && bkc) SREC_GEMM_GO(false, true);
    else SREC_GEMM_GO(false, false);
#undef SREC_GEMM_GO
    if (nsplit > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * N / 4 + 255) / 256)), dim3(256), 0, st, ws,
                           nsplit, C, ldc, bias, M, N, dyn, dyn_mode, alpha, beta);
    SREC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int srec_gemm_f32(const float* A, int a_rs, int a_cs, const float* B, int b_rs, int b_cs, float* C,
                             int ldc, const float* bias, int M, int N, int K, const int* dyn, int dyn_mode,
                             float alpha, float beta, float* ws, long ws_floats, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if ((a_rs != 1 && a_cs != 1) || (b_rs != 1 && b_cs != 1)) return SREC_BAD_ARG;
    // float4 paths: the non-unit stride and the contiguous extent must be multiples of 4 floats
    const int a_ld = (a_cs == 1) ? a_rs : a_cs, b_ld = (b_cs == 1) ? b_rs : b_cs;
    if ((a_ld & 3) || (b_ld & 3)) return SREC_BAD_ARG;
    if ((a_cs == 1 && (K & 3)) || (a_cs != 1 && (M & 3))) return SREC_BAD_ARG;
    if ((b_cs == 1 && (K & 3)) || (b_cs != 1 && (N & 3))) return SREC_BAD_ARG;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // small problems: 64x64 tiles keep more CUs busy; large: 128x128 for operand reuse
    const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (tiles128 >= 192)
        return launch<128, 128>(A, a_rs, a_cs, B, b_rs, b_cs, C, ldc, bias, M, N, K, dyn, dyn_mode, alpha, beta,
                                nullptr, 1, st);
    // skinny problems (few output tiles, long K - every weight-gradient GEMM): split K across
    // workgroups so all 256 CUs work; partial slabs in ws, reduced deterministically.
    const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
    int nsplit = 1;
    if (ws != nullptr && tiles64 < 256 && K >= 256 && (N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0) {
        nsplit = (int)(512 / tiles64);
        if (nsplit > K / 64) nsplit = K / 64;
        if (nsplit > 32) nsplit = 32;
        while (nsplit > 1 && (long)nsplit * M * N > ws_floats) --nsplit;
        if (nsplit < 1) nsplit = 1;
    }
    return launch<64, 64>(A, a_rs, a_cs, B, b_rs, b_cs, C, ldc, bias, M, N, K, dyn, dyn_mode, alpha, beta, ws, nsplit,
                          st);
}

// Grouped form of srec_gemm_f32 (same operand conventions per problem).  The split of skinny long-K problems is chosen
// here (g->nsplit is ignored on input); ws / ws_floats = the shared slab workspace.
// (split-K sums inside the launch - "the last workgroup to arrive at a tile adds the slabs" - were measured at +130 us per
// step: the device-scope fences write back / invalidate whole L2s, profiles/r03_notes.md; the reduce stays a launch of its own)
static int gemm_f32_group_go(const void* desc, float* ws, long ws_floats, long* slab_off, int* slab_n, void* stream);

extern "C" int srec_gemm_f32_group_run(const void* desc, float* ws, long ws_floats, void* stream) {
    return gemm_f32_group_go(desc, ws, ws_floats, nullptr, nullptr, stream);
}

// the same, but the slab sums of the PLAIN split problems (alpha = 1, beta = 0, no bias, no row clamp, contiguous C: the weight
// gradients of a backward group) are left to the caller: slab_off[p] (floats into ws; -1: problem p is finished) and slab_n[p]
// slabs of M N floats each, to be summed into C[p] (srec_sum_slabs_multi - e.g. ONE launch at the end of the backward pass for
// every group of it).  ws must then stay untouched until that sum has run: a private buffer of the call, not a shared one.
extern "C" int srec_gemm_f32_group_run_defer(const void* desc, float* ws, long ws_floats, long* slab_off, int* slab_n,
                                             void* stream) {
    if (slab_off == nullptr || slab_n == nullptr) return SREC_BAD_ARG;
    return gemm_f32_group_go(desc, ws, ws_floats, slab_off, slab_n, stream);
}

static int gemm_f32_group_go(const void* desc, float* ws, long ws_floats, long* slab_off, int* slab_n, void* stream) {
    const srec_gemm_f32_group* gin = (const srec_gemm_f32_group*)desc;
    if (gin->np <= 0) return 0;
    if (gin->np > SREC_GEMM32_MAXP) return SREC_BAD_ARG;
    GroupK k;
    k.g = *gin;
    k.g.ws = ws;
    k.defer_mask = 0u;
    int order[SREC_GEMM32_MAXP];
    {   // Workgroups start in index order: the problems with the longest per-workgroup k-loop go first, so that an unsplit
        // long-K product is not left running alone after the large products have drained.
        for (int p = 0; p < gin->np; ++p) order[p] = p;
        for (int i = 1; i < gin->np; ++i)
            for (int j = i; j > 0 && gin->K[order[j]] > gin->K[order[j - 1]]; --j) std::swap(order[j], order[j - 1]);
        for (int q = 0; q < gin->np; ++q) {
            const int p = order[q];
            srec_gemm_f32_group& g = k.g;
            g.A[q] = gin->A[p]; g.a_rs[q] = gin->a_rs[p]; g.a_cs[q] = gin->a_cs[p];
            g.B[q] = gin->B[p]; g.b_rs[q] = gin->b_rs[p]; g.b_cs[q] = gin->b_cs[p];
            g.C[q] = gin->C[p]; g.ldc[q] = gin->ldc[p]; g.bias[q] = gin->bias[p];
            g.M[q] = gin->M[p]; g.N[q] = gin->N[p]; g.K[q] = gin->K[p];
            g.dyn[q] = gin->dyn[p]; g.dyn_mode[q] = gin->dyn_mode[p];
            g.alpha[q] = gin->alpha[p]; g.beta[q] = gin->beta[p];
        }
    }
    long total64 = 0;
    for (int p = 0; p < k.g.np; ++p) {
        const srec_gemm_f32_group& g = k.g;
        if (g.M[p] <= 0 || g.N[p] <= 0) return SREC_BAD_ARG;
        if ((g.a_rs[p] != 1 && g.a_cs[p] != 1) || (g.b_rs[p] != 1 && g.b_cs[p] != 1)) return SREC_BAD_ARG;
        const int a_ld = (g.a_cs[p] == 1) ? g.a_rs[p] : g.a_cs[p], b_ld = (g.b_cs[p] == 1) ? g.b_rs[p] : g.b_cs[p];
        if ((a_ld & 3) || (b_ld & 3)) return SREC_BAD_ARG;
        if ((g.a_cs[p] == 1 && (g.K[p] & 3)) || (g.a_cs[p] != 1 && (g.M[p] & 3))) return SREC_BAD_ARG;
        if ((g.b_cs[p] == 1 && (g.K[p] & 3)) || (g.b_cs[p] != 1 && (g.N[p] & 3))) return SREC_BAD_ARG;
        if (((uintptr_t)g.A[p] & 15) || ((uintptr_t)g.B[p] & 15)) return SREC_BAD_ARG;
        total64 += (long)cdiv(g.M[p], 64) * cdiv(g.N[p], 64);
    }
    long ws_used = 0, ws_plan = 0;
    int end = 0, max_red = 0, any_split = 0;
    for (int p = 0; p < k.g.np; ++p) {
        srec_gemm_f32_group& g = k.g;
        const int tm = cdiv(g.M[p], 64), tn = cdiv(g.N[p], 64);
        const long tiles = (long)tm * tn;
        int nsplit = 1;
        // problems with few output tiles get a k-split (>= 64 of K per workgroup): always for long K (weight gradients over
        // all the nodes), for short K only when the group does not fill the chip anyway (a small product beside a large one
        // runs in its shadow and saves the reduce)
        if (ws != nullptr && tiles < 128 && (g.K[p] >= 1024 || (g.K[p] >= 256 && total64 < 512)) && (g.N[p] & 3) == 0 && (g.ldc[p] & 3) == 0 &&
            ((uintptr_t)g.C[p] & 15) == 0) {
            nsplit = (int)(512 / tiles);
            if (nsplit > g.K[p] / 64) nsplit = g.K[p] / 64;
            if (nsplit > 32) nsplit = 32;
            while (nsplit > 1 && ws_plan + (long)nsplit * g.M[p] * g.N[p] > ws_floats) --nsplit;
            if (nsplit < 1) nsplit = 1;
        } else if (ws != nullptr && tiles < 128 && g.K[p] >= 512 && (g.N[p] & 3) == 0 && (g.ldc[p] & 3) == 0 &&
                   ((uintptr_t)g.C[p] & 15) == 0 && ws_plan + (long)(g.K[p] / 256) * g.M[p] * g.N[p] <= ws_floats) {
            // a full chip: still cut a medium k-loop down to the 8 k-tiles of the large products beside it - the launch is one
            // round of workgroups and lasts as long as its longest k-loop (the head's K = 512 weight gradients: 16 k-tiles)
            nsplit = g.K[p] / 256;
        }
        g.nsplit[p] = nsplit;
        if (nsplit > 1) ws_plan += (long)nsplit * g.M[p] * g.N[p];     // (trimming below only shrinks it)
        k.tiles_n[p] = tn;
        k.tiles_mn[p] = (int)tiles;
        end += (int)tiles * nsplit;
    }
    {   // 34 KB of LDS -> 4 workgroups per CU = 128 slots per XCD, 1024 on the chip; workgroups are latency bound, so a grid a
        // little over a multiple of that pays a whole extra round (the head's backward group: 1172 workgroups, 40 us): trim
        // the k-splits until the grid fits the rounds it almost fits
        const int slots = 1024, rounds = end / slots;
        if (rounds >= 1 && end % slots != 0 && end % slots < slots * 35 / 100) {
            bool moved = true;
            while (end > rounds * slots && moved) {
                moved = false;
                int best = -1;
                for (int p = 0; p < k.g.np; ++p)
                    if (k.g.nsplit[p] > 2 && (best < 0 || k.g.nsplit[p] * k.tiles_mn[p] > k.g.nsplit[best] * k.tiles_mn[best])) best = p;
                if (best >= 0) { --k.g.nsplit[best]; end -= k.tiles_mn[best]; moved = true; }
            }
        }
    }
    end = 0;
    for (int p = 0; p < k.g.np; ++p) {
        srec_gemm_f32_group& g = k.g;
        const int nsplit = g.nsplit[p];
        k.ws_off[p] = ws_used;
        bool may = false;                   // slab_n[] on entry: non-zero = the caller may sum this problem's slabs later
        if (slab_off != nullptr) { may = slab_n[order[p]] != 0; slab_off[order[p]] = -1; slab_n[order[p]] = 0; }
        if (nsplit > 1) {
            ws_used += (long)nsplit * g.M[p] * g.N[p];
            if (may && g.alpha[p] == 1.f && g.beta[p] == 0.f && g.bias[p] == nullptr && g.dyn_mode[p] != 1) {
                k.defer_mask |= 1u << p;
                slab_off[order[p]] = k.ws_off[p];
                slab_n[order[p]] = nsplit;
            } else {
                if (slab_off != nullptr) slab_n[order[p]] = nsplit;          // (slab_off < 0: informational)
                any_split = 1;
                const int red = (int)(((size_t)g.M[p] * g.N[p] / 4 + 63) / 64);  // one-wave workgroups: 4 x the CUs at work
                if (red > max_red) max_red = red;
            }
        }
        end += k.tiles_mn[p] * nsplit;
        k.tile_end[p] = end;
    }
    hipStream_t st = (hipStream_t)stream;
    if (gin->split3) hipLaunchKernelGGL(gemm_f32_group_kernel<true>, dim3(end), dim3(256), 0, st, k);
    else hipLaunchKernelGGL(gemm_f32_group_kernel<false>, dim3(end), dim3(256), 0, st, k);
    if (any_split)
        hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(max_red, k.g.np), dim3(64), 0, st, k);
    SREC_LAUNCH_CHECK();
    return 0;
}
