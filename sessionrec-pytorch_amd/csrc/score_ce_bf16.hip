// bf16-operand variant of the fused full-catalog scoring + softmax-CE ("flash-CE", see score_ce.hip for the
// algorithm and the reference call sites it replaces).  BASELINE config C3's reduced-precision path.
//
// Differences to the exact-fp32 kernel, all driven by the gfx950 hardware model:
//   * v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 16x the fp32-MFMA rate, so the kernel is bound by staging and
//     LDS traffic, not by the matrix pipe;
//   * operands are pre-rounded ONCE per step into zero-padded bf16 copies (srec_bf16_prepare): row-major
//     [rows, d] for the S = X Y^T product and TRANSPOSED [d, rows] for ACC += P Y, whose B operand needs 8
//     consecutive reduction indices per lane - no transposes inside the hot loop, no bounds masks in the staging;
//   * the owner tile X lives in REGISTERS as ready-made MFMA fragments (64 VGPRs at d=256), LDS holds only
//     the streamed chunk twice (Ys, YsT) and P: 78 KB -> TWO workgroups per CU, whose staging / soft-max
//     phases overlap each other's MFMA phases (the fp32 kernel needs 148 KB and runs one per CU);
//   * every LDS fragment read is a 16-B ds_read_b128; row strides 528 B / 144 B put the 16 lanes of a b128 group
//     on 16 distinct 4-bank slots.
// Soft-max statistics, exp, the one-hot subtraction and all accumulation stay fp32; dE / d sr are written fp32.
#include "common.h"
#include "score_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { BMODE_FWD = 0, BMODE_DE = 1, BMODE_DSR = 2 };

__device__ __forceinline__ unsigned short f2bf(float a) {
    unsigned u = __float_as_uint(a);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// dst16[r, c] = bf16(src[r, c]),  dstT16[c, r] = bf16(src[r, c]);  rows >= live R and padding are zero.
__global__ void bf16_prepare_kernel(const float* __restrict__ src, int ld, int R, const int* __restrict__ dynR, int d,
                                    unsigned short* __restrict__ dst16, unsigned short* __restrict__ dstT16, int Rp) {
    __shared__ unsigned short tile[64][66];
    const int Rl = dyn_count(dynR, R);
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
    for (int rr = tr; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tc;
        unsigned short v = 0;
        if (r < Rl && c < d) v = f2bf(src[(size_t)r * ld + c]);
        tile[rr][tc] = v;
        if (r < Rp && c < d) dst16[(size_t)r * d + c] = v;
    }
    __syncthreads();
    for (int cc = tr; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tc;
        if (c < d && r < Rp) dstT16[(size_t)c * Rp + r] = tile[tc][cc];
    }
}

struct BArgs {
    const unsigned short* X16;    // owner rows   [Xp, d]
    const unsigned short* Y16;    // streamed rows [Yp, d]
    const unsigned short* YT16;   // streamed rows transposed [d, Yp]
    int Yp;
    const float* cs; const int* labels; const float* lse; const float* gscale; const float* ga; const float* gc;
    const int* dynB;
    int B, V, d;
    float* part_m; float* part_l; float* lab_logit;
    float* dE; int ld_de; int acc_dE;
    float* part_dsr;
    int chunks_per_range;
};

template <int NT, int MODE>
__global__ __launch_bounds__(256, 2) void flash_ce_bf16_kernel(BArgs a) {
    constexpr int D = NT * 32, KS = D / 16, LDY = D + 8, LDT = 72, LDP = 72;
    constexpr int NCB = (NT + 1) / 2;
    constexpr bool ITEMS_X = (MODE == BMODE_FWD || MODE == BMODE_DE);
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    unsigned short* Ys = smem16;                       // [64][LDY]
    unsigned short* YsT = Ys + 64 * LDY;               // [D][LDT]
    unsigned short* Ps = YsT + D * LDT;                // [64][LDP]  (or FWD scratch)
    float* scratch = reinterpret_cast<float*>(Ps);
    float* gaL = reinterpret_cast<float*>(Ps + 64 * LDP);   // [64] per-owner-row coefficients (DSR with ga/gc)
    float* gcL = gaL + 64;
    int* labL = reinterpret_cast<int*>(gcL + 64);          // [64] labels of the owner sessions (DSR)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int si = wave >> 1, sj = wave & 1;
    const int Bd = dyn_count(a.dynB, a.B);
    int nx, x0, ybeg, yend, tile_id;
    if (ITEMS_X) {
        nx = a.V; x0 = blockIdx.x * 64; ybeg = 0; yend = Bd; tile_id = blockIdx.x;
    } else {
        nx = Bd; x0 = blockIdx.y * 64; tile_id = 0;
        ybeg = blockIdx.x * a.chunks_per_range * 64;
        yend = min(a.V, ybeg + a.chunks_per_range * 64);
    }
    const bool x_empty = x0 >= nx;

    // owner tile as MFMA A-fragments: row si*32 + l31, k = ks*16 + 8*half .. +7   (padded buffer: no masks)
    bf16x8 xf[KS];
    {
        const unsigned short* xp = a.X16 + (size_t)(x0 + si * 32 + l31) * D + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xp + ks * 16);
    }
    float gs = 1.f;
    if (MODE != BMODE_FWD) gs = (a.gscale != nullptr ? *a.gscale : 1.f) / (float)(Bd > 0 ? Bd : 1);
    float xq[16];      // ITEMS_X: cs[item]   else: lse[session]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int xi = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (ITEMS_X) xq[r] = (a.cs != nullptr && xi < nx) ? a.cs[xi] : 1.f;
        else xq[r] = xi < nx ? a.lse[xi] : 0.f;
    }
    if (MODE == BMODE_DSR && tid < 64) labL[tid] = (x0 + tid < nx) ? a.labels[x0 + tid] : -1;
    const bool has_g = (MODE == BMODE_DSR) && a.ga != nullptr;     // per-session coefficients (rare: order fusion)
    if (has_g && tid < 64) {
        const int xi = x0 + tid;
        gaL[tid] = xi < nx ? a.ga[xi] : 0.f;
        gcL[tid] = xi < nx ? a.gc[xi] : 0.f;
    }
    f32x16 acc[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    constexpr int YQ = D / 8;                           // 16-B pieces per Ys row
    for (int y0 = ybeg; y0 < yend && !x_empty; y0 += 64) {
        // ---- stage the chunk: Ys[64][D] (rows y0..y0+63) and YsT[D][64] (columns y0..y0+63 of the transposed copy)
#pragma unroll 4
        for (int p = 0; p < NT; ++p) {
            const int idx = tid + p * 256;               // 64 * YQ = 256 * NT pieces
            const int row = idx / YQ, c8 = idx % YQ;
            const uint4 v = *reinterpret_cast<const uint4*>(a.Y16 + (size_t)(y0 + row) * D + c8 * 8);
            *reinterpret_cast<uint4*>(Ys + row * LDY + c8 * 8) = v;
        }
        if (MODE != BMODE_FWD) {
#pragma unroll 4
            for (int p = 0; p < NT; ++p) {
                const int idx = tid + p * 256;           // D * 8 = 256 * NT pieces
                const int row = idx >> 3, c8 = idx & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(a.YT16 + (size_t)row * a.Yp + y0 + c8 * 8);
                *reinterpret_cast<uint4*>(YsT + row * LDT + c8 * 8) = v;
            }
        }
        const int yj = y0 + sj * 32 + l31;
        const bool yvalid = yj < yend;
        float yq = 1.f, yga = gs, ygc = gs;
        int ylab = -1;
        if (MODE == BMODE_FWD) {
            ylab = yvalid ? a.labels[yj] : -1;
        } else if (MODE == BMODE_DE) {
            yq = yvalid ? a.lse[yj] : 0.f;
            ylab = yvalid ? a.labels[yj] : -1;
            if (a.ga != nullptr) { yga = yvalid ? a.ga[yj] : 0.f; ygc = yvalid ? a.gc[yj] : 0.f; }
        } else {
            yq = (a.cs != nullptr && yvalid) ? a.cs[yj] : 1.f;
        }
        __syncthreads();                                                  // (A)

        // ---- S = X Y^T (32x32 per wave), K = D in steps of 16
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            const unsigned short* yb = Ys + (sj * 32 + l31) * LDY + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(yb + ks * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], b, s, 0, 0, 0);
            }
        }
        if (MODE == BMODE_FWD) {
            float m = -INFINITY, z[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int item = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                z[r] = (item < nx) ? xq[r] * s[r] : -INFINITY;
                m = fmaxf(m, z[r]);
                if (yvalid && item == ylab) a.lab_logit[yj] = z[r];
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float ms = (m == -INFINITY) ? 0.f : m;
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) l += __expf(z[r] - ms);
            l += __shfl_xor(l, 32, 64);
            if (half == 0) {
                scratch[(si * 2 + 0) * 64 + sj * 32 + l31] = m;
                scratch[(si * 2 + 1) * 64 + sj * 32 + l31] = l;
            }
            __syncthreads();                                              // (B)
            if (tid < 64) {
                const int y = y0 + tid;
                if (y < yend) {
                    const float m0 = scratch[tid], l0 = scratch[64 + tid], m1 = scratch[128 + tid], l1 = scratch[192 + tid];
                    const float mm = fmaxf(m0, m1);
                    const float mms = (mm == -INFINITY) ? 0.f : mm;
                    a.part_m[(size_t)tile_id * a.B + y] = mm;
                    a.part_l[(size_t)tile_id * a.B + y] = l0 * __expf(m0 - mms) + l1 * __expf(m1 - mms);
                }
            }
            __syncthreads();                                              // (C)
        } else {
            // ---- P (bf16) -> LDS
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
                        const float ga_ = has_g ? gaL[xl] : gs, gc_ = has_g ? gcL[xl] : gs;
                        p = (__expf(zz - xq[r]) * ga_ - (labL[xl] == yj ? gc_ : 0.f)) * yq;
                    }
                }
                Ps[xl * LDP + sj * 32 + l31] = f2bf(p);
            }
            __syncthreads();                                              // (B)
            // ---- ACC += P Y : row block si, column blocks sj, sj+2, ...; K = 64 chunk rows in steps of 16
            {
                const unsigned short* pa = Ps + (si * 32 + l31) * LDP + 8 * half;
                const unsigned short* yt = YsT + (sj * 32 + l31) * LDT + 8 * half;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(pa + ks * 16);
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        if (sj + 2 * c < NT) {
                            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(yt + c * 64 * LDT + ks * 16);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();                                              // (C)
        }
    }

    if (MODE != BMODE_FWD) {
        const int d = a.d;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int cb = sj + 2 * c;
            if (cb >= NT) continue;
            const int col = cb * 32 + l31;
            if (col >= d) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xi = x0 + si * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (MODE == BMODE_DE) {
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
int launch_b(const BArgs& a, dim3 grid, hipStream_t st) {
    constexpr int D = NTV * 32;
    constexpr size_t lds = (size_t)(64 * (D + 8) + D * 72 + 64 * 72) * sizeof(unsigned short) + 192 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)flash_ce_bf16_kernel<NTV, MODE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((flash_ce_bf16_kernel<NTV, MODE>), grid, dim3(256), lds, st, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
int launch_bmode(const BArgs& a, dim3 grid, hipStream_t st) {
    switch (a.d / 32) {
        case 1: return launch_b<1, MODE>(a, grid, st);
        case 2: return launch_b<2, MODE>(a, grid, st);
        case 3: return launch_b<3, MODE>(a, grid, st);
        case 4: return launch_b<4, MODE>(a, grid, st);
        case 8: return launch_b<8, MODE>(a, grid, st);
        default: return SREC_BAD_ARG;
    }
}

int pick_ranges_b(int B, int V) {
    const int sess_tiles = cdiv(B, 64), chunks = cdiv(V, 64);
    int R = cdiv(1024, sess_tiles);
    if (R > chunks) R = chunks;
    return R < 1 ? 1 : R;
}

inline bool bad_d(int d) { return !(d == 32 || d == 64 || d == 96 || d == 128 || d == 256); }

}  // namespace

// rows [R, d] fp32 -> dst16 [Rp, d] and dstT16 [d, Rp] bf16 (RNE), zero for rows >= live R; Rp % 64 == 0.
extern "C" int srec_bf16_prepare(const float* src, int ld, int R, const int* dynR, int d, void* dst16, void* dstT16,
                                 int Rp, void* stream) {
    if (R <= 0 || (Rp & 63) || Rp < R) return SREC_BAD_ARG;
    hipLaunchKernelGGL(bf16_prepare_kernel, dim3(Rp / 64, cdiv(d, 64)), dim3(256), 0, (hipStream_t)stream, src, ld, R, dynR,
                       d, (unsigned short*)dst16, (unsigned short*)dstT16, Rp);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_ce_plan_bf16(int B, int V, int d, int* n_item_tiles, int* n_ranges) {
    if (bad_d(d)) return SREC_BAD_ARG;
    *n_item_tiles = cdiv(V, 64);
    *n_ranges = pick_ranges_b(B, V);
    return 0;
}

// sr16 [Bp,d], E16 [Vp,d] from srec_bf16_prepare.  Outputs as srec_score_ce_fwd.
extern "C" int srec_score_ce_fwd_bf16(const void* sr16, int Bp, const void* E16, int Vp, const float* cs,
                                      const int* labels, int B, int V, int d, const int* dynB, float* ws_stats,
                                      float* lab_logit, float* lse, float* lossvec, float* loss, void* stream) {
    if (bad_d(d) || (Bp & 63) || (Vp & 63)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nt = cdiv(V, 64);
    BArgs a{};
    a.X16 = (const unsigned short*)E16; a.Y16 = (const unsigned short*)sr16; a.YT16 = nullptr; a.Yp = Bp;
    a.cs = cs; a.labels = labels; a.dynB = dynB; a.B = B; a.V = V; a.d = d;
    a.part_m = ws_stats; a.part_l = ws_stats + (size_t)nt * B; a.lab_logit = lab_logit;
    int rc = launch_bmode<BMODE_FWD>(a, dim3(nt), st);
    if (rc) return rc;
    hipLaunchKernelGGL(ce_reduce_stats_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, a.part_m, a.part_l, lab_logit, nt, B,
                       dynB, lse, lossvec);
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, lossvec, B, dynB, loss);
    SREC_LAUNCH_CHECK();
    return 0;
}

// backward with the bf16 copies (row-major and transposed) of both operands; dE / dsr fp32 as in srec_score_ce_bwd.
extern "C" int srec_score_ce_bwd_bf16(const void* sr16, const void* srT16, int Bp, const void* E16, const void* ET16,
                                      int Vp, const float* cs, const int* labels, const float* lse,
                                      const float* gscale, const float* ga, const float* gc, int B, int V, int d,
                                      const int* dynB, float* dE, int ld_de, float* ws_dsr, float* dsr, int parts,
                                      void* stream) {
    if (bad_d(d) || (Bp & 63) || (Vp & 63)) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    BArgs a{};
    a.cs = cs; a.labels = labels; a.lse = lse; a.gscale = gscale; a.ga = ga; a.gc = gc; a.dynB = dynB;
    a.B = B; a.V = V; a.d = d; a.dE = dE; a.ld_de = ld_de; a.acc_dE = (parts & 4) ? 1 : 0; a.part_dsr = ws_dsr;
    int rc = 0;
    if (parts & 1) {
        a.X16 = (const unsigned short*)E16; a.Y16 = (const unsigned short*)sr16; a.YT16 = (const unsigned short*)srT16;
        a.Yp = Bp;
        rc = launch_bmode<BMODE_DE>(a, dim3(cdiv(V, 64)), st);
        if (rc) return rc;
    }
    if (!(parts & 2)) return 0;
    const int R = pick_ranges_b(B, V);
    a.chunks_per_range = cdiv(cdiv(V, 64), R);
    a.X16 = (const unsigned short*)sr16; a.Y16 = (const unsigned short*)E16; a.YT16 = (const unsigned short*)ET16;
    a.Yp = Vp;
    rc = launch_bmode<BMODE_DSR>(a, dim3(R, cdiv(B, 64)), st);
    if (rc) return rc;
    const size_t n = (size_t)B * d;
    hipLaunchKernelGGL(dsr_reduce_kernel, dim3((unsigned)cdiv((int)(n / 4), 256)), dim3(256), 0, st, ws_dsr, R, n, dsr);
    SREC_LAUNCH_CHECK();
    return 0;
}
