// The "weights of this step" copies as device functions: the hi / lo fragment-major operand copies of the read-out head
// (srec_head_wfrag, csrc/headf.hip) and the bf16 + transposed bf16 copies of the GEMM weights (srec_weights_bf16,
// csrc/gemm16.hip).  Shared with srec_step_weights (csrc/grufb.hip), which runs them and the GRU fragment copies in ONE launch:
// each is a once-per-optimizer-step pass over a few small matrices and a kernel node of the captured step otherwise.
#pragma once
#include "common.h"

namespace srec_frag {

constexpr int HEAD_NW = 4;               // waves per workgroup of the head kernels (wave w owns columns [w d/4, (w+1) d/4))

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = srec_pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = srec_pack_bf16(a - ah, b - bh);
}

// item idx (< N K / 8) of the hi / lo fragment-major copy of an operand matrix M [N, K] (trans = 0: M = W [rows = N, cols = K] as
// stored; trans = 1: M = W^T of the stored W [rows = K, cols = N]): fragment (((w KS + s) 2 + t) JB + j), JB = N / 128,
// KS = K / 16, holds for lane l the 8 bf16 of t(M[w N/4 + 32 j + (l & 31)][16 s + 8 (l >> 5) .. + 7]), t = hi / lo
__device__ __forceinline__ void head_frag_item(const float* __restrict__ W, unsigned short* __restrict__ dst, int rows, int cols,
                                               int trans, int idx) {
    const int N = trans ? cols : rows, K = trans ? rows : cols;
    const int JB = N / (32 * HEAD_NW), KS = K / 16;
    if (idx >= N * K / 8) return;
    const int lane = idx & 63, f = idx >> 6;
    const int j = f % JB, ws = f / JB, s = ws % KS, w = ws / KS;
    const int nrow = w * 32 * JB + 32 * j + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
    float v[8];
    if (!trans) {
        const float4 v0 = *reinterpret_cast<const float4*>(W + (size_t)nrow * K + kk), v1 = *reinterpret_cast<const float4*>(W + (size_t)nrow * K + kk + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = W[(size_t)(kk + i) * N + nrow];
    }
    uint4 h, l;
    split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y); split2(v[4], v[5], h.z, l.z); split2(v[6], v[7], h.w, l.w);
    unsigned short* o = dst + ((((size_t)(w * KS + s) * 2) * JB + j) * 64 + lane) * 8;
    *reinterpret_cast<uint4*>(o) = h;
    *reinterpret_cast<uint4*>(o + (size_t)JB * 512) = l;
}

// tile b (64 x 64, row-major tile order) of W [R, Cc] fp32 -> W16 [R, Cc] bf16 and (WT16 nullable) WT16 [Cc, R] bf16, through the
// workgroup's LDS tile: the operand copies of the bf16 GEMMs (srec_weights_bf16, csrc/gemm16.hip).  256 threads.
__device__ __forceinline__ void weights_bf16_tile(const float* __restrict__ W, unsigned short* __restrict__ W16,
                                                  unsigned short* __restrict__ WT16, int R, int Cc, int b,
                                                  unsigned short (*tile)[68]) {
    const int tc = (Cc + 63) / 64;
    const int r0 = (b / tc) * 64, c0 = (b % tc) * 64;
    if (((R | Cc) & 3) == 0) {
        // 4 columns per thread: float4 in, 8-byte bf16 stores in both layouts (2-byte stores ran at a third of this rate)
        const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
        for (int rr = y; rr < 64; rr += 16) {
            const int r = r0 + rr, c = c0 + 4 * x;
            uint2 v = make_uint2(0u, 0u);
            if (r < R && c < Cc) {
                const float4 f = *reinterpret_cast<const float4*>(W + (size_t)r * Cc + c);
                v = make_uint2(srec_pack_bf16(f.x, f.y), srec_pack_bf16(f.z, f.w));
                *reinterpret_cast<uint2*>(W16 + (size_t)r * Cc + c) = v;
            }
            *reinterpret_cast<uint2*>(&tile[rr][4 * x]) = v;
        }
        if (WT16 == nullptr) return;
        __syncthreads();
        for (int cc = y; cc < 64; cc += 16) {
            const int c = c0 + cc, r = r0 + 4 * x;
            if (c < Cc && r < R) {
                const unsigned lo = tile[4 * x][cc] | ((unsigned)tile[4 * x + 1][cc] << 16);
                const unsigned hi = tile[4 * x + 2][cc] | ((unsigned)tile[4 * x + 3][cc] << 16);
                *reinterpret_cast<uint2*>(WT16 + (size_t)c * R + r) = make_uint2(lo, hi);
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
            W16[(size_t)r * Cc + c] = v;
        }
        tile[rr][x] = v;
    }
    __syncthreads();
    if (WT16 != nullptr)
        for (int cc = y; cc < 64; cc += 4) {
            const int c = c0 + cc, r = r0 + x;
            if (c < Cc && r < R) WT16[(size_t)c * R + r] = tile[x][cc];
        }
}

}  // namespace srec_frag
