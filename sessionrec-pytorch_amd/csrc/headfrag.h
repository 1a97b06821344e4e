// hi / lo fragment-major operand copies of the read-out head's weights (csrc/headf.hip) as a device function: shared by
// srec_head_wfrag (headf.hip) and by srec_gru_wfrag_both (grufb.hip), which can take the head's copies along in its launch -
// both are "once per optimizer step" passes over a few small matrices, each a kernel node of the captured step otherwise.
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

}  // namespace srec_frag
