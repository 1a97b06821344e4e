// "Exact-fp32 work on the bf16 pipe": operand copies for the 3-term bf16 split products of the read-out / session-vector
// head (msgifsr.py:124-155 AttnReadout, :269-273 fc_sr; srgnn.py:76-91,142-144).
//
// On gfx950 the fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) runs at 1/16 of the bf16 rate.  A value x is written as
// hi = bf16(x), lo = bf16(x - hi): |x - hi - lo| <= 2^-18 |x| (two 8-bit significands), and
//     a b  =  a_hi b_hi + a_hi b_lo + a_lo b_hi  + O(2^-17 |a b|)
// so one exact-fp32 product becomes THREE bf16 products with fp32 accumulation - 3/16 of the fp32 MFMA time at ~2^-16
// relative accuracy (the single-bf16 rounding the head cannot afford is 2^-9: its output is the session vector that is
// scaled by 12 before the soft-max).  The three terms are the K-segments of ONE gemm16 problem (csrc/gemm16.hip:
// C = sum_s A_s B_s^T), so no new GEMM kernel is involved - only these operand copies.
#include "common.h"

namespace {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// hi / lo [n, ld16] (element offset col0 applied by the caller through the pointers) = split of src [n, d] (row stride ld);
// rows past the live count are written as zeros
__global__ void split_bf16_kernel(const float* __restrict__ src, int ld, int n, const int* __restrict__ dyn, int d,
                                  unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int ld16) {
    const int nl = dyn_count(dyn, n);
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long)n * d) return;
    const int r = (int)(i / d), c = (int)(i % d);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nl) v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
    const unsigned h0 = srec_pack_bf16(v.x, v.y), h1 = srec_pack_bf16(v.z, v.w);
    const float rx = v.x - bf2f((unsigned short)(h0 & 0xffffu)), ry = v.y - bf2f((unsigned short)(h0 >> 16));
    const float rz = v.z - bf2f((unsigned short)(h1 & 0xffffu)), rw = v.w - bf2f((unsigned short)(h1 >> 16));
    *reinterpret_cast<uint2*>(hi + (size_t)r * ld16 + c) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(lo + (size_t)r * ld16 + c) = make_uint2(srec_pack_bf16(rx, ry), srec_pack_bf16(rz, rw));
}

// up to 8 jobs of the kernel above in one launch (by-value descriptor)
struct SplitArgs {
    const float* src[8]; unsigned short* hi[8]; unsigned short* lo[8]; const int* dyn[8];
    int ld[8], n[8], d[8], ld16[8], start[9];
    int nj;
};
__global__ void split_bf16_multi_kernel(SplitArgs a) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.nj && (int)blockIdx.x >= a.start[i]) j = i;
    const int n = a.n[j], d = a.d[j];
    const int nl = dyn_count(a.dyn[j], n);
    const long i = ((long)(blockIdx.x - a.start[j]) * blockDim.x + threadIdx.x) * 4;
    if (i >= (long)n * d) return;
    const int r = (int)(i / d), c = (int)(i % d);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nl) v = *reinterpret_cast<const float4*>(a.src[j] + (size_t)r * a.ld[j] + c);
    const unsigned h0 = srec_pack_bf16(v.x, v.y), h1 = srec_pack_bf16(v.z, v.w);
    const float rx = v.x - bf2f((unsigned short)(h0 & 0xffffu)), ry = v.y - bf2f((unsigned short)(h0 >> 16));
    const float rz = v.z - bf2f((unsigned short)(h1 & 0xffffu)), rw = v.w - bf2f((unsigned short)(h1 >> 16));
    *reinterpret_cast<uint2*>(a.hi[j] + (size_t)r * a.ld16[j] + c) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(a.lo[j] + (size_t)r * a.ld16[j] + c) = make_uint2(srec_pack_bf16(rx, ry), srec_pack_bf16(rz, rw));
}

// weights: W [R, Cc] fp32 (contiguous) -> hi / lo [R, Cc] and the transposed hi / lo [Cc, R]; 64 x 64 tiles through LDS
struct WSplitArgs {
    const float* W[8];
    unsigned short* hi[8]; unsigned short* lo[8]; unsigned short* hiT[8]; unsigned short* loT[8];
    int R[8], Cc[8], start[9];
    int n;
};
__global__ void weights_split_kernel(WSplitArgs a) {
    __shared__ unsigned short th[64][68];
    __shared__ unsigned short tl[64][68];
    int t = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.n && (int)blockIdx.x >= a.start[i]) t = i;
    const int R = a.R[t], Cc = a.Cc[t];
    const int tc = (Cc + 63) / 64, b = blockIdx.x - a.start[t];
    const int r0 = (b / tc) * 64, c0 = (b % tc) * 64;
    const float* __restrict__ W = a.W[t];
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    for (int rr = y; rr < 64; rr += 16) {
        const int r = r0 + rr, c = c0 + 4 * x;
        uint2 vh = make_uint2(0u, 0u), vl = make_uint2(0u, 0u);
        if (r < R && c < Cc) {
            const float4 f = *reinterpret_cast<const float4*>(W + (size_t)r * Cc + c);
            vh = make_uint2(srec_pack_bf16(f.x, f.y), srec_pack_bf16(f.z, f.w));
            vl = make_uint2(srec_pack_bf16(f.x - bf2f((unsigned short)(vh.x & 0xffffu)), f.y - bf2f((unsigned short)(vh.x >> 16))),
                            srec_pack_bf16(f.z - bf2f((unsigned short)(vh.y & 0xffffu)), f.w - bf2f((unsigned short)(vh.y >> 16))));
            *reinterpret_cast<uint2*>(a.hi[t] + (size_t)r * Cc + c) = vh;
            *reinterpret_cast<uint2*>(a.lo[t] + (size_t)r * Cc + c) = vl;
        }
        *reinterpret_cast<uint2*>(&th[rr][4 * x]) = vh;
        *reinterpret_cast<uint2*>(&tl[rr][4 * x]) = vl;
    }
    if (a.hiT[t] == nullptr) return;
    __syncthreads();
    for (int cc = y; cc < 64; cc += 16) {
        const int c = c0 + cc, r = r0 + 4 * x;
        if (c < Cc && r < R) {
            *reinterpret_cast<uint2*>(a.hiT[t] + (size_t)c * R + r) =
                make_uint2(th[4 * x][cc] | ((unsigned)th[4 * x + 1][cc] << 16), th[4 * x + 2][cc] | ((unsigned)th[4 * x + 3][cc] << 16));
            *reinterpret_cast<uint2*>(a.loT[t] + (size_t)c * R + r) =
                make_uint2(tl[4 * x][cc] | ((unsigned)tl[4 * x + 1][cc] << 16), tl[4 * x + 2][cc] | ((unsigned)tl[4 * x + 3][cc] << 16));
        }
    }
}

}  // namespace

// nj <= 8 jobs: hi_j / lo_j [n_j, ld16_j] (bf16) = split of src_j [n_j, d_j] (fp32, row stride ld_j; rows past *dyn_j are
// zeros).  Every argument is a HOST array of nj entries; d, ld, ld16 multiples of 4, 8-byte aligned destinations.
extern "C" int srec_split_bf16(int nj, const void* src, const int* ld, const int* n, const void* dyn, const int* d, const void* hi,
                               const void* lo, const int* ld16, void* stream) {
    if (nj <= 0) return 0;
    if (nj > 8) return SREC_BAD_ARG;
    SplitArgs a{};
    a.nj = nj;
    int blocks = 0;
    for (int j = 0; j < nj; ++j) {
        a.src[j] = ((const float* const*)src)[j];
        a.hi[j] = ((unsigned short* const*)hi)[j];
        a.lo[j] = ((unsigned short* const*)lo)[j];
        a.dyn[j] = dyn != nullptr ? ((const int* const*)dyn)[j] : nullptr;
        a.ld[j] = ld[j]; a.n[j] = n[j]; a.d[j] = d[j]; a.ld16[j] = ld16[j];
        if (a.src[j] == nullptr || a.hi[j] == nullptr || a.lo[j] == nullptr || n[j] <= 0 || d[j] <= 0 || (d[j] & 3) || (ld[j] & 3) ||
            (ld16[j] & 3) || ((uintptr_t)a.hi[j] & 7) || ((uintptr_t)a.lo[j] & 7) || ((uintptr_t)a.src[j] & 15))
            return SREC_BAD_ARG;
        a.start[j] = blocks;
        blocks += (int)(((long)n[j] * d[j] / 4 + 255) / 256);
    }
    a.start[nj] = blocks;
    hipLaunchKernelGGL(split_bf16_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// n <= 8 weight matrices W_i [R_i, C_i] fp32 (contiguous, R_i and C_i multiples of 4) -> hi / lo bf16 copies [R_i, C_i] and
// their transposes hiT / loT [C_i, R_i] (hiT / loT arrays may hold NULLs: no transposed copy).  HOST arrays of n entries.
extern "C" int srec_weights_split_bf16(int n, const void* W, const int* R, const int* Cc, const void* hi, const void* lo,
                                       const void* hiT, const void* loT, void* stream) {
    if (n <= 0) return 0;
    if (n > 8 || W == nullptr || hi == nullptr || lo == nullptr) return SREC_BAD_ARG;
    WSplitArgs a{};
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i];
        a.hi[i] = ((unsigned short* const*)hi)[i];
        a.lo[i] = ((unsigned short* const*)lo)[i];
        a.hiT[i] = hiT != nullptr ? ((unsigned short* const*)hiT)[i] : nullptr;
        a.loT[i] = loT != nullptr ? ((unsigned short* const*)loT)[i] : nullptr;
        a.R[i] = R[i]; a.Cc[i] = Cc[i];
        if (a.W[i] == nullptr || a.hi[i] == nullptr || a.lo[i] == nullptr || R[i] <= 0 || Cc[i] <= 0 || ((R[i] | Cc[i]) & 3) ||
            ((a.hiT[i] == nullptr) != (a.loT[i] == nullptr)))
            return SREC_BAD_ARG;
        a.start[i] = blocks;
        blocks += cdiv(R[i], 64) * cdiv(Cc[i], 64);
    }
    a.start[n] = blocks;
    hipLaunchKernelGGL(weights_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}
