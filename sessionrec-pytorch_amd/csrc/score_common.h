// Reduction kernels shared by the fp32 (score_ce.hip) and bf16 (score_ce_bf16.hip) fused scoring paths.
#pragma once
#include "common.h"

namespace {

// lse_b from the per-item-tile partials; per-session loss term
__global__ void ce_reduce_stats_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                       const float* __restrict__ lab_logit, int ntiles, int B,
                                       const int* __restrict__ dynB, float* __restrict__ lse,
                                       float* __restrict__ lossvec) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int Bd = dyn_count(dynB, B);
    if (b >= Bd) {
        if (lane == 0) { lse[b] = 0.f; lossvec[b] = 0.f; }
        return;
    }
    float m = -INFINITY;
    for (int t = lane; t < ntiles; t += 64) m = fmaxf(m, part_m[(size_t)t * B + b]);
    m = wave_max(m);
    float l = 0.f;
    for (int t = lane; t < ntiles; t += 64) {
        const float pm = part_m[(size_t)t * B + b];
        if (pm != -INFINITY) l += part_l[(size_t)t * B + b] * expf(pm - m);
    }
    l = wave_sum(l);
    if (lane == 0) {
        const float v = m + logf(l);
        lse[b] = v;
        lossvec[b] = v - lab_logit[b];
    }
}

__global__ void ce_mean_kernel(const float* __restrict__ lossvec, int B, const int* __restrict__ dynB,
                               float* __restrict__ loss) {
    __shared__ float red[16];
    const int Bd = dyn_count(dynB, B);
    float s = 0.f;
    for (int i = threadIdx.x; i < Bd; i += blockDim.x) s += lossvec[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        loss[0] = t / (float)(Bd > 0 ? Bd : 1);
    }
}

// out = sum_r part[r]  (fixed order: deterministic).  64 float4 columns per block, the R slabs split over the 4 waves.
__global__ void dsr_reduce_kernel(const float* __restrict__ part, int R, size_t n, float* __restrict__ out) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const int per = (R + 3) / 4, r0 = g * per, r1 = min(R, r0 + per);
        for (int r = r0; r < r1; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)r * n + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && i < n) {
#pragma unroll
        for (int k = 1; k < 4; ++k) { s.x += red[k][lane].x; s.y += red[k][lane].y; s.z += red[k][lane].z; s.w += red[k][lane].w; }
        *reinterpret_cast<float4*>(out + i) = s;
    }
}

}  // namespace
