// EOPA neighbour aggregation (LESSR, lessr.py:20-27,35): for every node, a GRU runs over the
// features of its in-neighbours in edge-id (= click time) order, h0 = 0, and the last hidden state
// is the node's aggregated neighbourhood (zero for zero in-degree; DGL zero-fill).
//
// The input projection GI = ft W_ih^T + b_ih is ONE matrix-core GEMM over all nodes (a source
// feeds many edges); this kernel runs only the sequential part.  One 64-lane wavefront owns
// one destination node: h and the hidden projection live in LDS, W_hh is streamed from L2
// with consecutive lanes on consecutive addresses (k-major copy for the forward mat-vec, the
// native row-major layout for the transposed mat-vec of the backward).  Replaces DGL's
// degree-bucketed cuDNN GRU launches with a single launch, deterministic, no atomics.
//
// Per-edge records (indexed by edge id) are saved for BPTT: gates (r,z,n), h_prev, gh_n.
// Backward emits per-edge dGI / dGH; weight gradients are then MFMA GEMMs over the E records.
#include "common.h"

namespace {

constexpr int WPB = 4;
constexpr int MAXD = 256;

__global__ void gru_seq_fwd_kernel(const float* __restrict__ GI, int ld_gi, const float* __restrict__ WhhT,
                                   const float* __restrict__ bhh, const int* __restrict__ in_ptr,
                                   const int* __restrict__ in_idx, const int* __restrict__ esrc, int n_cap,
                                   const int* __restrict__ dyn, int D, float* __restrict__ neigh, int ld_n,
                                   float* __restrict__ gates, float* __restrict__ Hprev, float* __restrict__ ghn) {
    __shared__ float hs[WPB][MAXD];
    __shared__ float ghs[WPB][3 * MAXD];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * WPB + w;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? in_ptr[v + 1] - beg : 0;
    for (int c = lane; c < D; c += 64) hs[w][c] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int D3 = 3 * D;
    for (int j = 0; j < deg; ++j) {
        const int e = in_idx[beg + j];
        const int u = esrc[e];
        for (int o = lane; o < D3; o += 64) {
            float s = bhh[o];
            for (int k = 0; k < D; ++k) s += WhhT[(size_t)k * D3 + o] * hs[w][k];
            ghs[w][o] = s;
        }
        __builtin_amdgcn_wave_barrier();
        const float* gi = GI + (size_t)u * ld_gi;
        for (int c = lane; c < D; c += 64) {
            const float hp = hs[w][c];
            const float gn = ghs[w][2 * D + c];
            const float r = sigmoidf_(gi[c] + ghs[w][c]);
            const float z = sigmoidf_(gi[D + c] + ghs[w][D + c]);
            const float n = tanhf(gi[2 * D + c] + r * gn);
            gates[(size_t)e * D3 + c] = r;
            gates[(size_t)e * D3 + D + c] = z;
            gates[(size_t)e * D3 + 2 * D + c] = n;
            Hprev[(size_t)e * D + c] = hp;
            ghn[(size_t)e * D + c] = gn;
            hs[w][c] = (1.f - z) * n + z * hp;
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int c = lane; c < D; c += 64) neigh[(size_t)v * ld_n + c] = hs[w][c];
}

__global__ void gru_seq_bwd_kernel(const float* __restrict__ dneigh, int ld_dn, const float* __restrict__ Whh,
                                   const float* __restrict__ gates, const float* __restrict__ Hprev,
                                   const float* __restrict__ ghn, const int* __restrict__ in_ptr,
                                   const int* __restrict__ in_idx, int n_cap, const int* __restrict__ dyn, int D,
                                   float* __restrict__ dGIe, float* __restrict__ dGHe) {
    __shared__ float dhs[WPB][MAXD];
    __shared__ float dgs[WPB][3 * MAXD];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * WPB + w;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? in_ptr[v + 1] - beg : 0;
    const int D3 = 3 * D;
    for (int c = lane; c < D; c += 64) dhs[w][c] = live ? dneigh[(size_t)v * ld_dn + c] : 0.f;
    __builtin_amdgcn_wave_barrier();
    for (int j = deg - 1; j >= 0; --j) {
        const int e = in_idx[beg + j];
        for (int c = lane; c < D; c += 64) {
            const float r = gates[(size_t)e * D3 + c], z = gates[(size_t)e * D3 + D + c],
                        n = gates[(size_t)e * D3 + 2 * D + c];
            const float hp = Hprev[(size_t)e * D + c], gn = ghn[(size_t)e * D + c];
            const float dh = dhs[w][c];
            const float dpn = dh * (1.f - z) * (1.f - n * n);
            const float dpr = dpn * gn * r * (1.f - r);
            const float dpz = dh * (hp - n) * z * (1.f - z);
            dGIe[(size_t)e * D3 + c] = dpr; dGIe[(size_t)e * D3 + D + c] = dpz; dGIe[(size_t)e * D3 + 2 * D + c] = dpn;
            dGHe[(size_t)e * D3 + c] = dpr; dGHe[(size_t)e * D3 + D + c] = dpz; dGHe[(size_t)e * D3 + 2 * D + c] = dpn * r;
            dgs[w][c] = dpr; dgs[w][D + c] = dpz; dgs[w][2 * D + c] = dpn * r;
            dhs[w][c] = dh * z;                       // direct path; the W_hh^T term is added below
        }
        __builtin_amdgcn_wave_barrier();
        if (j > 0) {                                  // h_prev of step 0 is the constant 0
            for (int c = lane; c < D; c += 64) {
                float s = 0.f;
                for (int o = 0; o < D3; ++o) s += Whh[(size_t)o * D + c] * dgs[w][o];
                dhs[w][c] += s;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// GI [Nsrc,3D]; WhhT [D,3D] (k-major copy of weight_hh); per-edge outputs indexed by edge id.
extern "C" int srec_gru_seq_fwd(const float* GI, int ld_gi, const float* WhhT, const float* bhh, const int* in_ptr,
                                const int* in_idx, const int* esrc, int n_cap, const int* dyn, int D, float* neigh,
                                int ld_n, float* gates, float* Hprev, float* ghn, void* stream) {
    if (n_cap <= 0) return 0;
    if (D <= 0 || D > MAXD) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, GI, ld_gi, WhhT, bhh,
                       in_ptr, in_idx, esrc, n_cap, dyn, D, neigh, ld_n, gates, Hprev, ghn);
    SREC_LAUNCH_CHECK();
    return 0;
}

// Whh [3D,D] row-major (weight_hh as stored).  Outputs dGIe, dGHe [E,3D] (every live edge written).
extern "C" int srec_gru_seq_bwd(const float* dneigh, int ld_dn, const float* Whh, const float* gates,
                                const float* Hprev, const float* ghn, const int* in_ptr, const int* in_idx, int n_cap,
                                const int* dyn, int D, float* dGIe, float* dGHe, void* stream) {
    if (n_cap <= 0) return 0;
    if (D <= 0 || D > MAXD) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, dneigh, ld_dn, Whh,
                       gates, Hprev, ghn, in_ptr, in_idx, n_cap, dyn, D, dGIe, dGHe);
    SREC_LAUNCH_CHECK();
    return 0;
}
