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
// D <= 32 (the reference's default width): W_hh lives in REGISTERS - forward: lane o owns rows o and o + 64 of W_hh
// (2 x 32 floats), backward: lane (c, half) owns 48 entries of column c - so a time step is 64 / 48 FMAs per lane on a
// broadcast LDS read of h / dg instead of a D-long chain of L2 loads, and the next edge's records are requested while
// the current step computes.
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

// ---- D <= 32 ---------------------------------------------------------------------------------------------------
constexpr int SD = 32;

__global__ __launch_bounds__(256) void gru_seq_fwd_small_kernel(
    const float* __restrict__ GI, int ld_gi, const float* __restrict__ Whh, const float* __restrict__ bhh,
    const int* __restrict__ in_ptr, const int* __restrict__ in_idx, const int* __restrict__ esrc, int n_cap,
    const int* __restrict__ dyn, int D, float* __restrict__ neigh, int ld_n, float* __restrict__ gates,
    float* __restrict__ Hprev, float* __restrict__ ghn) {
    __shared__ __attribute__((aligned(16))) float hs[WPB][SD];
    __shared__ float ghs[WPB][3 * SD];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * WPB + w;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? in_ptr[v + 1] - beg : 0;
    const int D3 = 3 * D;
    if (deg == 0) {                                       // DGL zero-fill
        if (lane < D) neigh[(size_t)v * ld_n + lane] = 0.f;
        return;
    }
    // this lane's rows of W_hh (zero beyond D / 3D): outputs o0 = lane, o1 = lane + 64
    float w0[SD], w1[SD];
    const int o0 = lane, o1 = lane + 64;
    if (D == SD) {                                        // whole 128-byte rows: 16-byte loads
#pragma unroll
        for (int k4 = 0; k4 < SD / 4; ++k4) {
            const float4 a = *reinterpret_cast<const float4*>(Whh + (size_t)o0 * SD + 4 * k4);
            const float4 b = o1 < D3 ? *reinterpret_cast<const float4*>(Whh + (size_t)o1 * SD + 4 * k4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            w0[4 * k4] = a.x; w0[4 * k4 + 1] = a.y; w0[4 * k4 + 2] = a.z; w0[4 * k4 + 3] = a.w;
            w1[4 * k4] = b.x; w1[4 * k4 + 1] = b.y; w1[4 * k4 + 2] = b.z; w1[4 * k4 + 3] = b.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SD; ++k) {
            w0[k] = (o0 < D3 && k < D) ? Whh[(size_t)o0 * D + k] : 0.f;
            w1[k] = (o1 < D3 && k < D) ? Whh[(size_t)o1 * D + k] : 0.f;
        }
    }
    const float b0 = o0 < D3 ? bhh[o0] : 0.f, b1 = o1 < D3 ? bhh[o1] : 0.f;
    if (lane < SD) hs[w][lane] = 0.f;
    const int c = lane;                                   // gate column of lanes 0 .. D-1
    int e = in_idx[beg];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (c < D) {
        const float* gi = GI + (size_t)esrc[e] * ld_gi;
        g0 = gi[c]; g1 = gi[D + c]; g2 = gi[2 * D + c];
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < deg; ++j) {
        // the next step's edge and input projections, requested before this step's arithmetic
        int en = e;
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (j + 1 < deg) {
            en = in_idx[beg + j + 1];
            if (c < D) {
                const float* gi = GI + (size_t)esrc[en] * ld_gi;
                n0 = gi[c]; n1 = gi[D + c]; n2 = gi[2 * D + c];
            }
        }
        float s0 = b0, s1 = b1;
#pragma unroll
        for (int k4 = 0; k4 < SD / 4; ++k4) {
            const float4 h4 = *reinterpret_cast<const float4*>(&hs[w][4 * k4]);
            s0 += w0[4 * k4] * h4.x; s1 += w1[4 * k4] * h4.x;
            s0 += w0[4 * k4 + 1] * h4.y; s1 += w1[4 * k4 + 1] * h4.y;
            s0 += w0[4 * k4 + 2] * h4.z; s1 += w1[4 * k4 + 2] * h4.z;
            s0 += w0[4 * k4 + 3] * h4.w; s1 += w1[4 * k4 + 3] * h4.w;
        }
        if (o0 < D3) ghs[w][o0] = s0;
        if (o1 < D3) ghs[w][o1] = s1;
        __builtin_amdgcn_wave_barrier();
        if (c < D) {
            const float hp = hs[w][c];
            const float gn = ghs[w][2 * D + c];
            const float r = sigmoidf_(g0 + ghs[w][c]);
            const float z = sigmoidf_(g1 + ghs[w][D + c]);
            const float n = tanhf(g2 + r * gn);
            gates[(size_t)e * D3 + c] = r;
            gates[(size_t)e * D3 + D + c] = z;
            gates[(size_t)e * D3 + 2 * D + c] = n;
            Hprev[(size_t)e * D + c] = hp;
            ghn[(size_t)e * D + c] = gn;
            hs[w][c] = (1.f - z) * n + z * hp;
        }
        __builtin_amdgcn_wave_barrier();
        e = en; g0 = n0; g1 = n1; g2 = n2;
    }
    if (c < D) neigh[(size_t)v * ld_n + c] = hs[w][c];
}

__global__ __launch_bounds__(256) void gru_seq_bwd_small_kernel(
    const float* __restrict__ dneigh, int ld_dn, const float* __restrict__ Whh, const float* __restrict__ gates,
    const float* __restrict__ Hprev, const float* __restrict__ ghn, const int* __restrict__ in_ptr,
    const int* __restrict__ in_idx, int n_cap, const int* __restrict__ dyn, int D, float* __restrict__ dGIe,
    float* __restrict__ dGHe) {
    constexpr int HO = 3 * SD / 2;                        // 48 outputs per half-wave
    __shared__ __attribute__((aligned(16))) float dgs[WPB][3 * SD];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * WPB + w;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? in_ptr[v + 1] - beg : 0;
    if (deg == 0) return;
    const int D3 = 3 * D;
    const int c = lane & 31, half = lane >> 5;
    // column c of W_hh, outputs half * 48 .. + 47 in the PADDED numbering o' = 32 gate + column (zero beyond D)
    float wc[HO];
#pragma unroll
    for (int i = 0; i < HO; ++i) {
        const int op = half * HO + i, gate = op >> 5, col = op & 31;
        wc[i] = (c < D && col < D && deg > 1) ? Whh[(size_t)(gate * D + col) * D + c] : 0.f;
    }
    float dh = (half == 0 && c < D) ? dneigh[(size_t)v * ld_dn + c] : 0.f;      // lanes 0 .. D-1 carry dh
    const bool act = half == 0 && c < D;
    int e = in_idx[beg + deg - 1];
    float r = 0.f, z = 0.f, n = 0.f, hp = 0.f, gn = 0.f;
    if (act) {
        r = gates[(size_t)e * D3 + c]; z = gates[(size_t)e * D3 + D + c]; n = gates[(size_t)e * D3 + 2 * D + c];
        hp = Hprev[(size_t)e * D + c]; gn = ghn[(size_t)e * D + c];
    }
    if (lane < 3 * SD - 64) dgs[w][64 + lane] = 0.f;       // padded slots stay zero
    dgs[w][lane] = 0.f;
    __builtin_amdgcn_wave_barrier();
    for (int j = deg - 1; j >= 0; --j) {
        int en = e;
        float rn = 0.f, zn = 0.f, nn = 0.f, hpn = 0.f, gnn = 0.f;
        if (j > 0) {
            en = in_idx[beg + j - 1];
            if (act) {
                rn = gates[(size_t)en * D3 + c]; zn = gates[(size_t)en * D3 + D + c]; nn = gates[(size_t)en * D3 + 2 * D + c];
                hpn = Hprev[(size_t)en * D + c]; gnn = ghn[(size_t)en * D + c];
            }
        }
        if (act) {
            const float dpn = dh * (1.f - z) * (1.f - n * n);
            const float dpr = dpn * gn * r * (1.f - r);
            const float dpz = dh * (hp - n) * z * (1.f - z);
            dGIe[(size_t)e * D3 + c] = dpr; dGIe[(size_t)e * D3 + D + c] = dpz; dGIe[(size_t)e * D3 + 2 * D + c] = dpn;
            dGHe[(size_t)e * D3 + c] = dpr; dGHe[(size_t)e * D3 + D + c] = dpz; dGHe[(size_t)e * D3 + 2 * D + c] = dpn * r;
            dgs[w][c] = dpr; dgs[w][SD + c] = dpz; dgs[w][2 * SD + c] = dpn * r;
            dh = dh * z;                                  // direct path; the W_hh^T term is added below
        }
        __builtin_amdgcn_wave_barrier();
        if (j > 0) {                                      // h_prev of step 0 is the constant 0
            float s = 0.f;
#pragma unroll
            for (int i4 = 0; i4 < HO / 4; ++i4) {
                const float4 d4 = *reinterpret_cast<const float4*>(&dgs[w][half * HO + 4 * i4]);
                s += wc[4 * i4] * d4.x; s += wc[4 * i4 + 1] * d4.y; s += wc[4 * i4 + 2] * d4.z; s += wc[4 * i4 + 3] * d4.w;
            }
            s += __shfl_xor(s, 32, 64);
            if (act) dh += s;
        }
        __builtin_amdgcn_wave_barrier();
        e = en; r = rn; z = zn; n = nn; hp = hpn; gn = gnn;
    }
}

}  // namespace

// GI [Nsrc,3D]; Whh [3D,D] row-major (weight_hh as stored); WhhT [D,3D] k-major copy of it, read only when D > 32
// (nullable otherwise); per-edge outputs indexed by edge id.
extern "C" int srec_gru_seq_fwd(const float* GI, int ld_gi, const float* Whh, const float* WhhT, const float* bhh,
                                const int* in_ptr, const int* in_idx, const int* esrc, int n_cap, const int* dyn, int D,
                                float* neigh, int ld_n, float* gates, float* Hprev, float* ghn, void* stream) {
    if (n_cap <= 0) return 0;
    if (D <= 0 || D > MAXD) return SREC_BAD_ARG;
    if (D <= SD) {
        hipLaunchKernelGGL(gru_seq_fwd_small_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, GI, ld_gi, Whh,
                           bhh, in_ptr, in_idx, esrc, n_cap, dyn, D, neigh, ld_n, gates, Hprev, ghn);
        SREC_LAUNCH_CHECK();
        return 0;
    }
    if (WhhT == nullptr) return SREC_BAD_ARG;
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
    if (D <= SD) {
        hipLaunchKernelGGL(gru_seq_bwd_small_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, dneigh, ld_dn,
                           Whh, gates, Hprev, ghn, in_ptr, in_idx, n_cap, dyn, D, dGIe, dGHe);
        SREC_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, dneigh, ld_dn, Whh,
                       gates, Hprev, ghn, in_ptr, in_idx, n_cap, dyn, D, dGIe, dGHe);
    SREC_LAUNCH_CHECK();
    return 0;
}
