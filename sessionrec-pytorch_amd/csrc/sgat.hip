// SGAT shortcut-graph attention (LESSR, lessr.py:68-74) after the q/k/v GEMMs:
//   e_uv = fc_e(sigmoid(q_u + k_v)); a = softmax over the in-edges of v; out_v = sum_u a_uv v_u
// (DGL u_add_v + edge_softmax + u_mul_e_sum).  One wavefront per destination node, the
// backward as two gather-style passes (per destination, per source): deterministic, no atomics.
#include "common.h"

namespace {

constexpr int WPB = 4;
constexpr int MAXDEG = SREC_MAX_DEGREE_SGAT;

__global__ void sgat_fwd_kernel(const float* __restrict__ Q, int ld_q, const float* __restrict__ K, int ld_k,
                                const float* __restrict__ we, const float* __restrict__ Vf, int ld_v,
                                const int* __restrict__ in_ptr, const int* __restrict__ in_idx,
                                const int* __restrict__ esrc, int n_cap, const int* __restrict__ dyn, int Hh, int Do,
                                float* __restrict__ A, float* __restrict__ out, int ld_o) {
    __shared__ float sc[WPB][MAXDEG];
    __shared__ int su[WPB][MAXDEG];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * WPB + w;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? min(in_ptr[v + 1] - beg, MAXDEG) : 0;
    for (int j = lane; j < deg; j += 64) su[w][j] = esrc[in_idx[beg + j]];
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < deg; ++j) {
        float s = 0.f;
        for (int h = lane; h < Hh; h += 64)
            s += we[h] * sigmoidf_(Q[(size_t)su[w][j] * ld_q + h] + K[(size_t)v * ld_k + h]);
        s = wave_sum(s);
        if (lane == 0) sc[w][j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int j = lane; j < deg; j += 64) m = fmaxf(m, sc[w][j]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < deg; j += 64) z += expf(sc[w][j] - m);
    z = wave_sum(z);
    const float iz = deg > 0 ? 1.f / z : 0.f;
    for (int j = lane; j < deg; j += 64) {
        const float a = expf(sc[w][j] - m) * iz;
        sc[w][j] = a;
        A[in_idx[beg + j]] = a;
    }
    __builtin_amdgcn_wave_barrier();
    for (int c = lane * 4; c < Do; c += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < deg; ++j) {
            const float a = sc[w][j];
            const float4 f = *reinterpret_cast<const float4*>(Vf + (size_t)su[w][j] * ld_v + c);
            o.x += a * f.x; o.y += a * f.y; o.z += a * f.z; o.w += a * f.w;
        }
        *reinterpret_cast<float4*>(out + (size_t)v * ld_o + c) = o;
    }
}

__global__ void sgat_bwd_dst_kernel(const float* __restrict__ dout, int ld_o, const float* __restrict__ Q, int ld_q,
                                    const float* __restrict__ K, int ld_k, const float* __restrict__ we,
                                    const float* __restrict__ Vf, int ld_v, const float* __restrict__ A,
                                    const int* __restrict__ in_ptr, const int* __restrict__ in_idx,
                                    const int* __restrict__ esrc, int n_cap, const int* __restrict__ dyn, int Hh,
                                    int Do, float* __restrict__ dQe, float* __restrict__ dK, int ld_dk,
                                    float* __restrict__ dwe_part, int ld_dw) {
    __shared__ float ds[WPB][MAXDEG];
    __shared__ int su[WPB][MAXDEG];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * WPB + w;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? min(in_ptr[v + 1] - beg, MAXDEG) : 0;
    for (int j = lane; j < deg; j += 64) su[w][j] = esrc[in_idx[beg + j]];
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < deg; ++j) {
        float s = 0.f;
        for (int c = lane * 4; c < Do; c += 256) {
            const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)v * ld_o + c);
            const float4 f = *reinterpret_cast<const float4*>(Vf + (size_t)su[w][j] * ld_v + c);
            s += g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
        }
        s = wave_sum(s);
        if (lane == 0) ds[w][j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    float t = 0.f;
    for (int j = lane; j < deg; j += 64) t += A[in_idx[beg + j]] * ds[w][j];
    t = wave_sum(t);
    for (int j = lane; j < deg; j += 64) ds[w][j] = A[in_idx[beg + j]] * (ds[w][j] - t);
    __builtin_amdgcn_wave_barrier();
    for (int h = lane; h < Hh; h += 64) {
        const float kv = live ? K[(size_t)v * ld_k + h] : 0.f, wh = we[h];
        float dk = 0.f, dw = 0.f;
        for (int j = 0; j < deg; ++j) {
            const float sg = sigmoidf_(Q[(size_t)su[w][j] * ld_q + h] + kv);
            const float d = ds[w][j];
            dw += d * sg;
            const float dp = d * wh * sg * (1.f - sg);
            dQe[(size_t)in_idx[beg + j] * Hh + h] = dp;
            dk += dp;
        }
        dK[(size_t)v * ld_dk + h] = dk;
        dwe_part[(size_t)v * ld_dw + h] = dw;
    }
}

__global__ void sgat_bwd_src_kernel(const float* __restrict__ dout, int ld_o, const float* __restrict__ A,
                                    const float* __restrict__ dQe, const int* __restrict__ out_ptr,
                                    const int* __restrict__ out_idx, const int* __restrict__ edst, int n_cap,
                                    const int* __restrict__ dyn, int Hh, int Do, float* __restrict__ dQ, int ld_dq,
                                    float* __restrict__ dVf, int ld_dv) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u = blockIdx.x * WPB + w;
    if (u >= n_cap) return;
    const bool live = u < dyn_count(dyn, n_cap);
    const int beg = live ? out_ptr[u] : 0;
    const int deg = live ? out_ptr[u + 1] - beg : 0;
    for (int h = lane; h < Hh; h += 64) {
        float s = 0.f;
        for (int j = 0; j < deg; ++j) s += dQe[(size_t)out_idx[beg + j] * Hh + h];
        dQ[(size_t)u * ld_dq + h] = s;
    }
    for (int c = lane * 4; c < Do; c += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < deg; ++j) {
            const int e = out_idx[beg + j];
            const float a = A[e];
            const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)edst[e] * ld_o + c);
            o.x += a * g.x; o.y += a * g.y; o.z += a * g.z; o.w += a * g.w;
        }
        *reinterpret_cast<float4*>(dVf + (size_t)u * ld_dv + c) = o;
    }
}

}  // namespace

extern "C" int srec_sgat_fwd(const float* Q, int ld_q, const float* K, int ld_k, const float* we, const float* Vf,
                             int ld_v, const int* in_ptr, const int* in_idx, const int* esrc, int n_cap,
                             const int* dyn, int Hh, int Do, float* A, float* out, int ld_o, void* stream) {
    if (n_cap <= 0) return 0;
    if ((Do & 3) || (ld_v & 3) || (ld_o & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(sgat_fwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, Q, ld_q, K, ld_k, we, Vf,
                       ld_v, in_ptr, in_idx, esrc, n_cap, dyn, Hh, Do, A, out, ld_o);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_sgat_bwd_dst(const float* dout, int ld_o, const float* Q, int ld_q, const float* K, int ld_k,
                                 const float* we, const float* Vf, int ld_v, const float* A, const int* in_ptr,
                                 const int* in_idx, const int* esrc, int n_cap, const int* dyn, int Hh, int Do,
                                 float* dQe, float* dK, int ld_dk, float* dwe_part, int ld_dw, void* stream) {
    if (n_cap <= 0) return 0;
    if ((Do & 3) || (ld_v & 3) || (ld_o & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(sgat_bwd_dst_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, dout, ld_o, Q, ld_q,
                       K, ld_k, we, Vf, ld_v, A, in_ptr, in_idx, esrc, n_cap, dyn, Hh, Do, dQe, dK, ld_dk, dwe_part,
                       ld_dw);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_sgat_bwd_src(const float* dout, int ld_o, const float* A, const float* dQe, const int* out_ptr,
                                 const int* out_idx, const int* edst, int n_cap, const int* dyn, int Hh, int Do,
                                 float* dQ, int ld_dq, float* dVf, int ld_dv, void* stream) {
    if (n_cap <= 0) return 0;
    if ((Do & 3) || (ld_o & 3) || (ld_dv & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(sgat_bwd_src_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, dout, ld_o, A, dQe,
                       out_ptr, out_idx, edst, n_cap, dyn, Hh, Do, dQ, ld_dq, dVf, ld_dv);
    SREC_LAUNCH_CHECK();
    return 0;
}
