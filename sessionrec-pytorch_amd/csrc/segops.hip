// Per-session (segment) kernels: one 64-lane wavefront owns one session's nodes.
// Session graphs are tiny (<= 256 nodes), so attention scores live in LDS and all
// softmax / weighted-sum reductions are wavefront reductions - no atomics, no DGL
// segment kernels, deterministic.
//
//   srec_seg_attn_fwd/bwd     AttnReadout core after the fc_u / fc_v GEMMs:
//       e_i = fc_e(sigmoid(u_i + v_b)); alpha = softmax_session(e); out_b = sum_i alpha_i x_i
//       (srgnn.py:79-86, niser.py:77-84, lessr.py:106-113, msgifsr.py:139-146)
//   srec_seg_mean_add_fwd/bwd out_i = h_i + mean_{j in session(i)} f_j   (msgifsr.py:86-89)
#include "common.h"

namespace {

constexpr int WPB = 4;
constexpr int MAXN = SREC_MAX_SESSION_NODES;   // max nodes of one session (host checks: srec_limits)

// One 256-thread workgroup per session: the 4 waves split the session's nodes for the per-node dot products
// (float4 per lane along the hidden dim), the soft-max is one wave, the weighted sums run one column per thread.
__global__ void seg_attn_fwd_kernel(const float* __restrict__ U, int ld_u, const float* __restrict__ Vq, int ld_v,
                                    const float* __restrict__ we, const float* __restrict__ X, int ld_x,
                                    const int* __restrict__ seg, int B, const int* __restrict__ dynB, int h, int D,
                                    float* __restrict__ alpha, float* __restrict__ out, int ld_out) {
    __shared__ float e[MAXN];
    const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const bool live = b < dyn_count(dynB, B);
    const int base = live ? seg[b] : 0;
    const int n = live ? min(seg[b + 1] - base, MAXN) : 0;
    if (h <= 256) {
        // the usual case (hidden size <= 256: one float4 per lane): the rows of this wave's nodes are fetched four at a time
        const int k = lane * 4;
        const bool kok = k < h;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), wk = v;
        if (kok) {
            v = *reinterpret_cast<const float4*>(Vq + (size_t)b * ld_v + k);
            wk = *reinterpret_cast<const float4*>(we + k);
        }
        for (int i0 = w; i0 < n; i0 += 4 * WPB) {
            float4 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(i0 + j * WPB, n - 1);
                u[j] = kok ? *reinterpret_cast<const float4*>(U + (size_t)(base + i) * ld_u + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * WPB;
                float acc = kok ? wk.x * sigmoidf_(u[j].x + v.x) + wk.y * sigmoidf_(u[j].y + v.y) + wk.z * sigmoidf_(u[j].z + v.z) +
                                      wk.w * sigmoidf_(u[j].w + v.w) : 0.f;
                acc = wave_sum(acc);
                if (lane == 0 && i < n) e[i] = acc;
            }
        }
    } else {
        for (int i = w; i < n; i += WPB) {
            float acc = 0.f;
            for (int k = lane * 4; k < h; k += 256) {
                const float4 u = *reinterpret_cast<const float4*>(U + (size_t)(base + i) * ld_u + k);
                const float4 v = *reinterpret_cast<const float4*>(Vq + (size_t)b * ld_v + k);
                const float4 wk = *reinterpret_cast<const float4*>(we + k);
                acc += wk.x * sigmoidf_(u.x + v.x) + wk.y * sigmoidf_(u.y + v.y) + wk.z * sigmoidf_(u.z + v.z) +
                       wk.w * sigmoidf_(u.w + v.w);
            }
            acc = wave_sum(acc);
            if (lane == 0) e[i] = acc;
        }
    }
    __syncthreads();
    if (w == 0) {
        float m = -INFINITY;
        for (int i = lane; i < n; i += 64) m = fmaxf(m, e[i]);
        m = wave_max(m);
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += expf(e[i] - m);
        s = wave_sum(s);
        const float inv = n > 0 ? 1.f / s : 0.f;
        for (int i = lane; i < n; i += 64) {
            const float a = expf(e[i] - m) * inv;
            e[i] = a;
            alpha[base + i] = a;
        }
    }
    __syncthreads();
    for (int c = tid; c < D; c += 256) {
        float o = 0.f;
        const float* xp = X + (size_t)base * ld_x + c;
        int i = 0;
        for (; i + 8 <= n; i += 8) {                     // eight node rows in flight, added in node order (an in-order wave
            float xv[8];                                 // otherwise pays one memory round trip per node)
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = xp[(size_t)(i + k) * ld_x];
#pragma unroll
            for (int k = 0; k < 8; ++k) o += e[i + k] * xv[k];
        }
        for (; i < n; ++i) o += e[i] * xp[(size_t)i * ld_x];
        out[(size_t)b * ld_out + c] = o;
    }
}

// grid = B + 1: workgroup b < live B owns session b; all workgroups share the zeroing of the rows of dX / dU behind
// the last live node (padded layouts), so the outputs need no host-side zero fill.
__global__ void seg_attn_bwd_kernel(const float* __restrict__ dout, int ld_do, const float* __restrict__ X, int ld_x,
                                    const float* __restrict__ alpha, const float* __restrict__ U, int ld_u,
                                    const float* __restrict__ Vq, int ld_v, const float* __restrict__ we,
                                    const int* __restrict__ seg, int B, const int* __restrict__ dynB, int h, int D,
                                    int n_cap, float* __restrict__ dX, int ld_dx, float* __restrict__ dU, int ld_du,
                                    float* __restrict__ dVq, int ld_dv, float* __restrict__ dwe_part, int ld_dw) {
    __shared__ float de[MAXN];
    __shared__ float al[MAXN];
    const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int Bd = dyn_count(dynB, B);
    // rows behind the last live node: every workgroup zeroes its share (row r0 + b, r0 + b + grid, ...)
    for (int r = seg[Bd] + b; r < n_cap; r += gridDim.x) {
        for (int c = tid; c < D; c += 256) dX[(size_t)r * ld_dx + c] = 0.f;
        for (int k = tid; k < h; k += 256) dU[(size_t)r * ld_du + k] = 0.f;
    }
    if (b >= Bd) {
        if (b < B)
            for (int k = tid; k < h; k += 256) {
                dVq[(size_t)b * ld_dv + k] = 0.f; dwe_part[(size_t)b * ld_dw + k] = 0.f;
            }
        return;
    }
    const int base = seg[b];
    const int n = min(seg[b + 1] - base, MAXN);
    // dalpha_i = <dout_b, x_i>;  dX_i = alpha_i dout_b
    for (int i = w; i < n; i += WPB) {
        const float a = alpha[base + i];
        float acc = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)b * ld_do + c);
            const float4 x = *reinterpret_cast<const float4*>(X + (size_t)(base + i) * ld_x + c);
            acc += g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w;
            *reinterpret_cast<float4*>(dX + (size_t)(base + i) * ld_dx + c) =
                make_float4(a * g.x, a * g.y, a * g.z, a * g.w);
        }
        acc = wave_sum(acc);
        if (lane == 0) { de[i] = acc; al[i] = a; }
    }
    __syncthreads();
    if (w == 0) {
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += al[i] * de[i];
        s = wave_sum(s);
        for (int i = lane; i < n; i += 64) de[i] = al[i] * (de[i] - s);     // d e_i
    }
    __syncthreads();
    for (int k = tid; k < h; k += 256) {
        const float vq = Vq[(size_t)b * ld_v + k];
        const float wk = we[k];
        float dv = 0.f, dw = 0.f;
        const float* up = U + (size_t)base * ld_u + k;
        int i = 0;
        for (; i + 8 <= n; i += 8) {                     // eight node rows in flight, accumulated in node order
            float uv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) uv[j] = up[(size_t)(i + j) * ld_u];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sg = sigmoidf_(uv[j] + vq);
                const float dei = de[i + j];
                dw += dei * sg;
                const float dp = dei * wk * sg * (1.f - sg);
                dU[(size_t)(base + i + j) * ld_du + k] = dp;
                dv += dp;
            }
        }
        for (; i < n; ++i) {
            const float sg = sigmoidf_(up[(size_t)i * ld_u] + vq);
            const float dei = de[i];
            dw += dei * sg;
            const float dp = dei * wk * sg * (1.f - sg);
            dU[(size_t)(base + i) * ld_du + k] = dp;
            dv += dp;
        }
        dVq[(size_t)b * ld_dv + k] = dv;
        dwe_part[(size_t)b * ld_dw + k] = dw;
    }
}

__global__ void seg_mean_add_fwd_kernel(const float* __restrict__ H, int ld_h, const float* __restrict__ F, int ld_f,
                                        const int* __restrict__ seg, int B, const int* __restrict__ dynB, int D,
                                        float* __restrict__ out, int ld_out) {
    const int b = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= dyn_count(dynB, B)) return;
    const int base = seg[b], n = seg[b + 1] - base;
    const float inv = 1.f / (float)(n > 0 ? n : 1);
    for (int c = lane * 4; c < D; c += 256) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < n; ++i) {
            const float4 f = *reinterpret_cast<const float4*>(F + (size_t)(base + i) * ld_f + c);
            m.x += f.x; m.y += f.y; m.z += f.z; m.w += f.w;
        }
        m.x *= inv; m.y *= inv; m.z *= inv; m.w *= inv;
        for (int i = 0; i < n; ++i) {
            const float4 hv = *reinterpret_cast<const float4*>(H + (size_t)(base + i) * ld_h + c);
            *reinterpret_cast<float4*>(out + (size_t)(base + i) * ld_out + c) =
                make_float4(hv.x + m.x, hv.y + m.y, hv.z + m.z, hv.w + m.w);
        }
    }
}

// dF_j = (1/n) sum_{i in session(j)} dout_i     (dH = dout is a pass-through)
__global__ void seg_mean_add_bwd_kernel(const float* __restrict__ dout, int ld_do, const int* __restrict__ seg, int B,
                                        const int* __restrict__ dynB, int D, float* __restrict__ dF, int ld_df) {
    const int b = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= dyn_count(dynB, B)) return;
    const int base = seg[b], n = seg[b + 1] - base;
    const float inv = 1.f / (float)(n > 0 ? n : 1);
    for (int c = lane * 4; c < D; c += 256) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < n; ++i) {
            const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)(base + i) * ld_do + c);
            m.x += g.x; m.y += g.y; m.z += g.z; m.w += g.w;
        }
        m.x *= inv; m.y *= inv; m.z *= inv; m.w *= inv;
        for (int i = 0; i < n; ++i) *reinterpret_cast<float4*>(dF + (size_t)(base + i) * ld_df + c) = m;
    }
}

// SRGNN weighted-mean aggregation (srgnn.py:21-29,36-41): coef[e] = w_e / sum_{e' into dst(e)} w_e'
__global__ void edge_coef_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, const int* __restrict__ ew,
                                 int n_cap, const int* __restrict__ dyn, float* __restrict__ coef) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= dyn_count(dyn, n_cap)) return;
    const int beg = ptr[v], end = ptr[v + 1];
    float s = 0.f;
    for (int j = beg; j < end; ++j) s += (float)ew[idx[j]];
    for (int j = beg; j < end; ++j) coef[idx[j]] = (float)ew[idx[j]] / s;
}

// out[v,:] = sum_{e in list(v)} coef[e] * X[other[e],:]   (zero rows for empty lists: DGL zero fill)
__global__ void edge_agg_kernel(const float* __restrict__ X, int ld_x, const int* __restrict__ ptr,
                                const int* __restrict__ idx, const int* __restrict__ other,
                                const float* __restrict__ coef, int n_cap, const int* __restrict__ dyn, int D,
                                float* __restrict__ out, int ld_o) {
    const int v = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    const int beg = live ? ptr[v] : 0, end = live ? ptr[v + 1] : 0;
    for (int c = lane * 4; c < D; c += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = beg; j < end; ++j) {
            const int e = idx[j];
            const float a = coef[e];
            const float4 x = *reinterpret_cast<const float4*>(X + (size_t)other[e] * ld_x + c);
            o.x += a * x.x; o.y += a * x.y; o.z += a * x.z; o.w += a * x.w;
        }
        *reinterpret_cast<float4*>(out + (size_t)v * ld_o + c) = o;
    }
}

}  // namespace

extern "C" int srec_edge_coef(const int* ptr, const int* idx, const int* ew, int n_cap, const int* dyn, float* coef,
                              void* stream) {
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(edge_coef_kernel, dim3(cdiv(n_cap, 256)), dim3(256), 0, (hipStream_t)stream, ptr, idx, ew, n_cap,
                       dyn, coef);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_edge_agg(const float* X, int ld_x, const int* ptr, const int* idx, const int* other,
                             const float* coef, int n_cap, const int* dyn, int D, float* out, int ld_o, void* stream) {
    if (n_cap <= 0) return 0;
    if ((D & 3) || (ld_x & 3) || (ld_o & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(edge_agg_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, X, ld_x, ptr, idx, other,
                       coef, n_cap, dyn, D, out, ld_o);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_seg_attn_fwd(const float* U, int ld_u, const float* Vq, int ld_v, const float* we, const float* X,
                                 int ld_x, const int* seg, int B, const int* dynB, int h, int D, float* alpha,
                                 float* out, int ld_out, void* stream) {
    if (B <= 0) return 0;
    if ((D & 3) || (ld_x & 3) || (ld_out & 3)) return SREC_BAD_ARG;
    if ((h & 3) || (ld_u & 3) || (ld_v & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(seg_attn_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, U, ld_u, Vq, ld_v, we, X, ld_x, seg,
                       B, dynB, h, D, alpha, out, ld_out);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_seg_attn_bwd(const float* dout, int ld_do, const float* X, int ld_x, const float* alpha,
                                 const float* U, int ld_u, const float* Vq, int ld_v, const float* we, const int* seg,
                                 int B, const int* dynB, int h, int D, int n_cap, float* dX, int ld_dx, float* dU, int ld_du,
                                 float* dVq, int ld_dv, float* dwe_part, int ld_dw, void* stream) {
    if (B <= 0) return 0;
    if ((D & 3) || (ld_x & 3) || (ld_do & 3) || (ld_dx & 3) || dU == nullptr) return SREC_BAD_ARG;
    hipLaunchKernelGGL(seg_attn_bwd_kernel, dim3(B + 1), dim3(256), 0, (hipStream_t)stream, dout, ld_do, X, ld_x, alpha, U,
                       ld_u, Vq, ld_v, we, seg, B, dynB, h, D, n_cap, dX, ld_dx, dU, ld_du, dVq, ld_dv, dwe_part, ld_dw);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_seg_mean_add_fwd(const float* H, int ld_h, const float* F, int ld_f, const int* seg, int B,
                                     const int* dynB, int D, float* out, int ld_out, void* stream) {
    if (B <= 0) return 0;
    if ((D & 3) || (ld_h & 3) || (ld_f & 3) || (ld_out & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(seg_mean_add_fwd_kernel, dim3(cdiv(B, WPB)), dim3(256), 0, (hipStream_t)stream, H, ld_h, F, ld_f,
                       seg, B, dynB, D, out, ld_out);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_seg_mean_add_bwd(const float* dout, int ld_do, const int* seg, int B, const int* dynB, int D,
                                     float* dF, int ld_df, void* stream) {
    if (B <= 0) return 0;
    if ((D & 3) || (ld_do & 3) || (ld_df & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(seg_mean_add_bwd_kernel, dim3(cdiv(B, WPB)), dim3(256), 0, (hipStream_t)stream, dout, ld_do, seg,
                       B, dynB, D, dF, ld_df);
    SREC_LAUNCH_CHECK();
    return 0;
}

// the static budgets above, for the host-side checks (collate.py / ops.check_limits)
extern "C" int srec_limits(int* max_session_nodes, int* max_degree, int* max_degree_sgat) {
    if (max_session_nodes != nullptr) *max_session_nodes = SREC_MAX_SESSION_NODES;
    if (max_degree != nullptr) *max_degree = SREC_MAX_DEGREE;
    if (max_degree_sgat != nullptr) *max_degree_sgat = SREC_MAX_DEGREE_SGAT;
    return 0;
}
