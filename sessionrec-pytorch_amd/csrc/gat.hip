// MSGIFSR graph message passing: multi-head GAT over one relation of the batched
// session heterograph, plus the MSHGNN head-max combine.
//
// Reference path replaced (gatconv.py:267-311 via msgifsr.py:58-64,74-89), per relation:
//   el = <feat_src, a_l>_head, er = <feat_dst, a_r>_head            (srec_head_dot)
//   e_uv = LeakyReLU_0.2(el_u + er_v); a = softmax over in-edges of v; rst_v = sum_u a_uv feat_src_u
//                                                                     (srec_gat_agg_fwd)
//   h_v = max_head( sum_rel (rst_rel + bias_rel) + n_rel * x_v )      (srec_head_combine_*)
// Zero-in-degree destinations get rst = 0 (DGL zero fill).
//
// One 64-lane wavefront owns one (destination node, head) pair: its in-edge list (<= MAXDEG, sessions are
// tiny) is staged in LDS, the per-head softmax is a wavefront reduction and the aggregation
// streams 1 KiB feature rows with 16 B per lane.  The backward is two gather-style passes
// (per destination, then per source over the out-edge CSR): deterministic, no atomics.
// Feature layout: [N, H, D] row-major (the GEMM output of fc), ld = H*D.
#include "common.h"

namespace {

constexpr int WPB = 4;
constexpr int MAXDEG = SREC_MAX_DEGREE;
constexpr int MAXH = 8;

// out[n,h] = sum_d X[n,h,d] * a[h,d]
__global__ void head_dot_kernel(const float* __restrict__ X, int ld, const float* __restrict__ a, int n_cap,
                                const int* __restrict__ dyn, int H, int D, float* __restrict__ out) {
    const int gid = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = gid / H, h = gid % H;               // one wavefront per (node, head)
    if (n >= n_cap) return;
    const bool live = n < dyn_count(dyn, n_cap);
    float s = 0.f;
    if (live)
        for (int c = lane * 4; c < D; c += 256) {
            const float4 x = *reinterpret_cast<const float4*>(X + (size_t)n * ld + h * D + c);
            const float4 w = *reinterpret_cast<const float4*>(a + h * D + c);
            s += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
        }
    s = wave_sum(s);
    if (lane == 0) out[(size_t)n * H + h] = s;
}

// per destination node: edge softmax (saved to A[e,h]) + aggregation
__global__ void gat_agg_fwd_kernel(const float* __restrict__ Fs, int ld_s, const float* __restrict__ el,
                                   const float* __restrict__ er, const int* __restrict__ in_ptr,
                                   const int* __restrict__ in_idx, const int* __restrict__ esrc, int nd_cap,
                                   const int* __restrict__ dyn_nd, int H, int D, float slope, float* __restrict__ A,
                                   float* __restrict__ rst, int ld_r) {
    __shared__ float sc[WPB][MAXDEG];
    __shared__ int su[WPB][MAXDEG];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * WPB + w;
    const int v = gid / H, h = gid % H;               // one wavefront per (destination, head)
    if (v >= nd_cap) return;
    const bool live = v < dyn_count(dyn_nd, nd_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? min(in_ptr[v + 1] - beg, MAXDEG) : 0;
    for (int j = lane; j < deg; j += 64) su[w][j] = esrc[in_idx[beg + j]];
    __builtin_amdgcn_wave_barrier();
    {
        const float erv = live ? er[(size_t)v * H + h] : 0.f;
        float m = -INFINITY;
        for (int j = lane; j < deg; j += 64) {
            float s = el[(size_t)su[w][j] * H + h] + erv;
            s = s > 0.f ? s : slope * s;
            sc[w][j] = s;
            m = fmaxf(m, s);
        }
        m = wave_max(m);
        float z = 0.f;
        for (int j = lane; j < deg; j += 64) z += expf(sc[w][j] - m);
        z = wave_sum(z);
        const float iz = deg > 0 ? 1.f / z : 0.f;
        for (int j = lane; j < deg; j += 64) {
            const float a = expf(sc[w][j] - m) * iz;
            sc[w][j] = a;
            A[(size_t)in_idx[beg + j] * H + h] = a;
        }
        __builtin_amdgcn_wave_barrier();
        for (int c = lane * 4; c < D; c += 256) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = 0; j < deg; ++j) {
                const float a = sc[w][j];
                const float4 f = *reinterpret_cast<const float4*>(Fs + (size_t)su[w][j] * ld_s + h * D + c);
                o.x += a * f.x; o.y += a * f.y; o.z += a * f.z; o.w += a * f.w;
            }
            *reinterpret_cast<float4*>(rst + (size_t)v * ld_r + h * D + c) = o;
        }
    }
}

// backward pass 1, per destination: d(pre-activation score) of every in-edge -> DP[e,h]; der[v,h]
__global__ void gat_bwd_dst_kernel(const float* __restrict__ dR, int ld_r, const float* __restrict__ Fs, int ld_s,
                                   const float* __restrict__ el, const float* __restrict__ er,
                                   const float* __restrict__ A, const int* __restrict__ in_ptr,
                                   const int* __restrict__ in_idx, const int* __restrict__ esrc, int nd_cap,
                                   const int* __restrict__ dyn_nd, int H, int D, float slope, float* __restrict__ DP,
                                   float* __restrict__ der) {
    __shared__ float da[WPB][MAXDEG];
    __shared__ int su[WPB][MAXDEG];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * WPB + w;
    const int v = gid / H, h = gid % H;
    if (v >= nd_cap) return;
    const bool live = v < dyn_count(dyn_nd, nd_cap);
    const int beg = live ? in_ptr[v] : 0;
    const int deg = live ? min(in_ptr[v + 1] - beg, MAXDEG) : 0;
    for (int j = lane; j < deg; j += 64) su[w][j] = esrc[in_idx[beg + j]];
    __builtin_amdgcn_wave_barrier();
    {
        for (int j = 0; j < deg; ++j) {
            float s = 0.f;
            for (int c = lane * 4; c < D; c += 256) {
                const float4 g = *reinterpret_cast<const float4*>(dR + (size_t)v * ld_r + h * D + c);
                const float4 f = *reinterpret_cast<const float4*>(Fs + (size_t)su[w][j] * ld_s + h * D + c);
                s += g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
            }
            s = wave_sum(s);
            if (lane == 0) da[w][j] = s;
        }
        __builtin_amdgcn_wave_barrier();
        float t = 0.f;
        for (int j = lane; j < deg; j += 64) t += A[(size_t)in_idx[beg + j] * H + h] * da[w][j];
        t = wave_sum(t);
        const float erv = live ? er[(size_t)v * H + h] : 0.f;
        float dsum = 0.f;
        for (int j = lane; j < deg; j += 64) {
            const int e = in_idx[beg + j];
            const float a = A[(size_t)e * H + h];
            const float pre = el[(size_t)su[w][j] * H + h] + erv;
            const float dp = a * (da[w][j] - t) * (pre > 0.f ? 1.f : slope);
            DP[(size_t)e * H + h] = dp;
            dsum += dp;
        }
        dsum = wave_sum(dsum);
        if (lane == 0) der[(size_t)v * H + h] = dsum;
    }
}

// backward pass 2, per source: dFs[u,h,:] = sum_{e in out(u)} A[e,h] dR[dst_e,h,:] + del[u,h] * a_l[h,:]
__global__ void gat_bwd_src_kernel(const float* __restrict__ dR, int ld_r, const float* __restrict__ A,
                                   const float* __restrict__ DP, const float* __restrict__ attn_l,
                                   const int* __restrict__ out_ptr, const int* __restrict__ out_idx,
                                   const int* __restrict__ edst, int ns_cap, const int* __restrict__ dyn_ns, int H,
                                   int D, float* __restrict__ dFs, int ld_s, float* __restrict__ del,
                                   const float* __restrict__ der, const float* __restrict__ attn_r) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * WPB + w;
    const int u = gid / H, h = gid % H;
    if (u >= ns_cap) return;
    const bool live = u < dyn_count(dyn_ns, ns_cap);
    const int beg = live ? out_ptr[u] : 0;
    const int deg = live ? out_ptr[u + 1] - beg : 0;
    {
        float dl = 0.f;
        for (int j = lane; j < deg; j += 64) dl += DP[(size_t)out_idx[beg + j] * H + h];
        dl = wave_sum(dl);
        if (lane == 0) del[(size_t)u * H + h] = dl;
        // same-type relation (source features == destination features): the er-side term der[u,h]*a_r[h,:]
        // is added here instead of materialising a second [N,H,D] gradient
        const float dr = (der != nullptr && live) ? der[(size_t)u * H + h] : 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 al = *reinterpret_cast<const float4*>(attn_l + h * D + c);
            float4 o = make_float4(dl * al.x, dl * al.y, dl * al.z, dl * al.w);
            if (der != nullptr) {
                const float4 ar = *reinterpret_cast<const float4*>(attn_r + h * D + c);
                o.x += dr * ar.x; o.y += dr * ar.y; o.z += dr * ar.z; o.w += dr * ar.w;
            }
            for (int j = 0; j < deg; ++j) {
                const int e = out_idx[beg + j];
                const float a = A[(size_t)e * H + h];
                const float4 g = *reinterpret_cast<const float4*>(dR + (size_t)edst[e] * ld_r + h * D + c);
                o.x += a * g.x; o.y += a * g.y; o.z += a * g.z; o.w += a * g.w;
            }
            *reinterpret_cast<float4*>(dFs + (size_t)u * ld_s + h * D + c) = o;
        }
    }
}

// out[n,h,:] = w[n,h] * a[h,:]
__global__ void head_outer_kernel(const float* __restrict__ wgt, const float* __restrict__ a, int n_cap,
                                  const int* __restrict__ dyn, int H, int D, float* __restrict__ out, int ld) {
    const int gid = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = gid / H, h = gid % H;
    if (n >= n_cap) return;
    const bool live = n < dyn_count(dyn, n_cap);
    {
        const float wv = live ? wgt[(size_t)n * H + h] : 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 av = *reinterpret_cast<const float4*>(a + h * D + c);
            *reinterpret_cast<float4*>(out + (size_t)n * ld + h * D + c) =
                make_float4(wv * av.x, wv * av.y, wv * av.z, wv * av.w);
        }
    }
}

struct RstList {
    const float* p[8];
    int n;
};

// out[v,c] = max_h ( sum_i R_i[v,h,c] + bias[h,c] + nres * x[v,c] ); arg[v,c] = argmax head
__global__ void head_combine_fwd_kernel(RstList rl, int ld_r, const float* __restrict__ x, int ld_x,
                                        const float* __restrict__ bias, float nres, int n_cap,
                                        const int* __restrict__ dyn, int H, int D, float* __restrict__ out,
                                        int ld_o, unsigned char* __restrict__ arg) {
    const int v = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    for (int c = lane * 4; c < D; c += 256) {
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        if (live) {
            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)v * ld_x + c);
            best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            for (int h = 0; h < H; ++h) {
                float4 s = *reinterpret_cast<const float4*>(bias + h * D + c);
                s.x += nres * xv.x; s.y += nres * xv.y; s.z += nres * xv.z; s.w += nres * xv.w;
                for (int i = 0; i < rl.n; ++i) {
                    const float4 r = *reinterpret_cast<const float4*>(rl.p[i] + (size_t)v * ld_r + h * D + c);
                    s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
                }
                if (s.x > best.x) { best.x = s.x; bi.x = h; }
                if (s.y > best.y) { best.y = s.y; bi.y = h; }
                if (s.z > best.z) { best.z = s.z; bi.z = h; }
                if (s.w > best.w) { best.w = s.w; bi.w = h; }
            }
        }
        *reinterpret_cast<float4*>(out + (size_t)v * ld_o + c) = best;
        *reinterpret_cast<uchar4*>(arg + (size_t)v * D + c) = bi;
    }
}

// dR[v,h,c] = dout[v,c] * [h == arg[v,c]]   (the same tensor is the gradient of every R_i)
__global__ void head_combine_bwd_kernel(const float* __restrict__ dout, int ld_o,
                                        const unsigned char* __restrict__ arg, int n_cap,
                                        const int* __restrict__ dyn, int H, int D, float* __restrict__ dR, int ld_r) {
    const int v = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_cap) return;
    const bool live = v < dyn_count(dyn, n_cap);
    for (int c = lane * 4; c < D; c += 256) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        uchar4 bi = make_uchar4(255, 255, 255, 255);
        if (live) {
            g = *reinterpret_cast<const float4*>(dout + (size_t)v * ld_o + c);
            bi = *reinterpret_cast<const uchar4*>(arg + (size_t)v * D + c);
        }
        for (int h = 0; h < H; ++h)
            *reinterpret_cast<float4*>(dR + (size_t)v * ld_r + h * D + c) =
                make_float4(bi.x == h ? g.x : 0.f, bi.y == h ? g.y : 0.f, bi.z == h ? g.z : 0.f, bi.w == h ? g.w : 0.f);
    }
}

inline bool bad(int H, int D, int ld) { return H <= 0 || H > MAXH || D <= 0 || (D & 3) || (ld & 3); }

}  // namespace

extern "C" int srec_head_dot(const float* X, int ld, const float* a, int n_cap, const int* dyn, int H, int D,
                             float* out, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad(H, D, ld)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(head_dot_kernel, dim3(cdiv(n_cap * H, WPB)), dim3(256), 0, (hipStream_t)stream, X, ld, a, n_cap, dyn,
                       H, D, out);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_gat_agg_fwd(const float* Fs, int ld_s, const float* el, const float* er, const int* in_ptr,
                                const int* in_idx, const int* esrc, int nd_cap, const int* dyn_nd, int H, int D,
                                float slope, float* A, float* rst, int ld_r, void* stream) {
    if (nd_cap <= 0) return 0;
    if (bad(H, D, ld_s) || (ld_r & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gat_agg_fwd_kernel, dim3(cdiv(nd_cap * H, WPB)), dim3(256), 0, (hipStream_t)stream, Fs, ld_s, el, er,
                       in_ptr, in_idx, esrc, nd_cap, dyn_nd, H, D, slope, A, rst, ld_r);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_gat_bwd_dst(const float* dR, int ld_r, const float* Fs, int ld_s, const float* el, const float* er,
                                const float* A, const int* in_ptr, const int* in_idx, const int* esrc, int nd_cap,
                                const int* dyn_nd, int H, int D, float slope, float* DP, float* der, void* stream) {
    if (nd_cap <= 0) return 0;
    if (bad(H, D, ld_s) || (ld_r & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gat_bwd_dst_kernel, dim3(cdiv(nd_cap * H, WPB)), dim3(256), 0, (hipStream_t)stream, dR, ld_r, Fs, ld_s,
                       el, er, A, in_ptr, in_idx, esrc, nd_cap, dyn_nd, H, D, slope, DP, der);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_gat_bwd_src(const float* dR, int ld_r, const float* A, const float* DP, const float* attn_l,
                                const int* out_ptr, const int* out_idx, const int* edst, int ns_cap, const int* dyn_ns,
                                int H, int D, float* dFs, int ld_s, float* del, const float* der, const float* attn_r,
                                void* stream) {
    if (ns_cap <= 0) return 0;
    if (bad(H, D, ld_s) || (ld_r & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gat_bwd_src_kernel, dim3(cdiv(ns_cap * H, WPB)), dim3(256), 0, (hipStream_t)stream, dR, ld_r, A, DP,
                       attn_l, out_ptr, out_idx, edst, ns_cap, dyn_ns, H, D, dFs, ld_s, del, der, attn_r);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_head_outer(const float* wgt, const float* a, int n_cap, const int* dyn, int H, int D, float* out,
                               int ld, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad(H, D, ld)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(head_outer_kernel, dim3(cdiv(n_cap * H, WPB)), dim3(256), 0, (hipStream_t)stream, wgt, a, n_cap, dyn, H,
                       D, out, ld);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_head_combine_fwd(const float* const* rsts, int n_rst, int ld_r, const float* x, int ld_x,
                                     const float* bias, float nres, int n_cap, const int* dyn, int H, int D,
                                     float* out, int ld_o, unsigned char* arg, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad(H, D, ld_r) || n_rst < 0 || n_rst > 8 || (ld_x & 3) || (ld_o & 3)) return SREC_BAD_ARG;
    RstList rl{};
    rl.n = n_rst;
    for (int i = 0; i < n_rst; ++i) rl.p[i] = rsts[i];
    hipLaunchKernelGGL(head_combine_fwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, rl, ld_r, x,
                       ld_x, bias, nres, n_cap, dyn, H, D, out, ld_o, arg);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_head_combine_bwd(const float* dout, int ld_o, const unsigned char* arg, int n_cap, const int* dyn,
                                     int H, int D, float* dR, int ld_r, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad(H, D, ld_r) || (ld_o & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(head_combine_bwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, dout, ld_o, arg,
                       n_cap, dyn, H, D, dR, ld_r);
    SREC_LAUNCH_CHECK();
    return 0;
}
