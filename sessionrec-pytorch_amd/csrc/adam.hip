// Fused Adam with coupled L2 weight decay (torch.optim.Adam semantics used by the
// reference: train.py:70-75,101; fix_weight_decay groups train.py:12-23).
//
//   g = g + wd*p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g
//   p = p - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//
// Hyper-parameters live in a small device array so a captured hipGraph can be
// replayed while lr (StepLR) and the bias corrections change between steps:
//   hyper = {step_size = lr/(1-b1^t), beta1, beta2, eps, weight_decay, 1-beta1, 1-beta2, sqrt(1-b2^t)}
// (the derived scalars are computed in double on the host exactly like torch does; 1.f-0.999f in fp32
// is off by 4.7e-5 relative, a visible 2e-8 bias per step)
//
// srec_adam_flat : any parameter, viewed as a flat fp32 array (one streaming pass over p,g,m,v).
// srec_adam_rows : the item-embedding table, one wavefront per row, with the row-wise
//                  epilogues fused into the same pass over HBM: Embedding(max_norm) renorm
//                  (lessr.py:126, msgifsr.py:162) and the next step's cosine scale
//                  cs_v = scale/||E_v|| (niser.py:151, msgifsr.py:279) - so the reference's
//                  per-step full-table normalise passes cost no extra traffic.
#include "common.h"
#include "hyper_role.h"

namespace {

struct Hyper { float step, b1, b2, eps, wd, omb1, omb2, bc2s; };

__device__ __forceinline__ Hyper load_hyper(const float* __restrict__ h) {
    Hyper r{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    return r;
}

__device__ __forceinline__ float adam1(float& p, float g, float& m, float& v, const Hyper& h, float wd,
                                       float step, float bc2s) {
    g += wd * p;
    m = m + h.omb1 * (g - m);                     // exp_avg.lerp_(grad, 1 - beta1)
    v = h.b2 * v + h.omb2 * g * g;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    p -= step * (m / (sqrtf(v) / bc2s + h.eps));  // param.addcdiv_(exp_avg, denom, -step_size)
    return p;
}

__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, const float* __restrict__ hyper, int use_wd) {
    const Hyper h = load_hyper(hyper);
    const float wd = use_wd ? h.wd : 0.f, step = h.step, rs2 = h.bc2s;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            adam1(pp.x, gg.x, mm.x, vv.x, h, wd, step, rs2);
            adam1(pp.y, gg.y, mm.y, vv.y, h, wd, step, rs2);
            adam1(pp.z, gg.z, mm.z, vv.z, h, wd, step, rs2);
            adam1(pp.w, gg.w, mm.w, vv.w, h, wd, step, rs2);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (size_t j = i; j < n; ++j) adam1(p[j], g[j], m[j], v[j], h, wd, step, rs2);
        }
    }
}

// HOLD (d <= 1024): the updated row stays in registers (<= 4 float4 per lane) between the Adam arithmetic and the row-wise
// epilogues - W is stored ONCE, already renormalised when renorm_write is set, and dst16 (nullable) receives the bf16 operand
// copy of the row as the next forward's scoring kernels read it (zero in the columns d .. Dp): the stand-alone
// renorm + copy pass of the next step (srec_renorm_rows_bf16: 11 us, 57 MB at C3) disappears.
template <bool HOLD>
__global__ void adam_rows_kernel(float* __restrict__ W, const float* __restrict__ G, float* __restrict__ M,
                                 float* __restrict__ Vv, int n, int d, int ld, const float* __restrict__ hyper,
                                 int use_wd, float max_norm, int renorm_write, float* __restrict__ cs_out, float cs_scale,
                                 int eps_mode, float cs_eps, const float* __restrict__ proj_cs, float proj_inv_scale,
                                 float* __restrict__ radial, unsigned short* __restrict__ dst16, int Dp) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const Hyper h = load_hyper(hyper);
    const float wd = use_wd ? h.wd : 0.f, step = h.step, rs2 = h.bc2s;
    const size_t off = (size_t)i * ld;
    // proj_cs != NULL: the chain rule of the catalog-row normalisation (rownorm_project) is applied HERE, to the gradient row as
    // it is read: g = G - W (<W, G> - radial) inv^2, inv = proj_cs * proj_inv_scale (the column scale of this step's forward;
    // read before cs_out - possibly the same array - is rewritten below).  radial [n] holds <W, l> of whatever was added to G
    // after the scoring gradient (the lookup gradients, srec_scatter_add_sorted_ex) and is cleared for the next step.  Saves
    // the separate pass over the table gradient (read G + W, write G: 115 MB at C3).
    float pcoef = 0.f;
    float4 pw[4], gw[4];
    if (HOLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + j * 256;
            pw[j] = make_float4(0.f, 0.f, 0.f, 0.f); gw[j] = pw[j];
            if (c < d) { pw[j] = *reinterpret_cast<const float4*>(W + off + c); gw[j] = *reinterpret_cast<const float4*>(G + off + c); }
        }
    }
    if (proj_cs != nullptr) {
        float dot = 0.f;
        if (HOLD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dot += pw[j].x * gw[j].x + pw[j].y * gw[j].y + pw[j].z * gw[j].z + pw[j].w * gw[j].w;
        } else {
            for (int c = lane * 4; c < d; c += 256) {
                const float4 pp = *reinterpret_cast<const float4*>(W + off + c);
                const float4 gg = *reinterpret_cast<const float4*>(G + off + c);
                dot += pp.x * gg.x + pp.y * gg.y + pp.z * gg.z + pp.w * gg.w;
            }
        }
        dot = wave_sum(dot);
        const float iv = proj_cs[i] * proj_inv_scale;
        if (radial != nullptr) {
            dot -= radial[i];
            if (lane == 0) radial[i] = 0.f;
        }
        pcoef = dot * iv * iv;
    }
    float ss = 0.f;
    if (HOLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + j * 256;
            if (c < d) {
                float4 pp = pw[j], gg = gw[j];
                gg.x -= pp.x * pcoef; gg.y -= pp.y * pcoef; gg.z -= pp.z * pcoef; gg.w -= pp.w * pcoef;
                float4 mm = *reinterpret_cast<float4*>(M + off + c);
                float4 vv = *reinterpret_cast<float4*>(Vv + off + c);
                adam1(pp.x, gg.x, mm.x, vv.x, h, wd, step, rs2);
                adam1(pp.y, gg.y, mm.y, vv.y, h, wd, step, rs2);
                adam1(pp.z, gg.z, mm.z, vv.z, h, wd, step, rs2);
                adam1(pp.w, gg.w, mm.w, vv.w, h, wd, step, rs2);
                ss += pp.x * pp.x + pp.y * pp.y + pp.z * pp.z + pp.w * pp.w;
                pw[j] = pp;
                *reinterpret_cast<float4*>(M + off + c) = mm;
                *reinterpret_cast<float4*>(Vv + off + c) = vv;
            }
        }
        float nrm = 0.f;
        float sc = 1.f;
        if (max_norm > 0.f || cs_out != nullptr) {
            nrm = sqrtf(wave_sum(ss));
            if (max_norm > 0.f && nrm > max_norm) {
                // renorm_write == 0: W keeps the plain Adam result (the reference renormalises in the NEXT forward,
                // msgifsr.py:162,247) and only cs is that of the row as the next forward will see it
                const float s1 = max_norm / (nrm + 1e-7f);
                if (renorm_write) sc = s1;
                nrm *= s1;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + j * 256;
            float4 pp = pw[j];
            if (sc != 1.f) { pp.x *= sc; pp.y *= sc; pp.z *= sc; pp.w *= sc; }
            if (c < d) *reinterpret_cast<float4*>(W + off + c) = pp;
            if (dst16 != nullptr && c < Dp)
                *reinterpret_cast<uint2*>(dst16 + (size_t)i * Dp + c) = make_uint2(srec_pack_bf16(pp.x, pp.y), srec_pack_bf16(pp.z, pp.w));
        }
        if (cs_out != nullptr && lane == 0)
            cs_out[i] = cs_scale * (eps_mode == 0 ? 1.f / fmaxf(nrm, cs_eps) : 1.f / (nrm + cs_eps));
        return;
    }
    for (int c = lane * 4; c < d; c += 256) {
        float4 pp = *reinterpret_cast<float4*>(W + off + c);
        float4 gg = *reinterpret_cast<const float4*>(G + off + c);
        gg.x -= pp.x * pcoef; gg.y -= pp.y * pcoef; gg.z -= pp.z * pcoef; gg.w -= pp.w * pcoef;
        float4 mm = *reinterpret_cast<float4*>(M + off + c);
        float4 vv = *reinterpret_cast<float4*>(Vv + off + c);
        adam1(pp.x, gg.x, mm.x, vv.x, h, wd, step, rs2);
        adam1(pp.y, gg.y, mm.y, vv.y, h, wd, step, rs2);
        adam1(pp.z, gg.z, mm.z, vv.z, h, wd, step, rs2);
        adam1(pp.w, gg.w, mm.w, vv.w, h, wd, step, rs2);
        ss += pp.x * pp.x + pp.y * pp.y + pp.z * pp.z + pp.w * pp.w;
        *reinterpret_cast<float4*>(W + off + c) = pp;
        *reinterpret_cast<float4*>(M + off + c) = mm;
        *reinterpret_cast<float4*>(Vv + off + c) = vv;
    }
    if (max_norm <= 0.f && cs_out == nullptr) return;
    float nrm = sqrtf(wave_sum(ss));
    if (max_norm > 0.f && nrm > max_norm) {
        const float sc = max_norm / (nrm + 1e-7f);
        for (int c = lane * 4; renorm_write && c < d; c += 256) {       // same lane re-reads what it wrote
            float4 pp = *reinterpret_cast<float4*>(W + off + c);
            pp.x *= sc; pp.y *= sc; pp.z *= sc; pp.w *= sc;
            *reinterpret_cast<float4*>(W + off + c) = pp;
        }
        nrm *= sc;
    }
    if (cs_out != nullptr && lane == 0)
        cs_out[i] = cs_scale * (eps_mode == 0 ? 1.f / fmaxf(nrm, cs_eps) : 1.f / (nrm + cs_eps));
}

// Multi-tensor variant: ONE launch updates up to MT (small) parameter tensors of a group.  The descriptor travels BY
// VALUE in the kernel arguments (no device-side descriptor arrays to stage, nothing to re-upload when autograd hands
// out new gradient buffers; a captured hipGraph bakes it into the kernel node).  Workgroup b owns a 4096-element chunk
// of one tensor: blk0[] is the prefix sum of chunks, found by a scalar binary search; float4 accesses when the four
// pointers are 16-B aligned, scalar otherwise / for the tail.
constexpr int MT = 88;      // 88 x 40 B of descriptor + masks: inside the 4 KB kernel-argument budget
constexpr int MCHUNK = 4096;
struct MultiArgs {
    float* p[MT];
    const float* g[MT];
    float* m[MT];
    float* v[MT];
    int n[MT];
    int blk0[MT + 1];
    unsigned long long wd_mask[2];
    const float* hyper;
    int nt;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(MultiArgs a) {
    int lo = 0, hi = a.nt;                                   // largest t with blk0[t] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int t = lo;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const int n = a.n[t];
    const Hyper h = load_hyper(a.hyper);
    const float wd = ((a.wd_mask[t >> 6] >> (t & 63)) & 1ull) ? h.wd : 0.f, step = h.step, rs2 = h.bc2s;
    const int e0 = ((int)blockIdx.x - a.blk0[t]) * MCHUNK;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
#pragma unroll
    for (int u = 0; u < MCHUNK / 1024; ++u) {
        const int i = e0 + u * 1024 + (int)threadIdx.x * 4;
        if (i >= n) break;
        if (vec && i + 3 < n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            adam1(pp.x, gg.x, mm.x, vv.x, h, wd, step, rs2);
            adam1(pp.y, gg.y, mm.y, vv.y, h, wd, step, rs2);
            adam1(pp.z, gg.z, mm.z, vv.z, h, wd, step, rs2);
            adam1(pp.w, gg.w, mm.w, vv.w, h, wd, step, rs2);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int j = i; j < n && j < i + 4; ++j) adam1(p[j], g[j], m[j], v[j], h, wd, step, rs2);
        }
    }
}

}  // namespace

// desc: HOST srec_adam_multi_desc (srec.h) - nt tensors {p, g, m, v, numel, use_wd}; any nt (launched MT at a time).
extern "C" int srec_adam_multi(const void* desc_, const float* hyper, void* stream) {
    struct Desc { int nt; const int* use_wd; const long* numel; float* const* p; const float* const* g; float* const* m; float* const* v; };
    const Desc* d = (const Desc*)desc_;
    if (d == nullptr || d->nt < 0 || hyper == nullptr) return SREC_BAD_ARG;
    for (int t0 = 0; t0 < d->nt; t0 += MT) {
        MultiArgs a{};
        a.nt = d->nt - t0 < MT ? d->nt - t0 : MT;
        a.hyper = hyper;
        int blocks = 0;
        for (int t = 0; t < a.nt; ++t) {
            const int s = t0 + t;
            if (d->numel[s] <= 0 || d->numel[s] > 0x7fffffffL || d->p[s] == nullptr || d->g[s] == nullptr ||
                d->m[s] == nullptr || d->v[s] == nullptr || (((uintptr_t)d->p[s] | (uintptr_t)d->g[s] | (uintptr_t)d->m[s] | (uintptr_t)d->v[s]) & 3))
                return SREC_BAD_ARG;
            a.p[t] = d->p[s]; a.g[t] = d->g[s]; a.m[t] = d->m[s]; a.v[t] = d->v[s];
            a.n[t] = (int)d->numel[s];
            if (d->use_wd[s]) a.wd_mask[t >> 6] |= 1ull << (t & 63);
            a.blk0[t] = blocks;
            blocks += (a.n[t] + MCHUNK - 1) / MCHUNK;
        }
        a.blk0[a.nt] = blocks;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    SREC_LAUNCH_CHECK();
    return 0;
}

namespace {
// one thread: advance the DEVICE step counter and derive the step's scalars in double (like torch on the host).
// Keeping the counter on the device makes the optimizer step a pure function of device state: a replayed hipGraph
// (or a host running ahead of the GPU) can no longer pair a step with another step's bias corrections.
__global__ void adam_hyper_kernel(int* __restrict__ counter, const double* __restrict__ cfg, float* __restrict__ hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int t = *counter + 1;
    *counter = t;
    const double lr = cfg[0], b1 = cfg[1], b2 = cfg[2], eps = cfg[3], wd = cfg[4];
    hyper[0] = (float)(lr / (1.0 - pow(b1, (double)t)));
    hyper[1] = (float)b1; hyper[2] = (float)b2; hyper[3] = (float)eps; hyper[4] = (float)wd;
    hyper[5] = (float)(1.0 - b1); hyper[6] = (float)(1.0 - b2);
    hyper[7] = (float)sqrt(1.0 - pow(b2, (double)t));
}
}  // namespace

namespace {
__global__ void adam_hyper_multi_kernel(HyperArgs a) { hyper_role(a, (int)threadIdx.x); }
}  // namespace

// the same for n <= 16 (counter, cfg, hyper) slots in ONE launch (param groups x step offsets of an optimizer step);
// counter / cfg / hyper are HOST arrays of n device pointers.  Optional loss tap (all four or none): the slot whose counter
// is tap_counter also stores *tap_src into tap_ring[(its count before this step) % tap_n].  skip (nullable, device int32): when
// non-zero the step's scalars turn every Adam kernel that reads them into the identity (see the kernel).
extern "C" int srec_adam_hyper_multi(int n, const void* counter, const void* cfg, const void* hyper, const int* tap_counter,
                                     const float* tap_src, float* tap_ring, int tap_n, const int* skip, void* stream) {
    if (n <= 0) return 0;
    HyperArgs a{};
    if (int rc = hyper_fill(n, counter, cfg, hyper, tap_counter, tap_src, tap_ring, tap_n, skip, a)) return rc;
    hipLaunchKernelGGL(adam_hyper_multi_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// counter: device int32 (steps taken so far, incremented here); cfg: device double[5] = {lr, beta1, beta2, eps,
// weight_decay}; hyper: device float[8] consumed by srec_adam_flat / _rows / _multi of the same step.
extern "C" int srec_adam_hyper(int* counter, const void* cfg, float* hyper, void* stream) {
    if (counter == nullptr || cfg == nullptr || hyper == nullptr) return SREC_BAD_ARG;
    hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, (const double*)cfg, hyper);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_adam_flat(float* p, const float* g, float* m, float* v, long n, const float* hyper, int use_wd,
                              void* stream) {
    if (n <= 0) return 0;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return SREC_BAD_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (size_t)n, hyper, use_wd);
    SREC_LAUNCH_CHECK();
    return 0;
}

static int adam_rows_run(float* W, const float* G, float* M, float* V, int n, int d, int ld, const float* hyper, int use_wd,
                         float max_norm, int renorm_write, float* cs_out, float cs_scale, int eps_mode, float cs_eps,
                         const float* proj_cs, float proj_inv_scale, float* radial, void* dst16, int Dp, void* stream) {
    if (n <= 0) return 0;
    if (d <= 0 || (d & 3) || (ld & 3)) return SREC_BAD_ARG;
    if (dst16 != nullptr && (d > 1024 || Dp < d || (Dp & 3) || Dp > 1024)) return SREC_BAD_ARG;
    if (d <= 1024)
        hipLaunchKernelGGL(adam_rows_kernel<true>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, W, G, M, V, n, d, ld, hyper,
                           use_wd, max_norm, renorm_write, cs_out, cs_scale, eps_mode, cs_eps, proj_cs, proj_inv_scale, radial,
                           (unsigned short*)dst16, Dp);
    else
        hipLaunchKernelGGL(adam_rows_kernel<false>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, W, G, M, V, n, d, ld, hyper,
                           use_wd, max_norm, renorm_write, cs_out, cs_scale, eps_mode, cs_eps, proj_cs, proj_inv_scale, radial,
                           (unsigned short*)nullptr, 0);
    SREC_LAUNCH_CHECK();
    return 0;
}

// dst16 (nullable; d <= 1024): bf16 copy [n, Dp] of the rows as written (Dp >= d, zero padding columns) - with renorm_write the
// renormalised rows, i.e. what srec_renorm_rows_bf16 would produce at the start of the next forward
extern "C" int srec_adam_rows(float* W, const float* G, float* M, float* V, int n, int d, int ld, const float* hyper,
                              int use_wd, float max_norm, int renorm_write, float* cs_out, float cs_scale, int eps_mode,
                              float cs_eps, void* dst16, int Dp, void* stream) {
    return adam_rows_run(W, G, M, V, n, d, ld, hyper, use_wd, max_norm, renorm_write, cs_out, cs_scale, eps_mode, cs_eps, nullptr,
                         0.f, nullptr, dst16, Dp, stream);
}

// srec_adam_rows with the row-normalisation projection of the gradient fused in (see adam_rows_kernel)
extern "C" int srec_adam_rows_proj(float* W, const float* G, float* M, float* V, int n, int d, int ld, const float* hyper,
                                   int use_wd, float max_norm, int renorm_write, float* cs_out, float cs_scale, int eps_mode,
                                   float cs_eps, const float* proj_cs, float proj_inv_scale, float* radial, void* dst16, int Dp,
                                   void* stream) {
    if (proj_cs == nullptr) return SREC_BAD_ARG;
    return adam_rows_run(W, G, M, V, n, d, ld, hyper, use_wd, max_norm, renorm_write, cs_out, cs_scale, eps_mode, cs_eps, proj_cs,
                         proj_inv_scale, radial, dst16, Dp, stream);
}
