// k-gram GRU of MSGIFSR's SemanticExpander (msgifsr.py:25,32-45), all orders of a batch together - the bf16 path.
//
// The per-order, per-step formulation (gru.hip + one GEMM, one split-K reduction, one pointwise and one column-sum launch
// per step and order) cost ~55 launches and ~0.5 ms per training step at the C3 shapes, nearly all of it launch and
// latency overhead of ~5 us kernels.  Here ONE launch per time step serves every order (a "problem" = one order k with
// its n_k nodes), the projections run as grouped bf16-in-HBM GEMMs (gemm16.hip) and everything pointwise is folded into
// the step kernels:
//   forward  step t: gates from GI[:, t] (+ b_ih) and GH (+ b_hh; h_{-1} = 0 at t = 0), new hidden state as fp32 AND
//            as the bf16 operand of the next step's GEMM; at an order's last step also the expander output
//            0.5 * mean_t x + 0.5 * h_last (msgifsr.py:37,45).
//   backward step t: d(gi), d(gh) as bf16 GEMM operands, the direct term of d h_{t-1}, per-block partial column sums of
//            both (the bias gradients: one final reduction launch for all orders), and at the last step the mean term
//            of d x (the d(gi) W_ih GEMM accumulates onto it).
// Saved per step: gates [n, 4 d] = r, z, n, (gh_n + b_hh_n).
#include "common.h"
#include "hyper_role.h"
#include "../../include/srec_hg.h"

namespace {

constexpr int GX_MAXP = SREC_GRU_MAXP;

struct StepArgs {
    srec_gru_step_desc d;
    int start[GX_MAXP + 1];      // first block of each problem
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }

// one thread = 4 consecutive hidden units of one node
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(StepArgs a) {
    const srec_gru_step_desc& q = a.d;
    int p = 0;
#pragma unroll
    for (int i = 1; i < GX_MAXP; ++i)
        if (i < q.np && (int)blockIdx.x >= a.start[i]) p = i;
    const int d = q.d, d4 = d >> 2, n = q.n[p], k = q.k[p], t = q.t[p];
    const long idx = (long)((int)blockIdx.x - a.start[p]) * 256 + threadIdx.x;
    const int row = (int)(idx / d4), c = (int)(idx % d4) * 4;
    if (row >= n) return;
    float* Hn = q.Hn[p] + (size_t)row * d + c;
    unsigned short* Hn16 = (unsigned short*)q.Hn16[p];
    const bool last = t == k - 1 && q.out[p] != nullptr;
    if (row >= dyn_count(q.dyn[p], n)) {
        *reinterpret_cast<float4*>(Hn) = f4(0.f);
        if (Hn16 != nullptr) *reinterpret_cast<uint2*>(Hn16 + (size_t)row * d + c) = make_uint2(0u, 0u);
        if (last) *reinterpret_cast<float4*>(q.out[p] + (size_t)row * d + c) = f4(0.f);
        return;
    }
    const float* gi = q.GI[p] + ((size_t)row * k + t) * 3 * d + c;
    const float* bi = q.bih[p] + c;
    const float* bh = q.bhh[p] + c;
    float4 ir = ld4(gi), iz = ld4(gi + d), in = ld4(gi + 2 * d);
    const float4 bir = ld4(bi), biz = ld4(bi + d), bin = ld4(bi + 2 * d);
    float4 hr = ld4(bh), hz = ld4(bh + d), hn = ld4(bh + 2 * d), hp = f4(0.f);
    if (q.GH[p] != nullptr) {
        const float* gh = q.GH[p] + (size_t)row * 3 * d + c;
        const float4 gr = ld4(gh), gz = ld4(gh + d), gn = ld4(gh + 2 * d);
        hr.x += gr.x; hr.y += gr.y; hr.z += gr.z; hr.w += gr.w;
        hz.x += gz.x; hz.y += gz.y; hz.z += gz.z; hz.w += gz.w;
        hn.x += gn.x; hn.y += gn.y; hn.z += gn.z; hn.w += gn.w;
        hp = ld4(q.Hp[p] + (size_t)row * d + c);
    }
    float4 r, z, nn, h;
#define GX_LANE(e)                                                                 \
    r.e = sigmoidf_(ir.e + bir.e + hr.e);                                           \
    z.e = sigmoidf_(iz.e + biz.e + hz.e);                                           \
    nn.e = tanhf(in.e + bin.e + r.e * hn.e);                                        \
    h.e = (1.f - z.e) * nn.e + z.e * hp.e;
    GX_LANE(x) GX_LANE(y) GX_LANE(z) GX_LANE(w)
#undef GX_LANE
    *reinterpret_cast<float4*>(Hn) = h;
    if (Hn16 != nullptr) {
        uint2 o;
        o.x = srec_pack_bf16(h.x, h.y); o.y = srec_pack_bf16(h.z, h.w);
        *reinterpret_cast<uint2*>(Hn16 + (size_t)row * d + c) = o;
    }
    float* g = q.gates[p] + (size_t)row * 4 * d + c;
    *reinterpret_cast<float4*>(g) = r;
    *reinterpret_cast<float4*>(g + d) = z;
    *reinterpret_cast<float4*>(g + 2 * d) = nn;
    *reinterpret_cast<float4*>(g + 3 * d) = hn;
    if (last) {                                          // 0.5 * mean_t x[row, t, :] + 0.5 * h_last
        float4 s = f4(0.f);
        const float* x = q.X[p] + (size_t)row * k * d + c;
        for (int tt = 0; tt < k; ++tt) {
            const float4 v = ld4(x + (size_t)tt * d);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float ik = 0.5f / (float)k;
        *reinterpret_cast<float4*>(q.out[p] + (size_t)row * d + c) =
            make_float4(ik * s.x + 0.5f * h.x, ik * s.y + 0.5f * h.y, ik * s.z + 0.5f * h.z, ik * s.w + 0.5f * h.w);
    }
}

// backward of one step.  Block = 256 threads = RP rows x (d / 4) column quads at a time, RB rows per block; every thread
// keeps the column sums of d(gi) / d(gh) of its 4 columns over the block's rows, reduced over the RP row lanes through
// LDS and written as one partial row [6 d] per block (fixed order: deterministic).
constexpr int GX_RB = 8;
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(StepArgs a) {
    __shared__ float red[256 * 24];
    const srec_gru_step_desc& q = a.d;
    int p = 0;
#pragma unroll
    for (int i = 1; i < GX_MAXP; ++i)
        if (i < q.np && (int)blockIdx.x >= a.start[i]) p = i;
    const int d = q.d, d4 = d >> 2, n = q.n[p], k = q.k[p], t = q.t[p];
    const int RP = 256 / d4;                             // rows handled per pass
    const int RB = GX_RB > RP ? GX_RB : RP;              // rows per block
    const int blk = (int)blockIdx.x - a.start[p];
    const int lane_r = threadIdx.x / d4, c = (threadIdx.x % d4) * 4;
    const int nl = dyn_count(q.dyn[p], n);
    const bool last = t == k - 1 && q.dout[p] != nullptr;
    unsigned short* dGI16 = (unsigned short*)q.dGI16[p];
    unsigned short* dGH16 = (unsigned short*)q.dGH16[p];
    float si[12], sh[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) si[e] = sh[e] = 0.f;
    for (int r0 = 0; r0 < RB; r0 += RP) {
        const int row = blk * RB + r0 + lane_r;
        if (row >= n) continue;
        const size_t grow = (size_t)row * k + t;
        if (row >= nl) {                                 // capacity padding: zero operands (the GEMMs clamp rows anyway)
            for (int e = 0; e < 3; ++e) *reinterpret_cast<uint2*>(dGI16 + grow * 3 * d + e * d + c) = make_uint2(0u, 0u);
            if (dGH16 != nullptr)
                for (int e = 0; e < 3; ++e) *reinterpret_cast<uint2*>(dGH16 + (size_t)row * 3 * d + e * d + c) = make_uint2(0u, 0u);
            if (q.dHp[p] != nullptr) *reinterpret_cast<float4*>(q.dHp[p] + (size_t)row * d + c) = f4(0.f);
            if (last)
                for (int tt = 0; tt < k; ++tt) *reinterpret_cast<float4*>(q.dX[p] + ((size_t)row * k + tt) * d + c) = f4(0.f);
            continue;
        }
        const float* g = q.gates[p] + (size_t)row * 4 * d + c;
        const float4 r = ld4(g), z = ld4(g + d), nn = ld4(g + 2 * d), ghn = ld4(g + 3 * d);
        float4 hp = f4(0.f), dh;
        if (q.Hp[p] != nullptr) hp = ld4(q.Hp[p] + (size_t)row * d + c);
        if (last) {
            const float4 go = ld4(q.dout[p] + (size_t)row * d + c);
            dh = make_float4(0.5f * go.x, 0.5f * go.y, 0.5f * go.z, 0.5f * go.w);
            const float ik = 1.f / (float)k;
            const float4 gx = make_float4(dh.x * ik, dh.y * ik, dh.z * ik, dh.w * ik);
            for (int tt = 0; tt < k; ++tt) *reinterpret_cast<float4*>(q.dX[p] + ((size_t)row * k + tt) * d + c) = gx;
        } else {
            dh = ld4(q.dH[p] + (size_t)row * d + c);
        }
        float4 dpr, dpz, dpn, dgn;
#define GX_LANE(e)                                                                 \
    {                                                                              \
        const float dn = dh.e * (1.f - z.e), dz = dh.e * (hp.e - nn.e);            \
        dpn.e = dn * (1.f - nn.e * nn.e);                                          \
        dpr.e = dpn.e * ghn.e * r.e * (1.f - r.e);                                 \
        dpz.e = dz * z.e * (1.f - z.e);                                            \
        dgn.e = dpn.e * r.e;                                                       \
    }
        GX_LANE(x) GX_LANE(y) GX_LANE(z) GX_LANE(w)
#undef GX_LANE
        auto st16 = [](unsigned short* dst, const float4& v) {
            uint2 o;
            o.x = srec_pack_bf16(v.x, v.y); o.y = srec_pack_bf16(v.z, v.w);
            *reinterpret_cast<uint2*>(dst) = o;
        };
        st16(dGI16 + grow * 3 * d + c, dpr);
        st16(dGI16 + grow * 3 * d + d + c, dpz);
        st16(dGI16 + grow * 3 * d + 2 * d + c, dpn);
        if (dGH16 != nullptr) {
            st16(dGH16 + (size_t)row * 3 * d + c, dpr);
            st16(dGH16 + (size_t)row * 3 * d + d + c, dpz);
            st16(dGH16 + (size_t)row * 3 * d + 2 * d + c, dgn);
        }
        if (q.dHp[p] != nullptr)
            *reinterpret_cast<float4*>(q.dHp[p] + (size_t)row * d + c) = make_float4(dh.x * z.x, dh.y * z.y, dh.z * z.z, dh.w * z.w);
        si[0] += dpr.x; si[1] += dpr.y; si[2] += dpr.z; si[3] += dpr.w;
        si[4] += dpz.x; si[5] += dpz.y; si[6] += dpz.z; si[7] += dpz.w;
        si[8] += dpn.x; si[9] += dpn.y; si[10] += dpn.z; si[11] += dpn.w;
        sh[8] += dgn.x; sh[9] += dgn.y; sh[10] += dgn.z; sh[11] += dgn.w;
    }
    // d(gh) shares its r / z parts with d(gi)
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[e] = si[e];
#pragma unroll
    for (int e = 0; e < 12; ++e) { red[threadIdx.x * 24 + e] = si[e]; red[threadIdx.x * 24 + 12 + e] = sh[e]; }
    __syncthreads();
    if (lane_r == 0) {
        float* out = q.bias_part[p] + ((size_t)q.part_row0[p] + blk) * 6 * d;
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            float s = 0.f;
            for (int rr = 0; rr < RP; ++rr) s += red[(rr * d4 + threadIdx.x) * 24 + e];
            // e = 4 * gate + j (d(gi)), 12 + 4 * gate + j (d(gh)): column gate * d + c + j of that half
            const int half = e / 12, gate = (e % 12) / 4, j = e & 3;
            out[half * 3 * d + gate * d + c + j] = s;
        }
    }
}

// gb[p][col] = sum over the partial rows of problem p (fixed order): block = 64 columns x 16 row lanes (each a strided
// pass over its share of the rows, 8 independent loads in flight), LDS-combined
struct FinArgs { const float* part[GX_MAXP]; float* out[GX_MAXP]; int rows[GX_MAXP]; int ncol; int np; };
__global__ __launch_bounds__(1024) void gru_bias_final_kernel(FinArgs a) {
    __shared__ float red[16][64];
    const int p = blockIdx.y, cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (p < a.np && col < a.ncol) {
        const int R = a.rows[p];
        const float* q = a.part[p] + col;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        int r = rl;
        for (; r + 7 * 16 < R; r += 8 * 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += q[(size_t)(r + 16 * e) * a.ncol];
        }
        for (; r < R; r += 16) acc[0] += q[(size_t)r * a.ncol];
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && p < a.np && col < a.ncol) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) t += red[e][cl];
        a.out[p][col] = t;
    }
}

// out_i [n_i] = sum_r part_i [R_i, n_i].  tall_i: few columns, hundreds of rows (the GRU's bias partials: one row per gate-kernel
// workgroup) - a workgroup takes 64 columns with 4 row lanes of 8 independent accumulators instead of one thread per 4 columns
// walking all the rows.  w_i / ld_i: the output is a [n_i / w_i, w_i] block of a matrix with row stride ld_i (a column slice of a
// weight gradient); w_i = 0: contiguous
constexpr int SLAB_MAXP = 32;
struct SlabArgs {
    const float* part[SLAB_MAXP]; float* out[SLAB_MAXP]; long n[SLAB_MAXP]; int R[SLAB_MAXP]; int start[SLAB_MAXP + 1];
    int w[SLAB_MAXP]; int ld[SLAB_MAXP]; unsigned tall; int np;
};
// (h.n > 0: the workgroup behind the last task carries the optimizer's step-scalar role - hyper_role.h)
__global__ void sum_slabs_multi_kernel(SlabArgs a, HyperArgs h) {
    __shared__ float red[4][64];
    if ((int)blockIdx.x >= a.start[a.np]) {
        hyper_role(h, (int)threadIdx.x);
        return;
    }
    int p = 0;
    for (int i = 1; i < a.np; ++i)
        if ((int)blockIdx.x >= a.start[i]) p = i;
    if ((a.tall >> p) & 1u) {                              // (uniform per workgroup)
        const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
        const long col = (long)((int)blockIdx.x - a.start[p]) * 64 + cl;
        const long n = a.n[p];
        float s = 0.f;
        if (col < n) {
            const int R = a.R[p];
            const float* q = a.part[p] + col;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            int r = rl;
            for (; r + 7 * 4 < R; r += 8 * 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += q[(size_t)(r + 4 * e) * n];
            }
            for (; r < R; r += 4) acc[0] += q[(size_t)r * n];
            s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
        red[rl][cl] = s;
        __syncthreads();
        if (rl == 0 && col < n) a.out[p][col] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        return;
    }
    const long i = ((long)((int)blockIdx.x - a.start[p]) * blockDim.x + threadIdx.x) * 4;
    if (i >= a.n[p]) return;
    const float* q = a.part[p] + i;
    const size_t n = (size_t)a.n[p];
    const int R = a.R[p];
    float4 s = ld4(q);
    int r = 1;
    for (; r + 8 <= R; r += 8) {                 // 8 slab loads in flight (the k-split products leave 25 - 32 slabs: a latency chain
        float4 v[8];                             // otherwise); added in slab order: the same sum as the one-by-one loop
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ld4(q + (size_t)(r + e) * n);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s.x += v[e].x; s.y += v[e].y; s.z += v[e].z; s.w += v[e].w; }
    }
    for (; r < R; ++r) {
        const float4 v = ld4(q + (size_t)r * n);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int w = a.w[p];
    const long o = w > 0 ? (i / w) * (long)a.ld[p] + (i % w) : i;
    *reinterpret_cast<float4*>(a.out[p] + o) = s;
}

int check(const srec_gru_step_desc* q) {
    if (q == nullptr || q->np <= 0 || q->np > GX_MAXP || q->d <= 0 || (q->d & 3)) return SREC_BAD_ARG;
    const int d4 = q->d / 4;
    if (d4 > 256 || 256 % d4) return SREC_BAD_ARG;
    return 0;
}

}  // namespace

// one GRU time step of up to 4 orders: desc = host srec_gru_step_desc (srec_hg.h)
extern "C" int srec_gru_step_fwd(const void* desc, void* stream) {
    const srec_gru_step_desc* q = (const srec_gru_step_desc*)desc;
    if (int rc = check(q)) return rc;
    StepArgs a{};
    a.d = *q;
    int blocks = 0;
    for (int p = 0; p < q->np; ++p) {
        if (q->n[p] <= 0 || q->GI[p] == nullptr || q->bih[p] == nullptr || q->bhh[p] == nullptr || q->Hn[p] == nullptr ||
            q->gates[p] == nullptr || (q->GH[p] != nullptr && q->Hp[p] == nullptr))
            return SREC_BAD_ARG;
        a.start[p] = blocks;
        blocks += (int)(((long)q->n[p] * (q->d / 4) + 255) / 256);
    }
    a.start[q->np] = blocks;
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// bias_part[p] must hold (part_row0[p] + ceil(n / max(8, 1024 / d))) rows of 6 d floats: this launch writes rows part_row0[p] ...
extern "C" int srec_gru_step_bwd(const void* desc, void* stream) {
    const srec_gru_step_desc* q = (const srec_gru_step_desc*)desc;
    if (int rc = check(q)) return rc;
    StepArgs a{};
    a.d = *q;
    int blocks = 0;
    for (int p = 0; p < q->np; ++p) {
        if (q->n[p] <= 0 || q->gates[p] == nullptr || q->dGI16[p] == nullptr || q->bias_part[p] == nullptr ||
            (q->dout[p] == nullptr && q->dH[p] == nullptr) || (q->dout[p] != nullptr && q->t[p] == q->k[p] - 1 && q->dX[p] == nullptr))
            return SREC_BAD_ARG;
        a.start[p] = blocks;
        blocks += cdiv(q->n[p], (GX_RB > 1024 / q->d ? GX_RB : 1024 / q->d));
    }
    a.start[q->np] = blocks;
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// out[p][ncol] = column sums of part[p] [rows[p], ncol] for np <= 4 problems (HOST arrays), one launch
extern "C" int srec_gru_bias_final(int np, const void* part, const int* rows, int ncol, const void* out, void* stream) {
    if (np <= 0) return 0;
    if (np > GX_MAXP || part == nullptr || out == nullptr || rows == nullptr || ncol <= 0) return SREC_BAD_ARG;
    FinArgs a{};
    a.np = np; a.ncol = ncol;
    for (int p = 0; p < np; ++p) {
        a.part[p] = ((const float* const*)part)[p]; a.out[p] = ((float* const*)out)[p]; a.rows[p] = rows[p];
        if (a.part[p] == nullptr || a.out[p] == nullptr) return SREC_BAD_ARG;
    }
    hipLaunchKernelGGL(gru_bias_final_kernel, dim3(cdiv(ncol, 64), np), dim3(1024), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// np <= 32 outputs out_i [n_i] = sum_r part_i [R_i, n_i] (n_i % 4 == 0 and 16-byte aligned unless tall_i) in one launch; HOST arrays.  tall (nullable): tall_i != 0
// marks an output of few columns summed over many rows (row lanes instead of column threads); w / ld (nullable, both or neither):
// w_i > 0 = out_i is a block of w_i columns in rows of stride ld_i (both % 4 == 0; not with tall_i)
static int slab_sums(int np, const void* part, const int* R, const long* n, const void* out, const int* tall, const int* w,
                     const int* ld, const HyperArgs& h, void* stream) {
    if (np <= 0) return h.n > 0 ? SREC_BAD_ARG : 0;
    if (np > SLAB_MAXP || part == nullptr || out == nullptr || (w == nullptr) != (ld == nullptr)) return SREC_BAD_ARG;
    SlabArgs a{};
    a.np = np;
    int blocks = 0;
    for (int p = 0; p < np; ++p) {
        a.part[p] = ((const float* const*)part)[p]; a.out[p] = ((float* const*)out)[p]; a.R[p] = R[p]; a.n[p] = n[p];
        const bool tl = tall != nullptr && tall[p] != 0;
        if (tl) a.tall |= 1u << p;
        // (column-thread tasks move 16 bytes per access; the row-lane ("tall") tasks are scalar: any n, any alignment)
        if (a.part[p] == nullptr || a.out[p] == nullptr || R[p] <= 0 || n[p] <= 0 || (!tl && (n[p] & 3))) return SREC_BAD_ARG;
        if (!tl && ((((size_t)a.part[p]) | ((size_t)a.out[p])) & 15)) return SREC_BAD_ARG;
        a.w[p] = 0; a.ld[p] = 0;
        if (w != nullptr && w[p] > 0 && ld[p] != w[p]) {
            if (tl || (w[p] & 3) || (ld[p] & 3) || ld[p] < w[p] || n[p] % w[p]) return SREC_BAD_ARG;
            a.w[p] = w[p]; a.ld[p] = ld[p];
        }
        a.start[p] = blocks;
        blocks += tl ? (int)((n[p] + 63) / 64) : (int)((n[p] / 4 + 255) / 256);
    }
    a.start[np] = blocks;
    hipLaunchKernelGGL(sum_slabs_multi_kernel, dim3(blocks + (h.n > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, a, h);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_sum_slabs_multi_ld(int np, const void* part, const int* R, const long* n, const void* out, const int* tall,
                                       const int* w, const int* ld, void* stream) {
    return slab_sums(np, part, R, n, out, tall, w, ld, HyperArgs{}, stream);
}

// the same launch carrying the optimizer's step-scalar role (srec_adam_hyper_multi: arguments as there) in one more workgroup:
// the end-of-backward sums of a captured training step and the scalars of the Adam kernels behind them need nothing of each other
extern "C" int srec_sum_slabs_multi_hyper(int np, const void* part, const int* R, const long* n, const void* out, const int* tall,
                                          const int* w, const int* ld, int nh, const void* counter, const void* cfg,
                                          const void* hyper, const int* tap_counter, const float* tap_src, float* tap_ring, int tap_n,
                                          const int* skip, void* stream) {
    HyperArgs h{};
    if (int rc = hyper_fill(nh, counter, cfg, hyper, tap_counter, tap_src, tap_ring, tap_n, skip, h)) return rc;
    return slab_sums(np, part, R, n, out, tall, w, ld, h, stream);
}

extern "C" int srec_sum_slabs_multi(int np, const void* part, const int* R, const long* n, const void* out, const int* tall,
                                    void* stream) {
    return srec_sum_slabs_multi_ld(np, part, R, n, out, tall, nullptr, nullptr, stream);
}
