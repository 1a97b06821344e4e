// Row-structured HBM-bound kernels: one 64-lane wavefront per embedding / node row,
// 16 B per lane (float4), so a d=256 fp32 row is exactly one coalesced 1 KiB wave access.
//
//   srec_gather_rows         nn.Embedding lookup                (srgnn.py:133, niser.py:133, lessr.py:168, msgifsr.py:247)
//   srec_scatter_add_sorted  its backward: deterministic segmented sum over the positions of each
//                            distinct item (collate emits the item -> positions CSR), no atomics
//   srec_renorm_rows         Embedding(max_norm=1) in-place renorm (lessr.py:126, msgifsr.py:162)
//   srec_row_invnorm         1/||E_v|| (x scale) for the cosine-scored models (niser.py:151, msgifsr.py:279)
//   srec_normalize_fwd/bwd   F.normalize / x.div(norm) on node features (niser.py:135,142,148; msgifsr.py:253,263,273)
//   srec_rownorm_project     chain rule of the catalog-row normalisation applied to the dense dE
//   srec_col_sum             column sums (bias gradients, fc_e gradients)
#include "common.h"

namespace {

constexpr int WPB = 4;   // waves (rows) per 256-thread block

// rng.p > 0: feature dropout fused into the lookup (msgifsr.py:247 dropout(embedding(iid))): element (row, c) of the
// OUTPUT is masked with index row * d + c
__global__ void gather_rows_kernel(const float* __restrict__ src, int ld_src, const int* __restrict__ idx,
                                   float* __restrict__ out, int ld_out, int n_cap, const int* __restrict__ dyn,
                                   int d, srec_rng rng) {
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_cap) return;
    const int n = dyn_count(dyn, n_cap);
    const int s = row < n ? idx[row] : -1;            // negative index (row owned by another shard) -> zero row
    const bool live = s >= 0;
    const bool drop = rng.p > 0.f;
    const unsigned key = drop ? srec_rng_key(rng) : 0u;
    const float sc = drop ? 1.f / (1.f - rng.p) : 1.f;
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = *reinterpret_cast<const float4*>(src + (size_t)s * ld_src + c);
            if (drop) {
                const unsigned i0 = (unsigned)row * (unsigned)d + (unsigned)c;
                v.x *= srec_keep(key, i0, rng.p, sc); v.y *= srec_keep(key, i0 + 1, rng.p, sc);
                v.z *= srec_keep(key, i0 + 2, rng.p, sc); v.w *= srec_keep(key, i0 + 3, rng.p, sc);
            }
        }
        *reinterpret_cast<float4*>(out + (size_t)row * ld_out + c) = v;
    }
}

// backward of a row pick out = x[idx] with STRICTLY ASCENDING idx (the last node of every session): dst[idx[j]] = g[j], every
// other row of dst [nrows, d] zero - one wavefront per pick writes its row and the zero rows between the previous pick and
// itself; the rows behind the last live pick (capacity padding included) are dealt round-robin.  No zero-fill launch.
__global__ void expand_rows_sorted_kernel(const float* __restrict__ g, int ld_g, const int* __restrict__ idx, int n_cap,
                                          const int* __restrict__ dyn, int nrows, int d, float* __restrict__ dst, int ld_dst) {
    const int j = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n_cap) return;
    const int n = dyn_count(dyn, n_cap);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n) {
        const int hi = idx[j], lo = j > 0 ? idx[j - 1] + 1 : 0;
        for (int r = lo; r < hi; ++r)
            for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<float4*>(dst + (size_t)r * ld_dst + c) = z;
        if (hi >= 0 && hi < nrows)
            for (int c = lane * 4; c < d; c += 256)
                *reinterpret_cast<float4*>(dst + (size_t)hi * ld_dst + c) = *reinterpret_cast<const float4*>(g + (size_t)j * ld_g + c);
    }
    const int tail = n > 0 ? idx[n - 1] + 1 : 0;
    for (int r = tail + j; r < nrows; r += n_cap)
        for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<float4*>(dst + (size_t)r * ld_dst + c) = z;
}

// dst[items[u], :] (+)= sum_{p in pos[ptr[u]:ptr[u+1]]} g[p, :]
__global__ void scatter_add_sorted_kernel(const float* __restrict__ g, int ld_g, const int* __restrict__ items,
                                          const int* __restrict__ ptr, const int* __restrict__ pos,
                                          float* __restrict__ dst, int ld_dst, int u_cap,
                                          const int* __restrict__ dyn, int d, int accumulate, srec_rng rng,
                                          const float* __restrict__ projW = nullptr, int ld_w = 0,
                                          float* __restrict__ radial = nullptr) {
    // radial (nullable; deferred row-normalisation projection, see srec_adam_rows_proj): radial[item] += <W_item, sum> - the
    // part of THIS gradient along the item's row, which the projection applied later to the whole buffer must not remove
    const int u = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= dyn_count(dyn, u_cap)) return;
    const int beg = ptr[u], end = ptr[u + 1], item = items[u];
    if (item < 0) return;                             // row owned by another shard
    float rdot = 0.f;
    if (rng.p > 0.f) {
        // backward of the fused lookup dropout: the gradient row of position p is masked with the forward's mask
        const unsigned key = srec_rng_key(rng);
        const float sc = 1.f / (1.f - rng.p);
        for (int c = lane * 4; c < d; c += 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            auto add = [&](int pp, const float4& v) {
                const unsigned i0 = (unsigned)pp * (unsigned)d + (unsigned)c;
                s.x += v.x * srec_keep(key, i0, rng.p, sc); s.y += v.y * srec_keep(key, i0 + 1, rng.p, sc);
                s.z += v.z * srec_keep(key, i0 + 2, rng.p, sc); s.w += v.w * srec_keep(key, i0 + 3, rng.p, sc);
            };
            int e = beg;
            for (; e + 4 <= end; e += 4) {            // four row loads in flight, added in position order
                const int p0 = pos[e], p1 = pos[e + 1], p2 = pos[e + 2], p3 = pos[e + 3];
                const float4 v0 = *reinterpret_cast<const float4*>(g + (size_t)p0 * ld_g + c);
                const float4 v1 = *reinterpret_cast<const float4*>(g + (size_t)p1 * ld_g + c);
                const float4 v2 = *reinterpret_cast<const float4*>(g + (size_t)p2 * ld_g + c);
                const float4 v3 = *reinterpret_cast<const float4*>(g + (size_t)p3 * ld_g + c);
                add(p0, v0); add(p1, v1); add(p2, v2); add(p3, v3);
            }
            for (; e < end; ++e) {
                const int pp = pos[e];
                add(pp, *reinterpret_cast<const float4*>(g + (size_t)pp * ld_g + c));
            }
            if (radial != nullptr) {
                const float4 wv = *reinterpret_cast<const float4*>(projW + (size_t)item * ld_w + c);
                rdot += wv.x * s.x + wv.y * s.y + wv.z * s.z + wv.w * s.w;
            }
            float4* o = reinterpret_cast<float4*>(dst + (size_t)item * ld_dst + c);
            if (accumulate) {
                const float4 t = *o;
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *o = s;
        }
        if (radial != nullptr) {
            rdot = wave_sum(rdot);
            if (lane == 0) radial[item] += rdot;
        }
        return;
    }
    for (int c = lane * 4; c < d; c += 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = beg;
        for (; e + 16 <= end; e += 16) {              // a Zipf-hot item has hundreds of pieces and ONE wavefront sums them: 16 row
            int pp[16]; float4 v[16];                 // loads in flight, added in order
#pragma unroll
            for (int k = 0; k < 16; ++k) pp[k] = pos[e + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4*>(g + (size_t)pp[k] * ld_g + c);
#pragma unroll
            for (int k = 0; k < 16; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
        }
        for (; e + 4 <= end; e += 4) {                // four row loads in flight (a popular item has dozens of pieces)
            const int p0 = pos[e], p1 = pos[e + 1], p2 = pos[e + 2], p3 = pos[e + 3];
            const float4 v0 = *reinterpret_cast<const float4*>(g + (size_t)p0 * ld_g + c);
            const float4 v1 = *reinterpret_cast<const float4*>(g + (size_t)p1 * ld_g + c);
            const float4 v2 = *reinterpret_cast<const float4*>(g + (size_t)p2 * ld_g + c);
            const float4 v3 = *reinterpret_cast<const float4*>(g + (size_t)p3 * ld_g + c);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; e < end; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(g + (size_t)pos[e] * ld_g + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (radial != nullptr) {
            const float4 wv = *reinterpret_cast<const float4*>(projW + (size_t)item * ld_w + c);
            rdot += wv.x * s.x + wv.y * s.y + wv.z * s.z + wv.w * s.w;
        }
        float4* o = reinterpret_cast<float4*>(dst + (size_t)item * ld_dst + c);
        if (accumulate) {
            const float4 t = *o;
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *o = s;
    }
    if (radial != nullptr) {
        rdot = wave_sum(rdot);
        if (lane == 0) radial[item] += rdot;
    }
}

__device__ __forceinline__ float row_sumsq(const float* __restrict__ p, int d, int lane) {
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    return wave_sum(s);
}

// rows = idx[0:n] (distinct!) or all rows when idx == NULL
__global__ void renorm_rows_kernel(float* __restrict__ W, int ld, const int* __restrict__ idx, int n_cap,
                                   const int* __restrict__ dyn, int d, float max_norm) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= dyn_count(dyn, n_cap)) return;
    float* p = W + (size_t)(idx != nullptr ? idx[i] : i) * ld;
    const float nrm = sqrtf(row_sumsq(p, d, lane));
    if (nrm > max_norm) {
        const float sc = max_norm / (nrm + 1e-7f);
        for (int c = lane * 4; c < d; c += 256) {
            float4 v = *reinterpret_cast<float4*>(p + c);
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            *reinterpret_cast<float4*>(p + c) = v;
        }
    }
}

// Embedding(max_norm) renorm of EVERY row (max_norm > 0) fused with the per-step bf16 operand copy of the table for the
// bf16 scoring kernels: one pass over the table instead of renorm + a separate conversion pass.  dst16 [>= n, Dp], zero
// in the columns d .. Dp.  One wave per row, the row is held in registers between the norm and the two writes (d <= 1024).
__global__ void renorm_rows_bf16_kernel(float* __restrict__ W, int ld, int n, int d, float max_norm,
                                        unsigned short* __restrict__ dst16, int Dp) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float* p = W + (size_t)i * ld;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j * 256;
        v[j] = c < d ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    if (max_norm > 0.f) {
        const float nrm = sqrtf(wave_sum(s));
        if (nrm > max_norm) {
            const float sc = max_norm / (nrm + 1e-7f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane * 4 + j * 256;
                v[j].x *= sc; v[j].y *= sc; v[j].z *= sc; v[j].w *= sc;
                if (c < d) *reinterpret_cast<float4*>(p + c) = v[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j * 256;
        if (c < Dp)
            *reinterpret_cast<uint2*>(dst16 + (size_t)i * Dp + c) =
                make_uint2(srec_pack_bf16(v[j].x, v[j].y), srec_pack_bf16(v[j].z, v[j].w));
    }
}

// eps_mode 0: 1/max(||x||, eps) (F.normalize)   1: 1/(||x|| + eps) (niser.py)
__device__ __forceinline__ float inv_norm(float sumsq, int eps_mode, float eps) {
    const float n = sqrtf(sumsq);
    return eps_mode == 0 ? 1.f / fmaxf(n, eps) : 1.f / (n + eps);
}

__global__ void row_invnorm_kernel(const float* __restrict__ W, int ld, int n, int d, int eps_mode, float eps,
                                   float scale, float* __restrict__ out) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float s = row_sumsq(W + (size_t)i * ld, d, lane);
    if (lane == 0) out[i] = scale * inv_norm(s, eps_mode, eps);
}

// dst16 (nullable): bf16 copy of the normalised rows, row stride Dp (the session-vector operand of the bf16 scoring kernels:
// the normalised session vector is what they multiply, so the conversion pass of its own disappears)
__global__ void normalize_fwd_kernel(const float* __restrict__ X, int ld_x, float* __restrict__ Y, int ld_y,
                                     float* __restrict__ inv, int n_cap, const int* __restrict__ dyn, int d,
                                     int eps_mode, float eps, unsigned short* __restrict__ dst16, int Dp) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_cap) return;
    const bool live = i < dyn_count(dyn, n_cap);
    float iv = 0.f;
    if (live) iv = inv_norm(row_sumsq(X + (size_t)i * ld_x, d, lane), eps_mode, eps);
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = *reinterpret_cast<const float4*>(X + (size_t)i * ld_x + c);
            v.x *= iv; v.y *= iv; v.z *= iv; v.w *= iv;
        }
        *reinterpret_cast<float4*>(Y + (size_t)i * ld_y + c) = v;
        if (dst16 != nullptr)
            *reinterpret_cast<uint2*>(dst16 + (size_t)i * Dp + c) = make_uint2(srec_pack_bf16(v.x, v.y), srec_pack_bf16(v.z, v.w));
    }
    if (lane == 0) inv[i] = iv;
}

// dX = inv * (dY - Y (Y . dY))
__global__ void normalize_bwd_kernel(const float* __restrict__ Y, int ld_y, const float* __restrict__ dY, int ld_dy,
                                     const float* __restrict__ inv, float* __restrict__ dX, int ld_dx, int n_cap,
                                     const int* __restrict__ dyn, int d) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_cap) return;
    const bool live = i < dyn_count(dyn, n_cap);
    float dot = 0.f;
    if (live) {
        for (int c = lane * 4; c < d; c += 256) {
            const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)i * ld_y + c);
            const float4 g = *reinterpret_cast<const float4*>(dY + (size_t)i * ld_dy + c);
            dot += y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
        }
        dot = wave_sum(dot);
    }
    const float iv = live ? inv[i] : 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)i * ld_y + c);
            const float4 g = *reinterpret_cast<const float4*>(dY + (size_t)i * ld_dy + c);
            o.x = iv * (g.x - y.x * dot); o.y = iv * (g.y - y.y * dot);
            o.z = iv * (g.z - y.z * dot); o.w = iv * (g.w - y.w * dot);
        }
        *reinterpret_cast<float4*>(dX + (size_t)i * ld_dx + c) = o;
    }
}

// G_v <- G_v - e_v (e_v . G_v),  e_v = W_v * inv_v   (inv = cs / scale)
__global__ void rownorm_project_kernel(const float* __restrict__ W, int ld_w, const float* __restrict__ cs,
                                       float inv_scale, float* __restrict__ G, int ld_g, int n, int d,
                                       float* __restrict__ radial = nullptr) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float iv = cs[i] * inv_scale;
    float dot = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)i * ld_w + c);
        const float4 g = *reinterpret_cast<const float4*>(G + (size_t)i * ld_g + c);
        dot += w.x * g.x + w.y * g.y + w.z * g.z + w.w * g.w;
    }
    dot = wave_sum(dot);
    if (radial != nullptr) {                           // the part of G that was added after the scoring gradient stays
        dot -= radial[i];
        if (lane == 0) radial[i] = 0.f;
    }
    dot *= iv * iv;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)i * ld_w + c);
        float4 g = *reinterpret_cast<float4*>(G + (size_t)i * ld_g + c);
        g.x -= w.x * dot; g.y -= w.y * dot; g.z -= w.z * dot; g.w -= w.w * dot;
        *reinterpret_cast<float4*>(G + (size_t)i * ld_g + c) = g;
    }
}

// Column sums, two deterministic stages: grid (column blocks x NCHUNK row chunks) writes partial sums,
// a second tiny kernel adds the NCHUNK partials.  (w != NULL: per-(row, head) weights, head = col / D.)
constexpr int NCHUNK = 32;

__global__ void col_sum_part_kernel(const float* __restrict__ X, int ld, const float* __restrict__ wgt, int H, int D,
                                    int n_cap, const int* __restrict__ dyn, int ncol, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int n = dyn_count(dyn, n_cap);
    const int per = (n + NCHUNK - 1) / NCHUNK;
    const int r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    float s = 0.f;
    if (c < ncol) {
        if (wgt != nullptr) {
            const int h = c / D;
            for (int r = r0 + rg; r < r1; r += 4) s += wgt[(size_t)r * H + h] * X[(size_t)r * ld + c];
        } else {
            for (int r = r0 + rg; r < r1; r += 4) s += X[(size_t)r * ld + c];
        }
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < ncol)
        part[(size_t)blockIdx.y * ncol + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// unweighted column sums, vector path: a thread owns 4 adjacent columns (float4) and keeps 4 row loads in flight; the rows
// are split into NCHUNK_V chunks x 4 row groups so even a 768-column matrix gets ~400 workgroups (the scalar kernel above
// spent ~60 dependent 4-byte loads per thread: 12-18 us for 8 MB)
constexpr int NCHUNK_V = 128;
__global__ __launch_bounds__(256) void col_sum_part_v4_kernel(const float* __restrict__ X, int ld, int n_cap,
                                                              const int* __restrict__ dyn, int ncol, float* __restrict__ part) {
    __shared__ float4 red[3][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int n = dyn_count(dyn, n_cap);
    const int per = (n + NCHUNK_V - 1) / NCHUNK_V;
    const int r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < ncol) {
        int r = r0 + rg;
        for (; r + 12 < r1; r += 16) {
            float4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const float4*>(X + (size_t)(r + 4 * e) * ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s.x += v[e].x; s.y += v[e].y; s.z += v[e].z; s.w += v[e].w; }
        }
        for (; r < r1; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(X + (size_t)r * ld + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (rg > 0) red[rg - 1][lane] = s;
    __syncthreads();
    if (rg == 0 && c < ncol) {
#pragma unroll
        for (int g = 0; g < 3; ++g) { s.x += red[g][lane].x; s.y += red[g][lane].y; s.z += red[g][lane].z; s.w += red[g][lane].w; }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * ncol + c) = s;
    }
}

// 64 columns per workgroup, the chunks split over the 4 waves (fixed order: deterministic), 4 loads in flight
__global__ __launch_bounds__(256) void col_sum_final_v_kernel(const float* __restrict__ part, int nchunk, int ncol,
                                                              float* __restrict__ out, int accumulate) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < ncol) {
        int k = g;
        for (; k + 12 < nchunk; k += 16) {
            s0 += part[(size_t)k * ncol + c]; s1 += part[(size_t)(k + 4) * ncol + c];
            s2 += part[(size_t)(k + 8) * ncol + c]; s3 += part[(size_t)(k + 12) * ncol + c];
        }
        for (; k < nchunk; k += 4) s0 += part[(size_t)k * ncol + c];
    }
    const float s = (s0 + s1) + (s2 + s3);
    if (g > 0) red[g - 1][lane] = s;
    __syncthreads();
    if (g == 0 && c < ncol) {
        const float t = (s + red[0][lane]) + (red[1][lane] + red[2][lane]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ void col_sum_final_kernel(const float* __restrict__ part, int ncol, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    float s = 0.f;
    for (int k = 0; k < NCHUNK; ++k) s += part[(size_t)k * ncol + c];
    out[c] = accumulate ? out[c] + s : s;
}

// out[i] = idx[i] - lo when idx[i] is a row of this shard ([lo, lo + n_loc)), else -1 (also for the -1 padding)
template <typename I>
__global__ void localize_idx_kernel(const I* __restrict__ idx, long n, long long lo, int n_loc, int* __restrict__ out,
                                    float* __restrict__ zero = nullptr, long zero_from = 0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (zero != nullptr && i >= zero_from) zero[i - zero_from] = 0.f;
    const long long v = (long long)idx[i], r = v - lo;
    out[i] = (v >= 0 && r >= 0 && r < n_loc) ? (int)r : -1;
}

// st [w][2][B]: per-shard (log-sum-exp, label logit) of every session -> global lse[b] = logsumexp_r st[r][0][b],
// lab[b] = sum_r st[r][1][b] (non-zero on the one shard that owns the label), loss = mean over the LIVE sessions of
// (lse - lab).  lab_all (nullable) = the gathered global labels: a session with label < 0 is capacity padding of its
// rank's batch (dead): it is left out of the mean and gets weight 0 in gw (nullable) = d loss / d (lse_b - lab_b), i.e.
// 1 / n_live for the live sessions - the per-session coefficients the backward kernels take as ga / gc.  One workgroup.
// (1024 threads, the per-rank statistics of a session fetched side by side: at 8 ranks x 4 096 sessions the first version - 256
//  threads, one dependent load after the other - took 53 us of the rank's step, profiles/r06_rank8_weak_breakdown.txt)
template <typename I>
__global__ __launch_bounds__(1024) void merge_stats_kernel(const float* __restrict__ st, int w, int B, const I* __restrict__ lab_all,
                                                          float* __restrict__ lse, float* __restrict__ lab, float* __restrict__ loss,
                                                          float* __restrict__ gw) {
    __shared__ float red[16];
    __shared__ int cnt[16];
    const int nw = blockDim.x >> 6;
    int c = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) c += (lab_all == nullptr || lab_all[b] >= 0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    int n_live = 0;
    for (int i = 0; i < nw; ++i) n_live += cnt[i];
    const float inv = 1.f / (float)(n_live > 0 ? n_live : 1);
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const bool live = lab_all == nullptr || lab_all[b] >= 0;
        float m = -INFINITY, t = 0.f, l = 0.f;
        for (int r0 = 0; r0 < w; r0 += 8) {                              // 8 ranks' (lse, label logit) pairs in flight
            float sv[8], tv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = min(r0 + k, w - 1);
                sv[k] = st[((size_t)r * 2) * B + b];
                tv[k] = st[((size_t)r * 2 + 1) * B + b];
            }
            float mb = m;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (r0 + k < w) { mb = fmaxf(mb, sv[k]); t += tv[k]; }
            l *= expf(m - mb);                                           // (m = -inf in the first block: l = 0 stays 0)
            if (m == -INFINITY) l = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (r0 + k < w) l += expf(sv[k] - mb);
            m = mb;
        }
        const float v = m + logf(l);
        lse[b] = v;
        lab[b] = t;
        if (live) acc += v - t;
        if (gw != nullptr) gw[b] = live ? inv : 0.f;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nw; ++i) s += red[i];
        loss[0] = s * inv;
    }
}

// inv[pos[e]] = u for e in [ptr[u], ptr[u+1]); one wavefront per item.  inv is pre-filled with -1 by the first pass.
__global__ void inverse_index_kernel(const int* __restrict__ ptr, const int* __restrict__ pos, int U, int n,
                                     int* __restrict__ inv) {
    const int u = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= U) return;
    const int beg = ptr[u], end = ptr[u + 1];
    for (int e = beg + lane; e < end; e += 64) {
        const int p = pos[e];
        if (p >= 0 && p < n) inv[p] = u;
    }
}
__global__ void fill_int_kernel(int* __restrict__ p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline bool bad_row_args(int d, int ld) { return d <= 0 || (d & 3) || (ld & 3); }

}  // namespace

extern "C" int srec_gather_rows(const float* src, int ld_src, const int* idx, float* out, int ld_out, int n_cap,
                                const int* dyn, int d, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad_row_args(d, ld_src) || (ld_out & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx,
                       out, ld_out, n_cap, dyn, d, srec_rng{0u, nullptr, 0u, 0.f});
    SREC_LAUNCH_CHECK();
    return 0;
}

// ... with feature dropout (probability p, mask key from (seed, *counter, salt): common.h) fused into the written rows
extern "C" int srec_gather_rows_drop(const float* src, int ld_src, const int* idx, float* out, int ld_out, int n_cap,
                                     const int* dyn, int d, float p, int seed, const int* counter, int salt,
                                     void* stream) {
    if (n_cap <= 0) return 0;
    if (bad_row_args(d, ld_src) || (ld_out & 3) || p < 0.f || p >= 1.f || (long)n_cap * d > 0xffffffffL) return SREC_BAD_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx,
                       out, ld_out, n_cap, dyn, d, srec_rng{(unsigned)seed, counter, (unsigned)salt, p});
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_expand_rows_sorted(const float* g, int ld_g, const int* idx, int n_cap, const int* dyn, int nrows, int d,
                                       float* dst, int ld_dst, void* stream) {
    if (n_cap <= 0 || nrows <= 0) return 0;
    if (bad_row_args(d, ld_g) || (ld_dst & 3) || idx == nullptr) return SREC_BAD_ARG;
    hipLaunchKernelGGL(expand_rows_sorted_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, g, ld_g, idx, n_cap,
                       dyn, nrows, d, dst, ld_dst);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_scatter_add_sorted(const float* g, int ld_g, const int* items, const int* ptr, const int* pos,
                                       float* dst, int ld_dst, int u_cap, const int* dyn, int d, int accumulate,
                                       void* stream) {
    if (u_cap <= 0) return 0;
    if (bad_row_args(d, ld_g) || (ld_dst & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(scatter_add_sorted_kernel, dim3(cdiv(u_cap, WPB)), dim3(256), 0, (hipStream_t)stream, g, ld_g,
                       items, ptr, pos, dst, ld_dst, u_cap, dyn, d, accumulate, srec_rng{0u, nullptr, 0u, 0.f});
    SREC_LAUNCH_CHECK();
    return 0;
}

// ... reading g through the dropout mask of srec_gather_rows_drop (same p / seed / counter / salt; position index = pos)
extern "C" int srec_scatter_add_sorted_drop(const float* g, int ld_g, const int* items, const int* ptr, const int* pos,
                                            float* dst, int ld_dst, int u_cap, const int* dyn, int d, int accumulate,
                                            float p, int seed, const int* counter, int salt, void* stream) {
    if (u_cap <= 0) return 0;
    if (bad_row_args(d, ld_g) || (ld_dst & 3) || p < 0.f || p >= 1.f || ld_g != d) return SREC_BAD_ARG;
    hipLaunchKernelGGL(scatter_add_sorted_kernel, dim3(cdiv(u_cap, WPB)), dim3(256), 0, (hipStream_t)stream, g, ld_g,
                       items, ptr, pos, dst, ld_dst, u_cap, dyn, d, accumulate,
                       srec_rng{(unsigned)seed, counter, (unsigned)salt, p});
    SREC_LAUNCH_CHECK();
    return 0;
}

// ---- row-sharded lookup backward: the per-item gradient rows of ALL ranks into this shard's dense gradient in ONE launch.
// rows [w * ucap, d] = the all-gathered rows, rel [w * ucap] = local row of every request (-1: another shard's / padding), ids
// [w * ucap] = the requests' global item ids (ascending inside each rank's list, -1 padding behind them).  The same item may sit
// in several ranks' lists (Zipf-popular items in all of them): the LOWEST rank holding it owns the sum and adds the ranks' rows
// to dst[item] in rank order - the order, and so the bits, of w launches of srec_scatter_add_sorted rank by rank (what this
// replaces: w x ~4.6 us of launch floor in the rank step, profiles/r06_rank8_weak_breakdown.txt).  One wavefront per request:
// lane r < w binary-searches rank r's list for the item.
namespace {
__global__ void add_rows_ranks_kernel(const float* __restrict__ rows, int d, const int* __restrict__ rel,
                                      const int* __restrict__ ids, int w, int ucap, float* __restrict__ dst, int ld_dst,
                                      const float* __restrict__ projW, int ld_w, float* __restrict__ radial) {
    const int q = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= w * ucap) return;
    const int item = rel[q];
    if (item < 0) return;
    const int r_me = q / ucap, gid = ids[q];
    // slot of gid in rank `lane`'s list (ascending, -1 = +infinity), or -1
    int found = -1;
    if (lane < w) {
        if (lane == r_me) found = q;
        else {
            const int* L = ids + (size_t)lane * ucap;
            int lo = 0, hi = ucap;
            while (lo < hi) {                               // first position whose id is >= gid (padding counts as larger)
                const int mid = (lo + hi) >> 1;
                const int v = L[mid];
                if (v >= 0 && v < gid) lo = mid + 1; else hi = mid;
            }
            if (lo < ucap && L[lo] == gid) found = lane * ucap + lo;
        }
    }
    const unsigned long long have = __ballot(found >= 0);
    if ((have & ((1ull << r_me) - 1ull)) != 0ull) return;   // a lower rank holds the item too: it owns the sum
    float rd = 0.f;
    const bool proj = radial != nullptr;
    if (proj && lane == 0) rd = radial[item];
    float dots[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dots[k] = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        float4 s = *reinterpret_cast<const float4*>(dst + (size_t)item * ld_dst + c);
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (proj) wv = *reinterpret_cast<const float4*>(projW + (size_t)item * ld_w + c);
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {                      // the ranks' rows in flight together, added in rank order
            const int slot = __shfl(found, k, 64);
            v[k] = (k < w && slot >= 0) ? *reinterpret_cast<const float4*>(rows + (size_t)slot * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < w && ((have >> k) & 1ull)) {
                s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w;
                dots[k] += wv.x * v[k].x + wv.y * v[k].y + wv.z * v[k].z + wv.w * v[k].w;
            }
        }
        *reinterpret_cast<float4*>(dst + (size_t)item * ld_dst + c) = s;
    }
    if (proj) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < w && ((have >> k) & 1ull)) {            // (wave-uniform)
                const float t = wave_sum(dots[k]);
                rd += t;                                    // rank order, one wave sum per rank: as the w launches did
            }
        }
        if (lane == 0) radial[item] = rd;
    }
}
}  // namespace

extern "C" int srec_add_rows_ranks(const float* rows, int d, const int* rel, const int* ids, int w, int ucap, float* dst,
                                   int ld_dst, const float* projW, int ld_w, float* radial, void* stream) {
    if (w <= 0 || ucap <= 0) return 0;
    if (w > 16 || d <= 0 || (d & 3) || (ld_dst & 3) || rows == nullptr || rel == nullptr || ids == nullptr || dst == nullptr)
        return SREC_BAD_ARG;
    if (radial != nullptr && (projW == nullptr || (ld_w & 3))) return SREC_BAD_ARG;
    hipLaunchKernelGGL(add_rows_ranks_kernel, dim3(cdiv(w * ucap, WPB)), dim3(256), 0, (hipStream_t)stream, rows, d, rel, ids, w,
                       ucap, dst, ld_dst, projW, ld_w, radial);
    SREC_LAUNCH_CHECK();
    return 0;
}

// the general form: dropout mask (p = 0: none) and the radial side sum (radial = NULL: none) of the deferred projection
extern "C" int srec_scatter_add_sorted_ex(const float* g, int ld_g, const int* items, const int* ptr, const int* pos,
                                          float* dst, int ld_dst, int u_cap, const int* dyn, int d, int accumulate,
                                          float p, int seed, const int* counter, int salt, const float* projW, int ld_w,
                                          float* radial, void* stream) {
    if (u_cap <= 0) return 0;
    if (bad_row_args(d, ld_g) || (ld_dst & 3) || p < 0.f || p >= 1.f || (p > 0.f && ld_g != d)) return SREC_BAD_ARG;
    if (radial != nullptr && (projW == nullptr || (ld_w & 3))) return SREC_BAD_ARG;
    hipLaunchKernelGGL(scatter_add_sorted_kernel, dim3(cdiv(u_cap, WPB)), dim3(256), 0, (hipStream_t)stream, g, ld_g,
                       items, ptr, pos, dst, ld_dst, u_cap, dyn, d, accumulate,
                       srec_rng{(unsigned)seed, counter, (unsigned)salt, p}, projW, ld_w, radial);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_renorm_rows(float* W, int ld, const int* idx, int n_cap, const int* dyn, int d, float max_norm,
                                void* stream) {
    if (n_cap <= 0) return 0;
    if (bad_row_args(d, ld)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(renorm_rows_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, W, ld, idx, n_cap,
                       dyn, d, max_norm);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_renorm_rows_bf16(float* W, int ld, int n, int d, float max_norm, void* dst16, int Dp, void* stream) {
    if (n <= 0) return 0;
    if (bad_row_args(d, ld) || d > 1024 || Dp < d || (Dp & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(renorm_rows_bf16_kernel, dim3(cdiv(n, WPB)), dim3(256), 0, (hipStream_t)stream, W, ld, n, d, max_norm,
                       (unsigned short*)dst16, Dp);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_row_invnorm(const float* W, int ld, int n, int d, int eps_mode, float eps, float scale, float* out,
                                void* stream) {
    if (n <= 0) return 0;
    if (bad_row_args(d, ld)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(row_invnorm_kernel, dim3(cdiv(n, WPB)), dim3(256), 0, (hipStream_t)stream, W, ld, n, d, eps_mode,
                       eps, scale, out);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_normalize_fwd(const float* X, int ld_x, float* Y, int ld_y, float* inv, int n_cap, const int* dyn,
                                  int d, int eps_mode, float eps, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad_row_args(d, ld_x) || (ld_y & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(normalize_fwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, X, ld_x, Y, ld_y,
                       inv, n_cap, dyn, d, eps_mode, eps, (unsigned short*)nullptr, 0);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_normalize_fwd_bf16(const float* X, int ld_x, float* Y, int ld_y, float* inv, int n_cap, const int* dyn,
                                       int d, int eps_mode, float eps, void* dst16, int Dp, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad_row_args(d, ld_x) || (ld_y & 3) || dst16 == nullptr || Dp < d || (Dp & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(normalize_fwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, X, ld_x, Y, ld_y,
                       inv, n_cap, dyn, d, eps_mode, eps, (unsigned short*)dst16, Dp);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_normalize_bwd(const float* Y, int ld_y, const float* dY, int ld_dy, const float* inv, float* dX,
                                  int ld_dx, int n_cap, const int* dyn, int d, void* stream) {
    if (n_cap <= 0) return 0;
    if (bad_row_args(d, ld_y) || (ld_dy & 3) || (ld_dx & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(cdiv(n_cap, WPB)), dim3(256), 0, (hipStream_t)stream, Y, ld_y, dY,
                       ld_dy, inv, dX, ld_dx, n_cap, dyn, d);
    SREC_LAUNCH_CHECK();
    return 0;
}

namespace {
// normalisation of up to 4 row blocks of different tensors into / out of ONE stacked matrix (the node features of
// MSGIFSR's orders: msgifsr.py:253 per order, concatenated for the batched MSHGNN layer): block p = rows [row0, row0 + n)
// of the stacked side, its own tensor X[p] / dX[p] on the other side, its own live count
struct NormGroup {
    const float* X[4]; float* dX[4];
    const int* dyn[4];
    int ld[4], n[4], row0[4];
    int np;
};
__global__ void normalize_group_fwd_kernel(NormGroup g, float* __restrict__ Y, int ld_y, float* __restrict__ inv, int d,
                                           int eps_mode, float eps) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int p = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < g.np && i >= g.row0[q]) p = q;
    const int li = i - g.row0[p];
    if (li >= g.n[p]) return;
    const bool live = li < dyn_count(g.dyn[p], g.n[p]);
    const float* x = g.X[p] + (size_t)li * g.ld[p];
    float iv = 0.f;
    if (live) iv = inv_norm(row_sumsq(x, d, lane), eps_mode, eps);
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = *reinterpret_cast<const float4*>(x + c);
            v.x *= iv; v.y *= iv; v.z *= iv; v.w *= iv;
        }
        *reinterpret_cast<float4*>(Y + (size_t)i * ld_y + c) = v;
    }
    if (lane == 0) inv[i] = iv;
}
__global__ void normalize_group_bwd_kernel(NormGroup g, const float* __restrict__ Y, int ld_y, const float* __restrict__ dY,
                                           int ld_dy, const float* __restrict__ inv, int d) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int p = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < g.np && i >= g.row0[q]) p = q;
    const int li = i - g.row0[p];
    if (li >= g.n[p]) return;
    const bool live = li < dyn_count(g.dyn[p], g.n[p]);
    float* dx = g.dX[p] + (size_t)li * g.ld[p];
    float dot = 0.f;
    if (live)
        for (int c = lane * 4; c < d; c += 256) {
            const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)i * ld_y + c);
            const float4 gg = *reinterpret_cast<const float4*>(dY + (size_t)i * ld_dy + c);
            dot += y.x * gg.x + y.y * gg.y + y.z * gg.z + y.w * gg.w;
        }
    dot = wave_sum(dot);
    const float iv = live ? inv[i] : 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)i * ld_y + c);
            const float4 gg = *reinterpret_cast<const float4*>(dY + (size_t)i * ld_dy + c);
            o = make_float4(iv * (gg.x - y.x * dot), iv * (gg.y - y.y * dot), iv * (gg.z - y.z * dot), iv * (gg.w - y.w * dot));
        }
        *reinterpret_cast<float4*>(dx + c) = o;
    }
}
int fill_norm_group(NormGroup& g, int np, const void* X, const int* ld, const int* n, const void* dyn, int d, int* total) {
    if (np <= 0 || np > 4 || X == nullptr || ld == nullptr || n == nullptr) return SREC_BAD_ARG;
    g.np = np;
    int r = 0;
    for (int p = 0; p < np; ++p) {
        g.X[p] = ((const float* const*)X)[p];
        g.dX[p] = ((float* const*)X)[p];
        g.dyn[p] = dyn != nullptr ? ((const int* const*)dyn)[p] : nullptr;
        g.ld[p] = ld[p]; g.n[p] = n[p]; g.row0[p] = r;
        if (g.X[p] == nullptr || n[p] <= 0 || bad_row_args(d, ld[p])) return SREC_BAD_ARG;
        r += n[p];
    }
    *total = r;
    return 0;
}
}  // namespace

// Y [sum n_p, d] (stacked, ld_y) = row-normalised X_p [n_p, d]; inv [sum n_p].  X / dyn: HOST arrays of np <= 4 device pointers
// (dyn entries nullable), ld / n: HOST int arrays.  Rows past a block's live count are written as zeros.
extern "C" int srec_normalize_group_fwd(int np, const void* X, const int* ld, const int* n, const void* dyn, float* Y,
                                        int ld_y, float* inv, int d, int eps_mode, float eps, void* stream) {
    NormGroup g{};
    int total = 0;
    if (int rc = fill_norm_group(g, np, X, ld, n, dyn, d, &total)) return rc;
    if (ld_y & 3) return SREC_BAD_ARG;
    hipLaunchKernelGGL(normalize_group_fwd_kernel, dim3(cdiv(total, WPB)), dim3(256), 0, (hipStream_t)stream, g, Y, ld_y, inv,
                       d, eps_mode, eps);
    SREC_LAUNCH_CHECK();
    return 0;
}

// dX_p [n_p, d] = inv (dY - Y (Y . dY)) for the rows of block p of the stacked Y / dY
extern "C" int srec_normalize_group_bwd(int np, const void* dX, const int* ld, const int* n, const void* dyn, const float* Y,
                                        int ld_y, const float* dY, int ld_dy, const float* inv, int d, void* stream) {
    NormGroup g{};
    int total = 0;
    if (int rc = fill_norm_group(g, np, dX, ld, n, dyn, d, &total)) return rc;
    if ((ld_y & 3) || (ld_dy & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(normalize_group_bwd_kernel, dim3(cdiv(total, WPB)), dim3(256), 0, (hipStream_t)stream, g, Y, ld_y, dY,
                       ld_dy, inv, d);
    SREC_LAUNCH_CHECK();
    return 0;
}

namespace {
// out[r, :] = [a[r, :da] | b[r, :db]]
__global__ void cat_cols_kernel(const float* __restrict__ a, int lda, int da, const float* __restrict__ b, int ldb, int db,
                                int n, float* __restrict__ out) {
    const int w = da + db;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long)n * w) return;
    const int r = (int)(i / w), c = (int)(i % w);
    const float4 v = c < da ? *reinterpret_cast<const float4*>(a + (size_t)r * lda + c)
                            : *reinterpret_cast<const float4*>(b + (size_t)r * ldb + (c - da));
    *reinterpret_cast<float4*>(out + i) = v;
}
}  // namespace

// out [n, da + db] = [a | b] (feature-axis concatenation feeding fc_sr: srgnn.py:143, msgifsr.py:270-272); da, db % 4 == 0
extern "C" int srec_cat_cols(const float* a, int lda, int da, const float* b, int ldb, int db, int n, float* out,
                             void* stream) {
    if (n <= 0) return 0;
    if ((da & 3) || (db & 3) || (lda & 3) || (ldb & 3) || da <= 0 || db <= 0) return SREC_BAD_ARG;
    const long items = (long)n * (da + db) / 4;
    hipLaunchKernelGGL(cat_cols_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, lda, da, b,
                       ldb, db, n, out);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_rownorm_project(const float* W, int ld_w, const float* cs, float inv_scale, float* G, int ld_g,
                                    int n, int d, void* stream) {
    if (n <= 0) return 0;
    if (bad_row_args(d, ld_w) || (ld_g & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(rownorm_project_kernel, dim3(cdiv(n, WPB)), dim3(256), 0, (hipStream_t)stream, W, ld_w, cs,
                       inv_scale, G, ld_g, n, d, (float*)nullptr);
    SREC_LAUNCH_CHECK();
    return 0;
}

// the deferred form: G holds scoring gradient + later additions whose radial sums sit in radial [n] (cleared here)
extern "C" int srec_rownorm_project_radial(const float* W, int ld_w, const float* cs, float inv_scale, float* G, int ld_g,
                                           int n, int d, float* radial, void* stream) {
    if (n <= 0) return 0;
    if (bad_row_args(d, ld_w) || (ld_g & 3)) return SREC_BAD_ARG;
    hipLaunchKernelGGL(rownorm_project_kernel, dim3(cdiv(n, WPB)), dim3(256), 0, (hipStream_t)stream, W, ld_w, cs,
                       inv_scale, G, ld_g, n, d, radial);
    SREC_LAUNCH_CHECK();
    return 0;
}

// ws: NCHUNK(32) * ncol floats of scratch.  wgt (nullable) [n, H] weights per (row, head), head = col / D.
extern "C" int srec_col_sum(const float* X, int ld, const float* wgt, int H, int D, int n_cap, const int* dyn, int ncol,
                            float* out, int accumulate, float* ws, void* stream) {
    if (ncol <= 0) return 0;
    if (ws == nullptr) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (wgt == nullptr && (ncol & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)ws & 15) == 0) {
        hipLaunchKernelGGL(col_sum_part_v4_kernel, dim3(cdiv(ncol, 256), NCHUNK_V), dim3(256), 0, st, X, ld, n_cap, dyn, ncol,
                           ws);
        hipLaunchKernelGGL(col_sum_final_v_kernel, dim3(cdiv(ncol, 64)), dim3(256), 0, st, ws, NCHUNK_V, ncol, out,
                           accumulate);
        SREC_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(col_sum_part_kernel, dim3(cdiv(ncol, 64), NCHUNK), dim3(256), 0, st, X, ld, wgt, H, D, n_cap, dyn,
                       ncol, ws);
    hipLaunchKernelGGL(col_sum_final_kernel, dim3(cdiv(ncol, 256)), dim3(256), 0, st, ws, ncol, out, accumulate);
    SREC_LAUNCH_CHECK();
    return 0;
}

// Row-sharded table (dist.py): global item ids (int64, -1 = padding) -> row of THIS shard or -1, in one launch
// (replaces a subtract / three compares / two ands / where / cast chain per exchange).
extern "C" int srec_localize_idx(const long long* idx, long n, long lo, int n_loc, int* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(localize_idx_kernel<long long>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, n,
                       (long long)lo, n_loc, out);
    SREC_LAUNCH_CHECK();
    return 0;
}

// the same for int32 ids (the request lists of capacity-padded batches travel as the int32 words they are collated as);
// zero (nullable) [n - zero_from] floats cleared by the same launch: the label-logit array of the sharded scoring forward, whose
// kernels write an entry only where they meet the label - no fill launch of its own.  zero_from: the list is (requested ids |
// labels) of ONE exchange, localised together; the label part starts there
extern "C" int srec_localize_idx32(const int* idx, long n, long lo, int n_loc, int* out, float* zero, long zero_from, void* stream) {
    if (n <= 0) return 0;
    if (zero_from < 0 || zero_from > n) return SREC_BAD_ARG;
    hipLaunchKernelGGL(localize_idx_kernel<int>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, n,
                       (long long)lo, n_loc, out, zero, zero_from);
    SREC_LAUNCH_CHECK();
    return 0;
}

// Row-sharded scoring (dist.py): merge the per-shard soft-max statistics gathered from the w ranks.  st [w, 2, B].
extern "C" int srec_merge_stats(const float* st, int w, int B, const long long* lab_all, float* lse, float* lab,
                                float* loss, float* gw, void* stream) {
    if (w <= 0 || B <= 0) return SREC_BAD_ARG;
    hipLaunchKernelGGL(merge_stats_kernel<long long>, dim3(1), dim3(B > 1024 ? 1024 : 256), 0, (hipStream_t)stream, st, w, B, lab_all, lse,
                       lab, loss, gw);
    SREC_LAUNCH_CHECK();
    return 0;
}

// ... with int32 labels
extern "C" int srec_merge_stats32(const float* st, int w, int B, const int* lab_all, float* lse, float* lab, float* loss,
                                  float* gw, void* stream) {
    if (w <= 0 || B <= 0) return SREC_BAD_ARG;
    hipLaunchKernelGGL(merge_stats_kernel<int>, dim3(1), dim3(B > 1024 ? 1024 : 256), 0, (hipStream_t)stream, st, w, B, lab_all, lse, lab, loss,
                       gw);
    SREC_LAUNCH_CHECK();
    return 0;
}

// inv [n] (int32): inv[p] = u for the positions p = pos[ptr[u] .. ptr[u+1]) of item u (u < U), -1 for unclaimed positions
// (capacity padding).  The CSR (ptr, pos) is the uniq_ptr / uniq_pos pair of a FlatBatch.
extern "C" int srec_inverse_index(const int* ptr, const int* pos, int U, int n, int* inv, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, inv, n, -1);
    if (U > 0) hipLaunchKernelGGL(inverse_index_kernel, dim3(cdiv(U, WPB)), dim3(256), 0, st, ptr, pos, U, n, inv);
    SREC_LAUNCH_CHECK();
    return 0;
}

// ---- MSGIFSR after its last MSHGNN layer (msgifsr.py:260-264, 131-147): L2-normalise every node row, lay the nodes out
// per session [s1 | s2 | ...] (the read-out's concatenation, cat_perm) and pick each order's last node - ONE launch instead
// of normalize + permutation gather + pick gather,
// and ONE launch for the three backward kernels (inverse-permutation gather, pick scatter-add, normalize backward).
namespace {
struct NppArgs {
    const float* x; int ld_x;                 // stacked rows [NTs, D] (order-1 rows first)
    const int* perm;                          // [n_cap] concatenated row -> stacked row (-1: capacity padding)
    const int* dyn_t;                         // live concatenated rows
    int n_cap, D, eps_mode; float eps;
    float* allf; float* invr;                 // [n_cap, D] normalised rows in concatenated order, their 1 / norm
    int npick, B; const int* dyn_b;
    const int* pick[4];                       // [B] stacked row of each session's last node of that order
    float* pout[4]; int ld_p[4];              // [B, D] (row stride ld_p)
};

__device__ __forceinline__ void npp_row(const float* __restrict__ xr, bool live, int D, int eps_mode, float eps, int lane,
                                        float* __restrict__ out, float* __restrict__ inv_out) {
    float iv = 0.f;
    if (live) iv = inv_norm(row_sumsq(xr, D, lane), eps_mode, eps);
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = *reinterpret_cast<const float4*>(xr + c);
            v.x *= iv; v.y *= iv; v.z *= iv; v.w *= iv;
        }
        *reinterpret_cast<float4*>(out + c) = v;
    }
    if (inv_out != nullptr && lane == 0) *inv_out = iv;
}

// one wavefront per output row: rows [0, n_cap) of allf, then npick x B pick rows
__global__ void norm_perm_pick_fwd_kernel(NppArgs a) {
    const int i = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i < a.n_cap) {
        const int src = i < dyn_count(a.dyn_t, a.n_cap) ? a.perm[i] : -1;
        npp_row(a.x + (size_t)(src >= 0 ? src : 0) * a.ld_x, src >= 0, a.D, a.eps_mode, a.eps, lane, a.allf + (size_t)i * a.D,
                a.invr + i);
        return;
    }
    const int j = i - a.n_cap, k = j / a.B, b = j - k * a.B;
    if (k >= a.npick) return;
    const int src = b < dyn_count(a.dyn_b, a.B) ? a.pick[k][b] : -1;
    npp_row(a.x + (size_t)(src >= 0 ? src : 0) * a.ld_x, src >= 0, a.D, a.eps_mode, a.eps, lane, a.pout[k] + (size_t)b * a.ld_p[k],
            nullptr);
}

struct NppBwdArgs {
    const float* allf; const float* invr; const float* g_allf; int ld_g;      // concatenated order
    const int* perm; const int* cat_seg; int B; const int* dyn_b; int D;
    int npick; const int* pick[4]; const float* g_pick[4]; int ld_gp[4];     // pick k: stacked row of session b, its gradient row
    float* dx; int ld_dx;                                                       // stacked order, every row written
    int nt; int row0[4], ncap[4]; const int* dyn_n[4];                          // node types: padded stacked rows are zeroed
    int n_rows;                                                                 // stacked rows in total
};

// one wavefront per concatenated row r (workgroups [0, ceil(n_cap / 4))): the row's session is found by a binary search in
// the LDS-staged cat_seg (one load round trip per workgroup), the total gradient of the normalised row = the read-out's
// gradient of that row (+ the pick gradient when the row is its session's last node of an order) goes through the
// normalisation and is written to the row's place in the stacked matrix.  (A workgroup per SESSION walking its rows one
// after the other measured 14 us whatever the batch: a chain of dependent loads per row.)  Workgroups behind them zero the
// capacity padding of the stacked matrix (64 rows each).
constexpr int NPP_SEG = 2048;
__global__ void norm_perm_pick_bwd_kernel(NppBwdArgs a, int n_cap, int row_blocks) {
    __shared__ int sseg[NPP_SEG];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= row_blocks) {
        const int r0 = ((int)blockIdx.x - row_blocks) * 64;
        for (int rr = w; rr < 64; rr += WPB) {
            const int r = r0 + rr;
            if (r >= a.n_rows) break;
            int t = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (q < a.nt && r >= a.row0[q]) t = q;
            if (r - a.row0[t] < dyn_count(a.dyn_n[t], a.ncap[t])) continue;          // live row: a row wavefront owns it
            for (int c = lane * 4; c < a.D; c += 256)
                *reinterpret_cast<float4*>(a.dx + (size_t)r * a.ld_dx + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int nB = dyn_count(a.dyn_b, a.B);
    const bool staged = nB < NPP_SEG;
    if (staged)
        for (int i = threadIdx.x; i <= nB; i += blockDim.x) sseg[i] = a.cat_seg[i];
    const int r = blockIdx.x * WPB + w;
    const int src = r < n_cap ? a.perm[r] : -1;                   // (in flight next to the staging loads)
    __syncthreads();
    if (src < 0 || nB <= 0) return;
    int lo = 0, hi = nB;                                          // session b with cat_seg[b] <= r < cat_seg[b + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((staged ? sseg[mid] : a.cat_seg[mid]) <= r) lo = mid; else hi = mid;
    }
    const int b = lo;
    if (r >= (staged ? sseg[nB] : a.cat_seg[nB])) return;         // behind the last live session
    int hit = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < a.npick && a.g_pick[k] != nullptr && a.pick[k][b] == src) hit = k;
    const float iv = a.invr[r];
    float dot = 0.f;
    for (int c = lane * 4; c < a.D; c += 256) {
        const float4 yy = *reinterpret_cast<const float4*>(a.allf + (size_t)r * a.D + c);
        float4 gg = *reinterpret_cast<const float4*>(a.g_allf + (size_t)r * a.ld_g + c);
        if (hit >= 0) {
            const float4 p = *reinterpret_cast<const float4*>(a.g_pick[hit] + (size_t)b * a.ld_gp[hit] + c);
            gg.x += p.x; gg.y += p.y; gg.z += p.z; gg.w += p.w;
        }
        dot += yy.x * gg.x + yy.y * gg.y + yy.z * gg.z + yy.w * gg.w;
    }
    dot = wave_sum(dot);
    for (int c = lane * 4; c < a.D; c += 256) {
        const float4 yy = *reinterpret_cast<const float4*>(a.allf + (size_t)r * a.D + c);
        float4 gg = *reinterpret_cast<const float4*>(a.g_allf + (size_t)r * a.ld_g + c);
        if (hit >= 0) {
            const float4 p = *reinterpret_cast<const float4*>(a.g_pick[hit] + (size_t)b * a.ld_gp[hit] + c);
            gg.x += p.x; gg.y += p.y; gg.z += p.z; gg.w += p.w;
        }
        *reinterpret_cast<float4*>(a.dx + (size_t)src * a.ld_dx + c) =
            make_float4(iv * (gg.x - yy.x * dot), iv * (gg.y - yy.y * dot), iv * (gg.z - yy.z * dot), iv * (gg.w - yy.w * dot));
    }
}
}  // namespace

// allf [n_cap, D] = normalised rows of x in concatenated order (perm), invr [n_cap] their 1 / norm; pick k < npick <= 4:
// pout_k [B, D] (row stride ld_p[k]) = normalised row pick_k[b] of x.  pick / pout / ld_p are HOST arrays.  Replaces F.normalize + the per-session concatenation + filter_nodes(last) of msgifsr.py:131-147,260-264.
extern "C" int srec_norm_perm_pick_fwd(const float* x, int ld_x, const int* perm, int n_cap, const int* dyn_t, int D,
                                       int eps_mode, float eps, float* allf, float* invr, int npick, int B, const int* dyn_b,
                                       const void* pick, const void* pout, const int* ld_p, void* stream) {
    if (n_cap <= 0) return 0;
    if (npick < 0 || npick > 4 || (D & 3) || (ld_x & 3)) return SREC_BAD_ARG;
    NppArgs a{};
    a.x = x; a.ld_x = ld_x; a.perm = perm; a.dyn_t = dyn_t; a.n_cap = n_cap; a.D = D; a.eps_mode = eps_mode; a.eps = eps;
    a.allf = allf; a.invr = invr;
    a.npick = npick; a.B = B; a.dyn_b = dyn_b;
    for (int k = 0; k < npick; ++k) {
        a.pick[k] = ((const int* const*)pick)[k];
        a.pout[k] = ((float* const*)pout)[k];
        a.ld_p[k] = ld_p[k];
        if (a.pick[k] == nullptr || a.pout[k] == nullptr || (a.ld_p[k] & 3)) return SREC_BAD_ARG;
    }
    const int rows = n_cap + npick * B;
    hipLaunchKernelGGL(norm_perm_pick_fwd_kernel, dim3(cdiv(rows, WPB)), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// dx [n_rows, D] (stacked order, EVERY row written: capacity padding as zeros) = gradient of x given g_allf (gradient of the
// concatenated normalised rows) and the pick gradients g_pick_k [B, D] (HOST array of npick pointers, entries may be NULL).
// row0 / ncap / dyn_n (HOST arrays of nt <= 4 entries): the node types' row ranges and live counts inside the stacked matrix.
extern "C" int srec_norm_perm_pick_bwd(const float* allf, const float* invr, const float* g_allf, int ld_g, const int* perm,
                                       int n_cap, const int* cat_seg, int B, const int* dyn_b, int D, int npick, const void* pick,
                                       const void* g_pick, const int* ld_gp, float* dx, int ld_dx, int nt, const int* row0,
                                       const int* ncap, const void* dyn_n, int n_rows, void* stream) {
    if (B <= 0 || n_rows <= 0) return 0;
    if (npick < 0 || npick > 4 || nt <= 0 || nt > 4 || (D & 3) || (ld_g & 3) || (ld_dx & 3)) return SREC_BAD_ARG;
    NppBwdArgs a{};
    a.allf = allf; a.invr = invr; a.g_allf = g_allf; a.ld_g = ld_g; a.perm = perm; a.cat_seg = cat_seg; a.B = B; a.dyn_b = dyn_b;
    a.D = D; a.npick = npick; a.dx = dx; a.ld_dx = ld_dx; a.nt = nt; a.n_rows = n_rows;
    for (int k = 0; k < npick; ++k) {
        a.pick[k] = ((const int* const*)pick)[k];
        a.g_pick[k] = g_pick != nullptr ? ((const float* const*)g_pick)[k] : nullptr;
        a.ld_gp[k] = ld_gp != nullptr ? ld_gp[k] : D;
        if (a.pick[k] == nullptr || (a.ld_gp[k] & 3)) return SREC_BAD_ARG;
    }
    for (int t = 0; t < nt; ++t) {
        a.row0[t] = row0[t]; a.ncap[t] = ncap[t];
        a.dyn_n[t] = dyn_n != nullptr ? ((const int* const*)dyn_n)[t] : nullptr;
    }
    // (a pick row matched by several orders: only with one pick list per order, as here - at most one k matches a row's type)
    const int row_blocks = cdiv(n_cap, WPB);
    hipLaunchKernelGGL(norm_perm_pick_bwd_kernel, dim3(row_blocks + cdiv(n_rows, 64)), dim3(256), 0, (hipStream_t)stream, a,
                       n_cap, row_blocks);
    SREC_LAUNCH_CHECK();
    return 0;
}

// ---- batch intake: pinned host words -> device, as a KERNEL ---------------------------------------------------------------
// A captured training step takes its batch (one int32 buffer, ~1 MB: batch.py) from page-locked host memory.  The copy is done
// by a kernel that loads the host words over PCIe itself (page-locked memory is mapped into the device's address space) and
// stores them in HBM: it is ordered on the compute stream like any other kernel of the step - no DMA engine, no cross-stream
// event between a copy queue and the replayed graph.  16-byte loads, one per lane and iteration, ~n/4/256 workgroups so that
// enough reads are in flight to cover the link's latency.
namespace {
__global__ void copy_words_kernel(const int4* __restrict__ src, int4* __restrict__ dst, long n4, const int* __restrict__ src_tail,
                                  int* __restrict__ dst_tail, int tail) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
}  // namespace

// (srec_copy_words_mailbox - batch intake inside a captured step - is a role of the step's prologue launch: prep.hip)

// dst [n] (device) = src [n] (page-locked host memory or device memory), int32 words; both 16-byte aligned
extern "C" int srec_copy_words(const int* src, int* dst, long n, void* stream) {
    if (n <= 0) return 0;
    if (src == nullptr || dst == nullptr || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return SREC_BAD_ARG;
    const long n4 = n / 4;
    const int tail = (int)(n - 4 * n4);
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)(n4 > 0 ? (n4 + 255) / 256 : 1)), dim3(256), 0, (hipStream_t)stream,
                       (const int4*)src, (int4*)dst, n4, src + 4 * n4, dst + 4 * n4, tail);
    SREC_LAUNCH_CHECK();
    return 0;
}
