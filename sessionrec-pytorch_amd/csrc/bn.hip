// BatchNorm1d over the node / session rows of a batch and per-channel PReLU (LESSR:
// lessr.py:12,32 (EOPA), :56,66 (SGAT), :90,105 (readout), :162,179 (session vector);
// PReLU activations lessr.py:140,149,159).
//
// Column statistics over a [n, D] matrix are two-stage deterministic reductions: a grid of
// (column blocks x 32 row chunks) writes per-chunk partials, and the kernel that NEEDS the
// statistics (the normalising pass) adds the 32 partials of its columns itself - every workgroup
// redundantly, 32 x D floats out of L2 - so a BatchNorm is two launches each way.  Variance: per
// chunk sum and M2 about the chunk's own mean (two passes over the chunk's rows), combined as
// M2 = sum_k M2_k + n_k (mean_k - mean)^2: as accurate as the two-pass formula of torch's
// BatchNorm.  The live row count may come from device memory (capacity-padded batches).
//
//   srec_bn_fwd_train  batch statistics + y = (x - mean) * invstd * gamma + beta; workgroup 0 also stores mean / biased
//                      var for the backward and updates running_mean / running_var / num_batches_tracked exactly like
//                      nn.BatchNorm1d (momentum, unbiased var)
//   srec_bn_apply_fwd  the same normalisation with given statistics (eval mode)
//   srec_bn_bwd        d gamma, d beta, dx (training-mode formula) or dx = dy*gamma*invstd (eval)
//   srec_prelu_fwd/bwd y = x > 0 ? x : a[c] x ; da[c] = sum_{x<=0} dy x
#include "common.h"

namespace {

constexpr int NCHUNK = 32;

// thread -> (column lane cl of cw, row lane rg of nrg): 64 column lanes per workgroup, or the next power of two >= ncol
// when the matrix is narrower (d = 32: 8 row lanes instead of 4 with half the threads idle)
#define COL_LANES                                                                                    \
    const int cw = ncol > 32 ? 64 : (ncol > 16 ? 32 : 16), nrg = 256 / cw;                           \
    const int cl = threadIdx.x & (cw - 1), rg = threadIdx.x / cw, c = blockIdx.x * 64 + cl

inline __device__ float lane_groups_sum(const float* red, int cl, int cw, int nrg) {
    float s = 0.f;
    for (int g = 0; g < nrg; ++g) s += red[g * cw + cl];
    return s;
}

inline __device__ void chunk_rows(int n, int k, int& r0, int& r1) {
    const int per = (n + NCHUNK - 1) / NCHUNK;
    r0 = k * per;
    r1 = min(n, r0 + per);
}

// part[k][c] = sum of x over chunk k;  part[NCHUNK + k][c] = sum (x - chunk mean)^2
__global__ void bn_part_fwd_kernel(const float* __restrict__ X, int ld, int n_cap, const int* __restrict__ dyn, int ncol,
                                   float* __restrict__ part) {
    __shared__ float red[256];
    COL_LANES;
    int r0, r1;
    chunk_rows(dyn_count(dyn, n_cap), blockIdx.y, r0, r1);
    float s = 0.f;
    if (c < ncol)
        for (int r = r0 + rg; r < r1; r += nrg) s += X[(size_t)r * ld + c];
    red[threadIdx.x] = s;
    __syncthreads();
    const float tot = lane_groups_sum(red, cl, cw, nrg);
    const float cm = tot / (float)max(1, r1 - r0);
    __syncthreads();
    float q = 0.f;
    if (c < ncol)
        for (int r = r0 + rg; r < r1; r += nrg) {
            const float x = X[(size_t)r * ld + c] - cm;
            q += x * x;
        }
    red[threadIdx.x] = q;
    __syncthreads();
    if (rg == 0 && c < ncol) {
        part[(size_t)blockIdx.y * ncol + c] = tot;
        part[(size_t)(NCHUNK + blockIdx.y) * ncol + c] = lane_groups_sum(red, cl, cw, nrg);
    }
}

// mean and biased variance of column c from the chunk partials (64 independent loads, then arithmetic)
inline __device__ void bn_combine(const float* __restrict__ part, int ncol, int c, int n, float& mean, float& var) {
    float ps[NCHUNK], pq[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        ps[k] = part[(size_t)k * ncol + c];
        pq[k] = part[(size_t)(NCHUNK + k) * ncol + c];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) s += ps[k];
    mean = s / (float)max(1, n);
    const int per = (n + NCHUNK - 1) / NCHUNK;
    const float inv_per = 1.f / (float)max(1, per);
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int nk = min(n, (k + 1) * per) - k * per;       // rows of chunk k (<= 0: empty)
        if (nk > 0) {
            const float dm = ps[k] * (nk == per ? inv_per : 1.f / (float)nk) - mean;
            m2 += pq[k] + (float)nk * dm * dm;
        }
    }
    var = m2 / (float)max(1, n);
}

// part[k][c] = sum dy;  part[NCHUNK + k][c] = sum dy * (x - mean[c]) * rsqrt(var[c] + eps)
__global__ void bn_part_bwd_kernel(const float* __restrict__ dY, int ld_dy, const float* __restrict__ X, int ld_x,
                                   const float* __restrict__ mean, const float* __restrict__ var, float eps, int n_cap,
                                   const int* __restrict__ dyn, int ncol, float* __restrict__ part) {
    __shared__ float red[2][256];
    COL_LANES;
    int r0, r1;
    chunk_rows(dyn_count(dyn, n_cap), blockIdx.y, r0, r1);
    float s0 = 0.f, s1 = 0.f;
    if (c < ncol) {
        const float a0 = mean[c], a1 = rsqrtf(var[c] + eps);
        for (int r = r0 + rg; r < r1; r += nrg) {
            const float dy = dY[(size_t)r * ld_dy + c];
            s0 += dy;
            s1 += dy * (X[(size_t)r * ld_x + c] - a0) * a1;
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rg == 0 && c < ncol) {
        part[(size_t)blockIdx.y * ncol + c] = lane_groups_sum(red[0], cl, cw, nrg);
        part[(size_t)(NCHUNK + blockIdx.y) * ncol + c] = lane_groups_sum(red[1], cl, cw, nrg);
    }
}

// out[c] = sum of the chunk partials
__global__ void colstat_final_kernel(const float* __restrict__ part, int ncol, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    float s = 0.f;
    for (int k = 0; k < NCHUNK; ++k) s += part[(size_t)k * ncol + c];
    out[c] = s;
}

// training-mode normalisation: statistics from the chunk partials
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const float* __restrict__ X, int ld_x, const float* __restrict__ part, float eps,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float momentum,
                                      float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
                                      float* __restrict__ mean_out, float* __restrict__ var_out, int n_cap,
                                      const int* __restrict__ dyn, int D, float* __restrict__ Y, int ld_y) {
    const int n = dyn_count(dyn, n_cap);
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < D; c += blockDim.x) {
            float mean, var;
            bn_combine(part, D, c, n, mean, var);
            mean_out[c] = mean;
            var_out[c] = var;
            if (rmean != nullptr && rvar != nullptr) {
                const float unb = n > 1 ? var * (float)n / (float)(n - 1) : var;
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
            }
        }
        if (threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
    }
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(idx / D), c = (int)(idx % D);
    if (row >= n_cap) return;
    float y = 0.f;
    if (row < n) {
        const float x = X[(size_t)row * ld_x + c];
        float mean, var;
        bn_combine(part, D, c, n, mean, var);             // every thread for its own column: no LDS, no barrier
        y = (x - mean) * rsqrtf(var + eps) * gamma[c] + beta[c];
    }
    Y[(size_t)row * ld_y + c] = y;
}

__global__ void bn_apply_fwd_kernel(const float* __restrict__ X, int ld_x, const float* __restrict__ mean,
                                    const float* __restrict__ var, float eps, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, int n_cap, const int* __restrict__ dyn, int D,
                                    float* __restrict__ Y, int ld_y) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(idx / D), c = (int)(idx % D);
    if (row >= n_cap) return;
    float y = 0.f;
    if (row < dyn_count(dyn, n_cap))
        y = (X[(size_t)row * ld_x + c] - mean[c]) * rsqrtf(var[c] + eps) * gamma[c] + beta[c];
    Y[(size_t)row * ld_y + c] = y;
}

inline __device__ void bn_sums(const float* __restrict__ part, int ncol, int c, float& s0, float& s1) {
    float p0[NCHUNK], p1[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        p0[k] = part[(size_t)k * ncol + c];
        p1[k] = part[(size_t)(NCHUNK + k) * ncol + c];
    }
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) { s0 += p0[k]; s1 += p1[k]; }
}

// training: dx = gamma*invstd/n * (n*dy - dbeta - xhat*dgamma);  eval: dx = dy*gamma*invstd.  d beta / d gamma are the
// sums of the chunk partials; workgroup 0 stores them
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ dY, int ld_dy, const float* __restrict__ X, int ld_x,
                                    const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                    const float* __restrict__ gamma, const float* __restrict__ part,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int training, int n_cap,
                                    const int* __restrict__ dyn, int D, float* __restrict__ dX, int ld_dx) {
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < D; c += blockDim.x) {
            float s0, s1;
            bn_sums(part, D, c, s0, s1);
            dbeta[c] = s0;
            dgamma[c] = s1;
        }
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(idx / D), c = (int)(idx % D);
    if (row >= n_cap) return;
    const int n = dyn_count(dyn, n_cap);
    float o = 0.f;
    if (row < n) {
        const float is = rsqrtf(var[c] + eps);
        const float dy = dY[(size_t)row * ld_dy + c];
        if (training) {
            const float xh = (X[(size_t)row * ld_x + c] - mean[c]) * is;
            float s0, s1;
            bn_sums(part, D, c, s0, s1);                  // every thread for its own column
            o = gamma[c] * is / (float)n * ((float)n * dy - s0 - xh * s1);
        } else {
            o = dy * gamma[c] * is;
        }
    }
    dX[(size_t)row * ld_dx + c] = o;
}

__global__ void prelu_fwd_kernel(const float* __restrict__ X, int ld_x, const float* __restrict__ a, int n_cap,
                                 const int* __restrict__ dyn, int D, float* __restrict__ Y, int ld_y) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(idx / D), c = (int)(idx % D);
    if (row >= n_cap) return;
    float y = 0.f;
    if (row < dyn_count(dyn, n_cap)) {
        const float x = X[(size_t)row * ld_x + c];
        y = x > 0.f ? x : a[c] * x;
    }
    Y[(size_t)row * ld_y + c] = y;
}

// dX = dy * (x>0 ? 1 : a) over the capacity rows (zeros beyond the live ones);  part[k][c] = sum of dy * min(x, 0) over
// chunk k of the capacity rows (d a = their sum, colstat_final_kernel)
__global__ void prelu_bwd_kernel(const float* __restrict__ dY, int ld_dy, const float* __restrict__ X, int ld_x,
                                 const float* __restrict__ a, int n_cap, const int* __restrict__ dyn, int ncol,
                                 float* __restrict__ dX, int ld_dx, float* __restrict__ part) {
    __shared__ float red[256];
    COL_LANES;
    const int n = dyn_count(dyn, n_cap);
    int r0, r1;
    chunk_rows(n_cap, blockIdx.y, r0, r1);
    float t = 0.f;
    if (c < ncol) {
        const float ac = a[c];
        for (int r = r0 + rg; r < r1; r += nrg) {
            float dx = 0.f;
            if (r < n) {
                const float x = X[(size_t)r * ld_x + c], dy = dY[(size_t)r * ld_dy + c];
                dx = x > 0.f ? dy : ac * dy;
                t += x > 0.f ? 0.f : dy * x;
            }
            dX[(size_t)r * ld_dx + c] = dx;
        }
    }
    red[threadIdx.x] = t;
    __syncthreads();
    if (rg == 0 && c < ncol) part[(size_t)blockIdx.y * ncol + c] = lane_groups_sum(red, cl, cw, nrg);
}

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

// training-mode forward.  rmean / rvar / nbt (num_batches_tracked, int64) nullable; mean / var [D] out.  ws: 64*D floats
extern "C" int srec_bn_fwd_train(const float* X, int ld_x, int n_cap, const int* dyn, int D, const float* gamma,
                                 const float* beta, float eps, float momentum, float* rmean, float* rvar, long long* nbt,
                                 float* mean, float* var, float* Y, int ld_y, float* ws, void* stream) {
    if (D <= 0 || ws == nullptr) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_part_fwd_kernel, dim3(cdiv(D, 64), NCHUNK), dim3(256), 0, st, X, ld_x, n_cap, dyn, D, ws);
    hipLaunchKernelGGL(bn_train_apply_kernel, dim3(n_cap > 0 ? blocks_for((long)n_cap * D) : 1), dim3(256), 0,
                       st, X, ld_x, ws, eps, gamma, beta, momentum, rmean, rvar, nbt, mean, var, n_cap,
                       dyn, D, Y, ld_y);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_bn_apply_fwd(const float* X, int ld_x, const float* mean, const float* var, float eps,
                                 const float* gamma, const float* beta, int n_cap, const int* dyn, int D, float* Y,
                                 int ld_y, void* stream) {
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(blocks_for((long)n_cap * D)), dim3(256), 0, (hipStream_t)stream, X, ld_x,
                       mean, var, eps, gamma, beta, n_cap, dyn, D, Y, ld_y);
    SREC_LAUNCH_CHECK();
    return 0;
}

// dgamma, dbeta always produced; dx by the training formula (batch statistics) or the eval one.  ws: 64*D floats
extern "C" int srec_bn_bwd(const float* dY, int ld_dy, const float* X, int ld_x, const float* mean, const float* var,
                           float eps, const float* gamma, int training, int n_cap, const int* dyn, int D, float* dX,
                           int ld_dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    if (n_cap <= 0) return 0;
    if (ws == nullptr || D <= 0) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_part_bwd_kernel, dim3(cdiv(D, 64), NCHUNK), dim3(256), 0, st, dY, ld_dy, X, ld_x, mean, var, eps,
                       n_cap, dyn, D, ws);
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(blocks_for((long)n_cap * D)), dim3(256), 0, st, dY,
                       ld_dy, X, ld_x, mean, var, eps, gamma, ws, dgamma, dbeta, training, n_cap, dyn, D, dX, ld_dx);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_prelu_fwd(const float* X, int ld_x, const float* a, int n_cap, const int* dyn, int D, float* Y,
                              int ld_y, void* stream) {
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(prelu_fwd_kernel, dim3(blocks_for((long)n_cap * D)), dim3(256), 0, (hipStream_t)stream, X, ld_x, a,
                       n_cap, dyn, D, Y, ld_y);
    SREC_LAUNCH_CHECK();
    return 0;
}

// dX and d a[c] = sum dy * min(x, 0).  ws: 32*D floats = the chunk partials [32][D]; da NULL: they are left to the caller to sum
extern "C" int srec_prelu_bwd(const float* dY, int ld_dy, const float* X, int ld_x, const float* a, int n_cap,
                              const int* dyn, int D, float* dX, int ld_dx, float* da, float* ws, void* stream) {
    if (n_cap <= 0) return 0;
    if (ws == nullptr || D <= 0) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3(cdiv(D, 64), NCHUNK), dim3(256), 0, st, dY, ld_dy, X, ld_x, a, n_cap, dyn, D, dX,
                       ld_dx, ws);
    if (da != nullptr) hipLaunchKernelGGL(colstat_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, ws, D, da);
    SREC_LAUNCH_CHECK();
    return 0;
}
