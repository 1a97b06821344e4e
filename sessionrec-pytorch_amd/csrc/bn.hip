// BatchNorm1d over the node / session rows of a batch and per-channel PReLU (LESSR:
// lessr.py:12,32 (EOPA), :56,66 (SGAT), :90,105 (readout), :162,179 (session vector);
// PReLU activations lessr.py:140,149,159).
//
// Column statistics over a [n, D] matrix are two-stage deterministic reductions
// (grid = column blocks x 32 row chunks, then a 32-way add), two-pass variance
// (mean first, then sum (x-mean)^2) like torch's CPU/GPU BatchNorm.  The live row
// count may come from device memory (capacity-padded batches).
//
//   srec_bn_stats      mean[D], var[D] (biased) over the live rows; optionally updates the
//                      running statistics exactly like nn.BatchNorm1d (momentum, unbiased var)
//   srec_bn_apply_fwd  y = (x - mean) * invstd * gamma + beta
//   srec_bn_bwd        d gamma, d beta, dx (training-mode formula) or dx = dy*gamma*invstd (eval)
//   srec_prelu_fwd/bwd y = x > 0 ? x : a[c] x ; da[c] = sum_{x<=0} dy x
#include "common.h"

namespace {

constexpr int NCHUNK = 32;

// mode 0: sum x   1: sum (x - mean[c])^2   2: sum dy * (x - mean[c]) * rsqrt(var[c]+eps)  (X2 = dy)
template <int MODE>
__global__ void colstat_part_kernel(const float* __restrict__ X, int ld, const float* __restrict__ X2, int ld2,
                                    const float* __restrict__ aux0, const float* __restrict__ var, float eps, int n_cap,
                                    const int* __restrict__ dyn, int ncol, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int n = dyn_count(dyn, n_cap);
    const int per = (n + NCHUNK - 1) / NCHUNK;
    const int r0 = blockIdx.y * per, r1 = min(n, r0 + per);
    float s = 0.f;
    if (c < ncol) {
        const float a0 = MODE >= 1 ? aux0[c] : 0.f, a1 = MODE == 2 ? rsqrtf(var[c] + eps) : 0.f;
        for (int r = r0 + rg; r < r1; r += 4) {
            const float x = X[(size_t)r * ld + c];
            if (MODE == 0) s += x;
            else if (MODE == 1) s += (x - a0) * (x - a0);
            else s += X2[(size_t)r * ld2 + c] * (x - a0) * a1;
        }
    }
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < ncol)
        part[(size_t)blockIdx.y * ncol + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[c] = scale_by_n ? sum / n : sum
__global__ void colstat_final_kernel(const float* __restrict__ part, int ncol, int n_cap, const int* __restrict__ dyn,
                                     int divide, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    float s = 0.f;
    for (int k = 0; k < NCHUNK; ++k) s += part[(size_t)k * ncol + c];
    const int n = dyn_count(dyn, n_cap);
    out[c] = divide ? s / (float)(n > 0 ? n : 1) : s;
}

__global__ void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var, int ncol, int n_cap,
                                  const int* __restrict__ dyn, float momentum, float* __restrict__ rmean,
                                  float* __restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int n = dyn_count(dyn, n_cap);
    const float unb = n > 1 ? var[c] * (float)n / (float)(n - 1) : var[c];
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
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

// training: dx = gamma*invstd/n * (n*dy - dbeta - xhat*dgamma);  eval: dx = dy*gamma*invstd
__global__ void bn_apply_bwd_kernel(const float* __restrict__ dY, int ld_dy, const float* __restrict__ X, int ld_x,
                                    const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                    const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, int training, int n_cap,
                                    const int* __restrict__ dyn, int D, float* __restrict__ dX, int ld_dx) {
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
            o = gamma[c] * is / (float)n * ((float)n * dy - dbeta[c] - xh * dgamma[c]);
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

// dX = dy * (x>0 ? 1 : a);  T = dy * min(x,0)  (column-summed afterwards into da)
__global__ void prelu_bwd_kernel(const float* __restrict__ dY, int ld_dy, const float* __restrict__ X, int ld_x,
                                 const float* __restrict__ a, int n_cap, const int* __restrict__ dyn, int D,
                                 float* __restrict__ dX, int ld_dx, float* __restrict__ T, int ld_t) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(idx / D), c = (int)(idx % D);
    if (row >= n_cap) return;
    float dx = 0.f, t = 0.f;
    if (row < dyn_count(dyn, n_cap)) {
        const float x = X[(size_t)row * ld_x + c], dy = dY[(size_t)row * ld_dy + c];
        dx = x > 0.f ? dy : a[c] * dy;
        t = x > 0.f ? 0.f : dy * x;
    }
    dX[(size_t)row * ld_dx + c] = dx;
    T[(size_t)row * ld_t + c] = t;
}

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

// mean/var (biased) over the live rows; rmean/rvar nullable.  ws: 32*D floats.
extern "C" int srec_bn_stats(const float* X, int ld, int n_cap, const int* dyn, int D, float* mean, float* var,
                             float* rmean, float* rvar, float momentum, float* ws, void* stream) {
    if (D <= 0 || ws == nullptr) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(cdiv(D, 64), NCHUNK);
    hipLaunchKernelGGL((colstat_part_kernel<0>), g, dim3(256), 0, st, X, ld, nullptr, 0, nullptr, nullptr, 0.f, n_cap, dyn, D, ws);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, ws, D, n_cap, dyn, 1, mean);
    hipLaunchKernelGGL((colstat_part_kernel<1>), g, dim3(256), 0, st, X, ld, nullptr, 0, mean, nullptr, 0.f, n_cap, dyn, D, ws);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, ws, D, n_cap, dyn, 1, var);
    if (rmean != nullptr && rvar != nullptr)
        hipLaunchKernelGGL(bn_running_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, mean, var, D, n_cap, dyn, momentum,
                           rmean, rvar);
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

// dgamma, dbeta always produced; dx by the training formula (batch statistics) or the eval one.  ws: 32*D floats
extern "C" int srec_bn_bwd(const float* dY, int ld_dy, const float* X, int ld_x, const float* mean, const float* var,
                           float eps, const float* gamma, int training, int n_cap, const int* dyn, int D, float* dX,
                           int ld_dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    if (n_cap <= 0) return 0;
    if (ws == nullptr) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(cdiv(D, 64), NCHUNK);
    hipLaunchKernelGGL((colstat_part_kernel<0>), g, dim3(256), 0, st, dY, ld_dy, nullptr, 0, nullptr, nullptr, 0.f, n_cap,
                       dyn, D, ws);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, ws, D, n_cap, dyn, 0, dbeta);
    hipLaunchKernelGGL((colstat_part_kernel<2>), g, dim3(256), 0, st, X, ld_x, dY, ld_dy, mean, var, eps, n_cap, dyn, D, ws);
    hipLaunchKernelGGL(colstat_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, ws, D, n_cap, dyn, 0, dgamma);
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(blocks_for((long)n_cap * D)), dim3(256), 0, st, dY, ld_dy, X, ld_x, mean,
                       var, eps, gamma, dgamma, dbeta, training, n_cap, dyn, D, dX, ld_dx);
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

// dX and T = dy*min(x,0) (its column sums are d a: use srec_col_sum on T)
extern "C" int srec_prelu_bwd(const float* dY, int ld_dy, const float* X, int ld_x, const float* a, int n_cap,
                              const int* dyn, int D, float* dX, int ld_dx, float* T, int ld_t, void* stream) {
    if (n_cap <= 0) return 0;
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3(blocks_for((long)n_cap * D)), dim3(256), 0, (hipStream_t)stream, dY, ld_dy, X,
                       ld_x, a, n_cap, dyn, D, dX, ld_dx, T, ld_t);
    SREC_LAUNCH_CHECK();
    return 0;
}
