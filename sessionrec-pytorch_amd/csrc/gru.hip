// GRU gate math (torch.nn.GRU / GRUCell semantics) as fused pointwise kernels; the two
// projections GI = x W_ih^T + b_ih and GH = h W_hh^T + b_hh run on the matrix cores
// (srec_gemm_f32), one batched GEMM per time step over ALL nodes instead of the
// reference's cuDNN GRU launches on tiny sequences.
//
//   r = sigmoid(gi_r + gh_r); z = sigmoid(gi_z + gh_z); n = tanh(gi_n + r*gh_n); h' = (1-z)*n + z*h
//
// Used by: SemanticExpander's k-gram GRU (msgifsr.py:25,42: k <= order time steps, every k-gram
// node has exactly k steps), SRGNNLayer's GRUCell (srgnn.py:15,45).
//   srec_gru_pointwise_fwd/bwd   one time step; GH == NULL means h_prev == 0 (gh = b_hh)
//   srec_gram_combine_fwd/bwd    out = 0.5*mean_t x[n,t,:] + 0.5*h_last[n,:]   (msgifsr.py:37,45)
#include "common.h"

namespace {

__global__ void gru_pw_fwd_kernel(const float* __restrict__ GI, int ld_gi, const float* __restrict__ GH, int ld_gh,
                                  const float* __restrict__ bhh, const float* __restrict__ Hp, int ld_hp, int n_cap,
                                  const int* __restrict__ dyn, int d, float* __restrict__ Hn, int ld_hn,
                                  float* __restrict__ gates /* [n,3d]: r, z, n */) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = idx / d, c = idx % d;
    if (row >= n_cap) return;
    if (row >= dyn_count(dyn, n_cap)) {
        Hn[(size_t)row * ld_hn + c] = 0.f;
        return;
    }
    const float* gi = GI + (size_t)row * ld_gi;
    float ghr, ghz, ghn, hp;
    if (GH != nullptr) {
        const float* gh = GH + (size_t)row * ld_gh;
        ghr = gh[c]; ghz = gh[d + c]; ghn = gh[2 * d + c];
        hp = Hp[(size_t)row * ld_hp + c];
    } else {
        ghr = bhh[c]; ghz = bhh[d + c]; ghn = bhh[2 * d + c];
        hp = 0.f;
    }
    const float r = sigmoidf_(gi[c] + ghr);
    const float z = sigmoidf_(gi[d + c] + ghz);
    const float n = tanhf(gi[2 * d + c] + r * ghn);
    Hn[(size_t)row * ld_hn + c] = (1.f - z) * n + z * hp;
    float* g = gates + (size_t)row * 3 * d;
    g[c] = r; g[d + c] = z; g[2 * d + c] = n;
}

__global__ void gru_pw_bwd_kernel(const float* __restrict__ dHn, int ld_dh, const float* __restrict__ gates,
                                  const float* __restrict__ GH, int ld_gh, const float* __restrict__ bhh,
                                  const float* __restrict__ Hp, int ld_hp, int n_cap, const int* __restrict__ dyn,
                                  int d, float* __restrict__ dGI, int ld_dgi, float* __restrict__ dGH, int ld_dgh,
                                  float* __restrict__ dHp, int ld_dhp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = idx / d, c = idx % d;
    if (row >= n_cap) return;
    float* dgi = dGI + (size_t)row * ld_dgi;
    float* dgh = dGH + (size_t)row * ld_dgh;
    if (row >= dyn_count(dyn, n_cap)) {
        dgi[c] = dgi[d + c] = dgi[2 * d + c] = 0.f;
        dgh[c] = dgh[d + c] = dgh[2 * d + c] = 0.f;
        if (dHp != nullptr) dHp[(size_t)row * ld_dhp + c] = 0.f;
        return;
    }
    const float* g = gates + (size_t)row * 3 * d;
    const float r = g[c], z = g[d + c], n = g[2 * d + c];
    const float ghn = GH != nullptr ? GH[(size_t)row * ld_gh + 2 * d + c] : bhh[2 * d + c];
    const float hp = Hp != nullptr ? Hp[(size_t)row * ld_hp + c] : 0.f;
    const float dh = dHn[(size_t)row * ld_dh + c];
    const float dn = dh * (1.f - z);
    const float dz = dh * (hp - n);
    const float dpn = dn * (1.f - n * n);
    const float dpr = dpn * ghn * r * (1.f - r);
    const float dpz = dz * z * (1.f - z);
    dgi[c] = dpr; dgi[d + c] = dpz; dgi[2 * d + c] = dpn;
    dgh[c] = dpr; dgh[d + c] = dpz; dgh[2 * d + c] = dpn * r;
    if (dHp != nullptr) dHp[(size_t)row * ld_dhp + c] = dh * z;
}

// X: [n, k, d] contiguous; out[n,:] = 0.5/k * sum_t X[n,t,:] + 0.5 * Hl[n,:]
__global__ void gram_combine_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Hl, int ld_h, int n_cap,
                                        const int* __restrict__ dyn, int k, int d, float* __restrict__ out, int ld_o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = idx / d, c = idx % d;
    if (row >= n_cap) return;
    float v = 0.f;
    if (row < dyn_count(dyn, n_cap)) {
        float s = 0.f;
        for (int t = 0; t < k; ++t) s += X[((size_t)row * k + t) * d + c];
        v = 0.5f * (s / (float)k) + 0.5f * Hl[(size_t)row * ld_h + c];
    }
    out[(size_t)row * ld_o + c] = v;
}

__global__ void gram_combine_bwd_kernel(const float* __restrict__ dout, int ld_o, int n_cap,
                                        const int* __restrict__ dyn, int k, int d, float* __restrict__ dX,
                                        float* __restrict__ dHl, int ld_h) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = idx / d, c = idx % d;
    if (row >= n_cap) return;
    const float g = row < dyn_count(dyn, n_cap) ? dout[(size_t)row * ld_o + c] : 0.f;
    const float gx = 0.5f * g / (float)k;
    for (int t = 0; t < k; ++t) dX[((size_t)row * k + t) * d + c] = gx;
    dHl[(size_t)row * ld_h + c] = 0.5f * g;
}

}  // namespace

extern "C" int srec_gru_pointwise_fwd(const float* GI, int ld_gi, const float* GH, int ld_gh, const float* bhh,
                                      const float* Hp, int ld_hp, int n_cap, const int* dyn, int d, float* Hn,
                                      int ld_hn, float* gates, void* stream) {
    if (n_cap <= 0) return 0;
    if (GH == nullptr && bhh == nullptr) return SREC_BAD_ARG;
    const long total = (long)n_cap * d;
    hipLaunchKernelGGL(gru_pw_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, GI,
                       ld_gi, GH, ld_gh, bhh, Hp, ld_hp, n_cap, dyn, d, Hn, ld_hn, gates);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_gru_pointwise_bwd(const float* dHn, int ld_dh, const float* gates, const float* GH, int ld_gh,
                                      const float* bhh, const float* Hp, int ld_hp, int n_cap, const int* dyn, int d,
                                      float* dGI, int ld_dgi, float* dGH, int ld_dgh, float* dHp, int ld_dhp,
                                      void* stream) {
    if (n_cap <= 0) return 0;
    if (GH == nullptr && bhh == nullptr) return SREC_BAD_ARG;
    const long total = (long)n_cap * d;
    hipLaunchKernelGGL(gru_pw_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dHn,
                       ld_dh, gates, GH, ld_gh, bhh, Hp, ld_hp, n_cap, dyn, d, dGI, ld_dgi, dGH, ld_dgh, dHp, ld_dhp);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_gram_combine_fwd(const float* X, const float* Hl, int ld_h, int n_cap, const int* dyn, int k, int d,
                                     float* out, int ld_o, void* stream) {
    if (n_cap <= 0) return 0;
    const long total = (long)n_cap * d;
    hipLaunchKernelGGL(gram_combine_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       X, Hl, ld_h, n_cap, dyn, k, d, out, ld_o);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_gram_combine_bwd(const float* dout, int ld_o, int n_cap, const int* dyn, int k, int d, float* dX,
                                     float* dHl, int ld_h, void* stream) {
    if (n_cap <= 0) return 0;
    const long total = (long)n_cap * d;
    hipLaunchKernelGGL(gram_combine_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dout, ld_o, n_cap, dyn, k, d, dX, dHl, ld_h);
    SREC_LAUNCH_CHECK();
    return 0;
}
