// srec_score_topk: the K best catalog items of every session WITHOUT materialising the (B, V) score matrix
// (evaluate: train.py:36-55 = model forward -> logits.topk(20); SURVEY 8(a12), 8(f) rank 2: at V = 10 M the
// 20 GB score matrix cannot exist).  Ranking by z[b,v] = cs[v] * <sr_b, E_v> equals ranking by the log-probabilities.
//
// Pass 1 (topk_part_kernel): workgroup = 16 sessions x one item range.  The 16 session vectors sit in LDS; every
// thread scores ONE item against all 16 (its table row streams through registers once, LDS reads are broadcasts),
// then only scores that beat the session's current K-th best (kept in LDS) are pushed to a candidate list and merged
// by one wavefront - after the first chunk almost nothing passes the threshold.  Ties break towards the lower item id.
// Pass 2 (topk_merge_kernel): one wavefront per session merges the per-range lists.  fp32 throughout, no atomics on
// results (the candidate counter is an LDS slot index only; lists are re-sorted), deterministic output.
#include "common.h"

namespace {

constexpr int SB = 16;        // sessions per workgroup
constexpr int CHUNK = 256;    // items per step (one per thread)
constexpr int MAXK = 32;
constexpr int MAXC = 256;     // candidates per session and chunk

struct Cand { float v; int i; };

__device__ __forceinline__ bool better(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

// one wavefront: keep the K best of (list[0..K) U cand[0..n)) in list, sorted descending
__device__ void merge_topk(float* lv, int* li, int K, const float* cv, const int* ci, int n, int lane) {
    // selection by repeated arg-max over <= K + n <= 288 entries (5 per lane); taken entries are masked out
    float v[5]; int id[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        const int p = lane + 64 * e;
        if (p < K) { v[e] = lv[p]; id[e] = li[p]; }
        else if (p - K < n) { v[e] = cv[p - K]; id[e] = ci[p - K]; }
        else { v[e] = -INFINITY; id[e] = 0x7fffffff; }
    }
    for (int r = 0; r < K; ++r) {
        float bv = v[0]; int bi = id[0];
#pragma unroll
        for (int e = 1; e < 5; ++e)
            if (better(v[e], id[e], bv, bi)) { bv = v[e]; bi = id[e]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
#pragma unroll
        for (int e = 0; e < 5; ++e)
            if (id[e] == bi && v[e] == bv) { v[e] = -INFINITY; id[e] = 0x7fffffff; }
        if (lane == 0) { lv[r] = bv; li[r] = bi; }
    }
}

__global__ __launch_bounds__(256) void topk_part_kernel(const float* __restrict__ sr, int ld_sr,
                                                        const float* __restrict__ E, int ld_e,
                                                        const float* __restrict__ cs, int B, int V, int d, int K,
                                                        int items_per_range, float* __restrict__ pv,
                                                        int* __restrict__ pi) {
    extern __shared__ float smem[];
    float* ss = smem;                                   // [SB][d] session vectors
    float* lv = ss + SB * d;                            // [SB][MAXK] running best values
    int* li = reinterpret_cast<int*>(lv + SB * MAXK);   // [SB][MAXK] ... and item ids
    float* cv = reinterpret_cast<float*>(li + SB * MAXK);   // [SB][MAXC] candidates of this chunk
    int* ci = reinterpret_cast<int*>(cv + SB * MAXC);
    int* cn = ci + SB * MAXC;                           // [SB] candidate counts
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b0 = blockIdx.y * SB, range = blockIdx.x;
    const int v0 = range * items_per_range, v1 = min(V, v0 + items_per_range);
    for (int i = tid; i < SB * d; i += 256) {
        const int j = i / d, c = i % d;
        ss[i] = (b0 + j < B) ? sr[(size_t)(b0 + j) * ld_sr + c] : 0.f;
    }
    for (int i = tid; i < SB * MAXK; i += 256) { lv[i] = -INFINITY; li[i] = 0x7fffffff; }
    if (tid < SB) cn[tid] = 0;
    __syncthreads();
    for (int base = v0; base < v1; base += CHUNK) {
        const int v = base + tid;
        float acc[SB];
#pragma unroll
        for (int j = 0; j < SB; ++j) acc[j] = 0.f;
        if (v < v1) {
            const float* er = E + (size_t)v * ld_e;
            for (int c = 0; c < d; c += 4) {
                const float4 e4 = *reinterpret_cast<const float4*>(er + c);
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const float4 s4 = *reinterpret_cast<const float4*>(ss + j * d + c);
                    acc[j] += e4.x * s4.x + e4.y * s4.y + e4.z * s4.z + e4.w * s4.w;
                }
            }
            const float sc = cs != nullptr ? cs[v] : 1.f;
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const float z = sc * acc[j];
                if (better(z, v, lv[j * MAXK + K - 1], li[j * MAXK + K - 1])) {
                    const int slot = atomicAdd(&cn[j], 1);            // slot index only: the merge re-sorts
                    cv[j * MAXC + slot] = z;
                    ci[j * MAXC + slot] = v;
                }
            }
        }
        __syncthreads();
        for (int j = w; j < SB; j += 4) {
            const int n = cn[j];
            if (n > 0) merge_topk(lv + j * MAXK, li + j * MAXK, K, cv + j * MAXC, ci + j * MAXC, n, lane);
        }
        __syncthreads();
        if (tid < SB) cn[tid] = 0;
        __syncthreads();
    }
    for (int i = tid; i < SB * K; i += 256) {
        const int j = i / K, r = i % K;
        if (b0 + j < B) {
            pv[((size_t)range * B + b0 + j) * K + r] = lv[j * MAXK + r];
            pi[((size_t)range * B + b0 + j) * K + r] = li[j * MAXK + r];
        }
    }
}

__global__ void topk_merge_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int R, int B, int K,
                                  float* __restrict__ out_v, int* __restrict__ out_i) {
    __shared__ float lv[4][MAXK];
    __shared__ int li[4][MAXK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;
    for (int r = lane; r < MAXK; r += 64) { lv[w][r] = -INFINITY; li[w][r] = 0x7fffffff; }
    __builtin_amdgcn_wave_barrier();
    // fold the R lists in groups of up to MAXC candidates
    const int per = MAXC / K;                                   // lists per merge step
    for (int r0 = 0; r0 < R; r0 += per) {
        const int nl = min(per, R - r0);
        // candidates are addressed in place: list r, entry e  ->  ((r0 + r) * B + b) * K + e
        const int n = nl * K;
        __shared__ float cv[4][MAXC];
        __shared__ int ci[4][MAXC];
        for (int p = lane; p < n; p += 64) {
            const size_t src = ((size_t)(r0 + p / K) * B + b) * K + p % K;
            cv[w][p] = pv[src];
            ci[w][p] = pi[src];
        }
        __builtin_amdgcn_wave_barrier();
        merge_topk(lv[w], li[w], K, cv[w], ci[w], n, lane);
        __builtin_amdgcn_wave_barrier();
    }
    for (int r = lane; r < K; r += 64) {
        out_v[(size_t)b * K + r] = lv[w][r];
        out_i[(size_t)b * K + r] = li[w][r];
    }
}

inline int pick_ranges(int B, int V) {
    const int tiles = cdiv(B, SB);
    int R = cdiv(1024, tiles);                                  // ~4 workgroups per CU
    const int maxR = cdiv(V, 4 * CHUNK);                        // at least 4 chunks per range
    if (R > maxR) R = maxR;
    return R < 1 ? 1 : R;
}

}  // namespace

// ws: srec_score_topk_ws(B, V, K) bytes of scratch (per-range partial lists)
extern "C" int srec_score_topk_ws(int B, int V, int K, long* bytes) {
    if (B <= 0 || V <= 0 || K <= 0 || K > MAXK || bytes == nullptr) return SREC_BAD_ARG;
    *bytes = (long)pick_ranges(B, V) * B * K * 8;
    return 0;
}

// out_val [B, K] fp32 scores z = cs[v] * <sr_b, E_v> (descending), out_idx [B, K] int32 item ids; ties -> lower id.
extern "C" int srec_score_topk(const float* sr, int ld_sr, const float* E, int ld_e, const float* cs, int B, int V,
                               int d, int K, float* out_val, int* out_idx, void* ws, void* stream) {
    if (B <= 0 || V <= 0) return 0;
    if (K <= 0 || K > MAXK || K > V || (d & 3) || (ld_e & 3) || d > 1024 || ws == nullptr || ((uintptr_t)E & 15))
        return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int R = pick_ranges(B, V);
    const int ipr = cdiv(cdiv(V, R), CHUNK) * CHUNK;
    const int Ract = cdiv(V, ipr);
    float* pv = (float*)ws;
    int* pi = (int*)(pv + (size_t)R * B * K);
    const size_t lds = (size_t)SB * d * 4 + (size_t)SB * MAXK * 8 + (size_t)SB * MAXC * 8 + SB * 4;
    static std::atomic<unsigned long long> optin{0};
    if (int rc = srec_lds_optin((const void*)topk_part_kernel, 160 * 1024, optin)) return rc;
    hipLaunchKernelGGL(topk_part_kernel, dim3(Ract, cdiv(B, SB)), dim3(256), lds, st, sr, ld_sr, E, ld_e, cs, B, V, d, K, ipr,
                       pv, pi);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, pv, pi, Ract, B, K, out_val, out_idx);
    SREC_LAUNCH_CHECK();
    return 0;
}
