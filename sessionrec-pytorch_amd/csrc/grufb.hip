// k-gram GRU of MSGIFSR's SemanticExpander (msgifsr.py:25,32-45): backward of ALL time steps and orders in ONE launch (bf16
// path, d = 128 or 256) - the counterpart of gruf.hip.  What stays outside: the weight gradients (row reductions over all
// nodes: gemm16_tn on the bf16 d(gi) / d(gh) this kernel writes) and the final bias-gradient reduction.
//
// The step-by-step backward (grux.hip) is 3 gate kernels + 2 hidden-state GEMMs + the d x GEMM, ~100 us of latency-bound
// launches.  Per node the chain d h_t -> d(gi_t), d(gh_t) -> d h_{t-1} = d h_t z + d(gh_t) W_hh, d x_t = d(gi_t) W_ih is
// independent, so a workgroup OWNS 32 nodes and walks t = k - 1 .. 0:
//   E  gate derivatives, one thread per (node, 4 columns), 16-byte coalesced reads of the saved gates / h_{t-1}; d h_t comes
//      from LDS (fp32 [32, d]; 0.5 d out at the last step).  Results: the bf16 A operands of the two products in LDS
//      (d(gi) and d(gh) share their r / z parts: tiles RZ [32, 2 d], N_i [32, d], N_h [32, d], 16-B pieces XOR-swizzled by
//      row & 15), the same values as rows of dGI16 / dGH16 in HBM (operands of the weight-gradient GEMM), the direct term
//      d h_t z into the LDS d h tile, and running column sums (bias gradients);
//   G  wave w owns the output columns [w d/4, (w + 1) d/4) of BOTH products (2 JB blocks each): 3 d / 16 k-steps, the B
//      fragments (W_ih, W_hh in fragment-major order for this product, srec_gru_wfrag mode 1) streamed from L2 through a
//      register ring exactly as in the forward;
//   S  d x_t = 0.5 d out / k + d(gi_t) W_ih leaves through a per-wave LDS patch as 16-byte row stores; d(gh_t) W_hh is added
//      onto the direct term in the LDS d h tile (ds_add_f32) for the next step.
// Bias gradients: per-thread column sums over its rows and all steps, reduced over the row groups through LDS, one row
// [6 d] (d b_ih | d b_hh) per workgroup into bias_part (summed by srec_gru_bias_final).
#include "common.h"
#include "../../include/srec_hg.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int GB_MAXP = SREC_GRU_MAXP;
constexpr int RT = 32;             // nodes per workgroup
constexpr int NS = 4, PF = NS - 1; // register ring: stages, k-steps of B fragments in flight

struct BwdArgs {
    srec_gru_fused_bwd_desc d;
    int start[GB_MAXP + 1];
    int wide;                      // mixed launch: bit p = problem p runs in 32-node workgroups (gruf.hip)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

#ifdef SREC_GRUF_TIMING   // development probe (tools/gruf_timing.py): phase clocks of wave 0 of one workgroup, workgroup lives
__device__ unsigned long long g_grub_tim[16];
__device__ unsigned long long g_grub_blk[1024][2];
#define GBT(i) do { __builtin_amdgcn_sched_barrier(0); if (tim_on) tim_t[i] += __builtin_readcyclecounter() - tim_c; \
    tim_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GBT(i)
#endif

// NR: nodes per workgroup, 32 or 16 (16: the lower half of every 32-row MFMA tile idles - see gruf.hip)
// NW: waves per workgroup, 4 or 8 (D = 256; each wave then owns d / 8 = 32 output columns of both products - see gruf.hip)
template <int DD, int NR, int NW>
__device__ __forceinline__ void gru_fused_bwd_body(const BwdArgs& a) {
    constexpr int D = 128 * DD, JB = D / (32 * NW), NT = 64 * NW;   // JB: 32-column blocks per wave
    constexpr int KS = D / 16, TPR = D / 4, RPP = NT / TPR, NP = NR / RPP;
    static_assert(NP >= 2 && NP % 2 == 0, "phase E fetches its rows in two halves");
    constexpr int NRR = NR / 2;                  // accumulator registers per block that hold live nodes
    constexpr int PS = 32 * JB + 8;              // patch row stride (floats): rows r, r + 4 land in opposite bank halves
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* rz = sm;                     // [RT][2 D] bf16, swizzled
    unsigned short* ni = sm + RT * 2 * D;        // [RT][D]
    unsigned short* nh = ni + RT * D;            // [RT][D]
    float* dht = reinterpret_cast<float*>(nh + RT * D);         // [RT][DS] fp32: d(gh_{t+1}) W_hh, the product part of d h_t
    constexpr int DS = D + 8;                    // d h tile row stride (floats): rows r, r + 4 in opposite bank halves
    float* patches = dht + RT * DS;              // [NW waves][RT][PS]
    const srec_gru_fused_bwd_desc& q = a.d;
    int p = 0;
#pragma unroll
    for (int i = 1; i < GB_MAXP; ++i)
        if (i < q.np && (int)blockIdx.x >= a.start[i]) p = i;
    const int n = q.n[p], k = q.k[p];
    const int tile = (int)blockIdx.x - a.start[p];
    const int node0 = tile * NR;
    if (node0 >= n) return;
    const int nl = dyn_count(q.dyn[p], n);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const _Float16* gates = (const _Float16*)q.gates[p];        // saved as fp16 by the fused forward (values in [-1, 1] + gh_n)
    const float* H = q.H[p];
    const float* dout = q.dout[p];
    unsigned short* dGI16 = (unsigned short*)q.dGI16[p];
    unsigned short* dGH16 = (unsigned short*)q.dGH16[p];
    float* dX = q.dX[p];
    // (mixed launch, a.wide: the caller sized bias_part with one row per 16 nodes - this workgroup's sums go to row 2 tile and
    //  row 2 tile + 1, when the problem has it, is zero)
    const bool twin = NR == 32 && ((a.wide >> p) & 1);
    float* part = q.bias_part[p] + (size_t)(q.part_row0[p] + (twin ? 2 * tile : tile)) * 6 * D;
    if (twin && (2 * tile + 1) * 16 < n)
        for (int i = threadIdx.x; i < 6 * D; i += NT) part[6 * D + i] = 0.f;
    const int erow = tid / TPR, ec = (tid % TPR) * 4;           // phase E: this thread's row (per pass) and 4 columns

    if (node0 >= nl) {                           // capacity padding: zero operands and gradients, no arithmetic
        const int rows = min(NR, n - node0);
        for (int i = tid; i < rows * TPR; i += NT) {
            const int row = i / TPR, c = (i % TPR) * 4;
            const size_t node = (size_t)(node0 + row);
            for (int t = 0; t < k; ++t) {
                for (int g = 0; g < 3; ++g) {
                    *reinterpret_cast<uint2*>(dGI16 + (node * k + t) * 3 * D + g * D + c) = make_uint2(0u, 0u);
                    if (t > 0) *reinterpret_cast<uint2*>(dGH16 + ((size_t)(t - 1) * n + node) * 3 * D + g * D + c) = make_uint2(0u, 0u);
                }
                *reinterpret_cast<float4*>(dX + (node * k + t) * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int i = tid; i < 6 * D; i += NT) part[i] = 0.f;
        return;
    }

#ifdef SREC_GRUF_TIMING
    const bool tim_on = (int)blockIdx.x == a.start[a.d.np - 1] + 1;     // a workgroup of the last (longest) order
    unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_c = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_grub_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    float* patch = patches + wave * RT * PS;
    const int cbase = wave * 32 * JB;
    const unsigned short* wf_ih = (const unsigned short*)q.Wih_f[p] + (size_t)wave * 3 * KS * JB * 512;
    const unsigned short* wf_hh = (const unsigned short*)q.Whh_f[p] + (size_t)wave * 3 * KS * JB * 512;
    const bool full = node0 + NR <= nl;
    const float ik = 0.5f / (float)k;
    float si[12], shn[4];                        // column sums: d(gi) r, z, n and the n part of d(gh), this thread's 4 columns
#pragma unroll
    for (int e = 0; e < 12; ++e) si[e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) shn[e] = 0.f;

    // inputs of phase E (saved gates, h_{t-1}): the first half of the rows is fetched one step ahead, behind the d x store of
    // the step before (all of them would not fit the register file next to the products' operand ring); the
    // direct term d h_t z stays in this thread's registers (phase E of step t - 1 runs on the same (node, columns))
    uint2 pr[NP], pz[NP], pn[NP], phn[NP];      // 4 fp16 each
    float4 php[NP], dhz[NP];
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    auto h4f = [](uint2 u) {
        const h4_t v = __builtin_bit_cast(h4_t, u);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    };
    auto fetch = [&](int t, auto LO, auto HI) {
#pragma unroll
        for (int i = decltype(LO)::value; i < decltype(HI)::value; ++i) {
            const int node = node0 + i * RPP + erow;
            pr[i] = pz[i] = pn[i] = phn[i] = make_uint2(0u, 0u);
            php[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (full || node < nl) {
                const _Float16* g = gates + ((size_t)t * n + node) * 4 * D + ec;
                pr[i] = *reinterpret_cast<const uint2*>(g); pz[i] = *reinterpret_cast<const uint2*>(g + D);
                pn[i] = *reinterpret_cast<const uint2*>(g + 2 * D); phn[i] = *reinterpret_cast<const uint2*>(g + 3 * D);
                if (t > 0) php[i] = ld4(H + ((size_t)(t - 1) * n + node) * D + ec);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) dhz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, NP / 2>;
    using IN = std::integral_constant<int, NP>;
    fetch(k - 1, I0{}, IH{});
    constexpr int QPR = 8 * JB;                  // float4 per patch row
    constexpr int NQ = NR * QPR / 64;

    for (int t = k - 1; t >= 0; --t) {
        // (re-derived behind an opaque asm every step: keeps the per-element addresses from being hoisted and spilled)
        int tid_v = tid;
        asm volatile("" : "+v"(tid_v));
        const int er = tid_v / TPR, c = (tid_v % TPR) * 4;
        const bool has_h = t > 0;
        unsigned short* dGI16t = dGI16 + (size_t)t * 3 * D;                    // uniform bases + 32-bit element offsets
        unsigned short* dGH16t = dGH16 + (size_t)(has_h ? t - 1 : 0) * n * 3 * D;
        float* dXt = dX + (size_t)t * D;
        // ---- E: gate derivatives (second half of the rows: fetched now, consumed behind the first half)
        fetch(t, IH{}, IN{});
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = i * RPP + er;
            const int node = node0 + row;
            float4 dpr = make_float4(0.f, 0.f, 0.f, 0.f), dpz = dpr, dpn = dpr, dgn = dpr;
            if (full || node < nl) {
                const float4 r = h4f(pr[i]), z = h4f(pz[i]), nn = h4f(pn[i]), hn = h4f(phn[i]), hp = php[i];
                float4 dh;
                if (t == k - 1) {
                    const float4 go = ld4(dout + (size_t)node * D + c);
                    dh = make_float4(0.5f * go.x, 0.5f * go.y, 0.5f * go.z, 0.5f * go.w);
                } else {
                    const float4 pd = *reinterpret_cast<const float4*>(dht + row * DS + c);
                    dh = make_float4(dhz[i].x + pd.x, dhz[i].y + pd.y, dhz[i].z + pd.z, dhz[i].w + pd.w);
                }
#define GB_LANE(e)                                                                 \
    {                                                                              \
        const float dn = dh.e * (1.f - z.e), dz = dh.e * (hp.e - nn.e);            \
        dpn.e = dn * (1.f - nn.e * nn.e);                                          \
        dpr.e = dpn.e * hn.e * r.e * (1.f - r.e);                                  \
        dpz.e = dz * z.e * (1.f - z.e);                                            \
        dgn.e = dpn.e * r.e;                                                       \
        dhz[i].e = dh.e * z.e;                                                     \
    }
                GB_LANE(x) GB_LANE(y) GB_LANE(z) GB_LANE(w)
#undef GB_LANE
            }
            const uint2 br = make_uint2(srec_pack_bf16(dpr.x, dpr.y), srec_pack_bf16(dpr.z, dpr.w));
            const uint2 bz = make_uint2(srec_pack_bf16(dpz.x, dpz.y), srec_pack_bf16(dpz.z, dpz.w));
            const uint2 bn = make_uint2(srec_pack_bf16(dpn.x, dpn.y), srec_pack_bf16(dpn.z, dpn.w));
            const uint2 bg = make_uint2(srec_pack_bf16(dgn.x, dgn.y), srec_pack_bf16(dgn.z, dgn.w));
            const int sw = row & 15, pc = c >> 3, ho = c & 4;
            *reinterpret_cast<uint2*>(rz + row * 2 * D + ((pc ^ sw) * 8) + ho) = br;
            *reinterpret_cast<uint2*>(rz + row * 2 * D + (((D / 8 + pc) ^ sw) * 8) + ho) = bz;
            *reinterpret_cast<uint2*>(ni + row * D + ((pc ^ sw) * 8) + ho) = bn;
            if (has_h) *reinterpret_cast<uint2*>(nh + row * D + ((pc ^ sw) * 8) + ho) = bg;
            if (full || node < n) {
                const unsigned go = (unsigned)node * k * 3 * D + c;
                *reinterpret_cast<uint2*>(dGI16t + go) = br;
                *reinterpret_cast<uint2*>(dGI16t + go + D) = bz;
                *reinterpret_cast<uint2*>(dGI16t + go + 2 * D) = bn;
                if (has_h) {
                    const unsigned ho2 = (unsigned)node * 3 * D + c;
                    *reinterpret_cast<uint2*>(dGH16t + ho2) = br;
                    *reinterpret_cast<uint2*>(dGH16t + ho2 + D) = bz;
                    *reinterpret_cast<uint2*>(dGH16t + ho2 + 2 * D) = bg;
                }
            }
            si[0] += dpr.x; si[1] += dpr.y; si[2] += dpr.z; si[3] += dpr.w;
            si[4] += dpz.x; si[5] += dpz.y; si[6] += dpz.z; si[7] += dpz.w;
            si[8] += dpn.x; si[9] += dpn.y; si[10] += dpn.z; si[11] += dpn.w;
            shn[0] += dgn.x; shn[1] += dgn.y; shn[2] += dgn.z; shn[3] += dgn.w;
        }
        GBT(0);
        __syncthreads();                         // A tiles and the direct term of d h_{t-1} published
        GBT(1);

        // ---- G: d x_t = d(gi_t) W_ih, d h_{t-1} += d(gh_t) W_hh; this wave's 32 JB output columns of both
        float4 gmean[NQ];                        // d out of this wave's columns in the d x store layout (mean term 0.5 d out / k):
#pragma unroll                                   // fetched here, ahead of the next step's gate inputs (loads return in order)
        for (int i = 0; i < NQ; ++i) {
            const int idx = i * 64 + (tid_v & 63);
            const int node = node0 + idx / QPR, c4 = (idx % QPR) * 4;
            gmean[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (full || node < nl) gmean[i] = ld4(dout + (unsigned)node * D + cbase + c4);
        }
        f32x16 ax[JB], ah[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ax[j][r] = ah[j][r] = 0.f;
        auto products = [&](auto HH) {
            constexpr bool HAS_H = decltype(HH)::value;
            constexpr int NF = HAS_H ? 2 * JB : JB;
            constexpr int T = 3 * KS;
            bf16x8 Bq[NS][NF];
            auto load = [&](int i, int slot) {
                i = min(i, T - 1);
                const unsigned short* s0 = wf_ih + (size_t)i * JB * 512 + lane * 8;
                const unsigned short* s1 = wf_hh + (size_t)i * JB * 512 + lane * 8;
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    Bq[slot][j] = *reinterpret_cast<const bf16x8*>(s0 + j * 512);
                    if (HAS_H) Bq[slot][JB + j] = *reinterpret_cast<const bf16x8*>(s1 + j * 512);
                }
            };
#pragma unroll
            for (int i = 0; i < PF; ++i) load(i, i);
#pragma unroll 1
            for (int ib = 0; ib < T; ib += NS) {
                const bool npart = ib >= 2 * KS; // k-steps 0 .. 2 KS - 1: the shared r / z part; then the n parts
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    load(ib + u + PF, (u + PF) % NS);
                    const int s = ib + u;
                    bf16x8 Ai, Ah;
                    if (!npart) {
                        Ai = *reinterpret_cast<const bf16x8*>(rz + l31 * 2 * D + (((2 * s + half) ^ (l31 & 15)) * 8));
                        Ah = Ai;
                    } else {
                        const int s2 = s - 2 * KS;
                        Ai = *reinterpret_cast<const bf16x8*>(ni + l31 * D + (((2 * s2 + half) ^ (l31 & 15)) * 8));
                        if (HAS_H) Ah = *reinterpret_cast<const bf16x8*>(nh + l31 * D + (((2 * s2 + half) ^ (l31 & 15)) * 8));
                    }
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        ax[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ai, Bq[u][j], ax[j], 0, 0, 0);
                        if (HAS_H) ah[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bq[u][JB + j], ah[j], 0, 0, 0);
                    }
                    // keep the issue order: next stage's loads, this k-step's A reads, its MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x020, NF, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                }
            }
        };
        if (has_h) products(std::true_type{});
        else products(std::false_type{});

        GBT(2);
        // ---- S: next step's gate inputs on their way; d x_t rows out through the wave's patch (16-byte stores);
        //      d(gh_t) W_hh into the LDS d h tile (plain stores: the direct term waits in registers)
        if (has_h) fetch(t - 1, I0{}, IH{});
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int l31v = lane_v & 31, halfv = lane_v >> 5;
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < NRR; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                patch[row * PS + 32 * j + l31v] = ax[j][r];
                if (has_h) dht[row * DS + cbase + 32 * j + l31v] = ah[j][r];
            }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int idx = i * 64 + lane_v;
            const int row = idx / QPR, c4 = (idx % QPR) * 4;
            const int node = node0 + row;
            if (full || node < n) {
                float4 v = *reinterpret_cast<const float4*>(patch + row * PS + c4);
                if (full || node < nl) { v.x += ik * gmean[i].x; v.y += ik * gmean[i].y; v.z += ik * gmean[i].z; v.w += ik * gmean[i].w; }
                else v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(dXt + (unsigned)node * k * D + cbase + c4) = v;
            }
        }
        GBT(3);
        __syncthreads();                         // A tiles free again; d h_{t-1} complete
        GBT(4);
    }

    // ---- bias gradients: column sums over the row groups
    float* red = reinterpret_cast<float*>(sm);   // [RPP][16][TPR] floats (<= 16 KiB), the A tiles are dead
#pragma unroll
    for (int e = 0; e < 12; ++e) red[(erow * 16 + e) * TPR + tid % TPR] = si[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(erow * 16 + 12 + e) * TPR + tid % TPR] = shn[e];
    __syncthreads();
    for (int o = tid; o < 6 * D; o += NT) {
        // o = half * 3 D + gate * D + col; d(gh) shares its r / z sums with d(gi)
        const int hh = o / (3 * D), gate = (o % (3 * D)) / D, col = o % D;
        const int e = (hh == 1 && gate == 2 ? 12 : 4 * gate) + (col & 3);
        float s = 0.f;
#pragma unroll
        for (int rg = 0; rg < RPP; ++rg) s += red[(rg * 16 + e) * TPR + (col >> 2)];
        part[o] = s;
    }
    (void)ec;
#ifdef SREC_GRUF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GBT(5);
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_grub_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    if (tim_on && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_grub_tim[i] = tim_t[i];
#endif
}


// streaming accesses of the 16-node kernel: every input row is read once and every output row written once per launch - marked
// non-temporal so that they do not push the weight fragments (re-read by every workgroup of the XCD) out of L2
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 ldnt2(const void* p) {
    const u2_t v = __builtin_nontemporal_load(reinterpret_cast<const u2_t*>(p));
    return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ float4 ldnt4(const float* p) {
    const f4_t v = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stnt2(void* p, uint2 v) {
    const u2_t w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<u2_t*>(p));
}
__device__ __forceinline__ void stnt4(float* p, float4 v) {
    const f4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f4_t*>(p));
}

// ---- 16-node workgroups, k <= 4: d x for ALL time steps as one product behind the recurrence ---------------------------------
// The step loop above streams W_ih AND W_hh through every wave once per time step (2 k - 1 passes over 1.5 d^2 bf16 each) for
// MFMA tiles whose lower half idles.  d x_t = d(gi_t) W_ih is not part of the recurrence: here the loop keeps only
// d h_{t-1} += d(gh_t) W_hh (k - 1 passes over W_hh), phase E leaves d(gi_t) of every step in LDS (tile row = 16 t + node: full
// 32-row MFMA tiles, [16 k][3 d] bf16; d(gh_t) shares its r / z columns), and ONE pass over W_ih behind the loop produces the
// d x rows of all steps: k passes over the weights instead of 2 k - 1, (k - 1) + ceil(k / 2) MFMA tiles per k-step instead of
// 2 k - 1.  The patches of the d x store and the bias reduction reuse the d(gi) tiles once they are dead.
template <int DD, int NW>
__device__ __forceinline__ void gru_fused_bwd16_body(const BwdArgs& a) {
    constexpr int NR = 16, KMAX = 4;
    constexpr int D = 128 * DD, JB = D / (32 * NW), NT = 64 * NW;
    constexpr int KS = D / 16, TPR = D / 4, RPP = NT / TPR, NP = NR / RPP;
    static_assert(NP >= 2 && NP % 2 == 0, "phase E fetches its rows in two halves");
    constexpr int NRR = NR / 2;
    constexpr int PS = 32 * JB + 8;
    constexpr int DS = D + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* gi = sm;                                     // [16 KMAX][3 D] bf16, swizzled: d(gi_t) rows of node r at 16 t + r
    unsigned short* nh = gi + 16 * KMAX * 3 * D;                 // [16][D]: the n part of d(gh_t)
    float* dht = reinterpret_cast<float*>(nh + NR * D);          // [16][DS] fp32: d(gh_{t+1}) W_hh
    float* patches = reinterpret_cast<float*>(sm);               // [NW][32][PS], over the dead d(gi) tiles
    const srec_gru_fused_bwd_desc& q = a.d;
    int p = 0;
#pragma unroll
    for (int i = 1; i < GB_MAXP; ++i)
        if (i < q.np && (int)blockIdx.x >= a.start[i]) p = i;
    const int n = q.n[p], k = q.k[p];
    const int tile = (int)blockIdx.x - a.start[p];
    const int node0 = tile * NR;
    if (node0 >= n) return;
    const int nl = dyn_count(q.dyn[p], n);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const _Float16* gates = (const _Float16*)q.gates[p];
    const float* H = q.H[p];
    const float* dout = q.dout[p];
    unsigned short* dGI16 = (unsigned short*)q.dGI16[p];
    unsigned short* dGH16 = (unsigned short*)q.dGH16[p];
    float* dX = q.dX[p];
    float* part = q.bias_part[p] + (size_t)(q.part_row0[p] + tile) * 6 * D;
    const int erow = tid / TPR, ec = (tid % TPR) * 4;

    if (node0 >= nl) {                           // capacity padding: zero operands and gradients, no arithmetic
        const int rows = min(NR, n - node0);
        for (int i = tid; i < rows * TPR; i += NT) {
            const int row = i / TPR, c = (i % TPR) * 4;
            const size_t node = (size_t)(node0 + row);
            for (int t = 0; t < k; ++t) {
                for (int g = 0; g < 3; ++g) {
                    *reinterpret_cast<uint2*>(dGI16 + (node * k + t) * 3 * D + g * D + c) = make_uint2(0u, 0u);
                    if (t > 0) *reinterpret_cast<uint2*>(dGH16 + ((size_t)(t - 1) * n + node) * 3 * D + g * D + c) = make_uint2(0u, 0u);
                }
                *reinterpret_cast<float4*>(dX + (node * k + t) * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int i = tid; i < 6 * D; i += NT) part[i] = 0.f;
        return;
    }

#ifdef SREC_GRUF_TIMING
    const bool tim_on = (int)blockIdx.x == a.start[a.d.np - 1] + 1;
    unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_c = __builtin_readcyclecounter();
    const unsigned long long tim_r0 = __builtin_amdgcn_s_memrealtime(), tim_c0 = tim_c;
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_grub_blk[blockIdx.x][0] = tim_r0;
#endif
    const int cbase = wave * 32 * JB;
    const unsigned short* wf_ih = (const unsigned short*)q.Wih_f[p] + (size_t)wave * 3 * KS * JB * 512;
    const unsigned short* wf_hh = (const unsigned short*)q.Whh_f[p] + (size_t)wave * 3 * KS * JB * 512;
    const bool full = node0 + NR <= nl;
    const float ik = 0.5f / (float)k;
    float si[12], shn[4];
#pragma unroll
    for (int e = 0; e < 12; ++e) si[e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) shn[e] = 0.f;

    uint2 pr[NP], pz[NP], pn[NP], phn[NP];      // 4 fp16 each
    float4 php[NP], dhz[NP];
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    auto h4f = [](uint2 u) {
        const h4_t v = __builtin_bit_cast(h4_t, u);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    };
    auto fetch = [&](int t, auto LO, auto HI) {
#pragma unroll
        for (int i = decltype(LO)::value; i < decltype(HI)::value; ++i) {
            const int node = node0 + i * RPP + erow;
            pr[i] = pz[i] = pn[i] = phn[i] = make_uint2(0u, 0u);
            php[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (full || node < nl) {
                const _Float16* g = gates + ((size_t)t * n + node) * 4 * D + ec;
                pr[i] = ldnt2(g); pz[i] = ldnt2(g + D);
                pn[i] = ldnt2(g + 2 * D); phn[i] = ldnt2(g + 3 * D);
                if (t > 0) php[i] = ldnt4(H + ((size_t)(t - 1) * n + node) * D + ec);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) dhz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    using I0 = std::integral_constant<int, 0>;
    using IN = std::integral_constant<int, NP>;
    // inputs of phase E (saved gates, h_{t-1}) for ALL rows one step ahead, requested in front of the W_hh product that hides them
    // (the step's copy of this kernel finds none of its inputs in a cache: 54 us against 31 us for a warm loop, 41 us for an
    // immediate second launch - profiles/r05_notes.md)
    fetch(k - 1, I0{}, IN{});

    for (int t = k - 1; t >= 0; --t) {
        int tid_v = tid;
        asm volatile("" : "+v"(tid_v));
        const int er = tid_v / TPR, c = (tid_v % TPR) * 4;
        const bool has_h = t > 0;
        unsigned short* dGI16t = dGI16 + (size_t)t * 3 * D;
        unsigned short* dGH16t = dGH16 + (size_t)(has_h ? t - 1 : 0) * n * 3 * D;
        unsigned short* git = gi + (size_t)t * 16 * 3 * D;
        // ---- E: gate derivatives
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = i * RPP + er;
            const int node = node0 + row;
            float4 dpr = make_float4(0.f, 0.f, 0.f, 0.f), dpz = dpr, dpn = dpr, dgn = dpr;
            if (full || node < nl) {
                const float4 r = h4f(pr[i]), z = h4f(pz[i]), nn = h4f(pn[i]), hn = h4f(phn[i]), hp = php[i];
                float4 dh;
                if (t == k - 1) {
                    const float4 go = ld4(dout + (size_t)node * D + c);
                    dh = make_float4(0.5f * go.x, 0.5f * go.y, 0.5f * go.z, 0.5f * go.w);
                } else {
                    const float4 pd = *reinterpret_cast<const float4*>(dht + row * DS + c);
                    dh = make_float4(dhz[i].x + pd.x, dhz[i].y + pd.y, dhz[i].z + pd.z, dhz[i].w + pd.w);
                }
#define GB_LANE(e)                                                                 \
    {                                                                              \
        const float dn = dh.e * (1.f - z.e), dz = dh.e * (hp.e - nn.e);            \
        dpn.e = dn * (1.f - nn.e * nn.e);                                          \
        dpr.e = dpn.e * hn.e * r.e * (1.f - r.e);                                  \
        dpz.e = dz * z.e * (1.f - z.e);                                            \
        dgn.e = dpn.e * r.e;                                                       \
        dhz[i].e = dh.e * z.e;                                                     \
    }
                GB_LANE(x) GB_LANE(y) GB_LANE(z) GB_LANE(w)
#undef GB_LANE
            }
            const uint2 br = make_uint2(srec_pack_bf16(dpr.x, dpr.y), srec_pack_bf16(dpr.z, dpr.w));
            const uint2 bz = make_uint2(srec_pack_bf16(dpz.x, dpz.y), srec_pack_bf16(dpz.z, dpz.w));
            const uint2 bn = make_uint2(srec_pack_bf16(dpn.x, dpn.y), srec_pack_bf16(dpn.z, dpn.w));
            const uint2 bg = make_uint2(srec_pack_bf16(dgn.x, dgn.y), srec_pack_bf16(dgn.z, dgn.w));
            const int sw = row & 15, pc = c >> 3, ho = c & 4;
            *reinterpret_cast<uint2*>(git + row * 3 * D + ((pc ^ sw) * 8) + ho) = br;
            *reinterpret_cast<uint2*>(git + row * 3 * D + (((D / 8 + pc) ^ sw) * 8) + ho) = bz;
            *reinterpret_cast<uint2*>(git + row * 3 * D + (((2 * D / 8 + pc) ^ sw) * 8) + ho) = bn;
            if (has_h) *reinterpret_cast<uint2*>(nh + row * D + ((pc ^ sw) * 8) + ho) = bg;
            if (full || node < n) {
                const unsigned go = (unsigned)node * k * 3 * D + c;
                stnt2(dGI16t + go, br);
                stnt2(dGI16t + go + D, bz);
                stnt2(dGI16t + go + 2 * D, bn);
                if (has_h) {
                    const unsigned ho2 = (unsigned)node * 3 * D + c;
                    stnt2(dGH16t + ho2, br);
                    stnt2(dGH16t + ho2 + D, bz);
                    stnt2(dGH16t + ho2 + 2 * D, bg);
                }
            }
            si[0] += dpr.x; si[1] += dpr.y; si[2] += dpr.z; si[3] += dpr.w;
            si[4] += dpz.x; si[5] += dpz.y; si[6] += dpz.z; si[7] += dpz.w;
            si[8] += dpn.x; si[9] += dpn.y; si[10] += dpn.z; si[11] += dpn.w;
            shn[0] += dgn.x; shn[1] += dgn.y; shn[2] += dgn.z; shn[3] += dgn.w;
        }
        GBT(0);
        __syncthreads();                         // this step's A rows and the direct term of d h_{t-1} published
        GBT(1);
        if (!has_h) break;

        // ---- G: d h_{t-1} += d(gh_t) W_hh; this wave's 32 JB output columns (tile rows 16 .. 31 repeat 0 .. 15: ignored)
        fetch(t - 1, I0{}, IN{});                 // next step's gate inputs on their way
        f32x16 ah[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ah[j][r] = 0.f;
        {
            constexpr int T = 3 * KS;
            bf16x8 Bq[NS][JB];
            auto load = [&](int i, int slot) {
                i = min(i, T - 1);
                const unsigned short* s1 = wf_hh + (size_t)i * JB * 512 + lane * 8;
#pragma unroll
                for (int j = 0; j < JB; ++j) Bq[slot][j] = *reinterpret_cast<const bf16x8*>(s1 + j * 512);
            };
#pragma unroll
            for (int i = 0; i < PF; ++i) load(i, i);
            const unsigned short* arow = git + l15 * 3 * D;
            const unsigned short* nrow = nh + l15 * D;
#pragma unroll 1
            for (int ib = 0; ib < T; ib += NS) {
                const bool npart = ib >= 2 * KS;
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    load(ib + u + PF, (u + PF) % NS);
                    const int s = ib + u;
                    bf16x8 Ah;
                    if (!npart) Ah = *reinterpret_cast<const bf16x8*>(arow + (((2 * s + half) ^ l15) * 8));
                    else Ah = *reinterpret_cast<const bf16x8*>(nrow + (((2 * (s - 2 * KS) + half) ^ l15) * 8));
#pragma unroll
                    for (int j = 0; j < JB; ++j) ah[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bq[u][j], ah[j], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, JB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, JB, 0);
                }
            }
        }
        GBT(2);
        // ---- S: d(gh_t) W_hh into the LDS d h tile
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int l31v = lane_v & 31, halfv = lane_v >> 5;
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < NRR; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                dht[row * DS + cbase + 32 * j + l31v] = ah[j][r];
            }
        GBT(3);
        __syncthreads();                         // d h_{t-1} complete; nh free again
        GBT(4);
    }

    // ---- X: d x rows of all steps: [16 k, 3 d] x W_ih, one pass over the weight fragments
    const int ntb = (16 * k + 31) / 32;          // 32-row tiles (1 or 2)
    f32x16 ax[2][JB];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ax[b][j][r] = 0.f;
    auto xprod = [&](auto TWO) {
        constexpr bool TW = decltype(TWO)::value;
        constexpr int T = 3 * KS;
        bf16x8 Bq[NS][JB];
        auto load = [&](int i, int slot) {
            i = min(i, T - 1);
            const unsigned short* s0 = wf_ih + (size_t)i * JB * 512 + lane * 8;
#pragma unroll
            for (int j = 0; j < JB; ++j) Bq[slot][j] = *reinterpret_cast<const bf16x8*>(s0 + j * 512);
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) load(i, i);
        const int R0 = min(l31, 16 * k - 1), R1 = min(32 + l31, 16 * k - 1);     // rows past the last step repeat it: ignored
        const unsigned short* a0 = gi + (size_t)R0 * 3 * D;
        const unsigned short* a1 = gi + (size_t)R1 * 3 * D;
        const int x0 = R0 & 15, x1 = R1 & 15;
#pragma unroll 1
        for (int ib = 0; ib < T; ib += NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                load(ib + u + PF, (u + PF) % NS);
                const int s = ib + u;
                const bf16x8 A0 = *reinterpret_cast<const bf16x8*>(a0 + (((2 * s + half) ^ x0) * 8));
                bf16x8 A1;
                if (TW) A1 = *reinterpret_cast<const bf16x8*>(a1 + (((2 * s + half) ^ x1) * 8));
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    ax[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bq[u][j], ax[0][j], 0, 0, 0);
                    if (TW) ax[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bq[u][j], ax[1][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x020, JB, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TW ? 2 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TW ? 2 * JB : JB, 0);
            }
        }
    };
    // the mean term 0.5 d out / k of this wave's columns in the store layout: patch row r belongs to node r & 15 in either tile;
    // requested here, it arrives behind the product
    constexpr int XQPR = 8 * JB, XNQ = 32 * XQPR / 64;
    float4 gmean[XNQ];
#pragma unroll
    for (int i = 0; i < XNQ; ++i) {
        const int idx = i * 64 + lane;
        const int node = node0 + ((idx / XQPR) & 15), c4 = (idx % XQPR) * 4;
        gmean[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (full || node < nl) gmean[i] = ld4(dout + (unsigned)node * D + cbase + c4);
    }
    if (ntb == 2) xprod(std::true_type{});
    else xprod(std::false_type{});
    GBT(5);
    __syncthreads();                             // every wave is done with the d(gi) tiles: they become the store patches
    {
        float* patch = patches + wave * 32 * PS;
        constexpr int QPR = 8 * JB;              // float4 per patch row
        constexpr int NQ = 32 * QPR / 64;
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int l31v = lane_v & 31, halfv = lane_v >> 5;
        for (int b = 0; b < ntb; ++b) {
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                    patch[row * PS + 32 * j + l31v] = b == 0 ? ax[0][j][r] : ax[1][j][r];
                }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int idx = i * 64 + lane_v;
                const int prow = idx / QPR, c4 = (idx % QPR) * 4;
                const int R = 32 * b + prow, t = R >> 4, node = node0 + (R & 15);
                if (t < k && (full || node < n)) {
                    float4 v = *reinterpret_cast<const float4*>(patch + prow * PS + c4);
                    if (full || node < nl) { v.x += ik * gmean[i].x; v.y += ik * gmean[i].y; v.z += ik * gmean[i].z; v.w += ik * gmean[i].w; }
                    else v = make_float4(0.f, 0.f, 0.f, 0.f);
                    stnt4(dX + ((size_t)node * k + t) * D + cbase + c4, v);
                }
            }
        }
    }
    __syncthreads();                             // patches read: the bias reduction takes the same LDS

    // ---- bias gradients: column sums over the row groups
    float* red = reinterpret_cast<float*>(sm);   // [RPP][16][TPR] floats
#pragma unroll
    for (int e = 0; e < 12; ++e) red[(erow * 16 + e) * TPR + tid % TPR] = si[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(erow * 16 + 12 + e) * TPR + tid % TPR] = shn[e];
    __syncthreads();
    for (int o = tid; o < 6 * D; o += NT) {
        const int hh = o / (3 * D), gate = (o % (3 * D)) / D, col = o % D;
        const int e = (hh == 1 && gate == 2 ? 12 : 4 * gate) + (col & 3);
        float s = 0.f;
#pragma unroll
        for (int rg = 0; rg < RPP; ++rg) s += red[(rg * 16 + e) * TPR + (col >> 2)];
        part[o] = s;
    }
    (void)ec; (void)l31;
#ifdef SREC_GRUF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GBT(6);
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_grub_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    if (tim_on && threadIdx.x == 0) {
        for (int i = 0; i < 8; ++i) g_grub_tim[i] = tim_t[i];
        g_grub_tim[8] = __builtin_readcyclecounter() - tim_c0;              // shader clocks ...
        g_grub_tim[9] = __builtin_amdgcn_s_memrealtime() - tim_r0;          // ... over 100 MHz ticks of the same workgroup
    }
#endif
}

}  // namespace
extern "C" int srec_gru_fused_waves(int d, int* waves);
namespace {

template <int DD, int NR, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fused_bwd_kernel(BwdArgs a) { gru_fused_bwd_body<DD, NR, NW>(a); }

template <int DD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fused_bwd16_kernel(BwdArgs a) { gru_fused_bwd16_body<DD, NW>(a); }

// mixed launch (see gru_fused_fwd_mixed_kernel, gruf.hip): the problems of a.wide in 32-node workgroups, the others in 16-node ones
template <int DD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fused_bwd_mixed_kernel(BwdArgs a) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < GB_MAXP; ++i)
        if (i < a.d.np && (int)blockIdx.x >= a.start[i]) p = i;
    if ((a.wide >> p) & 1) gru_fused_bwd_body<DD, 32, NW>(a);
    else gru_fused_bwd16_body<DD, NW>(a);
}

struct WfbArgs {
    int d, jb;
    const float* W[2 * GB_MAXP];
    unsigned short* dst[2 * GB_MAXP];
};

// fragment-major bf16 copy of a GRU weight W [3 d, d] for the backward-data products d(g) W: fragment ((w 3 KS + s) JB + j)
// = the MFMA B operand of wave w, k-step s, output column block j: lane l <- W[16 s + 8 (l >> 5) .. + 7][w d/4 + 32 j + (l & 31)]
__global__ __launch_bounds__(256) void gru_wfrag_t_kernel(WfbArgs a) {
    const int d = a.d, JB = a.jb, KS3 = 3 * d / 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * d * d / 8) return;
    const int lane = idx & 63, frag = idx >> 6;
    const int j = frag % JB, ws = frag / JB, s = ws % KS3, w = ws / KS3;
    const int col = w * 32 * JB + 32 * j + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
    const float* src = a.W[blockIdx.y] + (size_t)kk * d + col;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[(size_t)e * d];
    uint4 o;
    o.x = srec_pack_bf16(v[0], v[1]); o.y = srec_pack_bf16(v[2], v[3]);
    o.z = srec_pack_bf16(v[4], v[5]); o.w = srec_pack_bf16(v[6], v[7]);
    *reinterpret_cast<uint4*>(a.dst[blockIdx.y] + (size_t)idx * 8) = o;
}

}  // namespace

// (srec_gru_wfrag_both - both layouts of the same weights in one launch - is a role of the step's prologue launch: prep.hip)

#ifdef SREC_GRUF_TIMING
extern "C" int srec_grub_timing(unsigned long long* tim16, unsigned long long* blk) {
    if (hipMemcpyFromSymbol(tim16, HIP_SYMBOL(g_grub_tim), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(blk, HIP_SYMBOL(g_grub_blk), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : 1;
}
#endif

// as srec_gru_wfrag, for the backward-data products of srec_gru_fused_bwd (B operand = W itself, k = its rows)
extern "C" int srec_gru_wfrag_t(int n, const void* W, const void* dst, int d, void* stream) {
    if (n <= 0) return 0;
    if (n > 2 * GB_MAXP || W == nullptr || dst == nullptr || (d != 128 && d != 256)) return SREC_BAD_ARG;
    WfbArgs a{};
    int nw = 4;
    if (int rc = srec_gru_fused_waves(d, &nw)) return rc;
    a.d = d; a.jb = d / (32 * nw);
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i]; a.dst[i] = ((unsigned short* const*)dst)[i];
        if (a.W[i] == nullptr || a.dst[i] == nullptr) return SREC_BAD_ARG;
    }
    hipLaunchKernelGGL(gru_wfrag_t_kernel, dim3((3 * d * d / 8 + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// nodes per workgroup (16 or 32) srec_gru_fused_fwd / _bwd use for np problems of n[p] nodes: 16-node workgroups while
// 32-node ones would leave much of the chip idle.  bias_part of srec_gru_fused_bwd holds one row per workgroup:
// sum_p ceil(n[p] / nodes) rows.  (SREC_GRU_NR = 16 / 32: development override.)
extern "C" int srec_gru_fused_waves(int d, int* waves) {
    if (waves == nullptr || (d != 128 && d != 256)) return SREC_BAD_ARG;
    *waves = d == 256 ? 8 : 4;            // (d = 256 with 4 waves measured 0.954 vs 0.942 ms per step: removed)
    return 0;
}

extern "C" int srec_gru_fused_nodes(int np, const int* n, int d, int* nodes) {
    if (np < 0 || np > GB_MAXP || (np > 0 && n == nullptr) || nodes == nullptr) return SREC_BAD_ARG;
    int nw = 4;
    if (int rc = srec_gru_fused_waves(d, &nw)) return rc;
    int blocks = 0;
    for (int p = 0; p < np; ++p) blocks += (n[p] + RT - 1) / RT;
    // measured at the bench shapes (ms per step, one box): 4 waves x 16 nodes 0.954, 8 x 32 0.947, 8 x 16 0.942
    (void)nw;
    *nodes = blocks <= 192 ? 16 : 32;
    return 0;
}

// Mixed launches (16-node workgroups, every problem with k <= 4): which problems - given in LAUNCH order, longest first - take
// 32-node tiles so that the workgroups of one launch are at most one per CU: the shortest first, as many as it takes.  Decided on
// the CAPACITIES (the launch shape of a captured step is fixed): at the bench shape 160 + 160 sixteen-node tiles become 160 + 80.
extern "C" int srec_gru_fused_wide(int np, const int* n, const int* k, int* mask) {
    if (np < 0 || np > GB_MAXP || (np > 0 && (n == nullptr || k == nullptr)) || mask == nullptr) return SREC_BAD_ARG;
    *mask = 0;
    static std::atomic<int> cu_count[64];                        // per device ordinal (the query costs ~0.1 ms: once)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int cus = cu_count[dev & 63].load(std::memory_order_relaxed);
    if (cus == 0) {
        cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cu_count[dev & 63].store(cus, std::memory_order_relaxed);
    }
    static const char* env = getenv("SREC_GRU_MIXED");           // development: 0 = never widen (A / B)
    if (env != nullptr && atoi(env) == 0) return 0;
    for (int p = 0; p < np; ++p)
        if (k[p] > 4) return 0;                                  // (the 16-node kernels of the mixed launch hold <= 4 time steps)
    // ... but never the longest order: its 16-node kernel batches the input projections of all steps (k weight passes), the 32-node
    // one streams 2 k - 1 - a launch with every order in 32-node tiles is slower than a second round of short 16-node ones (the
    // end-to-end loop's capacities, 192 + 192 tiles: 0.783 -> 0.800 ms per batch when both orders widened).  The count is taken on
    // the capacities, the live tiles are fewer.
    int tiles = 0, wide = 0, kmax = 0;
    for (int p = 0; p < np; ++p) { tiles += (n[p] + 15) / 16; kmax = k[p] > kmax ? k[p] : kmax; }
    while (tiles > cus) {
        int best = -1;
        for (int p = 0; p < np; ++p)
            if (!((wide >> p) & 1) && n[p] > 16 && k[p] < kmax && (best < 0 || k[p] < k[best] || (k[p] == k[best] && n[p] > n[best]))) best = p;
        if (best < 0) break;
        wide |= 1 << best;
        tiles -= (n[best] + 15) / 16 - (n[best] + 31) / 32;
    }
    *mask = wide;
    return 0;
}

// Workgroups start in index order and an order-k workgroup lives ~k time steps: with more live workgroups than CUs (one per CU:
// the d(gi) tiles fill most of the LDS) the ones that wait start when the first residents retire.  Longest problems first: the
// late starters are then the SHORT ones, behind the first short residents (measured on the C3 batches whose live 16-node tiles
// exceed 256: 52 us with the order-3 tiles last, ~36 us for the batches that fit one round - profiles/r05_notes.md).
static void longest_first(srec_gru_fused_bwd_desc& d) {
    int ord[GB_MAXP];
    for (int i = 0; i < d.np; ++i) ord[i] = i;
    for (int i = 1; i < d.np; ++i)
        for (int j = i; j > 0 && d.k[ord[j]] > d.k[ord[j - 1]]; --j) { const int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
    const srec_gru_fused_bwd_desc s = d;
    for (int q = 0; q < d.np; ++q) {
        const int p = ord[q];
        d.n[q] = s.n[p]; d.k[q] = s.k[p]; d.dyn[q] = s.dyn[p]; d.gates[q] = s.gates[p]; d.H[q] = s.H[p]; d.dout[q] = s.dout[p];
        d.Wih_f[q] = s.Wih_f[p]; d.Whh_f[q] = s.Whh_f[p]; d.dGI16[q] = s.dGI16[p]; d.dGH16[q] = s.dGH16[p]; d.dX[q] = s.dX[p];
        d.bias_part[q] = s.bias_part[p]; d.part_row0[q] = s.part_row0[p];
    }
}

// desc: HOST srec_gru_fused_bwd_desc (srec_hg.h)
extern "C" int srec_gru_fused_bwd(const void* desc, void* stream) {
    const srec_gru_fused_bwd_desc* q0 = (const srec_gru_fused_bwd_desc*)desc;
    if (q0 == nullptr || q0->np <= 0 || q0->np > GB_MAXP || (q0->d != 128 && q0->d != 256)) return SREC_BAD_ARG;
    BwdArgs a{};
    a.d = *q0;
    longest_first(a.d);
    const srec_gru_fused_bwd_desc* q = &a.d;
    int blocks = 0;
    for (int p = 0; p < q->np; ++p) {
        if (q->n[p] < 0 || q->k[p] < 1 || q->gates[p] == nullptr || q->H[p] == nullptr || q->dout[p] == nullptr ||
            q->Wih_f[p] == nullptr || q->Whh_f[p] == nullptr || q->dGI16[p] == nullptr || q->dGH16[p] == nullptr ||
            q->dX[p] == nullptr || q->bias_part[p] == nullptr)
            return SREC_BAD_ARG;
        a.start[p] = blocks;
        blocks += (q->n[p] + RT - 1) / RT;
    }
    // 16-node workgroups while 32-node ones would leave half of the chip idle (the caller sizes bias_part for either:
    // one row per 16 nodes is always enough)
    int NRv = 32;
    if (int rc = srec_gru_fused_nodes(q->np, q->n, q->d, &NRv)) return rc;
    bool batch = NRv == 16;                      // gru_fused_bwd16_kernel holds the d(gi) rows of <= 4 time steps
    for (int p = 0; p < q->np; ++p) batch = batch && q->k[p] <= 4;
    static const bool no16 = getenv("SREC_GRU_BWD16") != nullptr && atoi(getenv("SREC_GRU_BWD16")) == 0;   // development: A / B
    if (NRv == 16) {
        if (batch && !no16)                                                          // (the same choice as the forward's)
            if (int rc = srec_gru_fused_wide(q->np, q->n, q->k, &a.wide)) return rc;
        blocks = 0;
        for (int p = 0; p < q->np; ++p) { a.start[p] = blocks; blocks += (q->n[p] + (((a.wide >> p) & 1) ? 31 : 15)) / (((a.wide >> p) & 1) ? 32 : 16); }
    }
    for (int p = q->np; p <= GB_MAXP; ++p) a.start[p] = blocks;
    if (blocks == 0) return 0;
    const int D = q->d;
    int NWv = 4;
    if (int rc = srec_gru_fused_waves(D, &NWv)) return rc;
    const int JBw = D / (32 * NWv);
    const size_t lds = (size_t)RT * 4 * D * 2 + (size_t)RT * (D + 8) * 4 + (size_t)NWv * RT * (32 * JBw + 8) * 4;
    static std::atomic<unsigned long long> om[6];
#define SREC_GB(DDV, NRV, NWV, slot)                                                                                   \
    do {                                                                                                               \
        if (int rc = srec_lds_optin((const void*)gru_fused_bwd_kernel<DDV, NRV, NWV>, (int)lds, om[slot])) return rc;  \
        hipLaunchKernelGGL((gru_fused_bwd_kernel<DDV, NRV, NWV>), dim3(blocks), dim3(64 * NWV), lds, (hipStream_t)stream, a); \
    } while (0)
    if (batch && !no16 && a.wide) {
        const size_t lds16 = (size_t)(16 * 4 * 3 * D) * 2 + (size_t)16 * D * 2 + (size_t)16 * (D + 8) * 4;
        const size_t ldsm = lds16 > lds ? lds16 : lds;
        static std::atomic<unsigned long long> omm[2];
        if (D == 256) {
            if (int rc = srec_lds_optin((const void*)gru_fused_bwd_mixed_kernel<2, 8>, (int)ldsm, omm[0])) return rc;
            hipLaunchKernelGGL((gru_fused_bwd_mixed_kernel<2, 8>), dim3(blocks), dim3(64 * 8), ldsm, (hipStream_t)stream, a);
        } else {
            if (int rc = srec_lds_optin((const void*)gru_fused_bwd_mixed_kernel<1, 4>, (int)ldsm, omm[1])) return rc;
            hipLaunchKernelGGL((gru_fused_bwd_mixed_kernel<1, 4>), dim3(blocks), dim3(64 * 4), ldsm, (hipStream_t)stream, a);
        }
        SREC_LAUNCH_CHECK();
        return 0;
    }
    if (batch && !no16) {
        const size_t lds16 = (size_t)(16 * 4 * 3 * D) * 2 + (size_t)16 * D * 2 + (size_t)16 * (D + 8) * 4;
        static std::atomic<unsigned long long> om16[2];
        if (D == 256) {
            if (int rc = srec_lds_optin((const void*)gru_fused_bwd16_kernel<2, 8>, (int)lds16, om16[0])) return rc;
            hipLaunchKernelGGL((gru_fused_bwd16_kernel<2, 8>), dim3(blocks), dim3(64 * 8), lds16, (hipStream_t)stream, a);
        } else {
            if (int rc = srec_lds_optin((const void*)gru_fused_bwd16_kernel<1, 4>, (int)lds16, om16[1])) return rc;
            hipLaunchKernelGGL((gru_fused_bwd16_kernel<1, 4>), dim3(blocks), dim3(64 * 4), lds16, (hipStream_t)stream, a);
        }
        SREC_LAUNCH_CHECK();
        return 0;
    }
    if (D == 256) { if (NRv == 16) SREC_GB(2, 16, 8, 4); else SREC_GB(2, 32, 8, 5); }
    else { if (NRv == 16) SREC_GB(1, 16, 4, 2); else SREC_GB(1, 32, 4, 3); }
#undef SREC_GB
    SREC_LAUNCH_CHECK();
    return 0;
}
