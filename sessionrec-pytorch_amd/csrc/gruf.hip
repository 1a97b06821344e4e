// k-gram GRU of MSGIFSR's SemanticExpander (msgifsr.py:25,32-45), forward of ALL time steps and orders in ONE launch (bf16 path,
// d = 128 or 256).
//
// The step-by-step formulation (grux.hip: GI GEMM, then per time step one h W_hh^T GEMM + one gate kernel) is 8 launches of
// 5-20 us, each a latency-bound ~10 us workgroup life over ~3.5 k rows.  The recurrence is independent per node, so here a
// workgroup OWNS 32 nodes of one order and runs their whole sequence:
//   * wave w owns the hidden columns [w d/4, (w + 1) d/4) of all three gates: 2 JB column blocks of 32 (JB = d / 128), i.e.
//     per step the accumulators r, z (x W_ih^T and h W_hh^T summed in ONE chain), gi_n, gh_n = 8 JB x 16 VGPRs; the hidden
//     state of its columns stays in registers (fp32, MFMA result layout: lane = column, register = node) across steps;
//   * the A operands (x_t and h_{t-1} of the 32 nodes, bf16 [32, d]) sit in LDS, 16-B pieces XOR-swizzled by row & 15 so the
//     ds_read_b128 lane groups are conflict-free; x_t is rounded on the way in (its bf16 copy for the weight-gradient GEMM
//     is written from here: no separate conversion pass), h_t is written back by the gate epilogue;
//   * the weights are streamed from L2 - both orders' matrices are 1.5 MB - in FRAGMENT-MAJOR order (srec_gru_wfrag: the 64
//     lanes x 16 B of one MFMA B operand contiguous, fragments ordered wave / k-step / gate / column block): one plain
//     coalesced 1-KiB load puts a fragment straight into the registers that feed the MFMAs (the copy IS the operand; the first
//     version went through LDS-DMA + ds_read_b128).  Every wave runs a private 4-stage REGISTER ring (3 k-steps in flight),
//     no barrier in the k-loop.  At 113 16-node workgroups per order x 1.15 / 1.9 MB this stream is 345 MB per launch =
//     ~9 TB/s over the 37 us: the launch runs at what the L2s deliver to the CUs (profiles/r03_notes.md);
//   * the gate math is the epilogue of the step's products; H, the bf16 copy of H, the saved gates and (last step) the
//     expander output 0.5 mean_t x + 0.5 h_last leave as 128-byte row segments.
// Arithmetic = the step kernels': bf16 operands, fp32 accumulation, fp32 gates (the r / z pre-activations are summed in a
// different order: one accumulation chain instead of two).
#include "common.h"
#include "../../include/srec_hg.h"
#include <type_traits>

extern "C" int srec_gru_fused_nodes(int np, const int* n, int d, int* nodes);      // grufb.hip
extern "C" int srec_gru_fused_wide(int np, const int* n, const int* k, int* mask);   // grufb.hip: problems (in launch order) that take 32-node tiles
extern "C" int srec_gru_fused_waves(int d, int* waves);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int GF_MAXP = SREC_GRU_MAXP;
constexpr int RT = 32;            // nodes per workgroup
constexpr int NS = 4, PF = NS - 1; // register ring: stages, k-steps of B fragments in flight

struct FusedArgs {
    srec_gru_fused_desc d;
    int start[GF_MAXP + 1];
    int wide;                      // mixed launch: bit p = problem p runs in 32-node workgroups (the others in 16-node ones)
};

#ifdef SREC_GRUF_TIMING   // development probe (tools/gruf_timing.py): phase clocks of wave 0 of one workgroup, workgroup lives
__device__ unsigned long long g_gruf_tim[16];
__device__ unsigned long long g_gruf_blk[1024][2];
#define GFT(i) do { __builtin_amdgcn_sched_barrier(0); if (tim_on) tim_t[i] += __builtin_readcyclecounter() - tim_c; \
    tim_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GFT(i)
#endif

// NR: nodes per workgroup, 32 or 16.  With 16 the lower half of every 32-row MFMA tile is idle (its A rows are whatever LDS
// holds, its results are never read) - the k-loops are bound by the weight stream, not by the MFMAs, so halving the nodes
// halves staging and gate epilogues at an unchanged k-loop and puts a workgroup on twice as many CUs (the bench batch has
// 111 32-node tiles for 256 CUs).
// NW: waves per workgroup, 4 or 8 (D = 256 only).  With 8 each wave owns d / 8 = 32 columns (one block): the same weight bytes
// per node as with 4, half the gate epilogue per wave and twice the loads in flight for the k-loop; 256 registers per wave.
template <int DD, int NR, int NW>
__device__ __forceinline__ void gru_fused_fwd_body(const FusedArgs& a) {
    constexpr int D = 128 * DD, JB = D / (32 * NW), NT = 64 * NW;   // JB: 32-column blocks per wave
    constexpr int KS = D / 16, NF = 3 * JB;
    constexpr int NRR = NR / 2;                  // accumulator registers per block that hold live nodes (rows (r&3) + 8 (r>>2) + 4 half)
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* xt = sm;                     // [RT][D] bf16, swizzled
    unsigned short* ht = sm + RT * D;
    constexpr int PS = 40;                       // patch row stride (floats): rows r, r + 4 land in opposite bank halves
    const srec_gru_fused_desc& q = a.d;
    int p = 0;
#pragma unroll
    for (int i = 1; i < GF_MAXP; ++i)
        if (i < q.np && (int)blockIdx.x >= a.start[i]) p = i;
    const int n = q.n[p], k = q.k[p];
    const int node0 = ((int)blockIdx.x - a.start[p]) * NR;
    if (node0 >= n) return;
    const int nl = dyn_count(q.dyn[p], n);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* X = q.X[p];
    unsigned short* X16 = (unsigned short*)q.X16[p];
    float* H = q.H[p];
    unsigned short* H16 = (unsigned short*)q.H16[p];
    float* out = q.out[p];

    if (node0 >= nl) {                           // capacity padding: zero rows, no arithmetic
        const int rows = min(NR, n - node0);
        for (int i = tid; i < rows * (D / 4); i += NT) {
            const int row = i / (D / 4), c = (i % (D / 4)) * 4;
            const size_t node = (size_t)(node0 + row);
            for (int t = 0; t < k; ++t) {
                *reinterpret_cast<float4*>(H + ((size_t)t * n + node) * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < k - 1) *reinterpret_cast<uint2*>(H16 + ((size_t)t * n + node) * D + c) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(X16 + (node * k + t) * D + c) = make_uint2(0u, 0u);
            }
            *reinterpret_cast<float4*>(out + node * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

#ifdef SREC_GRUF_TIMING
    const bool tim_on = (int)blockIdx.x == a.start[a.d.np - 1] + 1;     // a workgroup of the last (longest) order
    unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_c = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_gruf_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    float* patch = reinterpret_cast<float*>(sm + 2 * RT * D) + wave * NR * PS;          // this wave's [NR][PS] patch
    const int cbase = wave * 32 * JB;
    float b_r[JB], b_z[JB], b_in[JB], b_hn[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int col = cbase + 32 * j + l31;
        b_r[j] = q.bih[p][col] + q.bhh[p][col];
        b_z[j] = q.bih[p][D + col] + q.bhh[p][D + col];
        b_in[j] = q.bih[p][2 * D + col];
        b_hn[j] = q.bhh[p][2 * D + col];
    }
    f32x16 hprev[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) hprev[j][r] = 0.f;

    const unsigned short* wf_ih = (const unsigned short*)q.Wih_f[p] + (size_t)wave * KS * NF * 512;
    const unsigned short* wf_hh = (const unsigned short*)q.Whh_f[p] + (size_t)wave * KS * NF * 512;

    // x_t of the 32 nodes: fetched one time step ahead (the loads fly under the previous step's products), rounded to bf16
    // into LDS (swizzled) and into the bf16 copy the weight-gradient GEMM reads
    constexpr int XV = NR * D / 4 / NT;          // float4 per thread and tile
    float4 xv[XV];
    auto fetch_x = [&](int t) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = i * NT + tid;
            const int row = idx / (D / 4), c4 = idx % (D / 4);
            const int node = node0 + row;
            xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (node < n) xv[i] = *reinterpret_cast<const float4*>(X + ((unsigned)node * k + t) * (size_t)D + c4 * 4);
        }
    };
    fetch_x(0);
    const bool full = node0 + NR <= nl;          // every node of the tile is live: no per-row predicates
    for (int t = 0; t < k; ++t) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = i * NT + tid;
            const int row = idx / (D / 4), c4 = idx % (D / 4);
            const int node = node0 + row;
            uint2 o;
            o.x = srec_pack_bf16(xv[i].x, xv[i].y); o.y = srec_pack_bf16(xv[i].z, xv[i].w);
            const int pos = (c4 >> 1) ^ (row & 15);
            *reinterpret_cast<uint2*>(xt + row * D + pos * 8 + (c4 & 1) * 4) = o;
            if (node < n) *reinterpret_cast<uint2*>(X16 + ((unsigned)node * k + t) * (size_t)D + c4 * 4) = o;
        }
        if (t + 1 < k) fetch_x(t + 1);
        GFT(0);
        __syncthreads();                         // x_t (and h_{t-1}, written behind the previous step's barrier) published
        GFT(1);

        f32x16 ar[JB], az[JB], ain[JB], ahn[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ar[j][r] = az[j][r] = ain[j][r] = ahn[j][r] = 0.f;
        // k-steps of this time step: x W_ih^T (KS of them), then h W_hh^T (t > 0).  The B fragments of the next PF k-steps
        // sit in a register ring fed by plain 1-KiB coalesced loads (the fragment-major copy IS the operand); the loop body
        // covers NS k-steps so the ring slots are compile-time registers.  The tail re-loads the last stage (no branch).
        const int T = t > 0 ? 2 * KS : KS;
        bf16x8 Bq[NS][NF];
        auto load = [&](int i, int slot) {
            i = min(i, T - 1);
            const unsigned short* src = (i < KS ? wf_ih + (size_t)i * NF * 512 : wf_hh + (size_t)(i - KS) * NF * 512) + lane * 8;
#pragma unroll
            for (int f = 0; f < NF; ++f) Bq[slot][f] = *reinterpret_cast<const bf16x8*>(src + f * 512);
        };
        auto pass = [&](int i0, const unsigned short* At, f32x16 (&an)[JB]) {
#pragma unroll 1
            for (int ib = 0; ib < KS; ib += NS) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    load(i0 + ib + u + PF, (u + PF) % NS);
                    const int s = ib + u;
                    const bf16x8 A = *reinterpret_cast<const bf16x8*>(At + l31 * D + (((2 * s + half) ^ (l31 & 15)) * 8));
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        ar[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bq[u][j], ar[j], 0, 0, 0);
                        az[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bq[u][JB + j], az[j], 0, 0, 0);
                        an[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bq[u][2 * JB + j], an[j], 0, 0, 0);
                    }
                    // keep the issue order: next stage's loads, this k-step's A read, its MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x020, NF, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                }
            }
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) load(i, i);
        pass(0, xt, ain);
        if (t > 0) pass(KS, ht, ahn);

        GFT(2);
        // ---- gates, new hidden state (this wave's columns of the 32 nodes).  sigmoid / tanh through v_exp_f32 + v_rcp_f32
        // (tanh x = 2 sigmoid(2 x) - 1: absolute error ~1e-7 on values of O(1)); uniform bases + 32-bit element offsets
        const bool last = t == k - 1;
        const float ik = 0.5f / (float)k;
        float* Ht = H + (size_t)t * n * D;
        unsigned short* H16t = H16 + (size_t)t * n * D;
        _Float16* Gt = (_Float16*)q.gates[p] + (size_t)t * n * 4 * D;       // saved as fp16: half the bytes of the largest output
        auto sig = [](float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); };
        // (the lane id is re-derived behind an opaque asm every time step: hipcc otherwise hoists the ~250 per-element store
        // addresses out of the time loop - and spills them all to scratch)
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int l31v = lane_v & 31, halfv = lane_v >> 5;
        auto epilogue = [&](auto FULL) {
            constexpr bool F = decltype(FULL)::value;
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                // gate math in the MFMA result layout (lane = column, register = node); the five results go one after the other
                // through this wave's LDS patch so that they leave as 16-byte stores, 8 rows x 128 B per instruction (as 4-byte
                // stores in the result layout the 192 store instructions per wave and step were 2/3 of the kernel)
                float v5[5][NRR];
#pragma unroll
                for (int r = 0; r < NRR; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                    const float rr = sig(ar[j][r] + b_r[j]);
                    const float zz = sig(az[j][r] + b_z[j]);
                    const float hn = ahn[j][r] + b_hn[j];
                    const float nn = 2.f * sig(2.f * (ain[j][r] + b_in[j] + rr * hn)) - 1.f;
                    const float h = (F || node0 + row < nl) ? (1.f - zz) * nn + zz * hprev[j][r] : 0.f;
                    hprev[j][r] = h;
                    v5[0][r] = h; v5[1][r] = rr; v5[2][r] = zz; v5[3][r] = nn; v5[4][r] = hn;
                }
#pragma unroll
                for (int ten = 0; ten < 5; ++ten) {
#pragma unroll
                    for (int r = 0; r < NRR; ++r)
                        patch[((r & 3) + 8 * (r >> 2) + 4 * halfv) * PS + l31v] = v5[ten][r];
#pragma unroll
                    for (int i = 0; i < NR / 8; ++i) {
                        const int row = 8 * i + (lane_v >> 3), c4 = (lane_v & 7) * 4;
                        const int node = node0 + row;
                        if (F || node < n) {
                            const float4 hv = *reinterpret_cast<const float4*>(patch + row * PS + c4);
                            if (ten == 0) {
                                const unsigned off = (unsigned)node * D + cbase + 32 * j + c4;
                                *reinterpret_cast<float4*>(Ht + off) = hv;
                                if (!last) *reinterpret_cast<uint2*>(H16t + off) = make_uint2(srec_pack_bf16(hv.x, hv.y), srec_pack_bf16(hv.z, hv.w));
                                if (last) {
                                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                                    if (F || node < nl) {
                                        const unsigned xo = (unsigned)node * k * D + cbase + 32 * j + c4;
                                        for (int tt = 0; tt < k; ++tt) {
                                            const float4 v = *reinterpret_cast<const float4*>(X + xo + tt * D);
                                            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                                        }
                                        o = make_float4(ik * o.x + 0.5f * hv.x, ik * o.y + 0.5f * hv.y, ik * o.z + 0.5f * hv.z, ik * o.w + 0.5f * hv.w);
                                    }
                                    *reinterpret_cast<float4*>(out + off) = o;
                                }
                            } else {
                                const unsigned goff = (unsigned)node * (4 * D) + cbase + 32 * j + c4;
                                typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                                const h4_t hq = {(_Float16)hv.x, (_Float16)hv.y, (_Float16)hv.z, (_Float16)hv.w};
                                *reinterpret_cast<h4_t*>(Gt + goff + (ten - 1) * D) = hq;
                            }
                        }
                    }
                }
            }
        };
        if (full) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        GFT(3);
        if (!last) {
            __syncthreads();                     // every wave is done reading x_t / h_{t-1}
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const int col = cbase + 32 * j + l31v;
#pragma unroll
                for (int r = 0; r < NRR; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                    ht[row * D + (((col >> 3) ^ (row & 15)) * 8) + (col & 7)] = srec_f2bf(hprev[j][r]);
                }
            }
        }
        GFT(4);
    }
#ifdef SREC_GRUF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GFT(5);
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_gruf_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    if (tim_on && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_gruf_tim[i] = tim_t[i];
#endif
}

// ---- 16-node workgroups with the INPUT projections of all time steps batched -----------------------------------------------
// x_t W_ih^T has no recurrence: the rows of all k <= 4 time steps of the 16 nodes (tile row = 16 t + node: 32 - 64 rows) go
// through ONE pass over W_ih, whose fragments then feed full 32-row MFMA tiles, instead of one half-empty pass per step.  A
// workgroup streams W_ih + (k - 1) W_hh instead of k W_ih + (k - 1) W_hh - 2 / 3 of the bytes at order 2, 3 / 5 at order 3, and
// this kernel runs at what the L2s deliver (see the head of the file).  The projections gi of every step wait in registers
// (MFMA result layout: registers 8 (t & 1) .. + 7 of tile t >> 1 hold the 16 nodes of step t); the recurrent half and the gate
// epilogue are those of gru_fused_fwd_kernel (r / z pre-activations = gi + gh, two chains added instead of one chain).
template <int DD, int NW>
__device__ __forceinline__ void gru_fused_fwd16_body(const FusedArgs& a) {
    constexpr int NR = 16;
    constexpr int D = 128 * DD, JB = D / (32 * NW), NT = 64 * NW;
    constexpr int KS = D / 16, NF = 3 * JB, NRR = NR / 2;
    constexpr int KMAX = 4;                      // time steps (k-gram order) this kernel holds projections for: 2 tiles of 32 rows
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* xt = sm;                     // [64][D] bf16, swizzled: rows 16 t + node
    unsigned short* ht = sm + 64 * D;            // [32][D]
    constexpr int PS = 40;
    const srec_gru_fused_desc& q = a.d;
    int p = 0;
#pragma unroll
    for (int i = 1; i < GF_MAXP; ++i)
        if (i < q.np && (int)blockIdx.x >= a.start[i]) p = i;
    const int n = q.n[p], k = q.k[p];
    const int node0 = ((int)blockIdx.x - a.start[p]) * NR;
    if (node0 >= n) return;
    const int nl = dyn_count(q.dyn[p], n);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* X = q.X[p];
    unsigned short* X16 = (unsigned short*)q.X16[p];
    float* H = q.H[p];
    unsigned short* H16 = (unsigned short*)q.H16[p];
    float* out = q.out[p];

    if (node0 >= nl) {                           // capacity padding: zero rows, no arithmetic
        const int rows = min(NR, n - node0);
        for (int i = tid; i < rows * (D / 4); i += NT) {
            const int row = i / (D / 4), c = (i % (D / 4)) * 4;
            const size_t node = (size_t)(node0 + row);
            for (int t = 0; t < k; ++t) {
                *reinterpret_cast<float4*>(H + ((size_t)t * n + node) * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < k - 1) *reinterpret_cast<uint2*>(H16 + ((size_t)t * n + node) * D + c) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(X16 + (node * k + t) * D + c) = make_uint2(0u, 0u);
            }
            *reinterpret_cast<float4*>(out + node * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    float* patch = reinterpret_cast<float*>(sm + 96 * D) + wave * NR * PS;              // this wave's [NR][PS] patch
    const int cbase = wave * 32 * JB;
    float b_r[JB], b_z[JB], b_in[JB], b_hn[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int col = cbase + 32 * j + l31;
        b_r[j] = q.bih[p][col] + q.bhh[p][col];
        b_z[j] = q.bih[p][D + col] + q.bhh[p][D + col];
        b_in[j] = q.bih[p][2 * D + col];
        b_hn[j] = q.bhh[p][2 * D + col];
    }
    const unsigned short* wf_ih = (const unsigned short*)q.Wih_f[p] + (size_t)wave * KS * NF * 512 + lane * 8;
    const unsigned short* wf_hh = (const unsigned short*)q.Whh_f[p] + (size_t)wave * KS * NF * 512 + lane * 8;
    bf16x8 Bq[NS][NF];
    auto load = [&](const unsigned short* wf, int i, int slot) {
        i = min(i, KS - 1);
#pragma unroll
        for (int f = 0; f < NF; ++f) Bq[slot][f] = *reinterpret_cast<const bf16x8*>(wf + ((size_t)i * NF + f) * 512);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load(wf_ih, i, i);            // (the first fragments fly under the staging of x)

    // ---- x of ALL time steps: bf16 into the two LDS tiles (row 16 t + node) and into the weight-gradient GEMM's operand copy
    for (int i = tid; i < k * NR * (D / 4); i += NT) {
        const int c4 = i % (D / 4), rn = i / (D / 4), node_l = rn % NR, t = rn / NR;
        const int node = node0 + node_l, row = NR * t + node_l;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (node < n) v = *reinterpret_cast<const float4*>(X + ((unsigned)node * k + t) * (size_t)D + c4 * 4);
        uint2 o;
        o.x = srec_pack_bf16(v.x, v.y); o.y = srec_pack_bf16(v.z, v.w);
        const int pos = (c4 >> 1) ^ (row & 15);
        *reinterpret_cast<uint2*>(xt + row * D + pos * 8 + (c4 & 1) * 4) = o;
        if (node < n) *reinterpret_cast<uint2*>(X16 + ((unsigned)node * k + t) * (size_t)D + c4 * 4) = o;
    }
    __syncthreads();

    // ---- gi of all steps: one pass over W_ih, every fragment feeds both row tiles (the second only when k > 2)
    f32x16 gi[3][2][JB];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) gi[g][tl][j][r] = 0.f;
    const bool two = k > 2;
#pragma unroll 1
    for (int ib = 0; ib < KS; ib += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            load(wf_ih, ib + u + PF, (u + PF) % NS);
            const int s = ib + u;
            const bf16x8 A0 = *reinterpret_cast<const bf16x8*>(xt + l31 * D + (((2 * s + half) ^ (l31 & 15)) * 8));
            const bf16x8 A1 = *reinterpret_cast<const bf16x8*>(xt + (32 + l31) * D + (((2 * s + half) ^ (l31 & 15)) * 8));
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int j = 0; j < JB; ++j) gi[g][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bq[u][g * JB + j], gi[g][0][j], 0, 0, 0);
            if (two) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int j = 0; j < JB; ++j) gi[g][1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bq[u][g * JB + j], gi[g][1][j], 0, 0, 0);
            }
        }
    }
    // the recurrent weights' first fragments: requested now, they land under the first gate epilogue
#pragma unroll
    for (int i = 0; i < PF; ++i) load(wf_hh, i, i);

    float hprev[JB][NRR];
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < NRR; ++r) hprev[j][r] = 0.f;
    const bool full = node0 + NR <= nl;
    auto sig = [](float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); };
    const float ik = 0.5f / (float)k;

#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        if (t >= k) break;
        f32x16 ar[JB], az[JB], ahn[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ar[j][r] = az[j][r] = ahn[j][r] = 0.f;
        if (t > 0) {
            // h_{t-1} W_hh^T (nodes in rows 0 .. 15 of the tile; the upper half is idle)
#pragma unroll 1
            for (int ib = 0; ib < KS; ib += NS) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    load(wf_hh, ib + u + PF, (u + PF) % NS);
                    const int s = ib + u;
                    const bf16x8 A = *reinterpret_cast<const bf16x8*>(ht + l31 * D + (((2 * s + half) ^ (l31 & 15)) * 8));
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        ar[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bq[u][j], ar[j], 0, 0, 0);
                        az[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bq[u][JB + j], az[j], 0, 0, 0);
                        ahn[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bq[u][2 * JB + j], ahn[j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x020, NF, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                }
            }
            if (t + 1 < k) {
#pragma unroll
                for (int i = 0; i < PF; ++i) load(wf_hh, i, i);            // next step's first fragments, under this epilogue
            }
        }
        const bool last = t == k - 1;
        float* Ht = H + (size_t)t * n * D;
        unsigned short* H16t = H16 + (size_t)t * n * D;
        _Float16* Gt = (_Float16*)q.gates[p] + (size_t)t * n * 4 * D;
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int l31v = lane_v & 31, halfv = lane_v >> 5;
        constexpr int TL = 0;
        (void)TL;
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            float v5[5][NRR];
#pragma unroll
            for (int r = 0; r < NRR; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                const float gr = gi[0][t >> 1][j][8 * (t & 1) + r], gz = gi[1][t >> 1][j][8 * (t & 1) + r], gn = gi[2][t >> 1][j][8 * (t & 1) + r];
                const float rr = sig(gr + ar[j][r] + b_r[j]);
                const float zz = sig(gz + az[j][r] + b_z[j]);
                const float hn = ahn[j][r] + b_hn[j];
                const float nn = 2.f * sig(2.f * (gn + b_in[j] + rr * hn)) - 1.f;
                const float h = (full || node0 + row < nl) ? (1.f - zz) * nn + zz * hprev[j][r] : 0.f;
                hprev[j][r] = h;
                v5[0][r] = h; v5[1][r] = rr; v5[2][r] = zz; v5[3][r] = nn; v5[4][r] = hn;
            }
#pragma unroll
            for (int ten = 0; ten < 5; ++ten) {
#pragma unroll
                for (int r = 0; r < NRR; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * halfv) * PS + l31v] = v5[ten][r];
#pragma unroll
                for (int i = 0; i < NR / 8; ++i) {
                    const int row = 8 * i + (lane_v >> 3), c4 = (lane_v & 7) * 4;
                    const int node = node0 + row;
                    if (full || node < n) {
                        const float4 hv = *reinterpret_cast<const float4*>(patch + row * PS + c4);
                        if (ten == 0) {
                            const unsigned off = (unsigned)node * D + cbase + 32 * j + c4;
                            *reinterpret_cast<float4*>(Ht + off) = hv;
                            if (!last) *reinterpret_cast<uint2*>(H16t + off) = make_uint2(srec_pack_bf16(hv.x, hv.y), srec_pack_bf16(hv.z, hv.w));
                            if (last) {
                                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (full || node < nl) {
                                    const unsigned xo = (unsigned)node * k * D + cbase + 32 * j + c4;
                                    for (int tt = 0; tt < k; ++tt) {
                                        const float4 v = *reinterpret_cast<const float4*>(X + xo + tt * D);
                                        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                                    }
                                    o = make_float4(ik * o.x + 0.5f * hv.x, ik * o.y + 0.5f * hv.y, ik * o.z + 0.5f * hv.z, ik * o.w + 0.5f * hv.w);
                                }
                                *reinterpret_cast<float4*>(out + off) = o;
                            }
                        } else {
                            const unsigned goff = (unsigned)node * (4 * D) + cbase + 32 * j + c4;
                            typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                            const h4_t hq = {(_Float16)hv.x, (_Float16)hv.y, (_Float16)hv.z, (_Float16)hv.w};
                            *reinterpret_cast<h4_t*>(Gt + goff + (ten - 1) * D) = hq;
                        }
                    }
                }
            }
        }
        if (!last) {
            __syncthreads();                     // every wave is done reading h_{t-1}
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const int col = cbase + 32 * j + l31v;
#pragma unroll
                for (int r = 0; r < NRR; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * halfv;
                    ht[row * D + (((col >> 3) ^ (row & 15)) * 8) + (col & 7)] = srec_f2bf(hprev[j][r]);
                }
            }
            __syncthreads();                     // h_t published
        }
    }
}

template <int DD, int NR, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fused_fwd_kernel(FusedArgs a) { gru_fused_fwd_body<DD, NR, NW>(a); }

template <int DD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fused_fwd16_kernel(FusedArgs a) { gru_fused_fwd16_body<DD, NW>(a); }

// Mixed launch: the long problems in 16-node workgroups (gru_fused_fwd16_body: the projections of all steps batched), the SHORT
// ones (a.wide) in 32-node workgroups (gru_fused_fwd_body) - half as many of them.  A workgroup takes a CU for itself (registers)
// and lives as long as its weight stream whatever its node count: with every problem in 16-node tiles the bench batches have 250 -
// 280 live workgroups for 256 CUs, and the ~45 % of them above 256 ran a second round (forward 26 -> 37 us, backward 34 -> 46 us:
// profiles/r05_notes.md 5, r06_kernel_spread.txt); with the order-2 problem in 32-node tiles every batch is one round.
template <int DD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fused_fwd_mixed_kernel(FusedArgs a) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < GF_MAXP; ++i)
        if (i < a.d.np && (int)blockIdx.x >= a.start[i]) p = i;
    if ((a.wide >> p) & 1) gru_fused_fwd_body<DD, 32, NW>(a);
    else gru_fused_fwd16_body<DD, NW>(a);
}

struct WfArgs {
    int d, jb;
    const float* W[2 * GF_MAXP];
    unsigned short* dst[2 * GF_MAXP];
};

// fragment-major bf16 copy of a GRU weight W [3 d, d]: fragment ((w KS + s) NF + g JB + j) = the MFMA B operand of wave w,
// k-step s, gate g, column block j: lane l <- W[g d + w d/4 + 32 j + (l & 31)][16 s + 8 (l >> 5) .. + 7]
__global__ __launch_bounds__(256) void gru_wfrag_kernel(WfArgs a) {
    const int d = a.d, JB = a.jb, KS = d / 16, NF = 3 * JB;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * d * d / 8) return;
    const int lane = idx & 63, frag = idx >> 6;
    const int f = frag % NF, ws = frag / NF, s = ws % KS, w = ws / KS, g = f / JB, j = f % JB;
    const int nrow = g * d + w * 32 * JB + 32 * j + (lane & 31), kk = 16 * s + 8 * (lane >> 5);
    const float* src = a.W[blockIdx.y] + (size_t)nrow * d + kk;
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    uint4 o;
    o.x = srec_pack_bf16(v0.x, v0.y); o.y = srec_pack_bf16(v0.z, v0.w);
    o.z = srec_pack_bf16(v1.x, v1.y); o.w = srec_pack_bf16(v1.z, v1.w);
    *reinterpret_cast<uint4*>(a.dst[blockIdx.y] + (size_t)idx * 8) = o;
}

}  // namespace

#ifdef SREC_GRUF_TIMING
extern "C" int srec_gruf_timing(unsigned long long* tim16, unsigned long long* blk) {
    if (hipMemcpyFromSymbol(tim16, HIP_SYMBOL(g_gruf_tim), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(blk, HIP_SYMBOL(g_gruf_blk), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : 1;
}
#endif

// n <= 8 GRU weight matrices W_i [3 d, d] fp32 (HOST array of device pointers) -> fragment-major bf16 copies dst_i [3 d d]
// (the B operands of srec_gru_fused_fwd), one launch; d = 128 or 256
extern "C" int srec_gru_wfrag(int n, const void* W, const void* dst, int d, void* stream) {
    if (n <= 0) return 0;
    if (n > 2 * GF_MAXP || W == nullptr || dst == nullptr || (d != 128 && d != 256)) return SREC_BAD_ARG;
    WfArgs a{};
    int nw = 4;
    if (int rc = srec_gru_fused_waves(d, &nw)) return rc;
    a.d = d; a.jb = d / (32 * nw);
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i]; a.dst[i] = ((unsigned short* const*)dst)[i];
        if (a.W[i] == nullptr || a.dst[i] == nullptr) return SREC_BAD_ARG;
    }
    hipLaunchKernelGGL(gru_wfrag_kernel, dim3((3 * d * d / 8 + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}


// desc: HOST srec_gru_fused_desc (srec_hg.h)
// longest problems first (workgroups start in index order; see srec_gru_fused_bwd)
static void longest_first(srec_gru_fused_desc& d) {
    int ord[GF_MAXP];
    for (int i = 0; i < d.np; ++i) ord[i] = i;
    for (int i = 1; i < d.np; ++i)
        for (int j = i; j > 0 && d.k[ord[j]] > d.k[ord[j - 1]]; --j) { const int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
    const srec_gru_fused_desc s = d;
    for (int q = 0; q < d.np; ++q) {
        const int p = ord[q];
        d.n[q] = s.n[p]; d.k[q] = s.k[p]; d.dyn[q] = s.dyn[p]; d.X[q] = s.X[p]; d.X16[q] = s.X16[p]; d.Wih_f[q] = s.Wih_f[p];
        d.Whh_f[q] = s.Whh_f[p]; d.bih[q] = s.bih[p]; d.bhh[q] = s.bhh[p]; d.H[q] = s.H[p]; d.H16[q] = s.H16[p];
        d.gates[q] = s.gates[p]; d.out[q] = s.out[p];
    }
}

extern "C" int srec_gru_fused_fwd(const void* desc, void* stream) {
    const srec_gru_fused_desc* q0 = (const srec_gru_fused_desc*)desc;
    if (q0 == nullptr || q0->np <= 0 || q0->np > GF_MAXP || (q0->d != 128 && q0->d != 256)) return SREC_BAD_ARG;
    FusedArgs a{};
    a.d = *q0;
    longest_first(a.d);
    const srec_gru_fused_desc* q = &a.d;
    int blocks = 0;
    for (int p = 0; p < q->np; ++p) {
        if (q->n[p] < 0 || q->k[p] < 1 || q->X[p] == nullptr || q->X16[p] == nullptr || q->Wih_f[p] == nullptr ||
            q->Whh_f[p] == nullptr || q->bih[p] == nullptr || q->bhh[p] == nullptr || q->H[p] == nullptr ||
            q->H16[p] == nullptr || q->gates[p] == nullptr || q->out[p] == nullptr)
            return SREC_BAD_ARG;
        a.start[p] = blocks;
        blocks += (q->n[p] + RT - 1) / RT;
    }
    // 16-node workgroups while 32-node ones would leave half of the chip idle
    int NRv = 32;
    if (int rc = srec_gru_fused_nodes(q->np, q->n, q->d, &NRv)) return rc;
    bool batch = NRv == 16;                      // gru_fused_fwd16_kernel holds the projections of <= 4 time steps
    for (int p = 0; p < q->np; ++p) batch = batch && q->k[p] <= 4;
    if (NRv == 16) {
        // one round of the chip: while the 16-node tiles of all problems exceed the CUs, the shortest problems go to 32-node tiles
        if (batch)
            if (int rc = srec_gru_fused_wide(q->np, q->n, q->k, &a.wide)) return rc;
        blocks = 0;
        for (int p = 0; p < q->np; ++p) { a.start[p] = blocks; blocks += (q->n[p] + (((a.wide >> p) & 1) ? 31 : 15)) / (((a.wide >> p) & 1) ? 32 : 16); }
    }
    a.start[q->np] = blocks;
    for (int p = q->np + 1; p <= GF_MAXP; ++p) a.start[p] = blocks;
    if (blocks == 0) return 0;
    const int D = q->d;
    int NWv = 4;
    if (int rc = srec_gru_fused_waves(D, &NWv)) return rc;
    const size_t lds = (size_t)(2 * RT * D) * 2 + (size_t)NWv * NRv * 40 * 4;
    static std::atomic<unsigned long long> om[6];
#define SREC_GF(DDV, NRV, NWV, slot)                                                                                   \
    do {                                                                                                               \
        if (int rc = srec_lds_optin((const void*)gru_fused_fwd_kernel<DDV, NRV, NWV>, (int)lds, om[slot])) return rc;  \
        hipLaunchKernelGGL((gru_fused_fwd_kernel<DDV, NRV, NWV>), dim3(blocks), dim3(64 * NWV), lds, (hipStream_t)stream, a); \
    } while (0)
    if (batch && a.wide) {
        const size_t lds16 = (size_t)(96 * D) * 2 + (size_t)NWv * 16 * 40 * 4, lds32 = (size_t)(2 * RT * D) * 2 + (size_t)NWv * 32 * 40 * 4;
        const size_t ldsm = lds16 > lds32 ? lds16 : lds32;
        static std::atomic<unsigned long long> omm[2];
        if (D == 256) {
            if (int rc = srec_lds_optin((const void*)gru_fused_fwd_mixed_kernel<2, 8>, (int)ldsm, omm[0])) return rc;
            hipLaunchKernelGGL((gru_fused_fwd_mixed_kernel<2, 8>), dim3(blocks), dim3(64 * 8), ldsm, (hipStream_t)stream, a);
        } else {
            if (int rc = srec_lds_optin((const void*)gru_fused_fwd_mixed_kernel<1, 4>, (int)ldsm, omm[1])) return rc;
            hipLaunchKernelGGL((gru_fused_fwd_mixed_kernel<1, 4>), dim3(blocks), dim3(64 * 4), ldsm, (hipStream_t)stream, a);
        }
    } else if (batch) {
        const size_t lds16 = (size_t)(96 * D) * 2 + (size_t)NWv * 16 * 40 * 4;
        if (D == 256) {
            if (int rc = srec_lds_optin((const void*)gru_fused_fwd16_kernel<2, 8>, (int)lds16, om[0])) return rc;
            hipLaunchKernelGGL((gru_fused_fwd16_kernel<2, 8>), dim3(blocks), dim3(64 * 8), lds16, (hipStream_t)stream, a);
        } else {
            if (int rc = srec_lds_optin((const void*)gru_fused_fwd16_kernel<1, 4>, (int)lds16, om[1])) return rc;
            hipLaunchKernelGGL((gru_fused_fwd16_kernel<1, 4>), dim3(blocks), dim3(64 * 4), lds16, (hipStream_t)stream, a);
        }
    } else if (D == 256) { if (NRv == 16) SREC_GF(2, 16, 8, 4); else SREC_GF(2, 32, 8, 5); }
    else { if (NRv == 16) SREC_GF(1, 16, 4, 2); else SREC_GF(1, 32, 4, 3); }
#undef SREC_GF
    SREC_LAUNCH_CHECK();
    return 0;
}
