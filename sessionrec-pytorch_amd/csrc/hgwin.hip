// MSGIFSR's MSHGNN layer (msgifsr.py:47-91; GATConv gatconv.py:267-311) per ROW WINDOW, without projections in memory.
//
// The batched formulation of hgat.hip projects every node row to [H * D] per GAT module (P = x W^T, 8 modules: ~90 MB of bf16
// at the C3 shape), and every pass of the layer - aggregation, score gradients, projection gradients, both backward GEMMs -
// streams a tensor of that size through HBM: ~0.9 GB per layer call for ~50 MB of inputs, outputs and gradients.  Here the
// aggregation runs FIRST, on the 8 x narrower input rows, and the projection is applied to the aggregate:
//     rst[v, h, :] = sum_{e = (u -> v)} a_eh (W_h x_u) = W_h (sum_e a_eh x_u) = W_h Y[v, h, :]
// (a_eh: edge soft-max of head h, from the logits of the folded attention vectors - hg_fold / hg_dots of hgat.hip).  A
// workgroup owns a window of WR = 64 DESTINATION rows of ONE node type and one half of the output columns (D = 256: two
// workgroups per window).  A type's rows receive messages through at most four GAT modules (conv1 / conv2 x {intra_k, inter}):
// the module "slots" of the window.  Per slot and per group of HG = 4 heads:
//   phase A (vector ALU): Y[v, h, :] = sum over the in-edges of the slot's relation instances of a_eh x16[u, :] - the source
//     rows are gathered as 512-byte bf16 rows (L2 resident: all rows of the batch are 5 MB), four edges in flight per
//     wavefront, from an edge list the workgroup builds once in LDS (sources + soft-max weights of every in-edge of its rows,
//     ordered by row) - and is written as a bf16 MFMA operand tile into LDS ([4 heads][64 rows][D], 128 KB, 16-byte pieces
//     XOR-swizzled by the row: conflict-free ds_read_b128);
//   phase B (matrix pipe): acc[h] (+)= W_{m,h}[this wave's 32 output columns, :] Y[., h, :]^T with v_mfma_f32_32x32x16_bf16,
//     computed TRANSPOSED (A = weight fragment, B = 32 rows of Y) so that a row lands in the lane with 16 output columns in
//     its registers; the weights stream from L2 as FRAGMENT-MAJOR bf16 copies (srec_hg_wfrag: the 64 lanes x 16 B of one MFMA
//     operand contiguous; plain 1-KiB loads straight into the operand registers through a register ring, the idiom of
//     gruf.hip / headf.hip); every fragment feeds the window's two 32-row tiles.
// The eight heads' accumulators (8 x 2 x 16 registers) stay in the register file across the slots - the relation sum of
// msgifsr.py:84 and HeteroGraphConv's 'sum' are the accumulation -, and the epilogue adds bias and identity residuals, takes
// the maximum over the heads (first maximum wins, as a scan over h) and adds the session mean: out [rows, D] and the arg-max
// bytes are the only things written besides the soft-max values A the backward reads.
#include "common.h"
#include "../../include/srec.h"
#include "../../include/srec_hg.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MAXT = SREC_HG_MAXT, MAXM = SREC_HG_MAXM, MAXI = SREC_HG_MAXI;
constexpr int WR = SREC_HGWIN_ROWS;      // destination rows per window (two 32-row MFMA tiles)
constexpr int NW = 4;                    // waves per workgroup; wave w owns 32 output columns
constexpr int HG = 4;                    // heads per pass of the LDS operand tile
constexpr int ECAP = 1024;               // in-edges of one window (all slots) held by the LDS edge list
constexpr int MAXS = 4;                  // module slots per node type
constexpr int MAXQ = 4;                  // relation instances per slot
constexpr int NS = 12, PF = NS - 1;      // register ring of weight fragments: stages, fragments in flight
constexpr int RB = 16;                   // source rows of a wave's edges of one slot kept in registers (phase A)

#ifndef SREC_HGWIN_KO
#define SREC_HGWIN_KO 0      // knock-out builds of tools/hgwin_timing.py: 1 = no weight loads in the matrix loop, 2 = no LDS reads
#endif
#ifdef SREC_HGWIN_TIMING   // development probe (tools/hgwin_timing.py): phase clocks of wave 0 of every workgroup
__device__ unsigned long long g_hgwin_tim[2048][8];
__device__ unsigned long long g_hgwin_blk[2048][2];
#define WT(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    tim_t[i] += __builtin_readcyclecounter() - tim_c; tim_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WT(i)
#endif

struct WinArgs {
    // node types
    int nt;
    int row0[MAXT + 1], ncap[MAXT], wstart[MAXT + 1], ninst[MAXT];
    const int* dyn_n[MAXT];
    const float* bsum[MAXT];             // summed bias rows of the instances into the type [H D] (hg_fold)
    const float* smean[MAXT];            // per (type, session) mean of the input rows [B, D] (hg_dots)
    int nslot[MAXT], slot_mod[MAXT][MAXS], slot_nq[MAXT][MAXS], slot_inst[MAXT][MAXS][MAXQ];
    // modules
    const unsigned short* x16[MAXM];     // bf16 input rows of the module's conv [NT, D]
    const unsigned short* Wf[MAXM];      // fragment-major bf16 fc weight (srec_hg_wfrag)
    // relation instances
    const float* eLs[MAXI]; const float* eRd[MAXI]; const float* Mk[MAXI];
    const int* in_ptr[MAXI]; const int* in_idx[MAXI]; const int* esrc[MAXI];
    float* A[MAXI];
    int src_row0[MAXI];
    const float* x; int ld_x;
    const float* xres;
    const int* sess;
    float* out; int ld_out;
    unsigned char* arg;
    float slope;
    int force_slow;
};

__device__ __forceinline__ void ld8(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void bf8_to_f(const uint4 v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS of the forward kernel (bytes): operand tile | edge sources | edge weights (fp16 x 8) | row pointers | scan scratch
template <int D> struct FwdLds {
    static constexpr int Y = 0;
    static constexpr int YB = HG * WR * D * 2 > ECAP * 80 ? HG * WR * D * 2 : ECAP * 80;    // (the list build's scratch: 80 B per edge)
    static constexpr int ESRC = YB;
    static constexpr int EAL = ESRC + ECAP * 4;
    static constexpr int RP = EAL + ECAP * 16;
    static constexpr int MISC = RP + MAXS * (WR + 1) * 4;
    static constexpr int BIAS = MISC + 64;                       // [8][NW * 32] summed bias of this workgroup's columns
    static constexpr int TOTAL = BIAS + 8 * NW * 32 * 4;
};

template <int D>
__global__ __launch_bounds__(64 * NW, 1) void hg_win_fwd_kernel(WinArgs a) {
    constexpr int KS = D / 16, NCT = D / 32, NCH = NCT / NW;
    typedef FwdLds<D> L;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* Y = reinterpret_cast<unsigned short*>(smem + L::Y);       // [HG][WR][D] bf16, pieces swizzled by row & 15
    int* e_src = reinterpret_cast<int*>(smem + L::ESRC);                      // [ECAP] stacked row of the edge's source
    _Float16* e_al = reinterpret_cast<_Float16*>(smem + L::EAL);              // [ECAP][8] soft-max weight x dropout multiplier
    int* rp = reinterpret_cast<int*>(smem + L::RP);                           // [MAXS][WR + 1] first edge of (slot, row)
    int* misc = reinterpret_cast<int*>(smem + L::MISC);
    float* bias_s = reinterpret_cast<float*>(smem + L::BIAS);
    // scratch of the list build, inside the (not yet used) operand tile
    int* t_inst = reinterpret_cast<int*>(smem);                               // [ECAP] instance of the edge
    int* t_pos = t_inst + ECAP;                                               // [ECAP] position in in_idx
    int* t_e = t_pos + ECAP;                                                  // [ECAP] edge id
    int* t_v = t_e + ECAP;                                                    // [ECAP] destination (type-local)
    float* t_s = reinterpret_cast<float*>(t_v + ECAP);                        // [ECAP][8] activated logits
    float* t_m = t_s + ECAP * 8;                                              // [ECAP][8] dropout multipliers

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = uni(tid >> 6);
    // work unit = (node type, column half, window), numbered in that order; workgroups are dealt to the 8 XCDs round-robin by
    // index and every XCD has its own 4-MB L2, so XCD x takes the x-th consecutive RUN of units: the workgroups behind one L2
    // then stream the same one or two (type, half) weight sets (2 MB each) and re-read them from that L2 however far they drift
    // apart (with unit = blockIdx an XCD saw all six streams, 12 MB: every re-read went to the fabric - 190 instead of ~90
    // cycles per k-step)
    const int upx = (int)gridDim.x / 8;
    const int unit = (a.force_slow & 2) ? (int)blockIdx.x : ((int)blockIdx.x % 8) * upx + (int)blockIdx.x / 8;
    if (unit >= a.wstart[a.nt] * NCH) return;
    int t = 0;
#pragma unroll
    for (int i = 1; i < MAXT; ++i)
        if (i < a.nt && unit >= a.wstart[i] * NCH) t = i;
    const int nwt = a.wstart[t + 1] - a.wstart[t];
    const int ch = (unit - a.wstart[t] * NCH) / nwt, win = a.wstart[t] + (unit - a.wstart[t] * NCH) % nwt;
    const int w0 = (win - a.wstart[t]) * WR;
    const int ncap = a.ncap[t];
    const int nlive = dyn_count(a.dyn_n[t], ncap);
    const int grow0 = a.row0[t] + w0;                                         // stacked row of the window's first row
    const int ct = NW * ch + wave;                                            // this wave's 32-column tile
#ifdef SREC_HGWIN_TIMING
    unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_c = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x < 2048) { g_hgwin_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime(); g_hgwin_blk[blockIdx.x][1] = 0; }
#endif
    if (w0 >= nlive) {                                                        // capacity padding: zero rows
        for (int i = tid; i < WR * (NW * 32 / 4); i += 64 * NW) {
            const int r = i / (NW * 32 / 4), c = NW * 32 * ch + 4 * (i % (NW * 32 / 4));
            if (w0 + r < ncap) {
                *reinterpret_cast<float4*>(a.out + (size_t)(grow0 + r) * a.ld_out + c) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<unsigned*>(a.arg + (size_t)(grow0 + r) * D + c) = 0u;
            }
        }
        return;
    }
    const int nslot = a.nslot[t];
    // early requests of what the epilogue needs: the summed bias of this workgroup's columns (LDS), the sessions of the lane's rows
    for (int i = tid; i < 8 * NW * 32; i += 64 * NW) {
        const int h = i / (NW * 32), c = i % (NW * 32);
        bias_s[i] = a.ninst[t] > 0 ? a.bsum[t][h * D + NW * 32 * ch + c] : 0.f;
    }
    int sess2[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) sess2[tt] = (w0 + 32 * tt + l31 < nlive) ? a.sess[grow0 + 32 * tt + l31] : 0;

    // ---------------------------------------------------------------- edge list of the window (all slots), ordered by (slot, row)
    // item = (slot, row) = thread: in-degrees of the slot's instances, exclusive scan -> first edge of every (slot, row)
    int beg[MAXQ], deg[MAXQ], tot = 0;
    {
        const int s = tid >> 6, r = tid & 63, v = w0 + r;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) { beg[q] = 0; deg[q] = 0; }
        if (s < nslot && v < nlive) {
#pragma unroll
            for (int q = 0; q < MAXQ; ++q)
                if (q < a.slot_nq[t][s]) {
                    const int i = a.slot_inst[t][s][q];
                    const int b0 = a.in_ptr[i][v], b1 = a.in_ptr[i][v + 1];
                    beg[q] = b0; deg[q] = b1 - b0; tot += b1 - b0;
                }
        }
        int inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o, 64);
            if (lane >= o) inc += n;
        }
        if (lane == 63) misc[wave] = inc;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) base += (w < wave) ? misc[w] : 0;
        const int off = base + inc - tot;
        rp[s * (WR + 1) + r] = off;
        if (r == WR - 1) rp[s * (WR + 1) + WR] = off + tot;
        if (tid == 64 * NW - 1) misc[4] = off + tot;
        // per-edge descriptors
        if (off + tot <= ECAP) {
            int k = off;
#pragma unroll
            for (int q = 0; q < MAXQ; ++q)
                for (int j = 0; j < deg[q]; ++j, ++k) {
                    t_inst[k] = a.slot_inst[t][s][q];
                    t_pos[k] = beg[q] + j;
                    t_v[k] = v;
                }
        }
    }
    __syncthreads();
    WT(0);
    const int etot = misc[4];
    const bool slow = etot > ECAP || (a.force_slow & 1) != 0;
    if (!slow) {
        // edge = thread: edge id, source, activated logits of the 8 heads, dropout multipliers
        for (int k = tid; k < etot; k += 64 * NW) {
            const int i = t_inst[k], v = t_v[k];
            const int e = a.in_idx[i][t_pos[k]];
            const int src = a.esrc[i][e];
            float el[8], er[8], mk[8];
            ld8(a.eLs[i] + (size_t)src * 8, el);
            ld8(a.eRd[i] + (size_t)v * 8, er);
            if (a.Mk[i] != nullptr) ld8(a.Mk[i] + (size_t)e * 8, mk);
            else {
#pragma unroll
                for (int h = 0; h < 8; ++h) mk[h] = 1.f;
            }
            t_e[k] = e;
            e_src[k] = a.src_row0[i] + src;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float sv = el[h] + er[h];
                t_s[k * 8 + h] = sv > 0.f ? sv : a.slope * sv;
                t_m[k * 8 + h] = mk[h];
            }
        }
        __syncthreads();
        // item = thread again: edge soft-max per (instance, destination, head) over the instance's run of edges
        {
            const int s = tid >> 6, r = tid & 63;
            int k0 = rp[s * (WR + 1) + r];
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int n = deg[q];
                if (n > 0) {
                    float m[8], z[8];
#pragma unroll
                    for (int h = 0; h < 8; ++h) { m[h] = -INFINITY; z[h] = 0.f; }
                    for (int j = 0; j < n; ++j)
#pragma unroll
                        for (int h = 0; h < 8; ++h) m[h] = fmaxf(m[h], t_s[(k0 + j) * 8 + h]);
                    for (int j = 0; j < n; ++j)
#pragma unroll
                        for (int h = 0; h < 8; ++h) z[h] += expf(t_s[(k0 + j) * 8 + h] - m[h]);
                    const int i = t_inst[k0];
                    for (int j = 0; j < n; ++j) {
                        float p[8];
#pragma unroll
                        for (int h = 0; h < 8; ++h) p[h] = expf(t_s[(k0 + j) * 8 + h] - m[h]) / z[h];
                        if (ch == 0) {                              // the soft-max values the backward reads
                            float* ap = a.A[i] + (size_t)t_e[k0 + j] * 8;
                            *reinterpret_cast<float4*>(ap) = make_float4(p[0], p[1], p[2], p[3]);
                            *reinterpret_cast<float4*>(ap + 4) = make_float4(p[4], p[5], p[6], p[7]);
                        }
#pragma unroll
                        for (int h = 0; h < 8; ++h) e_al[(k0 + j) * 8 + h] = (_Float16)(p[h] * t_m[(k0 + j) * 8 + h]);
                    }
                    k0 += n;
                }
            }
        }
    }
    __syncthreads();                                                          // list complete; the scratch inside Y is free
    WT(1);

    f32x16 acc[8][2];
#pragma unroll
    for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][tt][r] = 0.f;

    const int c8 = 8 * l31;                                                   // phase A: this lane's 8 columns (piece l31)
    const bool cok = c8 < D;
    constexpr int RW = WR / NW;                                               // rows of a wave in phase A
    const int rb = RW * wave;

    // source rows of the wave's first RB edges of a slot, requested one slot AHEAD (they land under the previous slot's
    // matrix phase) and kept in registers for both head groups of the slot
    uint4 xbuf[RB];
    auto xload = [&](int ms2) {
        const int* rp2 = rp + ms2 * (WR + 1);
        const int kb2 = uni(rp2[rb]), ke2 = uni(rp2[rb + RW]);
        const unsigned short* __restrict__ xs = a.x16[a.slot_mod[t][ms2]] + c8;
        int src[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) src[u] = e_src[min(kb2 + u, ECAP - 1)];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            xbuf[u] = make_uint4(0u, 0u, 0u, 0u);
            if (kb2 + u < ke2 && cok) xbuf[u] = *reinterpret_cast<const uint4*>(xs + (size_t)src[u] * D);
        }
    };
    if (!slow && nslot > 0) xload(0);

    for (int ms = 0; ms < nslot; ++ms) {
        const int m = a.slot_mod[t][ms];
        const unsigned short* __restrict__ x16m = a.x16[m];
        const unsigned short* __restrict__ wf = a.Wf[m] + (size_t)ct * 8 * KS * 512 + lane * 8;
        const int* rpm = rp + ms * (WR + 1);
#pragma unroll
        for (int hg = 0; hg < 2; ++hg) {
            // weight fragments of this pass: the first PF requested now, they land while phase A runs
            bf16x8 Aq[NS];
            const unsigned short* wsrc = wf + (size_t)(HG * hg) * KS * 512;
#pragma unroll
            for (int i = 0; i < PF; ++i) Aq[i] = *reinterpret_cast<const bf16x8*>(wsrc + (size_t)i * 512);

            // ---- phase A: rows [16 wave, 16 wave + 16); lane = (head pair half, 8 columns)
            {
                // ya[k][x]: columns 2x, 2x + 1 of head 2 half + k as packed pairs (v_pk_fma_f32); branch-free edge / row bodies: every
                // uniform branch inside them costs the in-order wave a drained wait at the block boundary
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 ya[2][4];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int x = 0; x < 4; ++x) ya[k][x] = f2{0.f, 0.f};
                auto flush = [&](int r) {
                    const int pc = (l31 ^ (r & 15)) * 8;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        uint4 o;
                        o.x = srec_pack_bf16(ya[k][0][0], ya[k][0][1]); o.y = srec_pack_bf16(ya[k][1][0], ya[k][1][1]);
                        o.z = srec_pack_bf16(ya[k][2][0], ya[k][2][1]); o.w = srec_pack_bf16(ya[k][3][0], ya[k][3][1]);
                        if (cok) *reinterpret_cast<uint4*>(Y + ((size_t)(2 * half + k) * WR + r) * D + pc) = o;
#pragma unroll
                        for (int x = 0; x < 4; ++x) ya[k][x] = f2{0.f, 0.f};
                    }
                };
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                auto fma2 = [&](const uint4 xv, const unsigned al2) {
                    const h2 alh = __builtin_bit_cast(h2, al2);
                    const float a0 = (float)alh[0], a1 = (float)alh[1];
                    const unsigned w[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const f2 f = {__uint_as_float(w[x] << 16), __uint_as_float(w[x] & 0xffff0000u)};
                        ya[0][x] += f * a0; ya[1][x] += f * a1;
                    }
                };
                if (!slow) {
                    const int kb = uni(rpm[rb]), ke = uni(rpm[rb + RW]);
                    const int rpv = rpm[rb + min(lane, RW)];                   // lane i <= RW: first edge of row rb + i
                    unsigned al[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        al[u] = *reinterpret_cast<const unsigned*>(e_al + min(kb + u, ECAP - 1) * 8 + HG * hg + 2 * half);
                    int r = rb;
                    int rnext = __builtin_amdgcn_readlane(rpv, 1);
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int k = kb + u;
                        if (k < ke) {
                            while (k >= rnext) { flush(r); ++r; rnext = __builtin_amdgcn_readlane(rpv, r - rb + 1); }
                            fma2(xbuf[u], al[u]);
                        }
                    }
                    // a wave with more than RB edges in this slot: the rest on demand, four rows in flight
                    for (int k0 = kb + RB; k0 < ke; k0 += 4) {
                        uint4 cur[4]; unsigned alc[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cur[u] = make_uint4(0u, 0u, 0u, 0u);
                            const int k = min(k0 + u, ke - 1);
                            if (cok) cur[u] = *reinterpret_cast<const uint4*>(x16m + (size_t)e_src[k] * D + c8);
                            alc[u] = *reinterpret_cast<const unsigned*>(e_al + k * 8 + HG * hg + 2 * half);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = k0 + u;
                            if (k < ke) {
                                while (k >= rnext) { flush(r); ++r; rnext = __builtin_amdgcn_readlane(rpv, r - rb + 1); }
                                fma2(cur[u], alc[u]);
                            }
                        }
                    }
                    for (; r < rb + RW; ++r) flush(r);
                } else {
                    // general path (a window with more in-edges than the LDS list holds): row by row from global memory
                    for (int r = rb; r < rb + RW; ++r) {
                        const int v = w0 + r;
                        if (v < nlive) {
                            for (int q = 0; q < a.slot_nq[t][ms]; ++q) {
                                const int i = a.slot_inst[t][ms][q];
                                const int b0 = a.in_ptr[i][v], n = a.in_ptr[i][v + 1] - b0;
                                const int* idx = a.in_idx[i] + b0;
                                float m2[2] = {-INFINITY, -INFINITY}, z2[2] = {0.f, 0.f};
                                const int h0 = HG * hg + 2 * half;
                                const float er0 = a.eRd[i][(size_t)v * 8 + h0], er1 = a.eRd[i][(size_t)v * 8 + h0 + 1];
                                for (int j = 0; j < n; ++j) {
                                    const int src = a.esrc[i][idx[j]];
                                    float s0 = a.eLs[i][(size_t)src * 8 + h0] + er0, s1 = a.eLs[i][(size_t)src * 8 + h0 + 1] + er1;
                                    s0 = s0 > 0.f ? s0 : a.slope * s0; s1 = s1 > 0.f ? s1 : a.slope * s1;
                                    m2[0] = fmaxf(m2[0], s0); m2[1] = fmaxf(m2[1], s1);
                                }
                                for (int j = 0; j < n; ++j) {
                                    const int src = a.esrc[i][idx[j]];
                                    float s0 = a.eLs[i][(size_t)src * 8 + h0] + er0, s1 = a.eLs[i][(size_t)src * 8 + h0 + 1] + er1;
                                    s0 = s0 > 0.f ? s0 : a.slope * s0; s1 = s1 > 0.f ? s1 : a.slope * s1;
                                    z2[0] += expf(s0 - m2[0]); z2[1] += expf(s1 - m2[1]);
                                }
                                for (int j = 0; j < n; ++j) {
                                    const int e = idx[j], src = a.esrc[i][e];
                                    float s0 = a.eLs[i][(size_t)src * 8 + h0] + er0, s1 = a.eLs[i][(size_t)src * 8 + h0 + 1] + er1;
                                    s0 = s0 > 0.f ? s0 : a.slope * s0; s1 = s1 > 0.f ? s1 : a.slope * s1;
                                    float p0 = expf(s0 - m2[0]) / z2[0], p1 = expf(s1 - m2[1]) / z2[1];
                                    if (ch == 0 && l31 == 0) { a.A[i][(size_t)e * 8 + h0] = p0; a.A[i][(size_t)e * 8 + h0 + 1] = p1; }
                                    if (a.Mk[i] != nullptr) { p0 *= a.Mk[i][(size_t)e * 8 + h0]; p1 *= a.Mk[i][(size_t)e * 8 + h0 + 1]; }
                                    {
                                        typedef _Float16 h2s __attribute__((ext_vector_type(2)));
                                        const h2s pp = {(_Float16)p0, (_Float16)p1};
                                        uint4 xv = make_uint4(0u, 0u, 0u, 0u);
                                        if (cok) xv = *reinterpret_cast<const uint4*>(x16m + (size_t)(a.src_row0[i] + src) * D + c8);
                                        fma2(xv, __builtin_bit_cast(unsigned, pp));
                                    }
                                }
                            }
                        }
                        flush(r);
                    }
                }
            }
            WT(2);
            __syncthreads();                                                  // the operand tile is written
            WT(3);
            // the next slot's source rows: requested now, they land under this pass's matrix phase
            if (!slow && hg == 1 && ms + 1 < nslot) xload(ms + 1);

            // ---- phase B: HG heads x KS k-steps, every fragment feeds both 32-row tiles; the tile rows of step i + 1 are read
            //      from LDS before the MFMAs of step i are issued (two MFMAs do not cover an LDS round trip)
            constexpr int BD = 4, BR = BD + 1;                                // tile rows requested BD k-steps ahead
            bf16x8 Bq[BR][2];
            auto bread = [&](int i, bf16x8 (&dst)[2]) {
                const int hh = i / KS, s = i % KS;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    dst[tt] = *reinterpret_cast<const bf16x8*>(Y + ((size_t)hh * WR + 32 * tt + l31) * D + (((2 * s + half) ^ (l31 & 15)) * 8));
            };
#pragma unroll
            for (int i = 0; i < BD; ++i) bread(i, Bq[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HG * KS; ++i) {
                const int nx = i + PF < HG * KS ? i + PF : HG * KS - 1;
#if !(SREC_HGWIN_KO & 1)
                Aq[(i + PF) % NS] = *reinterpret_cast<const bf16x8*>(wsrc + (size_t)nx * 512);
#endif
#if !(SREC_HGWIN_KO & 2)
                bread(i + BD < HG * KS ? i + BD : HG * KS - 1, Bq[(i + BD) % BR]);
#endif
                const int hh = i / KS;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[HG * hg + hh][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[i % NS], Bq[i % BR][tt], acc[HG * hg + hh][tt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            WT(4);
            __syncthreads();                                                  // the operand tile has been read
            WT(5);
        }
    }

    // ---------------------------------------------------------------- epilogue: + bias + residual, head max, + session mean
    // (every load of the epilogue is requested before the first is used: res / session-mean rows of the lane's two rows)
    const float nres = (float)a.ninst[t];
    float4 res[2][4], smv[2][4];
    bool rlive[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int r = 32 * tt + l31, v = w0 + r;
        rlive[tt] = v < nlive;
        const size_t grow = (size_t)(grow0 + r);
        const int sb = rlive[tt] ? sess2[tt] : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * ct + 8 * g + 4 * half;
            res[tt][g] = make_float4(0.f, 0.f, 0.f, 0.f); smv[tt][g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rlive[tt]) {
                res[tt][g] = *reinterpret_cast<const float4*>((a.xres != nullptr ? a.xres : a.x) + grow * a.ld_x + col);
                smv[tt][g] = *reinterpret_cast<const float4*>(a.smean[t] + (size_t)sb * D + col);
            }
        }
    }
    const float rsc = a.xres != nullptr ? 1.f : nres;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int r = 32 * tt + l31, v = w0 + r;
        if (v >= ncap) continue;
        const size_t grow = (size_t)(grow0 + r);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * ct + 8 * g + 4 * half;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned ab = 0u;
            if (rlive[tt]) {
                const float4 rs = make_float4(res[tt][g].x * rsc, res[tt][g].y * rsc, res[tt][g].z * rsc, res[tt][g].w * rsc);
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias_s + h * (NW * 32) + 32 * wave + 8 * g + 4 * half);
                    const float v0 = acc[h][tt][4 * g] + bv.x + rs.x, v1 = acc[h][tt][4 * g + 1] + bv.y + rs.y;
                    const float v2 = acc[h][tt][4 * g + 2] + bv.z + rs.z, v3 = acc[h][tt][4 * g + 3] + bv.w + rs.w;
                    if (v0 > best[0]) { best[0] = v0; bi[0] = h; }
                    if (v1 > best[1]) { best[1] = v1; bi[1] = h; }
                    if (v2 > best[2]) { best[2] = v2; bi[2] = h; }
                    if (v3 > best[3]) { best[3] = v3; bi[3] = h; }
                }
                o = make_float4(best[0] + smv[tt][g].x, best[1] + smv[tt][g].y, best[2] + smv[tt][g].z, best[3] + smv[tt][g].w);
                ab = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
            }
            *reinterpret_cast<float4*>(a.out + grow * a.ld_out + col) = o;
            *reinterpret_cast<unsigned*>(a.arg + grow * D + col) = ab;
        }
    }
#ifdef SREC_HGWIN_TIMING
    WT(6);
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
        g_hgwin_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 8; ++i) g_hgwin_tim[blockIdx.x][i] = tim_t[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- weight fragments
struct WfragArgs {
    const float* W[MAXM];
    unsigned short* F[MAXM];
    unsigned short* T[MAXM];
    int H, D;
};

// fragment-major bf16 copies of fc.weight W [H D, D] (fp32, row hD + j = output column j of head h):
//   F [ct][h][s][lane][8]: W[h D + 32 ct + (lane & 31)][16 s + 8 (lane >> 5) + i]   - A operand of the forward product
//   T [jt][h][s][lane][8]: W[h D + 16 s + 8 (lane >> 5) + i][32 jt + (lane & 31)]   - A operand of the backward-data product
__global__ __launch_bounds__(256) void hg_wfrag_kernel(WfragArgs a) {
    const int D = a.D, H = a.H, KS = D / 16, NCT = D / 32;
    const int m = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * D * D / 8) return;
    const int lane = idx & 63, f = idx >> 6;
    const int s = f % KS, h = (f / KS) % H, ct = f / (KS * H);
    (void)NCT;
    const float* W = a.W[m];
    const int kk = 16 * s + 8 * (lane >> 5);
    if (a.F[m] != nullptr) {
        const float* p = W + (size_t)(h * D + 32 * ct + (lane & 31)) * D + kk;
        const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
        *reinterpret_cast<uint4*>(a.F[m] + (size_t)idx * 8) =
            make_uint4(srec_pack_bf16(v0.x, v0.y), srec_pack_bf16(v0.z, v0.w), srec_pack_bf16(v1.x, v1.y), srec_pack_bf16(v1.z, v1.w));
    }
    if (a.T[m] != nullptr) {
        const float* p = W + (size_t)(h * D + kk) * D + 32 * ct + (lane & 31);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(size_t)i * D];
        *reinterpret_cast<uint4*>(a.T[m] + (size_t)idx * 8) =
            make_uint4(srec_pack_bf16(v[0], v[1]), srec_pack_bf16(v[2], v[3]), srec_pack_bf16(v[4], v[5]), srec_pack_bf16(v[6], v[7]));
    }
}

bool win_bad_desc(const srec_hg_desc* d) {
    return d == nullptr || d->H != 8 || (d->D != 128 && d->D != 256) || d->n_types <= 0 || d->n_types > MAXT || d->n_blocks < 0 ||
           d->n_blocks > SREC_HG_MAXB || d->n_inst < 0 || d->n_inst > MAXI || d->n_mods < 0 || d->n_mods > MAXM;
}

// slots of every node type: the modules with a relation instance INTO the type, in instance order
int win_slots(const srec_hg_desc* d, WinArgs& g) {
    for (int t = 0; t < d->n_types; ++t) { g.nslot[t] = 0; g.ninst[t] = 0; }
    for (int i = 0; i < d->n_inst; ++i) {
        const int m = d->inst_mod[i], t = d->blk_type[d->inst_dblk[i]];
        g.ninst[t]++;
        int s = -1;
        for (int q = 0; q < g.nslot[t]; ++q)
            if (g.slot_mod[t][q] == m) s = q;
        if (s < 0) {
            if (g.nslot[t] >= MAXS) return SREC_BAD_ARG;
            s = g.nslot[t]++;
            g.slot_mod[t][s] = m;
            g.slot_nq[t][s] = 0;
        }
        if (g.slot_nq[t][s] >= MAXQ) return SREC_BAD_ARG;
        g.slot_inst[t][s][g.slot_nq[t][s]++] = i;
    }
    return 0;
}

}  // namespace

#ifdef SREC_HGWIN_TIMING
extern "C" int srec_hgwin_timing(unsigned long long* tim, unsigned long long* blk) {
    if (hipMemcpyFromSymbol(tim, HIP_SYMBOL(g_hgwin_tim), sizeof(unsigned long long) * 2048 * 8) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(blk, HIP_SYMBOL(g_hgwin_blk), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : 1;
}
#endif

// n <= 8 fc weights W_i [H D, D] fp32 (HOST arrays of device pointers) -> fragment-major bf16 copies F_i (forward operand) and
// T_i (backward-data operand), each [H D D] bf16; F / T nullable per array (NULL array = none).  H = 8, D % 32 == 0.
extern "C" int srec_hg_wfrag(int n, const void* W, const void* F, const void* T, int H, int D, void* stream) {
    if (n <= 0) return 0;
    if (n > MAXM || W == nullptr || H <= 0 || D <= 0 || (D % 32)) return SREC_BAD_ARG;
    WfragArgs a{};
    a.H = H; a.D = D;
    for (int i = 0; i < n; ++i) {
        a.W[i] = ((const float* const*)W)[i];
        a.F[i] = F != nullptr ? ((unsigned short* const*)F)[i] : nullptr;
        a.T[i] = T != nullptr ? ((unsigned short* const*)T)[i] : nullptr;
        if (a.W[i] == nullptr) return SREC_BAD_ARG;
    }
    hipLaunchKernelGGL(hg_wfrag_kernel, dim3((H * D * D / 8 + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// forward of the layer without projections: desc as srec_hg_fwd, plus x16[m] (bf16 input rows of module m's conv) and Wf[m]
// (srec_hg_wfrag); P is not read.  H = 8, D = 128 / 256, <= 4 modules with an instance into any node type.
extern "C" int srec_hg_win_fwd(const void* desc_, const float* x, int ld_x, float* out, int ld_out, unsigned char* arg,
                               int force_slow, void* stream) {
    const srec_hg_desc* d = (const srec_hg_desc*)desc_;
    if (win_bad_desc(d) || (ld_x & 3) || (ld_out & 3) || d->sess == nullptr) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = srec_hg_logits(desc_, x, ld_x, stream)) return rc;
    const int D = d->D;
    WinArgs g{};
    g.nt = d->n_types; g.slope = d->slope; g.x = x; g.ld_x = ld_x; g.xres = d->xres; g.sess = d->sess;
    g.out = out; g.ld_out = ld_out; g.arg = arg; g.force_slow = force_slow;
    if (int rc = win_slots(d, g)) return rc;
    int rows = 0, wins = 0;
    for (int t = 0; t < d->n_types; ++t) {
        if (d->row0[t] != rows) return SREC_BAD_ARG;
        g.row0[t] = d->row0[t]; g.ncap[t] = d->ncap[t]; g.dyn_n[t] = d->dyn_n[t];
        g.bsum[t] = d->Z[t]; g.smean[t] = d->smean[t];
        if (g.smean[t] == nullptr || (g.ninst[t] > 0 && g.bsum[t] == nullptr)) return SREC_BAD_ARG;
        g.wstart[t] = wins;
        wins += cdiv(d->ncap[t], WR);
        rows += d->ncap[t];
    }
    g.row0[d->n_types] = rows; g.wstart[d->n_types] = wins;
    for (int m = 0; m < d->n_mods; ++m) {
        g.x16[m] = (const unsigned short*)d->x16[m]; g.Wf[m] = (const unsigned short*)d->Wf[m];
        if (g.x16[m] == nullptr || g.Wf[m] == nullptr) return SREC_BAD_ARG;
    }
    for (int i = 0; i < d->n_inst; ++i) {
        const int sb = d->inst_sblk[i], db = d->inst_dblk[i];
        g.eLs[i] = d->eL[sb]; g.eRd[i] = d->eR[db]; g.Mk[i] = d->Mk[i];
        g.in_ptr[i] = d->in_ptr[i]; g.in_idx[i] = d->in_idx[i]; g.esrc[i] = d->esrc[i];
        g.A[i] = d->A[i];
        g.src_row0[i] = d->row0[d->blk_type[sb]];
    }
    if (wins <= 0) return 0;
    static std::atomic<unsigned long long> om[2];
    if (D == 256) {
        if (int rc = srec_lds_optin((const void*)hg_win_fwd_kernel<256>, FwdLds<256>::TOTAL, om[0])) return rc;
        hipLaunchKernelGGL(hg_win_fwd_kernel<256>, dim3(cdiv(wins * 2, 8) * 8), dim3(64 * NW), FwdLds<256>::TOTAL, st, g);
    } else {
        if (int rc = srec_lds_optin((const void*)hg_win_fwd_kernel<128>, FwdLds<128>::TOTAL, om[1])) return rc;
        hipLaunchKernelGGL(hg_win_fwd_kernel<128>, dim3(cdiv(wins, 8) * 8), dim3(64 * NW), FwdLds<128>::TOTAL, st, g);
    }
    SREC_LAUNCH_CHECK();
    return 0;
}
