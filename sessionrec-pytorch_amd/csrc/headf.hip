// Attention read-out + session vector of MSGIFSR (msgifsr.py:124-155 AttnReadout, :269-273 fc_sr + F.normalize) for a GROUP OF
// SESSIONS per workgroup, forward in ONE launch (bf16 mode, d = 128 or 256):
//     Vq_b = v_b Wv^T;  U_i = x_i Wu^T + bu;  e_i = we . sigmoid(U_i + Vq_b(i));  alpha = softmax over the session's nodes;
//     g_b = sum_i alpha_i x_i;  s_b = [v_b | g_b] Wsr^T;  y_b = s_b / max(|s_b|, eps)   (+ the bf16 operand copy of y for the
//     scoring kernels)
// As grouped batch-wide launches (ops.ReadoutHead: {U, Vq} GEMM, read-out kernel, {s} GEMM, split-K sum, normalise) these are
// five latency-bound kernel nodes of ~50 us for 0.9 GFLOP.  Here a workgroup OWNS the sessions that start in its window of
// HWIN = 32 rows of the per-session concatenation `allf` (work is dealt by ROWS: 8 consecutive sessions of the bench batch hold
// 20 - 300 rows, and with one round of workgroups the launch is as long as its longest one; 32-row windows are ~220 workgroups
// on the 256 CUs at the bench shape - with 64-row windows half of the chip idled: forward 36.8 -> 32.8 us, backward 33.2 ->
// 29.5 us, medians of 60 replays) - a session's rows are contiguous - and runs the whole chain on them, HS = 16 sessions per
// pass, their rows in chunks of HR = 64:
//   * every product is computed TRANSPOSED on the bf16 matrix pipe, D^T = W . X^T with v_mfma_f32_32x32x16_bf16: the A
//     operand is a 32-column block of the weight, the B operand 32 rows of activations, so the result puts a ROW in the lane
//     and 16 hidden columns in its registers: the per-row reductions that follow (we . sigmoid(.), |s|^2) are sums over a
//     lane's own registers + one exchange between the wave halves - no LDS transpose, and U never exists in memory;
//   * exact-fp32 results on bf16 operands: both operands are split x = hi + lo (bf16 each, |lo| <= 2^-9 |x|) and the
//     product is the three terms hi hi + lo hi + hi lo (the dropped lo lo term is 2^-18 relative).  fp32 MFMA runs at 1/16
//     of the bf16 rate on gfx950: three bf16 products cost 3/16 - the kernel is then bound by what feeds the pipe;
//   * the weights stream from L2 in FRAGMENT-MAJOR hi / lo copies (srec_head_wfrag, once per step: the 64 lanes x 16 B of one
//     MFMA operand contiguous, ordered wave / k-step / {hi, lo} / column block), plain 1-KiB coalesced loads straight into
//     the registers that feed the MFMAs through a 4-stage register ring (the idiom of gruf.hip); ~220 workgroups x
//     (Wv 256 KB + 1 - 2 chunks x Wu 256 KB + Wsr 512 KB) ~ 250 MB out of L2 per launch at B = 512, d = 256;
//   * activations (32-row chunks of the group's node rows, the 8 query rows, the [v | g] rows) are split while they are staged
//     into LDS (row-major bf16, 16-B pieces XOR-swizzled by the row: conflict-free ds_read_b128).
// Saved for the backward: alpha [NT], cat = [v | g] [B, 2 d], Vq [B, d], y [B, d], 1 / |s| [B]; U only on request (the
// grouped backward of ops.ReadoutHead reads it; the fused backward recomputes it).
#include "common.h"
#include "../../include/srec_hg.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int HS = SREC_HEAD_SESSIONS;   // sessions per pass of a workgroup (MFMA columns of the B x d products)
constexpr int HR = SREC_HEAD_ROWS;       // rows of the per-session concatenation per chunk (2 MFMA tiles)
constexpr int HWIN = SREC_HEAD_WINDOW;   // rows per workgroup window: a workgroup owns the sessions that START in its window
constexpr int NW = 4;                    // waves per workgroup: wave w owns the hidden / output columns [w d/4, (w+1) d/4)
constexpr int NS = 4, PF = NS - 1;       // register ring of weight fragments: stages, k-steps in flight
constexpr int MAXN = SREC_MAX_SESSION_NODES;
constexpr int VQ_PAD = 8;                // floats of padding per Vq row in LDS: (session, wave half) -> distinct 16-B slots

struct HeadArgs {
    srec_head_desc d;
};

#ifdef SREC_HEADF_TIMING   // development probe (tools/headf_timing.py): phase clocks of wave 0 of every workgroup, workgroup lives
__device__ unsigned long long g_headf_tim[1024][8];
__device__ unsigned long long g_headf_blk[1024][2];
#define HFT(i) do { __builtin_amdgcn_sched_barrier(0); tim_t[i] += __builtin_readcyclecounter() - tim_c; \
    tim_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HFT(i)
#endif

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = srec_pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = srec_pack_bf16(a - ah, b - bh);
}

// 4 consecutive fp32 -> the hi / lo tiles (row-major [rows][W] bf16, 16-B pieces swizzled by row & 15), column c (multiple of 4)
__device__ __forceinline__ void stage4(unsigned short* hi_t, unsigned short* lo_t, int W, int row, int c, float4 v) {
    uint2 h, l;
    split2(v.x, v.y, h.x, l.x);
    split2(v.z, v.w, h.y, l.y);
    const int off = row * W + (((c >> 3) ^ (row & 15)) * 8) + (c & 4);
    *reinterpret_cast<uint2*>(hi_t + off) = h;
    *reinterpret_cast<uint2*>(lo_t + off) = l;
}

// D^T (+)= W . B^T over T k-steps on the bf16 matrix pipe, three terms of the hi / lo split per k-step: A = fragment-major weight
// (per wave and k-step: JBV hi fragments, then JBV lo fragments - the wave's JBV 32-row blocks of the operand matrix) through
// a register ring of NS stages fed by plain 1-KiB loads, B = NB 32-row tiles of the hi / lo LDS image bh / bl (row stride W
// elements, 16-B pieces swizzled by row & 15; tile t = rows brow + 32 t).  Result: lane = B row, registers = operand rows
// (r & 3) + 8 (r >> 2) + 4 half of block j.  ZERO: start from zero accumulators (else: accumulate onto acc).
template <int NB, int JBV, bool ZERO = true>
__device__ __forceinline__ void hl_product(f32x16 (&acc)[2][JBV], const unsigned short* __restrict__ wf, const int T,
                                           const unsigned short* bh, const unsigned short* bl, const int W, const int brow,
                                           const int wave, const int lane) {
    constexpr int NFV = 2 * JBV;
    const int half = lane >> 5;
    if (ZERO) {
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int j = 0; j < JBV; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    }
    bf16x8 Aq[NS][NFV];
    const unsigned short* wsrc = wf + (size_t)wave * T * NFV * 512 + lane * 8;
    auto load = [&](int i, int slot) {
        i = min(i, T - 1);
#pragma unroll
        for (int f = 0; f < NFV; ++f) Aq[slot][f] = *reinterpret_cast<const bf16x8*>(wsrc + ((size_t)i * NFV + f) * 512);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load(i, i);
#pragma unroll 1
    for (int ib = 0; ib < T; ib += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            load(ib + u + PF, (u + PF) % NS);
            const int s = ib + u;
            bf16x8 Bh[NB], Bl_[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int rw = brow + 32 * t;
                const int off = rw * W + (((2 * s + half) ^ (rw & 15)) * 8);
                Bh[t] = *reinterpret_cast<const bf16x8*>(bh + off);
                Bl_[t] = *reinterpret_cast<const bf16x8*>(bl + off);
            }
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int j = 0; j < JBV; ++j) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[u][JBV + j], Bh[t], acc[t][j], 0, 0, 0);     // lo hi
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[u][j], Bl_[t], acc[t][j], 0, 0, 0);          // hi lo
                    acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[u][j], Bh[t], acc[t][j], 0, 0, 0);           // hi hi
                }
            __builtin_amdgcn_sched_group_barrier(0x020, NFV, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * JBV * NB, 0);
        }
    }
}

template <int DD>
__global__ __launch_bounds__(64 * NW, 1) void head_fwd_kernel(HeadArgs a) {
    constexpr int D = 128 * DD, JB = D / (32 * NW), KS = D / 16, NF = 2 * JB;     // NF: fragments per (wave, k-step): hi | lo
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* x_hi = sm;                               // [HR][D]   node-row chunk (two 32-row MFMA tiles)
    unsigned short* x_lo = x_hi + HR * D;
    unsigned short* c_hi = x_lo + HR * D;                    // [HS][2 D] [v | g] of the pass's sessions
    unsigned short* c_lo = c_hi + HS * 2 * D;
    float* vq = reinterpret_cast<float*>(c_lo + HS * 2 * D); // [HS][D + VQ_PAD]
    float* bw = vq + HS * (D + VQ_PAD);                      // [2][D] fc_u bias | fc_e weight
    float* e = bw + 2 * D;                                   // [HR + MAXN] logits, then soft-max weights, of the pass's rows
    float* epart = e + HR + MAXN;                            // [NW][HR]
    int* segs = reinterpret_cast<int*>(epart + NW * HR);     // [HS + 1] first row of each session (+ end), relative to r0
    int* rsess = segs + HS + 1;                              // [HR] session (0 .. HS-1) of each row of the chunk
    int* cnt = rsess + HR;                                   // [2] sessions that start before / inside this row window
    int* segl = cnt + 2;                                     // [B + 1] copy of seg[]

    const srec_head_desc& q = a.d;
    const int hd = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Bl = dyn_count(q.dynB, q.B);
    float* cat = q.cat[hd];
    float* Y = q.y[hd];
    unsigned short* Y16 = (unsigned short*)q.y16[hd];
    const int ld16 = q.ld16;

    // this workgroup owns the sessions whose FIRST row lies in its window of HWIN rows of the per-session concatenation: work
    // is dealt by rows, not by sessions (a group of 8 sessions has 20 - 300 rows in the bench batch).  One pass over seg[]
    // (every thread a few entries, kept in LDS for the rest of the kernel) counts the sessions that start before / inside it
    const int w0 = (int)blockIdx.x * HWIN, w1 = w0 + HWIN;
#ifdef SREC_HEADF_TIMING
    unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_c = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_headf_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        for (int b = tid; b <= Bl; b += NT) {
            const int sb = q.seg[b];
            segl[b] = sb;
            c0 += (b < Bl && sb < w0) ? 1 : 0;
            c1 += (b < Bl && sb < w1) ? 1 : 0;
        }
        for (int c = tid; c < D; c += NT) {
            bw[c] = q.bu[hd] != nullptr ? q.bu[hd][c] : 0.f;
            bw[D + c] = q.we[hd][c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); }
        if (lane == 0) { atomicAdd(&cnt[0], c0); atomicAdd(&cnt[1], c1); }
    }
    // capacity padding (sessions past the live count): zero rows where the grouped path writes zeros - the read-out half of
    // cat, y, 1 / |s|, Vq, the bf16 operand; shared by all workgroups
    for (int b = Bl + (int)blockIdx.x; b < q.B; b += (int)gridDim.x) {
        for (int c = tid * 4; c < D; c += NT * 4) {
            *reinterpret_cast<float4*>(cat + (size_t)b * 2 * D + D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Y + (size_t)b * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Y16 != nullptr) *reinterpret_cast<uint2*>(Y16 + (size_t)b * ld16 + c) = make_uint2(0u, 0u);
            if (q.Vq[hd] != nullptr) *reinterpret_cast<float4*>(q.Vq[hd] + (size_t)b * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid == 0) q.inv[hd][b] = 0.f;
    }
    __syncthreads();
    if (Bl <= 0 || w0 >= segl[Bl]) return;
    const int bfirst = cnt[0], bend = cnt[1];
    const int cbase = wave * 32 * JB;
    const float* X = q.X;
    const int ld_x = q.ld_x;
    float* Uout = q.U[hd];
    auto sig = [](float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); };

    f32x16 acc[2][JB];

    for (int b0 = bfirst; b0 < bend; b0 += HS) {             // passes of HS sessions (one, unless sessions are very short)
        const int ns = min(HS, bend - b0);
        __syncthreads();                                     // LDS of the previous pass has been read
        const int r0 = segl[b0];
        if (tid <= HS) segs[tid] = segl[b0 + min(tid, ns)] - r0;
        const int nrows = segl[b0 + ns] - r0;
        // ---- the pass's query rows v_b (left half of cat) into the [v | g] tile
        for (int i = tid; i < HS * (D / 4); i += NT) {
            const int row = i / (D / 4), c = (i % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < ns) v = *reinterpret_cast<const float4*>(cat + (size_t)(b0 + row) * 2 * D + c);
            stage4(c_hi, c_lo, 2 * D, row, c, v);
        }
        // first chunk of node rows: in flight under the Vq product
        constexpr int XV = HR * D / 4 / NT;
        float4 xv[XV];
        auto fetch_x = [&](int c0) {
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                const int idx = i * NT + tid;
                const int row = idx / (D / 4), c4 = idx % (D / 4);
                xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + row < nrows) xv[i] = *reinterpret_cast<const float4*>(X + (size_t)(r0 + c0 + row) * ld_x + c4 * 4);
            }
        };
        fetch_x(0);
        __syncthreads();
        HFT(0);

        // ---- Vq^T = Wv . v^T: lane = session (l31 & (HS - 1)), registers = this wave's hidden columns
        hl_product<1, JB>(acc, (const unsigned short*)q.Wv_f[hd], KS, c_hi, c_lo, 2 * D, l31 & (HS - 1), wave, lane);
        HFT(1);
        if (l31 < HS) {
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = cbase + 32 * j + 8 * g + 4 * half;
                    const float4 v = make_float4(acc[0][j][4 * g], acc[0][j][4 * g + 1], acc[0][j][4 * g + 2], acc[0][j][4 * g + 3]);
                    *reinterpret_cast<float4*>(vq + l31 * (D + VQ_PAD) + col) = v;
                    if (q.Vq[hd] != nullptr && l31 < ns) *reinterpret_cast<float4*>(q.Vq[hd] + (size_t)(b0 + l31) * D + col) = v;
                }
        }

        // ---- node rows in chunks of HR = 2 x 32: U^T = Wu . X^T (every weight fragment feeds both tiles),
        //      e_i = we . sigmoid(U_i + bu + Vq_b(i))
        for (int c0 = 0; c0 < nrows; c0 += HR) {
            __syncthreads();                     // the previous chunk's tile / rsess / epart have been read; vq is published
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                const int idx = i * NT + tid;
                stage4(x_hi, x_lo, D, idx / (D / 4), (idx % (D / 4)) * 4, xv[i]);
            }
            if (tid < HR) {
                int s = 0;
#pragma unroll
                for (int k = 1; k < HS; ++k) s += (c0 + tid >= segs[k]) ? 1 : 0;
                rsess[tid] = min(s, ns - 1);
            }
            if (c0 + HR < nrows) fetch_x(c0 + HR);
            __syncthreads();
            HFT(2);
            const bool two = nrows - c0 > 32;
            if (two) hl_product<2, JB>(acc, (const unsigned short*)q.Wu_f[hd], KS, x_hi, x_lo, D, l31, wave, lane);
            else hl_product<1, JB>(acc, (const unsigned short*)q.Wu_f[hd], KS, x_hi, x_lo, D, l31, wave, lane);
            HFT(3);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
                const int row = c0 + 32 * t + l31;
                const float* vrow = vq + rsess[32 * t + l31] * (D + VQ_PAD);
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < JB; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = cbase + 32 * j + 8 * g + 4 * half;
                        const float4 vv = *reinterpret_cast<const float4*>(vrow + col);
                        const float4 bb = *reinterpret_cast<const float4*>(bw + col);
                        const float4 ww = *reinterpret_cast<const float4*>(bw + D + col);
                        const float u0 = acc[t][j][4 * g] + bb.x, u1 = acc[t][j][4 * g + 1] + bb.y;
                        const float u2 = acc[t][j][4 * g + 2] + bb.z, u3 = acc[t][j][4 * g + 3] + bb.w;
                        part += ww.x * sig(u0 + vv.x) + ww.y * sig(u1 + vv.y) + ww.z * sig(u2 + vv.z) + ww.w * sig(u3 + vv.w);
                        if (Uout != nullptr && row < nrows)
                            *reinterpret_cast<float4*>(Uout + (size_t)(r0 + row) * D + col) = make_float4(u0, u1, u2, u3);
                    }
                part += __shfl_xor(part, 32, 64);
                if (half == 0) epart[wave * HR + 32 * t + l31] = part;
            }
            __syncthreads();
            if (tid < HR && c0 + tid < nrows && (two || tid < 32)) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += epart[w * HR + tid];
                e[c0 + tid] = s;
            }
            HFT(4);
        }
        __syncthreads();
        HFT(2);

        // ---- soft-max over each session's rows, read-out row g_b = sum_i alpha_i x_i (node order, as seg_attn_fwd sums it)
        for (int sb = wave; sb < ns; sb += NW) {
            const int base = segs[sb], n = min(segs[sb + 1] - base, MAXN);
            float m = -INFINITY;
            for (int i = lane; i < n; i += 64) m = fmaxf(m, e[base + i]);
            m = wave_max(m);
            float ssum = 0.f;
            for (int i = lane; i < n; i += 64) ssum += expf(e[base + i] - m);
            ssum = wave_sum(ssum);
            const float inv = n > 0 ? 1.f / ssum : 0.f;
            for (int i = lane; i < n; i += 64) {
                const float al = expf(e[base + i] - m) * inv;
                e[base + i] = al;
                q.alpha[hd][r0 + base + i] = al;
            }
            __builtin_amdgcn_s_waitcnt(0);       // this wave's LDS writes of alpha before its own reads below (one wave per session)
            const int c = lane * 4;
            if (c < D) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* xp = X + (size_t)(r0 + base) * ld_x + c;
                int i = 0;
                for (; i + 8 <= n; i += 8) {
                    float4 t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const float4*>(xp + (size_t)(i + k) * ld_x);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float al = e[base + i + k];
                        o.x += al * t[k].x; o.y += al * t[k].y; o.z += al * t[k].z; o.w += al * t[k].w;
                    }
                }
                if (i < n) {                     // tail: the remaining <= 7 rows in flight together
                    float4 t[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) t[k] = *reinterpret_cast<const float4*>(xp + (size_t)min(i + k, n - 1) * ld_x);
#pragma unroll
                    for (int k = 0; k < 7; ++k)
                        if (i + k < n) {
                            const float al = e[base + i + k];
                            o.x += al * t[k].x; o.y += al * t[k].y; o.z += al * t[k].z; o.w += al * t[k].w;
                        }
                }
                *reinterpret_cast<float4*>(cat + (size_t)(b0 + sb) * 2 * D + D + c) = o;
                stage4(c_hi, c_lo, 2 * D, sb, D + c, o);
            }
        }
        if (ns < HS) {                           // tile rows of the pass's missing sessions: zeros (their lanes are never read)
            for (int i = tid; i < (HS - ns) * (D / 4); i += NT)
                stage4(c_hi, c_lo, 2 * D, ns + i / (D / 4), D + (i % (D / 4)) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        __syncthreads();
        HFT(5);

        // ---- s^T = Wsr . [v | g]^T (K = 2 d), y = s / max(|s|, eps)
        hl_product<1, JB>(acc, (const unsigned short*)q.Wsr_f[hd], 2 * KS, c_hi, c_lo, 2 * D, l31 & (HS - 1), wave, lane);
        HFT(6);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ss += acc[0][j][r] * acc[0][j][r];
        ss += __shfl_xor(ss, 32, 64);
        if (half == 0 && l31 < HS) epart[wave * HR + l31] = ss;
        __syncthreads();
        if (l31 < ns) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += epart[w * HR + l31];
            const float nrm = sqrtf(tot);
            const float iv = q.eps_mode == 0 ? 1.f / fmaxf(nrm, q.eps) : 1.f / (nrm + q.eps);
            const int b = b0 + l31;
            if (wave == 0 && half == 0) q.inv[hd][b] = iv;
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = cbase + 32 * j + 8 * g + 4 * half;
                    const float4 v = make_float4(acc[0][j][4 * g] * iv, acc[0][j][4 * g + 1] * iv, acc[0][j][4 * g + 2] * iv,
                                                 acc[0][j][4 * g + 3] * iv);
                    *reinterpret_cast<float4*>(Y + (size_t)b * D + col) = v;
                    if (Y16 != nullptr)
                        *reinterpret_cast<uint2*>(Y16 + (size_t)b * ld16 + col) = make_uint2(srec_pack_bf16(v.x, v.y), srec_pack_bf16(v.z, v.w));
                }
        }
        HFT(7);
    }
#ifdef SREC_HEADF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_headf_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 8; ++i) g_headf_tim[blockIdx.x][i] = tim_t[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------- backward, first half
// Everything of the head's backward that is per SESSION, in one launch over the same row windows as the forward (it replaces
// normalise-backward, the {d cat} GEMM + its split-K sum and srec_seg_attn_bwd of the grouped path, 43 us in four launches):
//     g_s = (g_y - y <y, g_y>) / |s|;   d cat = g_s Wsr  (transposed split product, Wsr^T streamed fragment-major: left half =
//     the direct part of d v, right half = d g);   d alpha_i = <d g_b, x_i>,  d e_i = alpha_i (d alpha_i - sum_j alpha_j d alpha_j);
//     dX_i = alpha_i d g_b;   dU_i = d e_i we sigma'(U_i + Vq_b),  dVq_b = sum_i dU_i,  dwp_b = sum_i d e_i sigma(U_i + Vq_b)
// The batch-wide products that remain (d allf += dU Wu, d v += dVq Wv, the weight gradients and the two column sums) stay one
// grouped launch + one split-K sum (ops.ReadoutHeadFused.backward).  The row phases run over ALL rows of a pass at once (no
// loop over sessions: every dependent trip to memory costs 1 - 2 us inside a step).
struct HeadBwdArgs {
    srec_head_bwd_desc d;
};

template <int DD>
__global__ __launch_bounds__(64 * NW, 1) void head_bwd_kernel(HeadBwdArgs a) {
    constexpr int D = 128 * DD, KS = D / 16, NT = 64 * NW;
    constexpr int JB4 = 2 * D / (32 * NW);                   // 32-row blocks of Wsr^T [2 D, D] per wave
    constexpr int GP = D + 8;                                // row stride (floats) of the d g tile
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* g_hi = sm;                               // [HS][D] g_s of the pass's sessions
    unsigned short* g_lo = g_hi + HS * D;
    float* dg = reinterpret_cast<float*>(g_lo + HS * D);     // [HS][GP] d g = right half of d cat
    float* vql = dg + HS * GP;                               // [HS][D] Vq of the pass's sessions
    float* de = vql + HS * D;                                // [HR + MAXN] d alpha, then d e, of the pass's rows
    float* al = de + HR + MAXN;                              // [HR + MAXN] alpha
    int* segs = reinterpret_cast<int*>(al + HR + MAXN);      // [HS + 1]
    int* cnt = segs + HS + 1;                                // [2]
    int* rs = cnt + 2;                                       // [HR + MAXN] session (0 .. HS-1) of each row of the pass
    int* segl = rs + HR + MAXN;                              // [B + 1]

    const srec_head_bwd_desc& q = a.d;
    const int hd = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Bl = dyn_count(q.dynB, q.B);
    const float* X = q.X;
    const int ld_x = q.ld_x;
    float* dX = q.dX[hd];
    float* dU = q.dU[hd];
    const int w0 = (int)blockIdx.x * HWIN, w1 = w0 + HWIN;
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        for (int b = tid; b <= Bl; b += NT) {
            const int sb = q.seg[b];
            segl[b] = sb;
            c0 += (b < Bl && sb < w0) ? 1 : 0;
            c1 += (b < Bl && sb < w1) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); }
        if (lane == 0) { atomicAdd(&cnt[0], c0); atomicAdd(&cnt[1], c1); }
    }
    // capacity padding: sessions past the live count get zero rows of g_s, d cat, dVq, dwp (shared by all workgroups) ...
    for (int b = Bl + (int)blockIdx.x; b < q.B; b += (int)gridDim.x)
        for (int c = tid * 4; c < D; c += NT * 4) {
            *reinterpret_cast<float4*>(q.gs[hd] + (size_t)b * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(q.gcat[hd] + (size_t)b * 2 * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(q.gcat[hd] + (size_t)b * 2 * D + D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(q.dVq[hd] + (size_t)b * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(q.dwp[hd] + (size_t)b * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    __syncthreads();
    // ... and the rows behind the last live node zero rows of dX / dU: every workgroup its own window
    const int ntl = Bl > 0 ? segl[Bl] : 0;
    for (int r = max(w0, ntl) + (tid >> 6); r < min(w1, q.NT); r += NW) {
        const int c = lane * 4;
        if (c < D) {
            *reinterpret_cast<float4*>(dX + (size_t)r * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(dU + (size_t)r * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (Bl <= 0 || w0 >= ntl) return;
#ifdef SREC_HEADF_TIMING
    unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_c = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_headf_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int bfirst = cnt[0], bend = cnt[1];
    auto sig = [](float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); };
    f32x16 acc[2][JB4];

    for (int b0 = bfirst; b0 < bend; b0 += HS) {
        const int ns = min(HS, bend - b0);
        __syncthreads();
        const int r0 = segl[b0];
        if (tid <= HS) segs[tid] = segl[b0 + min(tid, ns)] - r0;
        const int nrows = segl[b0 + ns] - r0;
        // ---- g_s = (g_y - y <y, g_y>) / |s|: one wavefront per session row, a wave's HS / NW rows requested together
        {
            constexpr int SPW = HS / NW;
            const int c = lane * 4;
            float4 yv[SPW], gv[SPW];
            float iv[SPW];
#pragma unroll
            for (int k = 0; k < SPW; ++k) {
                const int sb = wave + NW * k;
                yv[k] = gv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                iv[k] = 0.f;
                if (sb < ns && c < D) {
                    yv[k] = *reinterpret_cast<const float4*>(q.y[hd] + (size_t)(b0 + sb) * D + c);
                    gv[k] = *reinterpret_cast<const float4*>(q.gy[hd] + (size_t)(b0 + sb) * q.ld_gy + c);
                }
                if (sb < ns) iv[k] = q.inv[hd][b0 + sb];
            }
#pragma unroll
            for (int k = 0; k < SPW; ++k) {
                const int sb = wave + NW * k;
                const float dot = wave_sum(yv[k].x * gv[k].x + yv[k].y * gv[k].y + yv[k].z * gv[k].z + yv[k].w * gv[k].w);
                const float4 o = make_float4(iv[k] * (gv[k].x - yv[k].x * dot), iv[k] * (gv[k].y - yv[k].y * dot),
                                             iv[k] * (gv[k].z - yv[k].z * dot), iv[k] * (gv[k].w - yv[k].w * dot));
                if (c < D) {
                    if (sb < ns) *reinterpret_cast<float4*>(q.gs[hd] + (size_t)(b0 + sb) * D + c) = o;
                    stage4(g_hi, g_lo, D, sb, c, o);
                }
            }
        }
        // Vq rows of the pass (the column loop below switches session without a trip to memory), the pass's soft-max weights
        for (int i = tid; i < HS * (D / 4); i += NT) {
            const int sb = i / (D / 4), c = (i % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sb < ns) v = *reinterpret_cast<const float4*>(q.Vq[hd] + (size_t)(b0 + sb) * D + c);
            *reinterpret_cast<float4*>(vql + sb * D + c) = v;
        }
        for (int i = tid; i < nrows; i += NT) al[i] = q.alpha[hd][r0 + i];
        __syncthreads();
        for (int i = tid; i < nrows; i += NT) {          // (segs is published by the barrier above)
            int sb = 0;
#pragma unroll
            for (int k = 1; k < HS; ++k) sb += (i >= segs[k]) ? 1 : 0;
            rs[i] = min(sb, ns - 1);
        }
        HFT(0);
        // ---- d cat^T = Wsr^T . g_s^T: lane = session, registers = this wave's columns of [d v | d g]
        hl_product<1, JB4>(acc, (const unsigned short*)q.WsrT_f[hd], KS, g_hi, g_lo, D, l31 & (HS - 1), wave, lane);
        if (l31 < ns) {
            float* gc = q.gcat[hd] + (size_t)(b0 + l31) * 2 * D;
#pragma unroll
            for (int j = 0; j < JB4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = (wave * JB4 + j) * 32 + 8 * g + 4 * half;        // column of d cat
                    const float4 v = make_float4(acc[0][j][4 * g], acc[0][j][4 * g + 1], acc[0][j][4 * g + 2], acc[0][j][4 * g + 3]);
                    *reinterpret_cast<float4*>(gc + col) = v;
                    if (col >= D) *reinterpret_cast<float4*>(dg + l31 * GP + (col - D)) = v;
                }
        }
        __syncthreads();
        HFT(1);
        // ---- d alpha_i = <d g_b, x_i>, dX_i = alpha_i d g_b: one wavefront per row, 8 rows in flight per wave
        {
            const int c = lane * 4;
            for (int i0 = wave; i0 < nrows; i0 += 8 * NW) {
                float4 xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = min(i0 + k * NW, nrows - 1);
                    xv[k] = c < D ? *reinterpret_cast<const float4*>(X + (size_t)(r0 + i) * ld_x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k * NW;
                    const int sb = rs[min(i, nrows - 1)];
                    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < D) gv = *reinterpret_cast<const float4*>(dg + sb * GP + c);
                    const float d_ = wave_sum(gv.x * xv[k].x + gv.y * xv[k].y + gv.z * xv[k].z + gv.w * xv[k].w);
                    if (i < nrows) {
                        const float a_ = al[i];
                        if (lane == 0) de[i] = d_;
                        if (c < D) *reinterpret_cast<float4*>(dX + (size_t)(r0 + i) * D + c) = make_float4(a_ * gv.x, a_ * gv.y, a_ * gv.z, a_ * gv.w);
                    }
                }
            }
        }
        __syncthreads();
        HFT(2);
        // d e_i = alpha_i (d alpha_i - sum_j alpha_j d alpha_j): one wavefront per session
        for (int sb = wave; sb < ns; sb += NW) {
            const int base = segs[sb], n = min(segs[sb + 1] - base, MAXN);
            float ssum = 0.f;
            for (int i = lane; i < n; i += 64) ssum += al[base + i] * de[base + i];
            ssum = wave_sum(ssum);
            for (int i = lane; i < n; i += 64) de[base + i] = al[base + i] * (de[base + i] - ssum);
        }
        __syncthreads();
        HFT(3);
        // ---- dU, dVq, dwp: a wavefront takes whole sessions (sb = wave, wave + NW, ...: their sums are independent), a lane FOUR
        //      hidden columns - one 16-byte load of U and one 16-byte store of dU per row (a column per thread was a 4-byte load
        //      and a 4-byte store per row and wave: the phase was bound by the issue of ~200 vector-memory instructions per wave).
        //      Rows in node order per session, as srec_seg_attn_bwd sums them; 8 rows in flight.
        {
            const int k4 = lane * 4;
            if (k4 < D) {
                const float4 wk = *reinterpret_cast<const float4*>(q.we[hd] + k4);
                for (int sb = wave; sb < ns; sb += NW) {
                    const int base = segs[sb], n = segs[sb + 1] - base;
                    const float4 vq = *reinterpret_cast<const float4*>(vql + sb * D + k4);
                    const float* up = q.U[hd] + (size_t)(r0 + base) * D + k4;
                    float* dup = dU + (size_t)(r0 + base) * D + k4;
                    float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), dw = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int i = 0; i < n; i += 8) {
                        float4 uv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) uv[j] = *reinterpret_cast<const float4*>(up + (size_t)min(i + j, n - 1) * D);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (i + j < n) {
                                const float dei = de[base + i + j];
                                const float4 sg = make_float4(sig(uv[j].x + vq.x), sig(uv[j].y + vq.y), sig(uv[j].z + vq.z), sig(uv[j].w + vq.w));
                                dw.x += dei * sg.x; dw.y += dei * sg.y; dw.z += dei * sg.z; dw.w += dei * sg.w;
                                const float4 dp = make_float4(dei * wk.x * sg.x * (1.f - sg.x), dei * wk.y * sg.y * (1.f - sg.y),
                                                              dei * wk.z * sg.z * (1.f - sg.z), dei * wk.w * sg.w * (1.f - sg.w));
                                *reinterpret_cast<float4*>(dup + (size_t)(i + j) * D) = dp;
                                dv.x += dp.x; dv.y += dp.y; dv.z += dp.z; dv.w += dp.w;
                            }
                        }
                    }
                    *reinterpret_cast<float4*>(q.dVq[hd] + (size_t)(b0 + sb) * D + k4) = dv;
                    *reinterpret_cast<float4*>(q.dwp[hd] + (size_t)(b0 + sb) * D + k4) = dw;
                }
            }
        }
        HFT(4);
    }
#ifdef SREC_HEADF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_headf_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 8; ++i) g_headf_tim[blockIdx.x][i] = tim_t[i];
    }
#endif
}

}  // namespace

#ifdef SREC_HEADF_TIMING
extern "C" int srec_headf_timing(unsigned long long* tim, unsigned long long* blk) {
    if (hipMemcpyFromSymbol(tim, HIP_SYMBOL(g_headf_tim), sizeof(unsigned long long) * 8192) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(blk, HIP_SYMBOL(g_headf_blk), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : 1;
}
#endif

// (srec_head_wfrag - the hi / lo fragment-major operand copies of the weights - is a role of the step's prologue launch: prep.hip)

// desc: HOST srec_head_desc (srec_hg.h)
extern "C" int srec_head_fwd(const void* desc, void* stream) {
    const srec_head_desc* q = (const srec_head_desc*)desc;
    if (q == nullptr || q->nh <= 0 || q->nh > SREC_HEAD_MAXH || (q->d != 128 && q->d != 256) || q->B <= 0 || q->B > 8192 || q->NT <= 0) return SREC_BAD_ARG;
    if (q->X == nullptr || q->seg == nullptr || (q->ld_x & 3)) return SREC_BAD_ARG;
    for (int h = 0; h < q->nh; ++h) {
        if (q->cat[h] == nullptr || q->Wu_f[h] == nullptr || q->Wv_f[h] == nullptr || q->Wsr_f[h] == nullptr ||
            q->we[h] == nullptr || q->alpha[h] == nullptr || q->y[h] == nullptr || q->inv[h] == nullptr)
            return SREC_BAD_ARG;
        if (q->y16[h] != nullptr && (q->ld16 & 3)) return SREC_BAD_ARG;
    }
    HeadArgs a{};
    a.d = *q;
    const int D = q->d;
    const size_t lds = (size_t)(2 * HR * D + 2 * HS * 2 * D) * 2 + (size_t)(HS * (D + VQ_PAD) + 2 * D + HR + MAXN + NW * HR) * 4 +
                       (size_t)(HS + 1 + HR + 2 + q->B + 1) * 4;
    const dim3 grid((q->NT + HWIN - 1) / HWIN, q->nh);
    static std::atomic<unsigned long long> om[2];
    if (D == 256) {
        if (int rc = srec_lds_optin((const void*)head_fwd_kernel<2>, (int)lds, om[0])) return rc;
        hipLaunchKernelGGL(head_fwd_kernel<2>, grid, dim3(64 * NW), lds, (hipStream_t)stream, a);
    } else {
        if (int rc = srec_lds_optin((const void*)head_fwd_kernel<1>, (int)lds, om[1])) return rc;
        hipLaunchKernelGGL(head_fwd_kernel<1>, grid, dim3(64 * NW), lds, (hipStream_t)stream, a);
    }
    SREC_LAUNCH_CHECK();
    return 0;
}

// desc: HOST srec_head_bwd_desc (srec_hg.h)
extern "C" int srec_head_bwd(const void* desc, void* stream) {
    const srec_head_bwd_desc* q = (const srec_head_bwd_desc*)desc;
    if (q == nullptr || q->nh <= 0 || q->nh > SREC_HEAD_MAXH || (q->d != 128 && q->d != 256) || q->B <= 0 || q->B > 8192 || q->NT <= 0)
        return SREC_BAD_ARG;
    if (q->X == nullptr || q->seg == nullptr || (q->ld_x & 3) || (q->ld_gy & 3)) return SREC_BAD_ARG;
    for (int h = 0; h < q->nh; ++h)
        if (q->gy[h] == nullptr || q->y[h] == nullptr || q->inv[h] == nullptr || q->WsrT_f[h] == nullptr || q->alpha[h] == nullptr ||
            q->U[h] == nullptr || q->Vq[h] == nullptr || q->we[h] == nullptr || q->gs[h] == nullptr || q->gcat[h] == nullptr ||
            q->dX[h] == nullptr || q->dU[h] == nullptr || q->dVq[h] == nullptr || q->dwp[h] == nullptr)
            return SREC_BAD_ARG;
    HeadBwdArgs a{};
    a.d = *q;
    const int D = q->d;
    const size_t lds = (size_t)(2 * HS * D) * 2 + (size_t)(HS * (D + 8) + HS * D + 3 * (HR + MAXN)) * 4 + (size_t)(HS + 1 + 2 + q->B + 1) * 4;
    const dim3 grid((q->NT + HWIN - 1) / HWIN, q->nh);
    static std::atomic<unsigned long long> omb[2];        // (B-dependent dynamic LDS: above 64 KiB it needs the per-device opt-in)
    if (D == 256) {
        if (int rc = srec_lds_optin((const void*)head_bwd_kernel<2>, (int)lds, omb[0])) return rc;
        hipLaunchKernelGGL(head_bwd_kernel<2>, grid, dim3(64 * NW), lds, (hipStream_t)stream, a);
    } else {
        if (int rc = srec_lds_optin((const void*)head_bwd_kernel<1>, (int)lds, omb[1])) return rc;
        hipLaunchKernelGGL(head_bwd_kernel<1>, grid, dim3(64 * NW), lds, (hipStream_t)stream, a);
    }
    SREC_LAUNCH_CHECK();
    return 0;
}
