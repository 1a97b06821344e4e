// bf16-in-HBM grouped GEMMs for the GAT projections of the MSHGNN layer (gatconv.py:166-175,282-283 and their
// backward) - the FLOP-heavy part of the MSGIFSR encoder (~48 GFLOP per step at the C3 shapes).
//
// gemm_group_bf16.hip rounds fp32 operands to bf16 WHILE staging them through registers (global -> VGPR -> cvt ->
// ds_write), one barrier per 32-deep k-step: it measured 3-5x above its traffic floor.  Here every operand already
// lives in HBM as bf16 (activations are written as bf16 by their producers; the small fc weights get a bf16 copy and a
// transposed bf16 copy once per step, srec_weights_bf16), so staging is pure LDS-DMA:
//   * global_load_lds_dwordx4 (1 KiB per wave instruction) into a lane-linear LDS image, no staging registers, no
//     ds_write pass; bank conflicts of the ds_read_b128 fragment reads are removed by XOR-swizzling the SOURCE piece of
//     every 16-B slot (piece p of tile row r sits in slot p ^ ((r >> 1) & 7): with 128-B rows the 16 lanes of each
//     ds_read_b128 lane group then hit 16 distinct 16-B slots);
//   * 64-deep k-steps (4 MFMA k-steps per barrier), double buffered, ONE barrier per k-step: wait own DMA, barrier,
//     issue the next stage, compute;
//   * 4 waves (2 x 2) of v_mfma_f32_32x32x16_bf16, 128 x 128 (or 64 x 128) output tiles, fp32 accumulation.
// Two kernels:
//   nt16  C[M, N] (+)= sum_s A_s[M, K] B_s[N, K]^T      both operands k-contiguous.  Forward projections (B = W,
//         bf16 output: the accumulator is kept TRANSPOSED - lane = output row, registers = 4 consecutive columns - so
//         the 63 MB of projections leave as 8-byte packed stores) and backward-data (B = W^T copy, fp32 output, the
//         sum over the modules that project a node type is the segment loop: no split-K, no beta chains).
//   tn16  C[N1, N2] = sum over rows m of A[m, N1] B[m, N2]   (weight gradient dW = dP^T x; the reduction runs over the
//         node rows, clamped by the live count).  Both operands are read ROW-major as they are - the 63 MB dP is never
//         transposed in memory; the MFMA fragments (8 consecutive reduction rows of one column) are gathered from the
//         row-major LDS tile by eight 16-bit LDS reads (d16 / d16_hi pairs fill the four fragment registers without
//         a packing pass): consecutive lanes read consecutive 2-byte columns, conflict free.
#include "common.h"
#include "../../include/srec_hg.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short2_t __attribute__((ext_vector_type(2)));
constexpr int G16_MAXP = 16, G16_MAXS = 4;

struct G16Args {
    const unsigned short* A[G16_MAXP][G16_MAXS];
    const unsigned short* B[G16_MAXP][G16_MAXS];
    void* C[G16_MAXP];
    const int* dyn[G16_MAXP];
    int M[G16_MAXP], N[G16_MAXP], K[G16_MAXP], nseg[G16_MAXP], koff[G16_MAXP], nsplit[G16_MAXP], start[G16_MAXP + 1];
    int np, lda, ldb, ldc;
    int ldap[G16_MAXP], ldbp[G16_MAXP], ldcp[G16_MAXP];   // per-problem leading dimensions (the group's lda / ldb / ldc unless overridden)
    float beta;
    int keep_dead;                // leave output rows past the live count unwritten (nobody reads them)
    int per_xcd;                  // > 0: XCD-aware tile order (see xcd_tile), tiles per XCD
    int xcd_gs;                   // > 0: sibling groups of xcd_gs tiles dealt to the XCDs round-robin instead of runs
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// one 1-KiB LDS-DMA: lane l copies 16 B from sbase + voff[l] to LDS address lds_dst + 16 l (wave-uniform destination
// in M0).  Inline asm: hipcc would otherwise drain every outstanding DMA (vmcnt(0)) at the next LDS read it cannot prove
// disjoint; the only wait needed is the explicit one in front of the barrier that publishes the stage.
__device__ __forceinline__ void glds16(const unsigned short* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst_s = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);   // wave-uniform by construction: pin it to an SGPR
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst_s) : "memory");
}

// Workgroups are dealt to the 8 XCDs round-robin by index, and every XCD has its own L2.  With tile = blockIdx the tiles that
// share an operand slab (the two 128-column tiles of a row block in backward-data, the two column tiles of a dP slab in the
// weight gradient) sit on different XCDs and each pulls its own copy from HBM (PMC: 177 MB of reads for the 63 MB dP).  XCD x
// takes the x-th CONSECUTIVE run of tiles instead: siblings run side by side behind one L2.
// Capacity-padded problems end in tiles of dead rows; consecutive runs then hand some XCDs mostly dead tiles (backward-data at
// 45 % padding: 61 -> 85 us, profiles/r03e_caps).  With a host hint that the padding is large the sibling GROUPS (xcd_gs tiles that
// share an operand slab) are dealt round-robin instead: siblings still sit behind one L2, dead tails spread over all XCDs.
__device__ __forceinline__ int xcd_tile(const G16Args& g) {
    const int b = (int)blockIdx.x;
    if (g.xcd_gs > 0) {
        const int x = b & 7, l = b >> 3;
        return ((l / g.xcd_gs) * 8 + x) * g.xcd_gs + l % g.xcd_gs;
    }
    return g.per_xcd > 0 ? (b & 7) * g.per_xcd + (b >> 3) : b;
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// wait until at most `ahead` later stages (IPS LDS-DMA instructions each, per wave) are still in flight
template <int IPS, int PDV>
__device__ __forceinline__ void wait_stage(int ahead) {
    if (PDV >= 5 && ahead >= 4) wait_vm<(4 * IPS > 63 ? 63 : 4 * IPS)>();
    else if (PDV >= 4 && ahead >= 3) wait_vm<(3 * IPS > 63 ? 63 : 3 * IPS)>();
    else if (PDV >= 3 && ahead >= 2) wait_vm<2 * IPS>();
    else if (PDV >= 2 && ahead >= 1) wait_vm<IPS>();
    else wait_vm<0>();
}

#ifdef SREC_G16_TIMING   // development probe (tools/g16_timing.py): phase clocks of wave 0 + wall-clock life of every workgroup
__device__ unsigned long long g_g16_tim[8];
__device__ unsigned long long g_g16_blk[8192][2];
#define G16T(i) do { __builtin_amdgcn_sched_barrier(0); g16t[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G16T(i)
#endif

constexpr int NS = 4, PD = 3;        // default LDS ring: 4 stages, 3 stages of LDS-DMA in flight ahead of the MFMAs

// -------------------------------------------------------------------------------------------------- nt16
// C16: the accumulator is D^T (rows of D <-> B rows = output columns, lane <-> A row = output row): the bf16 output
// then leaves as 4 consecutive columns (8 B) per lane and register quad.
// k-steps of 32 (64-B tile rows, 4 pieces; one DMA instruction = 16 rows): the reductions here are short (K = D = 256
// for the forward) or streamed from HBM (K = H D for backward-data), so the kernel is bound by load latency, not by
// barriers - a 4-stage ring keeps 3 stages in flight per workgroup, and 16 KB stages leave room for 2-3 workgroups
// per CU.  Piece p of tile row r sits in slot p ^ ((r >> 2) & 3): the 16 lanes of every ds_read_b128 group hit 16
// distinct 16-B slots.
// NW: waves per workgroup, 2 x (NW / 2)
template <int TM, int TN, bool C16, int BK = 32, int NS = 4, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm16_nt_kernel(G16Args g) {
    constexpr int WN = NW / 2, NT = 64 * NW;          // waves along N, threads
    constexpr int PD = NS - 1;
    constexpr int RPI = 512 / BK;                        // tile rows per DMA instruction (1 KiB)
    constexpr int PPR = BK / 8;                          // 16-B pieces per tile row
    constexpr int FS = BK == 32 ? 2 : 1;                 // swizzle: slot = piece ^ ((row >> FS) & (PPR - 1))
    constexpr int IM = TM / 64, IN = TN / (32 * WN);     // 32x32 accumulators per wave and dimension (2 x WN waves)
    constexpr int STG = (TM + TN) * BK;                  // bf16 elements per stage
    constexpr int NIA = TM / RPI, NI = (TM + TN) / RPI;  // DMA instructions per stage: A, total
    constexpr int IPS = NI / NW;                         // ... per wave
    static_assert(NI % NW == 0, "every wave must issue the same number of DMA instructions per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

    const int bid = xcd_tile(g);
    if (bid >= g.start[g.np]) return;
#ifdef SREC_G16_TIMING
    unsigned long long g16t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g16w = 0, g16d = 0, g16c = 0;
    if (threadIdx.x == 0 && bid < 8192) g_g16_blk[bid][0] = __builtin_amdgcn_s_memrealtime();
    G16T(0);
#endif
    int p = 0;
#pragma unroll
    for (int i = 1; i < G16_MAXP; ++i)
        if (i < g.np && bid >= g.start[i]) p = i;
    const int M = g.M[p], N = g.N[p], K = g.K[p];
    const int tn = (N + TN - 1) / TN, tile = bid - g.start[p];
    const int m0 = (tile / tn) * TM, n0 = (tile % tn) * TN;
    const int Ml = dyn_count(g.dyn[p], M);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    float* __restrict__ C = static_cast<float*>(g.C[p]);
    unsigned short* __restrict__ C16p = static_cast<unsigned short*>(g.C[p]);
    if (m0 >= Ml) {                                      // tile of capacity padding: zero rows when overwriting
        if (!g.keep_dead && (C16 || g.beta == 0.f))
            for (int i = tid; i < TM * TN / 4; i += NT) {
                const int r = m0 + (i * 4) / TN, c = n0 + (i * 4) % TN;
                if (r < M && c + 3 < N) {
                    if (C16) *reinterpret_cast<uint2*>(C16p + (size_t)r * g.ldcp[p] + c) = make_uint2(0u, 0u);
                    else *reinterpret_cast<float4*>(C + (size_t)r * g.ldcp[p] + c) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (r < M) {
                    for (int e = 0; e < 4; ++e)
                        if (c + e < N) { if (C16) C16p[(size_t)r * g.ldcp[p] + c + e] = 0; else C[(size_t)r * g.ldcp[p] + c + e] = 0.f; }
                }
            }
        return;
    }

    f32x16 acc[IM][IN];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < IN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK, total = nk * g.nseg[p];
    const unsigned lds0 = lds_addr(smem);
    // DMA instruction i covers tile rows 16 i .. 16 i + 15 of the concatenated (A rows, then B rows) stage; lane -> (row
    // l >> 2, slot l & 3).  Rows past an operand are clamped to its last row (their products land in rows / columns the
    // epilogue does not store).
    const int rl = lane / PPR, sl = lane % PPR;
    // Issue state of the ring (wave-uniform): the operand bases of the current K segment live in SGPRs and advance by BK per
    // stage; the per-lane source offsets do not depend on k.  (Recomputing segment = it / nk and re-reading g.A[p][s] from
    // the kernel arguments for every stage put an integer division and a scalar-memory round trip in front of every
    // group of LDS-DMA instructions: 720 cycles per stage next to 256 cycles of MFMA, tools/g16_timing.py.)
    unsigned voff[IPS];
#pragma unroll
    for (int ii = 0; ii < IPS; ++ii) {
        const int i = ii * NW + wave;                    // wave-uniform
        const int r = RPI * (i < NIA ? i : i - NIA) + rl;
        const unsigned pc = (unsigned)((sl ^ ((r >> FS) & (PPR - 1))) * 8);
        voff[ii] = i < NIA ? ((unsigned)min(r, Ml - 1 - m0) * (unsigned)g.ldap[p] + pc) * 2u
                           : ((unsigned)min(r, N - 1 - n0) * (unsigned)g.ldbp[p] + pc) * 2u;
    }
    int is_seg = 0, is_k = 0;
    const unsigned short* Aseg = g.A[p][0] + (size_t)m0 * g.ldap[p];
    const unsigned short* Bseg = g.B[p][0] + (size_t)n0 * g.ldbp[p];
    auto stage = [&](int it) {
        const unsigned dst = lds0 + (unsigned)((it % NS) * STG) * 2u;
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
            const int i = ii * NW + wave;
            glds16((i < NIA ? Aseg : Bseg) + is_k, voff[ii], dst + (unsigned)i * 1024u);
        }
        is_k += BK;
        if (is_k >= K) {                                 // next K segment (another module's projection / weight)
            is_k = 0;
            if (++is_seg < g.nseg[p]) {
                Aseg = g.A[p][is_seg] + (size_t)m0 * g.ldap[p];
                Bseg = g.B[p][is_seg] + (size_t)n0 * g.ldbp[p];
            }
        }
    };

    for (int s = 0; s < PD && s < total; ++s) stage(s);
    G16T(1);
    for (int it = 0; it < total; ++it) {
        G16T(2);
        wait_stage<IPS, PD>(min(total - it - 1, PD - 1));      // this wave's pieces of stage `it` have landed ...
        __syncthreads();                                       // ... everyone's have; stage it - 1's buffer is free again
        G16T(3);
        if (it + PD < total) stage(it + PD);
        G16T(4);
        const unsigned short* As = smem + (it % NS) * STG;
        const unsigned short* Bs = As + TM * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[IM], b[IN];
#pragma unroll
            for (int i = 0; i < IM; ++i) {
                const int r = wm * (TM / 2) + i * 32 + l31;
                a[i] = *reinterpret_cast<const bf16x8*>(As + r * BK + (((2 * ks + half) ^ ((r >> FS) & (PPR - 1))) << 3));
            }
#pragma unroll
            for (int j = 0; j < IN; ++j) {
                const int r = wn * (TN / WN) + j * 32 + l31;
                b[j] = *reinterpret_cast<const bf16x8*>(Bs + r * BK + (((2 * ks + half) ^ ((r >> FS) & (PPR - 1))) << 3));
            }
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < IN; ++j)
                    acc[i][j] = C16 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#ifdef SREC_G16_TIMING
        G16T(5);
        g16w += g16t[3] - g16t[2]; g16d += g16t[4] - g16t[3]; g16c += g16t[5] - g16t[4];
#endif
    }
    G16T(6);

    if (C16) {
        // acc[i][j] = D^T: lane <-> output row, register r <-> output column (r & 3) + 8 (r >> 2) + 4 half.  The 4 KB row
        // stride of P makes per-lane stores 16-B fragments of 32 different rows per instruction (the forward measured
        // bound by write transactions, not bytes): every wave transposes its tile through its own LDS patch (rows padded
        // to 144 B: conflict-free 8-B writes) and stores 16 B per lane = 8 full 128-B row segments per instruction.
        constexpr int WR = TM / 2, WC = TN / WN, LDP = WC + 8;        // wave tile, padded LDS row (bf16 elements)
        __syncthreads();                                               // every wave is done with the ring buffers
        unsigned short* patch = smem + wave * (WR * LDP);
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < IN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 v;
                    v.x = srec_pack_bf16(acc[i][j][4 * q], acc[i][j][4 * q + 1]);
                    v.y = srec_pack_bf16(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    *reinterpret_cast<uint2*>(patch + (i * 32 + l31) * LDP + j * 32 + 8 * q + 4 * half) = v;
                }
        constexpr int PPW = WC / 8;                                    // 16-B pieces per patch row
#pragma unroll
        for (int t = 0; t < WR * PPW / 64; ++t) {
            const int idx = t * 64 + lane, rr = idx / PPW, pc = idx % PPW;
            const int row = m0 + wm * WR + rr, col = n0 + wn * WC + pc * 8;
            if (row >= M || (g.keep_dead && row >= Ml)) continue;
            uint4 v = *reinterpret_cast<const uint4*>(patch + rr * LDP + pc * 8);
            if (row >= Ml) v = make_uint4(0u, 0u, 0u, 0u);
            if (col + 7 < N) {
                *reinterpret_cast<uint4*>(C16p + (size_t)row * g.ldcp[p] + col) = v;
            } else {
                const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
                for (int k = 0; k < 8; ++k)
                    if (col + k < N) C16p[(size_t)row * g.ldcp[p] + col + k] = e[k];
            }
        }
#ifdef SREC_G16_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G16T(7);
        if (threadIdx.x == 0 && bid < 8192) g_g16_blk[bid][1] = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && bid == 17) { g_g16_tim[0] = g16t[1] - g16t[0]; g_g16_tim[1] = g16w; g_g16_tim[2] = g16d; g_g16_tim[3] = g16c; g_g16_tim[4] = g16t[7] - g16t[6]; g_g16_tim[5] = g16t[7] - g16t[0]; g_g16_tim[6] = total; }
#endif
        return;
    }
    // fp32 output (backward-data, beta = 1: d x += ...): the MFMA layout has ONE float of a row per lane - 16 four-byte loads and
    // 16 stores per MFMA tile and wave, each a vector-memory instruction of ~40 CU cycles whatever its width.  Through a per-wave
    // LDS patch (the ring is free now) the tile is read / written as 16-byte accesses: 4 + 4 instructions per MFMA tile.
    if ((N & 3) == 0 && (g.ldcp[p] & 3) == 0 && ((uintptr_t)C & 15) == 0) {
        constexpr int PS = 36;
        __syncthreads();                                               // every wave is done with the ring buffers
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PS);
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < IN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PS + l31] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int idx = q4 * 64 + lane, rl = idx >> 3, c4 = (idx & 7) * 4;
                    const int row = m0 + wm * (TM / 2) + i * 32 + rl, col = n0 + wn * (TN / WN) + j * 32 + c4;
                    if (row < M && col < N) {
                        float4* q = reinterpret_cast<float4*>(C + (size_t)row * g.ldcp[p] + col);
                        if (row < Ml) {
                            float4 v = *reinterpret_cast<const float4*>(patch + rl * PS + c4);
                            if (g.beta != 0.f) {
                                const float4 o = *q;
                                v.x += g.beta * o.x; v.y += g.beta * o.y; v.z += g.beta * o.z; v.w += g.beta * o.w;
                            }
                            *q = v;
                        } else if (g.beta == 0.f) {
                            *q = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < IN; ++j) {
            const int col = n0 + wn * (TN / WN) + j * 32 + l31;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) {
                    float* q = C + (size_t)row * g.ldcp[p] + col;
                    if (row < Ml) *q = g.beta != 0.f ? acc[i][j][r] + g.beta * *q : acc[i][j][r];
                    else if (g.beta == 0.f) *q = 0.f;
                }
            }
        }
#ifdef SREC_G16_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G16T(7);
    if (threadIdx.x == 0 && bid < 8192) g_g16_blk[bid][1] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && bid == 17) { g_g16_tim[0] = g16t[1] - g16t[0]; g_g16_tim[1] = g16w; g_g16_tim[2] = g16d; g_g16_tim[3] = g16c; g_g16_tim[4] = g16t[7] - g16t[6]; g_g16_tim[5] = g16t[7] - g16t[0]; g_g16_tim[6] = total; }
#endif
}

// -------------------------------------------------------------------------------------------------- tn16
// C[N1, N2] = sum_{m < live} A[m, N1] B[m, N2]; tile 128 x 128, reduction staged 32 rows at a time through the same
// 4-stage ring.  Both operand tiles sit in LDS ROW-major ([32 reduction rows][128 columns], 256-B rows, filled by
// LDS-DMA); an MFMA operand fragment (8 consecutive reduction rows of ONE column per lane) is two ds_read_b64_tr_b16:
// within a 16-lane group lane i supplies the address of (row i >> 2, 4 contiguous columns 4 (i & 3) ..) and receives
// (rows 0..3, column i) - measured lane map, tools/probes/tr_probe.hip.  The four rows of a group are 256 B apart, i.e.
// on the same banks: piece p of row r is stored in slot p ^ (4 (r & 3)), which moves them to the four 64-B quarters.
typedef short short4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const unsigned short* base, unsigned byte_off) {
    // compiler builtin (not inline asm): hipcc tracks the read in lgkmcnt and schedules it against the MFMAs itself
    const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4_t*)(reinterpret_cast<const char*>(base) + byte_off));
    return __builtin_bit_cast(uint2, v);
}

template <int BR, int NS>
__global__ __launch_bounds__(256) void gemm16_tn_kernel(G16Args g) {
    constexpr int T = 128;                               // output tile; BR = reduction rows per stage
    constexpr int PD = NS - 1;
    constexpr int STG = 2 * BR * T;                      // bf16 elements per stage (A tile + B tile)
    constexpr int NIO = BR / 4;                          // DMA instructions (4 rows each) per operand and stage
    constexpr int IPS = 2 * NIO / 4;                     // ... per stage and wave
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int bid = xcd_tile(g);
    if (bid >= g.start[g.np]) return;
    int p = 0;
#pragma unroll
    for (int i = 1; i < G16_MAXP; ++i)
        if (i < g.np && bid >= g.start[i]) p = i;
    const int N1 = g.M[p], N2 = g.N[p];
    const int tn = (N2 + T - 1) / T, ntile = ((N1 + T - 1) / T) * tn;
    const int split = (bid - g.start[p]) / ntile, tile = (bid - g.start[p]) % ntile;
    const int i0 = (tile / tn) * T, j0 = (tile % tn) * T;
    // row split s of nsplit: reduction rows [s chunk, (s + 1) chunk) of the problem, written to slab s of C
    // (the LIVE rows are split: with capacity padding an even split of K would leave the last slabs' workgroups idle)
    const int Klive = g.dyn[p] == nullptr ? g.K[p] : max(0, min(g.K[p], *g.dyn[p] - g.koff[p]));
    const int chunk = ((Klive + g.nsplit[p] - 1) / g.nsplit[p] + 63) / 64 * 64;
    const int kbeg = split * chunk, klen = max(0, min(chunk, g.K[p] - kbeg));
    // live reduction rows: rows koff .. koff + K of an operand with *dyn live rows in total
    const int Kr = g.dyn[p] == nullptr ? klen : max(0, min(klen, *g.dyn[p] - g.koff[p] - kbeg));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    float* __restrict__ C = static_cast<float*>(g.C[p]) + (size_t)split * N1 * g.ldcp[p];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nst = (Kr + BR - 1) / BR;
    const int total = nst * g.nseg[p];
    const unsigned lds0 = lds_addr(smem);
    // DMA instruction i: operand i / NIO, reduction rows 4 (i % NIO) .. + 3; lane -> (row l >> 4, slot l & 15)
    const int rl = lane >> 4, sl = lane & 15;
    auto stage = [&](int it) {
        const int s = it / nst, r0 = (it - s * nst) * BR;
        const unsigned dst = lds0 + (unsigned)((it % NS) * STG) * 2u;
        const int pc = sl ^ (4 * rl);                    // global piece of this lane's slot (row & 3 == rl)
#pragma unroll
        for (int ii = 0; ii < IPS; ++ii) {
            const int i = ii * 4 + wave;
            const int gr = kbeg + min(r0 + 4 * (i % NIO) + rl, Kr - 1);
            if (i < NIO) glds16(g.A[p][s], ((unsigned)gr * (unsigned)g.ldap[p] + (unsigned)min(i0 + pc * 8, N1 - 8)) * 2u, dst + (unsigned)i * 1024u);
            else glds16(g.B[p][s], ((unsigned)gr * (unsigned)g.ldbp[p] + (unsigned)min(j0 + pc * 8, N2 - 8)) * 2u, dst + (unsigned)i * 1024u);
        }
    };
    // per-lane LDS byte offsets of the transposing reads: fragment f (32 columns) of an operand tile, reduction rows
    // kb + 4 e + (i >> 2) with kb = 16 ks + 8 half, columns cbase + 16 ((lane >> 4) & 1) + 4 (i & 3), i = lane & 15
    const int ti = lane & 15, tr = ti >> 2;
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (ti & 3);
    auto tr_off = [&](int col, int row) {                // byte offset of (row, col..col+3) inside an operand tile
        return (unsigned)(row * 256 + ((((col >> 3) ^ (4 * (row & 3))) << 4) | ((col & 7) << 1)));
    };

    for (int s = 0; s < PD && s < total; ++s) stage(s);
    for (int it = 0; it < total; ++it) {
        wait_stage<IPS, PD>(min(total - it - 1, PD - 1));
        __syncthreads();
        if (it + PD < total) stage(it + PD);
        const int r0 = (it % nst) * BR;
        const unsigned short* abase = smem + (it % NS) * STG;
        const unsigned short* bbase = abase + BR * T;
        const bool tail = r0 + BR > Kr;                  // the last stage of a segment may hold clamped (repeated) rows
#pragma unroll
        for (int ks = 0; ks < BR / 16; ++ks) {
            const int kb = ks * 16 + 8 * half;
            uint2 ra[2][2], rb[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ra[f][e] = lds_tr16(abase, tr_off(wm * 64 + f * 32 + tcol, kb + 4 * e + tr));
                    rb[f][e] = lds_tr16(bbase, tr_off(wn * 64 + f * 32 + tcol, kb + 4 * e + tr));
                }
            if (tail) {                                  // rows at or past the live count contribute nothing: zero the A side
                const int nvalid = Kr - r0 - kb;         // this lane's element q of read e is row kb + 4 e + q
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int b0 = 4 * e;
                        if (b0 + 0 >= nvalid) ra[f][e].x &= 0xffff0000u;
                        if (b0 + 1 >= nvalid) ra[f][e].x &= 0x0000ffffu;
                        if (b0 + 2 >= nvalid) ra[f][e].y &= 0xffff0000u;
                        if (b0 + 3 >= nvalid) ra[f][e].y &= 0x0000ffffu;
                    }
            }
            bf16x8 a[2], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f] = __builtin_bit_cast(bf16x8, make_uint4(ra[f][0].x, ra[f][0].y, ra[f][1].x, ra[f][1].y));
                b[f] = __builtin_bit_cast(bf16x8, make_uint4(rb[f][0].x, rb[f][0].y, rb[f][1].x, rb[f][1].y));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if ((N2 & 3) == 0 && (g.ldcp[p] & 3) == 0 && ((uintptr_t)C & 15) == 0) {      // 16-byte accesses through a per-wave LDS patch
        constexpr int PS = 36;
        __syncthreads();                                               // every wave is done with the ring buffers
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PS);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PS + l31] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int idx = q4 * 64 + lane, rl = idx >> 3, c4 = (idx & 7) * 4;
                    const int row = i0 + wm * 64 + i * 32 + rl, col = j0 + wn * 64 + j * 32 + c4;
                    if (row < N1 && col < N2) {
                        float4* q = reinterpret_cast<float4*>(C + (size_t)row * g.ldcp[p] + col);
                        float4 v = *reinterpret_cast<const float4*>(patch + rl * PS + c4);
                        if (g.beta != 0.f) {
                            const float4 o = *q;
                            v.x += g.beta * o.x; v.y += g.beta * o.y; v.z += g.beta * o.z; v.w += g.beta * o.w;
                        }
                        *q = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j0 + wn * 64 + j * 32 + l31;
            if (col >= N2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < N1) {
                    float* q = C + (size_t)row * g.ldcp[p] + col;
                    *q = g.beta != 0.f ? acc[i][j][r] + g.beta * *q : acc[i][j][r];
                }
            }
        }
}

// -------------------------------------------------------------------------------------------------- operand copies
// fp32 rows -> bf16 rows (zero past the live count): the activations a GEMM reads that no producer wrote as bf16
__global__ void rows_bf16_kernel(const float* __restrict__ src, int ld, int n, const int* __restrict__ dyn, int d,
                                 unsigned short* __restrict__ dst) {
    const int nl = dyn_count(dyn, n);
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long)n * d) return;
    const int r = (int)(i / d), c = (int)(i % d);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nl) v = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
    uint2 o;
    o.x = srec_pack_bf16(v.x, v.y); o.y = srec_pack_bf16(v.z, v.w);
    *reinterpret_cast<uint2*>(dst + i) = o;
}

// out[c][i] = sum_r part[c][r][i] (fixed order: deterministic), i < n (n % 4 == 0); grid.y = c
__global__ void sum_slabs_kernel(const float* __restrict__ part, int R, long n, float* __restrict__ out) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float* pc = part + (size_t)blockIdx.y * R * n;
    float4 s = *reinterpret_cast<const float4*>(pc + i);
    for (int r = 1; r < R; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(pc + (size_t)r * n + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * n + i) = s;
}

int fill(G16Args& g, const void* desc_, int tm, int tn, bool tnmode, int& blocks) {
    const srec_gemm16_group* d = (const srec_gemm16_group*)desc_;
    if (d == nullptr || d->np <= 0 || d->np > G16_MAXP || (d->lda & 7) || (d->ldb & 7)) return SREC_BAD_ARG;
    g.np = d->np; g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.beta = d->beta;
    g.keep_dead = (d->c16 >> 1) & 1;
    blocks = 0;
    for (int p = 0; p < d->np; ++p) {
        if (d->nseg[p] <= 0 || d->nseg[p] > G16_MAXS || d->M[p] <= 0 || d->N[p] <= 0 || d->K[p] <= 0) return SREC_BAD_ARG;
        if (!tnmode && (d->K[p] & 31)) return SREC_BAD_ARG;                 // 32-deep k-steps
        if (tnmode && ((d->M[p] & 7) || (d->N[p] & 7) || d->M[p] < 8 || d->N[p] < 8)) return SREC_BAD_ARG;
        g.M[p] = d->M[p]; g.N[p] = d->N[p]; g.K[p] = d->K[p]; g.nseg[p] = d->nseg[p]; g.C[p] = d->C[p]; g.dyn[p] = d->dyn[p];
        g.ldap[p] = d->lda_p[p] > 0 ? d->lda_p[p] : d->lda; g.ldbp[p] = d->ldb_p[p] > 0 ? d->ldb_p[p] : d->ldb;
        g.ldcp[p] = d->ldc_p[p] > 0 ? d->ldc_p[p] : d->ldc;
        if ((g.ldap[p] & 7) || (g.ldbp[p] & 7)) return SREC_BAD_ARG;
        g.koff[p] = tnmode ? d->koff[p] : 0;
        g.nsplit[p] = tnmode && d->nsplit[p] > 1 ? d->nsplit[p] : 1;
        if (g.koff[p] < 0 || g.nsplit[p] > 64) return SREC_BAD_ARG;
        for (int s = 0; s < d->nseg[p]; ++s) {
            if (((uintptr_t)d->A[p][s] & 15) || ((uintptr_t)d->B[p][s] & 15)) return SREC_BAD_ARG;
            g.A[p][s] = (const unsigned short*)d->A[p][s]; g.B[p][s] = (const unsigned short*)d->B[p][s];
        }
        g.start[p] = blocks;
        blocks += cdiv(d->M[p], tm) * cdiv(d->N[p], tn) * g.nsplit[p];
    }
    g.start[d->np] = blocks;
    return 0;
}

template <typename K>
int optin(K kernel, int bytes, std::atomic<unsigned long long>& done) {
    return bytes > 64 * 1024 ? srec_lds_optin((const void*)kernel, bytes, done) : 0;
}

}  // namespace

// desc: host srec_gemm_group (srec_hg.h) with bf16 A and B.  C_p [M, N] (+)= sum_s A_ps [M, K] B_ps [N, K]^T; K % 32 == 0,
// lda / ldb % 8 == 0, 16-B aligned bases; c16: bf16 output (beta ignored); dyn clamps the output rows.
extern "C" int srec_gemm16_nt(const void* desc_, void* stream) {
    const srec_gemm16_group* h = (const srec_gemm16_group*)desc_;
    if (h == nullptr || h->np <= 0 || h->np > G16_MAXP) return SREC_BAD_ARG;
    G16Args g{};
    int blocks = 0;
    // small-N problems (backward-data, N = D): 64-row tiles keep every CU busy
    // (tile-shape heuristics count LIVE rows where the caller knows them: mhint = expected *dyn of a capacity-padded problem)
    auto Mh = [&](int p) { return (h->mhint[p] > 0 && h->mhint[p] < h->M[p]) ? h->mhint[p] : h->M[p]; };
    long t128 = 0, rows_cap = 0, rows_live = 0;
    for (int p = 0; p < h->np; ++p) { t128 += (long)cdiv(Mh(p), 128) * cdiv(h->N[p], 128); rows_cap += h->M[p]; rows_live += Mh(p); }
    // skinny outputs (backward-data: N = D) take 64-row tiles: twice the workgroups over the same HBM stream
    int nmax = 0;
    for (int p = 0; p < h->np; ++p) nmax = h->N[p] > nmax ? h->N[p] : nmax;
    int tm = (t128 >= 384 && nmax > 256) ? 128 : 64;
    // tiny products (the GRU hidden-state GEMMs: ~2k x 256 outputs): 64 x 64 tiles double the workgroups in flight
    long t64x128 = 0;
    for (int p = 0; p < h->np; ++p) t64x128 += (long)cdiv(Mh(p), 64) * cdiv(h->N[p], 128);
    // Workgroups are dealt to the 8 XCDs round-robin and a workgroup is latency bound on its own DMA ring (a lone one takes
    // as long as one of three on its CU), so a launch costs (rounds on the fullest XCD) x (workgroup life).  The fp32-output
    // kernels: 64 x 128 tiles with 2 x 64-deep stages = 48 KB -> 3 per CU = 96 slots per XCD; 128 x 128 = 64 KB -> 2 per CU =
    // 64 slots, workgroup life +5 % for twice the work (tools/gemm16_bench.py).  The step's backward-data launch (960 64-row
    // tiles = 120 per XCD) ran 1.25 -> 2 rounds: 77 us; as 480 128-row tiles (60 per XCD) it is one round.
    if (!(h->c16 & 1) && tm == 64 && t64x128 > 384) {
        const long r64 = cdiv((int)cdiv((int)t64x128, 8), 96), r128 = cdiv((int)cdiv((int)t128, 8), 64);
        // ... and when the rounds tie, the 128-row tiles win on BYTES: these launches run at what the L2s deliver to the CUs
        // (8 - 12 TB/s in every staging variant, profiles/r03_notes.md) and a 128 x 128 tile fetches 2/3 of the operand bytes of two
        // 64 x 128 tiles per output - as long as there is at least one workgroup per CU (65.4 -> 55.1 us at the bench shapes)
        if (r128 * 105 < r64 * 100 || (r128 <= r64 && t128 >= 256)) tm = 128;
    }
    // (measured and removed again, numbers in profiles/r03_notes.md: a forward kernel with the weights in registers, a
    //  backward-data kernel fed with fragment-major weights from L2, 128 x 256 8-wave tiles, column tiles dealt to XCDs)
    const int tn = (tm == 64 && t64x128 < 192) ? 64 : 128;
    if (int rc = fill(g, desc_, tm, tn, false, blocks)) return rc;
    // fp32-output launches (backward-data: two column tiles share every dP row block): 38.7 -> 34.5 us at the bench shapes; the
    // bf16-output forward (sixteen column tiles per row block, all of x fits any L2) measured 1.5 us slower that way
    if (!(h->c16 & 1)) {
        bool same_tn = true;
        const int tn0 = cdiv(h->N[0], tn);
        for (int p = 1; p < h->np; ++p) same_tn = same_tn && cdiv(h->N[p], tn) == tn0;
        if (same_tn && rows_live * 10 < rows_cap * 9) {          // > 10 % capacity padding: round-robin sibling groups
            g.xcd_gs = tn0;
            blocks = cdiv(blocks, 8 * tn0) * 8 * tn0;
        } else {
            g.per_xcd = cdiv(blocks, 8); blocks = 8 * g.per_xcd;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const bool c16 = h->c16 & 1;
    static std::atomic<unsigned long long> optin_mask[8];
#define SREC_G16(TMV, TNV, C16V, BKV, NSV, slot)                                                                       \
    do {                                                                                                               \
        const size_t lds = (size_t)NSV * (TMV + TNV) * BKV * 2;                                                        \
        if (int rc = optin(gemm16_nt_kernel<TMV, TNV, C16V, BKV, NSV>, (int)lds, optin_mask[slot])) return rc;         \
        hipLaunchKernelGGL((gemm16_nt_kernel<TMV, TNV, C16V, BKV, NSV>), dim3(blocks), dim3(256), lds, st, g);         \
    } while (0)
    // measured at the bench's GAT shapes (tools/gemm16_bench.py): every ring lands at 33-40 us for the 16-GFLOP forward
    // (deeper rings lose more in occupancy than they gain in flight): 3 x 32-deep stages for the bf16-output forward,
    // 2 x 64-deep stages for backward-data.  Few workgroups (<= 1.5 per CU): LDS is plentiful and every workgroup is latency
    // bound on its own DMA - a 4-stage ring keeps three 64-deep stages in flight instead of one
    const bool deep = blocks <= 384;
    if (tn == 64) {
        if (c16) SREC_G16(64, 64, true, 64, 2, 0); else if (deep) SREC_G16(64, 64, false, 64, 4, 1); else SREC_G16(64, 64, false, 64, 2, 2);
    } else if (deep && tm == 64 && !c16) {
        SREC_G16(64, 128, false, 64, 4, 3);
    } else if (tm == 128) {
        if (c16) SREC_G16(128, 128, true, 32, 3, 4); else SREC_G16(128, 128, false, 64, 2, 5);
    } else {
        if (c16) SREC_G16(64, 128, true, 32, 3, 6); else SREC_G16(64, 128, false, 64, 2, 7);
    }
#undef SREC_G16
    SREC_LAUNCH_CHECK();
    return 0;
}

#ifdef SREC_G16_TIMING
extern "C" int srec_g16_timing(unsigned long long* tim8, unsigned long long* blk) {
    if (hipMemcpyFromSymbol(tim8, HIP_SYMBOL(g_g16_tim), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(blk, HIP_SYMBOL(g_g16_blk), sizeof(unsigned long long) * 16384) == hipSuccess ? 0 : 1;
}
extern "C" int srec_g16_timing_reset() {
    static unsigned long long z[16384];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_g16_blk), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

// C_p [M, N] = sum_s sum_{m < min(K, *dyn)} A_ps [m, M] B_ps [m, N]  (fp32 output; M, N % 8 == 0)
extern "C" int srec_gemm16_tn(const void* desc_, void* stream) {
    G16Args g{};
    int blocks = 0;
    if (int rc = fill(g, desc_, 128, 128, true, blocks)) return rc;
    g.per_xcd = cdiv(blocks, 8); blocks = 8 * g.per_xcd;
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<unsigned long long> om;
    // 2 stages of 64 reduction rows (34 us at the bench shapes vs 37-38 for the 32-row rings)
    const size_t lds = (size_t)2 * 2 * 64 * 128 * 2;
    if (int rc = optin(gemm16_tn_kernel<64, 2>, (int)lds, om)) return rc;
    hipLaunchKernelGGL((gemm16_tn_kernel<64, 2>), dim3(blocks), dim3(256), lds, st, g);
    SREC_LAUNCH_CHECK();
    return 0;
}

// out [C, n] = sum over the R slabs of part [C, R, n] (split weight-gradient products of one module), n % 4 == 0
extern "C" int srec_sum_slabs(const float* part, int C, int R, long n, float* out, void* stream) {
    if (C <= 0 || R <= 0 || n <= 0) return 0;
    if (n & 3) return SREC_BAD_ARG;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n / 4 + 255) / 256), C), dim3(256), 0, (hipStream_t)stream, part, R, n,
                       out);
    SREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int srec_rows_bf16(const float* src, int ld, int n, const int* dyn, int d, void* dst16, void* stream) {
    if (n <= 0) return 0;
    if ((d & 3) || (ld & 3)) return SREC_BAD_ARG;
    const long items = (long)n * d / 4;
    hipLaunchKernelGGL(rows_bf16_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, n,
                       dyn, d, (unsigned short*)dst16);
    SREC_LAUNCH_CHECK();
    return 0;
}

// (srec_weights_bf16 - the bf16 / transposed copies of the fc weights - is a role of the step's prologue launch: prep.hip)
