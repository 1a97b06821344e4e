// bf16-operand variant of the fused full-catalog scoring + softmax-CE ("flash-CE", see score_ce.hip for the
// algorithm and the reference call sites it replaces).  BASELINE config C3's reduced-precision path.
//
// Design (gfx950): with v_mfma_f32_32x32x16_bf16 the matrix pipe is 16x faster than fp32, so the kernel is
// shaped around LDS traffic and occupancy instead:
//   * operands are pre-rounded ONCE per step into zero-padded bf16 copies (srec_bf16_prepare), row-major [rows, D] only,
//     D = d padded to 32/64/96/128/256: the same LDS image feeds S (ds_read_b128 along k = D) and ACC += P Y, whose B
//     fragments (8 consecutive STREAMED rows of one column) are gathered by ds_read_b64_tr_b16 transposing reads - no
//     transposed copy in HBM, half the chunk traffic of the first version of this kernel.
//   * every wave OWNS 32 rows (items for dE, sessions for d sr / forward) as ready-made MFMA fragments in
//     registers and keeps its 32 x D accumulator in registers; a workgroup is 4 independent waves (128 owner
//     rows) that only share the streamed chunk in LDS.  256 VGPRs -> two workgroups per CU.
//   * the product is computed TRANSPOSED, S^T = Y X^T (streamed rows x owner rows): the MFMA result layout
//     (lane = owner row, registers = streamed rows) is exactly the A-operand layout of the second product, so P
//     never touches LDS - it is exponentiated, rounded to bf16 and fed back from registers.  The k-order this
//     implies (register r <-> streamed row (r&3) + 8(r>>2) + 4(lane>>5)) is matched on the B side by pointing the
//     two transposing reads of a fragment at rows 16 t + 4 half + {0..3} and 16 t + 8 + 4 half + {0..3}.
//   * streamed chunks (32 rows) go global -> LDS by LDS-DMA (global_load_lds_dwordx4), double
//     buffered, one barrier per chunk, no staging registers and no ds_write pass.  LDS-DMA fills lane-linear,
//     so the tiles are unpadded and bank conflicts are removed by an XOR swizzle applied to the SOURCE address
//     of each 16-B piece and to the fragment reads (piece p of row r sits at p ^ f(r)); with f below the 16
//     lanes of every ds_read_b128 group hit 16 distinct 16-B slots.
//   * backward = ONE launch: workgroups [0, nDE) own item tiles (dE), the rest own (session tile, item range)
//     pairs (d sr partial slabs, reduced by dsr_reduce_kernel); the second kind fills the CUs the 128-row item
//     tiling leaves idle (293 tiles on 256 CUs at V = 37 484).  Range workgroups are numbered so the session
//     tiles of one item range share an XCD (blockIdx % 8) and therefore an L2.
// Soft-max statistics, exp, the one-hot subtraction and all accumulation stay fp32; dE / d sr are written fp32.
#include "common.h"
#include "score_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));

constexpr int CH = 32;    // streamed rows per chunk
constexpr int NSC = 2, PDC = 1;   // chunk ring: buffers, chunks in flight.  Measured at C3 (r02n): a 4-deep ring (3 chunks in flight)
                                  // changes nothing (98 vs 96 us) - the chunk loop is not waiting for its DMA
constexpr int SB = 512;   // streamed rows per side-data block (lse / labels / column scales in LDS)
constexpr int OWN = 128;  // owner rows per workgroup (4 waves x 32)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ unsigned short f2bf(float a) { return srec_f2bf(a); }
__device__ __forceinline__ unsigned pack2(float a, float b) { return srec_pack_bf16(a, b); }
__device__ __forceinline__ int kperm(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// dst16[r, c] = bf16(src[r, c]) (c < Dp, zero beyond d);  dstT16[c, kperm(r)] = bf16(src[r, c]);
// rows >= live R are zero.
__global__ void bf16_prepare_kernel(const float* __restrict__ src, int ld, int R, const int* __restrict__ dynR, int d,
                                    int Dp, unsigned short* __restrict__ dst16, unsigned short* __restrict__ dstT16,
                                    int Rp) {
    __shared__ unsigned short tile[64][66];
    const int Rl = dyn_count(dynR, R);
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
    for (int rr = tr; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tc;
        unsigned short v = 0;
        if (r < Rl && c < d) v = f2bf(src[(size_t)r * ld + c]);
        if (dstT16 != nullptr) tile[rr][tc] = v;
        if (c < Dp) dst16[(size_t)r * Dp + c] = v;
    }
    if (dstT16 == nullptr) return;
    __syncthreads();
    for (int cc = tr; cc < 64; cc += 4) {
        const int c = c0 + cc;
        if (c < Dp) dstT16[(size_t)c * Rp + r0 + tc] = tile[kperm(tc)][cc];   // kperm is an involution
    }
}

struct BArgs {
    const unsigned short* E16;   // items   [Vp, D]
    const unsigned short* ET16;  // items   [D, Vp]  (k-permuted)
    const unsigned short* S16;   // sessions [Bp, D]
    const unsigned short* ST16;  // sessions [D, Bp] (k-permuted)
    int Vp, Bp;
    const float* cs; const int* labels; const float* lse; const float* gscale; const float* ga; const float* gc;
    const int* dynB;
    int B, V, d;
    float* part_m; float* part_l; float* lab_logit;
    float* dE; int ld_de; int acc_dE;
    float* de_slab; int de_split, de_per;   // backward: every item tile's sessions are split over de_split workgroups (de_per sessions
    //                                         each), workgroup (tile, sb) writes slab sb [V, D] of de_slab (summed by de_reduce_kernel)
    float* part_dsr;
    int n_de_pad;                // item-tile workgroups (padded to a multiple of 8); 0 = none
    int n_ranges, chunks_per_range, n_sess_tiles;
    int rx_pref[9];              // backward: item ranges are counted per XCD (prefix sums; rx_pref[8] = slots enumerated)
};

enum { KIND_FWD = 0, KIND_BWD = 1, KIND_BWD_G = 2 };

#ifdef SREC_FLASH_TIMING   // development probe (tools/flash_timing.py): per-phase clocks of wave 0 of two workgroups
__device__ unsigned long long g_flash_tim[2][8];
__device__ unsigned long long g_flash_blk[1024][2];    // wall-clock (s_memrealtime, 100 MHz) start / end of every workgroup
#define TIM_DECL unsigned long long tim_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tim_a[6] = {0, 0, 0, 0, 0, 0}; \
    const bool tim_on = KIND == KIND_BWD && (blockIdx.x == 8 || (int)blockIdx.x == a.n_de_pad + 8)
#define TIM(i) do { __builtin_amdgcn_sched_barrier(0); tim_t[i] = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TIM_DECL
#define TIM(i)
#endif   // _G: per-session coefficients ga / gc (order fusion)

// development probe (tools/flash_knockout.py): builds of the backward with one cost removed - bit 0: no exp (P = its
// argument), 1: no accumulate product, 2: no S product, 3: no result stores, 4: fragments from registers instead of LDS.
// Results are garbage; only the launch time is read.
#ifndef SREC_FLASH_KO
#define SREC_FLASH_KO 0
#endif
constexpr int KO = SREC_FLASH_KO;
#define KO_EXP2(x) ((KO & 1) ? (x) : __builtin_amdgcn_exp2f(x))

template <int NT, int KIND_>
__global__ __launch_bounds__(256, 2) void flash_ce_bf16_kernel(BArgs a) {
    constexpr int PFD = (NT == 8 && KIND_ != KIND_FWD) ? 2 : 4;   // fragment prefetch depth (D = 256 backward sits at 256 VGPRs)
    constexpr int KIND = KIND_ == KIND_FWD ? KIND_FWD : KIND_BWD;
    constexpr bool has_g = KIND_ == KIND_BWD_G;
    constexpr int D = NT * 32, KS = D / 16, PR = D / 8;          // PR: 16-B pieces per row-major row
    constexpr int YS = CH * D;                                    // elements of one row-major chunk (= one transposed)
    constexpr int BUF = YS;                                       // ONE row-major image per chunk serves both products
    constexpr int NI = D / 16;                                    // 1-KiB LDS-DMA instructions per layout per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    float* sideF = reinterpret_cast<float*>(smem16 + NSC * BUF); // [SB] lse (dE) or column scale (d sr, fwd)
    int* sideI = reinterpret_cast<int*>(sideF + SB);             // [SB] labels (dE)
    float* sideGa = reinterpret_cast<float*>(sideI + SB);        // [SB] per-session coefficients (dE, order fusion)
    float* sideGc = sideGa + SB;
    int* sideHit = reinterpret_cast<int*>(sideGc + SB);          // [SB / CH] chunk may contain a label of this item tile (dE)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int Bd = dyn_count(a.dynB, a.B);
    TIM_DECL;
    TIM(6);
#ifdef SREC_FLASH_TIMING
    if (KIND == KIND_BWD && threadIdx.x == 0 && blockIdx.x < 1024) g_flash_blk[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif

    bool role_de = false;
    int x0, ybeg, yend, range = 0;
    if (KIND == KIND_BWD) {
        // item tiles first, then the session tiles: the dispatcher fills an XCD's CUs one workgroup each before the second
        // round, so this order already pairs an item tile with a session tile on most CUs (alternating the roles per XCD
        // position measured worse: 55.7 -> 58.4 us)
        if ((int)blockIdx.x < a.n_de_pad) {
            role_de = true;
            // (a rank of an N-GPU job scores N x 512 sessions against V / N rows: few item tiles, each with N x the sessions -
            //  34 workgroups of 128 chunks at N = 8, 244 us; split over the sessions they are 272 of 16 chunks again)
            const int tile = (int)blockIdx.x / a.de_split;
            range = (int)blockIdx.x - tile * a.de_split;            // (the slab this workgroup writes)
            x0 = tile * OWN;
            if (x0 >= a.V) return;
            ybeg = min(Bd, range * a.de_per); yend = min(Bd, ybeg + a.de_per);
        } else {
            const int b = blockIdx.x - a.n_de_pad;
            const int xcd = b & 7, j = b >> 3;
            const int tile = j % a.n_sess_tiles, kth = j / a.n_sess_tiles;   // the k-th range of this XCD (XCDs differ)
            if (kth >= a.rx_pref[xcd + 1] - a.rx_pref[xcd]) return;
            range = a.rx_pref[xcd] + kth;
            if (range >= a.n_ranges) return;
            x0 = tile * OWN;
            ybeg = range * a.chunks_per_range * CH;
            yend = min(a.V, ybeg + a.chunks_per_range * CH);
        }
    } else {
        const int b = blockIdx.x;
        const int xcd = b & 7, j = b >> 3;
        const int tile = j % a.n_sess_tiles;
        range = (j / a.n_sess_tiles) * 8 + xcd;
        if (range >= a.n_ranges) return;
        x0 = tile * OWN;
        ybeg = range * a.chunks_per_range * CH;
        yend = min(a.V, ybeg + a.chunks_per_range * CH);
    }
    // the session-tile workgroups are the long pole of the launch (23 chunks against 16): they win the issue arbitration
    // against a co-resident item-tile wave (57.8 -> 56.5 us)
    if (KIND == KIND_BWD && !role_de) __builtin_amdgcn_s_setprio(1);
    const unsigned short* X16 = role_de ? a.E16 : a.S16;
    const unsigned short* Y16 = role_de ? a.S16 : a.E16;
    const int xi = x0 + wave * 32 + l31;                          // this lane's owner row (column of S^T)

    // owner rows as MFMA B-fragments of S^T = Y X^T: row xi, k = ks*16 + 8*half .. +7 (padded copy: no masks)
    bf16x8 xf[KS];
    {
        const unsigned short* xp = X16 + (size_t)xi * D + 8 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xp + ks * 16);
    }
    // per-lane owner constants
    const float gs = (KIND == KIND_BWD) ? (a.gscale != nullptr ? *a.gscale : 1.f) / (float)(Bd > 0 ? Bd : 1) : 1.f;
    float csL = 0.f, lseL = INFINITY, gaL = gs, gcL = gs;
    int labL = -1;
    if (role_de) {
        csL = xi < a.V ? (a.cs != nullptr ? a.cs[xi] : 1.f) : 0.f;
    } else if (xi < Bd) {
        labL = a.labels[xi];
        if (KIND == KIND_BWD) {
            lseL = a.lse[xi];
            if (has_g) { const float gm = a.gscale != nullptr ? *a.gscale : 1.f; gaL = a.ga[xi] * gm; gcL = a.gc[xi] * gm; }
        }
    }

    const float csL2 = csL * LOG2E;
    if (!role_de) { lseL *= LOG2E; gaL *= LN2; gcL *= LN2; }      // d sr: the streamed column scales carry log2(e)
    f32x16 acc[KIND == KIND_BWD ? NT : 1];
    if (KIND == KIND_BWD) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    }
    float m_run = -INFINITY, l_run = 0.f;                         // forward: online soft-max of this lane's rows

    // LDS-DMA of one chunk: piece p of row r lands at position p ^ f(r) (row-major) / p ^ ((r>>2)&3) (transposed).
    // Issued as inline asm (SGPR base + 32-bit lane offset): hipcc would otherwise drain every outstanding DMA at
    // the next LDS read it cannot prove disjoint (a vmcnt(0) in the middle of the chunk); the only wait needed is
    // the explicit vmcnt(0) in front of the barrier that publishes the chunk.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem16;
    auto glds = [&](const unsigned short* sbase, unsigned voff, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    // Chunk image in LDS, k-major in 16-element blocks: block ks (1 KiB) = [32 chunk rows][k = 16 ks .. + 15], row r
    // stored at position r ^ 4 (ks & 1).  One LDS-DMA instruction fills one block (lane L <- row L >> 1, 16-B half L & 1).
    //   * product 1 reads block ks with one ds_read_b128 per lane (row l31, half = lane half): 64 lanes = the block's
    //     1 KiB, contiguous -> conflict free, address = per-lane base (even / odd ks) + immediate 1 KiB ks;
    //   * product 2's transposing reads take 4 rows x 32 B from each of the two blocks of a 32-column group per half-wave:
    //     the row flip of the odd blocks puts those 2 x 128 B into disjoint bank halves, address = ONE per-lane base +
    //     immediate (column block, k-step, read).
    // The two 16-B halves of a block row are swapped in rows 8..15 and 24..31: the 16 rows of a ds_read_b128 lane group
    // ({0-3, 12-15, 20-27}, ...) then cover all 16 slots of the 256-B bank cycle (un-swapped: 2-way conflicts, 2.4 M of 7.9 M
    // LDS cycles by SQ_LDS_BANK_CONFLICT).  For the transposing reads that swap depends on the read index e only (row bit
    // 3): two per-lane bases.
    // No XOR between lane and compile-time parts anywhere: nothing to compute per read.
    auto stage = [&](int y0, int bufsel, int lane) {
        const int rowl = (lane >> 1) ^ (4 * (wave & 1));          // instruction i = 4 ii + wave fills block i: i & 1 == wave & 1
        const unsigned voff = (unsigned)(rowl * D + (((lane & 1) ^ ((lane >> 4) & 1)) * 8)) * 2u;   // 16-B halves of rows 8..15, 24..31 swapped
#pragma unroll
        for (int ii = 0; ii < (NI + 3) / 4; ++ii) {
            const int i = ii * 4 + wave;
            if (i < NI) glds(Y16 + (size_t)y0 * D + i * 16, voff, lds0 + (unsigned)(bufsel * BUF + i * 512) * 2u);
        }
    };

    // LDS ring of NSC chunk images, PDC chunks of LDS-DMA in flight ahead of the one being multiplied
    constexpr int IPS = NI >= 4 ? NI / 4 : 1;                     // DMA instructions per chunk of the busiest wave
    const int nchunks = yend > ybeg ? (yend - ybeg + CH - 1) / CH : 0;
    for (int pc = 0; pc < PDC && pc < nchunks; ++pc) stage(ybeg + pc * CH, pc, lane);
    int lane_v = lane;   // re-derived per chunk (see the asm below): keeps ~40 loop-invariant address registers
                         // from being hoisted out of the chunk loop - the kernel lives at the 256-VGPR limit

    TIM(7);
#ifdef SREC_FLASH_TIMING
    const unsigned long long tim_loop0 = tim_t[7];
#endif
    // probe (KO bit 5): the logits come back from memory (fp16, fragment-major: 32 B per lane and chunk, as the forward would have
    // left them) instead of being recomputed, one chunk ahead; the item-tile role pays 16 two-byte LDS reads for the transposition
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 sv0 = {}, sv1 = {};
    constexpr bool KO_S = (KO & 32) && KIND == KIND_BWD;
    const size_t ko_nblk = KO_S ? (size_t)a.n_ranges * a.B * D / (64 * 8) / 2 : 1;        // 1-KiB pairs inside part_dsr
    auto ko_sload = [&](int c) {
        const h8* sp = reinterpret_cast<const h8*>(a.part_dsr) + ((((size_t)blockIdx.x * 61 + c) * 4 + wave) % ko_nblk) * 128 + lane * 2;
        sv0 = sp[0]; sv1 = sp[1];
    };
    if (KO_S && nchunks > 0) ko_sload(0);
    for (int c = 0; c < nchunks; ++c) {
        TIM(0);
        const int y0 = ybeg + c * CH;
        asm volatile("" : "+v"(lane_v));
        const int l31v = lane_v & 31, halfv = lane_v >> 5;
        unsigned short* cur = smem16 + (c % NSC) * BUF;
        const int sbase = (c % (SB / CH)) * CH;                   // offset of this chunk inside the side block
        if (sbase == 0) {
            if (c > 0) __syncthreads();                           // everyone is done with the previous side block
            for (int i = tid; i < SB; i += 256) {
                const int row = y0 + i;
                const bool ok = row < yend;
                if (role_de) {
                    const int lab = ok ? a.labels[row] : -1;
                    sideF[i] = ok ? a.lse[row] * LOG2E : INFINITY;
                    sideI[i] = lab;
                    if (has_g) { const float gm = a.gscale != nullptr ? *a.gscale : 1.f; sideGa[i] = ok ? a.ga[row] * gm : 0.f; sideGc[i] = ok ? a.gc[row] * gm : 0.f; }
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64(lab >= x0 && lab < x0 + OWN);
                    if ((lane & 31) == 0) sideHit[i >> 5] = ((hit >> (lane & 32)) & 0xffffffffull) != 0;
                } else {
                    sideF[i] = ok ? (a.cs != nullptr ? a.cs[row] * LOG2E : LOG2E) : 0.f;
                }
            }
        }
        {                                                         // this wave's share of chunk c has landed ...
            const int ahead = min(nchunks - c - 1, PDC - 1);      // (later chunks may stay in flight)
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IPS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                          // ... everyone's has; chunk c - 1's buffer is free again
        TIM(1);
        float s_pre[16];
        if (KO_S) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { s_pre[r] = (float)sv0[r]; s_pre[8 + r] = (float)sv1[r]; }
            if (role_de) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_pre[r] += (float)(smem16 + (c % NSC) * BUF)[(r * 32 + l31v) * 16 + halfv];
            }
        }
        if (c + PDC < nchunks) stage(y0 + PDC * CH, (c + PDC) % NSC, lane_v);
        if (KO_S && c + 1 < nchunks) ko_sload(c + 1);
        TIM(2);

        // ---- S^T = Y X^T : 32 streamed rows x 32 owner rows per wave, K = D.  Fragment reads run PF steps ahead of
        // the MFMAs that consume them (LDS latency ~ 4 MFMA issue slots).
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if (KO_S) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = s_pre[r];
        } else {
            constexpr int PF = KS < PFD ? KS : PFD;
            const int hsw = halfv ^ ((l31v >> 3) & 1);
            const unsigned short* ybe = cur + l31v * 16 + hsw * 8;            // even blocks
            const unsigned short* ybo = cur + (l31v ^ 4) * 16 + hsw * 8;      // odd blocks (rows flipped by 4)
            bf16x8 af[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j)
                af[j] = (KO & 16) ? xf[j] : *reinterpret_cast<const bf16x8*>(((j & 1) ? ybo : ybe) + j * 512);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (!(KO & 4)) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks % PF], xf[ks], s, 0, 0, 0);
                if (ks + PF < KS)
                    af[ks % PF] = (KO & 16) ? xf[(ks + PF) % KS]
                                            : *reinterpret_cast<const bf16x8*>((((ks + PF) & 1) ? ybo : ybe) + (ks + PF) * 512);
            }
#pragma unroll
            for (int ks = 0; ks < KS - PF; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            }
        }
        TIM(3);
        // register r of s <-> streamed row (r&3) + 8*(r>>2) + 4*half, owner row = xi
        if (KIND == KIND_FWD) {
            const int nvalid = yend - y0 - 4 * half;              // rows >= this are beyond the range
            const int labrel = labL - y0 - 4 * half;
            float z[16], m = m_run;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 c4 = *reinterpret_cast<const float4*>(sideF + sbase + 8 * q + 4 * half);
                const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e, rr = 8 * q + e;
                    z[r] = rr < nvalid ? cv[e] * s[r] : -INFINITY;          // base-2 logits (cv carries log2 e)
                    m = fmaxf(m, z[r]);
                    if (rr == labrel) a.lab_logit[xi] = z[r] * LN2;
                }
            }
            if (m != -INFINITY) {
                float l = l_run * __builtin_amdgcn_exp2f(m_run - m);
#pragma unroll
                for (int r = 0; r < 16; ++r) l += __builtin_amdgcn_exp2f(z[r] - m);
                l_run = l; m_run = m;
            }
        } else {
            // P = (softmax - onehot) * scales, in base 2: the side arrays / lane constants carry log2(e) so one fma +
            // v_exp_f32 per element; the one-hot compare runs only in the rare chunk that can contain a label.
            float p[16];
            if (role_de) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sb = sbase + 8 * q + 4 * halfv;
                    const float4 l4 = *reinterpret_cast<const float4*>(sideF + sb);
                    float4 a4 = make_float4(gs, gs, gs, gs);
                    if (has_g) a4 = *reinterpret_cast<const float4*>(sideGa + sb);
                    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q + e;
                        p[r] = KO_EXP2(__builtin_fmaf(csL2, s[r], -lv[e])) * (av[e] * csL);
                    }
                }
                if (sideHit[sbase >> 5]) {                        // some label of this chunk lies in this item tile
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sb = sbase + 8 * q + 4 * halfv;
                        const int4 i4 = *reinterpret_cast<const int4*>(sideI + sb);
                        float4 c4 = make_float4(gs, gs, gs, gs);
                        if (has_g) c4 = *reinterpret_cast<const float4*>(sideGc + sb);
                        const int iv[4] = {i4.x, i4.y, i4.z, i4.w};
                        const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (iv[e] == xi) p[4 * q + e] -= cv[e] * csL;
                    }
                }
            } else {
                const int labrel = labL - y0 - 4 * halfv;
                const float gaS = gaL;                            // already divided by log2(e): cv below carries it
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c4 = *reinterpret_cast<const float4*>(sideF + sbase + 8 * q + 4 * halfv);
                    const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q + e;
                        p[r] = KO_EXP2(__builtin_fmaf(cv[e], s[r], -lseL)) * (gaS * cv[e]);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(labrel >= 0 && labrel < 28 + 4) != 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 c4 = *reinterpret_cast<const float4*>(sideF + sbase + 8 * q + 4 * halfv);
                        const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (8 * q + e == labrel) p[4 * q + e] -= gcL * cv[e];
                    }
                }
            }
            // P as two A fragments (k-step t <-> registers 8t .. 8t+7), straight from registers
            bf16x8 pf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x8 w;
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = p[8 * t + e];
                pf[t] = __builtin_convertvector(w, bf16x8);       // v_cvt_pk_bf16_f32 (RNE)
            }
            TIM(4);
            // ---- ACC (32 owner rows x D) += P Y, K = 32 streamed rows in 2 steps.  B fragment (k-step t, column block cb):
            // lane <-> column cb*32 + l31, k slots 0..3 / 4..7 <-> streamed rows 16t + 4 half + {0..3} / 16t + 8 + 4 half +
            // {0..3} (the register order of P above).  One ds_read_b64_tr_b16 per 4 rows: inside a 16-lane group lane i
            // points at (row base + (i >> 2), columns 4 (i & 3) .. + 3 of the group's 16) and receives (rows base .. + 3,
            // column i) - measured lane map, tools/probes/tr_probe.hip.  Fragments prefetched PFB ahead of their MFMAs.
            {
                constexpr int NB = 2 * NT, PFB = NB < PFD ? NB : PFD;
                const int ti = lane_v & 15, trow = ti >> 2, b4 = (lane_v >> 4) & 1;
                // read e of fragment (cb, t): rows 16 t + 8 e + 4 half + {0..3} of the 16-column group 2 cb + b4 (= block ks);
                // lane i of the group addresses row (i >> 2), elements 4 (i & 3) .. + 3 of the block row
                const unsigned short* tb = cur + b4 * 512 + (4 * (halfv ^ b4) + trow) * 16 + 4 * (ti & 3);
                const unsigned short* tb1 = cur + b4 * 512 + (4 * (halfv ^ b4) + trow) * 16 + 4 * ((ti & 3) ^ 2);   // reads e = 1: rows 8..15 / 24..31
                auto frag = [&](int j) {
                    const int cb = j % NT, t = j / NT;
                    if (KO & 16) return xf[j % KS];
                    uint2 rr[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)((e ? tb1 : tb) + cb * 1024 + (16 * t + 8 * e) * 16));
                        rr[e] = __builtin_bit_cast(uint2, v);
                    }
                    return __builtin_bit_cast(bf16x8, make_uint4(rr[0].x, rr[0].y, rr[1].x, rr[1].y));
                };
                bf16x8 bf[PFB];
#pragma unroll
                for (int j = 0; j < PFB; ++j) bf[j] = frag(j);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (!(KO & 2)) acc[j % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[j / NT], bf[j % PFB], acc[j % NT], 0, 0, 0);
                    else acc[j % NT][0] += p[j % 16] + __builtin_bit_cast(float4, bf[j % PFB]).x;
                    if (j + PFB < NB) bf[j % PFB] = frag(j + PFB);
                }
#pragma unroll
                for (int j = 0; j < NB - PFB; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
        }
#ifdef SREC_FLASH_TIMING
        TIM(5);
        for (int i = 0; i < 5; ++i) tim_a[i] += tim_t[i + 1] - tim_t[i];
#endif
    }
#ifdef SREC_FLASH_TIMING
    TIM(0);
    const unsigned long long tim_loop1 = tim_t[0];
#endif

    if (KIND == KIND_FWD) {
        // the two halves of a wave hold disjoint rows of the same session: merge, one partial per (range, session)
        const float mo = __shfl_xor(m_run, 32, 64), lo = __shfl_xor(l_run, 32, 64);
        const float mm = fmaxf(m_run, mo);
        float l = 0.f;
        if (mm != -INFINITY) l = l_run * __builtin_amdgcn_exp2f(m_run - mm) + lo * __builtin_amdgcn_exp2f(mo - mm);
        if (half == 0 && xi < a.B) {
            a.part_m[(size_t)range * a.B + xi] = mm * LN2;        // statistics leave in natural-log units
            a.part_l[(size_t)range * a.B + xi] = l;
        }
        return;
    }
    if ((KO & 8) && acc[0][0] != 12345.678f) return;          // (keeps the accumulators alive)
    const int d = a.d;
    const bool slab = role_de && a.de_split > 1;
    float* const outp = role_de ? (slab ? a.de_slab + (size_t)range * a.V * d : a.dE) : a.part_dsr + (size_t)range * a.B * d;
    const int ld_out = role_de ? (slab ? d : a.ld_de) : d, nrows = role_de ? a.V : a.B;
    const bool accum = role_de && a.acc_dE && !slab;
    if (d == D && (ld_out & 3) == 0 && ((size_t)outp & 15) == 0) {
        // Rows leave as 16-byte stores (a 1 KiB row per wave instruction at D = 256) through a per-wave LDS patch of 8 rows
        // (the chunk ring is free now): the accumulator layout has ONE float of a row per lane, and 4-byte stores of 128
        // instructions per wave made this epilogue store-issue bound (21 k cycles of a 80 - 115 k-cycle workgroup life).
        __syncthreads();                                          // every wave is done with the last chunk image
        float* patch = reinterpret_cast<float*>(smem16) + wave * 8 * D;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) patch[(e + 4 * half) * D + cb * 32 + l31] = acc[cb][4 * q + e];
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int idx = k * 256 + lane * 4, rl = idx / D, col = idx % D;
                const int xr = x0 + wave * 32 + 8 * q + rl;
                float4 v = *reinterpret_cast<const float4*>(patch + idx);
                if (xr < nrows) {
                    float4* g = reinterpret_cast<float4*>(outp + (size_t)xr * ld_out + col);
                    if (accum) { const float4 o = *g; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *g = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
            const int col = cb * 32 + l31;
            if (col >= d) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xr = x0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (xr < nrows) {
                    float* q = outp + (size_t)xr * ld_out + col;
                    *q = accum ? *q + acc[cb][r] : acc[cb][r];
                }
            }
        }
    }
#ifdef SREC_FLASH_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TIM(1);
    if (KIND == KIND_BWD && tid == 0 && blockIdx.x < 1024) g_flash_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    if (tim_on && tid == 0) {
        unsigned long long* o = g_flash_tim[role_de ? 0 : 1];
        for (int i = 0; i < 5; ++i) o[i] = tim_a[i];
        o[5] = tim_loop0 - tim_t[6];         // prologue
        o[6] = tim_t[1] - tim_loop1;         // epilogue (stores drained)
        o[7] = tim_t[1] - tim_t[6];          // workgroup lifetime
    }
#endif
}

template <int NTV, int KIND>
int launch_b(const BArgs& a, int nblocks, hipStream_t st) {
    if (KIND == KIND_BWD && a.ga != nullptr) return launch_b<NTV, KIND == KIND_BWD ? KIND_BWD_G : KIND>(a, nblocks, st);
    constexpr int D = NTV * 32;
    constexpr size_t lds = (size_t)NSC * CH * D * sizeof(unsigned short) + 4 * SB * sizeof(float) + (SB / CH) * sizeof(int);
    static std::atomic<unsigned long long> optin{0};   // per (kernel instantiation, device)
    if (int rc = srec_lds_optin((const void*)flash_ce_bf16_kernel<NTV, KIND>, (int)lds, optin)) return rc;
    hipLaunchKernelGGL((flash_ce_bf16_kernel<NTV, KIND>), dim3(nblocks), dim3(256), lds, st, a);
    SREC_LAUNCH_CHECK();
    return 0;
}

// (96: BASELINE config C2, SRGNN / NISER at d = 96 - padded to 128 a quarter of both products multiplied zero columns)
inline int dpad(int d) { return d <= 32 ? 32 : d <= 64 ? 64 : d <= 96 ? 96 : d <= 128 ? 128 : 256; }

template <int KIND>
int launch_kind(const BArgs& a, int nblocks, hipStream_t st) {
    switch (dpad(a.d)) {
        case 32: return launch_b<1, KIND>(a, nblocks, st);
        case 64: return launch_b<2, KIND>(a, nblocks, st);
        case 96: return launch_b<3, KIND>(a, nblocks, st);
        case 128: return launch_b<4, KIND>(a, nblocks, st);
        default: return launch_b<8, KIND>(a, nblocks, st);
    }
}

// item ranges of the session-owner workgroups: `slots` workgroups wanted in total, at most `rmax` ranges
void pick_ranges_b(int B, int V, int slots, int rmax, int* n_ranges, int* chunks_per_range) {
    const int T = cdiv(B, OWN), chunks = cdiv(V, CH);
    int R = slots / T;
    if (R > rmax) R = rmax;
    if (R > chunks) R = chunks;
    if (R < 1) R = 1;
    const int cpr = cdiv(chunks, R);
    *chunks_per_range = cpr;
    *n_ranges = cdiv(chunks, cpr);
}

constexpr int FWD_SLOTS = 512, FWD_RMAX = 1024, BWD_RMAX = 64;
constexpr int XCDS = 8, XCD_SLOTS = 64;                   // 32 CUs x 2 resident workgroups per XCD

// Item ranges of the backward's session-owner workgroups.  Workgroups go to the XCDs round-robin by index and a
// workgroup waits for a slot on ITS XCD: 293 item tiles + 54 ranges x 4 session tiles (509 of the chip's 512 slots)
// put 65 live workgroups on five of the XCDs, whose 65th then ran after the first had finished - the launch took two
// workgroup lifetimes (95 us) instead of one (68 us).  So count per XCD: ranges per XCD = free slots of the fullest
// XCD / session tiles.
// rx_pref (nullable): prefix sums of the ranges per XCD; returns the largest per-XCD count.  An XCD hosts
// floor(free slots / session tiles) ranges, free = 64 - (its item tiles in the last round): at V = 37 484 five XCDs carry 37
// item tiles (6 ranges) and three carry 36 (7 ranges) - 51 ranges instead of 8 x 6.
int pick_ranges_bwd(int B, int V, bool with_de, int* n_ranges, int* chunks_per_range, int* rx_pref = nullptr, int de_split = 1) {
    const int T = cdiv(B, OWN), chunks = cdiv(V, CH), de_tiles = cdiv(V, OWN) * de_split;
    int rx[XCDS], total = 0;
    for (int x = 0; x < XCDS; ++x) {
        int free_slots = XCD_SLOTS;
        if (with_de) {
            const int cnt = x < de_tiles ? (de_tiles - x + XCDS - 1) / XCDS : 0;
            if (cnt % XCD_SLOTS) free_slots = XCD_SLOTS - cnt % XCD_SLOTS;
        }
        rx[x] = free_slots / T;
        total += rx[x];
    }
    if (total == 0) { for (int x = 0; x < XCDS; ++x) rx[x] = 1; total = XCDS; }
    int cap = BWD_RMAX < chunks ? BWD_RMAX : chunks;
    while (total > cap) {                                  // trim the fullest XCDs first
        int best = 0;
        for (int x = 1; x < XCDS; ++x) if (rx[x] > rx[best]) best = x;
        --rx[best]; --total;
    }
    const int cpr = cdiv(chunks, total);
    *chunks_per_range = cpr;
    *n_ranges = cdiv(chunks, cpr);                         // ranges behind this one are empty: their workgroups exit
    int mx = 0, acc = 0;
    for (int x = 0; x < XCDS; ++x) {
        if (rx_pref != nullptr) rx_pref[x] = acc;
        acc += rx[x];
        if (rx[x] > mx) mx = rx[x];
    }
    if (rx_pref != nullptr) rx_pref[XCDS] = acc;
    return mx;
}

inline bool bad_d(int d) { return d <= 0 || d > 256 || (d & 3); }

// session split of the backward's item tiles: ~293 workgroups of that kind fill the chip next to the d-sr ranges (the C3
// shape: 293 tiles x 16 chunks); fewer, longer tiles (a row shard scored against the sessions of all ranks) are cut into
// pieces of >= 512 sessions
inline int de_split_for(int B, int V) {
    const int tiles = cdiv(V, OWN);
    int s = (293 + tiles / 2) / tiles;
    const int cap = B / 512;
    if (s > cap) s = cap;
    if (s > 16) s = 16;
    // ... and the launch stays ONE round of the chip: per XCD its share of the item-tile pieces + one range of every session tile
    // within the 64 resident workgroups (8 pieces x 34 tiles + 8 x 32 session tiles = 528 on 512 slots took two workgroup lives:
    // 83.6 us at 4 096 x 4 332)
    const int T = cdiv(B, OWN);
    while (s > 1 && cdiv(tiles * s, XCDS) + T > XCD_SLOTS) --s;
    return s < 1 ? 1 : s;
}

// dE (+)= sum of the session-split slabs, in slab order (deterministic)
__global__ void de_reduce_kernel(const float* __restrict__ part, int R, size_t n, float* __restrict__ out, int acc) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 s = acc ? *reinterpret_cast<const float4*>(out + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = 0; r0 < R; r0 += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = r0 + k < R ? *reinterpret_cast<const float4*>(part + (size_t)(r0 + k) * n + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    }
    *reinterpret_cast<float4*>(out + i) = s;
}

}  // namespace

// rows [R, d] fp32 -> dst16 [Rp, Dp] and dstT16 [Dp, Rp] bf16 (RNE), zero for rows >= live R and columns >= d;
// Dp = d padded to 32/64/96/128/256 (srec_ce_plan_bf16), Rp % 128 == 0.  The transposed copy is stored in the
// MFMA k-order (bits 2 and 3 of the row index swapped).
extern "C" int srec_bf16_prepare(const float* src, int ld, int R, const int* dynR, int d, void* dst16, void* dstT16,
                                 int Rp, void* stream) {
    if (R <= 0 || (Rp & 127) || Rp < R || bad_d(d)) return SREC_BAD_ARG;
    const int Dp = dpad(d);
    hipLaunchKernelGGL(bf16_prepare_kernel, dim3(Rp / 64, cdiv(Dp, 64)), dim3(256), 0, (hipStream_t)stream, src, ld, R,
                       dynR, d, Dp, (unsigned short*)dst16, (unsigned short*)dstT16, Rp);
    SREC_LAUNCH_CHECK();
    return 0;
}

#ifdef SREC_FLASH_TIMING
extern "C" int srec_flash_timing(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_flash_tim), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
extern "C" int srec_flash_blocks(unsigned long long* out2048) {
    return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_flash_blk), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : 1;
}
#endif

// workspace plan: ws_stats >= 2 * n_stat_slabs * B floats, ws_dsr >= n_ranges * B * d floats, copies [.., d_pad]
extern "C" int srec_ce_plan_bf16(int B, int V, int d, int* n_stat_slabs, int* n_ranges, int* d_pad) {
    if (bad_d(d) || B <= 0 || V <= 0) return SREC_BAD_ARG;
    int R, cpr, R2;
    pick_ranges_b(B, V, FWD_SLOTS, FWD_RMAX, &R, &cpr);
    *n_stat_slabs = R;
    pick_ranges_bwd(B, V, true, &R, &cpr);
    pick_ranges_bwd(B, V, false, &R2, &cpr);
    *n_ranges = R > R2 ? R : R2;
    if (d == dpad(d) && de_split_for(B, V) > 1) {          // (the session-split launch counts its item-tile workgroups differently)
        pick_ranges_bwd(B, V, true, &R2, &cpr, nullptr, de_split_for(B, V));
        if (R2 > *n_ranges) *n_ranges = R2;
    }
    *d_pad = dpad(d);
    return 0;
}

// sr16 [Bp, Dp], E16 [Vp, Dp] from srec_bf16_prepare.  Outputs as srec_score_ce_fwd.
extern "C" int srec_score_ce_fwd_bf16(const void* sr16, int Bp, const void* E16, int Vp, const float* cs,
                                      const int* labels, int B, int V, int d, const int* dynB, float* ws_stats,
                                      float* lab_logit, float* lse, float* lossvec, float* loss, void* stream) {
    if (bad_d(d) || (Bp & 127) || (Vp & 127) || Bp < B || Vp < V) return SREC_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    BArgs a{};
    a.S16 = (const unsigned short*)sr16; a.E16 = (const unsigned short*)E16; a.Bp = Bp; a.Vp = Vp;
    a.cs = cs; a.labels = labels; a.dynB = dynB; a.B = B; a.V = V; a.d = d;
    a.n_sess_tiles = cdiv(B, OWN);
    pick_ranges_b(B, V, FWD_SLOTS, FWD_RMAX, &a.n_ranges, &a.chunks_per_range);
    const int nt = a.n_ranges;
    a.part_m = ws_stats; a.part_l = ws_stats + (size_t)nt * B; a.lab_logit = lab_logit;
    int rc = launch_kind<KIND_FWD>(a, 8 * cdiv(nt, 8) * a.n_sess_tiles, st);
    if (rc) return rc;
    // (one single-workgroup launch for both was measured: 16.8 us against 5 + 5 - a lone workgroup cannot hide the latency
    // of its ~100 loads per session behind other workgroups; profiles/r03b_*)
    hipLaunchKernelGGL(ce_reduce_stats_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, a.part_m, a.part_l, lab_logit, nt, B,
                       dynB, lse, lossvec);
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, st, lossvec, B, dynB, loss);
    SREC_LAUNCH_CHECK();
    return 0;
}

// backward with the bf16 copies (row-major and transposed) of both operands; dE / dsr fp32 as in srec_score_ce_bwd.
// workgroups per item tile the backward should split the sessions over at this shape (1: none) - the caller allocates
// ws_de >= that many x V x d floats and passes the count in `parts` bits 8 - 15
extern "C" int srec_ce_de_split(int B, int V, int d, int* split) {
    if (split == nullptr || bad_d(d) || B <= 0 || V <= 0) return SREC_BAD_ARG;
    *split = d == dpad(d) ? de_split_for(B, V) : 1;
    return 0;
}

extern "C" int srec_score_ce_bwd_bf16(const void* sr16, float* ws_de, int Bp, const void* E16, const void* ET16,
                                      int Vp, const float* cs, const int* labels, const float* lse,
                                      const float* gscale, const float* ga, const float* gc, int B, int V, int d,
                                      const int* dynB, float* dE, int ld_de, float* ws_dsr, float* dsr, int parts,
                                      void* stream) {
    if (bad_d(d) || (Bp & 127) || (Vp & 127) || Bp < B || Vp < V) return SREC_BAD_ARG;
    if (!(parts & 3)) return 0;
    hipStream_t st = (hipStream_t)stream;
    BArgs a{};
    a.S16 = (const unsigned short*)sr16; a.ST16 = nullptr; a.Bp = Bp;
    a.E16 = (const unsigned short*)E16; a.ET16 = (const unsigned short*)ET16; a.Vp = Vp;
    a.cs = cs; a.labels = labels; a.lse = lse; a.gscale = gscale; a.ga = ga; a.gc = gc; a.dynB = dynB;
    a.B = B; a.V = V; a.d = d; a.dE = dE; a.ld_de = ld_de; a.acc_dE = (parts & 4) ? 1 : 0; a.part_dsr = ws_dsr;
    a.n_sess_tiles = cdiv(B, OWN);
    const bool with_de = parts & 1, with_dsr = parts & 2;
    // session split of the item tiles: the caller says how many slabs ws_de holds (parts bits 8 - 15, from srec_ce_de_split)
    a.de_split = 1; a.de_slab = nullptr; a.de_per = B;
    const int ds = (parts >> 8) & 0xff;
    if (with_de && ws_de != nullptr && ds > 1 && ld_de == d && d == dpad(d) && ((size_t)V * d) % 4 == 0) {
        a.de_split = ds; a.de_slab = ws_de;
        a.de_per = (cdiv(B, ds) + CH - 1) / CH * CH;
    }
    a.n_de_pad = with_de ? 8 * cdiv(cdiv(V, OWN) * a.de_split, 8) : 0;
    int nblocks = a.n_de_pad;
    if (with_dsr) {
        const int rmax = pick_ranges_bwd(B, V, with_de, &a.n_ranges, &a.chunks_per_range, a.rx_pref, a.de_split);
        nblocks += 8 * rmax * a.n_sess_tiles;
    } else {
        a.n_ranges = 0; a.chunks_per_range = 1;
    }
    int rc = launch_kind<KIND_BWD>(a, nblocks, st);
    if (rc) return rc;
    if (a.de_split > 1) {
        const size_t n = (size_t)V * d;
        hipLaunchKernelGGL(de_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, a.de_slab, a.de_split, n, dE,
                           (parts & 4) ? 1 : 0);
    }
    if (with_dsr && !(parts & 8)) {
        const size_t n = (size_t)B * d;
        hipLaunchKernelGGL(dsr_reduce_kernel, dim3((unsigned)cdiv((int)(n / 4), 64)), dim3(256), 0, st, ws_dsr,
                           a.n_ranges, n, dsr);
    }
    SREC_LAUNCH_CHECK();
    return 0;
}
