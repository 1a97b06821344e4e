// MSHGNN layer backward WITHOUT the projection gradients dP in memory - the weight-gradient product.
//
// Reference: /root/reference/src/models/msgifsr.py:47-91 (MSHGNN: conv1 + conv2 over the reversed graph, sum over relations,
// max over the 8 heads) and /root/reference/src/models/gnn_models/gatconv.py:267-311 (fc projection, edge soft-max, u_mul_e /
// sum aggregation) under autograd: d fc.weight = dP^T x with dP [rows, H D] the gradient of the projection P = x fc^T.
//
// Round 5 materialised dP (90.8 MB of bf16 per layer call at the C3 shapes, written by hg_bwd_src, read back by the
// backward-data and the weight-gradient GEMM).  But dP has structure:
//     dP[u, h, j] = sum_{e: u -> v} a[e, h] g[v, j] [arg[v, j] == h]  +  wL[u, h] a_l[h, j]  +  wR[u, h] a_r[h, j]
// (a = edge soft-max x attention-dropout multiplier, g = gradient of the layer output, arg = the head that won the max, wL / wR
// = the attention-logit gradients; the a_l / a_r terms because the attention vectors are folded into fc: hgat.hip).  So
//     dW[hD + j, c] = sum_e a[e, h] g[v_e, j] [arg[v_e, j] == h] x[u_e, c]              (a GEMM whose reduction runs over EDGES)
//                   + a_l[hD + j] Z_l[h, c] + a_r[hD + j] Z_r[h, c]                     (rank-1 per head; Z = x^T wL / x^T wR is
//                                                                                        already computed for d attn: hg_colsum)
// The edge GEMM's A operand is GENERATED while it is staged: a workgroup owns 16 feature columns j x all 8 heads (128 columns
// of dW^T in the order m = 8 j + h: the one non-zero of an (edge, j) pair then lies inside ONE 16-byte piece) x all D input
// columns c; per 32-edge step every thread builds two 16-byte pieces of the A tile from g / arg / a (L2-resident: 5.6 MB /
// 1.4 MB / 0.6 MB) and fetches four 16-byte pieces of the gathered source rows x16[u_e, :] (the B tile), all through registers
// into a 2-stage LDS ring; the MFMA side is gemm16_tn's (row-major tiles, ds_read_b64_tr_b16 fragments, 32x32x16 bf16).
// There are fewer edges (18.7 k) than projected rows (22.2 k: every module projects all rows of its source types), nothing is
// read from HBM but 14 MB of L2-resident operands, and the 2 x 90.8 MB dP round trip of this product is gone.
#include "common.h"
#include "../../include/srec_hg.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
constexpr int MAXM = SREC_HG_MAXM, MAXI = SREC_HG_MAXI;

struct WgArgs {
    // modules
    const unsigned short* x16[MAXM];     // bf16 input rows [NT, D] the module projected (the conv's dropped copy)
    float* out[MAXM];                    // [nsplit][H D, D] slabs (nsplit == 1: the gradient itself)
    const float* al[MAXM]; const float* ar[MAXM]; const float* Z[MAXM];
    int nsplit[MAXM], ninst[MAXM], inst[MAXM][4];
    int n_units, unit_mod[64], unit_split[64];      // (module, split) pairs, 16 column tiles each
    // relation instances
    const unsigned short* am_base;       // Am[i] [E, 8] bf16 = am_base + am_off[i]: soft-max weight x attention-dropout
    int am_off[MAXI];                    // multiplier per edge and head (written by srec_hg_bwd)
    const int* esrc[MAXI]; const int* edst[MAXI]; const int* out_ptr[MAXI];
    const int* dyn_s[MAXI];
    int ncap_s[MAXI], row0_s[MAXI], row0_d[MAXI];
    int nm, D;
    const float* g; int ld_g;
    const unsigned char* arg;
};

__device__ __forceinline__ uint2 lds_tr16(const unsigned short* base, unsigned byte_off) {
    const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4_t*)(reinterpret_cast<const char*>(base) + byte_off));
    return __builtin_bit_cast(uint2, v);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// one 1-KiB LDS-DMA (gemm16.hip): lane l copies 16 B from sbase + voff[l] to LDS address lds_dst + 16 l
__device__ __forceinline__ void glds16(const unsigned short* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst_s = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst_s) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

#ifdef SREC_HGW_TIMING
__device__ unsigned long long g_hgw_tim[9];
#define HGWT(i) do { __builtin_amdgcn_sched_barrier(0); tm[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
// per-step phases of the steady-state loop, accumulated: barrier, load issue, reads + MFMA, wait for chunk it + 1, A pieces + stores
#define HGWP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                     if ((i) > 0) ph[(i) - 1] += t_ - tp; tp = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HGWT(i)
#define HGWP(i)
#endif

// one 16-byte piece of the A tile: the 8 heads of (edge, feature column j); the one non-zero is head `hw` (the arg-max byte;
// hw >= 8: none - a row past the live edges).  am = the edge's 8 bf16 weights A x Mk.
// (the weight of head hw as a SUM of selects: a chain `s = hw == h ? w[h] : s` is recognised as w[hw] by hipcc and turned into
//  a scratch array with a dynamic index - a memory round trip per piece inside the loop)
__device__ __forceinline__ uint4 a_piece(const uint4& am, unsigned hw, float gv) {
    const unsigned d = hw >> 1;
    const unsigned pr = (d == 0u ? am.x : 0u) | (d == 1u ? am.y : 0u) | (d == 2u ? am.z : 0u) | (d == 3u ? am.w : 0u);
    const float w = __uint_as_float((hw & 1u) ? (pr & 0xffff0000u) : (pr << 16));
    const unsigned v = (unsigned)srec_f2bf(w * gv) << ((hw & 1u) * 16u);
    return make_uint4(d == 0u ? v : 0u, d == 1u ? v : 0u, d == 2u ? v : 0u, d == 3u ? v : 0u);
}

constexpr int BR = 32;                    // edges per step
constexpr int TM = 128;                   // columns of dW^T per workgroup: 16 feature columns x 8 heads
constexpr int A_STG = BR * TM;            // bf16 elements of an A tile
constexpr int STG = A_STG + 2 * BR * 128; // ... of a stage: A tile + the two 128-column halves of the B tile
constexpr int LDS_ELEMS = 2 * STG;        // two stages: stage it + 1 is written while stage it is read
constexpr int NIDX = 48;                  // chunks whose edge ids fit the LDS table (a longer list is processed in rounds)

// grid: for every module 16 column tiles x nsplit[m] interleaved edge-chunk subsets.  D == 256, H == 8.
__global__ __launch_bounds__(256, 2) void hg_wgrad_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
#ifdef SREC_HGW_TIMING
    unsigned long long tm[8], ph[5] = {0, 0, 0, 0, 0}, tp = 0;
    HGWT(0);
    auto wait_landed = [&]() { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); };    // (what the compiler's own wait amounts to)
#else
    auto wait_landed = [&]() {};
#endif
    // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin by index and every XCD has its own L2.  The 16 column tiles
    // of a (module, split) unit gather the SAME source rows x16[u_e, :] (9.6 MB over the launch, 16 x re-read): they sit on one
    // XCD, side by side, so that one of them pulls a row from the fabric and fifteen find it in L2 (plain order: 68 us, the
    // gathers ran at the fabric's random-access rate).
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int unit = (slot >> 4) * 8 + xcd, jt = slot & 15;
    if (unit >= a.n_units) return;
    const int m = a.unit_mod[unit], split = a.unit_split[unit];
    const int D = a.D, HD = 8 * D;
    const int nsplit = a.nsplit[m];
    const int j0 = jt * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;

    // ---- the module's edge chunks: instance q holds ceil(E_q / BR) chunks; this workgroup takes chunks c = split (mod nsplit).
    // The MFMA loop must not chase pointers: every dependent load in it is an exposed memory round trip (first version: 12 k
    // cycles per 32-edge step, 211 us for the launch).  The ids of up to NIDX chunks are fetched up front into an LDS table
    // (one pass, coalesced), the loop then issues nothing but independent operand loads.
    __shared__ int idt[NIDX][3][BR];                 // per chunk: stacked destination row (-1: past the live edges), stacked
    //                                                  source row, element offset of the edge's weights in Am
    int cbeg[5], Eq[4];
    cbeg[0] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool on = q < a.ninst[m];
        const int i = on ? a.inst[m][q] : a.inst[m][0];
        Eq[q] = on ? a.out_ptr[i][dyn_count(a.dyn_s[i], a.ncap_s[i])] : 0;
        cbeg[q + 1] = cbeg[q] + (Eq[q] + BR - 1) / BR;
    }
    const int total = cbeg[4];
    const int nit_all = total > split ? (total - split + nsplit - 1) / nsplit : 0;
    auto fill_ids = [&](int base, int cnt) {         // table rows 0 .. cnt - 1 <- chunks base .. base + cnt - 1 of this workgroup
        for (int t = tid; t < cnt * BR; t += 256) {
            const int k = t / BR, r = t % BR;
            const int c = split + (base + k) * nsplit;
            const int q = (c >= cbeg[1] ? 1 : 0) + (c >= cbeg[2] ? 1 : 0) + (c >= cbeg[3] ? 1 : 0);
            const int i = a.inst[m][q];
            const int E = q == 0 ? Eq[0] : q == 1 ? Eq[1] : q == 2 ? Eq[2] : Eq[3];
            const int cb = q == 0 ? cbeg[0] : q == 1 ? cbeg[1] : q == 2 ? cbeg[2] : cbeg[3];
            const int e = (c - cb) * BR + r, ec = min(e, E - 1);
            idt[k][0][r] = e < E ? a.row0_d[i] + a.edst[i][ec] : -1;
            idt[k][1][r] = a.row0_s[i] + a.esrc[i][ec];                // (rows past the live edges repeat a live row: times zero)
            idt[k][2][r] = a.am_off[i] + ec * 8;
        }
    };

    // thread roles.  A tile: edge row ge = tid >> 3 of the chunk, feature columns j0 + 2 jq, + 1 (jq = tid & 7).
    //                B tile: rows brow + 8 k (k < 4), 16-byte piece bp = tid & 31 of the 512-byte source row x16[u_e, :].
    const int ge = tid >> 3, jq = tid & 7;
    const int bp = tid & 31, brow = tid >> 5;
    // LDS element offsets (bf16): A row-major [BR][128], piece p of row r in slot p ^ (4 (r & 3)); B halves [2][BR][128] alike
    const unsigned a_off = (unsigned)(ge * TM + (((2 * jq) ^ (4 * (ge & 3))) << 3));
    unsigned b_off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = brow + 8 * k;
        b_off[k] = (unsigned)((bp >> 4) * (BR * 128) + r * 128 + (((bp & 15) ^ (4 * (r & 3))) << 3));
    }
    typedef const __attribute__((address_space(1))) float* gfp;
    typedef const __attribute__((address_space(1))) int* gip;
    typedef const __attribute__((address_space(1))) unsigned short* gup;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // ---- register staging, three chunks deep.  A gather out of L2 / MALL takes 1.5 - 2 us here against ~0.3 us of MFMA + LDS work
    // per chunk: with one chunk in flight the launch took 101 us (64 us with two workgroups per CU).  Every load is a plain load
    // the compiler tracks (an LDS-DMA for the B tile, invisible to hipcc's vmcnt bookkeeping, made it wait for the newest
    // operations where the oldest were meant - the counter retires in order; and a per-step id load, consumed one step later,
    // drained the operand loads issued before it).  Three named register sets rotate with the step (unrolled by three).
    struct L2 { float gx, gy; unsigned av; uint4 am; uint4 b[4]; bool live; };
    L2 da{}, db{};
    auto load_data = [&](int k, L2& o) {         // chunk k of the table: 4 pieces of the B tile + what the A pieces are made of
        const int vrow = idt[k][0][ge];
        const int amo = idt[k][2][ge];
        int srow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) srow[q] = idt[k][1][brow + 8 * q];
        o.live = vrow >= 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 w = *(const __attribute__((address_space(1))) u32x4*)((gup)a.x16[m] + (size_t)srow[q] * D + bp * 8);
            o.b[q] = make_uint4(w.x, w.y, w.z, w.w);
        }
        const size_t v = (size_t)max(vrow, 0);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 t = *(const __attribute__((address_space(1))) f32x2*)((gfp)a.g + v * a.ld_g + j0 + 2 * jq);
        o.gx = t.x; o.gy = t.y;
        o.av = *(const __attribute__((address_space(1))) unsigned short*)(
            (const __attribute__((address_space(1))) unsigned char*)a.arg + v * D + j0 + 2 * jq);
        const u32x4 w = *(const __attribute__((address_space(1))) u32x4*)((gup)a.am_base + amo);
        o.am = make_uint4(w.x, w.y, w.z, w.w);
    };
    auto store_stage = [&](const L2& c, int st) {    // registers -> LDS stage st (of two)
        unsigned short* base = smem + st * STG;
        const uint4 p0 = a_piece(c.am, c.live ? (c.av & 0xffu) : 8u, c.gx);
        const uint4 p1 = a_piece(c.am, c.live ? (c.av >> 8) : 8u, c.gy);
        *reinterpret_cast<uint4*>(base + a_off) = p0;
        *reinterpret_cast<uint4*>(base + a_off + 8) = p1;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(base + A_STG + b_off[k]) = c.b[k];
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing fragment reads (gemm16.hip tn16): reduction rows kb + 4 e + (i >> 2), columns cbase + 16 ((lane >> 4) & 1) + 4 (i & 3)
    const int ti = lane & 15, tr = ti >> 2;
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (ti & 3);
    auto tr_off = [&](int col, int row) {
        return (unsigned)(row * 256 + ((((col >> 3) ^ (4 * (row & 3))) << 4) | ((col & 7) << 1)));
    };
    auto compute = [&](int st) {
        const unsigned short* abase = smem + st * STG;
        const unsigned short* bbase = abase + A_STG + wn * (BR * 128);
#pragma unroll
        for (int ks = 0; ks < BR / 16; ++ks) {
            const int kb = ks * 16 + 8 * half;
            uint2 ra[2][2], rb[4][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int f = 0; f < 2; ++f) ra[f][e] = lds_tr16(abase, tr_off(wm * 64 + f * 32 + tcol, kb + 4 * e + tr));
#pragma unroll
                for (int f = 0; f < 4; ++f) rb[f][e] = lds_tr16(bbase, tr_off(f * 32 + tcol, kb + 4 * e + tr));
            }
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int f = 0; f < 2; ++f) fa[f] = __builtin_bit_cast(bf16x8, make_uint4(ra[f][0].x, ra[f][0].y, ra[f][1].x, ra[f][1].y));
#pragma unroll
            for (int f = 0; f < 4; ++f) fb[f] = __builtin_bit_cast(bf16x8, make_uint4(rb[f][0].x, rb[f][0].y, rb[f][1].x, rb[f][1].y));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    HGWT(1);
    for (int base = 0; base < nit_all; base += NIDX) {
        const int nit = min(NIDX, nit_all - base);
        __syncthreads();                                   // (the previous round is done with the table and the stages)
        fill_ids(base, nit);
        __syncthreads();
        // prologue: chunk 0 in LDS stage 0, chunk 1 in flight (set db)
        load_data(0, da);
        load_data(min(1, nit - 1), db);
        store_stage(da, 0);
        // steady state, branch-free, two chunks per trip.  At the top of step it: stage it is complete in LDS, the data of
        // chunk it + 1 (set DN) is in flight or landed, the other set (DF) is free after this step's store... so the loads of
        // chunk it + 2 go BEHIND the store of chunk it + 1 (same registers): one chunk in flight under the MFMAs, a second
        // workgroup on the CU fills the gaps.
#define HGW_STEP(IT, DN, DF)                                                                                               \
        {                                                                                                                  \
            HGWP(0);                                                                                                       \
            __syncthreads();                               /* stage IT visible; stage IT + 1 free (read in IT - 1) */      \
            HGWP(1);                                                                                                       \
            load_data(min((IT) + 2, nit - 1), DF);         /* chunk IT + 2 */                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            HGWP(2);                                                                                                       \
            compute((IT) & 1);                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            HGWP(3);                                                                                                       \
            wait_landed();                                                                                                 \
            HGWP(4);                                                                                                       \
            store_stage(DN, ((IT) + 1) & 1);               /* chunk IT + 1: issued a step ago */                            \
            HGWP(5);                                                                                                       \
        }
        // set use: chunk c lives in set c % 2 = (da, db); step it stores chunk it + 1 and loads chunk it + 2 into the set that
        // held chunk it (stored one step earlier)
        int it = 0;
        for (; it + 2 < nit; it += 2) {
            HGW_STEP(it, db, da)
            HGW_STEP(it + 1, da, db)
        }
        // tail: at most two chunks left (it in LDS, it + 1 in db), nothing more to fetch
        __syncthreads();
        compute(it & 1);
        if (it + 1 < nit) {
            store_stage(db, (it + 1) & 1);
            __syncthreads();
            compute((it + 1) & 1);
        }
#undef HGW_STEP
    }
    HGWT(3);
    // ---- epilogue: column m = 8 jl + h of the tile is row h D + j0 + jl of dW; 16-byte stores through a per-wave LDS patch.
    // Split 0 adds the rank-1 terms a_l[r] Z_l[h, c] + a_r[r] Z_r[h, c].
    constexpr int PS = 36;
    __syncthreads();
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PS);
    float* __restrict__ C = a.out[m] + (size_t)split * HD * D;
    const float* __restrict__ Z = a.Z[m];
    const bool rank1 = split == 0 && Z != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * PS + l31] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int idx = q4 * 64 + lane, rl = idx >> 3, c4 = (idx & 7) * 4;
                const int ml = wm * 64 + i * 32 + rl;              // tile column m = 8 jl + h
                const int h = ml & 7, row = h * D + j0 + (ml >> 3), col = wn * 128 + j * 32 + c4;
                float4 v = *reinterpret_cast<const float4*>(patch + rl * PS + c4);
                if (rank1) {
                    const float l = a.al[m][row], rr = a.ar[m][row];
                    const float4 zl = *reinterpret_cast<const float4*>(Z + (size_t)h * D + col);
                    const float4 zr = *reinterpret_cast<const float4*>(Z + (size_t)HD + (size_t)h * D + col);
                    v.x += l * zl.x + rr * zr.x; v.y += l * zl.y + rr * zr.y; v.z += l * zl.z + rr * zr.z; v.w += l * zl.w + rr * zr.w;
                }
                *reinterpret_cast<float4*>(C + (size_t)row * D + col) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    HGWT(4);
#ifdef SREC_HGW_TIMING
    if (blockIdx.x == 8 * 16 + 5 * 8 && tid == 0) {          // (XCD 0, second unit of it, column tile 5)
        g_hgw_tim[0] = tm[1] - tm[0]; g_hgw_tim[1] = tm[3] - tm[1]; g_hgw_tim[2] = tm[4] - tm[3];
        for (int i = 0; i < 5; ++i) g_hgw_tim[3 + i] = ph[i];
        g_hgw_tim[8] = (unsigned long long)nit_all;
    }
#endif
}

}  // namespace

#ifdef SREC_HGW_TIMING
extern "C" int srec_hgw_timing(unsigned long long* tim9) {
    return (int)hipMemcpyFromSymbol(tim9, HIP_SYMBOL(g_hgw_tim), sizeof(unsigned long long) * 9);
}
#endif

// d fc.weight of every GAT module of one MSHGNN layer call from the edge lists (see the head of this file).
//   desc     the layer descriptor of srec_hg_bwd (A / Mk / esrc / edst / out_ptr per instance, attn_l / attn_r / Z per module: Z as
//            srec_hg_bwd left it)
//   x16      HOST array of n_mods device pointers: the bf16 input rows [NT, D] module m projected
//   g, arg   gradient of the layer output [NT, ld_g] fp32 and the arg-max head bytes [NT, D]
//   out      HOST array of n_mods device pointers: [nsplit[m]][H D, D] fp32 slabs (slab 0 carries the rank-1 attention terms);
//            nsplit: HOST int array (>= 1 each; the caller sums the slabs: srec_sum_slabs_multi)
// H == 8, D == 256, <= 4 instances per module.
extern "C" int srec_hg_wgrad(const void* desc_, const void* x16, const float* g, int ld_g, const unsigned char* arg,
                             const void* out, const int* nsplit, void* stream) {
    const srec_hg_desc* d = (const srec_hg_desc*)desc_;
    if (d == nullptr || x16 == nullptr || out == nullptr || nsplit == nullptr || g == nullptr || arg == nullptr) return SREC_BAD_ARG;
    if (d->H != 8 || d->D != 256 || d->n_mods <= 0 || d->n_mods > MAXM || d->n_inst > MAXI || (ld_g & 1)) return SREC_BAD_ARG;
    WgArgs a{};
    a.nm = d->n_mods; a.D = d->D; a.g = g; a.ld_g = ld_g; a.arg = arg;
    for (int m = 0; m < d->n_mods; ++m) {
        a.x16[m] = ((const unsigned short* const*)x16)[m];
        a.out[m] = ((float* const*)out)[m];
        a.al[m] = d->attn_l[m]; a.ar[m] = d->attn_r[m]; a.Z[m] = d->Z[m];
        if (a.x16[m] == nullptr || a.out[m] == nullptr || nsplit[m] < 1 || nsplit[m] > 8) return SREC_BAD_ARG;
        a.nsplit[m] = nsplit[m];
    }
    // units longest first is not needed: one unit = 16 workgroups with the same trip count; modules in order, splits side by side
    for (int m = 0; m < d->n_mods; ++m)
        for (int sp = 0; sp < nsplit[m]; ++sp) {
            if (a.n_units >= 64) return SREC_BAD_ARG;
            a.unit_mod[a.n_units] = m; a.unit_split[a.n_units] = sp; ++a.n_units;
        }
    const int blocks = ((a.n_units + 7) / 8) * 16 * 8;
    for (int i = 0; i < d->n_inst; ++i) {
        const int m = d->inst_mod[i], sb = d->inst_sblk[i], db = d->inst_dblk[i];
        if (a.ninst[m] >= 4) return SREC_BAD_ARG;
        a.inst[m][a.ninst[m]++] = i;
        a.esrc[i] = d->esrc[i];
        if (d->Am[i] == nullptr) return SREC_BAD_ARG; a.edst[i] = d->edst[i]; a.out_ptr[i] = d->out_ptr[i];
        const int ts = d->blk_type[sb], td = d->blk_type[db];
        a.dyn_s[i] = d->dyn_n[ts]; a.ncap_s[i] = d->ncap[ts]; a.row0_s[i] = d->row0[ts]; a.row0_d[i] = d->row0[td];
    }
    {   // the Am arrays live in one scratch allocation: offsets (bf16 elements) from the lowest of them
        const unsigned short* lo = (const unsigned short*)d->Am[0];
        for (int i = 1; i < d->n_inst; ++i) if ((const unsigned short*)d->Am[i] < lo) lo = (const unsigned short*)d->Am[i];
        a.am_base = lo;
        for (int i = 0; i < d->n_inst; ++i) {
            const long off = (const unsigned short*)d->Am[i] - lo;
            if (off < 0 || off > 0x3fffffffL) return SREC_BAD_ARG;
            a.am_off[i] = (int)off;
        }
    }
    const size_t lds = (size_t)LDS_ELEMS * sizeof(unsigned short);    // 48 KB: two stages; the epilogue patches reuse it
    hipLaunchKernelGGL(hg_wgrad_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, a);
    SREC_LAUNCH_CHECK();
    return 0;
}
