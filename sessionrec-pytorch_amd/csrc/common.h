// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32 -> bf16, round to nearest even, with the gfx950 conversion instruction (v_cvt_pk_bf16_f32: one VALU op per PAIR
// instead of ~6 integer ops per element - operand staging in the MFMA kernels is VALU-bound without it)
typedef __bf16 srec_bf16x2 __attribute__((ext_vector_type(2)));
typedef float srec_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned srec_pack_bf16(float lo, float hi) {
    const srec_f32x2 v = {lo, hi};
    const srec_bf16x2 r = __builtin_convertvector(v, srec_bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned short srec_f2bf(float a) { return (unsigned short)(srec_pack_bf16(a, 0.f) & 0xffffu); }

#define SREC_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

#define SREC_BAD_ARG 1001

// Static LDS budgets of the one-wavefront-per-session / per-destination kernels (srec_limits): the host checks a batch
// against them when it is collated (collate.py) and before a model's forward (ops.check_limits), so an oversized session
// is a clean error instead of a silently truncated reduction.  Sessions of <= 50 clicks (config C5) at order 3 have at
// most 50 + 49 + 48 = 147 read-out nodes and degree <= 50.
#define SREC_MAX_SESSION_NODES 256     // nodes of ONE session in a read-out (all n-gram orders of MSGIFSR concatenated)
#define SREC_MAX_DEGREE 128            // in- / out-degree of a node in one relation (GAT kernels)
#define SREC_MAX_DEGREE_SGAT 256       // in-degree in LESSR's shortcut graph (SGAT)

// > 64 KiB of dynamic LDS needs hipFuncSetAttribute once PER DEVICE (function attributes are per device; one process
// may drive several GPUs): `done` is a per-kernel bit mask indexed by the current device ordinal.
#include <atomic>
static inline int srec_lds_optin(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
static __device__ __forceinline__ int dyn_count(const int* dyn, int n_static) {
    if (dyn == nullptr) return n_static;
    int v = *dyn;
    return v < n_static ? v : n_static;
}
static __device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- dropout masks without a generator state: a counter-based hash.  key = f(host seed, per-step device counter, call-site
// salt); element i of the masked tensor keeps its value (scaled by 1 / (1 - p)) iff hash(key, i) / 2^24 >= p.  The forward
// and the backward of a step recompute the same mask from the same (key, i) - nothing is stored, nothing is exchanged with
// the framework's Philox state (whose two per-replay fill kernels a captured step would otherwise carry).  The step counter
// is the optimizer's device-side step count (NULL: 0): constant inside a step, different in the next, replay-safe.
struct srec_rng { unsigned seed; const int* counter; unsigned salt; float p; };
__device__ __forceinline__ unsigned srec_hash32(unsigned x) {          // "lowbias32" integer finaliser
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned srec_rng_key(const srec_rng& r) {
    const unsigned c = r.counter != nullptr ? (unsigned)*r.counter : 0u;
    return srec_hash32(r.seed ^ srec_hash32(c * 0x9E3779B9u + r.salt * 0x85EBCA6Bu + 0x165667B1u));
}
__device__ __forceinline__ float srec_keep(unsigned key, unsigned idx, float p, float scale) {
    const unsigned h = srec_hash32(key ^ (idx * 0x9E3779B1u + 0x7F4A7C15u));
    return (float)(h >> 8) * (1.f / 16777216.f) >= p ? scale : 0.f;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
