// The optimizer's step-scalar role (adam.hip: srec_adam_hyper_multi) as a body other launches can carry: one thread per (counter,
// cfg, hyper) slot advances the DEVICE step counter and derives the step's Adam scalars in double (like torch on the host).  Keeping
// the counter on the device makes the optimizer step a pure function of device state: a replayed hipGraph (or a host running ahead of
// the GPU) cannot pair a step with another step's bias corrections.  Carried by adam_hyper_multi_kernel and - in a captured training
// step - by the end-of-backward slab-sum launch (grux.hip: srec_sum_slabs_multi_hyper), which is otherwise followed by a launch of
// its own for these <= 16 threads (~6 us of a graph node).
#pragma once
#include "common.h"

namespace {
constexpr int HYPER_MAX = 16;
struct HyperArgs {
    int* counter[HYPER_MAX]; const double* cfg[HYPER_MAX]; float* hyper[HYPER_MAX]; int n;
    const int* tap_counter; const float* tap_src; float* tap_ring; int tap_n;
    const int* skip;
};

__device__ __forceinline__ void hyper_role(const HyperArgs& a, const int i) {
    if (i >= a.n) return;
    const int t = *a.counter[i] + 1;
    *a.counter[i] = t;
    // loss tap: the scalar of THIS step (the training loss) into slot (steps taken before it) % tap_n of a device ring - a
    // replayed step overwrites its loss tensor, and a host that runs ahead would otherwise clone it after every replay
    // *skip != 0 (the batch intake of this step found another step count in its mailbox entry than the host wrote: the step ran
    // on a STALE batch): the counters still advance - the device stays in step with the host's numbering - but the scalars make
    // every Adam kernel of the step the identity (step size 0, exp_avg / exp_avg_sq kept: m + 0 (g - m), 1 v + 0 g g), and the
    // loss slot reads NaN.  The flag is sticky; the host raises at its next readback (graph.GraphedTrainStep.check).
    const bool skip = a.skip != nullptr && *a.skip != 0;
    if (a.tap_ring != nullptr && a.counter[i] == a.tap_counter) a.tap_ring[(t - 1) % a.tap_n] = skip ? __int_as_float(0x7fc00000) : *a.tap_src;
    const double* cfg = a.cfg[i];
    float* hyper = a.hyper[i];
    const double lr = cfg[0], b1 = cfg[1], b2 = cfg[2], eps = cfg[3], wd = cfg[4];
    if (skip) {
        hyper[0] = 0.f; hyper[1] = 1.f; hyper[2] = 1.f; hyper[3] = (float)eps; hyper[4] = 0.f; hyper[5] = 0.f; hyper[6] = 0.f;
        hyper[7] = 1.f;
        return;
    }
    hyper[0] = (float)(lr / (1.0 - pow(b1, (double)t)));
    hyper[1] = (float)b1; hyper[2] = (float)b2; hyper[3] = (float)eps; hyper[4] = (float)wd;
    hyper[5] = (float)(1.0 - b1); hyper[6] = (float)(1.0 - b2);
    hyper[7] = (float)sqrt(1.0 - pow(b2, (double)t));
}

// n <= 16 (counter, cfg, hyper) slots: HOST arrays of n device pointers; optional loss tap (all four or none); skip (nullable)
inline int hyper_fill(int n, const void* counter, const void* cfg, const void* hyper, const int* tap_counter, const float* tap_src,
                      float* tap_ring, int tap_n, const int* skip, HyperArgs& a) {
    a = HyperArgs{};
    if (n <= 0) return 0;
    if (n > HYPER_MAX || counter == nullptr || cfg == nullptr || hyper == nullptr) return SREC_BAD_ARG;
    if (tap_ring != nullptr && (tap_counter == nullptr || tap_src == nullptr || tap_n <= 0)) return SREC_BAD_ARG;
    a.n = n;
    a.tap_counter = tap_counter; a.tap_src = tap_src; a.tap_ring = tap_ring; a.tap_n = tap_n; a.skip = skip;
    for (int i = 0; i < n; ++i) {
        a.counter[i] = ((int* const*)counter)[i];
        a.cfg[i] = ((const double* const*)cfg)[i];
        a.hyper[i] = ((float* const*)hyper)[i];
        if (a.counter[i] == nullptr || a.cfg[i] == nullptr || a.hyper[i] == nullptr) return SREC_BAD_ARG;
    }
    return 0;
}
}  // namespace
