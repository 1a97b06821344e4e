// Development probe: what does the end-of-kernel L2 write-back cost, and do store cache-policy bits (nt / sc0 sc1) move it?
// Each workgroup writes its slice of an N-byte buffer with 16-byte stores of one policy; timed as back-to-back launches (events).
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/spp tools/probes/store_policy_probe.hip && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void wr(f4* p, size_t n4, float x) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 v = {x, x + 1.f, x + 2.f, (float)i};
        if (MODE == 0) p[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, p + i);
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p + i), "v"(v) : "memory");
    }
}
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[0] += 1.f; }
template <int MODE>
float run(f4* p, size_t n4, int blocks, bool with_tiny, float* t) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wr<MODE>, dim3(blocks), dim3(256), 0, 0, p, n4, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) {
        hipLaunchKernelGGL(wr<MODE>, dim3(blocks), dim3(256), 0, 0, p, n4, (float)i);
        if (with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, t);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20.f * 1e3f;
}
int main() {
    float* t; hipMalloc(&t, 4); hipMemset(t, 0, 4);
    const size_t sizes[] = {(size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20};
    for (size_t N : sizes) {
        f4* p; hipMalloc(&p, N);
        const size_t n4 = N / 16;
        for (int blocks : {256, 2048}) {
            printf("%4zu MB, %4d workgroups: plain %.1f us, nt %.1f, sc0 sc1 %.1f, sc0 sc1 nt %.1f   | + tiny kernel: %.1f %.1f %.1f %.1f\n", N >> 20, blocks,
                   run<0>(p, n4, blocks, false, t), run<1>(p, n4, blocks, false, t), run<2>(p, n4, blocks, false, t), run<3>(p, n4, blocks, false, t),
                   run<0>(p, n4, blocks, true, t), run<1>(p, n4, blocks, true, t), run<2>(p, n4, blocks, true, t), run<3>(p, n4, blocks, true, t));
        }
        hipFree(p);
    }
    return 0;
}
