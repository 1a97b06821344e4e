// probe of ds_read_b64_tr_b16 lane mapping on gfx950: LDS holds element id = index (u16); every lane reads 8 B at a
// lane-specific address; prints what each lane received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr_bytes, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + (unsigned)addr_bytes[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    // pattern 1: row-major tile [rows][64 cols] (128 B per row); lane i of each 16-lane group g: row = (i >> 2) + 4*?; let
    // lane l read row (l & 15) >> 2 ... we just try: lane l -> row = (l%16)/4 + 4*(l/16), col chunk = (l%4)*4
    for (int variant = 0; variant < 2; ++variant) {
        for (int l = 0; l < 64; ++l) {
            int i = l % 16, g = l / 16;
            int row, col;
            if (variant == 0) { row = i / 4 + 4 * g; col = (i % 4) * 4; }
            else { row = i % 4 + 4 * g; col = (i / 4) * 4; }
            h_addr[l] = (row * 64 + col) * 2;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("variant %d (element id = row*64+col)\n", variant);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d addr(r%d,c%2d):", l, h_addr[l] / 128, (h_addr[l] % 128) / 2);
            for (int e = 0; e < 4; ++e) printf(" (r%d,c%2d)", h_out[l * 4 + e] / 64, h_out[l * 4 + e] % 64);
            printf("\n");
        }
    }
    return 0;
}
