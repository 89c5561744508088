// Which SIMD does wave w of a 512-thread (and a 256-thread) workgroup run on?  HW_REG_HW_ID bits 5:4 = SIMD id, 3:0 = wave slot.
// hipcc -O2 --offload-arch=gfx950 tools/wave_simd.cpp -o gpurun_out/wave_simd && gpurun_out/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned id = __builtin_amdgcn_s_getreg((5 << 11) | (0 << 6) | 4);   // bits 5:0 of HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    for (int threads : {512, 256}) {
        hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
        unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const int nw = threads / 64;
        for (int b = 0; b < 4; ++b) {
            printf("%d threads, workgroup %d: wave -> SIMD:", threads, b);
            for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3);
            printf("   (wave slot:");
            for (int w = 0; w < nw; ++w) printf(" %u", h[b * nw + w] & 15);
            printf(")\n");
        }
    }
    return 0;
}
