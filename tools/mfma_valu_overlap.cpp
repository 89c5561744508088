// Do the matrix pipe and the vector ALU of ONE SIMD overlap when the MFMAs come from one wave and the VALU work from the other?
// 256 workgroups x 512 threads, one per CU (96 KiB of LDS requested): waves 0-3 (one per SIMD) run a chain of
// v_mfma_f32_32x32x16_bf16 (accumulators in arch VGPRs or in AGPRs), waves 4-7 (their SIMD partners) a stream of one VALU
// instruction kind.  Times: MFMA waves alone, VALU waves alone, both - overlap = both ~ max, none = both ~ sum.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_valu_overlap.cpp -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int ACC, int VK, int PRIO>
__global__ __launch_bounds__(512, 2) void k(float* out, int n_mfma, int n_valu, int run_mfma, int run_valu) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!run_mfma) return;
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);     // the MFMA waves above
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < n_mfma; ++i) {
            if (ACC) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                             "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                             : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                             "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
            }
        }
        float s = 0.f;
        for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
        if (s == 12345.f) out[threadIdx.x] = s;
    } else {
        if (!run_valu) return;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);     // the VALU waves (the younger half) above the MFMA waves
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
        f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        for (int i = 0; i < n_valu; ++i) {
            if (VK == 0) {        // v_exp_f32 x 8
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                             "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (VK == 1) { // v_fma_f32 x 8
                asm volatile("v_fma_f32 %0, %0, %0, %1\n\tv_fma_f32 %1, %1, %1, %2\n\tv_fma_f32 %2, %2, %2, %3\n\tv_fma_f32 %3, %3, %3, %4\n\t"
                             "v_fma_f32 %4, %4, %4, %5\n\tv_fma_f32 %5, %5, %5, %6\n\tv_fma_f32 %6, %6, %6, %7\n\tv_fma_f32 %7, %7, %7, %0"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (VK == 2) { // v_pk_fma_f32 x 4 (8 values)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n\tv_pk_fma_f32 %1, %1, %1, %2\n\tv_pk_fma_f32 %2, %2, %2, %3\n\tv_pk_fma_f32 %3, %3, %3, %0"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            } else if (VK == 3) { // v_cvt_pk_bf16_f32 x 4 + v_max3_f32 x 4
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_cvt_pk_bf16_f32 %4, %4, %5\n\tv_cvt_pk_bf16_f32 %6, %6, %7\n\t"
                             "v_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %3, %3, %4, %5\n\tv_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %7, %7, %0, %1"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
        const float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
        if (s == 12345.f) out[threadIdx.x] = s;
    }
}

template <int ACC, int VK, int PRIO>
void run(const char* name, float* d, int n_mfma, int n_valu) {
    hipFuncSetAttribute((const void*)k<ACC, VK, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
        const int rm = mode != 1, rv = mode != 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<ACC, VK, PRIO>), dim3(256), dim3(512), 96 * 1024, 0, d, n_mfma, n_valu, rm, rv);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<ACC, VK, PRIO>), dim3(256), dim3(512), 96 * 1024, 0, d, n_mfma, n_valu, rm, rv);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1); ms[mode] /= 5;
    }
    printf("%-52s MFMA alone %.3f ms | VALU alone %.3f ms | both %.3f ms  (max %.3f, sum %.3f)\n", name, ms[0], ms[1], ms[2],
           ms[0] > ms[1] ? ms[0] : ms[1], ms[0] + ms[1]);
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int NM = 20000;    // x 4 MFMAs x 32 cycles = 2.56 M cycles
    run<0, 0, 0>("C in VGPRs | v_exp_f32", d, NM, 20000);            // 8 x 20000 trans
    run<1, 0, 0>("C in AGPRs | v_exp_f32", d, NM, 20000);
    run<0, 0, 1>("C in VGPRs | v_exp_f32 | VALU waves prio 3", d, NM, 20000);
    run<0, 0, 2>("C in VGPRs | v_exp_f32 | MFMA waves prio 3", d, NM, 20000);
    run<0, 1, 0>("C in VGPRs | v_fma_f32", d, NM, 40000);
    run<0, 1, 1>("C in VGPRs | v_fma_f32 | VALU waves prio 3", d, NM, 40000);
    run<1, 1, 1>("C in AGPRs | v_fma_f32 | VALU waves prio 3", d, NM, 40000);
    run<0, 2, 0>("C in VGPRs | v_pk_fma_f32", d, NM, 40000);
    run<0, 2, 1>("C in VGPRs | v_pk_fma_f32 | VALU waves prio 3", d, NM, 40000);
    run<0, 3, 0>("C in VGPRs | v_cvt_pk_bf16_f32 + v_max3_f32", d, NM, 40000);
    run<0, 3, 1>("C in VGPRs | cvt_pk + max3 | VALU waves prio 3", d, NM, 40000);
    return 0;
}
