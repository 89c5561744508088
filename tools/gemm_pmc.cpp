// Standalone (torch-free) driver of tdc_gemm for rocprofv3 --pmc runs: rocprofv3's counter collection segfaults inside
// a torch process on this image (torch bundles ROCm 7.0 HSA, the profiler is 7.2), so the HBM-traffic counters of the
// dominant kernel are collected here on the exact GEMM shapes bench.py launches (tools/gemm_shapes_T512.txt, written by
// `python bench.py --dump-gemm-shapes ...`).  Build: see tools/run_gemm_pmc.sh.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include "../include/tdc_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// operands are filled ON THE DEVICE (the host LCG + a 3-GB copy per shape took minutes per pass; bench.py runs four passes of
// this replay inside its own run): bf16 values uniform in [-scale, scale], a hash of the element index
__global__ void fill_bf16_kernel(unsigned short* v, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)i * 2654435761u + seed * 40503u + (unsigned)(i >> 32) * 97u;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
        const float f = ((int)(s >> 9) % 2001 - 1000) * 0.001f * scale;
        unsigned u = __float_as_uint(f);
        v[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
}
__global__ void fill_fp8_kernel(unsigned char* v, size_t n, unsigned seed) {   // random e4m3 bytes without the NaN encodings
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)i * 2654435761u + seed * 40503u;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
        unsigned char b = (unsigned char)(s >> 11);
        if ((b & 0x7f) == 0x7f) b &= 0xf7;
        v[i] = b;
    }
}

int main(int argc, char** argv) {
    if (const char* e = getenv("TDC_GEMM_DEBUG")) tdc_gemm_set_debug(atoi(e));   // the tool, not the library, reads the environment
    if (argc < 2) { fprintf(stderr, "usage: gemm_pmc shapes.txt [reps]\n"); return 2; }
    int reps = argc > 2 ? atoi(argv[2]) : 2;
    const bool fp8 = getenv("TDC_PMC_FP8") != nullptr;      // e4m3 operands (K % 128 == 0 shapes only; others are skipped)
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror("shapes"); return 2; }
    int M, N, K, act, res, outf32, count;
    hipStream_t st; CK(hipStreamCreate(&st));
    while (fscanf(f, "%d %d %d %d %d %d %d", &M, &N, &K, &act, &res, &outf32, &count) == 7) {
        if (fp8 && (K % 128 != 0 || (outf32 && !res) || act == TDC_ACT_GELU_ERF)) {
            printf("skip M=%d N=%d K=%d (not an fp8 tower shape)\n", M, N, K);
            continue;
        }
        const bool pad8 = (N % 8 == 4) && !res && !outf32 && act == TDC_ACT_NONE && !fp8;   // the transposed value projection of the Q-Former
        const int ldc_full = pad8 ? (N + 7) / 8 * 8 : N;
        size_t nA = (size_t)M * K, nW = (size_t)N * K, nC = (size_t)M * ldc_full;
        void *A, *W, *C; float* bias;
        CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&W, nW * 2)); CK(hipMalloc(&C, nC * (outf32 ? 4 : 2))); CK(hipMalloc((void**)&bias, N * 4));
        if (fp8) {
            hipLaunchKernelGGL(fill_fp8_kernel, dim3(2048), dim3(256), 0, st, (unsigned char*)A, nA * 2, (unsigned)(M + K));
            hipLaunchKernelGGL(fill_fp8_kernel, dim3(2048), dim3(256), 0, st, (unsigned char*)W, nW * 2, (unsigned)(N + K));
        } else {
            hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, st, (unsigned short*)A, nA, (unsigned)(M + K), 1.0f);
            hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, st, (unsigned short*)W, nW, (unsigned)(N + K), 0.05f);
        }
        CK(hipStreamSynchronize(st));
        CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(C, 0, nC * (outf32 ? 4 : 2)));
        tdc_gemm_desc d = {};
        d.A = A; d.lda = K; d.W = W; d.ldw = K; d.C = C; d.ldc = (act == TDC_ACT_SWIGLU) ? N / 2 : ldc_full; d.bias = pad8 ? nullptr : bias;
        d.c_pad8 = pad8 ? 1 : 0;
        d.M = M; d.N = N; d.K = K; d.dtype = TDC_BF16; d.out_f32 = outf32; d.act = act;
        if (res) { d.res = C; d.ldres = N; d.res_f32 = outf32; }   // in-place residual stream update
        if (res == 2) d.c16_dtype_p1 = TDC_F16 + 1;                // the towers' fp16 residual stream beside bf16 operands
        float *stats = nullptr, *c1 = nullptr;
        if (fp8) {
            std::vector<float> hs((size_t)M * 2);
            for (int m = 0; m < M; ++m) { hs[2 * m] = 0.f; hs[2 * m + 1] = 1e-4f; }
            CK(hipMalloc((void**)&stats, (size_t)M * 8)); CK(hipMalloc((void**)&c1, (size_t)N * 4));
            CK(hipMemcpy(stats, hs.data(), (size_t)M * 8, hipMemcpyHostToDevice)); CK(hipMemset(c1, 0, (size_t)N * 4));
            d.in_fp8 = 1; d.ln_stats = stats; d.ln_c1 = c1;
        }
        for (int r = 0; r < reps; ++r) { int rc = tdc_gemm(&d, st); if (rc) { fprintf(stderr, "tdc_gemm rc=%d\n", rc); return 1; } }
        CK(hipStreamSynchronize(st));
        printf("ran M=%d N=%d K=%d act=%d res=%d outf32=%d x%d\n", M, N, K, act, res, outf32, reps);
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(bias));
        if (stats) { CK(hipFree(stats)); CK(hipFree(c1)); }
    }
    return 0;
}
