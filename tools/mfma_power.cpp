// What the matrix pipes sustain at the chip's power limit: back-to-back MFMAs on register operands (no memory traffic at all),
// every CU busy, dense random vs all-zero operands, both bf16 shapes.  hipcc -O3 --offload-arch=gfx950 tools/mfma_power.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const bf16x8* in, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xFFFF]; b[i] = in[(t * 8 + 4 + i) & 0xFFFF]; }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) s += c[i][0];
    } else {
        f32x16 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i + 2 * r) & 3], c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += c[i][0];
    }
    if (s == 123.456f) out[t] = s;
}

int main(int argc, char** argv) {
    const int n = 1 << 16;
    std::vector<unsigned short> h(n * 8);
    bf16x8* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = argc > 1 ? atoi(argv[1]) : 40000;     // 40000: ~20 ms per launch
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (auto& x : h) x = zero ? 0 : (unsigned short)((rand() & 0x807F) | ((120 + rand() % 12) << 7));   // random sign / mantissa, exponents near 1
        hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
        for (int shape : {16, 32}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 0, 0, d, o, iters / 2 * 2);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // per wave and iteration: 16 x (16x16x32) = 16 x 16384 FLOP, or 8 x (32x32x16) = 8 x 32768 FLOP: the same
                const double fl = 2.0 * 16 * 16 * 16 * 32 * (double)iters * blocks * 8;
                if (rep) printf("%s operands, %dx%dx%d: %8.2f ms  %7.1f TFLOP/s\n", zero ? "all-zero" : "dense random", shape, shape,
                                shape == 16 ? 32 : 16, ms, fl / ms / 1e9);
            }
        }
    }
    return 0;
}
