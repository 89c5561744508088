// Common device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef __bf16 bf16;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// dtype codes shared with include/tdc_hip.h
#define TDC_F16 0
#define TDC_BF16 1

template <class T> struct VecOf;
template <> struct VecOf<f16> { typedef f16x8 v8; typedef f16x4 v4; };
template <> struct VecOf<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };

// D(16x16 f32) += A(16x32) * B(32x16); lane l: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15], j=0..7;
// D[row 4(l>>4)+reg][col l&15].
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_tanh(float x) {
    const float k = 0.7978845608028654f;  // sqrt(2/pi)
    float u = k * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// generic row map: row(m) = (m / seg) * stride + off + (m % seg) * inner ; identity when seg == 0
struct RowMap {
    int seg, stride, off, inner;
    __device__ __forceinline__ long long operator()(int m) const {
        if (seg == 0) return m;
        int q = m / seg;
        return (long long)q * stride + off + (long long)(m - q * seg) * inner;
    }
};

#define HIP_CHECK_RET(x)                                                                      \
    do {                                                                                      \
        hipError_t _e = (x);                                                                  \
        if (_e != hipSuccess) {                                                               \
            fprintf(stderr, "[tdc_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e),   \
                    __FILE__, __LINE__);                                                      \
            return (int)_e;                                                                   \
        }                                                                                     \
    } while (0)
