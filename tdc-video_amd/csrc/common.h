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

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <class T> struct VecOf;
template <> struct VecOf<f16> { typedef f16x8 v8; typedef f16x4 v4; typedef f16x2 v2; };
template <> struct VecOf<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; typedef bf16x2 v2; };

// fp32 -> 16-bit (round to nearest even, the same result as an elementwise (T) cast) on explicit PAIRS: one
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 per two values.  Written as four scalar casts, plain -O3 pairs elements 1 and 2 and
// converts 0 and 3 alone, then reassembles the registers with v_pk_mov / v_perm / v_alignbit: 10 VALU instructions per four
// values (with the bias add) instead of 4 - in the GEMM epilogues, where the MFMA pipe idles, that was most of the work.
template <class T>
__device__ __forceinline__ typename VecOf<T>::v2 cvt2(float a, float b) {
    return __builtin_convertvector((f32x2){a, b}, typename VecOf<T>::v2);
}
template <class T>
__device__ __forceinline__ typename VecOf<T>::v4 cvt4(f32x4 v) {
    const u32x2 r = {__builtin_bit_cast(unsigned, cvt2<T>(v[0], v[1])), __builtin_bit_cast(unsigned, cvt2<T>(v[2], v[3]))};
    return __builtin_bit_cast(typename VecOf<T>::v4, r);
}

// D(16x16 f32) += A(16x32) * B(32x16); lane l: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15], j=0..7;
// D[row 4(l>>4)+reg][col l&15].
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// fp8 (OCP e4m3) operands: the same 16-byte fragment holds 16 values = two K=32 steps of v_mfma_f32_16x16x32_fp8_fp8
// (lane l: bytes 8 h .. 8 h + 7 of its fragment are k-slot (l >> 4) of step h).  A and W fragments are cut identically, so
// the contraction pairs the right bytes whatever the order of K inside the 128-byte row.
typedef __attribute__((ext_vector_type(2))) long i64x2;
template <bool FP8, class V8>
__device__ __forceinline__ f32x4 mma16(V8 a, V8 b, f32x4 c) {
    if constexpr (FP8) {
        const i64x2 a2 = __builtin_bit_cast(i64x2, a), b2 = __builtin_bit_cast(i64x2, b);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[0], b2[0], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[1], b2[1], c, 0, 0, 0);
    } else {
        return mfma16(a, b, c);
    }
}

// ... and both 16-byte fragments of a 128-byte K tile (32 e4m3 values per lane) feed ONE v_mfma_f32_16x16x128_f8f6f4
// (unit block scales): 32 cycles for four times the K of the 16-cycle bf16 / fp8 16x16x32 forms - twice their rate.
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <class V8>
__device__ __forceinline__ f32x4 mma128_fp8(V8 a0, V8 a1, V8 b0, V8 b1, f32x4 c) {
    const u32x4 x0 = __builtin_bit_cast(u32x4, a0), x1 = __builtin_bit_cast(u32x4, a1);
    const u32x4 y0 = __builtin_bit_cast(u32x4, b0), y1 = __builtin_bit_cast(u32x4, b1);
    const i32x8 a = {(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
    const i32x8 b = {(int)y0[0], (int)y0[1], (int)y0[2], (int)y0[3], (int)y1[0], (int)y1[1], (int)y1[2], (int)y1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

// Activations for GEMM epilogues.  They run once per output element inside an MFMA-bound kernel, so they are written
// with the two quarter-rate instructions v_exp_f32 / v_rcp_f32 and a handful of FMAs instead of libm calls
// (erff / tanhf cost 30-40 VALU instructions per element and dominated the fc1 epilogues).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf(x) by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. below fp32 round-off of the surrounding math)
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = fast_rcp(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = fast_exp2(-1.4426950408889634f * ax * ax);
    const float r = __builtin_fmaf(-poly, e, 1.0f);
    return copysignf(r, x);
}
// nn.GELU() / ACT2FN["gelu"]: 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// "gelu_pytorch_tanh": 0.5 x (1 + tanh(u)) == x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)   (exact identity),
// as x / (1 + 2^(x (c0 + c1 x^2))).  The GEMM epilogues are VALU-bound on this, so the arithmetic around the two
// transcendentals is written on float pairs (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_tanh2(f32x2_t x) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
    const float c1 = c0 * 0.044715f;
    const f32x2_t z = x * __builtin_elementwise_fma(x * x, (f32x2_t){c1, c1}, (f32x2_t){c0, c0});
    const f32x2_t d = (f32x2_t){fast_exp2(z[0]), fast_exp2(z[1])} + 1.0f;
    return x * (f32x2_t){fast_rcp(d[0]), fast_rcp(d[1])};
}
__device__ __forceinline__ float gelu_tanh(float x) { return gelu_tanh2((f32x2_t){x, x})[0]; }
__device__ __forceinline__ f32x4 gelu_tanh4(f32x4 v) {
    const f32x2_t a = gelu_tanh2((f32x2_t){v[0], v[1]}), b = gelu_tanh2((f32x2_t){v[2], v[3]});
    return (f32x4){a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ float silu(float x) { return x * fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * x)); }
// SwiGLU on interleaved (gate, up) columns: (silu(v0) v1, silu(v2) v3)
__device__ __forceinline__ f32x2_t swiglu2(f32x4 v) {
    const f32x2_t gate = {v[0], v[2]}, up = {v[1], v[3]};
    const f32x2_t z = gate * -1.4426950408889634f;
    const f32x2_t d = (f32x2_t){fast_exp2(z[0]), fast_exp2(z[1])} + 1.0f;
    return gate * (f32x2_t){fast_rcp(d[0]), fast_rcp(d[1])} * up;
}

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
// The quotient comes from a host-computed reciprocal (inv = floor(2^32 / seg), a kernel argument = an SGPR) and two
// corrections: a plain `m / seg` expands into a float-reciprocal sequence whose seg-only part the compiler hoists to kernel
// entry as a VGPR that then lives through the whole kernel - in the persistent GEMM that register was spilled and its
// scratch reload at the tile seam waited (in-order vmcnt) for every staged load in flight.
struct RowMap {
    int seg, stride, off, inner;
    unsigned inv;
    __host__ static RowMap make(int seg, int stride, int off, int inner) {
        RowMap r;
        r.seg = seg; r.stride = stride; r.off = off; r.inner = inner;
        r.inv = seg > 0 ? (unsigned)((1ull << 32) / (unsigned long long)seg > 0xFFFFFFFFull ? 0xFFFFFFFFull
                                                                                             : (1ull << 32) / (unsigned long long)seg)
                        : 0u;
        return r;
    }
    __device__ __forceinline__ long long operator()(int m) const {        // m >= 0
        if (seg == 0) return m;
        unsigned q = __umulhi((unsigned)m, inv);                            // q_true - 2 <= q <= q_true
        unsigned r = (unsigned)m - q * (unsigned)seg;
        if (r >= (unsigned)seg) { q += 1; r -= (unsigned)seg; }
        if (r >= (unsigned)seg) { q += 1; r -= (unsigned)seg; }
        return (long long)q * stride + off + (long long)r * inner;
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
