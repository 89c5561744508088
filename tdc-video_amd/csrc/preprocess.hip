// Frame pre-processing on the device: the reference's `process_images` for one tower
// (tdc/mm_datautils.py:286-314): expand2square(mean colour) -> PIL Image.resize((R, R)) (bicubic, antialiased, 8 bits
// per channel) -> rescale 1/255 -> normalize -> 16-bit NCHW.  Byte-exact with Pillow: same 22-bit fixed-point
// coefficients (computed on the host exactly as Resample.c does), same pass order (horizontal, then vertical), same uint8
// intermediate and clipping.  HBM-bound byte work: one thread per output pixel, all three channels.
#include "common.h"
#include "../../include/tdc_hip.h"
#include <stdio.h>

namespace {

constexpr int PREC = 22;

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PREC;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass over the virtual S x S padded square: out [T][S][R][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* frames, int H, int W, int S, int R,
                                                       const int* bounds, const int* coeffs, int ksize, int pr, int pg,
                                                       int pb, unsigned char* out) {
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, t = blockIdx.z;
    if (xo >= R) return;
    const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
    const int* k = coeffs + (long long)xo * ksize;
    const int y0 = W > H ? (W - H) / 2 : 0, x0 = H > W ? (H - W) / 2 : 0;
    const int fy = y - y0;
    const bool row_in = fy >= 0 && fy < H;
    const unsigned char* row = frames + ((long long)t * H + (row_in ? fy : 0)) * W * 3;
    int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
    for (int i = 0; i < n; ++i) {
        const int fx = xmin + i - x0;
        int r = pr, g = pg, b = pb;
        if (row_in && fx >= 0 && fx < W) { r = row[fx * 3]; g = row[fx * 3 + 1]; b = row[fx * 3 + 2]; }
        const int c = k[i];
        s0 += r * c; s1 += g * c; s2 += b * c;
    }
    unsigned char* o = out + (((long long)t * S + y) * R + xo) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// vertical pass + rescale/normalize (per-channel 256-entry fp32 table) -> out [T][3][R][R]
template <class T>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const unsigned char* mid, int S, int R, const int* bounds,
                                                            const int* coeffs, int ksize, const float* lut, void* out,
                                                            int out_f32) {
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int yo = blockIdx.y, t = blockIdx.z;
    if (xo >= R) return;
    const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
    const int* k = coeffs + (long long)yo * ksize;
    int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
    const unsigned char* col = mid + (((long long)t * S + ymin) * R + xo) * 3;
    for (int i = 0; i < n; ++i) {
        const int c = k[i];
        s0 += col[0] * c; s1 += col[1] * c; s2 += col[2] * c;
        col += (long long)R * 3;
    }
    const float v0 = lut[clip8(s0)], v1 = lut[256 + clip8(s1)], v2 = lut[512 + clip8(s2)];
    const long long plane = (long long)R * R, base = (long long)t * 3 * plane + (long long)yo * R + xo;
    if (out_f32) {
        float* o = (float*)out;
        o[base] = v0; o[base + plane] = v1; o[base + 2 * plane] = v2;
    } else {
        T* o = (T*)out;
        o[base] = (T)v0; o[base + plane] = (T)v1; o[base + 2 * plane] = (T)v2;
    }
}

// no resize needed (S == R): pad to square + normalize
template <class T>
__global__ __launch_bounds__(256) void pad_norm_kernel(const unsigned char* frames, int H, int W, int S, int pr, int pg,
                                                       int pb, const float* lut, void* out, int out_f32) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, t = blockIdx.z;
    if (x >= S) return;
    const int y0 = W > H ? (W - H) / 2 : 0, x0 = H > W ? (H - W) / 2 : 0;
    const int fy = y - y0, fx = x - x0;
    int r = pr, g = pg, b = pb;
    if (fy >= 0 && fy < H && fx >= 0 && fx < W) {
        const unsigned char* p = frames + (((long long)t * H + fy) * W + fx) * 3;
        r = p[0]; g = p[1]; b = p[2];
    }
    const float v0 = lut[r], v1 = lut[256 + g], v2 = lut[512 + b];
    const long long plane = (long long)S * S, base = (long long)t * 3 * plane + (long long)y * S + x;
    if (out_f32) {
        float* o = (float*)out;
        o[base] = v0; o[base + plane] = v1; o[base + 2 * plane] = v2;
    } else {
        T* o = (T*)out;
        o[base] = (T)v0; o[base + plane] = (T)v1; o[base + 2 * plane] = (T)v2;
    }
}

}  // namespace

extern "C" size_t tdc_preprocess_scratch_bytes(int T, int H, int W, int R) {
    const int S = H > W ? H : W;
    return S == R ? 0 : (size_t)T * S * R * 3;
}

extern "C" int tdc_preprocess_frames(const unsigned char* frames, int T, int H, int W, int R, const int* bounds,
                                     const int* coeffs, int ksize, int pad_r, int pad_g, int pad_b, const float* lut,
                                     void* out, int out_f32, int dtype, unsigned char* scratch, void* stream) {
    if (!frames || !out || !lut || T <= 0 || H <= 0 || W <= 0 || R <= 0 || T > 65535) return TDC_E_BADARG;
    const int S = H > W ? H : W;
    if (S > 65535 || R > 65535) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (S == R) {   // PIL returns a copy when the size does not change
        dim3 grid((S + 255) / 256, S, T);
        if (dtype == TDC_F16) hipLaunchKernelGGL(pad_norm_kernel<f16>, grid, dim3(256), 0, st, frames, H, W, S, pad_r, pad_g, pad_b, lut, out, out_f32);
        else if (dtype == TDC_BF16) hipLaunchKernelGGL(pad_norm_kernel<bf16>, grid, dim3(256), 0, st, frames, H, W, S, pad_r, pad_g, pad_b, lut, out, out_f32);
        else return TDC_E_BADARG;
        return (int)hipGetLastError();
    }
    if (!bounds || !coeffs || !scratch || ksize <= 0) return TDC_E_BADARG;
    hipLaunchKernelGGL(resize_h_kernel, dim3((R + 255) / 256, S, T), dim3(256), 0, st, frames, H, W, S, R, bounds, coeffs,
                       ksize, pad_r, pad_g, pad_b, scratch);
    dim3 grid((R + 255) / 256, R, T);
    if (dtype == TDC_F16) hipLaunchKernelGGL(resize_v_norm_kernel<f16>, grid, dim3(256), 0, st, scratch, S, R, bounds, coeffs, ksize, lut, out, out_f32);
    else if (dtype == TDC_BF16) hipLaunchKernelGGL(resize_v_norm_kernel<bf16>, grid, dim3(256), 0, st, scratch, S, R, bounds, coeffs, ksize, lut, out, out_f32);
    else return TDC_E_BADARG;
    return (int)hipGetLastError();
}
