// BEATs audio front end on the device: kaldi fbank (tdc/audio_models/beats/BEATs.py:115-129 ->
// torchaudio.compliance.kaldi.fbank) and the gate of the gated relative position bias
// (tdc/audio_models/beats/backbone.py:652-657).  Both are tiny HBM/latency-bound fp32 kernels; the encoder's matmuls
// run on tdc_gemm, its attention on tdc_attention (bias variant).
//
// fbank: one workgroup (256 threads) per 25-ms frame.  400 samples -> DC removal -> pre-emphasis -> povey window ->
// zero pad to 512 -> radix-2 FFT in LDS (fp32, host-built twiddles) -> power spectrum -> 128 triangular mel bins
// (host-built weights, each bin only walks its own FFT-bin range) -> log(max(., eps)) -> (x - mean) / (2 std).
// The result is written in the layout the patch-embedding GEMM consumes: token (frame/16)*8 + mel/16, column
// (frame%16)*16 + mel%16 (the 16x16 stride-16 conv of BEATs.py:147 as a GEMM), and optionally as plain fp32
// [frames, 128].
#include "common.h"
#include "../../include/tdc_hip.h"
#include <stdio.h>

namespace {

constexpr int FB_WIN = 400, FB_SHIFT = 160, FB_PAD = 512, FB_MEL = 128, FB_BINS = FB_PAD / 2 + 1;

struct FbankArgs {
    const void* wav; int wav_f32; long long wav_bs;
    int frames;                     // frames per item
    const float* window;            // [400]
    const float* tw;                // [256][2] cos, sin of 2 pi k / 512
    const float* banks;             // [128][257]
    const int* range;               // [128][2] first / one-past-last non-zero FFT bin
    float* plain;                   // [B, frames, 128] or NULL
    void* patches; int ldp;         // 16-bit [B, (frames/16)*8, ldp] or NULL
    int dtype;
    float mean, inv;                // (x - mean) * inv
    float preemph;
};

__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs p) {
    __shared__ float re[FB_PAD], im[FB_PAD], red[4];
    const int tid = threadIdx.x, frame = blockIdx.x, b = blockIdx.y;
    const long long base = (long long)b * p.wav_bs + (long long)frame * FB_SHIFT;
    float x[2];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tid + r * 256;
        float v = 0.f;
        if (i < FB_WIN) v = (p.wav_f32 ? ((const float*)p.wav)[base + i] : (float)((const f16*)p.wav)[base + i]) * 32768.0f;
        x[r] = v;
        s += v;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) * (1.0f / FB_WIN);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tid + r * 256;
        if (i < FB_WIN) re[i] = x[r] - mu;
    }
    __syncthreads();
    float y[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tid + r * 256;
        y[r] = 0.f;
        if (i < FB_WIN) y[r] = (re[i] - p.preemph * re[i > 0 ? i - 1 : 0]) * p.window[i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tid + r * 256;
        const int j = (int)(__brev((unsigned)i) >> 23);   // 9-bit reversal
        re[j] = y[r];
        im[j] = 0.f;
    }
    __syncthreads();
    // radix-2 decimation in time, 9 stages, 256 butterflies each
#pragma unroll
    for (int st = 0; st < 9; ++st) {
        const int half = 1 << st;
        const int pos = tid & (half - 1);
        const int i0 = ((tid >> st) << (st + 1)) + pos, i1 = i0 + half;
        const int k = pos << (8 - st);                    // twiddle exp(-2 pi i pos / (2 half))
        const float c = p.tw[2 * k], sn = p.tw[2 * k + 1];
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        const float tr = br * c + bi * sn, ti = bi * c - br * sn;
        re[i0] = ar + tr; im[i0] = ai + ti;
        re[i1] = ar - tr; im[i1] = ai - ti;
        __syncthreads();
    }
    // power spectrum of bins 0..256 (in place in `re`; bin 256 by thread 0 into im[0] first to avoid the race)
    const float pw = re[tid] * re[tid] + im[tid] * im[tid];
    float p256 = 0.f;
    if (tid == 0) p256 = re[256] * re[256] + im[256] * im[256];
    __syncthreads();
    re[tid] = pw;
    if (tid == 0) re[256] = p256;
    __syncthreads();
    if (tid < FB_MEL) {
        const int lo = p.range[2 * tid], hi = p.range[2 * tid + 1];
        const float* w = p.banks + tid * FB_BINS;
        float acc = 0.f;
        for (int k = lo; k < hi; ++k) acc = __builtin_fmaf(re[k], w[k], acc);
        const float v = (logf(fmaxf(acc, 1.1920928955078125e-07f)) - p.mean) * p.inv;
        if (p.plain) p.plain[((long long)b * p.frames + frame) * FB_MEL + tid] = v;
        if (p.patches) {
            const int ty = frame >> 4, ky = frame & 15, fx = tid >> 4, kx = tid & 15;
            const int ntok_t = p.frames >> 4;
            if (ty < ntok_t) {
                const long long row = (long long)b * ntok_t * 8 + ty * 8 + fx;
                const int col = ky * 16 + kx;
                if (p.dtype == TDC_F16) ((f16*)p.patches)[row * p.ldp + col] = (f16)v;
                else ((bf16*)p.patches)[row * p.ldp + col] = (bf16)v;
            }
        }
    }
}

// gate[r, h] = ga * (gb * a[h] - 1) + 2 with (ga, gb) = sigmoid(w2 q_h + b2)   (w2 = the grep_linear rows summed in
// groups of 4, backbone.py:654-656).  One thread per (row, head).
template <class T>
__global__ void relpos_gate_kernel(const T* q, int ldq, int rows, int heads, int d, const float* w2, const float* b2,
                                   const float* a, float* gate, int ldg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * heads) return;
    const int r = idx / heads, h = idx - r * heads;
    const T* qp = q + (long long)r * ldq + h * d;
    float s0 = b2[0], s1 = b2[1];
    for (int c = 0; c < d; ++c) {
        const float v = (float)qp[c];
        s0 = __builtin_fmaf(v, w2[c], s0);
        s1 = __builtin_fmaf(v, w2[d + c], s1);
    }
    const float ga = 1.0f / (1.0f + expf(-s0)), gb = 1.0f / (1.0f + expf(-s1));
    gate[(long long)r * ldg + h] = ga * (gb * a[h] - 1.0f) + 2.0f;
}

}  // namespace

extern "C" int tdc_fbank_frames(long long n_samples) {
    return n_samples < FB_WIN ? 0 : (int)(1 + (n_samples - FB_WIN) / FB_SHIFT);
}

extern "C" int tdc_fbank(const void* wav, int wav_f32, long long n_samples, long long wav_bs, int B, const float* window,
                         const float* twiddle, const float* banks, const int* range, float* plain, void* patches,
                         int ldp, int dtype, float mean, float inv_scale, void* stream) {
    if (!wav || !window || !twiddle || !banks || !range || B <= 0 || B > 65535 || (!plain && !patches)) return TDC_E_BADARG;
    if (patches && (ldp < 256 || (dtype != TDC_F16 && dtype != TDC_BF16))) return TDC_E_BADARG;
    const int frames = tdc_fbank_frames(n_samples);
    if (frames <= 0 || wav_bs < n_samples) return TDC_E_BADARG;
    FbankArgs a;
    a.wav = wav; a.wav_f32 = wav_f32; a.wav_bs = wav_bs; a.frames = frames; a.window = window; a.tw = twiddle;
    a.banks = banks; a.range = range; a.plain = plain; a.patches = patches; a.ldp = ldp; a.dtype = dtype;
    a.mean = mean; a.inv = inv_scale; a.preemph = 0.97f;
    hipLaunchKernelGGL(fbank_kernel, dim3(frames, B), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int tdc_relpos_gate(const void* q, int ldq, int rows, int heads, int head_dim, const float* w2,
                               const float* b2, const float* grep_a, float* gate, int ldg, int dtype, void* stream) {
    if (!q || !w2 || !b2 || !grep_a || !gate || rows <= 0 || heads <= 0 || head_dim <= 0 || ldg < heads ||
        ldq < heads * head_dim)
        return TDC_E_BADARG;
    const int n = rows * heads, bs = 256;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TDC_F16)
        hipLaunchKernelGGL(relpos_gate_kernel<f16>, dim3((n + bs - 1) / bs), dim3(bs), 0, st, (const f16*)q, ldq, rows,
                           heads, head_dim, w2, b2, grep_a, gate, ldg);
    else if (dtype == TDC_BF16)
        hipLaunchKernelGGL(relpos_gate_kernel<bf16>, dim3((n + bs - 1) / bs), dim3(bs), 0, st, (const bf16*)q, ldq, rows,
                           heads, head_dim, w2, b2, grep_a, gate, ldg);
    else
        return TDC_E_BADARG;
    return (int)hipGetLastError();
}
