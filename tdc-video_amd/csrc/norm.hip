// LayerNorm / L2-normalise / Q-Former embedding+LN: HBM-bound row kernels, one wave per row, fp32 statistics.
#include "common.h"
#include "../../include/tdc_hip.h"
#include "profile.h"
#include <stdio.h>

namespace {

// One wave (64 lanes) per row; a row of up to 64*MAXV columns is kept in registers (MAXV elements per lane,
// strided by 64 so that consecutive lanes touch consecutive addresses).

struct LnArgs {
    const void* x; int ldx; int x_f32;
    void* y16; int ldy16;
    float* y32; int ldy32;
    const float* gamma; const float* beta; float eps;
    const float* add; int ldadd; int add_period; int add_mode; int add_side;
    int rows, cols, pad_cols;
    RowMap xm, ym;
    // fp8 (OCP e4m3) output for an fp8-operand GEMM: y8[row] = LN(x) / s_a with s_a = max|LN(x)| / 448 per row, and
    // st8[row] = (0, s_a * wscale): the (mean, rstd) pair the consuming tdc_gemm folds into its epilogue
    unsigned char* y8; int ldy8; float* st8; float wscale;
};

// NV4 = number of 4-column groups per lane: lane l owns columns 4*(l + 64*i) .. +3 (16-byte fp32 / 8-byte 16-bit
// accesses, consecutive lanes on consecutive addresses).  Requires cols % 4 == 0 and 4-element aligned rows.
// Q8: the e4m3 output path (y8), a separate instantiation - as a run-time branch it kept the normalised row and the
// quantisation temporaries live in the 16-bit kernel as well (208 VGPRs instead of 46 at NV4 = 6: a third of the waves,
// 4 TB/s instead of 6).
template <class T, int NV4, bool Q8>
__global__ __launch_bounds__(256) void ln_kernel(LnArgs p) {
    typedef typename VecOf<T>::v4 v4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const long long xrow = p.xm(row), yrow = p.ym(row);
    f32x4 v[NV4];
    const float* addrow = nullptr;
    if (p.add) {
        int t = row % p.add_period;
        if (p.add_mode == 1) {
            int y = t / p.add_side, x = t - y * p.add_side;
            t = (y & 1) * 2 + (x & 1);
        }
        addrow = p.add + (long long)t * p.ldadd;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = (lane + i * 64) * 4;
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < p.cols) {
            if (p.x_f32) {
                a = __builtin_nontemporal_load((const f32x4*)((const float*)p.x + xrow * p.ldx + c));
            } else {
                v4 h = *(const v4*)((const T*)p.x + xrow * p.ldx + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (float)h[e];
            }
            if (addrow) a += *(const f32x4*)(addrow + c);
        }
        v[i] = a;
        s += (a[0] + a[1]) + (a[2] + a[3]);
    }
    s = wave_sum(s);
    const float mean = s / (float)p.cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = (lane + i * 64) * 4;
        if (c < p.cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)p.cols + p.eps);
    float amax = 0.f, osq = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = (lane + i * 64) * 4;
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < p.cols) {
            const f32x4 gm = *(const f32x4*)(p.gamma + c), bt = *(const f32x4*)(p.beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
        }
        if (Q8) {           // keep the normalised row (zero beyond cols) for the quantisation pass
            v[i] = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(o[e]));
            osq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        } else if (c < p.pad_cols) {
            if (p.y16) {
                *(v4*)((T*)p.y16 + yrow * p.ldy16 + c) = cvt4<T>(o);
            }
            if (p.y32) *(f32x4*)(p.y32 + yrow * p.ldy32 + c) = o;
        }
    }
    if (Q8) {
        amax = wave_max(amax);          // the two reductions are independent: their cross-lane steps interleave
        osq = wave_sum(osq);
        const float sa = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
        const float inv = 1.0f / sa;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = (lane + i * 64) * 4;
            if (c < p.pad_cols) {
                const f32x4 q = v[i] * inv;                 // zero beyond cols
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true);
                *(int*)(p.y8 + yrow * p.ldy8 + c) = w;
            }
        }
        // (bound of the 2-norm of the quantised row y / s_a - e4m3 rounds by at most 2^-4 -, scale): tdc_gemm_desc ln_stats /
        // out_fp8
        if (lane == 0) *(float2*)(p.st8 + 2 * yrow) = make_float2(1.07f * inv * sqrtf(osq), sa * p.wscale);
    }
}

template <class T>
int launch_ln(const LnArgs& a, hipStream_t st) {
    const int nv = (a.pad_cols + 255) / 256;
    dim3 grid((a.rows + 3) / 4), block(256);
#define LN_LAUNCH(NV)                                                                          \
    do {                                                                                       \
        if (a.y8) hipLaunchKernelGGL((ln_kernel<T, NV, true>), grid, block, 0, st, a);         \
        else hipLaunchKernelGGL((ln_kernel<T, NV, false>), grid, block, 0, st, a);             \
    } while (0)
    if (nv <= 1) LN_LAUNCH(1);
    else if (nv <= 3) LN_LAUNCH(3);
    else if (nv <= 4) LN_LAUNCH(4);
    else if (nv <= 5) LN_LAUNCH(5);
    else if (nv <= 6) LN_LAUNCH(6);
    else if (nv <= 16) LN_LAUNCH(16);
    else return TDC_E_BADARG;
#undef LN_LAUNCH
    return (int)hipGetLastError();
}

// 16-bit rows in, 16-bit rows out (the towers' LayerNorms over a 16-bit residual stream, tdc_vit_model.res_dtype_p1): 4 B per
// element instead of 6, every access 16 bytes per lane (lane l owns columns 8 (l + 64 i) .. + 7).  TI / TO: input / output type
// (an fp16 stream normalised into bf16 GEMM operands).  Same arithmetic as ln_kernel: fp32 statistics, two passes over the
// register-held row.
template <class TI, class TO, int NV8>
__global__ __launch_bounds__(256) void ln16_kernel(LnArgs p) {
    typedef typename VecOf<TI>::v8 v8i;
    typedef typename VecOf<TO>::v4 v4o;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const long long xrow = p.xm(row), yrow = p.ym(row);
    float v[NV8][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        const int c = (lane + i * 64) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (c < p.cols) {
            const v8i h = __builtin_nontemporal_load((const v8i*)((const TI*)p.x + xrow * p.ldx + c));
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = (float)h[e];
        }
        s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    }
    s = wave_sum(s);
    const float mean = s / (float)p.cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        const int c = (lane + i * 64) * 8;
        if (c < p.cols) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)p.cols + p.eps);
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        const int c = (lane + i * 64) * 8;
        if (c < p.pad_cols) {
            f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
            if (c < p.cols) {
                const f32x4 g0 = *(const f32x4*)(p.gamma + c), g1 = *(const f32x4*)(p.gamma + c + 4);
                const f32x4 b0 = *(const f32x4*)(p.beta + c), b1 = *(const f32x4*)(p.beta + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
                    o1[e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
                }
            }
            const v4o x = cvt4<TO>(o0), y = cvt4<TO>(o1);
            const u32x2 xr = __builtin_bit_cast(u32x2, x), yr = __builtin_bit_cast(u32x2, y);
            *(u32x4*)((TO*)p.y16 + yrow * p.ldy16 + c) = (u32x4){xr[0], xr[1], yr[0], yr[1]};
        }
    }
}

template <class TI, class TO>
int launch_ln16(const LnArgs& a, hipStream_t st) {
    const int nv = (a.pad_cols + 511) / 512;
    dim3 grid((a.rows + 3) / 4), block(256);
    if (nv <= 1) hipLaunchKernelGGL((ln16_kernel<TI, TO, 1>), grid, block, 0, st, a);
    else if (nv <= 2) hipLaunchKernelGGL((ln16_kernel<TI, TO, 2>), grid, block, 0, st, a);
    else if (nv <= 3) hipLaunchKernelGGL((ln16_kernel<TI, TO, 3>), grid, block, 0, st, a);
    else if (nv <= 4) hipLaunchKernelGGL((ln16_kernel<TI, TO, 4>), grid, block, 0, st, a);
    else if (nv <= 8) hipLaunchKernelGGL((ln16_kernel<TI, TO, 8>), grid, block, 0, st, a);
    else return TDC_E_BADARG;
    return (int)hipGetLastError();
}

// ---- L2 normalise in place --------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void l2n_kernel(T* x, int ld, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    T* r = x + (long long)row * ld;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { float a = (float)r[c]; s += a * a; }
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < cols; c += 64) r[c] = (T)((float)r[c] * inv);
}

// ---- Q-Former embeddings + LayerNorm -------------------------------------------------------------------------
struct QeArgs {
    const void* query; int ldq; const int* qsrc;
    const float* word; const float* pos; int ldw; const int* ids; int Lt;
    const float* gamma; const float* beta; float eps;
    float* h32; void* h16; int ld;
    int F, K, cols;
};

template <class T, int NV>
__global__ __launch_bounds__(256) void qembed_kernel(QeArgs p) {
    const int lane = threadIdx.x & 63;
    const int S = p.K + p.Lt;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.F * S) return;
    const int f = row / S, s = row - f * S;
    float v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane + i * 64;
        float a = 0.f;
        if (c < p.cols) {
            if (s < p.K) {
                a = (float)((const T*)p.query)[((long long)p.qsrc[f] * p.K + s) * p.ldq + c];
            } else {
                int t = s - p.K;
                a = p.word[(long long)p.ids[t] * p.ldw + c] + p.pos[(long long)t * p.ldw + c];
            }
        }
        v[i] = a;
        sum += a;
    }
    sum = wave_sum(sum);
    const float mean = sum / (float)p.cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane + i * 64;
        float d = (c < p.cols) ? v[i] - mean : 0.f;
        q += d * d;
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)p.cols + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = lane + i * 64;
        if (c < p.ld) {
            float o = (c < p.cols) ? (v[i] - mean) * rstd * p.gamma[c] + p.beta[c] : 0.f;
            p.h32[(long long)row * p.ld + c] = o;
            ((T*)p.h16)[(long long)row * p.ld + c] = (T)o;
        }
    }
}

}  // namespace

extern "C" int tdc_layernorm(const tdc_ln_desc* d, void* stream) {
    if (!d || !d->x || d->rows <= 0 || d->cols <= 0 || (!d->y16 && !d->y32 && !d->y8)) return TDC_E_BADARG;
    if (d->y8 && (d->y16 || d->y32)) return TDC_E_BADARG;      /* the e4m3 instantiation writes y8 only */
    if (d->y8 && (!d->y8_stats || (d->ldy8 & 3) || ((uintptr_t)d->y8 & 3) || ((uintptr_t)d->y8_stats & 7))) return TDC_E_BADARG;
    LnArgs a;
    a.y8 = (unsigned char*)d->y8; a.ldy8 = d->ldy8; a.st8 = d->y8_stats; a.wscale = d->y8_wscale;
    a.x = d->x; a.ldx = d->ldx; a.x_f32 = d->x_f32;
    a.y16 = d->y16; a.ldy16 = d->ldy16; a.y32 = d->y32; a.ldy32 = d->ldy32;
    a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps;
    a.add = d->add; a.ldadd = d->ldadd; a.add_period = d->add_period > 0 ? d->add_period : 1;
    a.add_mode = d->add_mode;
    a.add_side = 1;
    if (d->add_mode == 1) {
        int s = 1;
        while (s * s < a.add_period) ++s;
        a.add_side = s;
    }
    a.rows = d->rows; a.cols = d->cols;
    if (d->x_map.seg < 0 || d->y_map.seg < 0) return TDC_E_BADARG;
    a.xm = RowMap::make(d->x_map.seg, d->x_map.stride, d->x_map.off, d->x_map.inner);
    a.ym = RowMap::make(d->y_map.seg, d->y_map.stride, d->y_map.off, d->y_map.inner);
    a.pad_cols = (d->cols + 63) / 64 * 64;
    int ldmin = a.pad_cols;
    if ((d->y16 && d->ldy16 < ldmin) || (d->y32 && d->ldy32 < ldmin) || (d->y8 && d->ldy8 < ldmin)) {
        // outputs narrower than the padded width: only write the real columns
        a.pad_cols = d->cols;
    }
    // vector accesses: 4-element groups
    if ((d->cols & 3) || (d->ldx & 3) || (d->y16 && (d->ldy16 & 3)) || (d->y32 && (d->ldy32 & 3)) ||
        (d->add && (d->ldadd & 3)) || ((uintptr_t)d->x & 15 & (d->x_f32 ? 15 : 7)) ||
        ((uintptr_t)d->gamma & 15) || ((uintptr_t)d->beta & 15)) {
        fprintf(stderr, "[tdc_hip] tdc_layernorm: cols/ld must be multiples of 4 and pointers 16-byte aligned\n");
        return TDC_E_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype != TDC_F16 && d->dtype != TDC_BF16) return TDC_E_BADARG;
    TdcProfScope prof(TDC_PROF_LN, st, d->rows, d->cols, 0, 0, 0, 0, nullptr, 0.0);
    if (d->x_dtype_p1) {
        // 16-bit input of its own type -> 16-bit output (ln16_kernel): 16-byte accesses on both sides
        const int xt = d->x_dtype_p1 - 1;
        if (d->x_f32 || (xt != TDC_F16 && xt != TDC_BF16) || !d->y16 || d->y32 || d->y8 || d->add || (d->cols & 7) ||
            (d->ldx & 7) || (d->ldy16 & 7) || (a.pad_cols & 7) || ((uintptr_t)d->x & 15) || ((uintptr_t)d->y16 & 15))
            return TDC_E_BADARG;
        if (xt == TDC_F16) return d->dtype == TDC_F16 ? launch_ln16<f16, f16>(a, st) : launch_ln16<f16, bf16>(a, st);
        return d->dtype == TDC_F16 ? launch_ln16<bf16, f16>(a, st) : launch_ln16<bf16, bf16>(a, st);
    }
    if (d->dtype == TDC_F16) return launch_ln<f16>(a, st);
    return launch_ln<bf16>(a, st);
}

extern "C" int tdc_l2_normalize(void* x, int ld, int rows, int cols, int dtype, void* stream) {
    if (!x || rows <= 0 || cols <= 0) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    if (dtype == TDC_F16) hipLaunchKernelGGL(l2n_kernel<f16>, grid, block, 0, st, (f16*)x, ld, rows, cols);
    else if (dtype == TDC_BF16) hipLaunchKernelGGL(l2n_kernel<bf16>, grid, block, 0, st, (bf16*)x, ld, rows, cols);
    else return TDC_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int tdc_qformer_embed(const tdc_qembed_desc* d, void* stream) {
    if (!d || !d->query || !d->h32 || !d->h16 || d->F <= 0 || d->K <= 0) return TDC_E_BADARG;
    if (d->Lt > 0 && (!d->word || !d->pos || !d->ids)) return TDC_E_BADARG;
    QeArgs a;
    a.query = d->query; a.ldq = d->ldq; a.qsrc = d->qsrc;
    a.word = d->word; a.pos = d->pos; a.ldw = d->ldw; a.ids = d->ids; a.Lt = d->Lt;
    a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps;
    a.h32 = d->h32; a.h16 = d->h16; a.ld = d->ld;
    a.F = d->F; a.K = d->K; a.cols = d->cols;
    const int nv = (d->ld + 63) / 64;
    const int rows = d->F * (d->K + d->Lt);
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
#define QE(T)                                                                                   \
    if (nv <= 2) hipLaunchKernelGGL((qembed_kernel<T, 2>), grid, block, 0, st, a);              \
    else if (nv <= 12) hipLaunchKernelGGL((qembed_kernel<T, 12>), grid, block, 0, st, a);       \
    else if (nv <= 16) hipLaunchKernelGGL((qembed_kernel<T, 16>), grid, block, 0, st, a);       \
    else return TDC_E_BADARG;
    if (d->dtype == TDC_F16) { QE(f16) }
    else if (d->dtype == TDC_BF16) { QE(bf16) }
    else return TDC_E_BADARG;
#undef QE
    return (int)hipGetLastError();
}

// ---- LayerNorm fusion: row statistics from the per-slot partials of the producer GEMM ----------------------------------
// (tdc_gemm with x16 / ln_part).  One thread per row; slots of 64 columns are combined in index order (Chan):
// mean = sum(mean_s) / S, M2 = sum(M2_s) + 64 * sum((mean_s - mean)^2); rstd as in ln_kernel.
namespace {
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int slots, int rows, float eps,
                                                          float* __restrict__ stats) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float2* p = (const float2*)part + r;          // slot-major: part[s][rows][2] -> coalesced across the rows
    float sm = 0.f;
    for (int s = 0; s < slots; ++s) sm += p[(long long)s * rows].x;
    const float mean = sm / (float)slots;
    float m2 = 0.f, dv = 0.f;
    for (int s = 0; s < slots; ++s) {
        const float2 v = p[(long long)s * rows];
        const float d = v.x - mean;
        m2 += v.y;
        dv = __builtin_fmaf(d, d, dv);
    }
    const float q = m2 + 64.0f * dv;
    *(float2*)(stats + 2 * (long long)r) = make_float2(mean, rsqrtf(q / (float)(slots * 64) + eps));
}
}  // namespace

extern "C" int tdc_ln_finalize(const float* ln_part, int slots, int rows, float eps, float* stats, void* stream) {
    if (!ln_part || !stats || slots <= 0 || rows <= 0 || ((uintptr_t)ln_part & 7) || ((uintptr_t)stats & 7))
        return TDC_E_BADARG;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, ln_part, slots,
                       rows, eps, stats);
    return (int)hipGetLastError();
}

// ---- per-row e4m3 quantisation of a 16-bit matrix (fp8 operands for GEMMs whose input is not produced by a LayerNorm:
// the attention output and the MLP hidden).  One wave per row, the row in registers (8 values per lane and step):
// y8 = x / s_a with s_a = max|x| / 448, stats = (0, s_a * wscale) exactly as tdc_layernorm's y8 output.
namespace {
template <class T, int NV8>
__global__ __launch_bounds__(256) void quant_rows_kernel(const T* __restrict__ x, int ldx, int rows, int cols,
                                                         unsigned char* __restrict__ y8, int ldy8, int pad_cols,
                                                         float* __restrict__ stats, float wscale) {
    typedef typename VecOf<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NV8][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        const int c = (lane + i * 64) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (c < cols) {
            const v8 h = *(const v8*)(x + (long long)row * ldx + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)h[e]; amax = fmaxf(amax, fabsf(v[i][e])); }
        }
    }
    amax = wave_max(amax);
    const float sa = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sa;
#pragma unroll
    for (int i = 0; i < NV8; ++i) {
        const int c = (lane + i * 64) * 8;
        if (c < pad_cols) {
            int w0 = 0, w1 = 0;
            if (c < cols) {
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, w1, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, w1, true);
            }
            *(int2*)(y8 + (long long)row * ldy8 + c) = make_int2(w0, w1);
        }
    }
    if (lane == 0) *(float2*)(stats + 2 * (long long)row) = make_float2(0.f, sa * wscale);
}
}  // namespace

extern "C" int tdc_quantize_rows_fp8(const void* x, int ldx, int rows, int cols, int dtype, void* y8, int ldy8,
                                     float* stats, float wscale, void* stream) {
    if (!x || !y8 || !stats || rows <= 0 || cols <= 0 || (cols & 7) || (ldx & 7) || (ldy8 & 7) || ldy8 < cols ||
        ((uintptr_t)x & 15) || ((uintptr_t)y8 & 7) || ((uintptr_t)stats & 7))
        return TDC_E_BADARG;
    int pad = (cols + 127) / 128 * 128;            // zero bytes up to the K tile of the consuming GEMM
    if (pad > ldy8) pad = cols;
    const int nv = (pad + 511) / 512;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    unsigned char* y = (unsigned char*)y8;
#define QLAUNCH(TT, NV) hipLaunchKernelGGL((quant_rows_kernel<TT, NV>), grid, block, 0, st, (const TT*)x, ldx, rows, cols, y, ldy8, pad, stats, wscale)
#define QDISPATCH(TT)                                        \
    if (nv <= 3) QLAUNCH(TT, 3); else if (nv <= 6) QLAUNCH(TT, 6); else if (nv <= 9) QLAUNCH(TT, 9); else return TDC_E_BADARG;
    if (dtype == TDC_F16) { QDISPATCH(f16) } else if (dtype == TDC_BF16) { QDISPATCH(bf16) } else return TDC_E_BADARG;
#undef QDISPATCH
#undef QLAUNCH
    return (int)hipGetLastError();
}
