// Data-movement and small reduction kernels of the path (HBM-bound or tiny): im2col, cls rows, token-grid resample,
// adjacent-frame cosine similarity, token mean / adaptive pooling, row gather (unpad+newline, token emission) and the
// SVA 2x2-window cross-attention core.
#include "common.h"
#include "../../include/tdc_hip.h"
#include <stdio.h>

namespace {

// ------------------------------------------------------------------------------------------------ im2col
// patches[(b*gh+gy)*gw+gx][c*p*p + ky*p + kx] = px[b][c][gy*p+ky][gx*p+kx]; one workgroup per patch row (b,gy):
// reads are contiguous along W, writes contiguous along the patch vector.
template <class T> struct Other16;
template <> struct Other16<f16> { typedef bf16 type; };
template <> struct Other16<bf16> { typedef f16 type; };
// px_f32: 0 = pixels of type T, 1 = fp32, 2 = the other 16-bit type (fp16 frames into bf16 towers and vice versa)
template <class T>
__global__ __launch_bounds__(256) void im2col_kernel(const void* px, int px_f32, T* out, int ldp, int H, int W,
                                                     int patch, int gh, int gw) {
    typedef typename Other16<T>::type TX;
    const int b = blockIdx.y, gy = blockIdx.x;
    const int kdim = 3 * patch * patch;
    const long long img = (long long)b * 3 * H * W;
    for (int idx = threadIdx.x; idx < gw * ldp; idx += 256) {
        int gx = idx / ldp, k = idx - gx * ldp;
        float v = 0.f;
        if (k < kdim) {
            int c = k / (patch * patch), rem = k - c * patch * patch;
            int ky = rem / patch, kx = rem - ky * patch;
            long long src = img + ((long long)c * H + gy * patch + ky) * W + gx * patch + kx;
            v = px_f32 == 1 ? ((const float*)px)[src] : px_f32 == 2 ? (float)((const TX*)px)[src] : (float)((const T*)px)[src];
        }
        out[((long long)(b * gh + gy) * gw + gx) * ldp + k] = (T)v;
    }
}

// the same for 16-bit pixels (the towers' own launches) through LDS: a workgroup loads the 3 x patch pixel rows of its patch row
// (b, gy) with 4-byte loads - a 378-pixel DINOv2 row starts on a 4-byte boundary only - and writes every patch vector as 16-byte
// pieces assembled from LDS.  The element-wise kernel above reads 28-byte runs (14 pixels) with 2-byte loads: 1.6 TB/s on the
// 0.93 GB of a 512-frame batch.  W, patch even; 3 * patch * W * 2 bytes of LDS (32 KB at 384 px).
template <class T, class TP>
__global__ __launch_bounds__(256) void im2col_lds_kernel(const TP* px, T* out, int ldp, int H, int W, int patch, int gh, int gw) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char im2col_smem[];
    TP* slab = (TP*)im2col_smem;                                            // [3 * patch][W]
    const int b = blockIdx.y, gy = blockIdx.x;
    const int rows = 3 * patch, wd = W >> 1;                                 // dwords per pixel row
    const long long img = (long long)b * 3 * H * W;
    for (int i = threadIdx.x; i < rows * wd; i += 256) {
        const int rr = i / wd, xd = i - rr * wd;
        const int c = rr / patch, ky = rr - c * patch;
        const unsigned* src = (const unsigned*)(px + img + ((long long)c * H + gy * patch + ky) * W);
        ((unsigned*)slab)[rr * wd + xd] = src[xd];
    }
    __syncthreads();
    const int kdim = 3 * patch * patch, chunks = ldp >> 3, pp = patch * patch;
    for (int i = threadIdx.x; i < gw * chunks; i += 256) {
        const int gx = i / chunks, k0 = (i - gx * chunks) * 8;
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float v = 0.f;
            if (k < kdim) {
                const int c = k / pp, rem = k - c * pp;
                const int ky = rem / patch, kx = rem - ky * patch;
                v = (float)slab[(c * patch + ky) * W + gx * patch + kx];
            }
            o[e] = (T)v;
        }
        *(v8*)(out + ((long long)(b * gh + gy) * gw + gx) * ldp + k0) = o;
    }
}

__global__ void set_rows_kernel(float* x, int ld, int S, int row, const float* vec) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) x[((long long)b * S + row) * ld + c] = vec[c];
}

// ------------------------------------------------------------------------------------------------ resample
template <class T, class TO>   // T: 16-bit type of x (when !x_f32), TO: 16-bit type of y (they differ when the towers run in
                               // bf16 and the connector behind them in fp16)
__global__ __launch_bounds__(256) void resample_kernel(const void* x, int x_f32, int ldx, int tok_off, int n_in,
                                                       TO* y, int ldy, int n_out, const int* i0, const int* i1,
                                                       const float* fr, int cols) {
    // F.interpolate(bilinear) is separable; torch evaluates w00*a + w01*b + w10*c + w11*d per output pixel
    const int b = blockIdx.y, o = blockIdx.x;
    const int oy = o / n_out, ox = o - oy * n_out;
    const int y0 = i0[oy], y1 = i1[oy], x0 = i0[ox], x1 = i1[ox];
    const float fy = fr[oy], fx = fr[ox];
    const long long base = (long long)b * (tok_off + n_in * n_in) + tok_off;
    const long long r00 = (base + y0 * n_in + x0) * ldx, r01 = (base + y0 * n_in + x1) * ldx;
    const long long r10 = (base + y1 * n_in + x0) * ldx, r11 = (base + y1 * n_in + x1) * ldx;
    TO* out = y + ((long long)b * n_out * n_out + o) * ldy;
    for (int c = threadIdx.x; c < ldy; c += 256) {
        float v = 0.f;
        if (c < cols) {
            float a, bb, cc, d;
            if (x_f32) {
                const float* p = (const float*)x;
                a = p[r00 + c]; bb = p[r01 + c]; cc = p[r10 + c]; d = p[r11 + c];
            } else {
                const T* p = (const T*)x;
                a = (float)p[r00 + c]; bb = (float)p[r01 + c]; cc = (float)p[r10 + c]; d = (float)p[r11 + c];
            }
            v = (1.f - fy) * ((1.f - fx) * a + fx * bb) + fy * ((1.f - fx) * cc + fx * d);
        }
        out[c] = (TO)v;
    }
}

// the same for 16-bit inputs with 16-byte aligned rows (every tower launch): a thread resamples 8 consecutive columns - four
// 16-byte loads, one 16-byte store; two output tokens per 256-thread workgroup at the towers' widths (the element-wise kernel
// above ran the 27 x 27 -> 24 x 24 resample of a 512-frame batch at 1.4 TB/s).  Same expression per element: same bits.
template <class T, class TO>
__global__ __launch_bounds__(256) void resample8_kernel(const T* x, int ldx, int tok_off, int n_in, TO* y, int ldy, int n_out,
                                                        const int* i0, const int* i1, const float* fr, int cols, int tpt,
                                                        long long n_tok) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<TO>::v8 v8o;
    const int tl = threadIdx.x / tpt, ct = threadIdx.x - tl * tpt;          // token within the workgroup, 8-column chunk
    const long long tk = (long long)blockIdx.x * (256 / tpt) + tl;            // flat output token (b, oy, ox)
    const int c = ct * 8;
    if (tk >= n_tok || c >= ldy) return;
    const int per = n_out * n_out;
    const int b = (int)(tk / per), o = (int)(tk - (long long)b * per);
    const int oy = o / n_out, ox = o - oy * n_out;
    const int y0 = i0[oy], y1 = i1[oy], x0 = i0[ox], x1 = i1[ox];
    const float fy = fr[oy], fx = fr[ox];
    const long long base = (long long)b * (tok_off + n_in * n_in) + tok_off;
    v8o out8;
    if (c < cols) {      // cols % 8 == 0 (host): a chunk is valid as a whole
        const v8 a8 = *(const v8*)(x + (base + y0 * n_in + x0) * ldx + c), b8 = *(const v8*)(x + (base + y0 * n_in + x1) * ldx + c);
        const v8 c8 = *(const v8*)(x + (base + y1 * n_in + x0) * ldx + c), d8 = *(const v8*)(x + (base + y1 * n_in + x1) * ldx + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = (float)a8[e], bb = (float)b8[e], cc = (float)c8[e], d = (float)d8[e];
            const float v = (1.f - fy) * ((1.f - fx) * a + fx * bb) + fy * ((1.f - fx) * cc + fx * d);
            out8[e] = (TO)v;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) out8[e] = (TO)0.f;
    }
    *(v8o*)(y + tk * ldy + c) = out8;
}

// ------------------------------------------------------------------------------------------------ cos-sim
// pass 1: partial sums of <f_t, f_t> and <f_t, f_{t+1}> (fp32, fixed reduction order => deterministic ranking);
// grid (chunks, T).  pass 2: combine.
constexpr int CS_CHUNKS = 32;
template <class T>
__global__ __launch_bounds__(256) void cossim_partial(const T* f, long long n, int Tn, float* part) {
    typedef typename VecOf<T>::v8 v8;
    const int t = blockIdx.y, ch = blockIdx.x;
    const long long per = ((n / 8 + CS_CHUNKS - 1) / CS_CHUNKS) * 8;
    const long long lo = ch * per, hi = (lo + per < n) ? lo + per : n;
    const T* a = f + (long long)t * n;
    const T* b = f + (long long)(t + 1 < Tn ? t + 1 : t) * n;
    float saa = 0.f, sab = 0.f;
    for (long long i = lo + threadIdx.x * 8; i < hi; i += 256 * 8) {
        v8 va = *(const v8*)(a + i), vb = *(const v8*)(b + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = (float)va[e], y = (float)vb[e];
            saa += x * x;
            sab += x * y;
        }
    }
    __shared__ float red[2][4];
    saa = wave_sum(saa); sab = wave_sum(sab);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = saa; red[1][w] = sab; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long long)t * CS_CHUNKS + ch) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[((long long)t * CS_CHUNKS + ch) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ void cossim_final(const float* part, int Tn, float* sims) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tn - 1) return;
    float aa = 0.f, ab = 0.f, bb = 0.f;
    for (int c = 0; c < CS_CHUNKS; ++c) {
        aa += part[((long long)t * CS_CHUNKS + c) * 2];
        ab += part[((long long)t * CS_CHUNKS + c) * 2 + 1];
        bb += part[((long long)(t + 1) * CS_CHUNKS + c) * 2];
    }
    // F.cosine_similarity: x.y / max(|x| |y|, eps) with eps = 1e-8
    sims[t] = ab / fmaxf(sqrtf(aa) * sqrtf(bb), 1e-8f);
}

// ------------------------------------------------------------------------------------------------ pooling
// one workgroup per frame; thread = (16-byte column chunk, token phase): 16-byte loads, several independent rows in flight
// per thread, the token phases of a chunk combined through LDS in a fixed order (deterministic).  The previous form - one
// thread per column walking all P tokens with 2-byte loads - was a 576-deep dependent load chain: 0.9 ms whatever the frame
// count (rocprof at T = 64).
template <class T>
__global__ __launch_bounds__(256) void token_mean_kernel(const T* x, int P, int ld, T* y) {
    typedef typename VecOf<T>::v8 v8;
    __shared__ float part[256][8];
    const int b = blockIdx.x;
    const int nch = ld >> 3;                                   // 16-byte chunks per row (ld % 8 == 0: host-checked)
    const int nchb = nch < 256 ? nch : 256;                    // chunks handled per sweep
    const int phases = 256 / nchb;                             // token phases (1 when a row has >= 256 chunks)
    const int c0 = threadIdx.x % nchb, ph = threadIdx.x / nchb;
    const T* base = x + (long long)b * P * ld;
    for (int cb = 0; cb < nch; cb += nchb) {                   // uniform trip count: barriers inside
        const int c = cb + c0;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ph < phases && c < nch) {
#pragma unroll 4
            for (int t = ph; t < P; t += phases) {
                const v8 v = *(const v8*)(base + (long long)t * ld + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[threadIdx.x][e] = acc[e];
        __syncthreads();
        if (ph == 0 && c < nch) {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = part[c0][e];
                for (int q = 1; q < phases; ++q) s += part[q * nchb + c0][e];
                o[e] = (T)(s / (float)P);
            }
            *(v8*)(y + (long long)b * ld + c * 8) = o;
        }
        __syncthreads();
    }
}
template <class T>
__global__ __launch_bounds__(256) void adaptive_pool_kernel(const T* x, int N, int frame_rows, int ld, T* y, int K,
                                                            const int* src_row) {
    const int b = blockIdx.y, k = blockIdx.x;
    const int s = (k * N) / K, e = ((k + 1) * N + K - 1) / K;
    const long long xb = (long long)(src_row ? src_row[b] : b) * frame_rows;
    for (int c = threadIdx.x; c < ld; c += 256) {
        float acc = 0.f;
        for (int t = s; t < e; ++t) acc += (float)x[(xb + t) * ld + c];
        y[((long long)b * K + k) * ld + c] = (T)(acc / (float)(e - s));
    }
}

// ------------------------------------------------------------------------------------------------ gather
template <class T>
__global__ __launch_bounds__(256) void gather_kernel(tdc_gather_tables t, const int* src, T* out, int ldo, int n,
                                                     int cols) {
    const int i = blockIdx.x;
    const int k = src[2 * i], r = src[2 * i + 1];
    const T* row = (const T*)t.base[k] + (long long)r * t.ld[k];
    T* o = out + (long long)i * ldo;
    typedef typename VecOf<T>::v8 v8;
    const bool vec = ((cols & 7) == 0) && ((t.ld[k] & 7) == 0) && ((ldo & 7) == 0) &&
                     (((uintptr_t)t.base[k] & 15) == 0) && (((uintptr_t)out & 15) == 0);
    if (vec) {
        for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) *(v8*)(o + c) = *(const v8*)(row + c);
    } else {
        for (int c = threadIdx.x; c < cols; c += 256) o[c] = row[c];
    }
}

// ------------------------------------------------------------------------------------------------ SVA core
// one wave per (query, head-group): q_len = 1, kv = n_towers * r * r <= 8 keys.  Each lane owns dim/64 channels;
// head h covers channels [h*hd, (h+1)*hd): per-head dot products are reduced inside the lanes that share a head.
struct SvaArgs {
    const void* q; int ldq;
    const void* kv[2]; int ldkv;
    const unsigned char* mask;
    void* out; int ldo;
    int T, side, r, n_towers, dim, heads;
};
template <class T>
__global__ __launch_bounds__(256) void sva_kernel(SvaArgs p) {
    // thread <-> channel c (strided by 256) ; per-head reduction through LDS (heads <= 64, kv <= 8)
    __shared__ float sc[8][64];  // [key][head]
    const int qi = blockIdx.x;   // query index = (t*side + i)*side + j
    const int nq = p.side * p.side;
    const int t = qi / nq, w = qi - t * nq;
    const int wi = w / p.side, wj = w - wi * p.side;
    const int n = p.side * p.r;
    const int hd = p.dim / p.heads;
    const int nkv = p.n_towers * p.r * p.r;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) ((float*)sc)[i] = 0.f;
    __syncthreads();
    const T* q = (const T*)p.q + (long long)qi * p.ldq;
    // scores: partial dot per thread then atomicAdd in LDS per head
    const bool wave_per_head = (hd % 64) == 0;  // a wave's 64 channels then lie in one head: deterministic reduce
    for (int c0 = 0; c0 < p.dim; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const bool live = c < p.dim;
        const float qc = live ? (float)q[c] : 0.f;
        const int h = (live ? c : p.dim - 1) / hd;
        int key = 0;
        for (int tw = 0; tw < p.n_towers; ++tw)
            for (int a = 0; a < p.r; ++a)
                for (int b = 0; b < p.r; ++b, ++key) {
                    long long tok = (long long)t * n * n + (wi * p.r + a) * n + (wj * p.r + b);
                    const T* kr = (const T*)p.kv[tw] + tok * p.ldkv;
                    float prod = live ? qc * (float)kr[c] : 0.f;
                    if (wave_per_head) {
                        prod = wave_sum(prod);
                        if ((threadIdx.x & 63) == 0 && live) atomicAdd(&sc[key][h], prod);  // hd == 64: one wave per head
                    } else if (live) {
                        atomicAdd(&sc[key][h], prod);
                    }
                }
    }
    __syncthreads();
    // softmax per head over the kv keys (masked), one thread per head
    if (threadIdx.x < p.heads) {
        const int h = threadIdx.x;
        const float scale = rsqrtf((float)hd);
        float mx = -INFINITY;
        float s[8];
        for (int k = 0; k < nkv; ++k) {
            bool ok = p.mask[(long long)qi * nkv + k] != 0;
            s[k] = ok ? sc[k][h] * scale : -INFINITY;
            mx = fmaxf(mx, s[k]);
        }
        float sum = 0.f;
        for (int k = 0; k < nkv; ++k) { s[k] = __expf(s[k] - mx); sum += s[k]; }
        for (int k = 0; k < nkv; ++k) sc[k][h] = s[k] / sum;
    }
    __syncthreads();
    T* o = (T*)p.out + (long long)qi * p.ldo;
    for (int c = threadIdx.x; c < p.dim; c += 256) {
        const int h = c / hd;
        float acc = 0.f;
        int key = 0;
        for (int tw = 0; tw < p.n_towers; ++tw)
            for (int a = 0; a < p.r; ++a)
                for (int b = 0; b < p.r; ++b, ++key) {
                    long long tok = (long long)t * n * n + (wi * p.r + a) * n + (wj * p.r + b);
                    const T* vr = (const T*)p.kv[tw] + tok * p.ldkv + p.dim;
                    acc += sc[key][h] * (float)vr[c];
                }
        o[c] = (T)acc;
    }
}

// The same attention for the shapes the path runs it on (C = 1024, 16 heads of 64: head dim % 8 == 0, dim / 8 threads per query
// dividing 256, 16-byte aligned rows): HBM-bound byte work - per query 2 KB of q, 8 keys x (2 KB K + 2 KB V), 2 KB out - done
// with 16-byte accesses and every load of a lane in flight at once.  A thread owns 8 consecutive channels of one query; the
// hd / 8 lanes of a head combine their partial dot products with xor-shuffles (a fixed tree: deterministic), every lane of the
// head then holds the head's <= 8 scores and runs the masked softmax redundantly, and the weighted V rows leave as one 16-byte
// store per lane.  No LDS, no atomics, no barrier.  (The kernel above - 2-byte loads, 32 wave reductions and LDS atomics per
// thread - moved 2.5 GB per launch in 1.86 ms at the bench's size: 1.3 TB/s; it stays for the odd shapes of the small fixtures.)
// NT towers x R x R keys are compile-time: the per-key registers below must be indexed by constants (a register array indexed
// by a run-time loop variable lives in scratch: the first version of this kernel ran at 0.4 TB/s)
template <class T, int NT, int R>
__global__ __launch_bounds__(256) void sva8_kernel(SvaArgs p, long long nq_total) {
    constexpr int NKV = NT * R * R;
    typedef typename VecOf<T>::v8 v8;
    const int tpq = p.dim >> 3;                              // threads per query
    const int qpw = 256 / tpq;                               // queries per workgroup
    const int ql = threadIdx.x / tpq, ct = threadIdx.x - ql * tpq;
    long long qi = (long long)blockIdx.x * qpw + ql;
    const bool valid = qi < nq_total;
    if (!valid) qi = nq_total - 1;                           // whole heads (lane groups) are valid or not: no exit before the shuffles
    const int nq = p.side * p.side;
    const int t = (int)(qi / nq), w = (int)(qi - (long long)t * nq);
    const int wi = w / p.side, wj = w - wi * p.side;
    const int n = p.side * R;
    const int hd = p.dim / p.heads, lph = hd >> 3;           // lanes per head: a power of two (host)
    const int c = ct * 8;
    const v8 q8 = *(const v8*)((const T*)p.q + qi * p.ldq + c);
    v8 k8[NKV], v8r[NKV];
    bool ok[NKV];
#pragma unroll
    for (int tw = 0; tw < NT; ++tw)
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) {
                const int key = (tw * R + a) * R + b;
                const long long tok = (long long)t * n * n + (wi * R + a) * n + (wj * R + b);
                const T* row = (const T*)p.kv[tw] + tok * p.ldkv + c;
                k8[key] = *(const v8*)row;
                v8r[key] = *(const v8*)(row + p.dim);
                ok[key] = p.mask[qi * NKV + key] != 0;
            }
    float qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)q8[e];
    const float scale = rsqrtf((float)hd);
    float sc[NKV];
    float mx = -INFINITY;
#pragma unroll
    for (int key = 0; key < NKV; ++key) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = __builtin_fmaf(qf[e], (float)k8[key][e], d);
        for (int o = 1; o < lph; o <<= 1) d += __shfl_xor(d, o);
        sc[key] = ok[key] ? d * scale : -INFINITY;
        mx = fmaxf(mx, sc[key]);
    }
    float sum = 0.f;
#pragma unroll
    for (int key = 0; key < NKV; ++key) { sc[key] = __expf(sc[key] - mx); sum += sc[key]; }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int key = 0; key < NKV; ++key) {
        const float pr = sc[key] / sum;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pr * (float)v8r[key][e];
    }
    if (valid) {
        v8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (T)acc[e];
        *(v8*)((T*)p.out + qi * p.ldo + c) = o8;
    }
}

}  // namespace

#define DISPATCH(dtype, CALL)                       \
    if ((dtype) == TDC_F16) { typedef f16 TT; CALL; }    \
    else if ((dtype) == TDC_BF16) { typedef bf16 TT; CALL; } \
    else return TDC_E_BADARG;

extern "C" int tdc_im2col(const void* px, int px_f32, void* patches, int ldp, int B, int H, int W, int patch,
                          int dtype, void* stream) {
    if (!px || !patches || B <= 0 || patch <= 0 || ldp < 3 * patch * patch || px_f32 < 0 || px_f32 > 2) return TDC_E_BADARG;
    const int gh = H / patch, gw = W / patch;
    hipStream_t st = (hipStream_t)stream;
    const size_t slab = (size_t)3 * patch * W * 2;
    if (px_f32 != 1 && !(W & 1) && !(ldp & 7) && !((uintptr_t)px & 3) && !((uintptr_t)patches & 15) && slab <= 64 * 1024 &&
        (dtype == TDC_F16 || dtype == TDC_BF16)) {
        // 16-bit pixels: through LDS (px_f32 == 2: pixels of the other 16-bit type, converted on the way out)
        const bool px_is_f16 = (dtype == TDC_F16) == (px_f32 == 0);
#define IM2COL_LDS(TO_, TP_) hipLaunchKernelGGL((im2col_lds_kernel<TO_, TP_>), dim3(gh, B), dim3(256), slab, st, (const TP_*)px, \
                                                (TO_*)patches, ldp, H, W, patch, gh, gw)
        if (dtype == TDC_F16) { if (px_is_f16) IM2COL_LDS(f16, f16); else IM2COL_LDS(f16, bf16); }
        else { if (px_is_f16) IM2COL_LDS(bf16, f16); else IM2COL_LDS(bf16, bf16); }
#undef IM2COL_LDS
        return (int)hipGetLastError();
    }
    DISPATCH(dtype, hipLaunchKernelGGL(im2col_kernel<TT>, dim3(gh, B), dim3(256), 0, st, px, px_f32, (TT*)patches,
                                       ldp, H, W, patch, gh, gw));
    return (int)hipGetLastError();
}

extern "C" int tdc_set_rows(float* x32, int ld, int B, int S, int row, const float* vec, void* stream) {
    if (!x32 || !vec || B <= 0) return TDC_E_BADARG;
    hipLaunchKernelGGL(set_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x32, ld, S, row, vec);
    return (int)hipGetLastError();
}

namespace {
template <class T>
__global__ void set_rows16_kernel(T* x, int ld, int S, int row, const float* vec) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) x[((long long)b * S + row) * ld + c] = (T)vec[c];
}
}  // namespace

extern "C" int tdc_set_rows16(void* x16, int ld, int B, int S, int row, const float* vec, int dtype, void* stream) {
    if (!x16 || !vec || B <= 0) return TDC_E_BADARG;
    if (dtype == TDC_F16) hipLaunchKernelGGL(set_rows16_kernel<f16>, dim3(B), dim3(256), 0, (hipStream_t)stream, (f16*)x16, ld, S, row, vec);
    else if (dtype == TDC_BF16) hipLaunchKernelGGL(set_rows16_kernel<bf16>, dim3(B), dim3(256), 0, (hipStream_t)stream, (bf16*)x16, ld, S, row, vec);
    else return TDC_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int tdc_resample_tokens(const void* x, int x_f32, int ldx, int tok_off, int n_in, void* y, int ldy,
                                   int n_out, const int* idx0, const int* idx1, const float* frac, int B, int cols,
                                   int dtype, int out_dtype, void* stream) {
    if (!x || !y || !idx0 || !idx1 || !frac || B <= 0) return TDC_E_BADARG;
    if ((dtype != TDC_F16 && dtype != TDC_BF16) || (out_dtype != TDC_F16 && out_dtype != TDC_BF16)) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (!x_f32 && cols % 8 == 0 && !(ldx & 7) && !(ldy & 7) && !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && ldy <= 2048) {
        // threads per token: ldy / 8 rounded up to a power of two that divides 256
        int tpt = 1;
        while (tpt * 8 < ldy) tpt <<= 1;
        const long long n_tok = (long long)B * n_out * n_out;
        const int per_wg = 256 / tpt;
        const dim3 grid8((unsigned)((n_tok + per_wg - 1) / per_wg));
#define RESAMPLE8(TI, TO_) hipLaunchKernelGGL((resample8_kernel<TI, TO_>), grid8, dim3(256), 0, st, (const TI*)x, ldx, tok_off, \
                                              n_in, (TO_*)y, ldy, n_out, idx0, idx1, frac, cols, tpt, n_tok)
        if (dtype == TDC_F16) { if (out_dtype == TDC_F16) RESAMPLE8(f16, f16); else RESAMPLE8(f16, bf16); }
        else { if (out_dtype == TDC_F16) RESAMPLE8(bf16, f16); else RESAMPLE8(bf16, bf16); }
#undef RESAMPLE8
        return (int)hipGetLastError();
    }
    const dim3 grid(n_out * n_out, B);
#define RESAMPLE(TI, TO_) hipLaunchKernelGGL((resample_kernel<TI, TO_>), grid, dim3(256), 0, st, x, x_f32, ldx, tok_off, \
                                             n_in, (TO_*)y, ldy, n_out, idx0, idx1, frac, cols)
    if (dtype == TDC_F16) { if (out_dtype == TDC_F16) RESAMPLE(f16, f16); else RESAMPLE(f16, bf16); }
    else { if (out_dtype == TDC_F16) RESAMPLE(bf16, f16); else RESAMPLE(bf16, bf16); }
#undef RESAMPLE
    return (int)hipGetLastError();
}

extern "C" int tdc_frame_cossim(const void* f, long long n, int T, float* sims, float* scratch, int dtype,
                                void* stream) {
    if (!f || !sims || !scratch || T < 2 || n <= 0 || (n % 8) != 0) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(dtype, hipLaunchKernelGGL(cossim_partial<TT>, dim3(CS_CHUNKS, T), dim3(256), 0, st, (const TT*)f, n, T,
                                       scratch));
    hipLaunchKernelGGL(cossim_final, dim3((T + 255) / 256), dim3(256), 0, st, scratch, T, sims);
    return (int)hipGetLastError();
}
extern "C" size_t tdc_frame_cossim_scratch_floats(int T) { return (size_t)T * CS_CHUNKS * 2; }

extern "C" int tdc_token_mean(const void* x, int P, int ld, void* y, int B, int dtype, void* stream) {
    if (!x || !y || B <= 0 || P <= 0 || ld <= 0 || (ld & 7) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(dtype, hipLaunchKernelGGL(token_mean_kernel<TT>, dim3(B), dim3(256), 0, st, (const TT*)x, P, ld, (TT*)y));
    return (int)hipGetLastError();
}

extern "C" int tdc_adaptive_pool_tokens(const void* x, int N, int frame_rows, int ld, void* y, int K, int B,
                                        const int* src_row, int dtype, void* stream) {
    if (!x || !y || B <= 0 || N <= 0 || K <= 0 || frame_rows < N) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(dtype, hipLaunchKernelGGL(adaptive_pool_kernel<TT>, dim3(K, B), dim3(256), 0, st, (const TT*)x, N,
                                       frame_rows, ld, (TT*)y, K, src_row));
    return (int)hipGetLastError();
}

extern "C" int tdc_gather_rows(const tdc_gather_tables* t, const int* src, void* out, int ldo, int n, int cols,
                               int dtype, void* stream) {
    if (!t || !src || !out || n <= 0) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(dtype, hipLaunchKernelGGL(gather_kernel<TT>, dim3(n), dim3(256), 0, st, *t, src, (TT*)out, ldo, n, cols));
    return (int)hipGetLastError();
}

extern "C" int tdc_sva_attention(const tdc_sva_attn_desc* d, void* stream) {
    if (!d || !d->q || !d->kv[0] || !d->mask || !d->out) return TDC_E_BADARG;
    if (d->n_towers < 1 || d->n_towers > 2 || d->n_towers * d->r * d->r > 8 || d->heads > 64 ||
        d->dim % d->heads != 0)
        return TDC_E_BADARG;
    SvaArgs a;
    a.q = d->q; a.ldq = d->ldq; a.kv[0] = d->kv[0]; a.kv[1] = d->kv[1]; a.ldkv = d->ldkv; a.mask = d->mask;
    a.out = d->out; a.ldo = d->ldo; a.T = d->T; a.side = d->side; a.r = d->r; a.n_towers = d->n_towers;
    a.dim = d->dim; a.heads = d->heads;
    hipStream_t st = (hipStream_t)stream;
    const int nq = d->T * d->side * d->side;
    // the 16-byte-access kernel where the shape allows it (the path's own: C = 1024, 16 heads): 8-channel threads must tile whole
    // heads, a query's threads a workgroup, and every row start must be 16-byte aligned
    const int hd = d->dim / d->heads, lph = hd >> 3, tpq = d->dim >> 3;
    const bool fast = d->r == 2 && d->dim % 8 == 0 && hd % 8 == 0 && lph >= 1 && lph <= 64 && (lph & (lph - 1)) == 0 && tpq <= 256 &&
                      256 % tpq == 0 && tpq % lph == 0 && !(d->ldq & 7) && !(d->ldkv & 7) && !(d->ldo & 7) &&
                      !((uintptr_t)d->q & 15) && !((uintptr_t)d->kv[0] & 15) && !((uintptr_t)d->out & 15) &&
                      (d->n_towers < 2 || !((uintptr_t)d->kv[1] & 15));
    if (fast) {
        const int qpw = 256 / tpq;
        const dim3 grid((nq + qpw - 1) / qpw);
        if (d->n_towers == 2) { DISPATCH(d->dtype, hipLaunchKernelGGL((sva8_kernel<TT, 2, 2>), grid, dim3(256), 0, st, a, (long long)nq)); }
        else { DISPATCH(d->dtype, hipLaunchKernelGGL((sva8_kernel<TT, 1, 2>), grid, dim3(256), 0, st, a, (long long)nq)); }
        return (int)hipGetLastError();
    }
    DISPATCH(d->dtype, hipLaunchKernelGGL(sva_kernel<TT>, dim3(nq), dim3(256), 0, st, a));
    return (int)hipGetLastError();
}

extern "C" const char* tdc_version(void) { return "tdc_hip 0.1 (gfx950)"; }

extern "C" int tdc_device_info(int* cu_count, size_t* hbm_bytes) {
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return 0;
}

// every row of out [rows, ld] (16-bit) := row[0 .. ld): the SVA queries start as `vision_query` on every window of every
// frame (cambrian_arch.py:1018-1023).  ld % 8 == 0, 16-byte aligned.
namespace {
__global__ __launch_bounds__(256) void fill_rows_kernel(const u32x4* __restrict__ row, u32x4* __restrict__ out, int chunks,
                                                        long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = row[i % chunks];
}
}  // namespace

extern "C" int tdc_fill_rows(const void* row, void* out, int ld, int rows, void* stream) {
    if (!row || !out || ld <= 0 || rows <= 0 || (ld & 7) || ((uintptr_t)row & 15) || ((uintptr_t)out & 15)) return TDC_E_BADARG;
    const int chunks = ld / 8;
    const long long total = (long long)rows * chunks;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)row, (u32x4*)out, chunks, total);
    return (int)hipGetLastError();
}
