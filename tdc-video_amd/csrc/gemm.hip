// tdc_gemm: C = act(A W^T + bias) + res  on MFMA (gfx950), 16-bit operands, fp32 accumulate.
//
// Replaces every nn.Linear of the path (ViT qkv/out/fc1/fc2, projector MLPs, SVA k/v/q/o/proj, Q-Former
// q/k/v/dense/FFN, query_proj/vision_proj).  W is the nn.Linear weight as stored ([N, K], K contiguous), so both
// operands are K-contiguous "row" tiles in LDS.
//
// Structure (v1, "2-phase" of cdna_hip_programming.md T3/T4): 128x128x64 tile, 256 threads = 4 waves (2x2), each
// wave a 64x64 sub-tile = 4x4 MFMA 16x16x32 tiles.  Tiles are staged HBM -> LDS with global_load_lds_dwordx4
// (16 B/lane, no VGPR round trip) into a double buffer; the LDS image is lane-linear, so the XOR swizzle that makes
// the ds_read_b128 fragment reads bank-conflict free is applied on the per-lane SOURCE address (rule 21):
// physical 16-B chunk p of row r holds logical chunk p ^ (r & 7).
// The MFMA is issued "swapped" (A-operand = W rows, B-operand = activation rows) so that every lane ends up with
// 4 CONSECUTIVE output columns of one output row: bias / residual / store are 8- or 16-byte vector accesses.
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous range of tiles, row-major over
// (tile_m, tile_n): the tiles that share an activation row-panel run back to back on one L2.
#include "common.h"
#include "../../include/tdc_hip.h"
#include <stdio.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
    const void* A; const void* W; void* C; const float* bias; const void* res;
    int lda, ldw, ldc, ldres;
    int M, N, K;
    int out_f32, res_f32, act;
    RowMap am, cm, rm;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "contiguous chunk per XCD" remap (cdna_hip_programming.md T1): blocks b, b+8, ... share an XCD
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

template <class T>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // smem: A[2][16K] | W[2][16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave w issues 4 glds for A and 4 for W per K tile; instruction i covers rows 8i..8i+7
    const int srow = lane >> 3;                        // row within the 8-row group (== row & 7)
    const int schunk = (lane & 7) ^ srow;              // logical 16-B chunk loaded into physical chunk (lane & 7)
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = (wave * 4 + j) * 8 + srow;
        int am = m0 + r; if (am > p.M - 1) am = p.M - 1;
        int wn = n0 + r; if (wn > p.N - 1) wn = p.N - 1;
        a_src[j] = (const char*)p.A + (p.am(am) * (long long)p.lda + schunk * 8) * 2;
        w_src[j] = (const char*)p.W + ((long long)wn * p.ldw + schunk * 8) * 2;
    }
    auto stage = [&](int buf, int kt) {
        const long long koff = (long long)kt * BK * 2;
        char* la = smem + buf * TILE_BYTES + wave * 4 * 1024;
        char* lw = smem + 2 * TILE_BYTES + buf * TILE_BYTES + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[j] + koff), LDS_PTR(la + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(w_src[j] + koff), LDS_PTR(lw + j * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses (bytes within a tile): row r, logical chunk c -> r*128 + ((c ^ (r&7)) * 16)
    const int wm = wave >> 1, wn_ = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    int a_off[4], w_off[4];  // byte offset of (row, chunk g) for k-step 0; k-step 1 flips chunk bit 2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = wm * 64 + i * 16 + fr;
        int rw = wn_ * 64 + i * 16 + fr;
        a_off[i] = ra * 128 + ((g ^ (ra & 7)) << 4);
        w_off[i] = rw * 128 + ((g ^ (rw & 7)) << 4);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* la = smem + cur * TILE_BYTES;
        const char* lw = smem + 2 * TILE_BYTES + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 xa[4], xw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xa[i] = *(const v8*)(la + (a_off[i] ^ (ks << 6)));
                xw[i] = *(const v8*)(lw + (w_off[i] ^ (ks << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(xw[j], xa[i], acc[i][j]);
        }
        __syncthreads();  // drains the glds of tile kt+1 (vmcnt(0)) and fences the reads of buffer `cur`
    }

    // ---- epilogue: lane holds C[m = ... + fr][n = ... + 4g .. 4g+3] for each (i, j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
        const long long crow = p.cm(m);
        const long long rrow = p.res ? p.rm(m) : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn_ * 64 + j * 16 + g * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.bias) {
                f32x4 b = *(const f32x4*)(p.bias + n);
                v += b;
            }
            if (p.act == TDC_ACT_GELU_ERF) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            } else if (p.act == TDC_ACT_GELU_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            } else if (p.act == TDC_ACT_SWIGLU) {
                // columns are interleaved (x1_j, x2_j): two outputs per lane at column n/2
                float o0 = silu(v[0]) * v[1], o1 = silu(v[2]) * v[3];
                const int nc = n >> 1;
                if (p.out_f32) {
                    float* c = (float*)p.C + crow * p.ldc + nc;
                    c[0] = o0; c[1] = o1;
                } else {
                    T* c = (T*)p.C + crow * p.ldc + nc;
                    c[0] = (T)o0; c[1] = (T)o1;
                }
                continue;
            }
            if (p.res) {
                if (p.res_f32) {
                    f32x4 r = *(const f32x4*)((const float*)p.res + rrow * p.ldres + n);
                    v += r;
                } else {
                    typename VecOf<T>::v4 r = *(const typename VecOf<T>::v4*)((const T*)p.res + rrow * p.ldres + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                }
            }
            if (p.out_f32) {
                *(f32x4*)((float*)p.C + crow * p.ldc + n) = v;
            } else {
                typename VecOf<T>::v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
                *(typename VecOf<T>::v4*)((T*)p.C + crow * p.ldc + n) = o;
            }
        }
    }
}

template <class T>
int launch(const tdc_gemm_desc* d, hipStream_t st) {
    GemmArgs a;
    a.A = d->A; a.W = d->W; a.C = d->C; a.bias = d->bias; a.res = d->res;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.out_f32 = d->out_f32; a.res_f32 = d->res_f32; a.act = d->act;
    a.am = {d->a_map.seg, d->a_map.stride, d->a_map.off, d->a_map.inner};
    a.cm = {d->c_map.seg, d->c_map.stride, d->c_map.off, d->c_map.inner};
    a.rm = {d->r_map.seg, d->r_map.stride, d->r_map.off, d->r_map.inner};
    a.tiles_m = (d->M + BM - 1) / BM;
    a.tiles_n = (d->N + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          4 * TILE_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_kernel<T>, dim3(a.tiles_m * a.tiles_n), dim3(256), 4 * TILE_BYTES, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int tdc_gemm(const tdc_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->W || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0) return TDC_E_BADARG;
    if (d->K % BK != 0 || d->N % 4 != 0 || (d->act == TDC_ACT_SWIGLU && d->N % 8 != 0)) {
        fprintf(stderr, "[tdc_hip] tdc_gemm: K %% 64 / N %% 4 violated (M=%d N=%d K=%d)\n", d->M, d->N, d->K);
        return TDC_E_BADARG;
    }
    if ((d->lda % 8) || (d->ldw % 8) || (d->ldc % 4) || (d->res && (d->ldres % 4)) || d->lda < d->K || d->ldw < d->K) return TDC_E_BADARG;
    if (d->act == TDC_ACT_SWIGLU && d->res) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == TDC_F16) return launch<f16>(d, st);
    if (d->dtype == TDC_BF16) return launch<bf16>(d, st);
    return TDC_E_BADARG;
}
