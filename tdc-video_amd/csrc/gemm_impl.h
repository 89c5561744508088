// Kernel templates and launchers of tdc_gemm, shared by the two translation units that instantiate them: gemm.hip (16-bit
// operands) and gemm_fp8.hip (e4m3 operands) - compiled in parallel, each takes about a minute.
#pragma once
#include "common.h"
#include "../../include/tdc_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <mutex>
#include <unordered_map>

// Diagnostic switch of the epilogue (timing experiments only), set through tdc_gemm_set_debug() - never from the
// environment, so a stale variable cannot silently change what the production library computes.  Defined in gemm.hip.
extern int tdc_gemm_debug_mode;
extern int tdc_gemm_persist_grid_override;    // tdc_gemm_set_persistent_grid (gemm.hip): 0 = one workgroup per CU of the device

namespace {

#ifndef TDC_FAST32
#define TDC_FAST32 1
#endif
constexpr int kMaxDev = 64;      // per-device caches below are indexed by hipGetDevice()
inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    return dev;
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
    const void* A; const void* W; void* C; const float* bias; const void* res;
    int lda, ldw, ldc, ldres;
    int M, N, K;
    int out_f32, res_f32, act;
    RowMap am, cm, rm;
    int tiles_m, tiles_n;
    int debug;   // tdc_gemm_set_debug(): 1 = skip the epilogue, 2 = direct (un-staged) epilogue - timing experiments only
    int group_m;             // tile rows per group of the grouped tile order (tile_coords): 4 or 8, chosen per shape in launch()
    int c_pad8;              // tdc_gemm_desc.c_pad8: rows of C are writable up to round_up(N, 8) columns
    int ctype;               // TDC_F16 / TDC_BF16: 16-bit type of C and of a 16-bit res (tdc_gemm_desc.c16_dtype_p1; default = T)
    // LayerNorm fusion (see EpiOps / slot_stats_*): producer side x16 + ln_part, consumer side ln_stats + ln_c1
    void* x16; int ldx16; float* ln_part;
    const float* ln_stats; const float* ln_c1;
    // e4m3 output with analytic per-row scales (tdc_gemm_desc.out_fp8; fp8 operands + LayerNorm-fold operands only)
    int out_fp8; float* out_stats; float out_w2max, out_bmax, out_wscale;
#ifdef TDC_GEMM_DIAG
    unsigned long long* stamps;   // diagnostics build only (tools/gemm_stamps.cpp): 8 x u64 per workgroup
    unsigned long long* wstamps;  // ... per tile and WAVE (persistent kernel): 4 x u64 = epilogue start / issued / acknowledged
    int diag_mode;                // diagnostics build only, TDC_GEMM_DIAGMODE: 1 = no staging (stale LDS), 2 = no MFMAs
#endif
};

// In-kernel timeline stamps (s_memrealtime, 100 MHz) of wave 0 - compiled only into the diagnostics build
#ifdef TDC_GEMM_DIAG
#define TDC_STAMP(k)                                                                              \
    if (p.stamps && threadIdx.x == 0) {                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        p.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    }
#else
#define TDC_STAMP(k)
#endif

// lane id from v_mbcnt: a fresh value wherever it is needed, no register live from kernel entry (threadIdx.x & 63 kept across
// the K loop of the persistent kernel gets spilled)
__device__ __forceinline__ int fresh_lane() {
    int l;      // volatile: recomputed at every use, never hoisted out of the tile loop (neither it nor what is derived from it)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "contiguous chunk per XCD" remap (cdna_hip_programming.md T1): blocks b, b+8, ... share an XCD
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// ---- epilogue operands ----------------------------------------------------------------------------------------------
// Per-column / per-row operands of a wave's sub-tile in the MFMA layout: lane (fr, g) holds, per accumulator tile (i, j),
// row 16 i + fr and columns 16 j + 4 g .. +3.
//   bias[j]          - the nn.Linear bias (LayerNorm fusion: beta . W^T + b);
//   c1[j], mean[i], rstd[i] (FOLD) - LayerNorm fused into the consumer GEMM: A holds the RAW 16-bit rows x, W the
//                      gamma-folded weight W' = W diag(gamma); LN(x) W^T + b == rstd (x W'^T - mean c1) + bias with
//                      c1[n] = sum_k W'[n, k]; (mean, rstd) of row m come from ln_stats [M, 2] (tdc_ln_finalize).
// In the persistent kernel these values are already in registers, one column / row per lane (EpiLane, loaded one tile
// ahead so that no vector-memory load - whose in-order vmcnt wait would also wait for the next tile's staged operands -
// sits at the start of the epilogue) and are gathered with ds_bpermute; the other kernels load them here.
struct EpiLane {
    float bias, c1;        // column nbase + lane
    float mean0, rstd0;    // row mbase + lane
    float mean1, rstd1;    // row mbase + 64 + lane
};

__device__ __forceinline__ float lane_get(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane * 4, __builtin_bit_cast(int, v)));
}

// fp8 operands (gemm_fp8.hip's instantiations): the fold carries only the dequantisation scale rstd = s_a s_w - ln_c1 is
// not read and the first fma is gone (this epilogue is VALU-bound); `mean` is the row-norm bound out_fp8 uses.
#ifdef TDC_GEMM_FP8_TU
constexpr bool kScaleOnly = true;
#else
constexpr bool kScaleOnly = false;
#endif

template <int MI, int NJ, bool LANE, bool FOLD>
struct EpiOps {
    static constexpr bool kFold = FOLD;
    f32x4 bias[NJ], c1[NJ];
    float mean[MI], rstd[MI];
    __device__ __forceinline__ void load(const GemmArgs& p, int mbase, int nbase, int fr, int g, const EpiLane& el) {
        if (!FOLD) return;          // the bias went into the accumulators before the first MFMA (acc_init_bias)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (LANE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bias[j][e] = lane_get(el.bias, j * 16 + g * 4 + e);
                    if (!kScaleOnly) c1[j][e] = lane_get(el.c1, j * 16 + g * 4 + e);
                }
            } else {
                const int n = nbase + j * 16 + g * 4;
                bias[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                c1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (n < p.N) {
                    if (p.bias) bias[j] = *(const f32x4*)(p.bias + n);
                    if (!kScaleOnly) c1[j] = *(const f32x4*)(p.ln_c1 + n);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (LANE) {
                const int r = (i & 3) * 16 + fr;
                mean[i] = lane_get(i < 4 ? el.mean0 : el.mean1, r);
                rstd[i] = lane_get(i < 4 ? el.rstd0 : el.rstd1, r);
            } else {
                int m = mbase + i * 16 + fr;
                if (m > p.M - 1) m = p.M - 1;
                const float2 st = *(const float2*)(p.ln_stats + 2 * (long long)m);
                mean[i] = st.x; rstd[i] = st.y;
            }
        }
    }
    // the linear output of accumulator tile (i, j).  Without the fold the accumulators started from the bias (below) and ARE
    // the linear output.  FOLD: two explicit fmas per element - the same instruction sequence in every kernel and layout (a row
    // must not depend on which kernel computed it)
    __device__ __forceinline__ f32x4 lin(f32x4 acc, int i, int j) const {
        if (!FOLD) return acc;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            r[e] = kScaleOnly ? __builtin_fmaf(rstd[i], acc[e], bias[j][e])
                              : __builtin_fmaf(rstd[i], __builtin_fmaf(-mean[i], c1[j][e], acc[e]), bias[j][e]);
        return r;
    }
};

// ---- the bias is the accumulators' initial value -------------------------------------------------------------------------
// C = A W^T + b with the fp32 accumulators of column n starting at b[n] instead of 0: the epilogue's 128 `acc + bias` adds per
// lane and tile (and, in the persistent kernel, the 16 ds_bpermute that gathered the bias behind the main loop) are gone, at
// the price of moving b[n] instead of 0 into the accumulators - the same number of moves.  EVERY kernel does this (a row must
// not depend on the kernel that computed it); the LayerNorm-fold / fp8-scale forms (ln_stats != NULL) need the bare product
// and start from 0, their bias enters in EpiOps::lin.  Lane (fr, g) holds columns nbase + 16 j + 4 g .. + 3 of tile (i, j).
template <int NJ>
__device__ __forceinline__ void bias_cols_load(f32x4 (&b)[NJ], const GemmArgs& p, int nbase, int g) {
    const bool on = p.bias != nullptr && p.ln_stats == nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nbase + j * 16 + g * 4;
        b[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (on && n < p.N) b[j] = *(const f32x4*)(p.bias + n);
    }
}
template <int MI, int NJ>
__device__ __forceinline__ void acc_init_bias(f32x4 (&acc)[MI][NJ], const f32x4 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = b[j];
}

// ---- LayerNorm fusion, producer side: (mean, M2) of one row x 64-column slot --------------------------------------------
// The fp32 residual-stream GEMMs (out-projection, fc2) also write the updated row as 16-bit (x16, the next GEMM's A
// operand) and, per row and 64-column slot, the slot's mean and sum of squared deviations; tdc_ln_finalize combines the
// N/64 slots of a row (Chan) into (mean, rstd).  The two layouts below run the SAME balanced tree over the 16 four-column
// groups c = 0..15 of a slot (pairs c^1, c^2, c^4, c^8 in that order, no fp contraction), so the partials - like every
// other output - are bit-identical whichever kernel computes a row.
#pragma clang fp contract(off)
// staged layout: the 16 lanes of a row segment hold groups c = lane & 15 (DPP exchanges inside the 16-lane row)
__device__ __forceinline__ float row16_xch(float v, int level) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    if (level == 0) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if (level == 1) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if (level == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false);                  // row_mirror
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ void slot_stats_row16(f32x4 v, float& mean, float& m2) {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int l = 0; l < 4; ++l) s = s + row16_xch(s, l);
    mean = s * 0.015625f;
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int l = 0; l < 4; ++l) q = q + row16_xch(q, l);
    m2 = q;
}
// 16-bit read-modify-write layout (epi_staged_rmw16): the 8 lanes of a row hold groups c = 2 (lane & 7) and c + 1 - the pair
// c ^ 1 sits in the lane, c ^ 2, c ^ 4, c ^ 8 are the lanes l ^ 1, l ^ 2, l ^ 4 (quad_perm, quad_perm, row_half_mirror)
__device__ __forceinline__ void slot_stats_row8(f32x4 a, f32x4 b, float& mean, float& m2) {
    float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
#pragma unroll
    for (int l = 0; l < 3; ++l) s = s + row16_xch(s, l);
    mean = s * 0.015625f;
    const float d0 = a[0] - mean, d1 = a[1] - mean, d2 = a[2] - mean, d3 = a[3] - mean;
    const float e0 = b[0] - mean, e1 = b[1] - mean, e2 = b[2] - mean, e3 = b[3] - mean;
    float q = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3));
#pragma unroll
    for (int l = 0; l < 3; ++l) q = q + row16_xch(q, l);
    m2 = q;
}
// MFMA layout: group c = 4 j + g: g across the lanes l ^ 16, l ^ 32, j across the four accumulator tiles of the row
__device__ __forceinline__ void slot_stats_mfma(const f32x4 (&v)[4], float& mean, float& m2) {
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j] = (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        s[j] = s[j] + __shfl_xor(s[j], 16);
        s[j] = s[j] + __shfl_xor(s[j], 32);
    }
    mean = ((s[0] + s[1]) + (s[2] + s[3])) * 0.015625f;
    float q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d0 = v[j][0] - mean, d1 = v[j][1] - mean, d2 = v[j][2] - mean, d3 = v[j][3] - mean;
        q[j] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        q[j] = q[j] + __shfl_xor(q[j], 16);
        q[j] = q[j] + __shfl_xor(q[j], 32);
    }
    m2 = (q[0] + q[1]) + (q[2] + q[3]);
}
#pragma clang fp contract(fast)

// ---- epilogue, MFMA layout (128^2 kernel; fallback of the 256^2 kernels) ---------------------------------------------------
// Each lane holds, per (i, j) accumulator tile, 4 consecutive output columns n..n+3 of ONE output row m.  The epilogue
// is specialised at compile time on (activation, residual kind, output type, LayerNorm fusion): the run-time flags select
// one of the branch-free instantiations once per kernel.
// TC: 16-bit type of C / of a 16-bit residual (T unless tdc_gemm_desc.c16_dtype_p1 says otherwise)
template <class T, int ACT, int RES, bool OUTF32, class TC = T>
__device__ __forceinline__ void epi_store(const GemmArgs& p, f32x4 v, long long crow, long long rrow, int n) {
    if (ACT == TDC_ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    } else if (ACT == TDC_ACT_GELU_TANH) {
        v = gelu_tanh4(v);
    } else if (ACT == TDC_ACT_SWIGLU) {
        // columns are interleaved (x1_j, x2_j): two outputs per lane at column n/2
        const f32x2_t sg = swiglu2(v);
        const float o0 = sg[0], o1 = sg[1];
        const int nc = n >> 1;
        if (OUTF32) {
            float* c = (float*)p.C + crow * p.ldc + nc;
            c[0] = o0; c[1] = o1;
        } else {
            typedef typename VecOf<T>::v2 v2;
            *(v2*)((T*)p.C + crow * p.ldc + nc) = cvt2<T>(o0, o1);
        }
        return;
    }
    if (RES == 1) {
        v += *(const f32x4*)((const float*)p.res + rrow * p.ldres + n);
    } else if (RES == 2) {
        typename VecOf<TC>::v4 r = *(const typename VecOf<TC>::v4*)((const TC*)p.res + rrow * p.ldres + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    }
    if (OUTF32) {
        *(f32x4*)((float*)p.C + crow * p.ldc + n) = v;
    } else {
        *(typename VecOf<TC>::v4*)((TC*)p.C + crow * p.ldc + n) = cvt4<TC>(v);
    }
}

template <class T, int MI, int NJ, int ACT, int RES, bool OUTF32, bool LB, bool FOLD, class TC = T>
__device__ __forceinline__ void epi_tile(const GemmArgs& p, f32x4 (&acc)[MI][NJ], int mbase, int nbase, int fr,
                                         int g, const EpiLane& el) {
    EpiOps<MI, NJ, LB, FOLD> ops;
    ops.load(p, mbase, nbase, fr, g, el);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mbase + i * 16 + fr;
        if (m < p.M) {
            const long long crow = p.cm(m);
            const long long rrow = RES ? p.rm(m) : 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nbase + j * 16 + g * 4;
                if (n < p.N) epi_store<T, ACT, RES, OUTF32, TC>(p, ops.lin(acc[i][j], i, j), crow, rrow, n);
            }
        }
    }
}

// fp32 residual stream update + 16-bit copy + per-slot LayerNorm partials (N % 64 == 0, identity c_map / r_map)
template <class T, int MI, bool LB>
__device__ __forceinline__ void epi_tile_emit(const GemmArgs& p, f32x4 (&acc)[MI][4], int mbase, int nbase, int fr,
                                              int g, const EpiLane& el) {
    typedef typename VecOf<T>::v4 v4;
    if (nbase >= p.N) return;                       // wave-uniform
    EpiOps<MI, 4, LB, false> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    const int slot = nbase >> 6;      // partials are slot-major: ln_part[slot][M][2]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mbase + i * 16 + fr;
        const int mc = m < p.M ? m : p.M - 1;       // clamped rows compute (the exchanges need every lane), never store
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nbase + j * 16 + g * 4;
            v[j] = ops.lin(acc[i][j], i, j) + *(const f32x4*)((const float*)p.res + (long long)mc * p.ldres + n);
        }
        float mean, m2;
        slot_stats_mfma(v, mean, m2);
        if (m < p.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nbase + j * 16 + g * 4;
                *(f32x4*)((float*)p.C + (long long)m * p.ldc + n) = v[j];
                *(v4*)((T*)p.x16 + (long long)m * p.ldx16 + n) = cvt4<T>(v[j]);
            }
            if (g == 0) *(float2*)(p.ln_part + 2 * ((long long)slot * p.M + m)) = make_float2(mean, m2);
        }
    }
}

// 16-bit residual stream update (C = TC(acc + bias + float(res)), in place when C == res) + per-slot LayerNorm partials of the
// fp32 sums (ln_part != NULL, x16 == NULL; N % 64 == 0, identity c_map / r_map): the MFMA-layout twin of epi_staged_rmw16<EMIT>
template <class T, int MI, bool LB, class TC>
__device__ __forceinline__ void epi_tile_emit16(const GemmArgs& p, f32x4 (&acc)[MI][4], int mbase, int nbase, int fr,
                                                int g, const EpiLane& el) {
    typedef typename VecOf<TC>::v4 v4c;
    if (nbase >= p.N) return;                       // wave-uniform
    EpiOps<MI, 4, LB, false> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    const int slot = nbase >> 6;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mbase + i * 16 + fr;
        const int mc = m < p.M ? m : p.M - 1;       // clamped rows compute (the exchanges need every lane), never store
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nbase + j * 16 + g * 4;
            const v4c r = *(const v4c*)((const TC*)p.res + (long long)mc * p.ldres + n);
            v[j] = ops.lin(acc[i][j], i, j);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] += (float)r[e];
        }
        float mean, m2;
        slot_stats_mfma(v, mean, m2);
        if (m < p.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nbase + j * 16 + g * 4;
                *(v4c*)((TC*)p.C + (long long)m * p.ldc + n) = cvt4<TC>(v[j]);
            }
            if (g == 0) *(float2*)(p.ln_part + 2 * ((long long)slot * p.M + m)) = make_float2(mean, m2);
        }
    }
}

// ---- e4m3 output with analytic per-row scales (tdc_gemm_desc.out_fp8) -----------------------------------------------------
// Row m of act(A W^T + b) is bounded by B = rstd * ||a8||_2 * max_n ||w8_n||_2 + max|b| (Cauchy-Schwarz on the quantised
// operands; ops.mean carries the row norm, ops.rstd the row's s_a * s_w), squared for SwiGLU; the row goes out as
// e4m3(v * 448 / B^p).  A loose bound costs nothing: e4m3 keeps its 3 mantissa bits over 15 binades.
template <int MI, int NJ, bool LB, bool SW>
struct RowScale8 {
    float inv[MI], out[MI];
    __device__ __forceinline__ void set(const GemmArgs& p, const EpiOps<MI, NJ, LB, true>& ops) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float b = __builtin_fmaf(ops.rstd[i] * ops.mean[i], p.out_w2max, p.out_bmax);
            if (SW) b = b * b;
            b = fmaxf(b, 1e-30f);
            inv[i] = 448.0f / b;
            out[i] = b * (1.0f / 448.0f) * p.out_wscale;
        }
    }
};
__device__ __forceinline__ int pack_fp8x4(f32x4 v) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
}
__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    if (act == TDC_ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    } else if (act == TDC_ACT_GELU_TANH) {
        v = gelu_tanh4(v);
    }
    return v;
}

// MFMA-layout form (128^2 kernel, fallback): 4 bytes (2 for SwiGLU) per lane and accumulator tile
template <class T, int MI, int NJ, int ACT, bool LB>
__device__ __forceinline__ void epi_tile8(const GemmArgs& p, f32x4 (&acc)[MI][NJ], int mbase, int nbase, int fr, int g,
                                          const EpiLane& el) {
    EpiOps<MI, NJ, LB, true> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    RowScale8<MI, NJ, LB, ACT == TDC_ACT_SWIGLU> rs;
    rs.set(p, ops);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mbase + i * 16 + fr;
        if (m >= p.M) continue;
        unsigned char* crow = (unsigned char*)p.C + (long long)m * p.ldc;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nbase + j * 16 + g * 4;
            if (n >= p.N) continue;
            const f32x4 v = ops.lin(acc[i][j], i, j);
            if (ACT == TDC_ACT_SWIGLU) {
                const f32x2_t sg = swiglu2(v) * rs.inv[i];
                const int w = __builtin_amdgcn_cvt_pk_fp8_f32(sg[0], sg[1], 0, false);
                *(unsigned short*)(crow + (n >> 1)) = (unsigned short)w;
            } else {
                *(int*)(crow + n) = pack_fp8x4(act4(v, ACT) * rs.inv[i]);
            }
        }
        if (nbase == 0 && g == 0) *(float2*)(p.out_stats + 2 * (long long)m) = make_float2(0.f, rs.out[i]);
    }
}

template <class T, int MI, int NJ, bool LB, bool FOLD>
__device__ __forceinline__ void epilogue_f(const GemmArgs& p, f32x4 (&acc)[MI][NJ], int mbase, int nbase, int fr, int g,
                                           const EpiLane& el) {
    const int res = p.res ? (p.res_f32 ? 1 : 2) : 0;
    if constexpr (FOLD) {
        if (p.out_fp8) {
            if (p.act == TDC_ACT_GELU_ERF) epi_tile8<T, MI, NJ, TDC_ACT_GELU_ERF, LB>(p, acc, mbase, nbase, fr, g, el);
            else if (p.act == TDC_ACT_GELU_TANH) epi_tile8<T, MI, NJ, TDC_ACT_GELU_TANH, LB>(p, acc, mbase, nbase, fr, g, el);
            else if (p.act == TDC_ACT_SWIGLU) epi_tile8<T, MI, NJ, TDC_ACT_SWIGLU, LB>(p, acc, mbase, nbase, fr, g, el);
            else epi_tile8<T, MI, NJ, TDC_ACT_NONE, LB>(p, acc, mbase, nbase, fr, g, el);
            return;
        }
    }
    if (p.act == TDC_ACT_GELU_ERF) epi_tile<T, MI, NJ, TDC_ACT_GELU_ERF, 0, false, LB, FOLD>(p, acc, mbase, nbase, fr, g, el);
    else if (p.act == TDC_ACT_GELU_TANH) epi_tile<T, MI, NJ, TDC_ACT_GELU_TANH, 0, false, LB, FOLD>(p, acc, mbase, nbase, fr, g, el);
    else if (p.act == TDC_ACT_SWIGLU) epi_tile<T, MI, NJ, TDC_ACT_SWIGLU, 0, false, LB, FOLD>(p, acc, mbase, nbase, fr, g, el);
    else if (!p.out_f32 && res == 0) epi_tile<T, MI, NJ, 0, 0, false, LB, FOLD>(p, acc, mbase, nbase, fr, g, el);
    else if (p.out_f32 && res == 1 && FOLD) epi_tile<T, MI, NJ, 0, 1, true, LB, true>(p, acc, mbase, nbase, fr, g, el);
    else if (!p.out_f32 && res == 2 && FOLD) epi_tile<T, MI, NJ, 0, 2, false, LB, true>(p, acc, mbase, nbase, fr, g, el);   // fp8 operands: T = the stream's type
    else if (FOLD) return;          // (host-checked: the fold exists for 16-bit outputs and for the residual-stream updates)
    else if (p.out_f32) {
        if (res == 1) epi_tile<T, MI, NJ, 0, 1, true, LB, false>(p, acc, mbase, nbase, fr, g, el);
        else if (res == 2) epi_tile<T, MI, NJ, 0, 2, true, LB, false>(p, acc, mbase, nbase, fr, g, el);
        else epi_tile<T, MI, NJ, 0, 0, true, LB, false>(p, acc, mbase, nbase, fr, g, el);
    } else {
        // 16-bit output with a residual: C / a 16-bit res may be of the other 16-bit type (GemmArgs::ctype)
        typedef typename std::conditional<std::is_same<T, f16>::value, bf16, f16>::type TX;
        const bool other = p.ctype != (std::is_same<T, f16>::value ? TDC_F16 : TDC_BF16);
        if (res == 1) {
            if (other) epi_tile<T, MI, NJ, 0, 1, false, LB, false, TX>(p, acc, mbase, nbase, fr, g, el);
            else epi_tile<T, MI, NJ, 0, 1, false, LB, false>(p, acc, mbase, nbase, fr, g, el);
        } else {
            if (other) epi_tile<T, MI, NJ, 0, 2, false, LB, false, TX>(p, acc, mbase, nbase, fr, g, el);
            else epi_tile<T, MI, NJ, 0, 2, false, LB, false>(p, acc, mbase, nbase, fr, g, el);
        }
    }
}

template <class T, int MI, int NJ, bool LB = false>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x4 (&acc)[MI][NJ], int mbase, int nbase, int fr, int g,
                                         const EpiLane& el = EpiLane()) {
    if (NJ == 4 && p.x16) { epi_tile_emit<T, MI, LB>(p, (f32x4 (&)[MI][4])acc, mbase, nbase, fr, g, el); return; }
    if (NJ == 4 && p.ln_part) {      // (x16 == NULL) the fold's producer over a 16-bit stream
        if (p.ctype == TDC_F16) epi_tile_emit16<T, MI, LB, f16>(p, (f32x4 (&)[MI][4])acc, mbase, nbase, fr, g, el);
        else epi_tile_emit16<T, MI, LB, bf16>(p, (f32x4 (&)[MI][4])acc, mbase, nbase, fr, g, el);
        return;
    }
    if (!LB && p.ln_stats) epilogue_f<T, MI, NJ, LB, true>(p, acc, mbase, nbase, fr, g, el);   // lane-held: LNF kernel
    else epilogue_f<T, MI, NJ, LB, false>(p, acc, mbase, nbase, fr, g, el);
}

// tile id -> (tile_m, tile_n), "grouped" order: ids walk GROUP_M tile rows column by column before moving to the next
// group of rows.  The 32 workgroups that one XCD runs concurrently (consecutive ids after xcd_remap) then cover an
// 8 x 4 patch of tiles: 8 activation panels + 4 weight panels stream through that XCD's L2 instead of 1 + 32
// (PMC, N = 8192: 9.6x the algorithmic bytes crossed the fabric with the plain row-major order).
constexpr int GROUP_M_DEFAULT = 8;
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M = GROUP_M_DEFAULT) {
    const int per_group = GROUP_M * tiles_n;
    const int grp = id / per_group, within = id - grp * per_group;
    const int first = grp * GROUP_M;
    const int rows = (tiles_m - first < GROUP_M) ? tiles_m - first : GROUP_M;
    tn = within / rows;
    tm = first + (within - tn * rows);
}

// FP8: A and W hold e4m3 bytes; all addressing below stays in 2-byte units (the host passes K / 2, lda / 2, ldw / 2), only
// the MFMA differs (common.h: mma16).  T remains the 16-bit output / residual type.
template <class T, bool FP8>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // smem: A[2][16K] | W[2][16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = p.tiles_m * p.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    int tm, tn;
    tile_coords(id, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
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
    {   // the accumulators start from the bias (acc_init_bias): its load is in flight beside the first staged tile
        f32x4 bcol[4];
        bias_cols_load<4>(bcol, p, n0 + wn_ * 64, g);
        stage(0, 0);
        acc_init_bias<4, 4>(acc, bcol);
    }
    const int nk = p.K / BK;
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* la = smem + cur * TILE_BYTES;
        const char* lw = smem + 2 * TILE_BYTES + cur * TILE_BYTES;
        if constexpr (FP8) {        // both halves of the 128-byte K tile in one 16x16x128 MFMA (common.h)
            v8 xa[2][4], xw[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[ks][i] = *(const v8*)(la + (a_off[i] ^ (ks << 6)));
                    xw[ks][i] = *(const v8*)(lw + (w_off[i] ^ (ks << 6)));
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma128_fp8(xw[0][j], xw[1][j], xa[0][i], xa[1][i], acc[i][j]);
        } else {
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
        }
        __syncthreads();  // drains the glds of tile kt+1 (vmcnt(0)) and fences the reads of buffer `cur`
    }

    // ---- epilogue: lane holds C[m = m0 + wm*64 + 16 i + fr][n = n0 + wn*64 + 16 j + 4g .. +3]
    epilogue<T, 4, 4>(p, acc, m0 + wm * 64, n0 + wn_ * 64, fr, g);
}

// The staged epilogues carry an interior-tile fast path beside their general form.  With the LayerNorm fold both need
// `ops.lin(acc[i][j])`; the same expression on both sides of a branch is hoisted above it - here above the whole run-time
// dispatch of epilogue variants - into NEW registers: 128 more live values next to the 128 accumulators (measured: 300-600
// spilled VGPRs).  The fast paths therefore work on an opaque copy of the (16-register) column operands, which makes their
// arithmetic theirs alone.  Without the fold lin() is the identity (the bias is in the accumulators) and there is nothing to
// hoist - making the ACCUMULATORS opaque instead was tried and spills 1400-5200 registers.
template <class OPS>
__device__ __forceinline__ OPS opaque_ops(OPS o) {
    if (OPS::kFold) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(o.bias[j]));
    }
    return o;
}

// ---- LDS-staged epilogue of the 256^2 kernels -------------------------------------------------------------------------
// Row-per-lane stores straight from the MFMA layout touch 16 cache lines with 8 B each per instruction and are
// store-ISSUE bound (~7 B/clk/CU, cdna_hip_programming.md T21): the 128 KiB C tile cost ~9 us per workgroup, 25 % of a
// K=1152 GEMM.  Instead every wave transposes its 128x64 sub-tile through LDS (its own 16 KiB of the idle pipeline
// buffers; 4 KiB beside the live pipeline in the persistent kernel: ROWS rows per pass) and stores whole rows: one
// instruction = 8 rows x 128 B (16-bit out) or 4 rows x 256 B (fp32 out), 16 B per lane.  Bias / LayerNorm fold /
// activation run before staging (MFMA layout); the fp32 residual add runs after it, on full 256-B row segments.  XOR
// swizzle of the 16-B chunk with the row keeps both sides (nearly) conflict-free.
template <class T, int ACT, int RES, int ROWS, bool LB, bool FOLD>
__device__ __forceinline__ void epi_staged16(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase, int nbase,
                                             int lane, const EpiLane& el) {
    // ROWS = rows of the wave's 128x64 sub-tile staged per pass (128-byte rows): 128 (16 KiB region) or 32 (4 KiB)
    typedef typename VecOf<T>::v4 v4;
    typedef typename VecOf<T>::v8 v8;
    const int fr = lane & 15, g = lane >> 4;
    EpiOps<8, 4, LB, FOLD> ops;
    ops.load(p, mbase, nbase, fr, g, el);
#ifdef TDC_GEMM_DIAG     // decomposition experiments (tools/run_gemm_epi_parts.sh): 8 = no global stores, 16 = no LDS staging
    const bool diag_no_store = p.debug & 8, diag_no_stage = p.debug & 16;
#else
    constexpr bool diag_no_store = false, diag_no_stage = false;
#endif
    // Interior sub-tile with an identity c_map (every tile of the tower GEMMs but the last row / column of tiles): one running
    // store pointer, no per-row bounds test and no row-map arithmetic, so a pass is straight-line code - its LDS reads are
    // all in flight before the first store waits for one.  In the general form below every store sits in its own
    // exec-masked block behind its own LDS read and (possibly) a row-map division: ~250 cycles of latency per 1-KiB store
    // instruction, which made a wave's epilogue 4.3 us long (tools/run_gemm_epi_parts.sh).
    if (RES == 0 && !diag_no_store && !diag_no_stage && p.cm.seg == 0 && mbase + 128 <= p.M && nbase + 64 <= p.N) {
        const auto fops = opaque_ops(ops);
        const int sw = fr & 7, rrow = lane >> 3, rk = lane & 7;
        char* wbase = region + fr * 128 + (g & 1) * 8;
        const char* rbase = region + rrow * 128 + ((rk ^ rrow) << 4);
        T* cptr = (T*)p.C + (long long)(mbase + rrow) * p.ldc + nbase + rk * 8;
        const long long step = 8ll * p.ldc;
#pragma unroll
        for (int pass = 0; pass < 128 / ROWS; ++pass) {
#pragma unroll
            for (int ii = 0; ii < ROWS / 16; ++ii) {
                const int i = pass * (ROWS / 16) + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = fops.lin(acc[i][j], i, j);
                    if (ACT == TDC_ACT_GELU_ERF) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    } else if (ACT == TDC_ACT_GELU_TANH) {
                        v = gelu_tanh4(v);
                    }
                    *(v4*)(wbase + ii * 2048 + (((j * 2 + (g >> 1)) ^ sw) << 4)) = cvt4<T>(v);
                }
            }
            v8 val[ROWS / 8];
#pragma unroll
            for (int q = 0; q < ROWS / 8; ++q) val[q] = *(const v8*)(rbase + q * 1024);
#pragma unroll
            for (int q = 0; q < ROWS / 8; ++q) {
                __builtin_nontemporal_store(val[q], (v8*)cptr);      // plain (cached) stores measure the same
                cptr += step;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 128 / ROWS; ++pass) {
#pragma unroll
        for (int ii = 0; ii < ROWS / 16; ++ii) {
            const int i = pass * (ROWS / 16) + ii;
            const int r = ii * 16 + fr;                 // row within the staging region
            const int m = mbase + i * 16 + fr;
            long long rrow = 0;
            if (RES == 2) rrow = p.rm(m < p.M ? m : p.M - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = ops.lin(acc[i][j], i, j);
                if (ACT == TDC_ACT_GELU_ERF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                } else if (ACT == TDC_ACT_GELU_TANH) {
                    v = gelu_tanh4(v);
                }
                if (RES == 2) {   // 16-bit residual: add before the single rounding to T
                    const int n = nbase + j * 16 + g * 4;
                    if (n < p.N) {
                        v4 rr = *(const v4*)((const T*)p.res + rrow * p.ldres + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
                    }
                }
                const v4 o = cvt4<T>(v);
                const int chunk = (j * 2 + (g >> 1)) ^ (r & 7);
                if (!diag_no_stage) *(v4*)(region + r * 128 + chunk * 16 + (g & 1) * 8) = o;
                else asm volatile("" ::"v"(o));
            }
        }
        // same wave, in-order LDS queue: the reads below observe the writes above (and the next pass's writes follow
        // these reads)
#pragma unroll
        for (int q = 0; q < ROWS / 8; ++q) {
            const int r = q * 8 + (lane >> 3), k = lane & 7;
            v8 val;
            if (!diag_no_stage) val = *(const v8*)(region + r * 128 + ((k ^ (r & 7)) << 4));
            else { const f32x4 a0 = acc[pass][q & 3]; val = __builtin_bit_cast(v8, a0); }     // any register data, same addresses
            const int m = mbase + pass * ROWS + r, n = nbase + k * 8;
            if (diag_no_store) { asm volatile("" ::"v"(val)); continue; }
            if (m < p.M && n < p.N) __builtin_nontemporal_store(val, (v8*)((T*)p.C + p.cm(m) * p.ldc + n));
        }
    }
}

// SwiGLU: columns are interleaved (x1_j, x2_j); each lane produces 2 outputs per accumulator tile, the wave's 128x64
// sub-tile becomes 128 rows x 32 outputs (64-byte rows, 8 KiB).  Staging turns 32 four-byte stores per lane (16 rows
// x 16 B per instruction) into 8 sixteen-byte stores (16 rows x 64 B per instruction).
template <class T, int ROWS, bool LB, bool FOLD>
__device__ __forceinline__ void epi_staged_swiglu(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase,
                                                  int nbase, int lane, const EpiLane& el) {
    typedef __attribute__((ext_vector_type(2))) T v2;
    typedef typename VecOf<T>::v8 v8;
    const int fr = lane & 15, g = lane >> 4;
    EpiOps<8, 4, LB, FOLD> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    if (p.cm.seg == 0 && mbase + 128 <= p.M && nbase + 64 <= p.N) {      // interior sub-tile: see epi_staged16
        const auto fops = opaque_ops(ops);
        const int rrow = lane >> 2, rk = lane & 3;
        char* wbase = region + fr * 64 + g * 4;
        const int wsw = (fr >> 2) & 3;                                     // (r >> 2) & 3 with r = 16 ii + fr
        const char* rbase = region + rrow * 64 + ((rk ^ ((rrow >> 2) & 3)) << 4);
        T* cptr = (T*)p.C + (long long)(mbase + rrow) * p.ldc + (nbase >> 1) + rk * 8;
        const long long step = 16ll * p.ldc;
#pragma unroll
        for (int pass = 0; pass < 128 / ROWS; ++pass) {
#pragma unroll
            for (int ii = 0; ii < ROWS / 16; ++ii) {
                const int i = pass * (ROWS / 16) + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_t sg = swiglu2(fops.lin(acc[i][j], i, j));
                    *(v2*)(wbase + ii * 1024 + ((j ^ wsw) << 4)) = cvt2<T>(sg[0], sg[1]);
                }
            }
            v8 val[ROWS / 16];
#pragma unroll
            for (int q = 0; q < ROWS / 16; ++q) val[q] = *(const v8*)(rbase + q * 1024);
#pragma unroll
            for (int q = 0; q < ROWS / 16; ++q) {
                __builtin_nontemporal_store(val[q], (v8*)cptr);      // plain (cached) stores measure the same
                cptr += step;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 128 / ROWS; ++pass) {
#pragma unroll
        for (int ii = 0; ii < ROWS / 16; ++ii) {
            const int i = pass * (ROWS / 16) + ii;
            const int r = ii * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = ops.lin(acc[i][j], i, j);
                const f32x2_t sg = swiglu2(v);
                const v2 o = cvt2<T>(sg[0], sg[1]);
                // output column within the wave's 32: j*8 + 2g -> 16-B chunk j (4 per row), swizzled with the row
                const int chunk = j ^ ((r >> 2) & 3);
                *(v2*)(region + r * 64 + chunk * 16 + g * 4) = o;
            }
        }
#pragma unroll
        for (int q = 0; q < ROWS / 16; ++q) {
            const int r = q * 16 + (lane >> 2), k = lane & 3;
            const v8 val = *(const v8*)(region + r * 64 + ((k ^ ((r >> 2) & 3)) << 4));
            const int m = mbase + pass * ROWS + r, nc = (nbase >> 1) + k * 8;
            if (m < p.M && 2 * nc < p.N) __builtin_nontemporal_store(val, (v8*)((T*)p.C + p.cm(m) * p.ldc + nc));
        }
    }
}

// e4m3 output (out_fp8), staged: 4 bytes per lane and accumulator tile -> 64-byte rows of the wave's 128x64 sub-tile (the
// geometry of the SwiGLU staging: 16-B chunk j of row r at j ^ ((r >> 2) & 3)); read back 16 rows x 64 B per instruction.
// ROWS = rows per pass: 128 (8 KiB of the 16-KiB region) or 64 (4 KiB).
template <class T, int ACT, int ROWS, bool LB>
__device__ __forceinline__ void epi_staged8(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase, int nbase,
                                            int lane, const EpiLane& el) {
    const int fr = lane & 15, g = lane >> 4;
    EpiOps<8, 4, LB, true> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    RowScale8<8, 4, LB, false> rs;
    rs.set(p, ops);
#pragma unroll
    for (int pass = 0; pass < 128 / ROWS; ++pass) {
#pragma unroll
        for (int ii = 0; ii < ROWS / 16; ++ii) {
            const int i = pass * (ROWS / 16) + ii;
            const int r = ii * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = j ^ ((r >> 2) & 3);
                *(int*)(region + r * 64 + chunk * 16 + g * 4) = pack_fp8x4(act4(ops.lin(acc[i][j], i, j), ACT) * rs.inv[i]);
            }
            if (nbase == 0 && g == 0) {
                const int m = mbase + i * 16 + fr;
                if (m < p.M) *(float2*)(p.out_stats + 2 * (long long)m) = make_float2(0.f, rs.out[i]);
            }
        }
#pragma unroll
        for (int q = 0; q < ROWS / 16; ++q) {
            const int r = q * 16 + (lane >> 2), k = lane & 3;
            const u32x4 val = *(const u32x4*)(region + r * 64 + ((k ^ ((r >> 2) & 3)) << 4));
            const int m = mbase + pass * ROWS + r, n = nbase + k * 16;
            if (m < p.M && n < p.N) __builtin_nontemporal_store(val, (u32x4*)((unsigned char*)p.C + (long long)m * p.ldc + n));
        }
    }
}

// ... SwiGLU: 2 bytes per lane and accumulator tile -> 32-byte rows (4 KiB for the whole sub-tile), read back 32 rows x 32 B
template <class T, bool LB>
__device__ __forceinline__ void epi_staged8_swiglu(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase, int nbase,
                                                   int lane, const EpiLane& el) {
    const int fr = lane & 15, g = lane >> 4;
    EpiOps<8, 4, LB, true> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    RowScale8<8, 4, LB, true> rs;
    rs.set(p, ops);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = ops.lin(acc[i][j], i, j);
            const f32x2_t sg = swiglu2(v) * rs.inv[i];
            const int w = __builtin_amdgcn_cvt_pk_fp8_f32(sg[0], sg[1], 0, false);
            // output columns j*8 + 2g, +1 of the wave's 32: 16-B half (j >> 1) of the 32-byte row, swizzled with the row
            const int half = (j >> 1) ^ ((r >> 3) & 1);
            *(unsigned short*)(region + r * 32 + half * 16 + (j & 1) * 8 + g * 2) = (unsigned short)w;
        }
        if (nbase == 0 && g == 0) {
            const int m = mbase + r;
            if (m < p.M) *(float2*)(p.out_stats + 2 * (long long)m) = make_float2(0.f, rs.out[i]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * 32 + (lane >> 1), k = lane & 1;
        const u32x4 val = *(const u32x4*)(region + r * 32 + ((k ^ ((r >> 3) & 1)) << 4));
        const int m = mbase + r, nc = (nbase >> 1) + k * 16;
        if (m < p.M && 2 * nc < p.N) __builtin_nontemporal_store(val, (u32x4*)((unsigned char*)p.C + (long long)m * p.ldc + nc));
    }
}

// fp32 output (+ fp32 residual).  EMIT (LayerNorm fusion, producer side): the updated row also goes out as 16-bit
// (x16) together with the per-slot (mean, M2) partials - identity row maps, N % 64 == 0.
template <class T, int RES, int ROWS, bool LB, bool EMIT, bool FOLD = false, int RING = 8>
__device__ __forceinline__ void epi_staged32(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase, int nbase,
                                             int lane, const EpiLane& el) {
    // ROWS = rows staged per pass (256-byte fp32 rows): 64 (16 KiB region) or 16 (4 KiB).  FOLD: the row / column
    // operands of EpiOps (fp8 operands: the dequantisation scales of a residual-stream GEMM)
    typedef typename VecOf<T>::v4 v4;
    const int fr = lane & 15, g = lane >> 4;
    EpiOps<8, 4, LB, FOLD> ops;
    ops.load(p, mbase, nbase, fr, g, el);
#ifdef TDC_GEMM_DIAG     // 8 = no global stores, 16 = no LDS staging, 32 = no residual loads
    const bool diag_no_store = p.debug & 8, diag_no_stage = p.debug & 16, diag_no_res = p.debug & 32;
#else
    constexpr bool diag_no_store = false, diag_no_stage = false, diag_no_res = false;
#endif
    constexpr int NPASS = 128 / ROWS, QP = ROWS / 4;      // QP read-back instructions (4 rows x 256 B each) per pass
    // The residual loads run as a ring of RING loads ahead of the read-back, independent of the staging passes (they
    // touch no LDS); they come from clamped (always valid) addresses so that no branch - and no vmcnt(0) - separates them.
    if (TDC_FAST32 && !diag_no_store && !diag_no_stage && !diag_no_res && p.cm.seg == 0 && (RES == 0 || p.rm.seg == 0) &&
        mbase + 128 <= p.M && nbase + 64 <= p.N) {
        // interior sub-tile, identity row maps (see epi_staged16): running pointers for the residual loads and the stores, the
        // residual ring filled up front, a pass = 4 writes, 4 read-backs, 4 stores of straight-line code
        const auto fops = opaque_ops(ops);
        const int rrow = lane >> 4, rk = lane & 15;
        char* wbase = region + fr * 256;
        const char* rbase = region + rrow * 256;
        float* cptr = (float*)p.C + (long long)(mbase + rrow) * p.ldc + nbase + rk * 4;
        const float* rptr = (const float*)p.res + (long long)(mbase + rrow) * p.ldres + nbase + rk * 4;
        const long long cstep = 4ll * p.ldc, rstep = 4ll * p.ldres;
        // EMIT (LayerNorm fusion, producer side): the 16-bit row copy and the per-slot (mean, M2) partials of the same rows
        T* xptr = EMIT ? (T*)p.x16 + (long long)(mbase + rrow) * p.ldx16 + nbase + rk * 4 : nullptr;
        float* lptr = EMIT ? p.ln_part + 2 * ((long long)(nbase >> 6) * p.M + mbase + rrow) : nullptr;
        const long long xstep = 4ll * p.ldx16;
        // ring depth: 8 loads (32 VGPRs) ahead - with the 128 accumulator registers, a pass of read-backs and the lane-held
        // operands that is what fits without spilling; sched_barrier keeps the passes from being merged (the scheduler would
        // otherwise hoist every pass's LDS traffic and spill ~300 registers)
        constexpr int FR = RING < 8 ? RING : 8;
        f32x4 ring[FR];
        if (RES == 1) {
#pragma unroll
            for (int u = 0; u < FR; ++u) { ring[u] = *(const f32x4*)rptr; rptr += rstep; }
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int ii = 0; ii < ROWS / 16; ++ii) {
                const int i = pass * (ROWS / 16) + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(f32x4*)(wbase + ii * 4096 + (((j * 4 + g) ^ fr) << 4)) = fops.lin(acc[i][j], i, j);
            }
            f32x4 val[QP];
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const int r = q * 4 + rrow;                        // r & 15 = (4 q + rrow) & 15
                val[q] = *(const f32x4*)(rbase + q * 1024 + ((rk ^ (r & 15)) << 4));
            }
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const int u = pass * QP + q;
                if (RES == 1) {
                    val[q] += ring[u % FR];
                    if (u + FR < 32) { ring[u % FR] = *(const f32x4*)rptr; rptr += rstep; }
                }
                __builtin_nontemporal_store(val[q], (f32x4*)cptr);
                cptr += cstep;
                if (EMIT) {
                    float mean, m2;
                    slot_stats_row16(val[q], mean, m2);
                    *(v4*)xptr = cvt4<T>(val[q]);
                    xptr += xstep;
                    if (rk == 0) *(float2*)lptr = make_float2(mean, m2);
                    lptr += 8;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    f32x4 rr[RING];
    auto load_res = [&](int u) {                           // unit u = rows 4u .. 4u+3 of the wave's sub-tile
        const int k = lane & 15;
        int m = mbase + u * 4 + (lane >> 4), n = nbase + k * 4;
        if (m > p.M - 1) m = p.M - 1;
        if (n > p.N - 4) n = p.N - 4;
        if (diag_no_res) return (f32x4){0.f, 0.f, 0.f, 0.f};
        return *(const f32x4*)((const float*)p.res + p.rm(m) * p.ldres + n);
    };
    if (RES == 1) {
#pragma unroll
        for (int u = 0; u < RING; ++u) rr[u] = load_res(u);
    }
    const int slot = nbase >> 6;      // partials are slot-major: ln_part[slot][M][2]
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
        for (int ii = 0; ii < ROWS / 16; ++ii) {
            const int r = ii * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = (j * 4 + g) ^ (r & 15);
                const int i = pass * (ROWS / 16) + ii;
                if (!diag_no_stage) *(f32x4*)(region + r * 256 + chunk * 16) = ops.lin(acc[i][j], i, j);
            }
        }
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const int u = pass * QP + q;
            const int r = q * 4 + (lane >> 4), k = lane & 15;
            f32x4 val;
            if (!diag_no_stage) val = *(const f32x4*)(region + r * 256 + ((k ^ (r & 15)) << 4));
            else val = acc[pass & 7][q & 3];
            const int m = mbase + pass * ROWS + r, n = nbase + k * 4;
            if (RES == 1) {
                val += rr[u % RING];
                if (u + RING < 32) rr[u % RING] = load_res(u + RING);
            }
            if (diag_no_store) { asm volatile("" ::"v"(val)); }
            else if (m < p.M && n < p.N) __builtin_nontemporal_store(val, (f32x4*)((float*)p.C + p.cm(m) * p.ldc + n));
            if (EMIT) {
                float mean, m2;
                slot_stats_row16(val, mean, m2);          // every lane takes part (rows beyond M: clamped duplicates)
                if (m < p.M) {
                    *(v4*)((T*)p.x16 + (long long)m * p.ldx16 + n) = cvt4<T>(val);
                    if (k == 0) *(float2*)(p.ln_part + 2 * ((long long)slot * p.M + m)) = make_float2(mean, m2);
                }
            }
        }
    }
}

// 16-bit output + 16-bit residual (C = TC(acc + bias + float(res)), one rounding; in place when C == res): the read-modify-write
// of a 16-bit residual stream - 4 B per element against the 8 B of the fp32 form - and the Q-Former's / connector's 16-bit
// residual GEMMs.  The sub-tile is staged as fp32 (256-byte rows, the epi_staged32 image: chunk c of row r at c ^ (r & 15)); a
// lane reads back 8 consecutive columns of one row (two 16-byte LDS reads), adds 8 residual values it loaded with ONE 16-byte
// load, and stores 16 bytes: one wave instruction = 8 rows x 128 B on either side.  The residual loads run RING units (8 rows)
// ahead of their use.  ROWS = rows staged per pass (16: 4 KiB region, 32: 8 KiB of the 16-KiB region).
#ifndef TDC_RMW16_RING
#define TDC_RMW16_RING 8      // residual loads in flight per wave (of the 16 a 128 x 64 sub-tile needs)
#endif
#ifndef TDC_RMW32_FOLD_RING
#define TDC_RMW32_FOLD_RING 4 // fp32 read-modify-write in the fold form
#endif
#ifndef TDC_RMW16_FOLD_RING
#define TDC_RMW16_FOLD_RING 6 // ... in the fold form (fp8 operands)
#endif
// a += float(r[0..3]), b += float(r[4..7]) for eight 16-bit residual values.  fp16: one v_fma_mix_f32 per element (the
// half of the packed register converted on the fly, x 1.0, + the fp32 sum) instead of a conversion per element and a packed
// add per pair - 8 instead of 12 instructions per eight values, the same correctly rounded fp32 sum (the conversion is exact).
template <class TC>
__device__ __forceinline__ void res_add8(f32x4& a, f32x4& b, typename VecOf<TC>::v8 r) {
    if constexpr (std::is_same<TC, f16>::value) {
        const u32x4 w = __builtin_bit_cast(u32x4, r);
        auto mix = [](unsigned h, float lo, float hi, float& xl, float& xh) {
            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(xl) : "v"(h), "v"(lo));
            asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(xh) : "v"(h), "v"(hi));
        };
        float x[8];
        mix(w[0], a[0], a[1], x[0], x[1]); mix(w[1], a[2], a[3], x[2], x[3]);
        mix(w[2], b[0], b[1], x[4], x[5]); mix(w[3], b[2], b[3], x[6], x[7]);
        a = (f32x4){x[0], x[1], x[2], x[3]};
        b = (f32x4){x[4], x[5], x[6], x[7]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] += (float)r[e]; b[e] += (float)r[4 + e]; }
    }
}

// EMIT (LayerNorm fusion over a 16-bit residual stream, producer side: ln_part != NULL, x16 == NULL): per row and 64-column slot
// the (mean, M2) of the fp32 sums that are rounded into the stream - the consumer GEMM reads the stream itself as its A operand,
// so there is no 16-bit copy to write.  Identity row maps, N % 64 == 0 (host-checked).
// FOLD: the row / column operands of EpiOps (fp8 operands: the dequantisation scales of a residual-stream GEMM over a 16-bit stream)
template <class TC, int ROWS, bool LB, int RING, bool EMIT = false, bool FOLD = false>
__device__ __forceinline__ void epi_staged_rmw16(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase, int nbase,
                                                 int lane, const EpiLane& el) {
    typedef typename VecOf<TC>::v8 v8c;
    typedef typename VecOf<TC>::v4 v4c;
    const int fr = lane & 15, g = lane >> 4;
    EpiOps<8, 4, LB, FOLD> ops;
    ops.load(p, mbase, nbase, fr, g, el);
    constexpr int NPASS = 128 / ROWS, UP = ROWS / 8, NU = 16;     // unit = 8 rows x 64 columns = one 16-byte store per lane
    const int rrow = lane >> 3, ck = lane & 7;
    auto pack8 = [](f32x4 a, f32x4 b) {
        const v4c x = cvt4<TC>(a), y = cvt4<TC>(b);
        const u32x2 xr = __builtin_bit_cast(u32x2, x), yr = __builtin_bit_cast(u32x2, y);
        return __builtin_bit_cast(v8c, (u32x4){xr[0], xr[1], yr[0], yr[1]});
    };
    if (p.cm.seg == 0 && p.rm.seg == 0 && mbase + 128 <= p.M && nbase + 64 <= p.N) {
        // interior sub-tile, identity row maps: running pointers, straight-line passes (see epi_staged16 / epi_staged32)
        const auto fops = opaque_ops(ops);
        char* wbase = region + fr * 256;
        const char* rbase = region + rrow * 256;
        TC* cptr = (TC*)p.C + (long long)(mbase + rrow) * p.ldc + nbase + ck * 8;
        const TC* rptr = (const TC*)p.res + (long long)(mbase + rrow) * p.ldres + nbase + ck * 8;
        const long long cstep = 8ll * p.ldc, rstep = 8ll * p.ldres;
        float* lptr = EMIT ? p.ln_part + 2 * ((long long)(nbase >> 6) * p.M + mbase + rrow) : nullptr;
        v8c ring[RING];
#pragma unroll
        for (int u = 0; u < RING; ++u) { ring[u] = *(const v8c*)rptr; rptr += rstep; }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int ii = 0; ii < ROWS / 16; ++ii) {
                const int i = pass * (ROWS / 16) + ii;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(f32x4*)(wbase + ii * 4096 + (((j * 4 + g) ^ fr) << 4)) = fops.lin(acc[i][j], i, j);
            }
            f32x4 lo[UP], hi[UP];
#pragma unroll
            for (int q = 0; q < UP; ++q) {
                const int r = q * 8 + rrow;
                lo[q] = *(const f32x4*)(rbase + q * 2048 + (((2 * ck) ^ (r & 15)) << 4));
                hi[q] = *(const f32x4*)(rbase + q * 2048 + (((2 * ck + 1) ^ (r & 15)) << 4));
            }
#pragma unroll
            for (int q = 0; q < UP; ++q) {
                const int u = pass * UP + q;
                const v8c rr = ring[u % RING];
                if (u + RING < NU) { ring[u % RING] = *(const v8c*)rptr; rptr += rstep; }
                f32x4 a = lo[q], b = hi[q];
                res_add8<TC>(a, b, rr);
                __builtin_nontemporal_store(pack8(a, b), (v8c*)cptr);
                cptr += cstep;
                if (EMIT) {
                    float mean, m2;
                    slot_stats_row8(a, b, mean, m2);
                    if (ck == 0) *(float2*)lptr = make_float2(mean, m2);
                    lptr += 16;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // general form: row maps, partial tiles (clamped residual loads, guarded stores)
    auto load_res = [&](int u) {
        int m = mbase + u * 8 + rrow, n = nbase + ck * 8;
        if (m > p.M - 1) m = p.M - 1;
        if (n > p.N - 8) n = p.N - 8;
        return *(const v8c*)((const TC*)p.res + p.rm(m) * p.ldres + n);
    };
    v8c rr[RING];
#pragma unroll
    for (int u = 0; u < RING; ++u) rr[u] = load_res(u);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
        for (int ii = 0; ii < ROWS / 16; ++ii) {
            const int i = pass * (ROWS / 16) + ii;
            const int r = ii * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(region + r * 256 + (((j * 4 + g) ^ (r & 15)) << 4)) = ops.lin(acc[i][j], i, j);
        }
#pragma unroll
        for (int q = 0; q < UP; ++q) {
            const int u = pass * UP + q;
            const int r = q * 8 + rrow;
            f32x4 a = *(const f32x4*)(region + r * 256 + (((2 * ck) ^ (r & 15)) << 4));
            f32x4 b = *(const f32x4*)(region + r * 256 + (((2 * ck + 1) ^ (r & 15)) << 4));
            const v8c x = rr[u % RING];
            if (u + RING < NU) rr[u % RING] = load_res(u + RING);
            res_add8<TC>(a, b, x);
            const int m = mbase + pass * ROWS + r, n = nbase + ck * 8;
            if (m < p.M && n < p.N) __builtin_nontemporal_store(pack8(a, b), (v8c*)((TC*)p.C + p.cm(m) * p.ldc + n));
            if (EMIT) {
                float mean, m2;
                slot_stats_row8(a, b, mean, m2);          // every lane takes part (rows beyond M: clamped duplicates)
                if (m < p.M && ck == 0 && nbase < p.N)
                    *(float2*)(p.ln_part + 2 * ((long long)(nbase >> 6) * p.M + m)) = make_float2(mean, m2);
            }
        }
    }
}

// returns false when this (act, res, out) combination / alignment has no staged variant.  SMALL: 4 KiB staging region
// per wave (the persistent kernel stages beside the live pipeline buffers, lane-held operands), otherwise 16 KiB.
template <class T, bool SMALL, bool FOLD>
__device__ __forceinline__ bool epilogue_staged_f(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase,
                                                  int nbase, int lane, const EpiLane& el) {
    const int res = p.res ? (p.res_f32 ? 1 : 2) : 0;
    constexpr int R16 = SMALL ? 32 : 128, RSW = SMALL ? 64 : 128, R32 = SMALL ? 16 : 64;
    if constexpr (FOLD) {
        if (p.out_fp8) {           // e4m3 output with analytic per-row scales (host-checked alignment, whole wave tiles)
            constexpr int R8 = SMALL ? 64 : 128;
            if (p.act == TDC_ACT_SWIGLU) epi_staged8_swiglu<T, SMALL>(p, acc, region, mbase, nbase, lane, el);
            else if (p.act == TDC_ACT_GELU_TANH) epi_staged8<T, TDC_ACT_GELU_TANH, R8, SMALL>(p, acc, region, mbase, nbase, lane, el);
            else if (p.act == TDC_ACT_GELU_ERF) epi_staged8<T, TDC_ACT_GELU_ERF, R8, SMALL>(p, acc, region, mbase, nbase, lane, el);
            else epi_staged8<T, TDC_ACT_NONE, R8, SMALL>(p, acc, region, mbase, nbase, lane, el);
            return true;
        }
    }
    // fp8 operands of a residual-stream GEMM (fp32 stream here, 16-bit stream below): the fold + residual forms exist for the e4m3
    // instantiations only (kScaleOnly: gemm_fp8.hip) - the host refuses them for 16-bit operands, and compiled into the 16-bit
    // LayerNorm-fold instance they only cost it registers
    if constexpr (FOLD && kScaleOnly) {
        if (p.out_f32 && res == 1 && p.act == TDC_ACT_NONE) {
            epi_staged32<T, 1, R32, SMALL, false, true, TDC_RMW32_FOLD_RING>(p, acc, region, mbase, nbase, lane, el);
            return true;
        }
    }
    if (p.out_f32 || res == 1 || (p.ldc & 7) || ((uintptr_t)p.C & 15)) return false;
    if (p.act == TDC_ACT_SWIGLU) {
        if (p.N & 15) return false;
        epi_staged_swiglu<T, RSW, SMALL, FOLD>(p, acc, region, mbase, nbase, lane, el);
        return true;
    }
    if ((p.N & 7) && !(p.c_pad8 && !(p.N & 3))) return false;   // 16-byte stores of 8 columns: the last group may overhang N by 4
    if (p.act == TDC_ACT_GELU_ERF) epi_staged16<T, TDC_ACT_GELU_ERF, 0, R16, SMALL, FOLD>(p, acc, region, mbase, nbase, lane, el);
    else if (p.act == TDC_ACT_GELU_TANH) epi_staged16<T, TDC_ACT_GELU_TANH, 0, R16, SMALL, FOLD>(p, acc, region, mbase, nbase, lane, el);
    else if (res == 2) {
        if ((p.N & 7) || (p.ldres & 7) || ((uintptr_t)p.res & 15)) return false;
        constexpr int RR = SMALL ? 16 : 32;
        if constexpr (FOLD && !kScaleOnly) return false;
        if constexpr (FOLD) {    // fp8 operands of a residual-stream GEMM over a 16-bit stream (scales through the fold operands)
            // (ring of 4 residual loads instead of 8: the fold's row / column operands take 32 registers of their own; rings of 4-8
            //  measured equal on the plain form, profiles/r04_gemm_rmw16_ring_ab.log)
            if (p.ctype == TDC_F16) epi_staged_rmw16<f16, RR, SMALL, TDC_RMW16_FOLD_RING, false, true>(p, acc, region, mbase, nbase, lane, el);
            else epi_staged_rmw16<bf16, RR, SMALL, TDC_RMW16_FOLD_RING, false, true>(p, acc, region, mbase, nbase, lane, el);
            return true;
        }
        if (p.ln_part) {         // producer of the LayerNorm fold over a 16-bit stream
            if (p.ctype == TDC_F16) epi_staged_rmw16<f16, RR, SMALL, TDC_RMW16_RING, true>(p, acc, region, mbase, nbase, lane, el);
            else epi_staged_rmw16<bf16, RR, SMALL, TDC_RMW16_RING, true>(p, acc, region, mbase, nbase, lane, el);
        }
        else if (p.ctype == TDC_F16) epi_staged_rmw16<f16, RR, SMALL, TDC_RMW16_RING>(p, acc, region, mbase, nbase, lane, el);
        else epi_staged_rmw16<bf16, RR, SMALL, TDC_RMW16_RING>(p, acc, region, mbase, nbase, lane, el);
    }
    else epi_staged16<T, 0, 0, R16, SMALL, FOLD>(p, acc, region, mbase, nbase, lane, el);
    return true;
}

template <class T, bool SMALL>
__device__ __forceinline__ bool epilogue_staged(const GemmArgs& p, f32x4 (&acc)[8][4], char* region, int mbase,
                                                int nbase, int lane, const EpiLane& el = EpiLane()) {
    constexpr int R32 = SMALL ? 16 : 64;
    if (!SMALL && p.ln_stats && p.out_f32) return epilogue_staged_f<T, SMALL, true>(p, acc, region, mbase, nbase, lane, el);
    if (p.out_f32) {
        const int res = p.res ? (p.res_f32 ? 1 : 2) : 0;
        if (p.act != TDC_ACT_NONE || res == 2) return false;
        if (p.x16) epi_staged32<T, 1, R32, SMALL, true>(p, acc, region, mbase, nbase, lane, el);
        // residual ring: 12 loads ahead in the persistent kernel (same-box A/B against 8: out-projections -2.5 %, fc2 -1 %;
        // 16 spills), 8 in the one-tile-per-workgroup kernel (12 spills there)
        else if (res == 1) epi_staged32<T, 1, R32, SMALL, false, false, SMALL ? 12 : 8>(p, acc, region, mbase, nbase, lane, el);
        else epi_staged32<T, 0, R32, SMALL, false>(p, acc, region, mbase, nbase, lane, el);
        return true;
    }
    if (!SMALL && p.ln_stats) return epilogue_staged_f<T, SMALL, true>(p, acc, region, mbase, nbase, lane, el);
    return epilogue_staged_f<T, SMALL, false>(p, acc, region, mbase, nbase, lane, el);
}


// ======================================================================================================================
// 256x256x64 tile, 8 waves (2 M x 4 N, 128x64 per wave), 128 KiB LDS, "8-phase" schedule (cdna_hip_programming.md
// T3+T4): every K tile is 4 phases of 16 MFMAs (one 64x32 quadrant of the wave's tile x K=64); each phase stages ONE
// 16-KiB half-tile with global_load_lds, and the loads stay in flight ACROSS the raw s_barriers: a counted
// s_waitcnt vmcnt(6) once per K tile (3 half-tiles in flight), never vmcnt(0) in the steady state.
//
// LDS: buf[2] x {A-half0, A-half1, W-half0, W-half1} x 16 KiB.  Half h of A holds rows {wm*128 + h*64 + 0..63} of both
// wave rows wm, half h of W the columns {wn*64 + h*32 + 0..31} of all four wave columns, so quadrant (a_h, b_h') of
// every wave needs exactly one A half and one W half.  Quadrant order (a0,b0) (a0,b1) (a1,b1) (a1,b0): one new operand
// sub-block per phase, b0 stays in registers.  Staging order A0 B0 B1 A1; tile t+2's A0/B0/B1 are issued during tile
// t's phases 2/3/4 into the regions tile t has finished reading one phase earlier (all waves have passed a barrier
// after their lgkmcnt-retired reads), tile t+1's A1 during phase 1.
constexpr int T2_HALF = 128 * 64 * 2;         // 16 KiB
constexpr int T2_BUF = 4 * T2_HALF;           // A0 A1 W0 W1
constexpr int T2_LDS = 2 * T2_BUF;            // 128 KiB

// (Round 5 removed the one-tile-per-workgroup form of this kernel, `gemm256_kernel`: with the bias as the accumulators' initial
// value its register allocation parked accumulator tiles in scratch INSIDE the K loop - every reload a vmcnt(0) that drains the
// staging pipeline, the row-mapped Q-Former GEMMs ran 26-36 % slower - and four re-arrangements of its prologue did not move the
// spills out again.  The persistent form below (0 spills) now takes every launch of 256 x 256 tiles: row-mapped A operands through
// per-tile offsets, launches with fewer tiles than CUs as workgroups with one tile.)
#define T2_BARRIER() __builtin_amdgcn_s_barrier()
// end of a load segment: retire this wave's LDS reads BEFORE the barrier (so that a later stage by any wave, incl.
// the other, staggered, wave group, can never overwrite bytes still being read), then pin the MFMA cluster below it
#define T2_END_LOADS()                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
    __builtin_amdgcn_s_barrier();                               \
    __builtin_amdgcn_sched_barrier(0)
#define T2_LOAD_A(buf, h)                                                          \
    if (active) {                                                                   \
        const char* base = smem + (buf) * T2_BUF + (h) * T2_HALF;                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                             \
            fa[i][0] = *(const v8*)(base + a_off[i]);                               \
            fa[i][1] = *(const v8*)(base + (a_off[i] ^ 64));                        \
        }                                                                           \
    }
#define T2_LOAD_B(dst, buf, h)                                                      \
    if (active) {                                                                   \
        const char* base = smem + (buf) * T2_BUF + (2 + (h)) * T2_HALF;             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                             \
            dst[j][0] = *(const v8*)(base + w_off[j]);                              \
            dst[j][1] = *(const v8*)(base + (w_off[j] ^ 64));                       \
        }                                                                           \
    }
#ifdef TDC_GEMM_DIAG
#define TDC_DIAG_MMA_ON && !(p.diag_mode & 2)
#else
#define TDC_DIAG_MMA_ON
#endif
// raised wave priority around the MFMA clusters: same-box A/B in round 5 (tools/lib_ab.sh on builds with -DTDC_GEMM_NOPRIO /
// -DTDC_GEMM_PRIO_HI=3): without it the towers take 969 instead of 884 ms
#if defined(TDC_GEMM_NOPRIO)
#define TDC_SETPRIO(x)
#elif defined(TDC_GEMM_PRIO_HI)
#define TDC_SETPRIO(x) __builtin_amdgcn_s_setprio((x) ? TDC_GEMM_PRIO_HI : 0)
#else
#define TDC_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#define T2_MMA(MI0, NJ0, fbx)                                                       \
    if (active TDC_DIAG_MMA_ON) {                                                   \
        TDC_SETPRIO(1);                                                             \
        if constexpr (FP8) {                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                           \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                           \
                acc[(MI0) + i][(NJ0) + j] = mma128_fp8(fbx[j][0], fbx[j][1], fa[i][0], fa[i][1], acc[(MI0) + i][(NJ0) + j]); \
        } else {                                                                    \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                           \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                           \
                acc[(MI0) + i][(NJ0) + j] = mfma16(fbx[j][ks], fa[i][ks], acc[(MI0) + i][(NJ0) + j]); \
        }                                                                           \
        TDC_SETPRIO(0);                                                             \
    }

// ======================================================================================================================
// Persistent form of the 256^2 kernel: one workgroup per CU walks its tiles (static round-robin inside the XCD's contiguous
// chunk of the grouped tile order, so the 32 workgroups of an XCD still cover the same patch of tiles at any time).
// In-kernel stamps of the one-tile-per-workgroup kernel (tools/gemm_stamps.cpp) showed ~3 us per tile - 7 % of a K=1152
// tile - between the last store of one workgroup and the first MFMA of the next (dispatch + first loads from HBM), and
// that the C-tile drain itself is store-issue bound (~73 cycles per 1-KiB store instruction and CU), not latency bound.
// Here the staging stream simply runs on across the tile seam: the last two K iterations of a tile stage K tiles 0 and 1
// of the NEXT tile (the staging cursor switches its base pointers between phase 1 and phase 2 of iteration nk-2), the
// epilogue stages through 4 KiB per wave BESIDE the 128 KiB pipeline buffers, and the next main loop starts with its
// operands already in LDS.  Staging addresses are an SGPR tile base + 32-bit per-lane offsets (a row-mapped A: set_stage_tile).
constexpr int T2P_LDS = T2_LDS + 8 * 4096;    // 160 KiB

template <class T, bool LNF, bool FP8>  // LNF: LayerNorm-fold consumer (ln_stats != NULL): 5 more lane-held epilogue operands
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmArgs p) {
    typedef typename VecOf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.tiles_m * p.tiles_n;
    // ---- this workgroup's tiles: ids base + l + G8 * j of XCD x's chunk [base, base + len) (same chunks as xcd_remap)
    const int G8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int cq = nwg >> 3, cr = nwg & 7;
    const int chunk_base = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_len = cq + (xcd < cr ? 1 : 0);
    const int n_my = l < chunk_len ? (chunk_len - l + G8 - 1) / G8 : 0;
    if (n_my == 0) return;
    // (Start offsets between the tile-column groups of an XCD, between XCDs, and two half-XCD sets half a tile period apart were
    // measured in round 2 - profiles/r02_gemm_stagger*.log, DESIGN.md section 4 - and removed: the drain gets shorter and the
    // main loop loses as much, because CUs out of lockstep stop sharing operand panels in flight in L2.)
    // ---- staging cursor: SGPR bases + per-lane byte offsets of the tile being staged
    unsigned a_so[2][2], w_so[2][2];
    const char* a_base;
    const char* w_base;
    auto set_stage_tile = [&](int m0, int n0) {
        // the lane-derived terms are recomputed from an opaque copy of the lane id on every call (once per tile): kept in
        // registers across the K loop they would be spilled, and a scratch reload at the tile seam waits for every
        // staged load in flight (in-order vmcnt)
        const int sl = fresh_lane();
        const int srow = sl >> 3;
        const int schunk = (sl & 7) ^ srow;
        // A rows through the row map (tdc_gemm_desc.a_map; host: monotone, a tile's rows within 2 GiB of its first one): the
        // tile base is the mapped first row, the per-lane offsets the mapped rows relative to it
        const bool mapped = p.am.seg != 0;                                     // uniform
        const long long arow0 = mapped ? p.am(m0) : (long long)m0;
        a_base = (const char*)p.A + arow0 * p.lda * 2;
        w_base = (const char*)p.W + (long long)n0 * p.ldw * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (wave * 2 + j) * 8 + srow;                       // row within the half (0..127)
                int am = (r >> 6) * 128 + h * 64 + (r & 63);
                int wn = (r >> 5) * 64 + h * 32 + (r & 31);
                if (am > p.M - 1 - m0) am = p.M - 1 - m0;
                if (wn > p.N - 1 - n0) wn = p.N - 1 - n0;
                if (mapped) am = (int)(p.am(m0 + am) - arow0);
                a_so[h][j] = (unsigned)(am * p.lda + schunk * 8) * 2u;
                w_so[h][j] = (unsigned)(wn * p.ldw + schunk * 8) * 2u;
            }
    };
    const int lds_stage = wave * 2 * 1024;
    // one half-tile (2 x 1 KiB per wave) from the SGPR source `src` (tile base + K offset) + the per-lane offsets
    auto stage_a = [&](int buf, int h, const char* src) {
#ifdef TDC_GEMM_DIAG
        if (p.diag_mode & 1) return;
#endif
        char* dst = smem + buf * T2_BUF + h * T2_HALF + lds_stage;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + a_so[h][0]), LDS_PTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + a_so[h][1]), LDS_PTR(dst + 1024), 16, 0, 0);
    };
    auto stage_w = [&](int buf, int h, const char* src) {
#ifdef TDC_GEMM_DIAG
        if (p.diag_mode & 1) return;
#endif
        char* dst = smem + buf * T2_BUF + (2 + h) * T2_HALF + lds_stage;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + w_so[h][0]), LDS_PTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + w_so[h][1]), LDS_PTR(dst + 1024), 16, 0, 0);
    };

    // wave -> (row, column) of the 2 x 4 wave grid: the two waves of a SIMD (w and w+4) take wave columns c and c+2, so
    // in a tile whose upper half of the columns lies beyond N (N = 1152: the 5th column tile) every SIMD keeps exactly one
    // working wave and the idle partner's MFMA slots are not wasted
    const int wm = (wave >> 1) & 1, wn_ = (wave & 1) | ((wave >> 2) << 1);
    const int fr = lane & 15, g = lane >> 4;
    int a_off[4], w_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        a_off[i] = r * 128 + ((g ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wn_ * 32 + j * 16 + fr;
        w_off[j] = r * 128 + ((g ^ (r & 7)) << 4);
    }

    f32x4 acc[8][4];
    v8 fa[4][2], fb0[2][2], fb1[2][2];
    const int nk = p.K / 64;        // >= 2 (host)
    int par = 0;                    // LDS buffer of the current tile's K tile 0

    // running staging sources (SGPR pairs): pa1 = A source of the A1 half staged in phase 1 (K tile kt+1), pa2 / pw2 =
    // A / W source of K tile kt+2.  They advance by one K tile (128 B) per iteration behind the last MFMA cluster and are
    // re-based at the tile seam, so no address arithmetic sits in a load segment (the LDS-read -> barrier -> MFMA path).
    const char *pa1, *pa2, *pw2;
    // main loop of ONE tile; MORE = another tile follows (its first two K tiles are staged by the last two iterations),
    // (m1, n1) = that tile's origin
    auto tile_loop = [&](auto active_c, auto more_c, int m1, int n1) {
        constexpr bool active = decltype(active_c)::value;
        constexpr bool MORE = decltype(more_c)::value;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = (kt + par) & 1, nxt = cur ^ 1;
            const bool s1 = kt + 1 < nk, s2 = kt + 2 < nk;
            // ---- phase 1: quadrant (a0, b0)
            T2_LOAD_B(fb0, cur, 0);
            __builtin_amdgcn_sched_barrier(0);
            T2_LOAD_A(cur, 0);
            __builtin_amdgcn_sched_barrier(0);      // LDS reads first: an LDS-DMA issue ahead of them delays the barrier
            if (MORE || s1) stage_a(nxt, 1, pa1);
            T2_END_LOADS();
            T2_MMA(0, 0, fb0);
            T2_BARRIER();
            // the staging cursor crosses the tile seam here: everything staged from now on belongs to the next tile
            if (MORE && kt == nk - 2) {
                set_stage_tile(m1, n1);
                pa2 = a_base; pw2 = w_base;
            }
            // ---- phase 2: quadrant (a0, b1)
            T2_LOAD_B(fb1, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (MORE || s2) stage_a(cur, 0, pa2);
            T2_END_LOADS();
            T2_MMA(0, 2, fb1);
            T2_BARRIER();
            // ---- phase 3: quadrant (a1, b1)
            T2_LOAD_A(cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (MORE || s2) stage_w(cur, 0, pw2);
            T2_END_LOADS();
            T2_MMA(4, 2, fb1);
            T2_BARRIER();
            // ---- phase 4: quadrant (a1, b0); retire K tile kt+1 (3 half-tiles of kt+2 may stay in flight)
            if (MORE || s2) {
                stage_w(cur, 1, pw2);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            T2_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
            T2_MMA(4, 0, fb0);
            pa1 = pa2; pa2 += 128; pw2 += 128;
            __builtin_amdgcn_sched_barrier(0);
            T2_BARRIER();
        }
    };

    int id = chunk_base + l;
    int tm, tn;
    tile_coords(id, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
    int m0 = tm * 256, n0 = tn * 256;
    // lane L keeps the epilogue operands of the tile being computed (EpiLane: column n0 + wn*64 + L, rows m0 + wm*128 +
    // L and + 64 + L); the first set is waited for here, compiler-visibly, before any staging load is in flight
    auto load_epi_lane = [&](int m0_, int n0_) {
        EpiLane e = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ln = fresh_lane();       // not `lane`: kept live across the K loop it is spilled, and its scratch reload here
                                           // waits (vmcnt(0), in order) for every staged load of the next tile
        int n = n0_ + wn_ * 64 + ln;
        if (n > p.N - 1) n = p.N - 1;
        if (p.bias) e.bias = p.bias[n];
        if (LNF) {
            if (!kScaleOnly) e.c1 = p.ln_c1[n];
            int r0 = m0_ + wm * 128 + ln, r1 = r0 + 64;
            if (r0 > p.M - 1) r0 = p.M - 1;
            if (r1 > p.M - 1) r1 = p.M - 1;
            const float2 s0 = *(const float2*)(p.ln_stats + 2 * (long long)r0);
            const float2 s1 = *(const float2*)(p.ln_stats + 2 * (long long)r1);
            e.mean0 = s0.x; e.rstd0 = s0.y; e.mean1 = s1.x; e.rstd1 = s1.y;
        }
        return e;
    };
    // Between its load (one tile ahead, at the start of the previous epilogue) and its use the set is parked in the wave's
    // epilogue staging region, which is idle during the main loop: no VGPR stays live across the K loop for it.
    auto el_park = [&]() { return (EpiLane*)(smem + T2_LDS + wave * 4096) + fresh_lane(); };
    *el_park() = load_epi_lane(m0, n0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // ---- prologue of the first tile: K tile 0 complete + K tile 1's A0 B0 B1
    set_stage_tile(m0, n0);
    stage_a(0, 0, a_base); stage_w(0, 0, w_base); stage_w(0, 1, w_base); stage_a(0, 1, a_base);
    stage_a(1, 0, a_base + 128); stage_w(1, 0, w_base + 128); stage_w(1, 1, w_base + 128);
    pa1 = a_base + 128; pa2 = a_base + 256; pw2 = w_base + 256;
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    T2_BARRIER();
    for (int it = 0; it < n_my; ++it) {
        // stagger: waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA cluster while its partner is
        // in its LDS-read / staging segment (MI355X_MICROARCH.md "Two waves per SIMD").  Re-established for every tile: the two wave groups leave a tile's loop one barrier apart,
        // and a group's epilogue holds no barrier - left staggered, waves 4-7 would sit at their last barrier of the tile for the
        // whole length of waves 0-3's epilogue and waves 0-3 then at their first barrier of the next tile for the whole length of
        // waves 4-7's: the two halves of the C tile drained one after the other (per-wave stamps, tools/run_gemm_epi_parts.sh:
        // first epilogue start -> last store acknowledged = twice a wave's epilogue).  Closing the stagger behind the loop
        // (below) and opening it again here costs one barrier wait per tile and lets all eight epilogues run together.
        if (wave >= 4) T2_BARRIER();
        const bool more = it + 1 < n_my;
        int m1 = 0, n1 = 0;
        if (more) {
            int tm1, tn1;
            tile_coords(id + G8, p.tiles_m, p.tiles_n, tm1, tn1, p.group_m);
            m1 = tm1 * 256; n1 = tn1 * 256;
        }
        if (LNF) {        // the fold needs the bare product (EpiOps::lin)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {          // the accumulators start from the bias (acc_init_bias): gathered from the lane-held set parked in LDS
            const float bl = el_park()->bias;
            const int bg = fresh_lane() >> 4;
            f32x4 bcol[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bcol[j][e] = lane_get(bl, j * 16 + bg * 4 + e);
            acc_init_bias<8, 4>(acc, bcol);
        }
        const bool wave_active = n0 + wn_ * 64 < p.N;
#ifdef TDC_GEMM_DIAG
        if (p.stamps && threadIdx.x == 0) {
            p.stamps[(size_t)id * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            p.stamps[(size_t)id * 8 + 6] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            __builtin_amdgcn_sched_barrier(0);
            p.stamps[(size_t)id * 8 + 0] = p.stamps[(size_t)id * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (more) {
            if (wave_active) tile_loop(std::true_type(), std::true_type(), m1, n1);
            else tile_loop(std::false_type(), std::true_type(), m1, n1);
        } else {
            if (wave_active) tile_loop(std::true_type(), std::false_type(), m1, n1);
            else tile_loop(std::false_type(), std::false_type(), m1, n1);
        }
        par ^= nk & 1;
        if (wave < 4) T2_BARRIER();   // matches the stagger barrier of waves 4-7: both groups enter the epilogue together
#ifdef TDC_GEMM_DIAG
        if (p.stamps && threadIdx.x == 0) {
            __builtin_amdgcn_sched_barrier(0);
            p.stamps[(size_t)id * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        // ---- epilogue (no barrier: every wave stages through its own 4 KiB beside the pipeline buffers).  The lane id is
        // made opaque per tile so that the epilogue's lane-derived addresses are recomputed here instead of being hoisted
        // out of the tile loop, kept live across the main loop and spilled (their scratch reloads would sit behind the
        // next tile's staged loads in the in-order vmcnt queue).
        const int elane = fresh_lane();
#ifdef TDC_GEMM_DIAG
        if (p.wstamps && lane == 0) {
            __builtin_amdgcn_sched_barrier(0);
            p.wstamps[((size_t)id * 8 + wave) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        const EpiLane el = *el_park();
        EpiLane el_next = el;
        if (more) el_next = load_epi_lane(m1, n1);                       // complete by the vmcnt(0) below
        if (wave_active && p.debug != 1) {
            char* region = smem + T2_LDS + wave * 4096;
            const int mb = m0 + wm * 128, nb = n0 + wn_ * 64;
            if (LNF) {
                epilogue_staged_f<T, true, true>(p, acc, region, mb, nb, elane, el);   // host: a staged variant exists
            } else if (!epilogue_staged<T, true>(p, acc, region, mb, nb, elane, el)) {
                epilogue<T, 8, 4, true>(p, acc, mb, nb, elane & 15, elane >> 4, el);
            }
        } else if (p.debug == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
#ifdef TDC_GEMM_DIAG
        if (p.stamps && threadIdx.x == 0) {
            __builtin_amdgcn_sched_barrier(0);
            p.stamps[(size_t)id * 8 + 3] = p.stamps[(size_t)id * 8 + 4] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#ifdef TDC_GEMM_DIAG
        if (p.wstamps && lane == 0) {
            __builtin_amdgcn_sched_barrier(0);
            p.wstamps[((size_t)id * 8 + wave) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        // A compiler-visible vmcnt(0): without it the waitcnt pass protects the fragment registers of the next main loop
        // against this epilogue's (long finished) loads with a vmcnt(0) INSIDE the K loop, which would drain the staging
        // pipeline every iteration.  Here it only waits for the acknowledgement of the last stores.  Unconditional: the
        // pass cannot tell that !more leaves the loop.
        __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef TDC_GEMM_DIAG
        if (p.wstamps && lane == 0) {
            __builtin_amdgcn_sched_barrier(0);
            p.wstamps[((size_t)id * 8 + wave) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        *el_park() = el_next;
        id += G8; m0 = m1; n0 = n1;
    }
}
#undef T2_END_LOADS
#undef T2_BARRIER
#undef T2_LOAD_A
#undef T2_LOAD_B
#undef T2_MMA

// workgroups of the persistent kernel = CUs of the device rounded down to a multiple of 8 (one per CU: 160 KiB of LDS);
// 0 disables it (a device whose LDS cannot hold 160 KiB per workgroup; diagnostics builds: TDC_GEMM_PERSIST=0)
inline int persistent_grid() {
    static int grid[kMaxDev];
    static bool known[kMaxDev];
    const int dev = current_device();
    if (!known[dev]) {
        const char* e = nullptr;
#ifdef TDC_GEMM_DIAG
        e = getenv("TDC_GEMM_PERSIST");
#endif
        hipDeviceProp_t prop;
        if ((e && atoi(e) == 0) || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            prop.sharedMemPerBlock < (size_t)T2P_LDS)
            grid[dev] = 0;
        else
            grid[dev] = (prop.multiProcessorCount / 8) * 8;
        known[dev] = true;
    }
    // a process confined to part of the chip (ROC_GLOBAL_CU_MASK / HSA_CU_MASK: the device still reports every CU) says how many
    // CUs it really owns through tdc_gemm_set_persistent_grid - an explicit call that announces itself on stderr, never an
    // environment variable (a stale one would silently halve the GEMM rate of the whole process)
    const int o = tdc_gemm_persist_grid_override;
    if (o >= 8 && o <= grid[dev]) return (o / 8) * 8;
    return grid[dev];
}

// Group height of the tile order, per launch.  The 32 workgroups an XCD runs at a time take 32 CONSECUTIVE tile ids of the XCD's
// chunk (persistent kernel: ids base + l + 32 j, round j), so the operand panels that stream through that L2 in a round are the
// distinct tile rows + tile columns of a window of 32 ids.  With GROUP_M x tiles_n not a multiple of 32 the windows straddle
// groups (N = 1152 at GROUP_M = 8: 17.4 panels per window on average instead of 12); the host walks the first windows of a chunk
// with the kernel's own id -> tile arithmetic and takes the height with the fewest panels: 1 (row-major) for <= 7 column tiles
// (12.3 panels at 5 or 6 columns), 4 or 8 for the wide GEMMs.  PMC, fabric bytes / algorithmic bytes: see profiles/archive/NOTES_rounds1-4.md 9.4.
inline int choose_group_m(int tiles_m, int tiles_n) {
    // memo: a serving process sees new tiles_m values all the time (tail batches differ per video) - unbounded map, looked up
    // under the lock, computed OUTSIDE it (tdc_gemm is called from several host threads, dist.py; two threads computing the
    // same entry at once write the same value)
    static std::unordered_map<unsigned long long, int> memo;
    static std::mutex memo_mu;
    const unsigned long long key = ((unsigned long long)(unsigned)tiles_m << 32) | (unsigned)tiles_n;
    {
        std::lock_guard<std::mutex> lock(memo_mu);
        const auto it = memo.find(key);
        if (it != memo.end()) return it->second;
    }
    const int cand[4] = {4, 8, 1, 2};                                   // ties go to the earlier entry
    const long long n = (long long)tiles_m * tiles_n;
    const long long len = n / 8 < 32 * 256 ? n / 8 : 32 * 256;           // up to 256 rounds of XCD 0's chunk
    int best = tiles_n >= 8 ? 4 : 8;
    double best_cost = 1e30;
    for (int c = 0; c < 4 && len >= 32; ++c) {
        const int G = cand[c];
        long long panels = 0;
        for (long long j = 0; j + 32 <= len; j += 32) {
            int rows[32], cols[32], nr = 0, nc = 0;
            for (int i = 0; i < 32; ++i) {
                const long long id = j + i;
                const long long per = (long long)G * tiles_n, grp = id / per, within = id - grp * per;
                const int first = (int)grp * G, h = (tiles_m - first < G) ? tiles_m - first : G;
                const int tn = (int)(within / h), tm = first + (int)(within - (long long)tn * h);
                bool fr = false, fc = false;
                for (int q = 0; q < nr; ++q) fr |= rows[q] == tm;
                for (int q = 0; q < nc; ++q) fc |= cols[q] == tn;
                if (!fr) rows[nr++] = tm;
                if (!fc) cols[nc++] = tn;
            }
            panels += nr + nc;
        }
        if ((double)panels < best_cost * 0.95) { best_cost = (double)panels; best = G; }   // 5 % hysteresis towards the earlier entry
    }
    std::lock_guard<std::mutex> lock(memo_mu);
    memo[key] = best;
    return best;
}

// kernel choice: the 256^2 8-phase kernel needs enough tiles to fill the 256 CUs (diagnostics builds: TDC_GEMM_FORCE=128|256)
inline bool use_256(int M, int N, int K) {
#ifdef TDC_GEMM_DIAG
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("TDC_GEMM_FORCE");
        force = e ? atoi(e) : 0;
    }
    if (force == 128) return false;
    if (force == 256) return true;
#endif
    const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
    return t256 >= 192 && K >= 128;
}

#ifdef TDC_GEMM_DIAG
}
unsigned long long* tdc_gemm_diag_stamps = nullptr;   // set by tools/gemm_stamps.cpp
unsigned long long* tdc_gemm_diag_wstamps = nullptr;
namespace {
#endif
template <class T, bool FP8>
int launch(const tdc_gemm_desc* d, hipStream_t st, bool force128 = false) {
    GemmArgs a;
    a.A = d->A; a.W = d->W; a.C = d->C; a.bias = d->bias; a.res = d->res;
    a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
    a.M = d->M; a.N = d->N; a.K = d->K;
    if (FP8) { a.lda /= 2; a.ldw /= 2; a.K /= 2; }     // kernels address A / W in 2-byte units
    a.out_f32 = d->out_f32; a.res_f32 = d->res_f32; a.act = d->act;
    a.x16 = d->x16; a.ldx16 = d->ldx16; a.ln_part = d->ln_part; a.ln_stats = d->ln_stats; a.ln_c1 = d->ln_c1;
    a.out_fp8 = d->out_fp8; a.out_stats = d->out_stats; a.out_w2max = d->out_w2max; a.out_bmax = d->out_bmax;
    a.out_wscale = d->out_wscale;
    a.debug = tdc_gemm_debug_mode;
    a.group_m = GROUP_M_DEFAULT;
    a.c_pad8 = d->c_pad8;
    a.ctype = d->c16_dtype_p1 ? d->c16_dtype_p1 - 1 : d->dtype;
#ifdef TDC_GEMM_DIAG
    { const char* e = getenv("TDC_GEMM_DIAGMODE"); a.diag_mode = e ? atoi(e) : 0; }
#endif
#ifdef TDC_GEMM_DIAG
    a.stamps = tdc_gemm_diag_stamps;
    a.wstamps = tdc_gemm_diag_wstamps;
#endif
    a.am = RowMap::make(d->a_map.seg, d->a_map.stride, d->a_map.off, d->a_map.inner);
    a.cm = RowMap::make(d->c_map.seg, d->c_map.stride, d->c_map.off, d->c_map.inner);
    a.rm = RowMap::make(d->r_map.seg, d->r_map.stride, d->r_map.off, d->r_map.inner);
    // 256 x 256 tiles: the persistent kernel - one workgroup per CU walks its tiles; a launch with fewer tiles than CUs simply
    // leaves workgroups with one tile or none.  It addresses its operands as a tile base + 32-bit per-lane offsets: K >= 128,
    // a tile's 256 rows of A (through a_map: monotone maps only) and of W within 2 GiB of the tile's first row; the
    // LayerNorm-fold instance only carries the LDS-staged epilogues.  Anything else runs on the 128 x 128 kernel (64-bit pointers).
    if (!force128 && use_256(a.M, a.N, a.K)) {
        const int G = persistent_grid();
        const bool fold_ok = !d->ln_stats || d->out_fp8 || (d->out_f32 && d->res && d->res_f32 && d->act == TDC_ACT_NONE) ||
                             (!d->out_f32 && !(d->ldc & 7) && !((uintptr_t)d->C & 15) &&
                              d->N % (d->act == TDC_ACT_SWIGLU ? 16 : 8) == 0 &&
                              (!d->res || (!d->res_f32 && !(d->ldres & 7) && !((uintptr_t)d->res & 15))));
        // rows of A a tile can span: identity 256; mapped m -> (m / seg) * stride + off + (m % seg) * inner, monotone when
        // inner >= 1 and stride >= (seg - 1) * inner + 1: at most 256 / seg + 2 segments are touched
        long long a_span = 256;
        bool a_ok = true;
        if (d->a_map.seg > 0) {
            const long long seg = d->a_map.seg, stride = d->a_map.stride, inner = d->a_map.inner, off = d->a_map.off;
            a_ok = inner >= 1 && stride >= (seg - 1) * inner + 1 && off >= 0;
            // exact: the map is monotone, so a tile spans am(last row of the tile) - am(first row) + 1 rows; the maximum over the
            // row tiles (a few thousand integer operations at most, only for the row-mapped launches of the Q-Former)
            auto am = [&](long long m) { return (m / seg) * stride + off + (m % seg) * inner; };
            a_span = 1;
            if (a_ok)
                for (long long m0 = 0; m0 < a.M; m0 += 256) {
                    const long long m1 = m0 + 255 < a.M - 1 ? m0 + 255 : a.M - 1;
                    const long long sp = am(m1) - am(m0) + 1;
                    if (sp > a_span) a_span = sp;
                }
        }
        const bool span_ok = a_span * a.lda * 2 < (1ll << 31) && 256ll * a.ldw * 2 < (1ll << 31);
        if (!(G > 0 && a_ok && a.K >= 128 && fold_ok && span_ok)) {
            // a launch big enough for 256 x 256 tiles runs on the 128 x 128 kernel: correct, but well below the persistent
            // kernel's rate - said once per shape, so that it does not go unnoticed
            static std::mutex note_mu;
            static std::unordered_map<unsigned long long, bool> noted;
            const unsigned long long key = ((unsigned long long)(unsigned)a.M << 40) ^ ((unsigned long long)(unsigned)a.N << 20) ^
                                           (unsigned long long)(unsigned)a.K;
            std::lock_guard<std::mutex> lock(note_mu);
            if (noted.size() < 64 && !noted.count(key)) {
                noted[key] = true;
                fprintf(stderr, "[tdc_hip] note: tdc_gemm M=%d N=%d K=%d is large enough for 256 x 256 tiles but runs on the 128 x 128 "
                                "kernel (%s)\n", a.M, a.N, a.K,
                        G <= 0 ? "no persistent grid on this device" : !a_ok ? "a_map is not monotone / has a negative offset"
                        : a.K < 128 ? "K < 128" : !fold_ok ? "no LDS-staged epilogue for this LayerNorm-fold / fp8 form (alignment)"
                        : "a tile's rows of A or W span 2 GiB or more");
            }
        }
        if (G > 0 && a_ok && a.K >= 128 && fold_ok && span_ok) {
            a.tiles_m = (a.M + 255) / 256;
            a.tiles_n = (a.N + 255) / 256;
            // group height of the tile order (choose_group_m above): fewest operand panels per window of 32 concurrent tiles.
            // The order never changes a result.  (Diagnostics builds: TDC_GEMM_GROUP_M overrides.)
            a.group_m = choose_group_m(a.tiles_m, a.tiles_n);
#ifdef TDC_GEMM_DIAG
            { const char* e = getenv("TDC_GEMM_GROUP_M"); if (e && atoi(e) > 0) a.group_m = atoi(e); }
#endif
            // hipFuncSetAttribute is per device: one flag per device and instantiation
            static bool attr256p_dev[kMaxDev];
            bool& attr256p = attr256p_dev[current_device()];
            if (!attr256p) {
                HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm256p_kernel<T, false, FP8>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, T2P_LDS));
                HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm256p_kernel<T, true, FP8>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, T2P_LDS));
                attr256p = true;
            }
            if (d->ln_stats) hipLaunchKernelGGL((gemm256p_kernel<T, true, FP8>), dim3(G), dim3(512), T2P_LDS, st, a);
            else hipLaunchKernelGGL((gemm256p_kernel<T, false, FP8>), dim3(G), dim3(512), T2P_LDS, st, a);
            return (int)hipGetLastError();
        }
    }
    a.group_m = GROUP_M_DEFAULT;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    static bool attr_set_dev[kMaxDev];
    bool& attr_set = attr_set_dev[current_device()];
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_kernel<T, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          4 * TILE_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, FP8>), dim3(a.tiles_m * a.tiles_n), dim3(256), 4 * TILE_BYTES, st, a);
    return (int)hipGetLastError();
}

}  // namespace
