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

#include "gemm_impl.h"
#ifndef TDC_GEMM_DIAG      // (the diagnostics tools compile this file by itself, without the profiler's translation unit)
#include "profile.h"
#endif

// e4m3-operand instantiations live in gemm_fp8.hip (a translation unit of its own: the two compile in parallel)
#ifdef TDC_GEMM_DIAG
static int tdc_gemm_fp8_impl(const tdc_gemm_desc*, hipStream_t) { return TDC_E_BADARG; }   // diagnostics tool: 16-bit only
#else
int tdc_gemm_fp8_impl(const tdc_gemm_desc* d, hipStream_t st);
#endif

int tdc_gemm_debug_mode = 0;

extern "C" int tdc_gemm_set_debug(int mode) {
    const int old = tdc_gemm_debug_mode;
    tdc_gemm_debug_mode = mode;
    if (mode != 0)
        fprintf(stderr, "[tdc_hip] WARNING: tdc_gemm diagnostic mode %d is ON - GEMM outputs are NOT valid (timing experiments "
                        "only); call tdc_gemm_set_debug(0) to restore the product path\n", mode);
    return old;
}

int tdc_gemm_persist_grid_override = 0;

extern "C" int tdc_gemm_set_persistent_grid(int workgroups) {
    const int old = tdc_gemm_persist_grid_override;
    if (workgroups != 0 && (workgroups < 8 || workgroups % 8 != 0)) {
        fprintf(stderr, "[tdc_hip] tdc_gemm_set_persistent_grid(%d) refused: 0 (one workgroup per CU) or a multiple of 8\n", workgroups);
        return -1;
    }
    tdc_gemm_persist_grid_override = workgroups;
    if (workgroups != 0)
        fprintf(stderr, "[tdc_hip] NOTE: the persistent GEMM kernel is limited to %d workgroups for this process (a CU-masked "
                        "process on part of the chip); tdc_gemm_set_persistent_grid(0) restores one per CU\n", workgroups);
    return old;
}

extern "C" int tdc_gemm(const tdc_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->W || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0) return TDC_E_BADARG;
    const int kmul = d->in_fp8 ? 2 * BK : BK;                  /* one 128-byte K tile: 64 16-bit or 128 fp8 values */
    if (d->K % kmul != 0 || d->N % 4 != 0 || (d->act == TDC_ACT_SWIGLU && d->N % 8 != 0)) {
        fprintf(stderr, "[tdc_hip] tdc_gemm: K %% %d / N %% 4 violated (M=%d N=%d K=%d)\n", kmul, d->M, d->N, d->K);
        return TDC_E_BADARG;
    }
    const int amul = d->in_fp8 ? 16 : 8;                       /* rows start on 16-byte boundaries */
    if ((d->lda % amul) || (d->ldw % amul) || (d->ldc % 4) || (d->res && (d->ldres % 4)) || d->ldw < d->K) return TDC_E_BADARG;
    /* lda < K is legal: rows of A may overlap (sliding-window views, e.g. the BEATs conv positional embedding) */
    if (d->act != TDC_ACT_NONE && d->res) return TDC_E_BADARG;  /* activation epilogues take no residual */
    if (d->a_map.seg < 0 || d->c_map.seg < 0 || d->r_map.seg < 0) return TDC_E_BADARG;
    if (d->x16) {   /* LayerNorm fusion, producer: fp32 residual-stream update with identity row maps, whole 64-column slots */
        if (!d->ln_part || !d->out_f32 || !d->res || !d->res_f32 || d->act != TDC_ACT_NONE || d->N % 64 != 0 ||
            d->c_map.seg != 0 || d->r_map.seg != 0 || d->ldx16 % 4 != 0 || d->ldx16 < d->N || ((uintptr_t)d->x16 & 7) ||
            ((uintptr_t)d->ln_part & 7))
            return TDC_E_BADARG;
    }
    if (d->ln_part && !d->x16) {   /* the same producer over a 16-bit residual stream: C = TC(acc + bias + float(res)) with
                                      16-bit C and res, only the partials are emitted (the consumer reads the stream itself) */
        if (d->out_f32 || !d->res || d->res_f32 || d->act != TDC_ACT_NONE || d->N % 64 != 0 || d->c_map.seg != 0 ||
            d->r_map.seg != 0 || d->ln_stats || d->in_fp8 || d->out_fp8 || d->c_pad8 || ((uintptr_t)d->ln_part & 7) ||
            (d->ldc & 7) || (d->ldres & 7) || ((uintptr_t)d->C & 15) || ((uintptr_t)d->res & 15))
            return TDC_E_BADARG;
    }
    if (d->ln_stats) {   /* consumer: 16-bit output without residual - or, with fp8 operands, the fp32 residual-stream
                            update (identity row maps); row statistics indexed by the A / C row */
        const bool rmw = d->in_fp8 && d->res && d->act == TDC_ACT_NONE && !d->x16 && d->c_map.seg == 0 && d->r_map.seg == 0 &&
                         ((d->out_f32 && d->res_f32) ||                                       /* fp32 residual stream */
                          (!d->out_f32 && !d->res_f32 && !d->out_fp8 && d->N % 8 == 0));       /* 16-bit stream of type `dtype` */
        if ((!d->ln_c1 && !d->in_fp8) || (!rmw && (d->out_f32 || d->res)) || d->a_map.seg != 0 || ((uintptr_t)d->ln_stats & 7) ||
            ((uintptr_t)d->ln_c1 & 15))
            return TDC_E_BADARG;
    }
    if (d->c_pad8 && (d->out_f32 || d->res || d->bias || d->out_fp8 || d->act != TDC_ACT_NONE || d->c_map.seg != 0 ||
                      d->ldc < (d->N + 7) / 8 * 8))
        return TDC_E_BADARG;
    if (d->c16_dtype_p1) {   /* C / 16-bit res of the other 16-bit type: plain 16-bit-output GEMMs (+ residual) only */
        if (d->c16_dtype_p1 < 1 || d->c16_dtype_p1 > 2 || d->out_f32 || d->act != TDC_ACT_NONE || d->x16 || d->ln_stats ||
            d->in_fp8 || d->out_fp8 || d->c_pad8)
            return TDC_E_BADARG;
        /* the epilogues without a residual store C in the operand type: a C of the OTHER 16-bit type exists only on the
           residual paths (the towers' 16-bit stream) - refuse instead of writing `dtype` bit patterns into such a buffer */
        if (d->c16_dtype_p1 - 1 != d->dtype && !d->res) {
            fprintf(stderr, "[tdc_hip] tdc_gemm: c16_dtype_p1 names the other 16-bit type but there is no residual (M=%d N=%d K=%d)\n",
                    d->M, d->N, d->K);
            return TDC_E_BADARG;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->out_fp8) {    /* e4m3 output: fp8 operands with their scales, whole 64-column wave tiles, 16-byte rows */
        if (!d->in_fp8 || !d->ln_stats || !d->out_stats || d->out_f32 || d->res || d->c_map.seg != 0 || d->N % 64 != 0 ||
            (d->ldc & 15) || d->ldc < (d->act == TDC_ACT_SWIGLU ? d->N / 2 : d->N) || ((uintptr_t)d->C & 15) ||
            ((uintptr_t)d->out_stats & 7) || !(d->out_w2max > 0.f))
            return TDC_E_BADARG;
    }
    if (d->dtype != TDC_F16 && d->dtype != TDC_BF16) return TDC_E_BADARG;      /* with fp8 operands: the type of C / res */
    if (d->in_fp8 && !d->ln_stats) {   /* e4m3 operands carry their dequantisation scales in ln_stats: without them the product
                                          would leave unscaled */
        fprintf(stderr, "[tdc_hip] tdc_gemm: in_fp8 needs ln_stats (the row / weight scales) (M=%d N=%d K=%d)\n", d->M, d->N, d->K);
        return TDC_E_BADARG;
    }
    /* every argument check is behind us: a refused launch leaves no profiler record */
#ifndef TDC_GEMM_DIAG
    TdcProfScope prof(TDC_PROF_GEMM, st, d->M, d->N, d->K, d->act, d->res ? (d->res_f32 ? 1 : 2) : 0, d->out_f32, d->W,
                      2.0 * d->M * d->N * d->K);
#endif
    if (d->in_fp8) return tdc_gemm_fp8_impl(d, st);
    // (Round 3 tried sending the last 128 columns of an N = 256 t + 128 GEMM - SigLIP's 1152-wide out-projection and fc2, whose
    // fifth column tile is half empty - to the 128 x 128 kernel as a launch of its own: 7-10 % SLOWER on both shapes; the second
    // launch streams the whole A operand again for an eighth of the columns.  Removed.)
    if (d->dtype == TDC_F16) return launch<f16, false>(d, st);
    return launch<bf16, false>(d, st);
}
