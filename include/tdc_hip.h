/* tdc_hip.h — C ABI of libtdc_hip.so: the MI355X (gfx950) implementation of TDC-Video's video-encoding hot path.
 *
 * The reference (Hoar012/TDC-Video) is 100 % Python and has no FFI layer; its boundary for this path is the Python
 * mixin `CambrianMetaForCausalLM.{encode_images, prepare_inputs_labels_for_multimodal}` (tdc/cambrian_arch.py:698,
 * :864).  Each entry point below replaces the eager-op sequence of one stage of that function; the citation next to
 * it names the reference lines it replaces.  `tdc-video_amd/` (Python) marshals torch tensors' data_ptr() into these
 * calls; INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in `_host`; nothing is allocated inside a call
 *    (composite calls take a caller-provided workspace; query its size with the matching *_workspace_bytes);
 *  - `stream` is a hipStream_t passed as void*; calls only enqueue work (no synchronisation);
 *  - return value: 0 on success, otherwise a hipError_t / negative TDC_E* code (message on stderr);
 *  - 16-bit tensors are fp16 (TDC_F16) or bf16 (TDC_BF16), accumulation / LayerNorm / softmax / GELU are fp32;
 *  - activation matrices are row-major [rows, ld] with ld = round_up(cols, 64) and ZERO pad columns; weights are
 *    nn.Linear layout [out, in] padded the same way (prepared once by tdc-video_amd/weights.py).
 */
#ifndef TDC_HIP_H
#define TDC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDC_F16 0
#define TDC_BF16 1

#define TDC_ACT_NONE 0
#define TDC_ACT_GELU_ERF 1   /* nn.GELU() default, ACT2FN["gelu"]          */
#define TDC_ACT_GELU_TANH 2  /* "gelu_pytorch_tanh" (SigLIP MLP)            */
#define TDC_ACT_SWIGLU 3     /* silu(x1)*x2 on interleaved (x1,x2) columns  */

#define TDC_E_BADARG (-2)
#define TDC_E_WORKSPACE (-3)

/* row(m) = (m / seg) * stride + off + (m % seg) * inner ; identity when seg == 0 */
typedef struct { int seg, stride, off, inner; } tdc_rowmap;

/* C[c_map(m), n] = act(sum_k A[a_map(m), k] * W[n, k] + bias[n]) + res[r_map(m), n]        (nn.Linear + epilogue)
 * K % 64 == 0, N % 4 == 0 (N % 8 == 0 for SWIGLU, output col = n/2).  Replaces every nn.Linear on the path. */
typedef struct {
    const void* A; int lda;
    const void* W; int ldw;
    void* C; int ldc;
    const float* bias;
    const void* res; int ldres;
    int M, N, K;
    int dtype, out_f32, res_f32, act;
    tdc_rowmap a_map, c_map, r_map;
    /* LayerNorm fused into the GEMMs around it (ViT towers; all NULL = plain GEMM).
     * Producer (the fp32 residual-stream GEMMs: out_f32, fp32 res, identity c_map / r_map, N % 64 == 0): with x16 != NULL
     * the updated row is also written as 16-bit to x16 [rows, ldx16] and, per row m and 64-column slot s, (mean, M2) of
     * the slot's 64 values to ln_part[(s * M + m) * 2 ..] (slot-major); tdc_ln_finalize turns them into (mean, rstd) per row.
     * Producer over a 16-bit residual stream (16-bit C and res of one type - in place when C == res -, identity c_map / r_map,
     * N % 64 == 0, 16-byte aligned rows): ln_part != NULL with x16 == NULL - only the partials are emitted, of the fp32 sums
     * acc + bias + float(res) that are rounded into C; the consumer then reads the stream itself as its A operand.
     * Consumer (16-bit output, no residual, identity a_map): with ln_stats != NULL, A holds the RAW rows x (x16 of the
     * producer) and W the gamma-folded weight W diag(gamma); the epilogue computes
     *   act(rstd[m] * (acc[m, n] - mean[m] * ln_c1[n]) + bias[n])   == act(LayerNorm(x) W^T + b)
     * with ln_stats [M, 2] = (mean, rstd), ln_c1[n] = sum_k W'[n, k] (of the 16-bit W'), bias = beta W^T + b. */
    void* x16; int ldx16; float* ln_part;
    const float* ln_stats; const float* ln_c1;
    /* in_fp8 != 0: A and W hold OCP e4m3 bytes (K % 128 == 0, lda / ldw % 16 == 0, both counted in values), accumulated by
     * v_mfma_f32_16x16x128_f8f6f4 (unit block scales); `dtype` stays the 16-bit type of C / res.  The dequantisation scales ride on the
     * LayerNorm-fold operand ln_stats[m] = (*, s_a[m] * s_w): C = act(s_a[m] s_w acc + bias).  ln_c1 is not read (may be NULL)
     * and ln_stats[m].x is used by out_fp8 only. */
    int in_fp8;
    /* out_fp8 != 0 (with in_fp8 + ln_stats, no residual, N % 64 == 0): C receives OCP e4m3 bytes [M, ldc] (ldc in bytes,
     * % 16 == 0) scaled per row by s_h[m] = B[m]^p / 448, where B[m] = ln_stats[m].y * ln_stats[m].x * out_w2max + out_bmax
     * bounds |A W^T + b| on that row (Cauchy-Schwarz: ln_stats.x = the 2-norm of the quantised A row as tdc_layernorm's
     * y8 path writes it, out_w2max = the largest 2-norm of a quantised W row, out_bmax = max|bias|) and p = 2 for SWIGLU
     * (|silu(a) b| <= |a| |b|), 1 otherwise (|gelu(a)| <= |a|).  e4m3 is a floating-point format: a scale that is safe but
     * an order of magnitude loose costs no relative precision.  out_stats[m] = (0, s_h[m] * out_wscale) are the ln_stats
     * of the fp8-operand GEMM that consumes C (out_wscale = its per-tensor weight scale). */
    int out_fp8; float* out_stats; float out_w2max, out_bmax, out_wscale;
    /* c_pad8 != 0 (16-bit C, no residual, no bias, N % 4 == 0, ldc >= round_up(N, 8)): the rows of C are writable up to
     * round_up(N, 8) columns - the 16-byte row stores of the staged epilogue may then run 4 columns past an N with
     * N % 8 == 4 (unspecified values there) instead of falling back to 8-byte stores.  Used by the transposed value
     * projection of the Q-Former (N = frames x tokens). */
    int c_pad8;
    /* c16_dtype_p1 != 0: the 16-bit type of C and of a 16-bit res is (c16_dtype_p1 - 1) instead of `dtype` (0 = `dtype`, so a
     * zero-initialised descriptor keeps the operand type).  For the residual-stream GEMMs of a tower whose operands are bf16
     * and whose residual stream is kept in fp16 (tdc_vit_model.res_dtype_p1): C = fp16(acc + bias + float(res)), one rounding.
     * Needs 16-bit C (out_f32 == 0), act == NONE, no LayerNorm-fusion / fp8 operand. */
    int c16_dtype_p1;
} tdc_gemm_desc;
int tdc_gemm(const tdc_gemm_desc* d, void* stream);
/* Diagnostics only (tools/, never the product path): 1 = skip the C-tile epilogue (nothing is written), 2 = un-staged
 * epilogue; 0 (the default) = normal operation.  Prints a warning on stderr when switched on; returns the previous mode.
 * The library reads no environment variable for this. */
int tdc_gemm_set_debug(int mode);
/* Workgroups of the persistent GEMM kernel: 0 (default) = one per CU the device reports; a multiple of 8 (>= 8, at most the
 * device's) for a process confined to part of the chip by a CU mask, which the device properties do not show.  Results never
 * depend on it.  Prints a note on stderr when set; returns the previous value, -1 when refused.  The library reads no
 * environment variable for this. */
int tdc_gemm_set_persistent_grid(int workgroups);
/* per-row e4m3 quantisation of a 16-bit matrix x [rows, ldx] (cols % 8 == 0, cols <= 4608): y8 [rows, ldy8] = x / s_a[r]
 * with s_a[r] = max|x[r]| / 448 (zero bytes up to the next multiple of 128 columns when ldy8 allows), stats[r] =
 * (0, s_a[r] * wscale) - the operands of an fp8-operand tdc_gemm whose input does not come out of a LayerNorm (the
 * attention output, the MLP hidden). */
int tdc_quantize_rows_fp8(const void* x, int ldx, int rows, int cols, int dtype, void* y8, int ldy8, float* stats,
                          float wscale, void* stream);
/* (mean, M2) partials [slots, rows, 2] of `slots` 64-column slots per row -> stats [rows, 2] = (mean, 1 / sqrt(var + eps)), biased
 * variance over slots * 64 columns (nn.LayerNorm); Chan's parallel combination in a fixed order. */
int tdc_ln_finalize(const float* ln_part, int slots, int rows, float eps, float* stats, void* stream);

/* LayerNorm over `cols` real columns of x [rows, ldx] (fp32 or 16-bit), optional additive table before the norm:
 * x'[r] = x[r] + add[(r % add_period) mapped by add_mode]; y = LN(x') * gamma + beta.  Writes y16 (16-bit, pad
 * columns zeroed, may be NULL) and/or y32 (fp32, may alias x, may be NULL).
 * add_mode 0: table row = r % add_period; 1: SVA 2x2 window position ((y&1)*2 + (x&1)) of token r % add_period on a
 * sqrt(add_period)-wide grid (tdc/vision_sampler.py:375-384). */
typedef struct {
    const void* x; int ldx; int x_f32;
    void* y16; int ldy16;
    float* y32; int ldy32;
    const float* gamma; const float* beta; float eps;
    const float* add; int ldadd; int add_period; int add_mode;
    int rows, cols, dtype;
    tdc_rowmap x_map, y_map; /* row r reads x[x_map(r)] and writes y*[y_map(r)] (identity when seg == 0) */
    /* fp8 output (may be NULL; when set it is the only output: y16 and y32 must be NULL): y8 [rows, ldy8] OCP e4m3 bytes = y / s_a[r] with the
     * per-row scale s_a[r] = max|y[r]| / 448, pad columns zero; y8_stats[r] = (1.07 ||y[r] / s_a[r]||_2, s_a[r] * y8_wscale) is
     * the ln_stats operand of the fp8-operand tdc_gemm that consumes y8 (y8_wscale = that GEMM's per-tensor weight scale;
     * the first entry - an upper bound of the 2-norm of the QUANTISED row - only matters to tdc_gemm_desc.out_fp8). */
    void* y8; int ldy8; float* y8_stats; float y8_wscale;
    /* x_dtype_p1 != 0 (with x_f32 == 0): the 16-bit type of x is (x_dtype_p1 - 1) instead of `dtype` (0 = `dtype`): an fp16
     * residual stream normalised into bf16 GEMM operands.  Needs y32 == NULL, y8 == NULL, add == NULL, cols % 8 == 0. */
    int x_dtype_p1;
} tdc_ln_desc;
int tdc_layernorm(const tdc_ln_desc* d, void* stream);

/* softmax(Q K^T * scale) V (no mask but the optional key padding mask of the biased form).  Q element (b, s, h, c) at q + b*q_bs + s*q_rs + h*head_dim + c (same for
 * k, v, o).  head_dim <= 80 (TDC_E_BADARG above).  Replaces HF SigLIP/DINOv2 attention (HF:models/siglip/modeling_siglip.py:273-308),
 * BertSelfAttention self and cross (tdc/Qformer.py:169-275). */
typedef struct {
    const void *q, *k, *v; void* o;
    long long q_bs, k_bs, v_bs, o_bs;
    int q_rs, k_rs, v_rs, o_rs;
    int batch, heads, head_dim, sq, sk;
    float scale; int dtype;
    /* optional additive score bias, fp32 (NULL = none): score(b,h,q,k) = scale*q.k + gate[(b*sq+q)*gate_rs + h] *
     * bias[h*bias_hs + q*bias_rs + k].  BEATs gated relative position bias
     * (tdc/audio_models/beats/backbone.py:650-661); needs sk % 4 == 0, head_dim <= 64. */
    const float* bias; long long bias_hs; int bias_rs;
    const float* gate; int gate_rs;
    /* optional key padding mask of the biased form (NULL = none; needs `bias`): key k of batch item b is excluded from the
     * softmax (score -inf BEFORE the bias is added, backbone.py:633-643) when key_mask[b*key_mask_bs + k] != 0.
     * key_mask_bs % 4 == 0, 4-byte aligned.  BEATs padded batches (BEATs.py:142-153: padding_mask of extract_features). */
    const unsigned char* key_mask; long long key_mask_bs;
    /* kernel form: TDC_ATTN_FORM_AUTO (0) picks by shape (the 32x32x16 kernel for long sequences at head dim 64 / 72, the
     * 16x16x32 kernels otherwise); TDC_ATTN_FORM_16X16 keeps the 16x16x32 kernels for every shape (tests compare the forms) */
    int form;
} tdc_attn_desc;
#define TDC_ATTN_FORM_AUTO 0
#define TDC_ATTN_FORM_16X16 1
/* (round 4's TDC_ATTN_FORM_PW = 2, a slower one-wave-per-SIMD prototype, left the library in round 5: any other value = AUTO;
 *  the prototype lives in tools/attention_pw/) */
int tdc_attention(const tdc_attn_desc* d, void* stream);

/* ---- small data-movement / reduction kernels ------------------------------------------------------------- */
/* pixels [B,3,H,W] (px_f32: 0 = the 16-bit type `dtype`, 1 = fp32, 2 = the OTHER 16-bit type - fp16 frames into bf16
 * towers) -> patches [B*gh*gw, ldp] 16-bit `dtype` in conv-weight order (c,ky,kx), zero pad.
 * Patch-embed conv as a GEMM (HF:models/siglip/modeling_siglip.py:124-130, dinov2/modeling_dinov2.py:140-151). */
int tdc_im2col(const void* px, int px_f32, void* patches, int ldp, int B, int H, int W, int patch, int dtype,
               void* stream);
/* x32 [B, S, ld] rows `row` of every batch := vec[ld] (cls token + pos[0]) */
int tdc_set_rows(float* x32, int ld, int B, int S, int row, const float* vec, void* stream);
/* the same into a 16-bit matrix x16 [B*S, ld] of type `dtype` (vec stays fp32): the cls row of a 16-bit residual stream */
int tdc_set_rows16(void* x16, int ld, int B, int S, int row, const float* vec, int dtype, void* stream);
/* separable 2-tap resample of a token grid: x (fp32 or 16-bit `dtype`) [B, tok_off + n_in*n_in, ldx] -> y 16-bit
 * (`out_dtype`: may differ from `dtype` - bf16 towers feeding an fp16 connector) [B, n_out*n_out, ldy]; idx0/idx1/frac are
 * device arrays [n_out] (bilinear, align_corners=False; built by host).
 * siglip_encoder.py:43-69, dino_encoder.py:81-107 (and the cls drop of feature_select :66-79). */
int tdc_resample_tokens(const void* x, int x_f32, int ldx, int tok_off, int n_in, void* y, int ldy, int n_out,
                        const int* idx0, const int* idx1, const float* frac, int B, int cols, int dtype, int out_dtype,
                        void* stream);
/* adjacent-frame cosine similarity on flattened features f [T, n] 16-bit: sims[t] = cos(f[t], f[t+1]), t < T-1
 * (tdc/cambrian_arch.py:832-842); n % 8 == 0.  scratch: tdc_frame_cossim_scratch_floats(T) floats.  The reduction
 * order is fixed (fp32), so the similarity ranking is reproducible run to run. */
int tdc_frame_cossim(const void* f, long long n, int T, float* sims, float* scratch, int dtype, void* stream);
size_t tdc_frame_cossim_scratch_floats(int T);
/* mean over the token axis: x [B, P, ld] 16-bit -> y [B, ld] 16-bit (global context, cambrian_arch.py:1009) */
int tdc_token_mean(const void* x, int P, int ld, void* y, int B, int dtype, void* stream);
/* adaptive_avg_pool1d over the token axis: y[b, k] = mean of x rows [floor(kN/K), ceil((k+1)N/K)) of frame
 * src_row[b] (b when NULL); frame f starts at row f*frame_rows (frame_rows >= N: extra rows, e.g. audio tokens, are
 * not pooled - the key frame is pooled BEFORE the audio concat, cambrian_arch.py:1609,1633-1637) */
int tdc_adaptive_pool_tokens(const void* x, int N, int frame_rows, int ld, void* y, int K, int B,
                             const int* src_row, int dtype, void* stream);
/* out[i, :cols] = table_k[row] where (k,row) = src[2i], src[2i+1]; tables: up to 4 16-bit matrices with their ld.
 * Used for unpad+newline (cambrian_arch.py:1195-1293) and token emission (:1668-1709). */
typedef struct { const void* base[4]; int ld[4]; } tdc_gather_tables;
int tdc_gather_rows(const tdc_gather_tables* t, const int* src, void* out, int ldo, int n, int cols, int dtype,
                    void* stream);
/* every row of out [rows, ld] (16-bit, ld % 8 == 0) := row[0 .. ld) (the SVA queries start as `vision_query` on every
 * window of every frame, cambrian_arch.py:1018-1023) */
int tdc_fill_rows(const void* row, void* out, int ld, int rows, void* stream);
/* rows of x [rows, ld] 16-bit scaled to unit L2 norm over `cols` (F.normalize, eps 1e-12; cambrian_arch.py:1664) */
int tdc_l2_normalize(void* x, int ld, int rows, int cols, int dtype, void* stream);
/* SVA core (tdc/vision_sampler.py:215-291): per query (frame t, window (i,j)) attend over the 2x2 windows of
 * n_towers towers: q [T*side*side, ldq]; kv[tower] [T*(side*r)^2, ldkv] with K at col 0 and V at col `dim`;
 * mask [T*side*side, n_towers*r*r] uint8; out [T*side*side, ldo].  heads x head_dim = dim. */
typedef struct {
    const void* q; int ldq;
    const void* kv[2]; int ldkv;
    const unsigned char* mask;
    void* out; int ldo;
    int T, side, r, n_towers, dim, heads, dtype;
} tdc_sva_attn_desc;
int tdc_sva_attention(const tdc_sva_attn_desc* d, void* stream);
/* Q-Former embeddings (tdc/Qformer.py:78-108): h[f, s] = LN(s < K ? query[qsrc[f], s] : word[ids[s-K]] + pos[s-K])
 * -> h32 [F*(K+Lt), ld] fp32 and h16 */
typedef struct {
    const void* query; int ldq; const int* qsrc;
    const float* word; const float* pos; int ldw; const int* ids; int Lt;
    const float* gamma; const float* beta; float eps;
    float* h32; void* h16; int ld;
    int F, K, cols, dtype;
} tdc_qembed_desc;
int tdc_qformer_embed(const tdc_qembed_desc* d, void* stream);

/* ---- composite: one ViT tower forward (a2-a4) --------------------------------------------------------------------
 * Replaces SiglipVisionTower._forward / DinoVisionTower._forward (tdc/multimodal_encoder/siglip_encoder.py:71-78,
 * dino_encoder.py:109-120) incl. the HF SiglipVisionModel / Dinov2Model they wrap: patch-embed GEMM (+pos, +cls),
 * n_layers x {LN, qkv GEMM, attention, out GEMM (+residual), LN, fc1 GEMM (+GELU-tanh / SwiGLU), fc2 GEMM (+residual)},
 * optional final LN - with `fused` the two LayerNorms of a block are not kernels: the residual-stream GEMM before them
 * emits the 16-bit row copy + per-slot statistics and the GEMM after them folds (mean, rstd) into its epilogue -, bilinear resample of the token grid (cls dropped).  Weights are the prepared (padded, fused,
 * LayerScale-folded) tensors of tdc-video_amd/weights.py; all pointers are device pointers except `layers`
 * (host array).  The residual stream is fp32 or 16-bit (res_dtype_p1).  Nothing is allocated: the caller passes a workspace of at least
 * tdc_vit_workspace_bytes() bytes (256-byte aligned). */
typedef struct { const void* w; const float* b; int n, k; } tdc_lin;      /* w [n, k] 16-bit (padded), b [n] or NULL */
typedef struct {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    tdc_lin qkv, out, fc1, fc2;
    /* LayerNorm fusion (tdc_vit_model.fused): row sums of the gamma-folded qkv / fc1 weight (tdc_gemm_desc.ln_c1); the
     * lin's weight is then W diag(gamma) and its bias beta W^T + b.  qkv_c1 == NULL: this layer's LN1 runs as a kernel
     * (layer 0, whose input comes from the patch embedding). */
    const float *qkv_c1, *fc1_c1;
    /* fp8 towers (tdc_vit_model.fp8): qkv.w / fc1.w hold e4m3 bytes [n, round_up(k, 128)] with these per-tensor scales;
     * zeros [>= max n] fp32 zeros (the ln_c1 operand of an fp8-operand tdc_gemm) */
    float qkv_wscale, fc1_wscale; const float* zeros;
    float out_wscale, fc2_wscale;    /* fp8 level 2: out.w / fc2.w are e4m3 too */
    float fc1_w2max, fc1_bmax;       /* fp8 level 3: max row 2-norm of fc1's e4m3 weight (in units of fc1_wscale), max |fc1.b|
                                        (tdc_gemm_desc.out_w2max / out_bmax) */
} tdc_vit_layer;
typedef struct {
    int dtype, dim, heads, head_dim, n_layers, patch, has_cls;
    int act;                         /* TDC_ACT_GELU_TANH (SigLIP), TDC_ACT_SWIGLU / TDC_ACT_GELU_ERF (DINOv2) */
    float eps;
    tdc_lin patch_lin;               /* [pad64(dim), pad64(3*patch*patch)] */
    const float* pos; int ldpos;     /* [P (+1), pad64(dim)] fp32 position rows for this patch grid (row 0 = cls) */
    const float* cls_row;            /* cls + pos[0], [pad64(dim)] (NULL when !has_cls) */
    const float *lnf_g, *lnf_b;      /* final LayerNorm (NULL: take the raw residual stream, SigLIP hidden_states[-1]) */
    const tdc_vit_layer* layers_host;
    int fused;                       /* pre-LayerNorms folded into the qkv / fc1 GEMMs (dim % 64 == 0) */
    int fp8;                         /* 1: the LayerNorms emit e4m3 rows + per-row scales and the qkv / fc1 GEMMs run on fp8
                                        operands (dim % 128 == 0; excludes `fused`); 2: out-proj / fc2 as well, their inputs
                                        (attention output, MLP hidden) quantised per row by tdc_quantize_rows_fp8; 3: as 2, but
                                        fc1 writes the MLP hidden as e4m3 itself (tdc_gemm_desc.out_fp8; needs fc1's output
                                        width == fc2.k) */
    int out_dtype_p1;                /* 16-bit type of tdc_vit_fwd's `out` rows, PLUS ONE: 0 = the same as `dtype` (so a
                                        zero-initialised struct behaves as before the field existed), TDC_F16 + 1 / TDC_BF16 + 1
                                        = that type (the towers - 98 % of the FLOPs - in bf16, the connector and the Q-Former
                                        behind them in fp16: compressed tokens within 1e-3 of the fp32 oracle) */
    int res_dtype_p1;                /* type of the residual stream in HBM: 0 = fp32 (read-modify-write in the out-projection /
                                        fc2 epilogues, 8 B per element), TDC_F16 + 1 / TDC_BF16 + 1 = that 16-bit type (4 B per
                                        element and half the LayerNorm input bytes; fp16 is the reference's own arithmetic: its
                                        HF towers run under torch_dtype=float16, tdc/builder.py:69, residual adds included).
                                        Sums are formed in fp32 and rounded once: x <- T16(acc + bias + float(x)).  With
                                        `fused` (round 6; needs the stream's type == `dtype`): the folded consumers read the
                                        stream itself, the read-modify-write epilogues emit only the per-slot partials and no
                                        LayerNorm kernel runs inside the layer loop.  Excludes `fp8`. */
} tdc_vit_model;
size_t tdc_vit_workspace_bytes(const tdc_vit_model* m, int B, int H, int W);
/* px [B,3,H,W] (px_f32 as in tdc_im2col: 0 = `dtype`, 1 = fp32, 2 = the other 16-bit type) -> out [B*out_grid*out_grid, ldo] 16-bit (`out_dtype_p1`); idx0/idx1/frac: bilinear tables
 * [out_grid] for the (H/patch)-wide grid (device).  H == W required (the reference pads frames to squares). */
int tdc_vit_fwd(const tdc_vit_model* m, const void* px, int px_f32, int B, int H, int W, int out_grid,
                const int* idx0, const int* idx1, const float* frac, void* out, int ldo, void* workspace,
                size_t workspace_bytes, void* stream);

/* ---- composite: the batched Q-Former TDC compressor (a13-a18) -------------------------------------------------------
 * Replaces `Qformer.bert(input_ids, query_embeds, encoder_hidden_states)` + `vision_proj` + `F.normalize`
 * (tdc/cambrian_arch.py:1653-1667, tdc/Qformer.py:804-965) for ALL compressed frames of a video at once:
 * enc [F*Nenc, ldenc] 16-bit (the frames' LLM-space tokens), query [nC*K, ldq] (query_proj of the pooled key frames),
 * qsrc[f] = chunk of frame f, ids[Lt] = BERT prompt ids (Lt may be 0) -> out [F*K, ldo] 16-bit unit-norm context rows.
 * Per layer: fused q|k|v GEMM, self-attention over S = K+Lt rows, dense+residual, LN; even layers: cross-attention of
 * the K query rows against the frame's Nenc tokens (all layers' K/V projections come from ONE stacked GEMM,
 * SURVEY D7); dual FFN (query rows / text rows).  Hidden stream fp32 + 16-bit copy. */
typedef struct {
    tdc_lin qkv, attn_out; const float *attn_ln_g, *attn_ln_b;
    int has_cross, cross_idx;
    tdc_lin cross_q, cross_out; const float *cross_ln_g, *cross_ln_b;
    tdc_lin fq1, fq2; const float *fq_ln_g, *fq_ln_b;      /* intermediate_query / output_query */
    tdc_lin ft1, ft2; const float *ft_ln_g, *ft_ln_b;      /* intermediate / output (text rows) */
    const void *cross_q_tiled, *cross_out_tiled;           /* fragment-major copies of cross_q.w / cross_out.w for the fused block
                                                              (tdc_qformer_xattn_tile_weight), NULL: the per-kernel sequence */
} tdc_qformer_layer;
typedef struct {
    int dtype, dim, heads, n_layers, H;
    float eps;
    const float *word, *pos; int ldw;
    const float *emb_ln_g, *emb_ln_b;
    tdc_lin cross_kv;                                       /* [n_cross*2*dim, pad64(H)]: (K_j | V_j) per cross layer */
    tdc_lin vision_proj;                                    /* [pad64(H), pad64(dim)] */
    const tdc_qformer_layer* layers_host;
    /* the fused cross-attention block (tdc_qformer_xattn; used when tdc_qformer_xattn_supported() and cross_k.w != NULL):
     * the same projections stacked per kind - cross_k [n_cross*dim, pad64(H)] with its bias, cross_v [n_cross*dim, pad64(H)]
     * WITHOUT bias (it is the A operand of the transposed GEMM V^T = Wv enc^T) and cross_bv [n_cross*dim] fp32 = the value
     * biases, added after the PV product */
    tdc_lin cross_k, cross_v; const float* cross_bv;
    /* how tdc_qformer_fwd runs the cross-attention block when the shape is one tdc_qformer_xattn supports (else 0):
     * 0 = per-kernel sequence; 1 = q GEMM + tdc_attention, then output projection + residual + LayerNorm in one kernel
     * (needs cross_out_tiled); 2 = the whole block in one kernel per layer (needs cross_k / cross_v / both tiled weights) */
    int xattn_mode;
} tdc_qformer_model;
size_t tdc_qformer_workspace_bytes(const tdc_qformer_model* m, int F, int K, int Lt, int Nenc);

/* ---- the Q-Former cross-attention block of one layer as ONE kernel (the north_star's kernel) ----------
 * Replaces, for the K query rows of every compressed frame, BertSelfAttention (cross form) + BertSelfOutput
 * (tdc/Qformer.py:128-130,185-188,205-264,285-289, called from BertLayer.forward :432-447):
 *     q = h[:, :K] Wq^T + bq;  ctx = softmax(q_h k_h^T * scale) v_h per head;  h[:, :K] = LayerNorm(ctx Wo^T + bo + h[:, :K])
 * h16 / h32: the 16-bit copy and the fp32 master of the hidden stream [F*S, ldh] (query rows of frame f = rows f*S + [0, K));
 * both are updated in place.  k [F*Nenc, ldk]: this layer's key rows (enc Wk^T + bk; pass the column-offset pointer);
 * vt [dim, ldvt]: this layer's values TRANSPOSED, vt[c][f*Nenc + key] = (enc Wv^T)[f*Nenc + key][c] WITHOUT the bias -
 * i.e. tdc_gemm with A = Wv, W = enc -, ldvt >= round_up(F*Nenc, 8) (with Nenc % 8 == 4 the last, half-valid key group of a
 * frame is read as one 16-byte piece: the last frame's read runs 4 columns past F*Nenc, values unused); bv [dim] fp32 or NULL.  wq / wo: the two [dim, dim] weights in the kernel's
 * FRAGMENT-MAJOR layout, dim*dim 16-bit values each, made once per weight by tdc_qformer_xattn_tile_weight from the
 * nn.Linear layout [dim, ldw] (a wave's MFMA operand is then 1 KiB of consecutive bytes instead of 16 rows x 64 B).
 * Supported: dim == 768, head dim 64, K % 16 == 0, Nenc % 4 == 0, 8 <= Nenc <= 224 (tdc_qformer_xattn_supported); the
 * composite tdc_qformer_fwd falls back to the per-kernel sequence (q GEMM, tdc_attention, dense GEMM, tdc_layernorm) otherwise. */
typedef struct {
    void* h16; float* h32; int ldh;
    int F, K, S;
    const void* wq; const float* bq; const void* wo; const float* bo;
    const void* k; int ldk;
    const void* vt; long long ldvt; const float* bv;
    int Nenc;
    const float *ln_g, *ln_b; float eps;
    int dim, heads; float scale; int dtype;
    /* ctx != NULL: the OUTPUT-PROJECTION-ONLY form - ctx [F*K, ldctx] 16-bit is the attention output of the flat query rows
     * (row f*K + k), produced by separate launches (q GEMM, tdc_attention); the kernel then computes
     * h[:, :K] = LayerNorm(ctx Wo^T + bo + h[:, :K]) alone: wq / bq / k / vt / bv / Nenc / scale are not read. */
    const void* ctx; int ldctx;
    /* res16 != 0 (ctx form only): the residual is the 16-bit hidden state - h16 is read as the residual and only h16 is written;
     * h32 is neither read nor written and may be NULL.  (The arithmetic of the reference's half-precision inference,
     * tdc/Qformer.py:285-289 on 16-bit tensors; half the bytes of the launch.  tdc_qformer_fwd uses it in its default form
     * and carries the query rows in 16 bits from here to the LayerNorm behind the layer's query FFN, which writes both copies again.) */
    int res16;
} tdc_xattn_desc;
int tdc_qformer_xattn_supported(int dim, int heads, int K, int Nenc);
int tdc_qformer_xattn(const tdc_xattn_desc* d, void* stream);
int tdc_qformer_xattn_tile_weight(const void* w, int ldw, void* out, int dtype, void* stream);
int tdc_qformer_fwd(const tdc_qformer_model* m, const void* enc, int ldenc, int F, int Nenc, const void* query, int ldq,
                    const int* qsrc, const int* ids, int Lt, int K, void* out, int ldo, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ---- composite: the connector (a6-a9) -------------------------------------------------------------------------------
 * Replaces mm_projector_aux_{0,1} + global context (tdc/cambrian_arch.py:80-90,1002-1013), the window gather + masks and
 * the 3-layer `vision_sampler_0` (cambrian_arch.py:601-695,1018-1053; tdc/vision_sampler.py:170-401,519-566) and
 * `mm_projector` (cambrian_arch.py:65-69,1149-1150) for T frames: the two towers' tokens sig [T*P, ld_s] / dino
 * [T*P, ld_d] (16-bit, P = (side*r)^2 tokens per frame) -> out [T*side*side, ldo] 16-bit in the LLM width H.
 * mask [T*side*side, 2*r*r] uint8 is the reference's window-validity mask (host geometry: tdc-video_amd/segment.py).
 * Per SVA layer: proj_context once per frame, proj_in split into its query / context column blocks (the context half is a
 * per-frame bias), k|v LayerNorm affines folded into ONE [P-token, 2C] GEMM per tower with the position embedding added
 * inside the LayerNorm launch, tdc_sva_attention, o_proj + residual, LayerNorm, proj_out MLP + residual. */
typedef struct { tdc_lin fc1, fc2; const float *ln_g, *ln_b; } tdc_aux_proj;
typedef struct {
    tdc_lin proj_context, proj_in_c, proj_in_q;
    const float* pos[2]; int ldpos;                  /* pos_embed_{0,1} [r*r, pad64(C)] fp32 */
    tdc_lin kv[2];                                   /* (K | V) projection of tower i, LayerNorm affines folded */
    const float *q_ln_g, *q_ln_b; tdc_lin q_proj, o_proj;
    const float *norm_g, *norm_b; tdc_lin out1, out2;
} tdc_sva_layer;
typedef struct {
    int dtype, C, side, r, heads, n_layers;
    tdc_aux_proj aux[2];
    const void* vision_query;                        /* [pad64(C)] 16-bit, pad columns zero */
    const float *ones_C, *zeros_C;                   /* [pad64(C)] (the un-affine LayerNorm of the k/v inputs) */
    const tdc_sva_layer* layers_host;
    tdc_lin mm1, mm2;                                /* mm_projector: Linear(C, H) + GELU, Linear(H, H) */
} tdc_connector_model;
size_t tdc_connector_workspace_bytes(const tdc_connector_model* m, int T);
int tdc_connector_fwd(const tdc_connector_model* m, const void* sig, int ld_s, const void* dino, int ld_d, int T,
                      const unsigned char* mask, void* out, int ldo, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ---- frame pre-processing on the device -----------------------------------------------------------
 * `process_images` for ONE tower (tdc/mm_datautils.py:270-314): frames uint8 [T, H, W, 3] (RGB, as decord / numpy give
 * them) -> expand2square with the pad colour -> Pillow `Image.resize((R, R))` (bicubic with antialiasing, 8 bpc:
 * byte-exact, same 22-bit fixed-point coefficients / pass order / uint8 intermediate as Resample.c) -> per-channel
 * table lut[c*256 + v] = ((v / 255) - mean[c]) / std[c] (fp32, built by the host exactly as the HF image processor
 * computes it) -> out [T, 3, R, R] 16-bit (or fp32).  bounds [R, 2] / coeffs [R, ksize] int32 are Resample.c's
 * precompute_coeffs + normalize_coeffs_8bpc for max(H,W) -> R (tdc-video_amd/preprocess.py).  scratch:
 * tdc_preprocess_scratch_bytes() bytes. */
size_t tdc_preprocess_scratch_bytes(int T, int H, int W, int R);
int tdc_preprocess_frames(const unsigned char* frames, int T, int H, int W, int R, const int* bounds, const int* coeffs,
                          int ksize, int pad_r, int pad_g, int pad_b, const float* lut, void* out, int out_f32,
                          int dtype, unsigned char* scratch, void* stream);

/* ---- BEATs audio front end --------------------------------------------------------------------------
 * kaldi fbank as BEATs.preprocess calls it (tdc/audio_models/beats/BEATs.py:115-129: torchaudio.compliance.kaldi.fbank,
 * 128 mel bins, 16 kHz, 25 ms / 10 ms, povey window, pre-emphasis 0.97, DC removal, dither 0, snip_edges) on B
 * equal-length waveforms wav [B, n_samples] (fp16 or fp32, amplitude +-1; item stride wav_bs), followed by
 * (x - mean) * inv_scale.  Host-built fp32 tables: window [400], twiddle [256][2] (cos, sin of 2 pi k / 512), banks
 * [128][257], range [128][2] (non-zero span of each mel row).  Outputs (either may be NULL): plain fp32
 * [B, frames, 128]; patches 16-bit [B * (frames/16) * 8, ldp] = the im2col of the 16x16 / stride-16 patch conv
 * (BEATs.py:145-148), row (frame/16)*8 + mel/16, column (frame%16)*16 + mel%16.  tdc_fbank_frames(n) = frame count. */
int tdc_fbank_frames(long long n_samples);
int tdc_fbank(const void* wav, int wav_f32, long long n_samples, long long wav_bs, int B, const float* window,
              const float* twiddle, const float* banks, const int* range, float* plain, void* patches, int ldp,
              int dtype, float mean, float inv_scale, void* stream);
/* gate of the gated relative position bias (backbone.py:652-657): q [rows, ldq] 16-bit (heads x head_dim columns, the
 * un-scaled q_proj output); w2 [2, head_dim], b2 [2] = grep_linear rows / biases summed in groups of 4; grep_a
 * [heads]; gate [rows, ldg] fp32 = ga * (gb * grep_a[h] - 1) + 2.  Consumed by tdc_attention's `gate`. */
int tdc_relpos_gate(const void* q, int ldq, int rows, int heads, int head_dim, const float* w2, const float* b2,
                    const float* grep_a, float* gate, int ldg, int dtype, void* stream);

/* ---- launch profiler -------------------------------------------------------------------------------------------------
 * Between tdc_profile_start and tdc_profile_stop every tdc_gemm / tdc_attention / tdc_layernorm / tdc_qformer_xattn launch -
 * called directly or from inside a composite (tdc_vit_fwd, tdc_connector_fwd, tdc_qformer_fwd) - is bracketed by two
 * hipEvents on its launch stream and leaves one record.  tdc_profile_stop waits for the recorded events, fills `recs` (at
 * most `cap`) and returns the number of launches seen (negative TDC_E* on error).  tdc_profile_tag sets the tag copied into
 * the records that follow (returns the previous one); tdc_qformer_fwd tags the launches of the cross-attention block
 *  with TDC_PROF_TAG_XATTN_BLOCK by itself.  One profile at a time per process; off by default (one global word
 * is tested per launch).  Measurement infrastructure of bench.py: the profiled step runs the same host path as the timed one. */
#define TDC_PROF_GEMM 1
#define TDC_PROF_ATTN 2
#define TDC_PROF_LN 3
#define TDC_PROF_XATTN 4
#define TDC_PROF_TAG_XATTN_BLOCK 1
typedef struct {
    int kind, tag;
    float ms;                       /* elapsed between the two events */
    int M, N, K, act, res, out_f32; /* GEMM: the launch's shape / epilogue (N, K as passed: padded); ATTN: M = batch * heads,
                                       N = sq, K = sk, act = head_dim; LN: M = rows, N = cols; XATTN: M = F * K rows, N = Nenc,
                                       act = 1 for the output-projection form */
    const void* W;                  /* GEMM: the weight pointer (callers map it to the un-padded dims) */
    double flops;                   /* algorithmic FLOPs of the launch from the dims as passed (GEMM 2 M N K, attention
                                       4 b h sq sk d, XATTN its GEMM + attention products); 0 for LN */
} tdc_prof_rec;
int tdc_profile_start(int max_records);
int tdc_profile_stop(tdc_prof_rec* recs, int cap);
int tdc_profile_tag(int tag);

/* library / device info */
const char* tdc_version(void);
int tdc_device_info(int* cu_count, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif
