// Composite entry points of the C ABI: whole-stage launch sequences over the primitive kernels (no Python between the
// launches; graph-capturable: nothing here allocates or synchronises).
#include "../../include/tdc_hip.h"
#include "profile.h"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <hip/hip_runtime.h>

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int pad64i(int x) { return (x + 63) / 64 * 64; }

struct VitWs {
    size_t patches, x32, h16, qkv, attn, mlp, part, stats, total;   // fp8 towers: h16 holds the e4m3 LayerNorm rows
};

// row stride of the 16-bit MLP hidden buffer: fc1's output width (SwiGLU halves it), at least fc2's (padded) K
int vit_mlp_ld(const tdc_vit_model* m) {
    if (!m->n_layers) return 0;
    const tdc_vit_layer& L = m->layers_host[0];
    const int n1 = m->act == TDC_ACT_SWIGLU ? L.fc1.n / 2 : L.fc1.n;
    return n1 > L.fc2.k ? n1 : L.fc2.k;
}

VitWs vit_layout(const tdc_vit_model* m, int B, int H, int W) {
    const int gh = H / m->patch, gw = W / m->patch;
    const size_t P = (size_t)gh * gw, S = P + m->has_cls, rows = (size_t)B * S;
    const int Dp = pad64i(m->dim);
    const int kp = m->patch_lin.k;
    const int qkv_w = m->n_layers ? m->layers_host[0].qkv.n : 0;
    const int mlp_w = vit_mlp_ld(m);
    VitWs w;
    size_t off = 0;
    w.patches = off; off += al256((size_t)B * P * kp * 2);
    w.x32 = off;     off += al256(rows * Dp * 4);
    w.h16 = off;     off += al256(rows * Dp * 2);
    {   // fp8 level 2 parks the e4m3 MLP hidden (rows x fc2.k bytes) in the qkv buffer
        size_t qb = rows * qkv_w * 2, mb = (m->fp8 >= 2 && m->n_layers) ? rows * (size_t)m->layers_host[0].fc2.k : 0;
        w.qkv = off; off += al256(qb > mb ? qb : mb);
    }
    w.attn = off;    off += al256(rows * Dp * 2);
    w.mlp = off;     off += al256(rows * mlp_w * 2);
    w.part = off;    off += m->fused ? al256(rows * (size_t)(Dp / 64) * 8) : 0;
    w.stats = off;   off += (m->fused || m->fp8) ? al256(rows * 8) * (m->fp8 >= 3 ? 2 : 1) : 0;   // level 3: fc1's out_stats
    w.total = off;
    return w;
}

#define RET_IF(x) do { int _rc = (x); if (_rc) { fprintf(stderr, "[tdc_hip] %s -> %d (%s:%d)\n", #x, _rc, __FILE__, __LINE__); return _rc; } } while (0)

int gemm(const void* A, int lda, const tdc_lin& L, void* C, int ldc, int M, int dtype, int act, int out_f32,
         const void* res, int ldres, int res_f32, tdc_rowmap cmap, tdc_rowmap rmap, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = C; d.ldc = ldc; d.bias = L.b; d.res = res; d.ldres = ldres;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = dtype; d.out_f32 = out_f32; d.res_f32 = res_f32; d.act = act;
    d.c_map = cmap; d.r_map = rmap;
    return tdc_gemm(&d, st);
}

// LayerNorm -> e4m3 rows + per-row scales (stats) for an fp8-operand GEMM with per-tensor weight scale `wscale`
int layernorm_fp8(const float* x, int ldx, void* y8, int ldy8, float* stats, float wscale, const float* g, const float* b,
                  float eps, int rows, int cols, int dtype, void* st) {
    tdc_ln_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.ldx = ldx; d.x_f32 = 1; d.gamma = g; d.beta = b; d.eps = eps;
    d.rows = rows; d.cols = cols; d.dtype = dtype;
    d.y8 = y8; d.ldy8 = ldy8; d.y8_stats = stats; d.y8_wscale = wscale;
    return tdc_layernorm(&d, st);
}

// fp32 residual-stream update x32 += s_a s_w (A8 W8^T) + b on fp8 operands
int gemm_fp8_rmw(const void* A8, int lda, const tdc_lin& L, float* x32, int ld, int M, int dtype, const float* stats,
                 const float* zeros, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A8; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = x32; d.ldc = ld; d.bias = L.b; d.res = x32; d.ldres = ld;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = dtype; d.out_f32 = 1; d.res_f32 = 1; d.in_fp8 = 1;
    d.ln_stats = stats; d.ln_c1 = zeros;
    return tdc_gemm(&d, st);
}

// ... over a 16-bit residual stream of type `rt` (tdc_vit_model.res_dtype_p1): x <- T16(s_a s_w acc + bias + float(x))
int gemm_fp8_rmw16(const void* A8, int lda, const tdc_lin& L, void* x16, int ld, int M, int rt, const float* stats,
                   const float* zeros, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A8; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = x16; d.ldc = ld; d.bias = L.b; d.res = x16; d.ldres = ld;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = rt; d.in_fp8 = 1;
    d.ln_stats = stats; d.ln_c1 = zeros;
    return tdc_gemm(&d, st);
}

int gemm_fp8(const void* A8, int lda, const tdc_lin& L, void* C, int ldc, int M, int dtype, int act, const float* stats,
             const float* zeros, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A8; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = C; d.ldc = ldc; d.bias = L.b;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = dtype; d.act = act; d.in_fp8 = 1; d.ln_stats = stats; d.ln_c1 = zeros;
    return tdc_gemm(&d, st);
}

// ... with e4m3 output C8 (row stride ldc bytes) + out_stats for the fp8-operand GEMM (weight scale next_wscale) after it
int gemm_fp8_out8(const void* A8, int lda, const tdc_lin& L, void* C8, int ldc, int M, int dtype, int act, const float* stats,
                  const float* zeros, float* out_stats, float w2max, float bmax, float next_wscale, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A8; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = C8; d.ldc = ldc; d.bias = L.b;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = dtype; d.act = act; d.in_fp8 = 1; d.ln_stats = stats; d.ln_c1 = zeros;
    d.out_fp8 = 1; d.out_stats = out_stats; d.out_w2max = w2max; d.out_bmax = bmax; d.out_wscale = next_wscale;
    return tdc_gemm(&d, st);
}

// tdc_gemm with the LayerNorm-fusion operands (identity row maps): producer side x16 / part, consumer side stats / c1
int gemm_ln(const void* A, int lda, const tdc_lin& L, void* C, int ldc, int M, int dtype, int act, int out_f32,
            const void* res, int ldres, void* x16, int ldx16, float* part, const float* stats, const float* c1, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = C; d.ldc = ldc; d.bias = L.b; d.res = res; d.ldres = ldres;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = dtype; d.out_f32 = out_f32; d.res_f32 = res ? 1 : 0; d.act = act;
    d.x16 = x16; d.ldx16 = ldx16; d.ln_part = part; d.ln_stats = stats; d.ln_c1 = c1;
    return tdc_gemm(&d, st);
}

int layernorm(const float* x, int ldx, void* y16, int ldy, const float* g, const float* b, float eps, int rows,
              int cols, int dtype, void* st) {
    tdc_ln_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.ldx = ldx; d.x_f32 = 1; d.y16 = y16; d.ldy16 = ldy; d.gamma = g; d.beta = b; d.eps = eps;
    d.rows = rows; d.cols = cols; d.dtype = dtype;
    return tdc_layernorm(&d, st);
}

struct QfWs {
    size_t h32, h16, kv, vt, ldvt, qkv, ctx, ctxq, cq, t32, t32b, mq, mt, total;
};

// form of the cross-attention block (tdc_qformer_model.xattn_mode, 0 when the weights or the shape do not allow the request):
// 2 = the whole block in one kernel per layer, 1 = its output projection + residual + LayerNorm in one kernel, 0 = per-kernel
int qf_mode(const tdc_qformer_model* m, int K, int Nenc) {
    if (m->xattn_mode <= 0) return 0;
    bool q_t = true, o_t = true;
    for (int l = 0; l < m->n_layers; ++l)
        if (m->layers_host[l].has_cross) {
            q_t = q_t && m->layers_host[l].cross_q_tiled;
            o_t = o_t && m->layers_host[l].cross_out_tiled;
        }
    if (m->xattn_mode >= 2 && q_t && o_t && m->cross_k.w && m->cross_v.w && m->cross_k.n == m->cross_v.n &&
        m->cross_k.k == m->cross_v.k && tdc_qformer_xattn_supported(m->dim, m->heads, K, Nenc))
        return 2;
    if (o_t && tdc_qformer_xattn_supported(m->dim, m->heads, K, 8)) return 1;
    return 0;
}
bool qf_fused(const tdc_qformer_model* m, int K, int Nenc) { return qf_mode(m, K, Nenc) == 2; }

QfWs qf_layout(const tdc_qformer_model* m, int F, int K, int Lt, int Nenc) {
    const size_t S = (size_t)K + Lt, rows = (size_t)F * S, Dp = pad64i(m->dim);
    const size_t ffn = m->layers_host[0].fq2.k;
    QfWs w;
    size_t off = 0;
    w.h32 = off;  off += al256(rows * Dp * 4);
    w.h16 = off;  off += al256(rows * Dp * 2);
    w.vt = 0; w.ldvt = 0;
    if (qf_fused(m, K, Nenc)) {     // keys [F*Nenc, n_cross*dim] and transposed values [n_cross*dim, ldvt]
        w.ldvt = ((size_t)F * Nenc + 4 + 63) / 64 * 64;     // + 4: the half-valid last key group of a frame is read as 8 columns
        w.kv = off;   off += al256((size_t)F * Nenc * m->cross_k.n * 2);
        w.vt = off;   off += al256((size_t)m->cross_v.n * w.ldvt * 2);
    } else {
        w.kv = off;   off += al256((size_t)F * Nenc * m->cross_kv.n * 2);
    }
    w.qkv = off;  off += al256(rows * m->layers_host[0].qkv.n * 2);
    w.ctx = off;  off += al256(rows * Dp * 2);
    w.ctxq = off; off += al256((size_t)F * K * Dp * 2);
    w.cq = off;   off += al256((size_t)F * K * Dp * 2);
    w.t32 = off;  off += al256(rows * Dp * 4);
    w.t32b = off; off += al256((size_t)F * (Lt > 0 ? Lt : 1) * Dp * 4);
    w.mq = off;   off += al256((size_t)F * K * ffn * 2);
    w.mt = off;   off += al256((size_t)F * (Lt > 0 ? Lt : 1) * ffn * 2);
    w.total = off;
    return w;
}

int ln_map(const void* x, int ldx, void* y16, float* y32, int ld, const float* g, const float* b, float eps, int rows,
           int cols, int dtype, tdc_rowmap ymap, void* st, int x_f32 = 1) {
    tdc_ln_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.ldx = ldx; d.x_f32 = x_f32; d.y16 = y16; d.ldy16 = ld; d.y32 = y32; d.ldy32 = ld; d.gamma = g; d.beta = b;
    d.eps = eps; d.rows = rows; d.cols = cols; d.dtype = dtype; d.y_map = ymap;
    return tdc_layernorm(&d, st);
}

int gemm_full(const void* A, int lda, const tdc_lin& L, void* C, int ldc, int M, int dtype, int act, int out_f32,
              const void* res, int ldres, int res_f32, tdc_rowmap amap, tdc_rowmap cmap, tdc_rowmap rmap, void* st) {
    tdc_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = C; d.ldc = ldc; d.bias = L.b; d.res = res; d.ldres = ldres;
    d.M = M; d.N = L.n; d.K = L.k; d.dtype = dtype; d.out_f32 = out_f32; d.res_f32 = res_f32; d.act = act;
    d.a_map = amap; d.c_map = cmap; d.r_map = rmap;
    return tdc_gemm(&d, st);
}

}  // namespace

extern "C" size_t tdc_qformer_workspace_bytes(const tdc_qformer_model* m, int F, int K, int Lt, int Nenc) {
    if (!m || !m->layers_host || F <= 0 || K <= 0 || Nenc <= 0) return 0;
    return qf_layout(m, F, K, Lt, Nenc).total;
}

extern "C" int tdc_qformer_fwd(const tdc_qformer_model* m, const void* enc, int ldenc, int F, int Nenc,
                               const void* query, int ldq, const int* qsrc, const int* ids, int Lt, int K, void* out,
                               int ldo, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !enc || !query || !qsrc || !out || !workspace || F <= 0 || K <= 0 || Lt < 0) return TDC_E_BADARG;
    const QfWs w = qf_layout(m, F, K, Lt, Nenc);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return TDC_E_WORKSPACE;
    char* ws = (char*)workspace;
    const int D = m->dim, Dp = pad64i(D), dt = m->dtype, S = K + Lt, rows = F * S;
    const int hd = D / m->heads;
    float* h32 = (float*)(ws + w.h32);
    void* h16 = ws + w.h16;
    char* kv = ws + w.kv;
    char* qkv = ws + w.qkv;
    void* ctx = ws + w.ctx;
    void* ctxq = ws + w.ctxq;
    void* cq = ws + w.cq;
    float* t32 = (float*)(ws + w.t32);
    float* t32b = (float*)(ws + w.t32b);
    void* mq = ws + w.mq;
    void* mt = ws + w.mt;
    const tdc_rowmap ident = {0, 0, 0, 0};
    const tdc_rowmap qmap = {K, S, 0, 1};
    const tdc_rowmap tmap = {Lt > 0 ? Lt : 1, S, K, 1};
    hipStream_t st = (hipStream_t)stream;
    // attention outputs only write the real columns: clear the K-padding columns once
    if (Dp != D) {
        if (hipMemsetAsync(ctx, 0, (size_t)rows * Dp * 2, st) != hipSuccess) return TDC_E_BADARG;
        if (hipMemsetAsync(ctxq, 0, (size_t)F * K * Dp * 2, st) != hipSuccess) return TDC_E_BADARG;
    }
    {
        tdc_qembed_desc e;
        memset(&e, 0, sizeof(e));
        e.query = query; e.ldq = ldq; e.qsrc = qsrc; e.word = m->word; e.pos = m->pos; e.ldw = m->ldw; e.ids = ids;
        e.Lt = Lt; e.gamma = m->emb_ln_g; e.beta = m->emb_ln_b; e.eps = m->eps; e.h32 = h32; e.h16 = h16; e.ld = Dp;
        e.F = F; e.K = K; e.cols = D; e.dtype = dt;
        RET_IF(tdc_qformer_embed(&e, stream));
    }
    const int mode = qf_mode(m, K, Nenc);
    const bool fused = mode == 2;
    {
    TdcProfTagGuard tag_kv(TDC_PROF_TAG_XATTN_BLOCK);   // the cross-attention block's launches, for tdc_profile_*
    if (fused) {
        // keys of all cross layers: one GEMM; values of all cross layers TRANSPOSED: one GEMM with the operands swapped
        // (A = Wv [n_cross*dim, H], "weight" = enc [F*Nenc, H]) - vt[c][f*Nenc + key], the A operand of the PV product
        RET_IF(gemm_full(enc, ldenc, m->cross_k, kv, m->cross_k.n, F * Nenc, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident,
                         ident, ident, stream));
        tdc_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.A = m->cross_v.w; d.lda = m->cross_v.k; d.W = enc; d.ldw = ldenc; d.C = ws + w.vt; d.ldc = (int)w.ldvt;
        d.M = m->cross_v.n; d.N = F * Nenc; d.K = m->cross_v.k; d.dtype = dt; d.c_pad8 = 1;
        RET_IF(tdc_gemm(&d, stream));
    } else {
        RET_IF(gemm_full(enc, ldenc, m->cross_kv, kv, m->cross_kv.n, F * Nenc, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident,
                         ident, ident, stream));
    }
    }
    const float scale = (float)(1.0 / sqrt((double)hd));          // float(hd ** -0.5), correctly rounded (1.0f / sqrtf is 1 ulp off at 72, 96)
    for (int l = 0; l < m->n_layers; ++l) {
        const tdc_qformer_layer& L = m->layers_host[l];
        bool q16 = false;   // this layer's query rows between the cross-attention output and the FFN LayerNorm: 16-bit only (t32 holds 16-bit rows)
        RET_IF(gemm_full(h16, Dp, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, ident,
                         stream));
        tdc_attn_desc a;
        memset(&a, 0, sizeof(a));
        a.q = qkv; a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2; a.o = ctx;
        a.q_bs = a.k_bs = a.v_bs = (long long)S * L.qkv.n; a.o_bs = (long long)S * Dp;
        a.q_rs = a.k_rs = a.v_rs = L.qkv.n; a.o_rs = Dp;
        a.batch = F; a.heads = m->heads; a.head_dim = hd; a.sq = S; a.sk = S; a.scale = scale; a.dtype = dt;
        RET_IF(tdc_attention(&a, stream));
        RET_IF(gemm_full(ctx, Dp, L.attn_out, t32, Dp, rows, dt, TDC_ACT_NONE, 1, h32, Dp, 1, ident, ident, ident,
                         stream));
        RET_IF(ln_map(t32, Dp, h16, h32, Dp, L.attn_ln_g, L.attn_ln_b, m->eps, rows, D, dt, ident, stream));
        {
        TdcProfTagGuard tag_x(L.has_cross ? TDC_PROF_TAG_XATTN_BLOCK : -1);
        if (L.has_cross && fused) {
            tdc_xattn_desc x;
            memset(&x, 0, sizeof(x));
            x.h16 = h16; x.h32 = h32; x.ldh = Dp; x.F = F; x.K = K; x.S = S;
            x.wq = L.cross_q_tiled; x.bq = L.cross_q.b; x.wo = L.cross_out_tiled; x.bo = L.cross_out.b;
            x.k = kv + (size_t)L.cross_idx * D * 2; x.ldk = m->cross_k.n;
            x.vt = ws + w.vt + (size_t)L.cross_idx * D * w.ldvt * 2; x.ldvt = (long long)w.ldvt;
            x.bv = m->cross_bv ? m->cross_bv + (size_t)L.cross_idx * D : nullptr;
            x.Nenc = Nenc; x.ln_g = L.cross_ln_g; x.ln_b = L.cross_ln_b; x.eps = m->eps;
            x.dim = D; x.heads = m->heads; x.scale = scale; x.dtype = dt;
            RET_IF(tdc_qformer_xattn(&x, stream));
        } else if (L.has_cross) {
            RET_IF(gemm_full(h16, Dp, L.cross_q, cq, Dp, F * K, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, qmap, ident, ident,
                             stream));
            tdc_attn_desc c;
            memset(&c, 0, sizeof(c));
            const size_t koff = (size_t)L.cross_idx * 2 * D * 2;
            c.q = cq; c.k = kv + koff; c.v = kv + koff + (size_t)D * 2; c.o = ctxq;
            c.q_bs = (long long)K * Dp; c.k_bs = c.v_bs = (long long)Nenc * m->cross_kv.n; c.o_bs = (long long)K * Dp;
            c.q_rs = Dp; c.k_rs = c.v_rs = m->cross_kv.n; c.o_rs = Dp;
            c.batch = F; c.heads = m->heads; c.head_dim = hd; c.sq = K; c.sk = Nenc; c.scale = scale; c.dtype = dt;
            RET_IF(tdc_attention(&c, stream));
            if (mode == 1) {     // output projection + residual + LayerNorm of the K query rows in one kernel
                tdc_xattn_desc x;
                memset(&x, 0, sizeof(x));
                x.h16 = h16; x.h32 = h32; x.ldh = Dp; x.F = F; x.K = K; x.S = S;
                x.wo = L.cross_out_tiled; x.bo = L.cross_out.b; x.ln_g = L.cross_ln_g; x.ln_b = L.cross_ln_b; x.eps = m->eps;
                x.dim = D; x.heads = m->heads; x.Nenc = Nenc; x.dtype = dt; x.ctx = ctxq; x.ldctx = Dp;
                // the residual of this kernel and of the query FFN behind it is the 16-bit hidden state (tdc_xattn_desc.res16):
                // the fp32 copy of the query rows is not touched until the FFN's LayerNorm rewrites both copies
                x.res16 = 1; x.h32 = nullptr;
                q16 = true;
                RET_IF(tdc_qformer_xattn(&x, stream));
            } else {
                RET_IF(gemm_full(ctxq, Dp, L.cross_out, t32, Dp, F * K, dt, TDC_ACT_NONE, 1, h32, Dp, 1, ident, ident, qmap,
                                 stream));
                RET_IF(ln_map(t32, Dp, h16, h32, Dp, L.cross_ln_g, L.cross_ln_b, m->eps, F * K, D, dt, qmap, stream));
            }
        }
        }
        RET_IF(gemm_full(h16, Dp, L.fq1, mq, L.fq2.k, F * K, dt, TDC_ACT_GELU_ERF, 0, nullptr, 0, 0, qmap, ident, ident,
                         stream));
        if (q16)
            RET_IF(gemm_full(mq, L.fq2.k, L.fq2, t32, Dp, F * K, dt, TDC_ACT_NONE, 0, h16, Dp, 0, ident, ident, qmap, stream));
        else
            RET_IF(gemm_full(mq, L.fq2.k, L.fq2, t32, Dp, F * K, dt, TDC_ACT_NONE, 1, h32, Dp, 1, ident, ident, qmap,
                             stream));
        if (Lt > 0) {
            RET_IF(gemm_full(h16, Dp, L.ft1, mt, L.ft2.k, F * Lt, dt, TDC_ACT_GELU_ERF, 0, nullptr, 0, 0, tmap, ident,
                             ident, stream));
            RET_IF(gemm_full(mt, L.ft2.k, L.ft2, t32b, Dp, F * Lt, dt, TDC_ACT_NONE, 1, h32, Dp, 1, ident, ident, tmap,
                             stream));
        }
        RET_IF(ln_map(t32, Dp, h16, h32, Dp, L.fq_ln_g, L.fq_ln_b, m->eps, F * K, D, dt, qmap, stream, q16 ? 0 : 1));
        if (Lt > 0)
            RET_IF(ln_map(t32b, Dp, h16, h32, Dp, L.ft_ln_g, L.ft_ln_b, m->eps, F * Lt, D, dt, tmap, stream));
    }
    RET_IF(gemm_full(h16, Dp, m->vision_proj, out, ldo, F * K, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, qmap, ident, ident,
                     stream));
    return tdc_l2_normalize(out, ldo, F * K, m->H, dt, stream);
}

extern "C" size_t tdc_vit_workspace_bytes(const tdc_vit_model* m, int B, int H, int W) {
    if (!m || B <= 0 || !m->layers_host) return 0;
    return vit_layout(m, B, H, W).total;
}

extern "C" int tdc_vit_fwd(const tdc_vit_model* m, const void* px, int px_f32, int B, int H, int W, int out_grid,
                           const int* idx0, const int* idx1, const float* frac, void* out, int ldo, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!m || !px || !out || !workspace || B <= 0 || H != W || H < m->patch) return TDC_E_BADARG;  // "valid" conv: H % patch pixels dropped
    const VitWs w = vit_layout(m, B, H, W);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return TDC_E_WORKSPACE;
    char* ws = (char*)workspace;
    const int g = H / m->patch, P = g * g, S = P + m->has_cls, rows = B * S;
    const int D = m->dim, Dp = pad64i(D), dt = m->dtype;
    void* patches = ws + w.patches;
    float* x32 = (float*)(ws + w.x32);
    void* h16 = ws + w.h16;
    char* qkv = ws + w.qkv;
    void* attn = ws + w.attn;
    void* mlp = ws + w.mlp;
    const tdc_rowmap ident = {0, 0, 0, 0};
    // patch embedding: im2col + GEMM, position rows added in the epilogue, output rows skip the cls slot
    RET_IF(tdc_im2col(px, px_f32, patches, m->patch_lin.k, B, H, W, m->patch, dt, stream));
    if (!m->res_dtype_p1) {
        tdc_rowmap cmap = {P, S, m->has_cls, 1}, rmap = {P, 0, m->has_cls, 1};
        RET_IF(gemm(patches, m->patch_lin.k, m->patch_lin, x32, Dp, B * P, dt, TDC_ACT_NONE, 1, m->pos, m->ldpos, 1,
                    cmap, rmap, stream));
        if (m->has_cls) RET_IF(tdc_set_rows(x32, Dp, B, S, 0, m->cls_row, stream));
    }
    // attention output pad columns must be zero (K padding of the out-projection)
    if (Dp != D) {
        // one-time clear through the LayerNorm kernel is not possible; the pad columns of `attn` are zeroed by writing
        // the whole buffer once with a gather-free memset on the stream
        hipError_t e = hipMemsetAsync(attn, 0, (size_t)rows * Dp * 2, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    const float scale = (float)(1.0 / sqrt((double)m->head_dim));  // float(head_dim ** -0.5), correctly rounded (1.0f / sqrtf is 1 ulp off at 72)
    const int out_dt = m->out_dtype_p1 ? m->out_dtype_p1 - 1 : dt;
    if (m->res_dtype_p1) {
        // 16-bit residual stream (tdc_vit_model.res_dtype_p1): x lives in the x32 region as rows of type `rt`; the out-projection
        // and fc2 GEMMs read-modify-write it in 16 bits (one rounding of acc + bias + float(x)), the LayerNorms read 16-bit rows
        const int rt = m->res_dtype_p1 - 1;
        if ((rt != TDC_F16 && rt != TDC_BF16) || (out_dt != TDC_F16 && out_dt != TDC_BF16)) return TDC_E_BADARG;
        if (m->fp8 && (m->fused || D % 128 != 0)) return TDC_E_BADARG;
        // fused over the 16-bit stream: the folded consumers (qkv of layers >= 1, every fc1) read the stream ITSELF as their A
        // operand - operands of the stream's type, whole 64-column slots - and the out-projection / fc2 epilogues emit only the
        // per-slot (mean, M2) partials: no LayerNorm kernel and no 16-bit row copy inside the layer loop
        const bool fused16 = m->fused != 0;
        if (fused16 && (rt != dt || D % 64 != 0)) return TDC_E_BADARG;
        void* x16 = x32;
        float* part16 = (float*)(ws + w.part);
        float* stats16 = (float*)(ws + w.stats);
        auto gemm_c16 = [&](const void* A, int lda, const tdc_lin& L, int M, const void* res, int ldres, int res_f32,
                            tdc_rowmap cmap, tdc_rowmap rmap, float* part = nullptr) {
            tdc_gemm_desc d;
            memset(&d, 0, sizeof(d));
            d.A = A; d.lda = lda; d.W = L.w; d.ldw = L.k; d.C = x16; d.ldc = Dp; d.bias = L.b; d.res = res; d.ldres = ldres;
            d.M = M; d.N = L.n; d.K = L.k; d.dtype = dt; d.res_f32 = res_f32; d.c_map = cmap; d.r_map = rmap;
            d.c16_dtype_p1 = rt + 1;
            d.ln_part = part;
            return tdc_gemm(&d, stream);
        };
        auto ln16 = [&](const float* g, const float* b) {
            tdc_ln_desc d;
            memset(&d, 0, sizeof(d));
            d.x = x16; d.ldx = Dp; d.x_f32 = 0; d.x_dtype_p1 = rt + 1; d.y16 = h16; d.ldy16 = Dp; d.gamma = g; d.beta = b;
            d.eps = m->eps; d.rows = rows; d.cols = D; d.dtype = dt;
            return tdc_layernorm(&d, stream);
        };
        {
            tdc_rowmap cmap = {P, S, m->has_cls, 1}, rmap = {P, 0, m->has_cls, 1};
            RET_IF(gemm_c16(patches, m->patch_lin.k, m->patch_lin, B * P, m->pos, m->ldpos, 1, cmap, rmap));
        }
        if (m->has_cls) RET_IF(tdc_set_rows16(x16, Dp, B, S, 0, m->cls_row, rt, stream));
        for (int l = 0; l < m->n_layers; ++l) {
            const tdc_vit_layer& L = m->layers_host[l];
            if (m->fp8) {
                // e4m3 operands over the 16-bit stream (BASELINE config 5 at the final code): the LayerNorm kernel reads the 16-bit
                // rows and writes e4m3 rows + scales; out-projection / fc2 read-modify-write the stream in 16 bits - on 16-bit
                // operands (level 1) or on e4m3 operands with the scales in the fold operands (levels 2, 3)
                auto ln8 = [&](const float* g_, const float* b_, int ldy8, float wscale) {
                    tdc_ln_desc d;
                    memset(&d, 0, sizeof(d));
                    d.x = x16; d.ldx = Dp; d.x_f32 = 0; d.dtype = rt; d.gamma = g_; d.beta = b_; d.eps = m->eps;
                    d.rows = rows; d.cols = D; d.y8 = h16; d.ldy8 = ldy8; d.y8_stats = stats16; d.y8_wscale = wscale;
                    return tdc_layernorm(&d, stream);
                };
                RET_IF(ln8(L.ln1_g, L.ln1_b, L.qkv.k, L.qkv_wscale));
                RET_IF(gemm_fp8(h16, L.qkv.k, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, stats16, L.zeros, stream));
                tdc_attn_desc a;
                memset(&a, 0, sizeof(a));
                const long long bs = (long long)S * L.qkv.n;
                a.q = qkv; a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2; a.o = attn;
                a.q_bs = a.k_bs = a.v_bs = bs; a.o_bs = (long long)S * Dp;
                a.q_rs = a.k_rs = a.v_rs = L.qkv.n; a.o_rs = Dp;
                a.batch = B; a.heads = m->heads; a.head_dim = m->head_dim; a.sq = S; a.sk = S; a.scale = scale; a.dtype = dt;
                RET_IF(tdc_attention(&a, stream));
                if (m->fp8 >= 2) {
                    RET_IF(tdc_quantize_rows_fp8(attn, Dp, rows, Dp, dt, h16, L.out.k, stats16, L.out_wscale, stream));
                    RET_IF(gemm_fp8_rmw16(h16, L.out.k, L.out, x16, Dp, rows, rt, stats16, L.zeros, stream));
                } else {
                    RET_IF(gemm_c16(attn, Dp, L.out, rows, x16, Dp, 0, ident, ident));
                }
                RET_IF(ln8(L.ln2_g, L.ln2_b, L.fc1.k, L.fc1_wscale));
                const int mlp_ld8 = vit_mlp_ld(m);
                const int mlp_n = m->act == TDC_ACT_SWIGLU ? L.fc1.n / 2 : L.fc1.n;
                if (m->fp8 >= 3) {
                    float* stats2 = stats16 + al256((size_t)rows * 8) / 4;
                    if (mlp_n != L.fc2.k) return TDC_E_BADARG;
                    RET_IF(gemm_fp8_out8(h16, L.fc1.k, L.fc1, qkv, L.fc2.k, rows, dt, m->act, stats16, L.zeros, stats2, L.fc1_w2max,
                                         L.fc1_bmax, L.fc2_wscale, stream));
                    RET_IF(gemm_fp8_rmw16(qkv, L.fc2.k, L.fc2, x16, Dp, rows, rt, stats2, L.zeros, stream));
                } else {
                    RET_IF(gemm_fp8(h16, L.fc1.k, L.fc1, mlp, mlp_ld8, rows, dt, m->act, stats16, L.zeros, stream));
                    if (m->fp8 == 2) {
                        RET_IF(tdc_quantize_rows_fp8(mlp, mlp_ld8, rows, mlp_n, dt, qkv, L.fc2.k, stats16, L.fc2_wscale, stream));
                        RET_IF(gemm_fp8_rmw16(qkv, L.fc2.k, L.fc2, x16, Dp, rows, rt, stats16, L.zeros, stream));
                    } else {
                        RET_IF(gemm_c16(mlp, L.fc2.k, L.fc2, rows, x16, Dp, 0, ident, ident));
                    }
                }
                continue;
            }
            if (fused16 && L.qkv_c1) {
                RET_IF(gemm_ln(x16, Dp, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, nullptr, 0, nullptr, stats16,
                               L.qkv_c1, stream));
            } else {
                RET_IF(ln16(L.ln1_g, L.ln1_b));
                RET_IF(gemm(h16, Dp, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream));
            }
            tdc_attn_desc a;
            memset(&a, 0, sizeof(a));
            const long long bs = (long long)S * L.qkv.n;
            a.q = qkv; a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2; a.o = attn;
            a.q_bs = a.k_bs = a.v_bs = bs; a.o_bs = (long long)S * Dp;
            a.q_rs = a.k_rs = a.v_rs = L.qkv.n; a.o_rs = Dp;
            a.batch = B; a.heads = m->heads; a.head_dim = m->head_dim; a.sq = S; a.sk = S; a.scale = scale; a.dtype = dt;
            RET_IF(tdc_attention(&a, stream));
            if (fused16) {
                RET_IF(gemm_c16(attn, Dp, L.out, rows, x16, Dp, 0, ident, ident, part16));
                RET_IF(tdc_ln_finalize(part16, D / 64, rows, m->eps, stats16, stream));
                RET_IF(gemm_ln(x16, Dp, L.fc1, mlp, L.fc2.k, rows, dt, m->act, 0, nullptr, 0, nullptr, 0, nullptr, stats16,
                               L.fc1_c1, stream));
                if (l + 1 < m->n_layers) {
                    RET_IF(gemm_c16(mlp, L.fc2.k, L.fc2, rows, x16, Dp, 0, ident, ident, part16));
                    RET_IF(tdc_ln_finalize(part16, D / 64, rows, m->eps, stats16, stream));
                } else {
                    RET_IF(gemm_c16(mlp, L.fc2.k, L.fc2, rows, x16, Dp, 0, ident, ident));
                }
                continue;
            }
            RET_IF(gemm_c16(attn, Dp, L.out, rows, x16, Dp, 0, ident, ident));
            RET_IF(ln16(L.ln2_g, L.ln2_b));
            RET_IF(gemm(h16, Dp, L.fc1, mlp, L.fc2.k, rows, dt, m->act, 0, nullptr, 0, 0, ident, ident, stream));
            RET_IF(gemm_c16(mlp, L.fc2.k, L.fc2, rows, x16, Dp, 0, ident, ident));
        }
        const void* src16 = x16;
        int src_dt = rt;
        if (m->lnf_g) {
            RET_IF(ln16(m->lnf_g, m->lnf_b));
            src16 = h16;
            src_dt = dt;
        }
        return tdc_resample_tokens(src16, 0, Dp, m->has_cls, g, out, ldo, out_grid, idx0, idx1, frac, B, D, src_dt, out_dt, stream);
    }
    // fused: the block's LayerNorms are folded into the GEMMs around them (tdc_gemm_desc: x16 / ln_part / ln_stats / ln_c1);
    // h16 then holds the 16-bit copy of the residual stream instead of the LayerNorm output
    const bool fused = m->fused != 0;
    if ((fused && (D % 64 != 0)) || (m->fp8 && (fused || D % 128 != 0))) return TDC_E_BADARG;
    float* part = (float*)(ws + w.part);
    float* stats = (float*)(ws + w.stats);
    const int slots = D / 64;
    const int mlp_ld = vit_mlp_ld(m);            // row stride of the MLP hidden buffer (16-bit values)
    for (int l = 0; l < m->n_layers; ++l) {
        const tdc_vit_layer& L = m->layers_host[l];
        if (m->fp8) {
            RET_IF(layernorm_fp8(x32, Dp, h16, L.qkv.k, stats, L.qkv_wscale, L.ln1_g, L.ln1_b, m->eps, rows, D, dt, stream));
            RET_IF(gemm_fp8(h16, L.qkv.k, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, stats, L.zeros, stream));
        } else if (!fused || !L.qkv_c1) {
            RET_IF(layernorm(x32, Dp, h16, Dp, L.ln1_g, L.ln1_b, m->eps, rows, D, dt, stream));
            RET_IF(gemm(h16, Dp, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream));
        } else {
            RET_IF(gemm_ln(h16, Dp, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, nullptr, 0, nullptr, stats,
                           L.qkv_c1, stream));
        }
        tdc_attn_desc a;
        memset(&a, 0, sizeof(a));
        const long long bs = (long long)S * L.qkv.n;
        a.q = qkv; a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2; a.o = attn;
        a.q_bs = a.k_bs = a.v_bs = bs; a.o_bs = (long long)S * Dp;
        a.q_rs = a.k_rs = a.v_rs = L.qkv.n; a.o_rs = Dp;
        a.batch = B; a.heads = m->heads; a.head_dim = m->head_dim; a.sq = S; a.sk = S; a.scale = scale; a.dtype = dt;
        RET_IF(tdc_attention(&a, stream));
        if (fused) {
            RET_IF(gemm_ln(attn, Dp, L.out, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, h16, Dp, part, nullptr, nullptr,
                           stream));
            RET_IF(tdc_ln_finalize(part, slots, rows, m->eps, stats, stream));
            RET_IF(gemm_ln(h16, Dp, L.fc1, mlp, L.fc2.k, rows, dt, m->act, 0, nullptr, 0, nullptr, 0, nullptr, stats,
                           L.fc1_c1, stream));
            if (l + 1 < m->n_layers) {
                RET_IF(gemm_ln(mlp, L.fc2.k, L.fc2, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, h16, Dp, part, nullptr,
                               nullptr, stream));
                RET_IF(tdc_ln_finalize(part, slots, rows, m->eps, stats, stream));
            } else {
                RET_IF(gemm(mlp, L.fc2.k, L.fc2, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
            }
        } else if (m->fp8) {
            if (m->fp8 >= 2) {   // attention output -> e4m3 rows (in h16: the LayerNorm rows are consumed) -> out-proj on fp8
                RET_IF(tdc_quantize_rows_fp8(attn, Dp, rows, Dp, dt, h16, L.out.k, stats, L.out_wscale, stream));
                RET_IF(gemm_fp8_rmw(h16, L.out.k, L.out, x32, Dp, rows, dt, stats, L.zeros, stream));
            } else {
                RET_IF(gemm(attn, Dp, L.out, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
            }
            RET_IF(layernorm_fp8(x32, Dp, h16, L.fc1.k, stats, L.fc1_wscale, L.ln2_g, L.ln2_b, m->eps, rows, D, dt, stream));
            const int mlp_n = m->act == TDC_ACT_SWIGLU ? L.fc1.n / 2 : L.fc1.n;   // columns fc1 writes (pad columns: 0)
            if (m->fp8 >= 3) {   // fc1 writes the e4m3 MLP hidden + its row scales itself (qkv buffer, second stats array)
                float* stats2 = stats + al256((size_t)rows * 8) / 4;
                if (mlp_n != L.fc2.k) return TDC_E_BADARG;
                RET_IF(gemm_fp8_out8(h16, L.fc1.k, L.fc1, qkv, L.fc2.k, rows, dt, m->act, stats, L.zeros, stats2, L.fc1_w2max,
                                     L.fc1_bmax, L.fc2_wscale, stream));
                RET_IF(gemm_fp8_rmw(qkv, L.fc2.k, L.fc2, x32, Dp, rows, dt, stats2, L.zeros, stream));
            } else {
                RET_IF(gemm_fp8(h16, L.fc1.k, L.fc1, mlp, mlp_ld, rows, dt, m->act, stats, L.zeros, stream));
                if (m->fp8 == 2) {   // MLP hidden -> e4m3 rows (in the qkv buffer, free after the attention) -> fc2 on fp8
                    RET_IF(tdc_quantize_rows_fp8(mlp, mlp_ld, rows, mlp_n, dt, qkv, L.fc2.k, stats, L.fc2_wscale, stream));
                    RET_IF(gemm_fp8_rmw(qkv, L.fc2.k, L.fc2, x32, Dp, rows, dt, stats, L.zeros, stream));
                } else {
                    RET_IF(gemm(mlp, L.fc2.k, L.fc2, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
                }
            }
        } else {
            RET_IF(gemm(attn, Dp, L.out, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
            RET_IF(layernorm(x32, Dp, h16, Dp, L.ln2_g, L.ln2_b, m->eps, rows, D, dt, stream));
            RET_IF(gemm(h16, Dp, L.fc1, mlp, L.fc2.k, rows, dt, m->act, 0, nullptr, 0, 0, ident, ident, stream));
            RET_IF(gemm(mlp, L.fc2.k, L.fc2, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
        }
    }
    const void* src = x32;
    int src_f32 = 1;
    if (m->lnf_g) {
        RET_IF(layernorm(x32, Dp, h16, Dp, m->lnf_g, m->lnf_b, m->eps, rows, D, dt, stream));
        src = h16;
        src_f32 = 0;
    }
    return tdc_resample_tokens(src, src_f32, Dp, m->has_cls, g, out, ldo, out_grid, idx0, idx1, frac, B, D, dt, out_dt, stream);
}

// ---- connector (a6-a9) -------------------------------------------------------------------------------------------------
namespace {
struct ConnWs {
    size_t h, y32, aux[2], ctx, cproj, cin, q16[2], qin, xn, kv[2], qn, qs, att, q2, mh, total;
};

ConnWs conn_layout(const tdc_connector_model* m, int T) {
    const size_t Cp = pad64i(m->C), P = (size_t)(m->side * m->r) * (m->side * m->r), nq = (size_t)m->side * m->side;
    const size_t rows = (size_t)T * P, qrows = (size_t)T * nq, Hp = m->mm1.n;
    ConnWs w;
    size_t off = 0;
    w.h = off;       off += al256(rows * Cp * 2);             // aux fc1 output
    w.y32 = off;     off += al256(rows * Cp * 4);             // aux fc2 output (fp32, LayerNorm input)
    for (int i = 0; i < 2; ++i) { w.aux[i] = off; off += al256(rows * Cp * 2); }
    w.ctx = off;     off += al256((size_t)T * Cp * 2);
    w.cproj = off;   off += al256((size_t)T * Cp * 2);
    w.cin = off;     off += al256((size_t)T * Cp * 4);
    for (int i = 0; i < 2; ++i) { w.q16[i] = off; off += al256(qrows * Cp * 2); }
    w.qin = off;     off += al256(qrows * Cp * 4);
    w.xn = off;      off += al256(rows * Cp * 2);
    for (int i = 0; i < 2; ++i) { w.kv[i] = off; off += al256(rows * 2 * Cp * 2); }
    w.qn = off;      off += al256(qrows * Cp * 2);
    w.qs = off;      off += al256(qrows * Cp * 2);
    w.att = off;     off += al256(qrows * Cp * 2);
    w.q2 = off;      off += al256(qrows * Cp * 4);
    w.mh = off;      off += al256(qrows * (Hp > Cp ? Hp : Cp) * 2);   // proj_out hidden / mm_projector hidden
    w.total = off;
    return w;
}

int ln16(const float* x, int ldx, void* y16, int ldy, const float* g, const float* b, float eps, int rows, int cols,
         int dtype, void* st) {
    return layernorm(x, ldx, y16, ldy, g, b, eps, rows, cols, dtype, st);
}
}  // namespace

extern "C" size_t tdc_connector_workspace_bytes(const tdc_connector_model* m, int T) {
    if (!m || T <= 0 || !m->layers_host) return 0;
    return conn_layout(m, T).total;
}

extern "C" int tdc_connector_fwd(const tdc_connector_model* m, const void* sig, int ld_s, const void* dino, int ld_d,
                                 int T, const unsigned char* mask, void* out, int ldo, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!m || !sig || !dino || !mask || !out || !workspace || T <= 0 || m->n_layers <= 0) return TDC_E_BADARG;
    const ConnWs w = conn_layout(m, T);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return TDC_E_WORKSPACE;
    char* ws = (char*)workspace;
    const int C = m->C, Cp = pad64i(C), dt = m->dtype, side = m->side, r = m->r;
    const int P = side * r * side * r, nq = side * side, rows = T * P, qrows = T * nq;
    const tdc_rowmap ident = {0, 0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    // a6: mm_projector_aux_i = Linear + GELU(erf), Linear, LayerNorm(1e-5); global context = mean over the tokens of aux_0
    const void* feats[2] = {sig, dino};
    const int lds[2] = {ld_s, ld_d};
    for (int i = 0; i < 2; ++i) {
        const tdc_aux_proj& a = m->aux[i];
        RET_IF(gemm(feats[i], lds[i], a.fc1, ws + w.h, Cp, rows, dt, TDC_ACT_GELU_ERF, 0, nullptr, 0, 0, ident, ident, stream));
        RET_IF(gemm(ws + w.h, Cp, a.fc2, ws + w.y32, Cp, rows, dt, TDC_ACT_NONE, 1, nullptr, 0, 0, ident, ident, stream));
        RET_IF(ln16((const float*)(ws + w.y32), Cp, ws + w.aux[i], Cp, a.ln_g, a.ln_b, 1e-5f, rows, C, dt, stream));
    }
    RET_IF(tdc_token_mean(ws + w.aux[0], P, Cp, ws + w.ctx, T, dt, stream));
    // queries start as vision_query broadcast to every window of every frame (cambrian_arch.py:1018-1023)
    RET_IF(tdc_fill_rows(m->vision_query, ws + w.q16[0], Cp, qrows, stream));
    int cur = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        const tdc_sva_layer& L = m->layers_host[l];
        char* q16 = ws + w.q16[cur];
        char* q16n = ws + w.q16[cur ^ 1];
        RET_IF(gemm(ws + w.ctx, Cp, L.proj_context, ws + w.cproj, Cp, T, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream));
        RET_IF(gemm(ws + w.cproj, Cp, L.proj_in_c, ws + w.cin, Cp, T, dt, TDC_ACT_NONE, 1, nullptr, 0, 0, ident, ident, stream));
        {   // proj_in(cat[q, ctx]) = proj_in_q(q) + (per-frame) proj_in_c(ctx)
            const tdc_rowmap rmap = {nq, 1, 0, 0};
            RET_IF(gemm(q16, Cp, L.proj_in_q, ws + w.qin, Cp, qrows, dt, TDC_ACT_NONE, 1, ws + w.cin, Cp, 1, ident, rmap, stream));
        }
        for (int tw = 0; tw < 2; ++tw) {
            tdc_ln_desc d;
            memset(&d, 0, sizeof(d));
            d.x = ws + w.aux[tw]; d.ldx = Cp; d.x_f32 = 0; d.y16 = ws + w.xn; d.ldy16 = Cp;
            d.gamma = m->ones_C; d.beta = m->zeros_C; d.eps = 1e-5f;
            d.add = L.pos[tw]; d.ldadd = L.ldpos; d.add_period = P; d.add_mode = 1;
            d.rows = rows; d.cols = C; d.dtype = dt;
            RET_IF(tdc_layernorm(&d, stream));
            RET_IF(gemm(ws + w.xn, Cp, L.kv[tw], ws + w.kv[tw], L.kv[tw].n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream));
        }
        RET_IF(ln16((const float*)(ws + w.qin), Cp, ws + w.qn, Cp, L.q_ln_g, L.q_ln_b, 1e-5f, qrows, C, dt, stream));
        RET_IF(gemm(ws + w.qn, Cp, L.q_proj, ws + w.qs, Cp, qrows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream));
        {
            tdc_sva_attn_desc a;
            memset(&a, 0, sizeof(a));
            a.q = ws + w.qs; a.ldq = Cp; a.kv[0] = ws + w.kv[0]; a.kv[1] = ws + w.kv[1]; a.ldkv = L.kv[0].n;
            a.mask = mask; a.out = ws + w.att; a.ldo = Cp;
            a.T = T; a.side = side; a.r = r; a.n_towers = 2; a.dim = C; a.heads = m->heads; a.dtype = dt;
            if (Cp != C && hipMemsetAsync(ws + w.att, 0, (size_t)qrows * Cp * 2, st) != hipSuccess) return TDC_E_BADARG;
            RET_IF(tdc_sva_attention(&a, stream));
        }
        RET_IF(gemm(ws + w.att, Cp, L.o_proj, ws + w.q2, Cp, qrows, dt, TDC_ACT_NONE, 1, ws + w.qin, Cp, 1, ident, ident, stream));
        RET_IF(ln16((const float*)(ws + w.q2), Cp, ws + w.qn, Cp, L.norm_g, L.norm_b, 1e-5f, qrows, C, dt, stream));
        RET_IF(gemm(ws + w.qn, Cp, L.out1, ws + w.mh, Cp, qrows, dt, TDC_ACT_GELU_ERF, 0, nullptr, 0, 0, ident, ident, stream));
        RET_IF(gemm(ws + w.mh, Cp, L.out2, q16n, Cp, qrows, dt, TDC_ACT_NONE, 0, q16, Cp, 0, ident, ident, stream));
        cur ^= 1;
    }
    // a9: mm_projector
    RET_IF(gemm(ws + w.q16[cur], Cp, m->mm1, ws + w.mh, m->mm1.n, qrows, dt, TDC_ACT_GELU_ERF, 0, nullptr, 0, 0, ident, ident, stream));
    return gemm(ws + w.mh, m->mm1.n, m->mm2, out, ldo, qrows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream);
}
