// Composite entry points of the C ABI: whole-stage launch sequences over the primitive kernels (no Python between the
// launches; graph-capturable: nothing here allocates or synchronises).
#include "../../include/tdc_hip.h"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <hip/hip_runtime.h>

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int pad64i(int x) { return (x + 63) / 64 * 64; }

struct VitWs {
    size_t patches, x32, h16, qkv, attn, mlp, total;
};

VitWs vit_layout(const tdc_vit_model* m, int B, int H, int W) {
    const int gh = H / m->patch, gw = W / m->patch;
    const size_t P = (size_t)gh * gw, S = P + m->has_cls, rows = (size_t)B * S;
    const int Dp = pad64i(m->dim);
    const int kp = m->patch_lin.k;
    const int qkv_w = m->n_layers ? m->layers_host[0].qkv.n : 0;
    const int mlp_w = m->n_layers ? m->layers_host[0].fc2.k : 0;
    VitWs w;
    size_t off = 0;
    w.patches = off; off += al256((size_t)B * P * kp * 2);
    w.x32 = off;     off += al256(rows * Dp * 4);
    w.h16 = off;     off += al256(rows * Dp * 2);
    w.qkv = off;     off += al256(rows * qkv_w * 2);
    w.attn = off;    off += al256(rows * Dp * 2);
    w.mlp = off;     off += al256(rows * mlp_w * 2);
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

int layernorm(const float* x, int ldx, void* y16, int ldy, const float* g, const float* b, float eps, int rows,
              int cols, int dtype, void* st) {
    tdc_ln_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.ldx = ldx; d.x_f32 = 1; d.y16 = y16; d.ldy16 = ldy; d.gamma = g; d.beta = b; d.eps = eps;
    d.rows = rows; d.cols = cols; d.dtype = dtype;
    return tdc_layernorm(&d, st);
}

}  // namespace

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
    {
        tdc_rowmap cmap = {P, S, m->has_cls, 1}, rmap = {P, 0, m->has_cls, 1};
        RET_IF(gemm(patches, m->patch_lin.k, m->patch_lin, x32, Dp, B * P, dt, TDC_ACT_NONE, 1, m->pos, m->ldpos, 1,
                    cmap, rmap, stream));
    }
    if (m->has_cls) RET_IF(tdc_set_rows(x32, Dp, B, S, 0, m->cls_row, stream));
    // attention output pad columns must be zero (K padding of the out-projection)
    if (Dp != D) {
        // one-time clear through the LayerNorm kernel is not possible; the pad columns of `attn` are zeroed by writing
        // the whole buffer once with a gather-free memset on the stream
        hipError_t e = hipMemsetAsync(attn, 0, (size_t)rows * Dp * 2, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    const float scale = 1.0f / sqrtf((float)m->head_dim);
    for (int l = 0; l < m->n_layers; ++l) {
        const tdc_vit_layer& L = m->layers_host[l];
        RET_IF(layernorm(x32, Dp, h16, Dp, L.ln1_g, L.ln1_b, m->eps, rows, D, dt, stream));
        RET_IF(gemm(h16, Dp, L.qkv, qkv, L.qkv.n, rows, dt, TDC_ACT_NONE, 0, nullptr, 0, 0, ident, ident, stream));
        tdc_attn_desc a;
        memset(&a, 0, sizeof(a));
        const long long bs = (long long)S * L.qkv.n;
        a.q = qkv; a.k = qkv + (size_t)D * 2; a.v = qkv + (size_t)2 * D * 2; a.o = attn;
        a.q_bs = a.k_bs = a.v_bs = bs; a.o_bs = (long long)S * Dp;
        a.q_rs = a.k_rs = a.v_rs = L.qkv.n; a.o_rs = Dp;
        a.batch = B; a.heads = m->heads; a.head_dim = m->head_dim; a.sq = S; a.sk = S; a.scale = scale; a.dtype = dt;
        RET_IF(tdc_attention(&a, stream));
        RET_IF(gemm(attn, Dp, L.out, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
        RET_IF(layernorm(x32, Dp, h16, Dp, L.ln2_g, L.ln2_b, m->eps, rows, D, dt, stream));
        RET_IF(gemm(h16, Dp, L.fc1, mlp, L.fc2.k, rows, dt, m->act, 0, nullptr, 0, 0, ident, ident, stream));
        RET_IF(gemm(mlp, L.fc2.k, L.fc2, x32, Dp, rows, dt, TDC_ACT_NONE, 1, x32, Dp, 1, ident, ident, stream));
    }
    const void* src = x32;
    int src_f32 = 1;
    if (m->lnf_g) {
        RET_IF(layernorm(x32, Dp, h16, Dp, m->lnf_g, m->lnf_b, m->eps, rows, D, dt, stream));
        src = h16;
        src_f32 = 0;
    }
    return tdc_resample_tokens(src, src_f32, Dp, m->has_cls, g, out, ldo, out_grid, idx0, idx1, frac, B, D, dt, stream);
}
