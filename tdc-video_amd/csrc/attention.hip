// tdc_attention: flash-style softmax(Q K^T * scale) V for the ViT towers (S = 729/730, d = 72/64) and the Q-Former
// self / cross attention (d = 64).  No mask on this path (BERT masks are all-zero).
//
// Workgroup = 4 waves; each wave owns QT tiles of 16 query rows (BQ = 64*QT rows per workgroup) of one (batch, head);
// K/V tiles of 64 keys are staged global -> registers -> LDS (loads for tile t+1 are issued before the MFMAs of tile
// t, written after the barrier: the "async-STAGE split", T14).
//  * S^T = K Q^T is computed with the KEY on the MFMA row: lane (g = lane>>4, i = lane&15) holds, for query i, the
//    scores of keys 16*kt + 4g + reg.  Row max / row sum are then 16 in-register ops + two xor-shuffles (16, 32).
//  * O^T = V^T P^T: the V^T operand comes from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose
//    read, T10); the P^T operand is the S^T accumulator converted in place (k-slot (g, j) <-> key
//    32s + 16(j>>2) + 4g + (j&3), the same permutation on both operands).  The output accumulator has the query on
//    the lane (same as the statistics: no cross-lane traffic for the rescale) and 4 consecutive head-dim columns in
//    its 4 registers (8-byte stores).
//  * K tile rows are XOR-swizzled in 16-byte chunks (chunk ^ (row & (NCH-1))), V rows padded to 160 B: both read
//    patterns are bank-conflict free for d = 64/72.
#include "common.h"
#include "../../include/tdc_hip.h"
#include "attention_args.h"
#include "profile.h"
#include <stdio.h>
#include <type_traits>

namespace {

constexpr int KT = 64;  // keys per tile

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements is returned column-major
// (lane 4q+p supplies the address of row q, cols 4p..4p+3; lane i receives column i, row q in element q).
template <class T> __device__ __forceinline__ typename VecOf<T>::v4 tr_read(const T* p) {
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(typename VecOf<T>::v4, r);
}

// DK: padded head dim for the QK^T contraction (32/64/96); NDV: number of 16-wide output column tiles; QT: q tiles/wave
template <class T, int DK, int NDV, int QT, bool VEC, bool BIAS>
__global__ __launch_bounds__(256, (QT <= 2 ? 2 : 1)) void attn_kernel(AttnArgs p) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    constexpr int NCH = DK / 8;                                 // real 16-B chunks per K row
    constexpr int NCHP = (NCH <= 4) ? 4 : (NCH <= 8 ? 8 : 16);  // chunks per row incl. padding (power of 2)
    constexpr int KROW = NCHP * 8;                              // K row stride in elements
    constexpr int VCH = NDV * 2;                                // 16-B chunks per V row
    constexpr int VROW = 80;                                    // V row stride in elements (160 B)
    static_assert(NDV * 16 <= VROW, "V row");
    constexpr int KSTEPS = DK / 32;
    __shared__ __attribute__((aligned(16))) T Ks[KT * KROW];
    __shared__ __attribute__((aligned(16))) T Vs[KT * VROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const int d = p.d;
    const T* Q = (const T*)p.q + b * p.q_bs + h * d;
    const T* K = (const T*)p.k + b * p.k_bs + h * d;
    const T* V = (const T*)p.v + b * p.v_bs + h * d;
    T* O = (T*)p.o + b * p.o_bs + h * d;

    // 8 elements row[c0..c0+7].  Loads are UNCONDITIONAL (address clamped into the row) so that the compiler can keep
    // all of a tile's loads in flight.  Only Q's columns beyond the head dim are zeroed (`zero_tail`): K's clamped
    // (finite, duplicated) tail then contributes 0 to the scores, V's tail only feeds output columns >= d that are
    // never stored, and key rows beyond sk are clamped duplicates whose probabilities are exactly 0.
    auto load8 = [&](const T* row, int c0) -> v8 {
        v8 r;
        if (VEC) {
            const int cc = c0 < d ? c0 : d - 8;
            r = *(const v8*)(row + cc);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int cc = c0 + e < d ? c0 + e : d - 1; r[e] = row[cc]; }
        }
        return r;
    };
    auto zero_tail = [&](v8 r, int c0) -> v8 {
        if (VEC) {
            if (c0 >= d)
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (T)0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (c0 + e >= d) r[e] = (T)0.f;
        }
        return r;
    };

    // ---- Q fragments (B operand: lane holds Q[q0 + li][32 ks + 8 g .. +7]); scores are scaled in fp32 afterwards
    v8 qf[QT][KSTEPS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int qr = q0 + t * 16 + li;
        if (qr > p.sq - 1) qr = p.sq - 1;
        const T* row = Q + (long long)qr * p.q_rs;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            qf[t][ks] = zero_tail(load8(row, ks * 32 + g * 8), ks * 32 + g * 8);
        }
    }

    // ---- staging: K tile = 64 x NCH chunks, V tile = 64 x VCH chunks, 256 threads
    constexpr int KLD = (KT * NCH + 255) / 256, VLD = (KT * VCH + 255) / 256;
    v8 kreg[KLD], vreg[VLD];
    // loads are issued unconditionally (indices clamped) so no exec-masked branch - and no vmcnt(0) - sits between them;
    // only the LDS writes of a partially used last round are guarded
    constexpr bool K_EXACT = (KT * NCH) % 256 == 0, V_EXACT = (KT * VCH) % 256 == 0;
    auto issue_loads = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            int idx = tid + i * 256;
            if (!K_EXACT && idx > KT * NCH - 1) idx = KT * NCH - 1;
            const int key = idx / NCH, c = idx - key * NCH;
            int kr = kv0 + key; if (kr > p.sk - 1) kr = p.sk - 1;
            kreg[i] = load8(K + (long long)kr * p.k_rs, c * 8);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            int idx = tid + i * 256;
            if (!V_EXACT && idx > KT * VCH - 1) idx = KT * VCH - 1;
            const int key = idx / VCH, c = idx - key * VCH;
            int kr = kv0 + key; if (kr > p.sk - 1) kr = p.sk - 1;
            vreg[i] = load8(V + (long long)kr * p.v_rs, c * 8);
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / NCH, c = idx - key * NCH;
            if (K_EXACT || idx < KT * NCH) *(v8*)(Ks + key * KROW + ((c ^ (key & (NCHP - 1))) << 3)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / VCH, c = idx - key * VCH;
            if (V_EXACT || idx < KT * VCH) *(v8*)(Vs + key * VROW + (c << 3)) = vreg[i];
        }
    };

    // BIAS: lane (g, li) needs bias[h][q0 + 16 t + li][kv0 + 16 kt + 4 g .. +3] and the gate of that query row
    const float* brow[QT];
    float g2[QT];
    if (BIAS) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            int qr = q0 + t * 16 + li;
            if (qr > p.sq - 1) qr = p.sq - 1;
            brow[t] = p.bias + h * p.bias_hs + (long long)qr * p.bias_rs + g * 4;
            g2[t] = p.gate[((long long)b * p.sq + qr) * p.gate_rs + h] * 1.4426950408889634f;
        }
    }
    f32x4 o_acc[QT][NDV];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDV; ++dt) o_acc[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (p.sk + KT - 1) / KT;
    const f32x4 c4 = {p.scale_log2, p.scale_log2, p.scale_log2, p.scale_log2};
    // one KV tile; PARTIAL (keys beyond sk exist) is only instantiated for the last tile so the steady-state loop
    // carries no mask compares
    auto do_tile = [&](const int tile, auto partial_c) {
        constexpr bool PARTIAL = decltype(partial_c)::value;
        __syncthreads();  // every wave finished reading the previous tile
        write_lds();
        __syncthreads();
        if (!PARTIAL) issue_loads((tile + 1) * KT);
        const int kv0 = tile * KT;

        // ---- S^T = K Q^T : s[t][kt] holds keys kv0 + 16 kt + 4 g + reg for query li
        f32x4 s[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = kt * 16 + li;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int c = ks * 4 + g;
                v8 kf = *(const v8*)(Ks + key * KROW + ((c ^ (key & (NCHP - 1))) << 3));
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][kt] = mfma16(kf, qf[t][ks], s[t][kt]);
            }
        }
        // ---- online softmax in base 2 on the RAW scores: p = exp2(s*c - m) with c = scale*log2(e) folded into one fma;
        //      keys beyond sk only exist in the last tile (wave-uniform branch)
        v8 pf[QT][2];
        // key padding mask (biased form only, wave-uniform test): 4 mask bytes per accumulator tile, the same for every query
        unsigned km[4] = {0u, 0u, 0u, 0u};
        const bool masked = BIAS && p.kmask != nullptr;
        if (masked) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                int kc = kv0 + kt * 16 + g * 4;
                if (PARTIAL && kc > p.sk - 4) kc = p.sk - 4;          // keys beyond sk are masked below anyway
                km[kt] = *(const unsigned*)(p.kmask + b * p.kmask_bs + kc);
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (BIAS) {
                // scores move to the base-2 domain here: s = s*c + gate*log2(e)*bias (keys beyond sk: address clamped,
                // value masked below)
                const f32x4 gg = {g2[t], g2[t], g2[t], g2[t]};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    int kc = kv0 + kt * 16;
                    if (PARTIAL && kc + g * 4 > p.sk - 4) kc = p.sk - 4 - g * 4;
                    const f32x4 bv = *(const f32x4*)(brow[t] + kc);
                    s[t][kt] = __builtin_elementwise_fma(s[t][kt], c4, gg * bv);
                }
                if (masked) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if ((km[kt] >> (8 * r)) & 0xffu) s[t][kt][r] = -INFINITY;
                }
            }
            if (PARTIAL) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + kt * 16 + g * 4 + r >= p.sk) s[t][kt][r] = -INFINITY;
            }
            float mx = fmaxf(s[t][0][0], s[t][0][1]);                              // v_max3_f32 chain: 8 issues / 16 values
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[t][0][2]), s[t][0][3]);
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) {
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[t][kt][0]), s[t][kt][1]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[t][kt][2]), s[t][kt][3]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[t], BIAS ? mx : mx * p.scale_log2);
            // every key so far masked: 0 is the exponent offset of THIS tile only (p = 0, no NaN from inf - inf); the running
            // maximum stays -inf, so the first tile with a finite score re-bases (alpha = 2^(-inf) = 0 on l = 0, O = 0) instead
            // of measuring its scores against a stale 0 - where valid scores below ~ -126 would all underflow
            const float m_use = (BIAS && m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_use);
            m_run[t] = m_new;
            const f32x4 nm4 = {-m_use, -m_use, -m_use, -m_use};
            f32x4 rs4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 z = BIAS ? s[t][kt] + nm4 : __builtin_elementwise_fma(s[t][kt], c4, nm4);   // v_pk_fma_f32
                f32x4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(z[r]);
                const typename VecOf<T>::v4 e16 = cvt4<T>(e);
#pragma unroll
                for (int r = 0; r < 4; ++r) pf[t][kt >> 1][(kt & 1) * 4 + r] = e16[r];
                rs4 += e;                                                       // v_pk_add_f32
            }
            const float rs = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
            l_run[t] = l_run[t] * alpha + rs;
            // the running max only moves in the first few tiles: skip the O rescale when no lane's max changed
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < NDV; ++dt) o_acc[t][dt] *= alpha;
            }
        }
        // ---- O^T += V^T P^T : A operand = V^T via transposed LDS reads
#pragma unroll
        for (int dt = 0; dt < NDV; ++dt) {
#pragma unroll
            for (int sstep = 0; sstep < 2; ++sstep) {
                // lane 4q+pp of group g supplies the address of row (key) 32 s + 4 g + q (+16), cols 16 dt + 4 pp ..
                const int qq = li >> 2, pp = li & 3;
                const T* a0 = Vs + (sstep * 32 + g * 4 + qq) * VROW + dt * 16 + pp * 4;
                v4 lo = tr_read<T>(a0);
                v4 hi = tr_read<T>(a0 + 16 * VROW);
                v8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
#pragma unroll
                for (int t = 0; t < QT; ++t) o_acc[t][dt] = mfma16(vf, pf[t][sstep], o_acc[t][dt]);
            }
        }
    };
    issue_loads(0);
    // tiles 0 .. ntiles-2 prefetch their successor; the last tile does not (it is run by the `true_type` instance,
    // which also masks; a full last tile goes through it too with every key valid)
    for (int tile = 0; tile < ntiles - 1; ++tile) do_tile(tile, std::false_type());
    do_tile(ntiles - 1, std::true_type());

    // ---- finalise: lane holds O[q = q0 + 16 t + li][16 dt + 4 g + reg]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int qr = q0 + t * 16 + li;
        if (qr >= p.sq) continue;
        T* orow = O + (long long)qr * p.o_rs;
#pragma unroll
        for (int dt = 0; dt < NDV; ++dt) {
            const int c = dt * 16 + g * 4;
            if (VEC && c + 3 < d) {
                *(v4*)(orow + c) = cvt4<T>(o_acc[t][dt] * inv);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < d) orow[c + e] = (T)(o_acc[t][dt][e] * inv);
            }
        }
    }
}

template <class T, int DK, int NDV, bool VEC, bool BIAS>
int launch_qt(const AttnArgs& a, int batch, hipStream_t st) {
    // 64 query rows per wave for head dim 64 and long sequences: K / V fragments are read from LDS half as often per MFMA
    // and each K / V tile is staged for 256 instead of 128 query rows, at 250 VGPRs - still two waves per SIMD (three at 32
    // rows): DINOv2 tower shape 603 -> 651 TFLOP/s at a 512-frame batch in isolation (2.78 -> 2.57 ms per launch; inside
    // the pipeline, behind the qkv GEMM, 2.76 -> 2.71 ms); head dim 72 needs 298 VGPRs (one wave per SIMD), loses 5 % and
    // stays at 32 rows per wave.
    if (DK == 64 && !BIAS && a.sq > 256) {
        dim3 grid((a.sq + 255) / 256, a.heads, batch);
        hipLaunchKernelGGL((attn_kernel<T, DK, NDV, 4, VEC, BIAS>), grid, dim3(256), 0, st, a);
    } else if (DK == 64 && !BIAS && a.sq > 128 && a.sq <= 192) {
        // 129 ... 192 query rows (the Q-Former's K = 144 queries against a frame's tokens): ONE workgroup per (batch, head) with 48
        // rows per wave - 144 rows are exactly three waves' worth - instead of two 128-row workgroups that stage the same K / V
        // twice, the second one for 16 rows
        dim3 grid(1, a.heads, batch);
        hipLaunchKernelGGL((attn_kernel<T, DK, NDV, 3, VEC, BIAS>), grid, dim3(256), 0, st, a);
    } else if (a.sq > 64) {
        dim3 grid((a.sq + 127) / 128, a.heads, batch);
        hipLaunchKernelGGL((attn_kernel<T, DK, NDV, 2, VEC, BIAS>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid((a.sq + 63) / 64, a.heads, batch);
        hipLaunchKernelGGL((attn_kernel<T, DK, NDV, 1, VEC, BIAS>), grid, dim3(256), 0, st, a);
    }
    return (int)hipGetLastError();
}

template <class T>
int launch(const AttnArgs& a, int batch, hipStream_t st) {
    const int d = a.d;
    if (a.bias) {     // biased scores: vector path only (BEATs: d = 64; reduced-size fixtures d = 16)
        if (!a.vec_ok) return TDC_E_BADARG;
        if (d <= 16) return launch_qt<T, 32, 1, true, true>(a, batch, st);
        if (d <= 32) return launch_qt<T, 32, 2, true, true>(a, batch, st);
        if (d <= 64) return launch_qt<T, 64, 4, true, true>(a, batch, st);
        return TDC_E_BADARG;
    }
    if (a.vec_ok) {   // head_dim % 8 == 0, 16-byte aligned rows
        if (d <= 16) return launch_qt<T, 32, 1, true, false>(a, batch, st);
        if (d <= 32) return launch_qt<T, 32, 2, true, false>(a, batch, st);
        if (d <= 64) return launch_qt<T, 64, 4, true, false>(a, batch, st);
        if (d <= 80) return launch_qt<T, 96, 5, true, false>(a, batch, st);
    } else {          // odd head dims / unaligned views (reduced-size fixtures only): element-wise loads
        if (d <= 16) return launch_qt<T, 32, 1, false, false>(a, batch, st);
        if (d <= 32) return launch_qt<T, 32, 2, false, false>(a, batch, st);
        if (d <= 64) return launch_qt<T, 64, 4, false, false>(a, batch, st);
        if (d <= 80) return launch_qt<T, 96, 5, false, false>(a, batch, st);
    }
    return TDC_E_BADARG;
}

}  // namespace

extern "C" int tdc_attention(const tdc_attn_desc* d, void* stream) {
    if (!d || !d->q || !d->k || !d->v || !d->o) return TDC_E_BADARG;
    if (d->batch <= 0 || d->heads <= 0 || d->sq <= 0 || d->sk <= 0 || d->head_dim <= 0 || d->head_dim > 80) {
        fprintf(stderr, "[tdc_hip] tdc_attention: bad shape (head_dim=%d)\n", d->head_dim);
        return TDC_E_BADARG;
    }
    if (d->heads > 65535 || d->batch > 65535) return TDC_E_BADARG;
    AttnArgs a;
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o;
    a.q_bs = d->q_bs; a.k_bs = d->k_bs; a.v_bs = d->v_bs; a.o_bs = d->o_bs;
    a.q_rs = d->q_rs; a.k_rs = d->k_rs; a.v_rs = d->v_rs; a.o_rs = d->o_rs;
    a.heads = d->heads; a.d = d->head_dim; a.sq = d->sq; a.sk = d->sk;
    a.scale_log2 = d->scale * 1.4426950408889634f;
    auto al = [](const void* p, int bytes) { return ((uintptr_t)p % bytes) == 0; };
    a.vec_ok = (d->head_dim % 8 == 0) && (d->head_dim >= 8) && (d->q_rs % 8 == 0) && (d->k_rs % 8 == 0) && (d->v_rs % 8 == 0) &&
               (d->o_rs % 4 == 0) && (d->q_bs % 8 == 0) && (d->k_bs % 8 == 0) && (d->v_bs % 8 == 0) &&
               (d->o_bs % 4 == 0) && al(d->q, 16) && al(d->k, 16) && al(d->v, 16) && al(d->o, 8);
    a.bias = d->bias; a.bias_hs = d->bias_hs; a.bias_rs = d->bias_rs; a.gate = d->gate; a.gate_rs = d->gate_rs;
    a.kmask = d->key_mask; a.kmask_bs = d->key_mask_bs;
    if (a.kmask && (!a.bias || d->key_mask_bs % 4 != 0 || d->key_mask_bs < d->sk || !al(d->key_mask, 4))) {
        fprintf(stderr, "[tdc_hip] tdc_attention: key_mask needs the biased form, key_mask_bs %% 4 == 0 (>= sk) and a 4-byte aligned base\n");
        return TDC_E_BADARG;
    }
    if (a.bias) {
        if (!a.gate || d->sk % 4 != 0 || d->sk < 4 || d->bias_rs % 4 != 0 || d->bias_hs % 4 != 0 || !al(d->bias, 16) ||
            d->gate_rs < d->heads) {
            fprintf(stderr, "[tdc_hip] tdc_attention: bias needs a gate, sk %% 4 == 0 and 16-byte aligned fp32 rows\n");
            return TDC_E_BADARG;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype != TDC_F16 && d->dtype != TDC_BF16) return TDC_E_BADARG;
    TdcProfScope prof(TDC_PROF_ATTN, st, d->batch * d->heads, d->sq, d->sk, d->head_dim, 0, 0, nullptr,
                      4.0 * d->batch * d->heads * (double)d->sq * d->sk * d->head_dim);
    if (d->form != TDC_ATTN_FORM_16X16) {   // long sequences at head dim 64 / 72 (the towers): the 32x32x16 form
        const int rc = tdc_attention32(a, d->batch, d->dtype, st);
        if (rc != -1) return rc;
    }
    if (d->dtype == TDC_F16) return launch<f16>(a, d->batch, st);
    if (d->dtype == TDC_BF16) return launch<bf16>(a, d->batch, st);
    return TDC_E_BADARG;
}
