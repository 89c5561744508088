// PROTOTYPE, not part of the library (tools/attn_proto_bench.py builds and times it): the tower attention at head dim 64 with the
// MFMAs of one 32-row query block issued BETWEEN the softmax instructions of the wave's other block - the only arrangement in which
// the matrix pipe and the vector ALU of a SIMD demonstrably overlap (tools/mfma_valu_overlap.cpp: across two waves they do not,
// unless the VALU wave outranks the MFMA wave; inside one wave's instruction stream an MFMA holds the issue port for 8 of its 32
// cycles).  Wave = 64 query rows = blocks A and B, software-pipelined across K / V tiles:
//     R1(t):  exp_A(t)   ||  QK^T_B(t) , PV_B(t-1)          (16 MFMAs under ~650 cycles of VALU)
//     R2(t):  exp_B(t)   ||  PV_A(t)   , QK^T_A(t+1)
// with the row maxima (head_A / head_B) between the regions and ONE barrier per tile.  K(t), V(t-1), K(t+1), V(t) are read while
// tile t+2 is written: a 4-slot ring (64 KiB per workgroup, two workgroups per CU).  Layouts as csrc/attention32.hip.
#include "../tdc-video_amd/csrc/common.h"
#include "../include/tdc_hip.h"
#include "../tdc-video_amd/csrc/attention_args.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <class T> __device__ __forceinline__ typename VecOf<T>::v4 tr_read32(const T* p) {
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(typename VecOf<T>::v4, r);
}
__device__ __forceinline__ float other_half(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((int)(threadIdx.x & 63) ^ 32) * 4, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

constexpr int KTS = 64;    // keys per tile
#ifndef SWP_VALU_PER_MFMA
#define SWP_VALU_PER_MFMA 3
#endif
#ifndef SWP_TRANS_PER_MFMA
#define SWP_TRANS_PER_MFMA 2
#endif
#ifndef SWP_AHEAD
#define SWP_AHEAD 3
#endif

template <class T>
__global__ __launch_bounds__(256, 2) void attn_swp_kernel(AttnArgs p) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    constexpr int DK = 64, KS = 4, NDB = 2;
    constexpr int KROW = 64, VROW = 64;
    constexpr int SLOT = KTS * (KROW + VROW);                // elements per ring slot: K tile then V tile (16 KiB)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* ring = (T*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = (p.sq + 255) / 256;
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int lid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int bh = lid / nqb, qblk = lid - bh * nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qblk * 256 + wave * 64;
    const T* Q = (const T*)p.q + b * p.q_bs + h * DK;
    const T* K = (const T*)p.k + b * p.k_bs + h * DK;
    const T* V = (const T*)p.v + b * p.v_bs + h * DK;
    T* O = (T*)p.o + b * p.o_bs + h * DK;

    v8 qf[2][KS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 32 + r;
        if (qr > p.sq - 1) qr = p.sq - 1;
        const T* row = Q + (long long)qr * p.q_rs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const v8*)(row + ks * 16 + hh * 8);
    }

    v8 kreg[2], vreg[2];
    auto issue_loads = [&](int kv0) __attribute__((always_inline)) {
        const int gt = opaque(tid);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = gt + i * 256;
            const int key = idx >> 3, c = idx & 7;
            int kr = kv0 + key; if (kr > p.sk - 1) kr = p.sk - 1;
            kreg[i] = *(const v8*)(K + (long long)kr * p.k_rs + c * 8);
            vreg[i] = *(const v8*)(V + (long long)kr * p.v_rs + c * 8);
        }
    };
    auto write_lds = [&](int slot) __attribute__((always_inline)) {
        T* kd = ring + slot * SLOT;
        T* vd = kd + KTS * KROW;
        const int gt = opaque(tid);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = gt + i * 256;
            const int key = idx >> 3, c = idx & 7;
            *(v8*)(kd + key * KROW + ((c ^ ((key >> 1) & 7)) << 3)) = kreg[i];
            *(v8*)(vd + key * VROW + ((c ^ (((key >> 1) & 1) << 2)) << 3)) = vreg[i];
        }
    };

    f32x16 o_acc[2][NDB];
    float m_run[2], l_run[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o_acc[qb][db][e] = 0.f;
    }
    const int ntiles = (p.sk + KTS - 1) / KTS;
    const float c = p.scale_log2;

    f32x16 sA[2], sB[2];         // S^T accumulators of the two blocks (key blocks 0 / 1)
    v8 pA[2][2], pB[2][2];       // P^T fragments [kb][st]

    // the 8 MFMAs of S^T = K Q^T for one block: K fragment (kb, ks) = row 32 kb + r, chunk (2 ks) ^ x
    auto qk = [&](int slot, const int qb, f32x16 (&s)[2]) __attribute__((always_inline)) {
        const int ln = opaque(lane);
        const int rr = ln & 31, x = (ln >> 5) ^ ((rr >> 1) & 7);
        const char* kt_ = (const char*)(ring + slot * SLOT) + rr * (KROW * 2);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const v8 kf = *(const v8*)(kt_ + (((2 * ks) ^ x) << 4) + kb * (32 * KROW * 2));
                s[kb] = mfma32(kf, qf[qb][ks], s[kb]);
            }
    };
    // the 8 MFMAs of O^T += V^T P^T for one block
    auto pv = [&](int slot, const int qb, v8 (&pf)[2][2]) __attribute__((always_inline)) {
        const int ln = opaque(lane);
        const int rr = ln & 31, h2 = ln >> 5, li = rr & 15, qq = li >> 2, pp = li & 3;
        const int y = ((qq >> 1) & 1) << 2;
        const char* vt_ = (const char*)(ring + slot * SLOT + KTS * KROW) + (h2 * 4 + qq) * (VROW * 2) + ((pp & 1) << 3);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const char* va = vt_ + (((4 * db + 2 * (rr >> 4) + (pp >> 1)) ^ y) << 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* a0 = va + j * 16 * (VROW * 2);
                const v4 lo = tr_read32<T>((const T*)a0);
                const v4 hi = tr_read32<T>((const T*)(a0 + 8 * VROW * 2));
                v8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                o_acc[qb][db] = mfma32(vf, pf[j >> 1][j & 1], o_acc[qb][db]);
            }
        }
    };
    // row maximum of a block's scores, running statistics, rescale of its output (rare after the first tiles): returns -m_new
    auto head = [&](const int qb, f32x16 (&s)[2], int kv0, auto partial_c) __attribute__((always_inline)) -> float {
        if (decltype(partial_c)::value) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh >= p.sk) s[kb][e] = -INFINITY;
        }
        float mx = fmaxf(s[0][0], s[0][1]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[0][e]), s[0][e + 1]);
#pragma unroll
        for (int e = 0; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[1][e]), s[1][e + 1]);
        mx = fmaxf(mx, other_half(mx));
        const float m_new = fmaxf(m_run[qb], mx * c);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o_acc[qb][db][e] *= alpha;
        }
        return -m_new;
    };
    // exponentials of one block: s -> P^T fragments, row sum
    auto expo = [&](const int qb, f32x16 (&s)[2], v8 (&pf)[2][2], const float nm) __attribute__((always_inline)) {
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][st * 8 + j], c, nm));
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][st * 8 + j + 1], c, nm));
                    rs0 += e0;
                    rs1 += e1;
                    const typename VecOf<T>::v2 e16 = cvt2<T>(e0, e1);
                    pf[kb][st][j] = e16[0];
                    pf[kb][st][j + 1] = e16[1];
                }
        l_run[qb] += rs0 + rs1;
    };
    // The region's MFMAs spread over its vector work, with the LDS reads of an MFMA issued AHEAD MFMAs before it: an in-order wave
    // that waits for a fragment directly in front of its MFMA also stalls the exponentials queued behind it (first version of this
    // prototype: every MFMA behind an s_waitcnt of a full LDS latency, 600 TFLOP/s).  ra / rb = LDS reads per MFMA of the first /
    // second product of the region (1 for a K fragment, 2 for a transposed V fragment), nb = MFMAs of the second product (8 or 0).
    auto spread = [&](auto ra_c, auto rb_c, auto nb_c) __attribute__((always_inline)) {
        constexpr int RA = decltype(ra_c)::value, RB = decltype(rb_c)::value, NB = decltype(nb_c)::value;
        constexpr int AHEAD = SWP_AHEAD;      // <= 8
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);
#pragma unroll
        for (int i = 0; i < 8 - AHEAD; ++i) {                  // MFMA i, reads of MFMA i + AHEAD (first product)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, SWP_VALU_PER_MFMA, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, SWP_TRANS_PER_MFMA, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {                         // reads of the second product
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, SWP_VALU_PER_MFMA, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, SWP_TRANS_PER_MFMA, 0);
        }
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {                      // the last MFMAs: everything is requested
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, SWP_VALU_PER_MFMA, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, SWP_TRANS_PER_MFMA, 0);
        }
    };
    auto pin = [&](v8 (&pf)[2][2], float& l) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4 x = __builtin_bit_cast(u32x4, pf[kb][st]);
                asm volatile("" : "+v"(x));
                pf[kb][st] = __builtin_bit_cast(v8, x);
            }
        asm volatile("" : "+v"(l));
    };

    // ---- prologue: tiles 0 and 1 -> slots 0 and 1, tile 2 in registers; S^T_A of tile 0
    issue_loads(0);
    write_lds(0);
    if (ntiles > 1) { issue_loads(KTS); write_lds(1); }
    if (ntiles > 2) issue_loads(2 * KTS);
    __syncthreads();
    qk(0, 0, sA);

    // one tile; FIRST: no PV of B of the tile before, LAST: keys past the end masked, no QK^T of A of the tile after.  The body is
    // straight-line code (a branch would split the scheduling region the MFMAs and the exponentials are interleaved in)
    auto tile = [&](const int t, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const int s0 = t & 3, s1 = (t + 1) & 3, s2 = (t + 2) & 3, sm = (t + 3) & 3;      // slots of tiles t, t+1, t+2, t-1
        if (t + 2 < ntiles) {
            write_lds(s2);
            if (t + 3 < ntiles) issue_loads((t + 3) * KTS);
        }
        const float nmA = head(0, sA, t * KTS, last_c);
        __builtin_amdgcn_sched_barrier(0);
        // ---- R1: exponentials of A beside QK^T of B (tile t) and PV of B (tile t - 1)
        expo(0, sA, pA, nmA);
        qk(s0, 1, sB);
        if (!FIRST) pv(sm, 1, pB);
        spread(std::integral_constant<int, 1>(), std::integral_constant<int, 2>(), std::integral_constant<int, FIRST ? 0 : 8>());
        pin(pA, l_run[0]);
        __builtin_amdgcn_sched_barrier(0);
        const float nmB = head(1, sB, t * KTS, last_c);
        __builtin_amdgcn_sched_barrier(0);
        // ---- R2: exponentials of B beside PV of A (tile t) and QK^T of A (tile t + 1)
        expo(1, sB, pB, nmB);
        pv(s0, 0, pA);
        if (!LAST) qk(s1, 0, sA);
        spread(std::integral_constant<int, 2>(), std::integral_constant<int, 1>(), std::integral_constant<int, LAST ? 0 : 8>());
        pin(pB, l_run[1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    tile(0, std::true_type(), std::false_type());
    for (int t = 1; t < ntiles - 1; ++t) tile(t, std::false_type(), std::false_type());
    tile(ntiles - 1, std::false_type(), std::true_type());
    // ---- PV of B of the last tile
    pv((ntiles - 1) & 3, 1, pB);

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float l = l_run[qb];
        l += other_half(l);
        const float inv = 1.0f / l;
        const int qr = q0 + qb * 32 + r;
        if (qr >= p.sq) continue;
        T* orow = O + (long long)qr * p.o_rs;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = db * 32 + g4 * 8 + hh * 4;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o_acc[qb][db][g4 * 4 + e] * inv;
                *(v4*)(orow + col) = cvt4<T>(o);
            }
    }
}

template <class T>
int launch_swp(const AttnArgs& a, int batch, hipStream_t st) {
    constexpr int lds = 4 * KTS * (64 + 64) * 2;       // 64 KiB
    static bool attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_dev[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_swp_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_dev[dev] = true;
    }
    const int nitems = ((a.sq + 255) / 256) * a.heads * batch;
    hipLaunchKernelGGL((attn_swp_kernel<T>), dim3(nitems), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int tdc_attention_proto(const tdc_attn_desc* d, void* stream) {
    if (!d || d->bias || d->head_dim != 64 || d->sq < 256 || d->sk < 192 || (d->q_rs | d->k_rs | d->v_rs | d->o_rs) % 8) return TDC_E_BADARG;
    AttnArgs a = {};
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o;
    a.q_bs = d->q_bs; a.k_bs = d->k_bs; a.v_bs = d->v_bs; a.o_bs = d->o_bs;
    a.q_rs = d->q_rs; a.k_rs = d->k_rs; a.v_rs = d->v_rs; a.o_rs = d->o_rs;
    a.heads = d->heads; a.d = d->head_dim; a.sq = d->sq; a.sk = d->sk;
    a.scale_log2 = d->scale * 1.4426950408889634f;
    a.vec_ok = 1;
    hipStream_t st = (hipStream_t)stream;
    return d->dtype == TDC_F16 ? launch_swp<f16>(a, d->batch, st) : launch_swp<bf16>(a, d->batch, st);
}
