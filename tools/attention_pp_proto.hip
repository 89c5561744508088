// PROTOTYPE, not part of the library (tools/attn_pp_diag.py builds and times it): tdc_attention in a phase-alternating form for
// the ViT towers at head dim 64 (S = 730) - the 32x32x16 flash kernel of csrc/attention32.hip rebuilt so that the matrix pipe and
// the vector ALU of a SIMD could work at the same time.  Correct (same error as the library kernel), 630-670 TFLOP/s against the
// library kernel's 685-705 on the same boxes: see profiles/r03_attention_experiments.log, experiment D, for the decomposition.
//
// Why: in attention32.hip the two waves of a SIMD belong to two independent workgroups.  Per 64-key tile a wave issues 32 MFMAs
// (1024 cycles of the matrix pipe, 256 cycles of vector issue) and ~1300 cycles of softmax VALU work, strictly one after the
// other (QK^T -> softmax -> PV is a dependency chain), and two free-running waves convoy: both want the matrix pipe, then both
// want the VALU.  Taken apart (round 2) the kernel costs the SUM of its MFMA and VALU time.
// Here a workgroup is TWO groups of four waves (waves w and w + 4 share SIMD w), each group with its own work item (256 query
// rows of one (batch, head)) and its own K / V ring, held half a tile apart by workgroup barriers:
//     group 0:   QK(0) | Y(0) | X(0) | Y(1) | X(1) | ...     X(t) = PV of tile t, then QK^T of tile t + 1 (32 MFMAs, LDS / global traffic)
//     group 1:         | QK(0) | Y(0) | X(0) | Y(1) | ...     Y(t) = softmax of tile t (VALU only)
// so that at any time one wave of a SIMD is in its MFMA phase and its partner in its VALU phase.  PV of a tile comes first in
// its phase, so the P fragments are dead before the next S accumulators are written (they share registers); V of tile t and K of
// tile t + 1 are read while tile t + 2 is written: a 3-slot ring.
// Layouts (Q^T / P^T as B operands, K rows and transposed V reads as A operands, swizzles) are those of attention32.hip.
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

// phase boundary: LDS traffic of the phase retired, then the workgroup barrier.  Global loads stay in flight across it.
__device__ __forceinline__ void phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// a copy of x the optimiser cannot see through: lane-derived LDS addresses built from it inside a phase are recomputed there
// instead of being hoisted out of the tile loop and kept (spilled) across it
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// PP_DIAG (tools/attn_pp_diag.py only, never the library build): pieces left out for timing - 1 = the softmax arithmetic,
// 2 = the MFMAs, 4 = the LDS fragment reads, 8 = the K / V staging (global loads + LDS writes).  Results are then wrong by design.
#ifndef PP_DIAG
#define PP_DIAG 0
#endif
// wave priorities of the two phases, and how many (e0, e1) pairs of the softmax run between two points at which the softmax wave
// drops to priority 0 for one instruction (0 = never)
#ifndef PP_YPRIO
#define PP_YPRIO 3
#endif
#ifndef PP_XPRIO
#define PP_XPRIO 0
#endif
#ifndef PP_YIELD
#define PP_YIELD 0
#endif

constexpr int KTP = 64;    // keys per tile

// DK = 64, two output blocks of 32 columns, two 32-row query blocks per wave (256 rows per group)
template <class T>
__global__ __launch_bounds__(512, 2) void attn_pp_kernel(AttnArgs p, int nitems) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    constexpr int DK = 64, KS = 4, NDB = 2, QB = 2;
    constexpr int KROW = 64, VROW = 64;                      // elements per LDS row (128 bytes)
    constexpr int SLOT = KTP * (KROW + VROW);                // elements per ring slot (K tile, then V tile)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gwave = wave & 3, gtid = tid & 255;
    const int r = lane & 31, hh = lane >> 5;
    T* ring = (T*)smem_raw + grp * (3 * SLOT);

    // XCD-contiguous logical workgroup ids (attention32.hip); the two groups take two consecutive work items
    const int nqb = (p.sq + 255) / 256;
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    int lid = 2 * wid + grp;
    const bool ghost = lid >= nitems;                        // odd item count: the last group recomputes the last item, stores nothing
    if (ghost) lid = nitems - 1;
    const int bh = lid / nqb, qblk = lid - bh * nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qblk * 256 + gwave * 64;
    const T* Q = (const T*)p.q + b * p.q_bs + h * DK;
    const T* K = (const T*)p.k + b * p.k_bs + h * DK;
    const T* V = (const T*)p.v + b * p.v_bs + h * DK;
    T* O = (T*)p.o + b * p.o_bs + h * DK;

    // ---- Q^T fragments (B operand): lane (r, hh) holds Q[q0 + 32 qb + r][16 ks + 8 hh .. +7]
    v8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = q0 + qb * 32 + r;
        if (qr > p.sq - 1) qr = p.sq - 1;
        const T* row = Q + (long long)qr * p.q_rs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const v8*)(row + ks * 16 + hh * 8);
    }

    // ---- staging by the group's 256 threads: K tile 64 x 8 chunks, V tile 64 x 8 chunks of 16 bytes; two of each per thread
    v8 kreg[2], vreg[2];
    auto issue_loads = [&](int kv0) __attribute__((always_inline)) {
        if (PP_DIAG & 8) return;
        const int gt = opaque(gtid);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = gt + i * 256;
            const int key = idx >> 3, c = idx & 7;
            int kr = kv0 + key; if (kr > p.sk - 1) kr = p.sk - 1;
            kreg[i] = *(const v8*)(K + (long long)kr * p.k_rs + c * 8);
            vreg[i] = *(const v8*)(V + (long long)kr * p.v_rs + c * 8);
        }
    };
    auto kswz = [](int key, int c) { return c ^ ((key >> 1) & 7); };
    auto vswz = [](int key, int c) { return c ^ (((key >> 1) & 1) << 2); };
    auto write_lds = [&](int slot) __attribute__((always_inline)) {
        if (PP_DIAG & 8) return;
        T* kd = ring + slot * SLOT;
        T* vd = kd + KTP * KROW;
        const int gt = opaque(gtid);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = gt + i * 256;
            const int key = idx >> 3, c = idx & 7;
            *(v8*)(kd + key * KROW + (kswz(key, c) << 3)) = kreg[i];
            *(v8*)(vd + key * VROW + (vswz(key, c) << 3)) = vreg[i];
        }
    };

    f32x16 o_acc[QB][NDB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o_acc[qb][db][e] = 0.f;
    }
    const int ntiles = (p.sk + KTP - 1) / KTP;
    const float c = p.scale_log2;

    f32x16 s[QB][2];
    v8 pf[QB][2][2];

    // ---- the MFMA phase: O^T += V^T P^T of the tile in slot_v (PV), then S^T = K Q^T of the tile in slot_k (QK); LDS fragments are
    // requested a batch (8 MFMAs = 256 cycles of the matrix pipe) ahead of their use, pinned in that order.
    //   K fragment of (kb, ks): row 32 kb + r, 16-byte chunk (2 ks + hh) ^ ((r >> 1) & 7) = (2 ks) ^ x with the lane constant
    //     x = hh ^ ((r >> 1) & 7): four lane addresses + immediates
    //   transposed V read of (db, kb, st): rows 32 kb + 16 st + 4 hh + qq (and + 8), 16-byte chunk (4 db + 2 (r >> 4) + (pp >> 1)) ^ y,
    //     y = 4 ((qq >> 1) & 1), 8-byte half pp & 1: two lane addresses (db) + immediates
    auto x_phase = [&](int slot_v, int slot_k, auto pv_c, auto qk_c) __attribute__((always_inline)) {
        constexpr bool PV = decltype(pv_c)::value, QK = decltype(qk_c)::value;
        const int ln = opaque(lane);
        const int rr = ln & 31, h2 = ln >> 5, li = rr & 15, qq = li >> 2, pp = li & 3;
        const int y = ((qq >> 1) & 1) << 2, x = h2 ^ ((rr >> 1) & 7);
        const char* vt_ = (const char*)(ring + slot_v * SLOT + KTP * KROW) + (h2 * 4 + qq) * (VROW * 2) + ((pp & 1) << 3);
        const char* kt_ = (const char*)(ring + slot_k * SLOT) + rr * (KROW * 2);
        v8 f[2][4];                                          // two batches of four fragments in flight
        auto load_v = [&](int db, int fb) __attribute__((always_inline)) {
            const char* va = vt_ + (((4 * db + 2 * (rr >> 4) + (pp >> 1)) ^ y) << 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // j = 2 kb + st
                if (PP_DIAG & 4) { asm volatile("" : "=v"(f[fb][j])); continue; }
                const char* a0 = va + j * 16 * (VROW * 2);
                const v4 lo = tr_read32<T>((const T*)a0);
                const v4 hi = tr_read32<T>((const T*)(a0 + 8 * VROW * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[fb][j][e] = lo[e]; f[fb][j][4 + e] = hi[e]; }
            }
        };
        auto load_k = [&](int half, int fb) __attribute__((always_inline)) {      // k-steps 2 half, 2 half + 1
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // j = 2 (ks & 1) + kb
                const int ks = 2 * half + (j >> 1), kb = j & 1;
                if (PP_DIAG & 4) { asm volatile("" : "=v"(f[fb][j])); continue; }
                f[fb][j] = *(const v8*)(kt_ + (((2 * ks) ^ x) << 4) + kb * (32 * KROW * 2));
            }
        };
        auto mma_v = [&](int db, int fb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (PP_DIAG & 2) { asm volatile("" : "+v"(o_acc[qb][db]) : "v"(f[fb][j]), "v"(pf[qb][j >> 1][j & 1])); continue; }
                    o_acc[qb][db] = mfma32(f[fb][j], pf[qb][j >> 1][j & 1], o_acc[qb][db]);
                }
        };
        auto mma_k = [&](int half, int fb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (PP_DIAG & 2) { asm volatile("" : "+v"(s[qb][j & 1]) : "v"(f[fb][j]), "v"(qf[qb][2 * half + (j >> 1)])); continue; }
                    s[qb][j & 1] = mfma32(f[fb][j], qf[qb][2 * half + (j >> 1)], s[qb][j & 1]);
                }
        };
        auto zero_s = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[qb][kb][e] = 0.f;
        };
        if (PV && QK) {
            load_v(0, 0); load_v(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_v(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_k(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_v(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_k(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            zero_s();
            mma_k(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_k(1, 1);
        } else if (PV) {
            load_v(0, 0); load_v(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_v(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_v(1, 1);
        } else {
            load_k(0, 0); load_k(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            zero_s();
            mma_k(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_k(1, 1);
        }
    };
    // online softmax in base 2 of the scores in s -> P^T fragments in pf; rescales the output accumulators
    auto softmax = [&](int kv0, auto partial_c) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial_c)::value;
        if (PP_DIAG & 1) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int st = 0; st < 2; ++st) asm volatile("" : "=v"(pf[qb][kb][st]) : "v"(s[qb][kb]));
            return;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (PARTIAL) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh >= p.sk) s[qb][kb][e] = -INFINITY;
            }
            float mx = fmaxf(s[qb][0][0], s[qb][0][1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][0][e]), s[qb][0][e + 1]);
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][1][e]), s[qb][1][e + 1]);
            mx = fmaxf(mx, other_half(mx));
            const float m_new = fmaxf(m_run[qb], mx * c);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            float rs0 = 0.f, rs1 = 0.f;             // single-value VALU only: packed f32 arithmetic does not overlap with MFMAs
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kb][st * 8 + j], c, -m_new));
                        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kb][st * 8 + j + 1], c, -m_new));
                        rs0 += e0;
                        rs1 += e1;
                        const typename VecOf<T>::v2 e16 = cvt2<T>(e0, e1);
                        pf[qb][kb][st][j] = e16[0];
                        pf[qb][kb][st][j + 1] = e16[1];
                        if (PP_YIELD && ((kb * 8 + st * 4 + (j >> 1)) % PP_YIELD) == PP_YIELD - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_setprio(0);
                            __builtin_amdgcn_s_setprio(PP_YPRIO);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            l_run[qb] = l_run[qb] * alpha + (rs0 + rs1);
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o_acc[qb][db][e] *= alpha;
            }
            __builtin_amdgcn_sched_barrier(0);     // one query block after the other: both in flight cost 64 more registers
        }
    };
    // results of a phase are pinned in it: LLVM otherwise moves pure arithmetic (the exponentials) across the barrier asm
    auto pin_p = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    u32x4 x = __builtin_bit_cast(u32x4, pf[qb][kb][st]);
                    asm volatile("" : "+v"(x));
                    pf[qb][kb][st] = __builtin_bit_cast(v8, x);
                }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(l_run[qb]), "+v"(m_run[qb]));   // the row sums too: left to sink, 64 exponentials stay live
    };
    auto pin_s = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+v"(s[qb][kb]));
    };

    // ---- prologue: tiles 0 and 1 -> slots 0 and 1, tile 2 in registers
    issue_loads(0);
    write_lds(0);
    if (ntiles > 1) { issue_loads(KTP); write_lds(1); }
    if (ntiles > 2) issue_loads(2 * KTP);
    phase_barrier();
    if (grp == 1) phase_barrier();             // group 1 runs one phase behind
    // ---- phase 0: QK^T of tile 0
    x_phase(0, 0, std::false_type(), std::true_type());
    pin_s();
    phase_barrier();
    int slot_v = 0;                             // t % 3
    for (int t = 0; t < ntiles - 1; ++t) {
        const int slot_k = slot_v == 2 ? 0 : slot_v + 1;        // (t + 1) % 3
        const int slot_w = slot_v == 0 ? 2 : slot_v - 1;        // (t + 2) % 3
        // ---- Y(t): softmax of tile t (s -> pf in place), above the partner's MFMA phase in priority (see attention32.hip)
        __builtin_amdgcn_s_setprio(PP_YPRIO);
        softmax(t * KTP, std::false_type());
        pin_p();
        __builtin_amdgcn_s_setprio(PP_XPRIO);
        phase_barrier();
        // ---- X(t): PV of tile t, then QK^T of tile t + 1; then tile t + 2 -> its slot, loads of tile t + 3
        x_phase(slot_v, slot_k, std::true_type(), std::true_type());
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntiles) {
            write_lds(slot_w);
            if (t + 3 < ntiles) issue_loads((t + 3) * KTP);
        }
        pin_s();
        phase_barrier();
        slot_v = slot_k;
    }
    // ---- the last tile: keys past the end masked
    __builtin_amdgcn_s_setprio(PP_YPRIO);
    softmax((ntiles - 1) * KTP, std::true_type());
    pin_p();
    __builtin_amdgcn_s_setprio(PP_XPRIO);
    phase_barrier();
    x_phase(slot_v, 0, std::true_type(), std::false_type());

    // ---- finalise: lane (r, hh) holds O[q = q0 + 32 qb + r][32 db + 8 (e >> 2) + 4 hh + (e & 3)]
    if (!ghost) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
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
    if (grp == 0) phase_barrier();             // matches group 1's last phase
}

template <class T>
int launch_pp(const AttnArgs& a, int batch, hipStream_t st) {
    constexpr int lds = 2 * 3 * KTP * (64 + 64) * 2;      // two groups x three slots x (K + V tile) = 96 KiB
    static bool attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_dev[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_pp_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_dev[dev] = true;
    }
    const int nitems = ((a.sq + 255) / 256) * a.heads * batch;
    hipLaunchKernelGGL((attn_pp_kernel<T>), dim3((nitems + 1) / 2), dim3(512), lds, st, a, nitems);
    return (int)hipGetLastError();
}

}  // namespace

// stand-alone entry of the prototype (same descriptor as tdc_attention; head dim 64, no bias, sq >= 256, sk >= 128)
extern "C" int tdc_attention_pp_proto(const tdc_attn_desc* d, void* stream) {
    if (!d || d->bias || d->head_dim != 64 || d->sq < 256 || d->sk < 128 || (d->q_rs | d->k_rs | d->v_rs | d->o_rs) % 8) return TDC_E_BADARG;
    AttnArgs a = {};
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o;
    a.q_bs = d->q_bs; a.k_bs = d->k_bs; a.v_bs = d->v_bs; a.o_bs = d->o_bs;
    a.q_rs = d->q_rs; a.k_rs = d->k_rs; a.v_rs = d->v_rs; a.o_rs = d->o_rs;
    a.heads = d->heads; a.d = d->head_dim; a.sq = d->sq; a.sk = d->sk;
    a.scale_log2 = d->scale * 1.4426950408889634f;
    a.vec_ok = 1;
    hipStream_t st = (hipStream_t)stream;
    return d->dtype == TDC_F16 ? launch_pp<f16>(a, d->batch, st) : launch_pp<bf16>(a, d->batch, st);
}
