// tdc_attention, tower form "pw": ONE wave per SIMD with the whole 512-entry register file, software-pipelined across K/V
// tiles inside the wave (cdna_hip_programming.md, "Fused attention prefill", the 4-wave structure; MI355X_MICROARCH.md, "one wave
// per SIMD: single-issue instructions HIDDEN per MFMA gap").
//
// Why: the two-waves-per-SIMD kernel (attention32.hip) spends a 64-key tile as QK^T (16 MFMAs) -> softmax (~280 VALU
// instructions) -> PV (16 MFMAs), one after the other per wave, and its two waves per SIMD do not overlap them either (the
// counters: matrix pipe ~30 % busy, VALU ~70 %, together ~100 %).  Here a wave owns 64 query rows (two 32-row blocks) and runs
// three tiles at once: while the VALU works through the softmax of tile t, the matrix pipe runs QK^T of tile t+1 and PV of tile
// t-1 - independent instruction streams of ONE wave, so an MFMA's 32 cycles are filled by the wave's own exponentials.  All
// MFMA operands of a tile sit in registers a full tile before they are used (K fragments of tile t+2 and V^T fragments of
// tile t are read from LDS during iteration t), so no MFMA waits on an LDS read issued just ahead of it.
//
// K / V tiles (64 keys) come by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip) into 4-slot rings, issued two
// iterations ahead, one raw barrier per tile behind a counted vmcnt; the XOR swizzles of the fragment reads are applied on the
// DMA's per-lane SOURCE address (the LDS image is lane-linear).
#include "common.h"
#include "../../include/tdc_hip.h"
#include "attention_args.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// ds_read_b64_tr_b16 as an asm statement: the builtin form makes the compiler wait vmcnt(0) in front of it whenever an LDS-DMA
// piece is in flight (it cannot tell the DMA's LDS write from the bytes being read), which drains the K / V stream once per
// tile.  The compiler does not count this read either: every fragment read this way is consumed one iteration later, behind
// that iteration's s_waitcnt lgkmcnt(0) (PW_SYNC), and the audit of the .s (no copy of the destination between the read and
// that wait) is part of the build notes in DESIGN.md.
template <class T> __device__ __forceinline__ typename VecOf<T>::v4 tr_read(unsigned lds_byte_addr) {
    typename VecOf<T>::v4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr));
    return r;
}
// value held by lane ^ 32 (v_permlane32_swap: a VALU exchange between the two lane halves, no LDS round trip)
// both lane halves' values of v (v_permlane32_swap: a VALU exchange between lanes l and l ^ 32, no LDS round trip): with both
// operands = v, result 0 holds the LOW half's value in every lane and result 1 the HIGH half's
__device__ __forceinline__ void both_halves(float v, float& lo, float& hi) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    lo = __builtin_bit_cast(float, (unsigned)sw[0]);
    hi = __builtin_bit_cast(float, (unsigned)sw[1]);
}

constexpr int PW_KT = 64;           // keys per tile
constexpr int PW_RING = 4;          // ring slots per operand
constexpr int PW_TILE_BYTES = 8192; // 64 keys x 128 B (head dim 64)

// head dim 64: DK = 64 (4 k-steps), 2 output blocks of 32 columns; a wave = 2 query blocks of 32 rows
template <class T>
__global__ __launch_bounds__(256, 1) void attn_pw64_kernel(AttnArgs p) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    constexpr int KS = 4, NDB = 2, QB = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kr = smem;                                   // [PW_RING][64 keys][128 B]
    char* Vr = smem + PW_RING * PW_TILE_BYTES;         // [PW_RING][64 keys][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // XCD-contiguous logical ids (the query blocks of one head share an L2), as attention32.hip
    const int nqb = (p.sq + 255) >> 8;
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int lid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int bh = lid / nqb, qblk = lid - bh * nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qblk * 256 + wave * 64;
    const T* Q = (const T*)p.q + b * p.q_bs + h * 64;
    const char* K = (const char*)((const T*)p.k + b * p.k_bs + h * 64);
    const char* V = (const char*)((const T*)p.v + b * p.v_bs + h * 64);
    T* O = (T*)p.o + b * p.o_bs + h * 64;

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
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), compiler-visible: no ordinary load is pending once the DMA stream starts

    // ---- LDS-DMA staging: per tile and operand 8 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces 2 w, 2 w + 1.
    //      lane -> row 8 pi + (lane >> 3), physical 16-B chunk lane & 7; the source chunk is the swizzle's inverse image:
    //      K: chunk ^ ((key >> 1) & 7)   (ds_read_b128 of 16-lane groups conflict-free), V: chunk ^ (((key >> 1) & 1) << 2)
    //      (the four keys of a transposed read fall on four 64-byte bank groups) - the images attention32.hip reads.
    const int srow = lane >> 3, sch = lane & 7;
    int krow_[2], kcb[2], vcb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = (wave * 2 + j) * 8 + srow;
        krow_[j] = key;
        kcb[j] = (sch ^ ((key >> 1) & 7)) * 16;
        vcb[j] = (sch ^ (((key >> 1) & 1) << 2)) * 16;
    }
    const long long k_rsb = (long long)p.k_rs * 2, v_rsb = (long long)p.v_rs * 2;
    auto dma_k = [&](int tile) {
        char* kd = Kr + (tile & (PW_RING - 1)) * PW_TILE_BYTES + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int kr = tile * PW_KT + krow_[j];
            if (kr > p.sk - 1) kr = p.sk - 1;
            __builtin_amdgcn_global_load_lds(GLB_PTR(K + kr * k_rsb + kcb[j]), LDS_PTR(kd + j * 1024), 16, 0, 0);
        }
    };
    auto dma_v = [&](int tile) {
        char* vd = Vr + (tile & (PW_RING - 1)) * PW_TILE_BYTES + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int kr = tile * PW_KT + krow_[j];
            if (kr > p.sk - 1) kr = p.sk - 1;
            __builtin_amdgcn_global_load_lds(GLB_PTR(V + kr * v_rsb + vcb[j]), LDS_PTR(vd + j * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a tile image)
    int k_off[2][KS];         // A operand of QK^T: key 32 kb + r, logical chunk 2 ks + hh
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kb * 32 + r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_off[kb][ks] = key * 128 + (((ks * 2 + hh) ^ ((key >> 1) & 7)) << 4);
    }
    // A operand of PV (V^T through the transposed read): 16-lane group (r >> 4) reads the 4-key x 16-column block at keys
    // 32 kb + 16 st + 4 hh (+ 8), columns 32 db + 16 (r >> 4): lane 4 q + pp supplies row q, columns 4 pp ..
    int v_off[NDB][2][2][2];
    {
        const int li = r & 15, qq = li >> 2, pp = li & 3;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int key = kb * 32 + st * 16 + hh * 4 + qq + u * 8;
                        const int col = db * 32 + (r >> 4) * 16 + pp * 4;
                        v_off[db][kb][st][u] = key * 128 + ((((col >> 3) ^ (((key >> 1) & 1) << 2))) << 4) + (col & 4) * 2;
                    }
    }

    // ---- MFMAs as asm statements: the register FILE of every operand is chosen here - the compiler's own choice parks the S
    // accumulators in AGPRs and copies them out for the softmax (136 v_accvgpr_read + 72 v_accvgpr_write per tile).  S (read by
    // the VALU) and P (written by it) live in VGPRs, O and the K / V / Q fragments in AGPRs ("a": DS loads can target them).
    // An MFMA's result is never read in the same iteration: S(t+1) is consumed by iteration t+1's softmax, O by the next
    // iteration's MFMAs (an accumulate chain needs no wait states) - the two places that read O with the VALU pad themselves.
#define PW_MFMA_NAME(T) (std::is_same<T, f16>::value ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_32x32x16_bf16")
    auto mfma_s0 = [&](f32x16& acc, const v8& a, const v8& b) {      // acc = a b   (S, VGPRs)
        if constexpr (std::is_same<T, f16>::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "a"(b));
    };
    auto mfma_s = [&](f32x16& acc, const v8& a, const v8& b) {       // acc += a b
        if constexpr (std::is_same<T, f16>::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b));
    };
    auto mfma_o = [&](f32x16& acc, const v8& a, const v8& b) {       // acc += a b   (O in AGPRs, P from VGPRs)
        if constexpr (std::is_same<T, f16>::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
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
    const int ntiles = (p.sk + PW_KT - 1) / PW_KT;
    const float c = p.scale_log2;

#define PW_SYNC(N)                                                       \
    asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                        \
    __builtin_amdgcn_sched_barrier(0)
#define PW_PIN() __builtin_amdgcn_sched_barrier(0)

    // ---- prologue.  Ring discipline: iteration t requests K(t+4) and V(t+2) - two K and two V pieces per wave - into the slots
    // whose tiles (K(t), V(t-2)) every wave finished reading before it reached iteration t's barrier; "all but my last four
    // pieces have landed" + that barrier then means K(<= t+2) and V(<= t) are in LDS for everyone.  The prologue requests
    // K(0..3), V(0..1) in that order of need and starts from the same state.
    dma_k(0); dma_k(1); dma_v(0); dma_k(2); dma_k(3); dma_v(1);
    PW_SYNC(4);
    v8 kf[2][KS], vf[NDB][2][2];
    f32x16 sA[QB][2], sB[QB][2];
    v8 pA[QB][2][2], pB[QB][2][2];
    {
        const char* kb0 = Kr;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const v8*)(kb0 + k_off[kb][ks]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (ks == 0) mfma_s0(sA[qb][kb], kf[kb][ks], qf[qb][ks]);
                    else mfma_s(sA[qb][kb], kf[kb][ks], qf[qb][ks]);
                }
        const char* kb1 = Kr + PW_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const v8*)(kb1 + k_off[kb][ks]);
        // S(0) is read by the VALU right below: no compiler padding behind asm MFMAs, and the operands keep the reads below it
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sA[0][0]), "+v"(sA[0][1]), "+v"(sA[1][0]), "+v"(sA[1][1]));
    }

    // iteration t: [barrier: K(<= t+2), V(<= t) landed] DMA K(t+4), V(t+2), then 32 slots of {one MFMA, a slice of the softmax,
    // a fragment read}: MFMAs 0-15 = S(t+1) = K(t+1) Q^T, 16-31 = O += V(t-1)^T P(t-1); VALU = P(t) = softmax(S(t)); LDS = the
    // K(t+2) / V(t) fragments, each register set re-read right behind the two MFMAs that used it.  The order is pinned
    // (sched_barrier): left to itself the scheduler issues the 32 MFMAs as one cluster in front of the whole softmax.
    auto iter = [&](int t, f32x16 (&s_cur)[QB][2], f32x16 (&s_nxt)[QB][2], v8 (&p_cur)[QB][2][2], v8 (&p_prv)[QB][2][2],
                    auto qk_c, auto pv_c, auto partial_c) {
        constexpr bool HAS_QK = decltype(qk_c)::value, HAS_PV = decltype(pv_c)::value, PARTIAL = decltype(partial_c)::value;
        PW_SYNC(4);
        dma_k(t + 4);
        dma_v(t + 2);
        PW_PIN();
        const char* kbase = Kr + ((t + 2) & (PW_RING - 1)) * PW_TILE_BYTES;
        const unsigned vbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vr + (t & (PW_RING - 1)) * PW_TILE_BYTES;
        const int kv0 = t * PW_KT;
        if (PARTIAL) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh >= p.sk) s_cur[qb][kb][e] = -INFINITY;
        }
        float mx[QB], nm[QB], alpha[QB], rs0[QB], rs1[QB];
        // softmax slices.  max_part(qb, part): running maximum over 8 of the row's 32 scores; max_fin(qb): both lane halves, the
        // new running maximum, the scale of the older sums; pair(qb, i): scores 2 i, 2 i + 1 of the row -> two exponentials,
        // their row-sum terms and one packed conversion (i = 16 kb + 8 st + j / 2 in the PV product's contraction order)
        auto max_part = [&](int qb, int part) {
            const int kb = part >> 1, e0 = (part & 1) * 8;
            float m = __builtin_fmaxf(__builtin_fmaxf(s_cur[qb][kb][e0], s_cur[qb][kb][e0 + 1]), s_cur[qb][kb][e0 + 2]);
            m = __builtin_fmaxf(__builtin_fmaxf(m, s_cur[qb][kb][e0 + 3]), s_cur[qb][kb][e0 + 4]);
            m = __builtin_fmaxf(__builtin_fmaxf(m, s_cur[qb][kb][e0 + 5]), s_cur[qb][kb][e0 + 6]);
            m = __builtin_fmaxf(m, s_cur[qb][kb][e0 + 7]);
            mx[qb] = part == 0 ? m : __builtin_fmaxf(mx[qb], m);
            asm volatile("" : "+v"(mx[qb]));
        };
        auto max_fin = [&](int qb) {
            float lo, hi;
            both_halves(mx[qb], lo, hi);
            const float m_new = __builtin_fmaxf(m_run[qb], __builtin_fmaxf(lo, hi) * c);
            alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            nm[qb] = -m_new;
            rs0[qb] = 0.f;
            rs1[qb] = 0.f;
            asm volatile("" : "+v"(nm[qb]), "+v"(alpha[qb]));
        };
        auto pair = [&](int qb, int i) {
            const int kb = i >> 3, st = (i >> 2) & 1, j = (i & 3) * 2;
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[qb][kb][st * 8 + j], c, nm[qb]));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[qb][kb][st * 8 + j + 1], c, nm[qb]));
            rs0[qb] += e0;
            rs1[qb] += e1;
            unsigned pk = __builtin_bit_cast(unsigned, cvt2<T>(e0, e1));
            // the slot's results made opaque HERE: pure arithmetic is otherwise sunk past the pinned MFMAs to its first use
            asm volatile("" : "+v"(pk), "+v"(rs0[qb]), "+v"(rs1[qb]));
            const typename VecOf<T>::v2 e16 = __builtin_bit_cast(typename VecOf<T>::v2, pk);
            p_cur[qb][kb][st][j] = e16[0];
            p_cur[qb][kb][st][j + 1] = e16[1];
        };
        auto row_fin = [&](int qb) { l_run[qb] = __builtin_fmaf(l_run[qb], alpha[qb], rs0[qb] + rs1[qb]); };
        // the VALU work of slot i (0..31): slots 0-1 the maximum of query block 0, slots 2-3 that of block 1 beside the first pairs
        // of block 0, then one pair per slot: block 0's 16 pairs in slots 2-17, block 1's in slots 16-31
        auto valu_slot = [&](int i) {
            if (i == 0) { max_part(0, 0); max_part(0, 1); }
            if (i == 1) { max_part(0, 2); max_part(0, 3); max_fin(0); }
            if (i == 2) { max_part(1, 0); max_part(1, 1); }
            if (i == 3) { max_part(1, 2); max_part(1, 3); max_fin(1); }
            if (i >= 2 && i < 18) pair(0, i - 2);
            if (i == 17) row_fin(0);
            if (i >= 16) pair(1, i - 16);
            if (i == 31) row_fin(1);
        };
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i < 16) {                                     // S(t+1)[qb][kb] += K(t+1)[kb][ks] Q[qb][ks]: i = 8 kb + 2 ks + qb
                const int kb = i >> 3, ks = (i >> 1) & 3, qb = i & 1;
                if (HAS_QK) {
                    if (ks == 0) mfma_s0(s_nxt[qb][kb], kf[kb][ks], qf[qb][ks]);
                    else mfma_s(s_nxt[qb][kb], kf[kb][ks], qf[qb][ks]);
                }
                PW_PIN();
                valu_slot(i);
                if (qb == 1) kf[kb][ks] = *(const v8*)(kbase + k_off[kb][ks]);      // K(t+2) fragment behind its last use
            } else {                                          // O[qb][db] += V(t-1)[db][kb][st] P(t-1)[qb][kb][st]: i - 16 = 8 db + 4 kb + 2 st + qb
                const int u = i - 16, db = u >> 3, kb = (u >> 2) & 1, st = (u >> 1) & 1, qb = u & 1;
                if (HAS_PV) mfma_o(o_acc[qb][db], vf[db][kb][st], p_prv[qb][kb][st]);
                PW_PIN();
                valu_slot(i);
                if (qb == 1) {                                // V(t) fragment behind its last use
                    const v4 lo = tr_read<T>(vbase + v_off[db][kb][st][0]);
                    const v4 hi = tr_read<T>(vbase + v_off[db][kb][st][1]);
                    v8 x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
                    vf[db][kb][st] = x;
                }
            }
            PW_PIN();
        }
        // the older sums follow the new maximum: O (complete up to tile t-1) scales by alpha - rarely, the maximum settles within
        // the first tiles.  O sits in AGPRs behind asm MFMAs: pad the read by hand.
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
            if (!__all(alpha[qb] == 1.0f)) {
                // (the operands keep every read of O below the pad: hoisted above it they would sit right behind an asm MFMA)
                asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o_acc[qb][0]), "+a"(o_acc[qb][1]));
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o_acc[qb][db][e] *= alpha[qb];
                asm volatile("s_nop 7" : "+a"(o_acc[qb][0]), "+a"(o_acc[qb][1]));      // v_accvgpr_write -> MFMA SrcC
            }
    };
    typedef std::true_type yes;
    typedef std::false_type no;
    // tile 0 has no PV yet; the last tile (partial) no further QK^T; ntiles >= 3
    iter(0, sA, sB, pA, pB, yes(), no(), no());
    int t = 1;
    for (; t + 2 < ntiles; t += 2) {
        iter(t, sB, sA, pB, pA, yes(), yes(), no());
        iter(t + 1, sA, sB, pA, pB, yes(), yes(), no());
    }
    // after the loop t is odd, S(t) in sB; one or two tiles are left
    if (t + 1 < ntiles) {
        iter(t, sB, sA, pB, pA, yes(), yes(), no());
        iter(t + 1, sA, sB, pA, pB, no(), yes(), yes());
        // O += V(last)^T P(last): the V fragments were read by asm statements the compiler does not count
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PW_PIN();
#pragma unroll
        for (int u = 0; u < 16; ++u) mfma_o(o_acc[u & 1][u >> 3], vf[u >> 3][(u >> 2) & 1][(u >> 1) & 1], pA[u & 1][(u >> 2) & 1][(u >> 1) & 1]);
    } else {
        iter(t, sB, sA, pB, pA, no(), yes(), yes());
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PW_PIN();
#pragma unroll
        for (int u = 0; u < 16; ++u) mfma_o(o_acc[u & 1][u >> 3], vf[u >> 3][(u >> 2) & 1][(u >> 1) & 1], pB[u & 1][(u >> 2) & 1][(u >> 1) & 1]);
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o_acc[0][0]), "+a"(o_acc[0][1]), "+a"(o_acc[1][0]), "+a"(o_acc[1][1]));   // O is read by the VALU below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA pieces requested past the last tile

    // ---- finalise: lane (r, hh) holds O[q = q0 + 32 qb + r][32 db + 8 (e >> 2) + 4 hh + (e & 3)]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float llo, lhi;
        both_halves(l_run[qb], llo, lhi);
        const float inv = 1.0f / (llo + lhi);
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
#undef PW_SYNC
#undef PW_PIN
#undef PW_MFMA_NAME
}

template <class T>
int launch_pw64(const AttnArgs& a, int batch, hipStream_t st) {
    constexpr int lds = 2 * PW_RING * PW_TILE_BYTES;
    static bool attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_dev[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_pw64_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_dev[dev] = true;
    }
    dim3 grid(((a.sq + 255) / 256) * a.heads * batch);
    hipLaunchKernelGGL((attn_pw64_kernel<T>), grid, dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

// entry for attention.hip: -1 when this form does not apply
int tdc_attention_pw(const AttnArgs& a, int batch, int dtype, hipStream_t st) {
    if (a.bias || !a.vec_ok || a.sq < 256 || a.sk < 3 * PW_KT || a.d != 64) return -1;
    if ((a.k_rs & 7) || (a.v_rs & 7)) return -1;
    return dtype == TDC_F16 ? launch_pw64<f16>(a, batch, st) : launch_pw64<bf16>(a, batch, st);
}
