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

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) - every index inside is a constant expression, so the
// register arrays below are never addressed at run time (a run-time index sends them to scratch)
template <int B, int E>
struct StaticFor {
    template <class F>
    static __device__ __forceinline__ void run(F&& f) {
        if constexpr (B < E) {
            f(std::integral_constant<int, B>());
            StaticFor<B + 1, E>::run(f);
        }
    }
};

constexpr int PW_KT = 64;           // keys per tile
constexpr int PW_RING = 4;          // ring slots per operand
constexpr int PW_TILE_BYTES = 8192; // 64 keys x 128 B (head dim 64)

// head dim 64: DK = 64 (4 k-steps), 2 output blocks of 32 columns; a wave = 2 query blocks of 32 rows
template <class T, int DBG = 0>
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
    // Addresses = a wave-uniform tile base (SGPR pair, advanced by 64 rows per tile) + a per-lane 32-bit offset computed once: the
    // DMA then issues in the `saddr` form and no 64-bit address arithmetic sits in the loop.  The last tile's rows >= sk are
    // clamped to row sk - 1 through a second offset set; requests past the last tile (the ring runs ahead) repeat the last tile.
    const int srow = lane >> 3, sch = lane & 7;
    const int ntiles = (p.sk + PW_KT - 1) / PW_KT;
    const int last_rows = p.sk - (ntiles - 1) * PW_KT;          // 1..64 valid rows in the last tile
    unsigned koff[2], voff[2], koff_l[2], voff_l[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = (wave * 2 + j) * 8 + srow;
        const int keyc = key < last_rows ? key : last_rows - 1;
        const unsigned kc = (unsigned)(sch ^ ((key >> 1) & 7)) * 16u, vc = (unsigned)(sch ^ (((key >> 1) & 1) << 2)) * 16u;
        koff[j] = (unsigned)key * (unsigned)p.k_rs * 2u + kc;
        voff[j] = (unsigned)key * (unsigned)p.v_rs * 2u + vc;
        koff_l[j] = (unsigned)keyc * (unsigned)p.k_rs * 2u + kc;
        voff_l[j] = (unsigned)keyc * (unsigned)p.v_rs * 2u + vc;
    }
    const long long k_tile = (long long)p.k_rs * 2 * PW_KT, v_tile = (long long)p.v_rs * 2 * PW_KT;
    auto dma_k_piece = [&](int tile, int j) {
        const int tb = tile < ntiles - 1 ? tile : ntiles - 1;
        const char* base = K + tb * k_tile;                                   // wave-uniform
        const unsigned off = tile < ntiles - 1 ? koff[j] : koff_l[j];
        char* kd = Kr + (tile & (PW_RING - 1)) * PW_TILE_BYTES + wave * 2048 + j * 1024;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(kd), 16, 0, 0);
    };
    auto dma_v_piece = [&](int tile, int j) {
        const int tb = tile < ntiles - 1 ? tile : ntiles - 1;
        const char* base = V + tb * v_tile;
        const unsigned off = tile < ntiles - 1 ? voff[j] : voff_l[j];
        char* vd = Vr + (tile & (PW_RING - 1)) * PW_TILE_BYTES + wave * 2048 + j * 1024;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(vd), 16, 0, 0);
    };
    auto dma_k = [&](int tile) { dma_k_piece(tile, 0); dma_k_piece(tile, 1); };
    auto dma_v = [&](int tile) { dma_v_piece(tile, 0); dma_v_piece(tile, 1); };

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
        if constexpr (!(DBG & 128)) { acc = mfma32(a, b, acc); return; }
        // (needed for correctness with hipcc 7.2: without this empty statement that re-defines the accumulator in front of every
        // MFMA, the first register of each O tuple loses what the loop accumulated - the O rescale's element-wise code and the asm
        // MFMAs' tied 512-bit AGPR operands do not mix; found with tools/debug_attn_pw.py, kept under test_attention_pw_form)
        asm volatile("" : "+a"(acc));
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

    // in-kernel cycle stamps of the timing build (DBG & 0x2000; tools/debug_attn_pw.py): cycles spent, summed over the
    // iterations, in the sync, slots 0-7, 8-15, 16-23, 24-31 and the tail of an iteration
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int k) {
        if constexpr (DBG & 0x2000) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (k >= 0) st_acc[k] += now - st_prev;
            st_prev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // iteration t: [barrier: K(<= t+2), V(<= t) landed] DMA K(t+4), V(t+2), then 32 slots of {one MFMA, a slice of the softmax,
    // a fragment read}: MFMAs 0-15 = S(t+1) = K(t+1) Q^T, 16-31 = O += V(t-1)^T P(t-1); VALU = P(t) = softmax(S(t)); LDS = the
    // K(t+2) / V(t) fragments, each register set re-read right behind the two MFMAs that used it.  The order is pinned
    // (sched_barrier): left to itself the scheduler issues the 32 MFMAs as one cluster in front of the whole softmax.
    auto iter = [&](int t, f32x16 (&s_cur)[QB][2], f32x16 (&s_nxt)[QB][2], v8 (&p_cur)[QB][2][2], v8 (&p_prv)[QB][2][2],
                    auto qk_c, auto pv_c, auto partial_c) {
        constexpr bool HAS_QK = decltype(qk_c)::value, HAS_PV = decltype(pv_c)::value, PARTIAL = decltype(partial_c)::value;
        stamp(-1);
        if constexpr (!(DBG & 0x800)) { PW_SYNC(4); }
        PW_PIN();
        stamp(0);
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
        float nm[QB], alpha[QB], rs0[QB], rs1[QB];
        // ---- the softmax of tile t as 106 micro-stages of 2-4 VALU instructions, dealt over the 32 MFMA slots in a fixed order.
        // With ONE wave on the SIMD nothing hides a VALU result's latency: an instruction that reads the result of the one just
        // in front of it stalls the wave (v_exp_f32 above all).  So the stream is software-pipelined over the 32 score pairs of
        // the wave's two query blocks: A(g) = the two exponent arguments of pair g, B(g) = its two exponentials, C(g) = its
        // row-sum terms + the packed conversion, issued as A(g), B(g-1), C(g-2); the running maxima (4 interleaved chains of
        // max3 per query block) come first.  Every micro-stage ends in an empty asm that makes its results opaque: pure
        // arithmetic is otherwise sunk past the pinned MFMAs to its first use.
        float pm[QB][4], z0[32], z1[32], e0[32], e1[32];
        auto max_step = [&](auto qb_c, auto step_c) {         // step 0..3 of the four 8-score chains of query block qb
            constexpr int qb = decltype(qb_c)::value, step = decltype(step_c)::value;
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const int kb = part >> 1, b0 = (part & 1) * 8;
                const f32x16& sv = s_cur[qb][kb];
                if (step == 0) pm[qb][part] = __builtin_fmaxf(__builtin_fmaxf(sv[b0], sv[b0 + 1]), sv[b0 + 2]);
                else if (step == 1) pm[qb][part] = __builtin_fmaxf(__builtin_fmaxf(pm[qb][part], sv[b0 + 3]), sv[b0 + 4]);
                else if (step == 2) pm[qb][part] = __builtin_fmaxf(__builtin_fmaxf(pm[qb][part], sv[b0 + 5]), sv[b0 + 6]);
                else pm[qb][part] = __builtin_fmaxf(pm[qb][part], sv[b0 + 7]);
            }
            asm volatile("" : "+v"(pm[qb][0]), "+v"(pm[qb][1]), "+v"(pm[qb][2]), "+v"(pm[qb][3]));
        };
        auto max_fin = [&](auto qb_c) {
            constexpr int qb = decltype(qb_c)::value;
            float lo, hi;
            both_halves(__builtin_fmaxf(__builtin_fmaxf(pm[qb][0], pm[qb][1]), __builtin_fmaxf(pm[qb][2], pm[qb][3])), lo, hi);
            const float m_new = __builtin_fmaxf(m_run[qb], __builtin_fmaxf(lo, hi) * c);
            alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            nm[qb] = -m_new;
            rs0[qb] = 0.f;
            rs1[qb] = 0.f;
            asm volatile("" : "+v"(nm[qb]), "+v"(alpha[qb]));
        };
        // pair g = 16 qb + i: scores 2 i, 2 i + 1 of the row (i = 8 kb + 4 st + j / 2: the PV product's contraction order)
        auto stage_a = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value, qb = g >> 4, i = g & 15, kb = i >> 3, st = (i >> 2) & 1, j = (i & 3) * 2;
            z0[g] = __builtin_fmaf(s_cur[qb][kb][st * 8 + j], c, nm[qb]);
            z1[g] = __builtin_fmaf(s_cur[qb][kb][st * 8 + j + 1], c, nm[qb]);
            asm volatile("" : "+v"(z0[g]), "+v"(z1[g]));
        };
        auto stage_b = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            e0[g] = __builtin_amdgcn_exp2f(z0[g]);
            e1[g] = __builtin_amdgcn_exp2f(z1[g]);
            asm volatile("" : "+v"(e0[g]), "+v"(e1[g]));
        };
        auto stage_c = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value, qb = g >> 4, i = g & 15, kb = i >> 3, st = (i >> 2) & 1, j = (i & 3) * 2;
            rs0[qb] += e0[g];
            rs1[qb] += e1[g];
            unsigned pk = __builtin_bit_cast(unsigned, cvt2<T>(e0[g], e1[g]));
            asm volatile("" : "+v"(pk), "+v"(rs0[qb]), "+v"(rs1[qb]));
            const typename VecOf<T>::v2 e16 = __builtin_bit_cast(typename VecOf<T>::v2, pk);
            p_cur[qb][kb][st][j] = e16[0];
            p_cur[qb][kb][st][j + 1] = e16[1];
            if (i == 15) l_run[qb] = __builtin_fmaf(l_run[qb], alpha[qb], rs0[qb] + rs1[qb]);
        };
        // micro-stage k of 106: 0-4 / 5-9 the maxima of query block 0 / 1 (4 chain steps + the finish), then the pair pipeline
        constexpr int NMICRO = 106;
        auto micro = [&](auto k_c) {
            constexpr int k = decltype(k_c)::value;
            if constexpr (k < 10) {
                constexpr int qb = k / 5, st = k - qb * 5;
                if constexpr (st < 4) max_step(std::integral_constant<int, qb>(), std::integral_constant<int, st>());
                else max_fin(std::integral_constant<int, qb>());
            } else {
                constexpr int pidx = k - 10;                   // 0..95: A0 | A1 B0 | (A(g) B(g-1) C(g-2)), g = 2..31 | B31 C30 | C31
                if constexpr (pidx == 0) stage_a(std::integral_constant<int, 0>());
                else if constexpr (pidx == 1) stage_a(std::integral_constant<int, 1>());
                else if constexpr (pidx == 2) stage_b(std::integral_constant<int, 0>());
                else if constexpr (pidx < 93) {
                    constexpr int q = pidx - 3, g = 2 + q / 3, w = q - (g - 2) * 3;
                    if constexpr (w == 0) stage_a(std::integral_constant<int, g>());
                    else if constexpr (w == 1) stage_b(std::integral_constant<int, g - 1>());
                    else stage_c(std::integral_constant<int, g - 2>());
                } else if constexpr (pidx == 93) stage_b(std::integral_constant<int, 31>());
                else if constexpr (pidx == 94) stage_c(std::integral_constant<int, 30>());
                else stage_c(std::integral_constant<int, 31>());
            }
        };
        // fragment f (0-7: K (kb, ks) = (f >> 2, f & 3); 8-15: V^T (db, kb, st)) is used by the MFMAs of slots 2 f and 2 f + 1 and
        // re-read for the next iteration in slot min(2 f + 3, 31): an LDS read that overwrites a register an in-flight MFMA still
        // reads as an operand waits for that MFMA (measured: re-read right behind its last use, the 24 fragment reads cost as much as
        // the whole MFMA stream).  The four LDS-DMA pieces of the iteration go out in slots 1, 9, 17, 25, beside running MFMAs.
        auto reload = [&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            if constexpr (!(DBG & 0x1000)) {
                if constexpr (f < 8) {
                    constexpr int kb = f >> 2, ks = f & 3;
                    kf[kb][ks] = *(const v8*)(kbase + k_off[kb][ks]);
                } else {
                    constexpr int u = f - 8, db = u >> 2, kb = (u >> 1) & 1, st = u & 1;
                    const v4 lo = tr_read<T>(vbase + v_off[db][kb][st][0]);
                    const v4 hi = tr_read<T>(vbase + v_off[db][kb][st][1]);
                    v8 x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
                    vf[db][kb][st] = x;
                }
            }
        };
        StaticFor<0, 32>::run([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (i < 16) {                           // S(t+1)[qb][kb] += K(t+1)[kb][ks] Q[qb][ks]: i = 8 kb + 2 ks + qb
                constexpr int kb = i >> 3, ks = (i >> 1) & 3, qb = i & 1;
                if constexpr (HAS_QK && !(DBG & 0x400)) {
                    if constexpr (ks == 0) mfma_s0(s_nxt[qb][kb], kf[kb][ks], qf[qb][ks]);
                    else mfma_s(s_nxt[qb][kb], kf[kb][ks], qf[qb][ks]);
                }
            } else {                                          // O[qb][db] += V(t-1)[db][kb][st] P(t-1)[qb][kb][st]: i - 16 = 8 db + 4 kb + 2 st + qb
                constexpr int u = i - 16, db = u >> 3, kb = (u >> 2) & 1, st = (u >> 1) & 1, qb = u & 1;
                if constexpr (HAS_PV && !(DBG & 0x400)) mfma_o(o_acc[qb][db], vf[db][kb][st], p_prv[qb][kb][st]);
            }
            PW_PIN();
            if constexpr (!(DBG & 0x200)) StaticFor<(i * NMICRO) / 32, ((i + 1) * NMICRO) / 32>::run(micro);
            if constexpr (i >= 3 && i < 31 && ((i - 3) & 1) == 0) reload(std::integral_constant<int, (i - 3) / 2>());
            if constexpr (i == 31) { reload(std::integral_constant<int, 14>()); reload(std::integral_constant<int, 15>()); }
            if constexpr (!(DBG & 0x100)) {
                if constexpr (i == 1) dma_k_piece(t + 4, 0);
                if constexpr (i == 9) dma_k_piece(t + 4, 1);
                if constexpr (i == 17) dma_v_piece(t + 2, 0);
                if constexpr (i == 25) dma_v_piece(t + 2, 1);
            }
            PW_PIN();
            if constexpr (i == 7) stamp(1);
            if constexpr (i == 15) stamp(2);
            if constexpr (i == 23) stamp(3);
            if constexpr (i == 31) stamp(4);
        });
        // the older sums follow the new maximum: O (complete up to tile t-1) scales by alpha - rarely, the maximum settles within
        // the first tiles.  O sits in AGPRs behind asm MFMAs: pad the read by hand.
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
            if (!(DBG & (1 | 0x200)) && ((DBG & 8) || !__all(alpha[qb] == 1.0f))) {
                // (the operands keep every read of O below the pad: hoisted above it they would sit right behind an asm MFMA)
                asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o_acc[qb][0]), "+a"(o_acc[qb][1]));
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o_acc[qb][db][e] *= alpha[qb];
                asm volatile("s_nop 7" : "+a"(o_acc[qb][0]), "+a"(o_acc[qb][1]));      // v_accvgpr_write -> MFMA SrcC
            }
        stamp(5);
    };
    typedef std::true_type yes;
    typedef std::false_type no;
    (void)0;
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

    if constexpr (DBG & 0x2000) {
        if (blockIdx.x == 0 && tid == 0) {
            unsigned long long* dst = (unsigned long long*)p.o;      // (timing build: the output is not valid anyway)
#pragma unroll
            for (int k = 0; k < 6; ++k) dst[k] = st_acc[k];
            dst[6] = (unsigned long long)ntiles;
        }
        return;
    }
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

template <class T, int DBG = 0>
int launch_pw64(const AttnArgs& a, int batch, hipStream_t st) {
    constexpr int lds = 2 * PW_RING * PW_TILE_BYTES;
    static bool attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_dev[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_pw64_kernel<T, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_dev[dev] = true;
    }
    dim3 grid(((a.sq + 255) / 256) * a.heads * batch);
    hipLaunchKernelGGL((attn_pw64_kernel<T, DBG>), grid, dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

// entry for attention.hip: -1 when this form does not apply
int tdc_attention_pw(const AttnArgs& a, int batch, int dtype, hipStream_t st, int dbg) {
    if (a.bias || !a.vec_ok || a.sq < 256 || a.sk < 3 * PW_KT || a.d != 64) return -1;
    if ((a.k_rs & 7) || (a.v_rs & 7)) return -1;
    // timing ablations (tools/debug_attn_pw.py; results invalid): form = 2 + bits
    if (dbg == 0x100) return launch_pw64<f16, 0x100>(a, batch, st);
    if (dbg == 0x200) return launch_pw64<f16, 0x200>(a, batch, st);
    if (dbg == 0x400) return launch_pw64<f16, 0x400>(a, batch, st);
    if (dbg == 0x800) return launch_pw64<f16, 0x900>(a, batch, st);       // no sync needs no DMA either
    if (dbg == 0x1000) return launch_pw64<f16, 0x1000>(a, batch, st);
    if (dbg == 0x1100) return launch_pw64<f16, 0x1b00>(a, batch, st);     // MFMAs + softmax only
    if (dbg == 0x1300) return launch_pw64<f16, 0x1b00 | 0x200>(a, batch, st);   // MFMAs only
    if (dbg == 0x2000) return launch_pw64<f16, 0x2000>(a, batch, st);           // full kernel with cycle stamps
    if (dbg == 0x2100) return launch_pw64<f16, 0x2100>(a, batch, st);           // ... without DMA
    if (dbg == 0x3000) return launch_pw64<f16, 0x3000>(a, batch, st);           // ... without fragment reads
    return dtype == TDC_F16 ? launch_pw64<f16>(a, batch, st) : launch_pw64<bf16>(a, batch, st);
}
