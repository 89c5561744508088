// tdc_attention, tower form "pw": ONE wave per SIMD with the whole 512-entry register file, software-pipelined across K/V
// tiles inside the wave (cdna_hip_programming.md, "Fused attention prefill", the 4-wave structure; MI355X_MICROARCH.md, "one wave
// per SIMD: single-issue instructions HIDDEN per MFMA gap").  Head dim 64 (DINOv2-giant: 24 heads x 64).
//
// Why: the two-waves-per-SIMD kernel (attention32.hip) spends a 64-key tile as QK^T (16 MFMAs) -> softmax (~280 VALU
// instructions) -> PV (16 MFMAs), one after the other per wave, and its two waves per SIMD do not overlap them either (the
// counters: matrix pipe ~30 % busy, VALU ~70 %, together ~100 %).  Here a wave owns 64 query rows (two 32-row blocks) and runs
// three tiles at once: while the VALU works through the softmax of tile t, the matrix pipe runs QK^T of tile t+1 and PV of tile
// t-1 - independent instruction streams of ONE wave, so an MFMA's 32 cycles are filled by the wave's own exponentials.
//
// STATUS: a PROTOTYPE outside the product library since round 5 (it was form 2 of tdc_attention in round 4).  Correct
// (tools/debug_attn_pw.py checks it against the library's kernel and fp32 SDPA) and SLOWER than attention32.hip on the tower shape
// (B = 512, 24 heads, S = 730: 3.0-3.4 ms against 2.45-2.6 ms; S = 4096: 1.14-1.30 ms against 1.05-1.10 ms).  Built by
// tools/attention_pw/build.sh into its own libtdc_attn_pw.so (entry: tdc_attn_pw_run, below), with tools/audit_asm_reads.py
// run on the generated code as part of that build.  What was measured on the way (profiles/archive/NOTES_rounds1-4.md 9.3):
//  * MFMA stream + softmax alone - K / V fragments and tiles frozen - runs at 1 600 TFLOP/s equivalent: the exponentials DO hide
//    under the wave's own MFMAs; MFMA stream + LDS-DMA + fragment reads without the softmax at ~1 300; all three together at
//    600-720: with one wave per SIMD every LDS-DMA issue (50-250 cycles each, four per tile), every barrier skew and the whole
//    per-item prologue (Q, first tiles: ~35 % of an item of 12 tiles) are exposed, and nothing on the SIMD covers them;
//  * structure kept from those measurements: the softmax as a software pipeline over the 32 score pairs (dependent VALU
//    instructions are never neighbours), the maximum of tile t+1 behind tile t's pairs, the row sums on the matrix pipe (a third
//    "output block" whose V^T operand is all ones), P in ONE buffer (a chunk's three MFMAs two slots behind the conversion that
//    completes it), fragment reads with per-lane bases + immediate offsets two slots behind the last MFMA that used the
//    register set, LDS-DMA by a uniform SGPR base + per-lane 32-bit offsets, deferred rescale (T13: the O rescale is an AGPR
//    round trip of ~1000 cycles, taken in most tiles without the threshold);
//  * S MFMAs are asm statements (VGPR destination, AGPR operands: the compiler's own choice parks S in AGPRs and copies it out,
//    136 v_accvgpr_read + 72 v_accvgpr_write per tile); the PV / row-sum MFMAs are builtins - as asm statements with tied
//    512-bit AGPR accumulators next to the element-wise rescale they were miscompiled (first / last register of each tuple
//    lost what the loop had accumulated), in two different ways;
//  * the transposed V reads are asm statements: the builtin makes the compiler wait vmcnt(0) in front of it whenever an LDS-DMA
//    piece is in flight.  tools/audit_asm_reads.py checks the generated code for what the compiler does not do for an asm
//    statement (count its LDS read, pad its MFMA's destination); an asm read whose result is unused must not be issued at all
//    (the compiler gives its destination to something else and the data still lands there).
//
// K / V tiles (64 keys) come by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip) into 4-slot rings, requested two
// iterations ahead, one raw barrier per tile behind a counted vmcnt; the XOR swizzles of the fragment reads are applied on the
// DMA's per-lane SOURCE address (the LDS image is lane-linear).
#include "common.h"            // -I tdc-video_amd/csrc
#include "tdc_hip.h"           // -I include
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
// ds_read_b64_tr_b16 as an asm statement with an immediate offset (see the header).  The compiler does not count this read:
// every fragment read this way is consumed one iteration later, behind that iteration's s_waitcnt lgkmcnt(0) (PW_SYNC).
template <class T, int OFF> __device__ __forceinline__ typename VecOf<T>::v4 tr_read(unsigned lds_byte_addr) {
    typename VecOf<T>::v4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "i"(OFF));
    return r;
}
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

constexpr float PW_THR = 8.0f;      // deferred-rescale threshold, in binades
constexpr int PW_KT = 64;           // keys per tile
constexpr int PW_RING = 4;          // ring slots per operand
constexpr int PW_TILE_BYTES = 8192; // 64 keys x 128 B (head dim 64)
constexpr int PW_NSLOT = 40;        // MFMAs per tile and wave: 16 QK^T + 16 PV + 8 row sums

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
    // Addresses = a wave-uniform tile base (SGPR pair) + a per-lane 32-bit offset computed once (the `saddr` form).  The last
    // tile's rows >= sk are clamped to row sk - 1 through a second offset set; requests past the last tile (the ring runs ahead)
    // repeat the last tile.
    const int ntiles = (p.sk + PW_KT - 1) / PW_KT;
    const int last_rows = p.sk - (ntiles - 1) * PW_KT;          // 1..64 valid rows in the last tile
    unsigned koff[2], voff[2];
    {
        const int srow = lane >> 3, sch = lane & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int key = (wave * 2 + j) * 8 + srow;
            koff[j] = (unsigned)key * (unsigned)p.k_rs * 2u + (unsigned)(sch ^ ((key >> 1) & 7)) * 16u;
            voff[j] = (unsigned)key * (unsigned)p.v_rs * 2u + (unsigned)(sch ^ (((key >> 1) & 1) << 2)) * 16u;
        }
    }
    const long long k_tile = (long long)p.k_rs * 2 * PW_KT, v_tile = (long long)p.v_rs * 2 * PW_KT;
    // rows of the last tile past the sequence end: how far lane's row lies beyond the last valid one (0 for valid rows),
    // recomputed from the lane id where it is needed - twice per item - instead of living in registers through the loop
    auto rows_beyond = [&](int j) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int key = (wave * 2 + j) * 8 + (l >> 3);
        return key < last_rows ? 0 : key - (last_rows - 1);
    };
    auto dma_k_piece = [&](int tile, int j) {
        const int tb = tile < ntiles - 1 ? tile : ntiles - 1;
        const char* base = K + tb * k_tile;                                   // wave-uniform
        unsigned off = koff[j];
        if (tile >= ntiles - 1) off -= (unsigned)rows_beyond(j) * (unsigned)p.k_rs * 2u;      // wave-uniform branch
        char* kd = Kr + (tile & (PW_RING - 1)) * PW_TILE_BYTES + wave * 2048 + j * 1024;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(kd), 16, 0, 0);
    };
    auto dma_v_piece = [&](int tile, int j) {
        const int tb = tile < ntiles - 1 ? tile : ntiles - 1;
        const char* base = V + tb * v_tile;
        unsigned off = voff[j];
        if (tile >= ntiles - 1) off -= (unsigned)rows_beyond(j) * (unsigned)p.v_rs * 2u;
        char* vd = Vr + (tile & (PW_RING - 1)) * PW_TILE_BYTES + wave * 2048 + j * 1024;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(vd), 16, 0, 0);
    };
    auto dma_k = [&](int tile) { dma_k_piece(tile, 0); dma_k_piece(tile, 1); };
    auto dma_v = [&](int tile) { dma_v_piece(tile, 0); dma_v_piece(tile, 1); };

    // ---- fragment read addresses (bytes inside a tile image)
    // A operand of QK^T: key 32 kb + r, logical chunk 2 ks + hh -> per lane one base per k-step (+ 4096 kb)
    int k_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = r * 128 + (((ks * 2 + hh) ^ ((r >> 1) & 7)) << 4);
    // A operand of PV (V^T through the transposed read): 16-lane group (r >> 4) reads the 4-key x 16-column block at keys
    // 32 kb + 16 st + 4 hh (+ 8 u), columns 32 db + 16 (r >> 4): lane 4 qq + pp supplies row qq, columns 4 pp ..  Byte offset =
    // key * 128 + (((col >> 3) ^ (((key >> 1) & 1) << 2)) << 4) + (col & 4) * 2; with key = 8 (4 kb + 2 st + u) + (4 hh + qq) the swizzle bit
    // is (qq >> 1) & 1 - a lane constant -, so the address is a per-lane base per output block + the immediate 4096 kb + 2048 st + 1024 u.
    unsigned v_lane[NDB];
    {
        const int li = r & 15, qq = li >> 2, pp = li & 3;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
            v_lane[db] = (unsigned)((hh * 4 + qq) * 128 + (((db * 4 + (r >> 4) * 2 + (pp >> 1)) ^ (((qq >> 1) & 1) << 2)) << 4) + (pp & 1) * 8);
    }

    // ---- S = K Q^T as asm MFMAs: destination in VGPRs (the VALU reads it), operands in AGPRs.  An MFMA's result is never read
    // in the iteration that issues it before slot 22 (its last MFMA is slot 15; tools/audit_asm_reads.py checks the distance).
    auto mfma_s0 = [&](f32x16& acc, const v8& a, const v8& bq) {      // acc = a b
        if constexpr (std::is_same<T, f16>::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "a"(bq));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "a"(bq));
    };
    auto mfma_s = [&](f32x16& acc, const v8& a, const v8& bq) {       // acc += a b
        if constexpr (std::is_same<T, f16>::value) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(bq));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(bq));
    };

    f32x16 o_acc[QB][NDB], l_acc[QB];       // O^T blocks and the row sums (every register of l_acc[qb] = the sum of query r)
    float m_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            l_acc[qb][e] = 0.f;
#pragma unroll
            for (int db = 0; db < NDB; ++db) o_acc[qb][db][e] = 0.f;
        }
    }
    v8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
    const float c = p.scale_log2;
    float nm_cur[QB], alpha_cur[QB];          // -(reference maximum) and the older sums' scale for the tile the VALU works on

    // running maximum of one tile's scores (keys >= sk masked first when the tile is the last one) -> nm / alpha of that tile.
    // Seven steps per query block: 4 chain steps over four interleaved 8-score chains, then the finish in three pieces.
    struct MaxState { float pm[QB][4]; float mx[QB]; float m_new[QB]; };
    auto mask_tail = [&](f32x16 (&sv)[QB][2], int kv0) {          // keys >= sk of the tile starting at kv0 -> -inf
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh >= p.sk) sv[qb][kb][e] = -INFINITY;
    };
    auto max_op = [&](auto op_c, f32x16 (&sv)[QB][2], MaxState& ms, float (&nm_out)[QB], float (&alpha_out)[QB]) {
        constexpr int op = decltype(op_c)::value, qb = op / 7, step = op - qb * 7;
        if constexpr (step < 4) {
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                const int kb = part >> 1, b0 = (part & 1) * 8;
                const f32x16& x = sv[qb][kb];
                if (step == 0) ms.pm[qb][part] = __builtin_fmaxf(__builtin_fmaxf(x[b0], x[b0 + 1]), x[b0 + 2]);
                else if (step == 1) ms.pm[qb][part] = __builtin_fmaxf(__builtin_fmaxf(ms.pm[qb][part], x[b0 + 3]), x[b0 + 4]);
                else if (step == 2) ms.pm[qb][part] = __builtin_fmaxf(__builtin_fmaxf(ms.pm[qb][part], x[b0 + 5]), x[b0 + 6]);
                else ms.pm[qb][part] = __builtin_fmaxf(ms.pm[qb][part], x[b0 + 7]);
            }
            asm volatile("" : "+v"(ms.pm[qb][0]), "+v"(ms.pm[qb][1]), "+v"(ms.pm[qb][2]), "+v"(ms.pm[qb][3]));
        } else if constexpr (step == 4) {
            ms.mx[qb] = __builtin_fmaxf(__builtin_fmaxf(ms.pm[qb][0], ms.pm[qb][1]), __builtin_fmaxf(ms.pm[qb][2], ms.pm[qb][3]));
            asm volatile("" : "+v"(ms.mx[qb]));
        } else if constexpr (step == 5) {
            float lo, hi;
            both_halves(ms.mx[qb], lo, hi);
            // deferred rescale (cdna_hip_programming.md T13): the reference maximum follows the scores only when they exceed it by
            // more than PW_THR binades; until then P = exp2(s c - m) may reach 2^PW_THR (a 16-bit float keeps its precision at any
            // scale, sums are fp32), and the rescale of O and l - an AGPR round trip here - stays confined to the first tiles
            const float m_cand = __builtin_fmaxf(lo, hi) * c;
            ms.m_new[qb] = m_cand > m_run[qb] + PW_THR ? m_cand : m_run[qb];
            asm volatile("" : "+v"(ms.m_new[qb]));
        } else {
            alpha_out[qb] = __builtin_amdgcn_exp2f(m_run[qb] - ms.m_new[qb]);
            m_run[qb] = ms.m_new[qb];
            nm_out[qb] = -ms.m_new[qb];
            asm volatile("" : "+v"(nm_out[qb]), "+v"(alpha_out[qb]));
        }
    };

#define PW_SYNC(N)                                                       \
    asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                        \
    __builtin_amdgcn_sched_barrier(0)
#define PW_PIN() __builtin_amdgcn_sched_barrier(0)

    // ---- prologue.  Ring discipline: iteration t requests K(t+4) and V(t+3) - two K and two V pieces per wave - into the slots
    // whose tiles (K(t), V(t-1)) every wave finished reading before it reached iteration t's barrier (fragments of K(t+2) and
    // V(t+1) are read DURING iteration t); "all but my last four pieces have landed" + that barrier then means K(<= t+2) and
    // V(<= t+1) are in LDS for everyone.  The prologue requests K(0..3), V(0..2) in that order of need and starts from the same
    // state.
    dma_k(0); dma_k(1); dma_v(0); dma_k(2); dma_v(1); dma_k(3); dma_v(2);
    PW_SYNC(4);
    v8 kf[2][KS], vf[NDB][4];                 // K(t+1) fragments (kb, ks), V(t)^T fragments (db, chunk)
    f32x16 sA[QB][2], sB[QB][2];
    v8 pf[QB][4];                             // P(t)^T fragments per 16-key chunk: ONE buffer (see the iteration)
    {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const v8*)(Kr + kb * 4096 + k_off[ks]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (ks == 0) mfma_s0(sA[qb][kb], kf[kb][ks], qf[qb][ks]);
                    else mfma_s(sA[qb][kb], kf[kb][ks], qf[qb][ks]);
                }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = *(const v8*)(Kr + PW_TILE_BYTES + kb * 4096 + k_off[ks]);
        const unsigned v0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vr;
        StaticFor<0, 8>::run([&](auto f_c) {
            constexpr int f = decltype(f_c)::value, cch = f >> 1, db = f & 1;
            const v4 lo = tr_read<T, (cch >> 1) * 4096 + (cch & 1) * 2048>(v0 + v_lane[db]);
            const v4 hi = tr_read<T, (cch >> 1) * 4096 + (cch & 1) * 2048 + 1024>(v0 + v_lane[db]);
            v8 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
            vf[db][cch] = x;
        });
        // S(0) is read by the VALU right below: no compiler padding behind asm MFMAs, and the operands keep the reads below it
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sA[0][0]), "+v"(sA[0][1]), "+v"(sA[1][0]), "+v"(sA[1][1]));
        MaxState ms0;
        StaticFor<0, 14>::run([&](auto op_c) { max_op(op_c, sA, ms0, nm_cur, alpha_cur); });
    }

    // iteration t: [barrier: K(<= t+2), V(<= t+1) landed], the rare rescale of O / l by alpha(t), then 40 slots of {one MFMA, a few
    // VALU micro-stages, sometimes a fragment read or an LDS-DMA piece}:
    //   MFMA  16 x S(t+1) = K(t+1) Q^T  and, per query block qb and 16-key chunk c of tile t, 3 x {O^T[qb][0], O^T[qb][1], l[qb]} +=
    //         {V(t)^T, V(t)^T, ones} P(t)[qb][c]^T - each chunk's three MFMAs two slots behind the conversion that completes its P
    //         fragment, so P needs ONE buffer and the V tile no extra iteration of lag;
    //   VALU  P(t) = exp2(S(t) c - m) as a software pipeline over the 32 score pairs, then the maximum of tile t+1;
    //   LDS   the K(t+2) / V(t+1) fragments, each two slots behind the last MFMA that used its register set;
    //   DMA   K(t+4), V(t+3).
    // The order is pinned (sched_barrier + opaque results): left to itself the scheduler issues the MFMAs as one cluster in front
    // of the whole softmax.  SLOT[i] < 16: the QK^T MFMA 8 kb + 2 ks + qb; >= 100: 100 + 3 (4 qb + c) + w.
    static constexpr int SLOT[PW_NSLOT] = {0, 1, 2, 3, 4, 5, 6, 100, 101, 102, 7, 103, 104, 105, 8, 9, 106, 107, 108, 10,
                                           109, 110, 111, 11, 12, 112, 113, 114, 13, 115, 116, 117, 14, 15, 118, 119, 120, 121, 122, 123};
    // fragment re-reads per slot: K fragment f = 4 kb + ks two slots behind its second MFMA (-1: none); V^T fragments 2 c + db two
    // slots behind the MFMA of query block 1 that used them (the last two share the last slot)
    static constexpr int KREL[PW_NSLOT] = {-1, -1, -1, 0, -1, 1, -1, 2, -1, -1, -1, -1, 3, -1, -1, -1, -1, 4, -1, -1,
                                           -1, -1, -1, -1, -1, 5, -1, -1, -1, -1, 6, -1, -1, -1, -1, 7, -1, -1, -1, -1};
    static constexpr int VREL[PW_NSLOT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                           -1, -1, -1, -1, -1, -1, -1, 0, 1, -1, -1, 2, 3, -1, -1, -1, 4, 5, -1, 6};
    auto iter = [&](int t, f32x16 (&s_cur)[QB][2], f32x16 (&s_nxt)[QB][2], auto qk_c) {
        constexpr bool HAS_QK = decltype(qk_c)::value;
        PW_SYNC(4);
        // the older sums follow the new reference maximum: O and l (complete up to tile t-1) scale by alpha(t) - rarely (PW_THR)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
            if (!__all(alpha_cur[qb] == 1.0f)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    l_acc[qb][e] *= alpha_cur[qb];
#pragma unroll
                    for (int db = 0; db < NDB; ++db) o_acc[qb][db][e] *= alpha_cur[qb];
                }
            }
        PW_PIN();
        const char* kbase = Kr + ((t + 2) & (PW_RING - 1)) * PW_TILE_BYTES;
        const unsigned vslot = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vr + ((t + 1) & (PW_RING - 1)) * PW_TILE_BYTES;
        unsigned vaddr[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) vaddr[db] = vslot + v_lane[db];
        float nm_nxt[QB], alpha_nxt[QB];
        MaxState ms;
        // ---- the VALU work of the iteration as 110 micro-stages of 1-4 instructions, dealt over the 40 MFMA slots in a fixed
        // order: the pair pipeline (A(g) = the two exponent arguments of pair g, B(g) = its two exponentials, C(g) = the packed
        // conversion, issued as A(g), B(g-1), C(g-2): 96 stages), then the 14 maximum steps of tile t+1, the two query blocks'
        // chains alternating (S(t+1) is complete behind slot 33).  With ONE wave on the SIMD nothing hides a VALU result's
        // latency, so dependent instructions are never neighbours.  Every micro-stage ends in an empty asm that makes its
        // results opaque: pure arithmetic is otherwise sunk past the pinned MFMAs to its first use.
        float z0[32], z1[32], e0[32], e1[32];
        // pair g = 16 qb + i: scores 2 i, 2 i + 1 of the row; chunk c = i >> 2 = 2 kb + st, registers j = 2 (i & 3), + 1 of its fragment
        auto stage_a = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value, qb = g >> 4, i = g & 15, kb = i >> 3, st = (i >> 2) & 1, j = (i & 3) * 2;
            z0[g] = __builtin_fmaf(s_cur[qb][kb][st * 8 + j], c, nm_cur[qb]);
            z1[g] = __builtin_fmaf(s_cur[qb][kb][st * 8 + j + 1], c, nm_cur[qb]);
            asm volatile("" : "+v"(z0[g]), "+v"(z1[g]));
        };
        auto stage_b = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            e0[g] = __builtin_amdgcn_exp2f(z0[g]);
            e1[g] = __builtin_amdgcn_exp2f(z1[g]);
            asm volatile("" : "+v"(e0[g]), "+v"(e1[g]));
        };
        auto stage_c = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value, qb = g >> 4, i = g & 15, cch = i >> 2, j = (i & 3) * 2;
            unsigned pk = __builtin_bit_cast(unsigned, cvt2<T>(e0[g], e1[g]));
            asm volatile("" : "+v"(pk));
            const typename VecOf<T>::v2 e16 = __builtin_bit_cast(typename VecOf<T>::v2, pk);
            pf[qb][cch][j] = e16[0];
            pf[qb][cch][j + 1] = e16[1];
        };
        constexpr int NMICRO = 110;
        auto micro = [&](auto k_c) {       // 0..95: A0 | A1 B0 | (A(g) B(g-1) C(g-2)), g = 2..31 | B31 C30 | C31; 96..109: the maximum steps
            constexpr int k = decltype(k_c)::value;
            if constexpr (k == 0) stage_a(std::integral_constant<int, 0>());
            else if constexpr (k == 1) stage_a(std::integral_constant<int, 1>());
            else if constexpr (k == 2) stage_b(std::integral_constant<int, 0>());
            else if constexpr (k < 93) {
                constexpr int q = k - 3, g = 2 + q / 3, w = q - (g - 2) * 3;
                if constexpr (w == 0) stage_a(std::integral_constant<int, g>());
                else if constexpr (w == 1) stage_b(std::integral_constant<int, g - 1>());
                else stage_c(std::integral_constant<int, g - 2>());
            } else if constexpr (k == 93) stage_b(std::integral_constant<int, 31>());
            else if constexpr (k == 94) stage_c(std::integral_constant<int, 30>());
            else if constexpr (k == 95) stage_c(std::integral_constant<int, 31>());
            else if constexpr (HAS_QK) {
                constexpr int m = k - 96, qb = m & 1, step = m >> 1;      // the two query blocks' chains alternate
                if constexpr (m == 0) {
                    if ((t + 2) * PW_KT > p.sk) mask_tail(s_nxt, (t + 1) * PW_KT);      // tile t+1 is the last one (wave-uniform)
                }
                max_op(std::integral_constant<int, 7 * qb + step>(), s_nxt, ms, nm_nxt, alpha_nxt);
            }
        };
        auto reload_k = [&](auto f_c) {
            constexpr int f = decltype(f_c)::value, kb = f >> 2, ks = f & 3;
            kf[kb][ks] = *(const v8*)(kbase + kb * 4096 + k_off[ks]);
        };
        auto reload_v = [&](auto f_c) {
            constexpr int f = decltype(f_c)::value, cch = f >> 1, db = f & 1;
            const v4 lo = tr_read<T, (cch >> 1) * 4096 + (cch & 1) * 2048>(vaddr[db]);
            const v4 hi = tr_read<T, (cch >> 1) * 4096 + (cch & 1) * 2048 + 1024>(vaddr[db]);
            v8 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
            vf[db][cch] = x;
        };
        StaticFor<0, PW_NSLOT>::run([&](auto i_c) {
            constexpr int i = decltype(i_c)::value, code = SLOT[i];
            if constexpr (code < 16) {                        // S(t+1)[qb][kb] += K(t+1)[kb][ks] Q[qb][ks]
                constexpr int kb = code >> 3, ks = (code >> 1) & 3, qb = code & 1;
                if constexpr (HAS_QK) {
                    if constexpr (ks == 0) mfma_s0(s_nxt[qb][kb], kf[kb][ks], qf[qb][ks]);
                    else mfma_s(s_nxt[qb][kb], kf[kb][ks], qf[qb][ks]);
                }
            } else {
                constexpr int u = code - 100, qc = u / 3, w = u - qc * 3, qb = qc >> 2, cch = qc & 3;
                if constexpr (w < 2) o_acc[qb][w] = mfma32(vf[w][cch], pf[qb][cch], o_acc[qb][w]);
                else l_acc[qb] = mfma32(ones, pf[qb][cch], l_acc[qb]);
            }
            PW_PIN();
            StaticFor<(i * NMICRO) / PW_NSLOT, ((i + 1) * NMICRO) / PW_NSLOT>::run(micro);
            // (not in the last iteration: an asm read whose result nobody uses still lands - in a register the compiler has
            // meanwhile given to something else)
            if constexpr (HAS_QK) {
                if constexpr (KREL[i] >= 0) reload_k(std::integral_constant<int, KREL[i]>());
                if constexpr (VREL[i] >= 0) reload_v(std::integral_constant<int, VREL[i]>());
                if constexpr (i == PW_NSLOT - 1) reload_v(std::integral_constant<int, 7>());
            }
            if constexpr (i == 2) dma_k_piece(t + 4, 0);
            if constexpr (i == 12) dma_k_piece(t + 4, 1);
            if constexpr (i == 22) dma_v_piece(t + 3, 0);
            if constexpr (i == 32) dma_v_piece(t + 3, 1);
            PW_PIN();
        });
        if constexpr (HAS_QK) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) { nm_cur[qb] = nm_nxt[qb]; alpha_cur[qb] = alpha_nxt[qb]; }
        }
    };
    typedef std::true_type yes;
    typedef std::false_type no;
    // tiles 0 .. ntiles - 2 also compute the next tile's scores; ntiles >= 3
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        iter(t, sA, sB, yes());
        iter(t + 1, sB, sA, yes());
    }
    if (t + 1 < ntiles) {          // two tiles left
        iter(t, sA, sB, yes());
        iter(t + 1, sB, sA, no());
    } else {
        iter(t, sA, sB, no());
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA pieces requested past the last tile

    // ---- finalise: lane (r, hh) holds O[q = q0 + 32 qb + r][32 db + 8 (e >> 2) + 4 hh + (e & 3)] and, in every register of
    // l_acc[qb], the row sum of query r
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float inv = 1.0f / l_acc[qb][0];
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

// -1 when this form does not apply
int tdc_attention_pw(const AttnArgs& a, int batch, int dtype, hipStream_t st) {
    if (a.bias || !a.vec_ok || a.sq < 256 || a.sk < 3 * PW_KT || a.d != 64) return -1;
    if ((a.k_rs & 7) || (a.v_rs & 7)) return -1;
    return dtype == TDC_F16 ? launch_pw64<f16>(a, batch, st) : launch_pw64<bf16>(a, batch, st);
}

// entry of the prototype library: the tdc_attn_desc of include/tdc_hip.h (no bias / mask), TDC_E_BADARG where the form does not apply
extern "C" int tdc_attn_pw_run(const tdc_attn_desc* d, void* stream) {
    if (!d || !d->q || !d->k || !d->v || !d->o || d->bias || d->key_mask || d->head_dim != 64) return TDC_E_BADARG;
    if (d->dtype != TDC_F16 && d->dtype != TDC_BF16) return TDC_E_BADARG;
    AttnArgs a;
    a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o;
    a.q_bs = d->q_bs; a.k_bs = d->k_bs; a.v_bs = d->v_bs; a.o_bs = d->o_bs;
    a.q_rs = d->q_rs; a.k_rs = d->k_rs; a.v_rs = d->v_rs; a.o_rs = d->o_rs;
    a.heads = d->heads; a.d = d->head_dim; a.sq = d->sq; a.sk = d->sk;
    a.scale_log2 = d->scale * 1.4426950408889634f;
    auto al = [](const void* p, int bytes) { return ((uintptr_t)p % bytes) == 0; };
    a.vec_ok = (d->q_rs % 8 == 0) && (d->k_rs % 8 == 0) && (d->v_rs % 8 == 0) && (d->o_rs % 4 == 0) && (d->q_bs % 8 == 0) &&
               (d->k_bs % 8 == 0) && (d->v_bs % 8 == 0) && (d->o_bs % 4 == 0) && al(d->q, 16) && al(d->k, 16) && al(d->v, 16) &&
               al(d->o, 8);
    a.bias = nullptr; a.bias_hs = 0; a.bias_rs = 0; a.gate = nullptr; a.gate_rs = 0; a.kmask = nullptr; a.kmask_bs = 0;
    const int rc = tdc_attention_pw(a, d->batch, d->dtype, (hipStream_t)stream);
    return rc == -1 ? TDC_E_BADARG : rc;
}
