// tdc_attention, long-sequence form for the ViT towers (S = 729 / 730 / 577, d = 64 / 72): flash-style
// softmax(Q K^T * scale) V on v_mfma_f32_32x32x16 tiles.
//
// Why a second kernel: the 16x16x32 form (attention.hip) is bound by the SIMD's vector issue port - per 64 x 64 score tile
// and wave it issues 64 MFMAs (8 issue cycles each) beside ~1000 cycles of softmax VALU work, and its row maximum needs
// two cross-lane exchanges per 16 values.  The 32x32x16 MFMA does twice the work per issue (8 of its 32 cycles), and in
// its C layout a lane holds 16 scores of ONE query (query = lane & 31, keys on the registers): the row maximum / row sum
// are in-register chains plus a single exchange between the two lane halves, the S accumulator converts in place into
// the B operand of the PV product (cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's operand") and the
// output accumulator has the query on the lane again, so the rescale needs no cross-lane traffic either.
//
// Workgroup = 4 waves; a wave owns QB blocks of 32 query rows of one (batch, head); K / V tiles of 64 keys go
// global -> registers -> LDS, double buffered: the loads of tile t+2 are in flight and tile t+1 is written to the other
// buffer while tile t is computed; ONE barrier per tile.
//   S^T[key, q] = K Q^T      A = K rows from LDS (ds_read_b128; 16-byte chunks XOR-swizzled against the 16-lane groups of
//                            a b128 read), B = Q^T held in registers for the whole kernel
//   O^T[d, q] += V^T P^T     A = V^T through ds_read_b64_tr_b16 from the row-major V tile (k-slot (h, j) of step s <-> key
//                            16 s + 8 (j >> 2) + 4 h + (j & 3), the order the S accumulator has), B = P^T = exp2(S^T c - m)
//                            converted pairwise in place
// Head dim 72 runs with the QK^T contraction padded to 80 (5 k-steps of 16; the 16x16x32 form pads to 96) and 3 output
// blocks of 32.
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

template <class T> __device__ __forceinline__ typename VecOf<T>::v4 tr_read32(const T* p) {
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(typename VecOf<T>::v4, r);
}

// value of the other lane half (lane ^ 32)
__device__ __forceinline__ float other_half(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((int)(threadIdx.x & 63) ^ 32) * 4, __builtin_bit_cast(int, v)));
}

constexpr int KT32 = 64;   // keys per tile (2 key blocks of 32)
#ifndef ATTN32_THR
#define ATTN32_THR 8.0f
#endif

// DK: QK^T contraction width (64 or 80), NDB: output blocks of 32 columns (2 or 3), QB: 32-row query blocks per wave
template <class T, int DK, int NDB, int QB>
__global__ __launch_bounds__(256, 2) void attn32_kernel(AttnArgs p) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    constexpr int KS = DK / 16;                              // k-steps of the QK^T product
    constexpr int NCH = DK / 8;                              // real 16-B chunks per K row
    // chunks per K row in LDS: 8 (128-byte rows, XOR swizzle) at head dim 64; at head dim 72 an ODD number of chunks (11 = 176-byte
    // rows, no swizzle): the 16 lanes a ds_read_b128 cycle serves hold 16 keys that are distinct mod 16, and 11 key mod 16 is a
    // permutation, so they hit 16 distinct 16-byte slots - and the staging writes (consecutive lanes = consecutive chunks of
    // consecutive keys) wrap onto an occupied slot once per 16 lanes instead of six times as with round 4's 256-byte rows, where
    // every row started on bank 0 (the d = 72 launches' 3.5 M conflict cycles, profiles/r04_attention_pmc_summary.txt)
    constexpr int KCH = (NCH <= 8) ? 8 : (NCH | 1);
    constexpr int KROW = KCH * 8;
    constexpr int VCH = NDB * 4;                             // 16-B chunks per V row (64 B per output block)
    constexpr int VROW = VCH * 8;                            // d = 64: 128-byte rows, swizzled; d = 72: 192-byte rows
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Ks = (T*)smem_raw;                                    // [2][KT32 * KROW]
    T* Vs = Ks + 2 * KT32 * KROW;                            // [2][KT32 * VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // Workgroups are dispatched round-robin over the 8 XCDs (private L2s): every XCD takes a contiguous range of logical ids, so
    // that the query blocks of one (batch, head) - consecutive logical ids, which read the same K / V - run on ONE L2 and at
    // about the same time (cdna_hip_programming.md T1).  In dispatch order they land on different XCDs and each fetches that
    // head's K / V through the fabric for itself: measured 494 -> 532 TFLOP/s at head dim 72, 686 -> 707 at 64 (same box).
    const int nqb = (p.sq + 128 * QB - 1) / (128 * QB);
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int lid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int bh = lid / nqb, qblk = lid - bh * nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qblk * (128 * QB) + wave * (32 * QB);
    const int d = p.d;
    const T* Q = (const T*)p.q + b * p.q_bs + h * d;
    const T* K = (const T*)p.k + b * p.k_bs + h * d;
    const T* V = (const T*)p.v + b * p.v_bs + h * d;
    T* O = (T*)p.o + b * p.o_bs + h * d;

    // 8 elements row[c0 .. c0+7], address clamped into the row (head dim % 8 == 0: the launcher's vec_ok)
    auto load8 = [&](const T* row, int c0) -> v8 {
        const int cc = c0 < d ? c0 : d - 8;
        return *(const v8*)(row + cc);
    };

    // ---- Q^T fragments (B operand): lane (r, hh) holds Q[q0 + 32 qb + r][16 ks + 8 hh .. +7]; columns >= d zeroed
    v8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = q0 + qb * 32 + r;
        if (qr > p.sq - 1) qr = p.sq - 1;
        const T* row = Q + (long long)qr * p.q_rs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = ks * 16 + hh * 8;
            v8 x = load8(row, c0);
            if (c0 >= d)
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (T)0.f;
            qf[qb][ks] = x;
        }
    }

    // ---- staging: K tile 64 x NCH chunks, V tile 64 x VCH chunks (V columns >= d: clamped duplicates that only feed
    //      output columns >= d, which are never stored)
    constexpr int KLD = (KT32 * NCH + 255) / 256, VLD = (KT32 * VCH + 255) / 256;
    constexpr bool K_EXACT = (KT32 * NCH) % 256 == 0, V_EXACT = (KT32 * VCH) % 256 == 0;
    v8 kreg[KLD], vreg[VLD];
    auto issue_loads = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            int idx = tid + i * 256;
            if (!K_EXACT && idx > KT32 * NCH - 1) idx = KT32 * NCH - 1;
            const int key = idx / NCH, c = idx - key * NCH;
            int kr = kv0 + key; if (kr > p.sk - 1) kr = p.sk - 1;
            kreg[i] = load8(K + (long long)kr * p.k_rs, c * 8);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            int idx = tid + i * 256;
            if (!V_EXACT && idx > KT32 * VCH - 1) idx = KT32 * VCH - 1;
            const int key = idx / VCH, c = idx - key * VCH;
            int kr = kv0 + key; if (kr > p.sk - 1) kr = p.sk - 1;
            vreg[i] = load8(V + (long long)kr * p.v_rs, c * 8);
        }
    };
    // K, 128-byte rows: physical chunk = chunk ^ ((key >> 1) & 7) - the 16 lanes one ds_read_b128 cycle serves ({0-3, 12-15,
    // 20-27} / {4-11, 16-19, 28-31} of each half) then hit 16 distinct 16-byte slots; 176-byte rows do that by their stride.
    // V: 128-byte rows (d = 64): the 64-byte half is XORed with (key >> 1) & 1 so that the four keys of one transposed read
    // fall on four different 64-byte bank groups; 192-byte rows (3 output blocks) do that by their stride.
    auto kswz = [](int key, int c) { return KCH == 8 ? (c ^ ((key >> 1) & 7)) : c; };
    auto vswz = [](int key, int c) { return NDB == 2 ? (c ^ (((key >> 1) & 1) << 2)) : c; };
    auto write_lds = [&](int buf) {
        T* kd = Ks + buf * (KT32 * KROW);
        T* vd = Vs + buf * (KT32 * VROW);
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / NCH, c = idx - key * NCH;
            if (K_EXACT || idx < KT32 * NCH) *(v8*)(kd + key * KROW + (kswz(key, c) << 3)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / VCH, c = idx - key * VCH;
            if (V_EXACT || idx < KT32 * VCH) *(v8*)(vd + key * VROW + (vswz(key, c) << 3)) = vreg[i];
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

    const int ntiles = (p.sk + KT32 - 1) / KT32;
    const float c = p.scale_log2;

    // NQ = query blocks of this wave that hold at least one row < sq (the first NQ ones), NKB = key blocks of the tile that hold
    // at least one key < sk: S = 729 / 730 / 577 leave 32-row blocks past the end of the sequence on both sides (the last
    // workgroup of a head, the last tile of the keys), whose MFMAs and softmax are skipped whole
    auto do_tile = [&](const int tile, auto partial_c, auto nq_c, auto nkb_c) {
        constexpr bool PARTIAL = decltype(partial_c)::value;
        constexpr int NQ = decltype(nq_c)::value, NKB = decltype(nkb_c)::value;
        const int buf = tile & 1;
        const T* kt_ = Ks + buf * (KT32 * KROW);
        const T* vt_ = Vs + buf * (KT32 * VROW);
        const int kv0 = tile * KT32;
        // tile t+1 (in registers since the previous iteration) -> the other buffer; every wave passed the barrier that ended
        // iteration t-1, so nobody reads that buffer any more.  Then the loads of tile t+2 go out.
        if (!PARTIAL) {
            write_lds(buf ^ 1);
            if (tile + 2 < ntiles) issue_loads((tile + 2) * KT32);
        }

        // ---- S^T = K Q^T: s[qb][kb] holds, for query r, keys kv0 + 32 kb + (e & 3) + 8 (e >> 2) + 4 hh
        f32x16 s[QB][2];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[qb][kb][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int key = kb * 32 + r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v8 kf = *(const v8*)(kt_ + key * KROW + (kswz(key, ks * 2 + hh) << 3));
#pragma unroll
                for (int qb = 0; qb < NQ; ++qb) s[qb][kb] = mfma32(kf, qf[qb][ks], s[qb][kb]);
            }
        }
        // ---- online softmax in base 2 on the raw scores; P^T fragments in place
        v8 pf[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
            if (PARTIAL) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh >= p.sk) s[qb][kb][e] = -INFINITY;
            }
            float mx = fmaxf(s[qb][0][0], s[qb][0][1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][0][e]), s[qb][0][e + 1]);
            if (NKB == 2) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][1][e]), s[qb][1][e + 1]);
            }
            mx = fmaxf(mx, other_half(mx));
            // deferred rescale (cdna_hip_programming.md T13): the running maximum only follows the tile's when that exceeds it by
            // more than ATTN32_THR (base-2 exponent units) - P stays <= 2^THR, O and l are rescaled in the first tile or two only
            const float m_cand = mx * c;
            const float m_new = (m_cand > m_run[qb] + ATTN32_THR) ? m_cand : m_run[qb];
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
#ifndef ATTN32_PACKED
            // exponent arguments and row sums as SINGLE-value fma / add (two accumulators): packed fp32 arithmetic (v_pk_fma_f32 /
            // v_pk_add_f32) halves the instruction count but never runs beside the partner wave's MFMAs (tools/mfma_valu_overlap.cpp,
            // MI355X_MICROARCH.md "packed f32 VALU ... an anti-lever beside MFMAs"); round 5, same box, three runs each
            // (profiles/r05_attention32_scalar_ab.log): d = 64 716-733 -> 744-756 TFLOP/s, d = 72 512-518 -> 512-522; s_setprio around the
            // softmax or around the MFMA sections: no gain.  The empty
            // asm keeps the SLP vectoriser from re-packing the pairs.  (-DATTN32_PACKED: the packed form of rounds 2-4.)
            float rs0 = 0.f, rs1 = 0.f;
            const float nm = -m_new;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        float z0 = __builtin_fmaf(s[qb][kb][st * 8 + j], c, nm), z1 = __builtin_fmaf(s[qb][kb][st * 8 + j + 1], c, nm);
                        asm volatile("" : "+v"(z0), "+v"(z1));
                        const float e0 = __builtin_amdgcn_exp2f(z0), e1 = __builtin_amdgcn_exp2f(z1);
                        rs0 += e0; rs1 += e1;
                        const typename VecOf<T>::v2 e16 = cvt2<T>(e0, e1);
                        pf[qb][kb][st][j] = e16[0];
                        pf[qb][kb][st][j + 1] = e16[1];
                    }
            l_run[qb] = l_run[qb] * alpha + (rs0 + rs1);
#else
            // exponent arguments and row sums on register pairs (v_pk_fma_f32 / v_pk_add_f32: one issue slot per two values)
            const f32x2_t c2 = {c, c}, nm2 = {-m_new, -m_new};
            f32x2_t rs2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2_t z = __builtin_elementwise_fma((f32x2_t){s[qb][kb][st * 8 + j], s[qb][kb][st * 8 + j + 1]}, c2, nm2);
                        const f32x2_t e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
                        rs2 += e;
                        const typename VecOf<T>::v2 e16 = cvt2<T>(e[0], e[1]);     // one v_cvt_pk per pair, f16 too
                        pf[qb][kb][st][j] = e16[0];
                        pf[qb][kb][st][j + 1] = e16[1];
                    }
            l_run[qb] = l_run[qb] * alpha + (rs2[0] + rs2[1]);
#endif
            // the running max only moves in the first few tiles: skip the O rescale when no lane's max changed
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o_acc[qb][db][e] *= alpha;
            }
        }
        // ---- O^T += V^T P^T.  16-lane group G of the wave (r >> 4 within each half) reads the 4-key x 16-column block at keys
        //      32 kb + 16 st + 4 hh (+ 8), columns 32 db + 16 (r >> 4): lane 4 q + pp supplies row q, columns 4 pp ..
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            if (NQ == 0) break;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int li = r & 15, qq = li >> 2, pp = li & 3;
                    const int key = kb * 32 + st * 16 + hh * 4 + qq;
                    const int col = db * 32 + (r >> 4) * 16 + pp * 4;      // multiple of 4: chunk col >> 3, 8-byte half (col >> 2) & 1
                    const T* a0 = vt_ + key * VROW + (vswz(key, col >> 3) << 3) + (col & 4);
                    const T* a1 = vt_ + (key + 8) * VROW + (vswz(key + 8, col >> 3) << 3) + (col & 4);
                    const v4 lo = tr_read32<T>(a0);
                    const v4 hi = tr_read32<T>(a1);
                    v8 vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
#pragma unroll
                    for (int qb = 0; qb < NQ; ++qb) o_acc[qb][db] = mfma32(vf, pf[qb][kb][st], o_acc[qb][db]);
                }
        }
        __syncthreads();    // tile t+1 is visible; everyone is done reading tile t
    };

    // prologue: tile 0 -> buffer 0, tile 1 in registers
    issue_loads(0);
    write_lds(0);
    if (ntiles > 1) issue_loads(KT32);
    __syncthreads();
    int nq = (p.sq - q0 + 31) >> 5;                              // wave-uniform
    nq = nq < 0 ? 0 : (nq > QB ? QB : nq);
    const bool last_two = p.sk - (ntiles - 1) * KT32 > 32;       // the last tile's second key block holds valid keys
    auto run = [&](auto nq_c) {
        typedef std::integral_constant<int, 2> two;
        typedef std::integral_constant<int, 1> one;
        for (int tile = 0; tile < ntiles - 1; ++tile) do_tile(tile, std::false_type(), nq_c, two());
        if (last_two) do_tile(ntiles - 1, std::true_type(), nq_c, two());
        else do_tile(ntiles - 1, std::true_type(), nq_c, one());
    };
    if (nq >= QB) run(std::integral_constant<int, QB>());
    else if (QB > 1 && nq == 1) run(std::integral_constant<int, 1>());
    else run(std::integral_constant<int, 0>());

    // ---- finalise: lane (r, hh) holds O[q = q0 + 32 qb + r][32 db + 8 (e >> 2) + 4 hh + (e & 3)]
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
                if (col + 3 < d) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o_acc[qb][db][g4 * 4 + e] * inv;
                    *(v4*)(orow + col) = cvt4<T>(o);
                }
            }
    }
}

template <class T, int DK, int NDB, int QB>
int launch32(const AttnArgs& a, int batch, hipStream_t st) {
    constexpr int KCH = (DK / 8 <= 8) ? 8 : ((DK / 8) | 1);
    constexpr int lds = 2 * KT32 * (KCH * 8 + NDB * 32) * 2;
    static bool attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_dev[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn32_kernel<T, DK, NDB, QB>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_dev[dev] = true;
    }
    dim3 grid(((a.sq + 128 * QB - 1) / (128 * QB)) * a.heads * batch);   // 1-D: the kernel maps ids to (batch, head, query block)
    hipLaunchKernelGGL((attn32_kernel<T, DK, NDB, QB>), grid, dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

// entry for attention.hip: returns -1 when this form does not apply (the caller falls through to the 16x16 kernels)
int tdc_attention32(const AttnArgs& a, int batch, int dtype, hipStream_t st) {
    if (a.bias || !a.vec_ok || a.sq < 256 || a.sk < 64) return -1;
    if (a.d == 64) return dtype == TDC_F16 ? launch32<f16, 64, 2, 2>(a, batch, st) : launch32<bf16, 64, 2, 2>(a, batch, st);
    if (a.d > 64 && a.d <= 80) return dtype == TDC_F16 ? launch32<f16, 80, 3, 1>(a, batch, st) : launch32<bf16, 80, 3, 1>(a, batch, st);
    return -1;
}
