// tdc_qformer_xattn: the Q-Former cross-attention block of one layer as ONE kernel:
//
//     q   = h[:, :K] Wq^T + bq                                   tdc/Qformer.py:128 (self.query), :205
//     ctx = softmax(q_h k_h^T / sqrt(64)) v_h   per head          :213-264 (key / value of the frame's encoder tokens, :185-188)
//     h'  = LayerNorm(ctx Wo^T + bo + h[:, :K])                   BertSelfOutput :285-289
//
// for the K query rows of every compressed frame.  The encoder-side projections k = enc Wk^T + bk and v = enc Wv^T (all six
// cross layers at once) stay ordinary tdc_gemm launches in front of it - they are 97 % of the block's FLOPs and run at the
// GEMM's rate - with one twist: V is produced TRANSPOSED (V^T = Wv enc^T, a GEMM with the operands swapped), so that the
// PV product reads its A operand (V^T rows = head-dim columns, keys contiguous) straight from global memory like every other
// operand here; bv is added after the PV product (softmax rows sum to 1).
//
// Workgroup = 8 waves = 64 consecutive query rows of the flat [F*K] row space (K % 16 == 0: a 16-row tile never straddles a
// frame).  One 96-KiB LDS buffer [64][768] (16-byte chunks XOR-swizzled by the row) holds, in turn, the rows' hidden state
// (A operand of the q projection), q (overwritten head by head by ctx) and ctx (A operand of the output projection):
//   P1  q-proj    64 x 768 x 768: wave w owns output columns [96 w, 96 w + 96); weight fragments come straight from global
//                 memory (L2-resident, no reuse between waves) out of a FRAGMENT-MAJOR copy of the weight (xattn_tile_weight:
//                 the 64 lanes' 16-byte operands of one MFMA are 1 KiB of consecutive bytes), activation fragments from LDS.
//                 A fragment load of a row-major matrix touches 16 rows x 64 B with consecutive lanes on different rows:
//                 the texture addresser takes ~56 cycles for such an instruction against 16 for a contiguous one
//   P2  attention 48 (head, row tile) units over the 8 waves; S^T = K Q^T with the key on the MFMA row (softmax statistics in
//                 registers + two cross-lane steps), the S^T accumulators converted in place into the B operand of
//                 O^T = V^T P^T; ctx written over q's columns of its own rows.  K and V^T are row-major GEMM outputs: they
//                 are loaded coalesced (lane 4 r + c takes chunk c of row r) and moved to their MFMA lanes (16 g + i <- 4 i + g)
//                 with ds_bpermute; the keys sit on the MFMA rows in the order that makes a lane's eight P values of a
//                 32-key step CONSECUTIVE keys, so the V^T operand is one 16-byte piece per lane
//   P3  out-proj  as P1, then + bias + fp32 residual, LayerNorm (two-pass statistics, partial sums across the 8 waves through
//                 LDS), fp32 + 16-bit rows stored.
#include "common.h"
#include "../../include/tdc_hip.h"
#include "profile.h"
#include <stdio.h>
#include <type_traits>

// XATTN_DIAG (tools/xattn_diag.py only, never the library build): bit mask of pieces left out for timing - 1 = the q-proj
// MFMA loop, 2 = the attention phase, 4 = the out-proj MFMA loop, 8 = the final stores, 128 = the staging loads; inside the
// attention phase 16 = consumer waves idle (barriers only), 32 = loader waves request nothing.  Results are then wrong by design.
#ifndef XATTN_DIAG
#define XATTN_DIAG 0
#endif

namespace {

constexpr int XD = 768;          // hidden size of the Q-Former (bert-base)
constexpr int XHD = 64;          // head dim
constexpr int XROWS = 64;        // query rows per workgroup
constexpr int XBUF = XROWS * XD * 2;

struct XattnArgs {
    const void* h16; float* h32; void* h16_out; int ldh;
    const void* ctx; int ldctx;           // OUT_ONLY form: the attention output of the flat query rows [rows, ldctx]
    int res16;                            // OUT_ONLY form: residual = the 16-bit rows h16, only h16 is written (no fp32 master)
    int rows, K, S;                       // rows = F*K flat query rows; global row of flat row r: (r / K) * S + r % K
    unsigned invK;                        // floor(2^32 / K)
    const void *wq, *wo; const float *bq, *bo;     // wq / wo: fragment-major copies (tdc_qformer_xattn_tile_weight)
    const void* k; int ldk;               // [F*Nenc, ldk], this layer's 768 columns
    const void* vt; long long ldvt;       // [768, ldvt]: vt[c][f*Nenc + key]
    const float* bv;
    int Nenc;
    const float *ln_g, *ln_b; float eps;
    float scale_log2;
    int nblocks;
};

// byte offset of element (row, col) in the swizzled [64][768] 16-bit buffer; col % 4 == 0 for 8-byte, % 8 for 16-byte accesses
__device__ __forceinline__ int buf_off(int row, int col) {
    const int ch = col >> 3;
    return row * (XD * 2) + ((((ch ^ row) & 15) | (ch & ~15)) << 4) + ((col & 7) << 1);
}

// flat / K through the host-computed reciprocal floor(2^32 / K) and two corrections (q_true - 2 <= q <= q_true)
__device__ __forceinline__ int div_of(int flat, int K, unsigned invK) {
    unsigned q = __umulhi((unsigned)flat, invK);
    unsigned r = (unsigned)flat - q * (unsigned)K;
    if (r >= (unsigned)K) { q += 1; r -= (unsigned)K; }
    if (r >= (unsigned)K) { q += 1; }
    return (int)q;
}
// row of flat query row `flat` in the [F*S, ld] hidden stream
__device__ __forceinline__ int grow_of(int flat, int K, int S, unsigned invK) {
    const int q = div_of(flat, K, invK);
    return q * S + (flat - q * K);
}

// acc[nt][mt] += W[96 w + 16 nt + i][k] * buf[16 mt + i'][k] over k = 0..767 (swapped MFMA: lane (g, i) ends up with output
// columns 96 w + 16 nt + 4 g + reg of row 16 mt + i).  K advances in steps of 64 = one 128-byte line of a weight row: lane
// group g takes the line's 16-byte chunks 2 g and 2 g + 1 (one MFMA each; the LDS side reads the same chunks, so the
// contraction pairs the right columns) - both halves of a line are requested back to back and hit L1 together; fetched
// one 32-column step apart, the second half found its line evicted by the 768 lines the workgroup touches per step.
// The weight fragments of step p + 1 are requested before the MFMAs of step p (sched_barrier: left to itself the scheduler
// sinks every load to its first use and waits vmcnt(0) in front of each group of four MFMAs).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <class T>
__device__ __forceinline__ void gemm_64x768(const T* __restrict__ Wt, const char* buf, int wave, int lane_in, f32x4 (&acc)[6][4]) {
    typedef typename VecOf<T>::v8 v8;
    // lane-derived addresses are rebuilt from an opaque copy of the lane id in each of the two GEMM phases: shared between
    // them they were kept alive across the attention phase - spilled, and reloaded inside this loop, where a scratch reload
    // waits (in-order vmcnt) for every weight fragment in flight
    const int lane = opaque(lane_in);
    const int g = lane >> 4, i = lane & 15;
    // fragment-major weight: piece ((nt * 12 + kp) * 2 + hh) of 1 KiB holds, for lane 16 g + i, W[16 nt + i][64 kp + 16 g + 8 hh ..+7]
    const char* wb = (const char*)Wt + (long long)(6 * wave) * (12 * 2 * 1024) + lane * 16;
    // activations: chunk 8 kp + 2 g + hh of row 16 mt + i - four bases (parity of kp, hh) + immediates: 256 (kp >> 1) bytes along
    // the row, 24576 mt down the rows
    unsigned ab[2][2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) ab[par][hh] = (unsigned)(i * (XD * 2) + ((((8 * par + 2 * g + hh) ^ i) & 15) << 4));
    // a stage = one 64-column K step of THREE of the wave's six column tiles (24 registers of weight fragments): PF stages
    // in flight beside the one being multiplied keep the register count under 256 with the 96 accumulators
    constexpr int NS = 2 * (XD / 64);   // 24 stages
    constexpr int PF = 2;
    v8 wf[PF + 1][3][2];
    auto issue = [&](int st) {
        const int kp = st >> 1, ng = st & 1;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const char* src = wb + ((3 * ng + j) * 12 + kp) * 2048;
            wf[st % (PF + 1)][j][0] = *(const v8*)(src);
            wf[st % (PF + 1)][j][1] = *(const v8*)(src + 1024);
        }
    };
#pragma unroll
    for (int st = 0; st < PF; ++st) issue(st);
    // activation fragments run one half-step ahead of their MFMAs (read from LDS under the previous half-step's MFMAs)
    v8 af[2][4];
    auto read_a = [&](int hs, int slot) {          // half-step hs = 2 st + hh
        const int kp = hs >> 2, hh = hs & 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[slot][mt] = *(const v8*)(buf + ab[kp & 1][hh] + 256 * (kp >> 1) + 24576 * mt);
    };
    read_a(0, 0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        if (st + PF < NS) issue(st + PF);
        const int ng = st & 1;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int hs = 2 * st + hh;
            if (hs + 1 < 2 * NS) read_a(hs + 1, (hs + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[3 * ng + j][mt] = mfma16(wf[st % (PF + 1)][j][hh], af[hs & 1][mt], acc[3 * ng + j][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// NKT: key tiles of 16 (even), Nenc <= 16 NKT.  OUT_ONLY: the block's last third alone - ctx (the attention output, read from
// global memory) -> output projection + residual + LayerNorm; q projection and attention stay separate launches in front of it.
// RES16 (with OUT_ONLY): the residual is read from the 16-bit rows and only they are written - half the bytes of the launch.
template <class T, int NKT, bool OUT_ONLY, bool RES16 = false>
__global__ __launch_bounds__(512, 2) void xattn_kernel(XattnArgs p) {
    typedef typename VecOf<T>::v8 v8;
    typedef typename VecOf<T>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* buf = smem;
    float* red = (float*)(smem + XBUF);            // [2][8 waves][64 rows]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    // XCD-contiguous block ids: the blocks of one frame (K / 64 of them) share its K / V^T through one L2
    const int nb = p.nblocks;
    int bid = blockIdx.x;
    {
        const int per = (nb + 7) >> 3, x = bid & 7, j = bid >> 3;
        bid = x * per + j;
        if (j >= per || bid >= nb) return;
    }
    const int r0 = bid * XROWS;

    // ---- stage the hidden rows (16-bit copy) of the block - or, OUT_ONLY, its attention output rows: 64 x 96 chunks of 16 bytes
    {
        const T* H = (const T*)(OUT_ONLY ? p.ctx : p.h16);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int idx = tid + j * 512;
            const int row = idx / 96, ch = idx - row * 96;
            int flat = r0 + row;
            if (flat > p.rows - 1) flat = p.rows - 1;
            const long long src = OUT_ONLY ? (long long)flat * p.ldctx : (long long)grow_of(flat, p.K, p.S, p.invK) * p.ldh;
            if (XATTN_DIAG & 128) continue;
            const v8 v = *(const v8*)(H + src + ch * 8);
            *(v8*)(buf + buf_off(row, ch * 8)) = v;
        }
    }
    __syncthreads();

    f32x4 acc[6][4];
    if constexpr (!OUT_ONLY) {
    // ---- P1: q = h Wq^T + bq
#pragma unroll
    for (int nt = 0; nt < 6; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(XATTN_DIAG & 1)) gemm_64x768<T>((const T*)p.wq, buf, wave, lane, acc);
    __syncthreads();                               // every wave has read its A fragments: the buffer may be overwritten
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int n0 = 96 * wave + 16 * nt + 4 * g;
        const f32x4 b4 = *(const f32x4*)(p.bq + n0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *(v4*)(buf + buf_off(16 * mt + i, n0)) = cvt4<T>(acc[nt][mt] + b4);
    }
    __syncthreads();

    // ---- P2: attention, one row tile per CONSUMER wave (waves 0-3); waves 4-7 are LOADERS that stage the head's keys and
    //      transposed values through a two-slot LDS ring, one item (K_h or V^T_h of one frame: 20 KiB at 160 keys) per step:
    //          step n:  barrier | loaders: write item n + 1 (loaded during step n - 1) into slot (n + 1) & 1, request item n + 2
    //                           | consumers: item n = K_h: S^T = K Q^T, softmax  /  item n = V^T_h: O^T = V^T P^T, ctx
    //      Why: a wave-level load of a row-major operand tile (16 rows x 64 B) costs the texture addresser / L1 ~90 cycles even
    //      on hits (measured: with every unit reading the SAME lines the loads still took 0.16 of the phase's 0.19 ms); loading
    //      each (frame, head) once per block instead of once per row-tile pair, by waves that do nothing else, removes 3/4 of
    //      those instructions and takes them off the compute waves' critical path.  Loader lane s of a 1-KiB piece fetches
    //      (row s >> 2, 16-byte chunk (s & 3) ^ ((s >> 4) & 3)): a quad reads one row's 64 bytes, and the MFMA lane (g, i) finds
    //      its operand at slot 4 i + (g ^ (i >> 2)) of the piece - conflict-free ds_read_b128.
    //      Keys sit on the MFMA rows in the order 32 (kt >> 1) + 8 (r >> 2) + 4 (kt & 1) + (r & 3): a lane's eight P values of a
    //      32-key step are then CONSECUTIVE keys and the V^T operand is one 16-byte piece per lane.
    {
        const T* Kg = (const T*)p.k;
        const T* Vt = (const T*)p.vt;
        const int Nenc = p.Nenc;
        char* ring = smem + XBUF + 2 * 8 * 64 * (int)sizeof(float);     // 2 slots of NKT * 2 KiB
        constexpr int SLOT = NKT * 2048;
        // frames of the block's row tiles (consecutive frames; tiles past the end do not count)
        const int f_first = div_of(r0, p.K, p.invK);
        int last_row = r0 + XROWS - 1;
        if (last_row > p.rows - 1) last_row = p.rows - 1;
        const int nfr = div_of(last_row, p.K, p.invK) - f_first + 1;
        const int NI = ((XATTN_DIAG & 2) ? 0 : 24) * nfr;               // items: (head, frame, K | V)
        typedef v8 __attribute__((aligned(8))) v8a;
        if (wave >= 4) {
            // ------------------------------------------------------------------------------------------------ loaders
            const int lw = wave - 4;
            const int srow = lane >> 2, sch = (lane & 3) ^ ((lane >> 4) & 3);
            constexpr int PPW = NKT / 2;            // pieces per loader wave and item: NKT * 2 pieces over 4 waves
            // FOUR items in flight in registers per loader wave (the ring only has to hold the item being consumed and the next
            // one): with one item in flight every step waited for a full memory round trip - the phase took LONGER than with
            // the loads inside the compute waves
            v8 reg[4][PPW];
            auto request = [&](int n, auto q_c) {
                constexpr int q = decltype(q_c)::value;
                if (XATTN_DIAG & 32) {
#pragma unroll
                    for (int j = 0; j < PPW; ++j) reg[q][j] = (v8){};
                    return;
                }
                const int h = n / (2 * nfr), fr = f_first + (n >> 1) % nfr;
                const long long kbase = (long long)fr * Nenc;
#pragma unroll
                for (int j = 0; j < PPW; ++j) {
                    const int piece = lw * PPW + j;
                    if (!(n & 1)) {                  // K: piece = (key tile kt, k-step ks)
                        const int kt = piece >> 1, ks = piece & 1;
                        int key = 32 * (kt >> 1) + 8 * (srow >> 2) + 4 * (kt & 1) + (srow & 3);
                        if (key > Nenc - 1) key = Nenc - 1;
                        reg[q][j] = *(const v8*)(Kg + (kbase + key) * p.ldk + 64 * h + 32 * ks + 8 * sch);
                    } else {                         // V^T: piece = (head-dim tile dt, 32-key step st)
                        const int dt = piece / (NKT / 2), st = piece - dt * (NKT / 2);
                        int k0 = 32 * st + 8 * sch;
                        const bool half = (Nenc - k0 == 4);   // the last group of eight keys may be half valid
                        if (k0 > Nenc - 4) k0 = Nenc - 8;     // a group past the end: probabilities are exactly 0 there
                        v8 v = *(const v8a*)(Vt + (long long)(64 * h + 16 * dt + srow) * p.ldvt + kbase + k0);
                        if (half)
#pragma unroll
                            for (int e = 4; e < 8; ++e) v[e] = (T)0.f;
                        reg[q][j] = v;
                    }
                }
            };
            auto deposit = [&](int n, auto q_c) {
                constexpr int q = decltype(q_c)::value;
#pragma unroll
                for (int j = 0; j < PPW; ++j) *(v8*)(ring + (n & 1) * SLOT + (lw * PPW + j) * 1024 + lane * 16) = reg[q][j];
            };
            typedef std::integral_constant<int, 0> Q0;
            typedef std::integral_constant<int, 1> Q1;
            typedef std::integral_constant<int, 2> Q2;
            typedef std::integral_constant<int, 3> Q3;
            if (NI > 0) {                           // NI is a multiple of 24: item n lives in register set n & 3
                request(0, Q0()); request(1, Q1()); request(2, Q2()); request(3, Q3());
                deposit(0, Q0());
                request(4, Q0());
            }
            auto step = [&](int n, auto qn_c) {     // qn = (n + 1) & 3
                __syncthreads();
                if (n + 1 < NI) deposit(n + 1, qn_c);
                if (n + 5 < NI) request(n + 5, qn_c);
            };
            for (int n = 0; n < NI; n += 4) {
                step(n, Q1()); step(n + 1, Q2()); step(n + 2, Q3()); step(n + 3, Q0());
            }
        } else {
            // ---------------------------------------------------------------------------------------------- consumers
            const int rt = wave;
            const int flat0 = r0 + 16 * rt;
            const bool valid = flat0 < p.rows;
            const int my_frame = valid ? div_of(flat0, p.K, p.invK) : -1;
            const f32x4 c4 = {p.scale_log2, p.scale_log2, p.scale_log2, p.scale_log2};
            const int slot_off = (4 * i + (g ^ (i >> 2))) * 16;
            v8 pf[NKT / 2];
            float inv = 0.f;
            for (int n = 0; n < NI; ++n) {
                __syncthreads();
                const int h = n / (2 * nfr), fr = f_first + (n >> 1) % nfr;
                if (fr != my_frame || (XATTN_DIAG & 16)) continue;      // wave-uniform
                const char* slot = ring + (n & 1) * SLOT + slot_off;
                if (!(n & 1)) {
                    v8 qf[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const v8*)(buf + buf_off(16 * rt + i, 64 * h + 32 * ks + 8 * g));
                    // every operand of the step is read from LDS before the first MFMA (one wave per SIMD computes here: left to
                    // the scheduler each key tile is a chain read -> wait -> MFMA with the LDS latency exposed twenty times)
                    v8 kf[NKT][2];
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) kf[kt][ks] = *(const v8*)(slot + (2 * kt + ks) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 s[NKT];
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {
                        f32x4 a = {0.f, 0.f, 0.f, 0.f};
                        a = mfma16(kf[kt][0], qf[0], a);
                        s[kt] = mfma16(kf[kt][1], qf[1], a);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {
                        if (32 * (kt >> 1) + 4 * (kt & 1) + 28 <= Nenc) continue;       // wave-uniform: every key of the tile exists
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r >= Nenc) s[kt][r] = -INFINITY;
                    }
                    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
                    for (int kt = 1; kt < NKT; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float nm = -mx * p.scale_log2;
                    const f32x4 nm4 = {nm, nm, nm, nm};
                    f32x4 rs4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {
                        const f32x4 z = __builtin_elementwise_fma(s[kt], c4, nm4);
                        f32x4 e;
#pragma unroll
                        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(z[r]);
                        const v4 e16 = cvt4<T>(e);
#pragma unroll
                        for (int r = 0; r < 4; ++r) pf[kt >> 1][(kt & 1) * 4 + r] = e16[r];
                        rs4 += e;
                    }
                    float l = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
                    l += __shfl_xor(l, 16);
                    l += __shfl_xor(l, 32);
                    inv = 1.0f / l;
                } else {
                    v8 vf[4][NKT / 2];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int st = 0; st < NKT / 2; ++st) vf[dt][st] = *(const v8*)(slot + (dt * (NKT / 2) + st) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int st = 0; st < NKT / 2; ++st) o = mfma16(vf[dt][st], pf[st], o);
                        const int c0 = 64 * h + 16 * dt + 4 * g;
                        f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
                        if (p.bv) bv4 = *(const f32x4*)(p.bv + c0);
                        *(v4*)(buf + buf_off(16 * rt + i, c0)) = cvt4<T>(o * inv + bv4);
                    }
                }
            }
        }
    }
    __syncthreads();
    }   // !OUT_ONLY

    // ---- P3: out = ctx Wo^T + bo + residual, LayerNorm
#pragma unroll
    for (int nt = 0; nt < 6; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(XATTN_DIAG & 4)) gemm_64x768<T>((const T*)p.wo, buf, wave, lane, acc);
    long long grow[4];
    bool live[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int flat = r0 + 16 * mt + i;
        live[mt] = flat < p.rows;
        if (!live[mt]) flat = p.rows - 1;
        grow[mt] = (long long)grow_of(flat, p.K, p.S, p.invK) * p.ldh;
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int n0 = 96 * wave + 16 * nt + 4 * g;
        const f32x4 b4 = *(const f32x4*)(p.bo + n0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 r4;
            if constexpr (RES16) {
                const v4 r16 = *(const v4*)((const T*)p.h16 + grow[mt] + n0);
                r4 = (f32x4){(float)r16[0], (float)r16[1], (float)r16[2], (float)r16[3]};
            } else {
                r4 = *(const f32x4*)(p.h32 + grow[mt] + n0);
            }
            acc[nt][mt] = (acc[nt][mt] + b4) + r4;
            sum[mt] += (acc[nt][mt][0] + acc[nt][mt][1]) + (acc[nt][mt][2] + acc[nt][mt][3]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        sum[mt] += __shfl_xor(sum[mt], 16);
        sum[mt] += __shfl_xor(sum[mt], 32);
        if (g == 0) red[wave * 64 + 16 * mt + i] = sum[mt];
    }
    __syncthreads();
    float mean[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w * 64 + 16 * mt + i];
        mean[mt] = t * (1.0f / XD);
    }
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 6; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 d = acc[nt][mt] - mean[mt];
            sq[mt] += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        sq[mt] += __shfl_xor(sq[mt], 16);
        sq[mt] += __shfl_xor(sq[mt], 32);
        if (g == 0) red[512 + wave * 64 + 16 * mt + i] = sq[mt];
    }
    __syncthreads();
    float rstd[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[512 + w * 64 + 16 * mt + i];
        rstd[mt] = rsqrtf(t * (1.0f / XD) + p.eps);
    }
    // the normalised rows leave through LDS: a lane holds 4 columns of 16 DIFFERENT rows per accumulator tile, i.e. every direct
    // store instruction would touch 16 rows x 64 B (48 of them per wave, 0.09 of this kernel's 0.45 ms); staged as fp32 in two
    // halves of 32 rows (the buffer is free: every wave passed two barriers since its last ctx read), each thread then stores
    // whole 16-byte pieces of consecutive addresses - the fp32 master and, converted on the way out, the 16-bit copy
    T* H16 = (T*)p.h16_out;
    if constexpr (RES16) {
        // 16-bit rows only: all 64 rows fit the image, one pass (8-byte pieces in, 16-byte pieces of consecutive addresses out)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            const int n0 = 96 * wave + 16 * nt + 4 * g;
            const f32x4 g4 = *(const f32x4*)(p.ln_g + n0), b4 = *(const f32x4*)(p.ln_b + n0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 y = (acc[nt][mt] - mean[mt]) * rstd[mt] * g4 + b4;
                *(v4*)(buf + buf_off(16 * mt + i, n0)) = cvt4<T>(y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int idx = tid + j * 512;
            const int row = idx / 96, ch = idx - row * 96;
            const int flat = r0 + row;
            if (flat >= p.rows || (XATTN_DIAG & 8)) continue;
            const v8 y = *(const v8*)(buf + buf_off(row, ch * 8));
            *(v8*)(H16 + (long long)grow_of(flat, p.K, p.S, p.invK) * p.ldh + ch * 8) = y;
        }
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();                 // the copy-out of the first half is done
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            const int n0 = 96 * wave + 16 * nt + 4 * g;
            const f32x4 g4 = *(const f32x4*)(p.ln_g + n0), b4 = *(const f32x4*)(p.ln_b + n0);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const int mt = 2 * half + m2, row = 16 * m2 + i, ch = n0 >> 2;      // 16-byte chunk of the 3072-byte row
                const f32x4 y = (acc[nt][mt] - mean[mt]) * rstd[mt] * g4 + b4;
                *(f32x4*)(buf + row * (XD * 4) + ((((ch ^ row) & 15) | (ch & ~15)) << 4)) = y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int idx = tid + j * 512;
            const int row = idx / 192, ch = idx - row * 192;
            const int flat = r0 + 32 * half + row;
            if (flat >= p.rows || (XATTN_DIAG & 8)) continue;
            const f32x4 y = *(const f32x4*)(buf + row * (XD * 4) + ((((ch ^ row) & 15) | (ch & ~15)) << 4));
            const long long go = (long long)grow_of(flat, p.K, p.S, p.invK) * p.ldh + ch * 4;
            *(f32x4*)(p.h32 + go) = y;
            *(v4*)(H16 + go) = cvt4<T>(y);
        }
    }
}

template <class T>
int launch(const XattnArgs& a, int nkt, hipStream_t st) {

    const int grid = ((a.nblocks + 7) / 8) * 8;
#define XLAUNCH(N, OO) XLAUNCH_(N, OO, false)
#define XLAUNCH_(N, OO, R16)                                                                                                  \
    do {                                                                                                                 \
        const size_t lds = XBUF + 2 * 8 * 64 * sizeof(float) + (OO ? 0 : 2 * N * 2048);  /* rows, LayerNorm partials, K / V^T ring */ \
        static bool attr[16];                                                                                            \
        int dev = 0;                                                                                                     \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return TDC_E_BADARG;                               \
        if (!attr[dev]) {                                                                                                \
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)xattn_kernel<T, N, OO, R16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr[dev] = true;                                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((xattn_kernel<T, N, OO, R16>), dim3(grid), dim3(512), lds, st, a);                                 \
    } while (0)
    if (a.ctx && a.res16) XLAUNCH_(10, true, true);
    else if (a.ctx) XLAUNCH(10, true);
    else if (nkt <= 10) XLAUNCH(10, false);
    else XLAUNCH(14, false);
#undef XLAUNCH
#undef XLAUNCH_
    return (int)hipGetLastError();
}

// fragment-major copy of a [768, ldw] 16-bit weight: piece ((nt * 12 + kp) * 2 + hh) of 1 KiB holds, for lane l = 16 g + i, the
// eight values W[16 nt + i][64 kp + 16 g + 8 hh .. + 7] - the A operand of one MFMA of gemm_64x768 as 64 consecutive 16-byte pieces
template <class T>
__global__ void xattn_tile_weight_kernel(const T* __restrict__ w, int ldw, T* __restrict__ out) {
    typedef typename VecOf<T>::v8 v8;
    const int piece = blockIdx.x, lane = threadIdx.x;
    const int hh = piece & 1, kp = (piece >> 1) % 12, nt = (piece >> 1) / 12;
    const int g = lane >> 4, i = lane & 15;
    *(v8*)(out + ((long long)piece * 64 + lane) * 8) = *(const v8*)(w + (long long)(16 * nt + i) * ldw + 64 * kp + 16 * g + 8 * hh);
}

}  // namespace

extern "C" int tdc_qformer_xattn_supported(int dim, int heads, int K, int Nenc) {
    return dim == XD && heads * XHD == XD && K > 0 && K % 16 == 0 && Nenc >= 8 && Nenc % 4 == 0 && Nenc <= 224;
}

extern "C" int tdc_qformer_xattn_tile_weight(const void* w, int ldw, void* out, int dtype, void* stream) {
    if (!w || !out || ldw < XD || ldw % 8 || ((uintptr_t)w & 15) || ((uintptr_t)out & 15)) return TDC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == TDC_F16) hipLaunchKernelGGL(xattn_tile_weight_kernel<f16>, dim3(48 * 12 * 2), dim3(64), 0, st, (const f16*)w, ldw, (f16*)out);
    else if (dtype == TDC_BF16) hipLaunchKernelGGL(xattn_tile_weight_kernel<bf16>, dim3(48 * 12 * 2), dim3(64), 0, st, (const bf16*)w, ldw, (bf16*)out);
    else return TDC_E_BADARG;
    return (int)hipGetLastError();
}

extern "C" int tdc_qformer_xattn(const tdc_xattn_desc* d, void* stream) {
    const bool out_only = d && d->ctx != nullptr;
    if (!d || !d->h16 || (!d->h32 && !(out_only && d->res16)) || !d->wo || !d->bo || !d->ln_g || !d->ln_b) return TDC_E_BADARG;
    if (d->res16 && !out_only) return TDC_E_BADARG;
    if (!out_only && (!d->wq || !d->bq || !d->k || !d->vt)) return TDC_E_BADARG;
    if (!tdc_qformer_xattn_supported(d->dim, d->heads, d->K, out_only ? 8 : d->Nenc) || d->F <= 0 || d->S < d->K) {
        fprintf(stderr, "[tdc_hip] tdc_qformer_xattn: unsupported shape (dim=%d heads=%d K=%d Nenc=%d)\n", d->dim, d->heads, d->K,
                d->Nenc);
        return TDC_E_BADARG;
    }
    auto al = [](const void* p, int bytes) { return ((uintptr_t)p % bytes) == 0; };
    if (d->ldh % 8 || d->ldh < XD || !al(d->h16, 16) || (d->h32 && !al(d->h32, 16)) || !al(d->wo, 16) || !al(d->bo, 16) || !al(d->ln_g, 16) ||
        !al(d->ln_b, 16) || (long long)d->F * d->K > 0x7fffffffll || (long long)d->F * d->S * d->ldh > 0x7fffffffll)
        return TDC_E_BADARG;
    if (out_only) {
        if (d->ldctx % 8 || d->ldctx < XD || !al(d->ctx, 16)) return TDC_E_BADARG;
    } else if (d->ldk % 8 || d->ldk < XD || d->ldvt % 4 || d->ldvt < ((long long)d->F * d->Nenc + 7) / 8 * 8 || !al(d->wq, 16) || !al(d->k, 16) ||
               !al(d->vt, 8) || !al(d->bq, 16) || (d->bv && !al(d->bv, 16))) {
        return TDC_E_BADARG;
    }
    XattnArgs a;
    a.h16 = d->h16; a.h32 = d->h32; a.h16_out = d->h16; a.ldh = d->ldh;
    a.ctx = d->ctx; a.ldctx = d->ldctx; a.res16 = d->res16;
    a.rows = d->F * d->K; a.K = d->K; a.S = d->S;
    a.invK = (unsigned)((1ull << 32) / (unsigned long long)d->K);
    a.wq = d->wq; a.wo = d->wo; a.bq = d->bq; a.bo = d->bo;
    a.k = d->k; a.ldk = d->ldk; a.vt = d->vt; a.ldvt = d->ldvt; a.bv = d->bv; a.Nenc = d->Nenc;
    a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.eps = d->eps;
    a.scale_log2 = d->scale * 1.4426950408889634f;
    a.nblocks = (a.rows + XROWS - 1) / XROWS;
    const int nkt = (d->Nenc + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    const double rows = (double)d->F * d->K;
    TdcProfScope prof(TDC_PROF_XATTN, st, d->F * d->K, d->Nenc, 0, out_only ? 1 : 0, 0, 0, nullptr,
                      out_only ? 2.0 * rows * d->dim * d->dim : 4.0 * rows * d->dim * d->dim + 4.0 * rows * d->Nenc * d->dim);
    if (d->dtype == TDC_F16) return launch<f16>(a, nkt, st);
    if (d->dtype == TDC_BF16) return launch<bf16>(a, nkt, st);
    return TDC_E_BADARG;
}
