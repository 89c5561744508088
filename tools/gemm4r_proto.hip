// Prototype 2 (round-1 experiment, not part of the library): REGISTER-STAGED operand loads.  As gemm4w_proto.hip (bf16
// GEMM C = A W^T, 256x256 tile computed by FOUR waves
// (2 x 2, 128x128 per wave, all 256 accumulator registers + one wave per SIMD, 512 VGPRs), persistent over tiles.
// Question it answers: does a one-wave-per-SIMD main loop (2/3 of the LDS read traffic of the 8-wave kernel, one
// barrier per k-step among 4 waves) reach at least the 8-wave kernel's main-loop rate?  If so, its spare registers can
// park a finished 16-bit tile and drain it under the next tile's MFMAs (the 8-wave kernel cannot: 226+ VGPRs of 256).
//
// Pipeline: ring of 4 LDS stages x 32 KiB (A 256x32 + W 256x32 bf16, 64-byte rows, 16-B chunk c of row r stored at
// c ^ ((r >> 2) & 3) -> conflict-free ds_read_b128), one k-step (K = 32 = one MFMA 16x16x32 depth) per stage.
// k-step s: vmcnt(8) + barrier (stage s+1 visible, stage s-1 free) | 8 LDS-DMA loads of stage s+3 | prefetch reads of
// stage s+1 (A fragments of all 8 row blocks, W fragments 3 columns ahead) | 64 MFMAs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gpurun_out/gemm4w tools/gemm4w_proto.hip && gpurun_out/gemm4w M N K
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// acc (AGPRs, in place) += a (W rows -> output columns) x b (activation rows)
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

struct Args {
    const bf16* A; const bf16* W; bf16* C;
    int M, N, K, lda, ldw, ldc;
    int tiles_m, tiles_n;
    int mode;   // 0 = store C, 1 = skip the epilogue (main loop timing)
};

constexpr int STAGE = 32768, NSTAGE = 4;
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_group = GROUP_M * tiles_n;
    const int grp = id / per_group, within = id - grp * per_group;
    const int first = grp * GROUP_M;
    const int rows = (tiles_m - first < GROUP_M) ? tiles_m - first : GROUP_M;
    tn = within / rows;
    tm = first + (within - tn * rows);
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int G8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int cq = nwg >> 3, cr = nwg & 7;
    const int chunk_base = (xcd < cr) ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_len = cq + (xcd < cr ? 1 : 0);
    const int n_my = l < chunk_len ? (chunk_len - l + G8 - 1) / G8 : 0;
    if (n_my == 0) return;
    const int nks = p.K / 32;                    // k-steps per tile (>= 4)
    const int total = n_my * nks;                // k-steps of this workgroup

    // ---- staging: instruction q (0..3) of this wave covers rows (wave*4+q)*16 .. +15 of the A and of the W stage image
    unsigned a_so[4], w_so[4];
    {
        const int srcchunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 16 + (lane >> 2);
            a_so[q] = (unsigned)(r * p.lda + srcchunk * 8) * 2u;
            w_so[q] = (unsigned)(r * p.ldw + srcchunk * 8) * 2u;
        }
    }
    // load cursor: tile and k-step being LOADED (global -> registers; runs 4 k-steps ahead of the MFMAs, across tile seams).
    // Stage X is loaded during k-step X-4 into register set X % 3, written to LDS slot X & 3 during k-step X-2 and read
    // (fragment prefetch) during k-steps X-1 and X: the LDS-DMA pieces of gemm4w_proto cost ~60 issue cycles each with
    // no partner wave to cover them; a global_load_dwordx4 + ds_write_b128 pair issues in a fraction of that.
    int s_it = 0, s_ks = 0;
    const char *s_a, *s_w;
    auto stage_tile = [&](int it) {
        int tm, tn;
        tile_coords(chunk_base + l + it * G8, p.tiles_m, p.tiles_n, tm, tn);
        s_a = (const char*)p.A + (long long)tm * 256 * p.lda * 2;
        s_w = (const char*)p.W + (long long)tn * 256 * p.ldw * 2;
    };
    stage_tile(0);
    u32x4 rg[3][8];
    auto load_piece = [&](auto rs_c, int q) {    // q = 0..7: pieces 0-3 A, 4-7 W of the cursor's k-step
        constexpr int RS = decltype(rs_c)::value;
        if (q < 4) rg[RS][q] = *(const u32x4*)(s_a + a_so[q & 3]);
        else rg[RS][q] = *(const u32x4*)(s_w + w_so[q & 3]);
    };
    auto load_advance = [&]() {
        s_a += 64; s_w += 64;
        if (++s_ks == nks) {
            s_ks = 0;
            if (++s_it < n_my) stage_tile(s_it);
        }
    };
    auto write_piece = [&](auto rs_c, int q, int slot_) {
        constexpr int RS = decltype(rs_c)::value;
        char* dst = smem + slot_ * STAGE + (q >> 2) * 16384 + (wave * 4 + (q & 3)) * 1024 + lane * 16;
        *(u32x4*)dst = rg[RS][q];
    };

    // ---- fragment read offsets: row block i adds i*1024 bytes (immediate)
    const int fr = lane & 15, g = lane >> 4;
    const int frag_off = fr * 64 + ((g ^ (fr >> 2)) & 3) * 16;
    const int a_rd = wm * 128 * 64 + frag_off;
    const int w_rd = 16384 + wn * 128 * 64 + frag_off;

    f32x4 acc[8][8];
    bf16x8 xa[2][8], xw[4];

    // ---- prologue: stages 0, 1 in LDS and visible, stage 2 in set 2, stage 3 in set 0 (in flight)
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
#pragma unroll
    for (int q = 0; q < 8; ++q) load_piece(I0(), q);
    load_advance();
#pragma unroll
    for (int q = 0; q < 8; ++q) load_piece(I1(), q);
    load_advance();
#pragma unroll
    for (int q = 0; q < 8; ++q) load_piece(I2(), q);
    load_advance();
#pragma unroll
    for (int q = 0; q < 8; ++q) { write_piece(I0(), q, 0); write_piece(I1(), q, 1); }
#pragma unroll
    for (int q = 0; q < 8; ++q) load_piece(I0(), q);
    load_advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const char* b = smem;
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[0][i] = *(const bf16x8*)(b + a_rd + i * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) xw[j] = *(const bf16x8*)(b + w_rd + j * 1024);
    }

    int slot = 0;                                // LDS slot of the k-step being computed
    int S = 0;                                   // k-steps done by this workgroup
    // one k-step; CUR = which xa buffer holds its A fragments, WS = register set of stage S+2 (written to LDS here),
    // LS = register set that receives stage S+4
    auto kstep = [&](auto cur_c, auto ws_c, auto ls_c) {
        constexpr int CUR = decltype(cur_c)::value;
        // this wave's loads of stage S+2 have landed (stage S+3's may still fly); its LDS writes of stage S+1 (first
        // half of the previous k-step) are older than the 4 fragment reads that followed them -> barrier: stage S+1
        // visible to all, every wave done with stage S-1
        if (S + 3 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* b0 = smem + slot * STAGE;                    // stage S   (this k-step's W columns 3..7)
        const char* b1 = smem + ((slot + 1) & 3) * STAGE;        // stage S+1 (next k-step's fragments)
        const int wslot = (slot + 2) & 3;
        const bool do_write = S + 2 < total, do_load = S + 4 < total;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            MFMA(acc[0][j], xw[j & 3], xa[CUR][0]);
            if (j < 5) xw[(j + 3) & 3] = *(const bf16x8*)(b0 + w_rd + (j + 3) * 1024);
            else xw[(j + 3) & 3] = *(const bf16x8*)(b1 + w_rd + (j + 3 - 8) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            MFMA(acc[1][j], xw[j & 3], xa[CUR][1]);
            if (j < 4) xa[CUR ^ 1][2 * j] = *(const bf16x8*)(b1 + a_rd + (2 * j) * 1024);
            else if (do_load) load_piece(ls_c, 2 * (j - 4));
            __builtin_amdgcn_sched_barrier(0);
            MFMA(acc[2][j], xw[j & 3], xa[CUR][2]);
            if (j < 4) xa[CUR ^ 1][2 * j + 1] = *(const bf16x8*)(b1 + a_rd + (2 * j + 1) * 1024);
            else if (do_load) load_piece(ls_c, 2 * (j - 4) + 1);
            __builtin_amdgcn_sched_barrier(0);
            MFMA(acc[3][j], xw[j & 3], xa[CUR][3]);
            if (j < 4 && do_write) write_piece(ws_c, 2 * j, wslot);
            __builtin_amdgcn_sched_barrier(0);
            MFMA(acc[4][j], xw[j & 3], xa[CUR][4]);
            if (j < 4 && do_write) write_piece(ws_c, 2 * j + 1, wslot);
            __builtin_amdgcn_sched_barrier(0);
            MFMA(acc[5][j], xw[j & 3], xa[CUR][5]);
            MFMA(acc[6][j], xw[j & 3], xa[CUR][6]);
            MFMA(acc[7][j], xw[j & 3], xa[CUR][7]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_load) load_advance();
        slot = (slot + 1) & 3;
        ++S;
    };

    for (int it = 0; it < n_my; ++it) {
        int tm, tn;
        tile_coords(chunk_base + l + it * G8, p.tiles_m, p.tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // CUR alternates 0/1; stage S+2 lives in set (S+2) % 3, stage S+4 goes to set (S+1) % 3: period 6 (nks % 6 == 0,
        // so every tile starts at S % 6 == 0)
        for (int ks = 0; ks < nks; ks += 6) {
            kstep(I0(), I2(), I1());     // S % 6 == 0
            kstep(I1(), I0(), I2());     // 1
            kstep(I0(), I1(), I0());     // 2
            kstep(I1(), I2(), I1());     // 3
            kstep(I0(), I0(), I2());     // 4
            kstep(I1(), I1(), I0());     // 5
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> accvgpr reads (the compiler cannot see asm MFMAs)
        // ---- epilogue (prototype: plain row-per-lane stores)
        if (MODE == 0) {
            int elane = lane;
            asm volatile("" : "+v"(elane));      // keep the store addresses out of the main loop's live ranges
            const int efr = elane & 15, eg = elane >> 4;
            const int m0 = tm * 256 + wm * 128, n0 = tn * 256 + wn * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16* crow = p.C + (long long)(m0 + i * 16 + efr) * p.ldc + n0 + eg * 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][e];
                    *(bf16x4*)(crow + j * 16) = o;
                }
                __builtin_amdgcn_sched_barrier(0);   // one row block at a time: 32 accumulator reads live, not 256
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"a"(acc[i][j]));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // compiler-visible: epilogue stores done before the fragment registers are reused
    }
}

// ---------------------------------------------------------------------------------------------------------------------
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void fill(std::vector<unsigned short>& v, unsigned seed, float scale) {   // same data as tools/gemm_stamps.cpp
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < v.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = f2bf(((int)(s >> 9) % 2001 - 1000) * 0.001f * scale);
    }
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 186624, N = argc > 2 ? atoi(argv[2]) : 3456, K = argc > 3 ? atoi(argv[3]) : 1152;
    int reps = argc > 4 ? atoi(argv[4]) : 20;
    if (M % 256 || N % 256 || K % 192) { fprintf(stderr, "M, N %% 256, K %% 192\n"); return 2; }
    size_t nA = (size_t)M * K, nW = (size_t)N * K, nC = (size_t)M * N;
    std::vector<unsigned short> hA(nA), hW(nW), hC(nC);
    fill(hA, M + K, 1.0f); fill(hW, N + K, 0.05f);
    void *A, *W, *C;
    CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&W, nW * 2)); CK(hipMalloc(&C, nC * 2));
    CK(hipMemcpy(A, hA.data(), nA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), nW * 2, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, nC * 2));
    Args a;
    a.A = (const bf16*)A; a.W = (const bf16*)W; a.C = (bf16*)C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N;
    a.tiles_m = M / 256; a.tiles_n = N / 256;
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int G = (prop.multiProcessorCount / 8) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        a.mode = mode;
        auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(G), dim3(256), NSTAGE * STAGE, 0, a); else hipLaunchKernelGGL(gemm4w_kernel<1>, dim3(G), dim3(256), NSTAGE * STAGE, 0, a); };
        for (int r = 0; r < 3; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("gemm4r M=%d N=%d K=%d %s: %.3f ms  %.1f TF/s\n", M, N, K, mode ? "no-epilogue" : "with plain epilogue", ms,
               2.0 * M * N * K / ms / 1e9);
        if (mode == 0) {
            CK(hipMemcpy(hC.data(), C, nC * 2, hipMemcpyDeviceToHost));
            // spot check 4096 outputs against a double-precision dot product
            double maxerr = 0; unsigned s = 777;
            for (int t = 0; t < 4096; ++t) {
                s = s * 1664525u + 1013904223u; int m = (s >> 8) % M;
                s = s * 1664525u + 1013904223u; int n = (s >> 8) % N;
                if (t < 8) { m = t & 1 ? M - 1 - t : t * 37 % M; n = t & 2 ? N - 1 - t : t * 91 % N; }
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
                double err = fabs(ref - bf2f(hC[(size_t)m * N + n])) / (fabs(ref) + 1.0);
                if (err > maxerr) maxerr = err;
            }
            printf("  spot check: max rel err %.3e %s\n", maxerr, maxerr < 1e-2 ? "OK" : "MISMATCH");
        }
    }
    return 0;
}
