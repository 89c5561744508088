// What bounds the C-tile drain of the GEMM epilogue: the CU's own store path, or a resource the CUs share?
// Every workgroup (512 threads, one per CU: 160 KiB of dynamic LDS) writes `tiles` C tiles of 256 x 256 16-bit values
// (row stride `ldc` elements) exactly as the staged epilogue does - one wave instruction = 8 rows x 128 B, 16 B per lane,
// 16 instructions per wave and tile - and stamps s_memrealtime around each tile (stores acknowledged: vmcnt(0)).
//   store_bw G tiles mode [mfma]      G = workgroups (8..256), mode 0 = nontemporal, 1 = plain, 2 = sc1 (write-through)
//                                     mfma = 1: waves 4-7 run an MFMA loop instead of storing (waves 0-3 store the whole tile)
// Prints the median / min / max time per tile and the bytes per clock (100 MHz realtime counter; the shader clock is not
// needed for the comparison across G).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
    if (MODE == 0) __builtin_nontemporal_store(v, p);
    else if (MODE == 1) *p = v;
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// spacer: every wave runs `spacer` x 64 MFMAs (~ spacer x 0.5 us) before each tile's drain (the "main loop" between two
// drains); stagger_ns: workgroup l of its XCD (blockIdx.x >> 3) starts (l >> 3) * stagger_ns late (tile-column groups of the
// GEMM's 8 x 4 patch per XCD); one_xcd >= 0: only workgroups running on that XCC_ID take part.
template <int MODE, bool MFMA>
__global__ __launch_bounds__(512, 2) void k(unsigned short* C, long long ldc, int tiles, int tiles_n, unsigned long long* stamps,
                                            float* sink, int spacer, int stagger_ns, int one_xcd) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (one_xcd >= 0 && (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) != one_xcd) return;
    if (stagger_ns > 0) {
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)((blockIdx.x >> 3) >> 3) * stagger_ns / 10;
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(4);
    }
    u32x4 v = {threadIdx.x, blockIdx.x, 0x3f803f80u, 0x40004000u};
    f32x4 acc[8];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < tiles; ++t) {
        const int id = blockIdx.x + gridDim.x * t;
        const int tm = id / tiles_n, tn = id % tiles_n;
        unsigned short* tile = C + (long long)tm * 256 * ldc + tn * 256;
        for (int it = 0; it < spacer * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        __syncthreads();
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (!MFMA || wave < 4) {
            // wave sub-tile: 128 rows x 64 columns (128 B per row) as in the kernel; with MFMA only 4 waves store, 2 sub-tiles each
            const int nsub = MFMA ? 2 : 1;
            for (int s = 0; s < nsub; ++s) {
                const int w = MFMA ? wave * 2 + s : wave;
                const int wm = (w >> 1) & 1, wn = (w & 1) | ((w >> 2) << 1);
                unsigned short* sub = tile + (long long)wm * 128 * ldc + wn * 64;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int r = q * 8 + (lane >> 3), kk = lane & 7;
                    st16<MODE>((u32x4*)(sub + (long long)r * ldc + kk * 8), v);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // ~ the MFMA work of a K = 1152 tile's share for this wave: 18 K tiles x 64 MFMAs
            for (int it = 0; it < 18 * 8; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            stamps[((size_t)blockIdx.x * tiles + t) * 16 + wave * 2] = t1 - t0;
            if (wave == 0) stamps[((size_t)blockIdx.x * tiles + t) * 16 + 1] = t2 - t0;
        }
    }
    {
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 123.456f) sink[0] = s;
    }
}

int main(int argc, char** argv) {
    int G = argc > 1 ? atoi(argv[1]) : 256, tiles = argc > 2 ? atoi(argv[2]) : 8, mode = argc > 3 ? atoi(argv[3]) : 0;
    int mfma = argc > 4 ? atoi(argv[4]) : 0;
    int spacer = argc > 5 ? atoi(argv[5]) : 0, stagger_ns = argc > 6 ? atoi(argv[6]) : 0, one_xcd = argc > 7 ? atoi(argv[7]) : -1;
    const int N = 3456, tiles_n = 13;                      // 13 whole column tiles of the SigLIP qkv output
    const long long ldc = N;
    const int tiles_m = (G * tiles + tiles_n - 1) / tiles_n + 1;
    unsigned short* C; unsigned long long* st; float* sink;
    CK(hipMalloc(&C, (size_t)tiles_m * 256 * ldc * 2));
    CK(hipMalloc(&st, (size_t)G * tiles * 16 * 8));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(st, 0, (size_t)G * tiles * 16 * 8));
    auto run = [&](auto kern) -> int {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(kern, dim3(G), dim3(512), 160 * 1024, 0, C, ldc, tiles, tiles_n, st, sink, spacer, stagger_ns, one_xcd);
            CK(hipDeviceSynchronize());
        }
        return 0;
    };
    int rc = 0;
    if (mfma) rc = mode == 0 ? run(k<0, true>) : mode == 1 ? run(k<1, true>) : run(k<2, true>);
    else rc = mode == 0 ? run(k<0, false>) : mode == 1 ? run(k<1, false>) : run(k<2, false>);
    if (rc) return rc;
    std::vector<unsigned long long> h((size_t)G * tiles * 16);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> wg, wv;
    for (int b = 0; b < G; ++b)
        for (int t = 1; t < tiles; ++t) {          // skip the first tile (cold)
            const unsigned long long* s = &h[((size_t)b * tiles + t) * 16];
            if (s[1] == 0) continue;               // workgroup did not take part (one_xcd)
            wg.push_back((double)s[1] * 0.01);     // us (100 MHz)
            for (int w = 0; w < (mfma ? 4 : 8); ++w) wv.push_back((double)s[w * 2] * 0.01);
        }
    std::sort(wg.begin(), wg.end()); std::sort(wv.begin(), wv.end());
    if (wg.empty()) { printf("no workgroup took part\n"); return 0; }
    const double med = wg[wg.size() / 2];
    printf("spacer=%d stagger=%dns one_xcd=%d n=%zu | ", spacer, stagger_ns, one_xcd, wg.size());
    printf("G=%3d mode=%d mfma=%d  tile drain (workgroup, barrier to barrier): median %.2f us  min %.2f  max %.2f | per storing wave "
           "median %.2f us | %.1f GB/s per CU, %.2f TB/s over %d CUs\n", G, mode, mfma, med, wg.front(), wg.back(), wv[wv.size() / 2],
           131072.0 / med * 1e-3, 131072.0 / med * 1e-6 * G, G);
    return 0;
}
