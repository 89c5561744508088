// Diagnostics build of tdc_gemm with in-kernel timeline stamps (s_memrealtime of wave 0 at: entry, end of prologue, end
// of the main loop, epilogue stores issued, [stores acknowledged]) + the CU each workgroup ran on.  Prints where a tile's
// time goes and how long a CU sits between two workgroups.  The shipped library never contains a stamp.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDC_GEMM_DIAG -o gpurun_out/gemm_stamps tools/gemm_stamps.cpp
//   gpurun_out/gemm_stamps M N K act res outf32 [reps]
#include "../tdc-video_amd/csrc/gemm.hip"
#include <algorithm>
#include <map>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static void fill_bf16(std::vector<unsigned short>& v, unsigned seed, float scale) {
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < v.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        float f = ((int)(s >> 9) % 2001 - 1000) * 0.001f * scale;
        unsigned u; memcpy(&u, &f, 4);
        v[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
}

int main(int argc, char** argv) {
    if (const char* e = getenv("TDC_GEMM_DEBUG")) tdc_gemm_set_debug(atoi(e));   // the tool, not the library, reads the environment
    if (argc < 7) { fprintf(stderr, "usage: gemm_stamps M N K act res outf32 [reps]\n"); return 2; }
    int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), act = atoi(argv[4]), res = atoi(argv[5]), outf32 = atoi(argv[6]);
    int reps = argc > 7 ? atoi(argv[7]) : 3;
    size_t nA = (size_t)M * K, nW = (size_t)N * K, nC = (size_t)M * N;
    std::vector<unsigned short> hA(nA), hW(nW);
    fill_bf16(hA, M + K, 1.0f); fill_bf16(hW, N + K, 0.05f);
    void *A, *W, *C; float* bias;
    CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&W, nW * 2)); CK(hipMalloc(&C, nC * (outf32 ? 4 : 2))); CK(hipMalloc((void**)&bias, N * 4));
    CK(hipMemcpy(A, hA.data(), nA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), nW * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(C, 0, nC * (outf32 ? 4 : 2)));
    const int nwg = ((M + 255) / 256) * ((N + 255) / 256);
    unsigned long long* st;
    CK(hipMalloc((void**)&st, (size_t)nwg * 64));
    CK(hipMemset(st, 0, (size_t)nwg * 64));
    tdc_gemm_diag_stamps = st;
    unsigned long long* wst;
    CK(hipMalloc((void**)&wst, (size_t)nwg * 8 * 4 * 8));
    CK(hipMemset(wst, 0, (size_t)nwg * 8 * 4 * 8));
    tdc_gemm_diag_wstamps = wst;
    tdc_gemm_desc d = {};
    d.A = A; d.lda = K; d.W = W; d.ldw = K; d.C = C; d.ldc = (act == TDC_ACT_SWIGLU) ? N / 2 : N; d.bias = bias;
    d.M = M; d.N = N; d.K = K; d.dtype = TDC_BF16; d.out_f32 = outf32; d.act = act;
    if (res) { d.res = C; d.ldres = N; d.res_f32 = outf32; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        int rc = tdc_gemm(&d, 0); if (rc) { fprintf(stderr, "tdc_gemm rc=%d\n", rc); return 1; }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h((size_t)nwg * 8);
    CK(hipMemcpy(h.data(), st, (size_t)nwg * 64, hipMemcpyDeviceToHost));
    // group by CU
    std::map<unsigned, std::vector<int>> by_cu;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nwg; ++b) {
        const unsigned long long* s = &h[(size_t)b * 8];
        unsigned hw = (unsigned)s[5], xcc = (unsigned)s[6];
        unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        tmin = std::min(tmin, s[0]); tmax = std::max(tmax, std::max(s[4], s[2]));
    }
    double sum[5] = {0, 0, 0, 0, 0}; long cnt = 0, gaps = 0; double gap_sum = 0, gap_max = 0;
    double first_start_max = 0;
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return h[(size_t)a * 8] < h[(size_t)b * 8]; });
        first_start_max = std::max(first_start_max, (double)(h[(size_t)v[0] * 8] - tmin));
        for (size_t i = 0; i < v.size(); ++i) {
            const unsigned long long* s = &h[(size_t)v[i] * 8];
            sum[0] += s[1] - s[0]; sum[1] += s[2] - s[1];
            if (s[3]) { sum[2] += s[3] - s[2]; sum[3] += s[4] - s[3]; }
            ++cnt;
            if (i + 1 < v.size()) {
                unsigned long long end = s[4] ? s[4] : s[2];
                double g = (double)h[(size_t)v[i + 1] * 8] - (double)end;
                gap_sum += g; gap_max = std::max(gap_max, g); ++gaps;
            }
        }
    }
    const double u = 0.01;  // 100 MHz ticks -> us
    printf("M=%d N=%d K=%d act=%d res=%d outf32=%d: %.3f ms (%.1f TF/s), %d workgroups on %zu CUs, kernel span by stamps %.1f us\n",
           M, N, K, act, res, outf32, ms, 2.0 * M * N * K / ms / 1e9, nwg, by_cu.size(), (tmax - tmin) * u);
    printf("  per workgroup (wave 0), us: prologue %.2f | main loop %.2f | epilogue issue %.2f | store ack wait %.2f | "
           "CU gap to next workgroup avg %.2f max %.2f | first-start skew max %.2f\n",
           sum[0] / cnt * u, sum[1] / cnt * u, sum[2] / cnt * u, sum[3] / cnt * u, gaps ? gap_sum / gaps * u : 0.0,
           gap_max * u, first_start_max * u);
    {   // per-wave epilogue stamps (persistent kernel only): duration per wave, skew of the waves' starts, tile-level span
        std::vector<unsigned long long> w((size_t)nwg * 32);
        CK(hipMemcpy(w.data(), wst, w.size() * 8, hipMemcpyDeviceToHost));
        double d_issue = 0, d_ack = 0, skew = 0, span = 0; long n_w = 0, n_t = 0;
        for (int b = 0; b < nwg; ++b) {
            unsigned long long lo = ~0ull, hi0 = 0, hi2 = 0;
            for (int wv = 0; wv < 8; ++wv) {
                const unsigned long long* x = &w[((size_t)b * 8 + wv) * 4];
                if (!x[0] || !x[2]) continue;
                d_issue += x[1] - x[0]; d_ack += x[2] - x[1]; ++n_w;
                lo = std::min(lo, x[0]); hi0 = std::max(hi0, x[0]); hi2 = std::max(hi2, x[2]);
            }
            if (hi2) { skew += hi0 - lo; span += hi2 - lo; ++n_t; }
        }
        if (n_w) printf("  epilogue per WAVE, us: issue %.2f | ack wait %.2f | start skew inside a workgroup %.2f | first start -> last ack "
                        "%.2f (tiles %ld)\n", d_issue / n_w * u, d_ack / n_w * u, skew / n_t * u, span / n_t * u, n_t);
    }
    // timeline of one CU
    auto& v0 = by_cu.begin()->second;
    printf("  CU %x:", by_cu.begin()->first);
    for (size_t i = 0; i < v0.size() && i < 6; ++i) {
        const unsigned long long* s = &h[(size_t)v0[i] * 8];
        printf(" [wg %d: %.1f %.1f %.1f %.1f %.1f]", v0[i], (s[0] - tmin) * u, (s[1] - tmin) * u, (s[2] - tmin) * u,
               s[3] ? (s[3] - tmin) * u : 0.0, s[4] ? (s[4] - tmin) * u : 0.0);
    }
    printf("\n");
    return 0;
}
