// tdc_profile_*: see profile.h.  Host-only state (events + records) behind one mutex; off by default.
#include "../../include/tdc_hip.h"
#include "profile.h"
#include <mutex>
#include <vector>

std::atomic<int> tdc_prof_on{0};

namespace {
std::mutex g_mu;
std::vector<hipEvent_t> g_ev;        // 2 per record
std::vector<tdc_prof_rec> g_rec;
int g_n = 0, g_cap = 0, g_tag = 0, g_dropped = 0;   // g_dropped: launches that found the table full
}  // namespace

int tdc_prof_begin(int kind, hipStream_t st, int M, int N, int K, int act, int res, int out_f32, const void* W, double flops) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tdc_prof_on.load()) return -1;
    if (g_n >= g_cap) { ++g_dropped; return -1; }
    const int i = g_n++;
    tdc_prof_rec& r = g_rec[i];
    r.kind = kind; r.tag = g_tag; r.ms = 0.f; r.M = M; r.N = N; r.K = K; r.act = act; r.res = res; r.out_f32 = out_f32;
    r.W = W; r.flops = flops;
    if (hipEventRecord(g_ev[2 * i], st) != hipSuccess) { --g_n; return -1; }
    return i;
}

void tdc_prof_end(int idx, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx >= 0 && idx < g_n) (void)hipEventRecord(g_ev[2 * idx + 1], st);
}

extern "C" int tdc_profile_start(int max_records) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (tdc_prof_on.load() || max_records <= 0) return TDC_E_BADARG;
    while ((int)g_ev.size() < 2 * max_records) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return TDC_E_BADARG;
        g_ev.push_back(e);
    }
    g_rec.assign(max_records, tdc_prof_rec());
    g_cap = max_records; g_n = 0; g_tag = 0; g_dropped = 0;
    tdc_prof_on = 1;
    return 0;
}

extern "C" int tdc_profile_tag(int tag) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int old = g_tag;
    g_tag = tag;
    return old;
}

extern "C" int tdc_profile_stop(tdc_prof_rec* recs, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!tdc_prof_on.load()) return TDC_E_BADARG;
    tdc_prof_on = 0;
    const int n = g_n < cap ? g_n : cap;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_ev[2 * i + 1]) != hipSuccess) return TDC_E_BADARG;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess) return TDC_E_BADARG;
        g_rec[i].ms = ms;
        if (recs) recs[i] = g_rec[i];
    }
    const int total = g_n + g_dropped;
    g_n = 0; g_dropped = 0;
    return total;       // > cap: a table (the caller's, or the one sized by tdc_profile_start) was too small - launches were dropped
}
