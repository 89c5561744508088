// Launch profiler of the C ABI (tdc_profile_start / _stop / _tag, include/tdc_hip.h): while it is on, every leaf entry point
// (tdc_gemm, tdc_attention, tdc_layernorm, tdc_qformer_xattn) brackets its launch with two hipEvents on the launch stream and
// leaves a record - whether Python called it directly or a composite (tdc_vit_fwd, tdc_connector_fwd, tdc_qformer_fwd) did.
// bench.py's roofline therefore times the host path it also measures end to end.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

extern std::atomic<int> tdc_prof_on;     // 0 = off: the entry points test this one word (relaxed) and do nothing else
// returns a record index (>= 0) or -1 (off / table full); `end` takes that index
int tdc_prof_begin(int kind, hipStream_t st, int M, int N, int K, int act, int res, int out_f32, const void* W, double flops);
void tdc_prof_end(int idx, hipStream_t st);

struct TdcProfScope {
    int idx; hipStream_t st;
    TdcProfScope(int kind, hipStream_t s, int M, int N, int K, int act, int res, int out_f32, const void* W, double flops)
        : idx(tdc_prof_on.load(std::memory_order_relaxed) ? tdc_prof_begin(kind, s, M, N, K, act, res, out_f32, W, flops) : -1), st(s) {}
    ~TdcProfScope() { if (idx >= 0) tdc_prof_end(idx, st); }
};

// tag of the records that follow, for the life of the guard; the previous tag comes back on every way out of the scope (an early
// error return included), so a tagged region inside a composite never clobbers a tag its caller set
extern "C" int tdc_profile_tag(int tag);
struct TdcProfTagGuard {      // tag < 0: leaves the current tag alone
    bool active; int old;
    explicit TdcProfTagGuard(int tag) : active(tag >= 0), old(tag >= 0 ? tdc_profile_tag(tag) : 0) {}
    ~TdcProfTagGuard() { if (active) tdc_profile_tag(old); }
    TdcProfTagGuard(const TdcProfTagGuard&) = delete;
    TdcProfTagGuard& operator=(const TdcProfTagGuard&) = delete;
};
