// Launch profiler of the C ABI (tdc_profile_start / _stop / _tag, include/tdc_hip.h): while it is on, every leaf entry point
// (tdc_gemm, tdc_attention, tdc_layernorm, tdc_qformer_xattn) brackets its launch with two hipEvents on the launch stream and
// leaves a record - whether Python called it directly or a composite (tdc_vit_fwd, tdc_connector_fwd, tdc_qformer_fwd) did.
// bench.py's roofline therefore times the host path it also measures end to end.
#pragma once
#include <hip/hip_runtime.h>

extern int tdc_prof_on;     // 0 = off: the entry points test this one word and do nothing else
// returns a record index (>= 0) or -1 (off / table full); `end` takes that index
int tdc_prof_begin(int kind, hipStream_t st, int M, int N, int K, int act, int res, int out_f32, const void* W, double flops);
void tdc_prof_end(int idx, hipStream_t st);

struct TdcProfScope {
    int idx; hipStream_t st;
    TdcProfScope(int kind, hipStream_t s, int M, int N, int K, int act, int res, int out_f32, const void* W, double flops)
        : idx(tdc_prof_on ? tdc_prof_begin(kind, s, M, N, K, act, res, out_f32, W, flops) : -1), st(s) {}
    ~TdcProfScope() { if (idx >= 0) tdc_prof_end(idx, st); }
};
