// tdc_gemm with OCP e4m3 operands (tdc_gemm_desc.in_fp8): the FP8 instantiations of the kernels in gemm_impl.h - the same
// tiles, staging, LDS image and epilogues as the 16-bit ones, one v_mfma_f32_16x16x128_f8f6f4 per pair of 16-byte
// fragments (common.h: mma128_fp8).  Validation happens in tdc_gemm (gemm.hip), which forwards here.
#define TDC_GEMM_FP8_TU 1
#include "gemm_impl.h"

int tdc_gemm_fp8_impl(const tdc_gemm_desc* d, hipStream_t st) {
    if (d->dtype == TDC_F16) return launch<f16, true>(d, st);
    if (d->dtype == TDC_BF16) return launch<bf16, true>(d, st);
    return TDC_E_BADARG;
}
