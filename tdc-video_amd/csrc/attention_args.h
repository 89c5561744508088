// Launch arguments shared by the two attention translation units (attention.hip: 16x16x32 MFMA form, every shape;
// attention32.hip: 32x32x16 form for the long sequences of the ViT towers).
#pragma once
#include <hip/hip_runtime.h>

struct AttnArgs {
    const void *q, *k, *v; void* o;
    long long q_bs, k_bs, v_bs, o_bs;
    int q_rs, k_rs, v_rs, o_rs;
    int heads, d, sq, sk;
    float scale_log2;
    int vec_ok;
    // additive score bias (BEATs gated relative position bias): score(b,h,q,k) += gate[(b*sq+q)*gate_rs + h] *
    // bias[h*bias_hs + q*bias_rs + k]; both fp32, bias rows 16-byte aligned (sk % 4 == 0)
    const float* bias; long long bias_hs; int bias_rs;
    const float* gate; int gate_rs;
    // key padding mask of the biased form: key k of batch b scores -inf when kmask[b*kmask_bs + k] != 0 (NULL = none)
    const unsigned char* kmask; long long kmask_bs;
};

// attention32.hip: returns -1 when the 32x32 form does not apply (bias, short sequences, other head dims)
int tdc_attention32(const AttnArgs& a, int batch, int dtype, hipStream_t st);
