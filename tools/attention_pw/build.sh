#!/bin/bash
# Builds the one-wave-per-SIMD attention prototype into tools/attention_pw/libtdc_attn_pw.so (NOT part of libtdc_hip.so) and
# audits its generated code: the kernel's correctness rests on two things the compiler does not do for asm statements (every
# asm LDS read waited for before its use, asm MFMA destinations untouched for the hazard window) - tools/audit_asm_reads.py
# checks both on the --save-temps assembly and this script fails when it reports a violation.
set -e
cd "$(dirname "$0")"
R=../..
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result \
    -I $R/tdc-video_amd/csrc -I $R/include --save-temps=obj -c attention_pw.hip -o $T/attention_pw.o
python $R/tools/audit_asm_reads.py $T/attention_pw-hip-amdgcn-amd-amdhsa-gfx950.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtdc_attn_pw.so $T/attention_pw.o
rm -rf $T
echo "built tools/attention_pw/libtdc_attn_pw.so"
