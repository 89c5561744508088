#!/bin/bash
# Same-box A/B of two builds of libtdc_hip.so: runs "$@" once with tools/ab/libtdc_hip_base.so (or $TDC_AB_BASE; a build of an earlier commit or of a variant,
# copied there by hand: *.so files travel with gpurun but stay out of git) in place of the tree's library, once with the tree's
# own.  Output of the two runs: gpurun_out/ab_base.log / gpurun_out/ab_new.log.  Run from the repo root on the GPU box.
set -e
mkdir -p gpurun_out
cp tdc-video_amd/libtdc_hip.so /tmp/libtdc_hip_new.so
restore() { cp /tmp/libtdc_hip_new.so tdc-video_amd/libtdc_hip.so; }
trap restore EXIT
for round in 1 2; do
cp ${TDC_AB_BASE:-tools/ab/libtdc_hip_base.so} tdc-video_amd/libtdc_hip.so
echo "== base (round $round)" >> gpurun_out/ab_base.log; "$@" >> gpurun_out/ab_base.log 2>&1
cp /tmp/libtdc_hip_new.so tdc-video_amd/libtdc_hip.so
echo "== new (round $round)" >> gpurun_out/ab_new.log; "$@" >> gpurun_out/ab_new.log 2>&1
done
