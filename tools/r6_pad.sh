#!/bin/bash
# experiment: k_tail padded by N us (tools/_ab/libbnr_pad.so, -DBNR_EXP_PAD): does WHEN the scalar branch's later kernels start matter for the sweep?
R=$GRAFT_REPO_ROOT
{
for r in 1 2; do
for pad in 0 10 20 30 45 60; do
  echo -n "pad $pad: "; BNR_HIP_LIB=$R/tools/_ab/libbnr_pad.so BNR_EXP_TAIL_PAD_US=$pad tools/quick_bench.sh
done
echo -n "base: "; BNR_HIP_LIB=$R/tools/_ab/libbnr_base.so tools/quick_bench.sh
done
} > gpurun_out/r6_pad.log 2>&1
cat gpurun_out/r6_pad.log
