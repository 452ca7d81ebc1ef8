#!/bin/bash
# quick try of a build: a few parity tests, then timings interleaved with the round-5 base
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_sweep_matches_oracle or deconstructed" 2>&1 | tail -5
for r in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export BNR_HIP_LIB=$R/tools/_ab/libbnr_base.so; else unset BNR_HIP_LIB; fi
    echo -n "$v: "; tools/quick_bench.sh
  done
done
unset BNR_HIP_LIB
} > gpurun_out/r6_try.log 2>&1
cat gpurun_out/r6_try.log
