#!/bin/bash
# the built library against tools/_ab/libbnr_head.so (the last commit's build), interleaved: one chain and the group of 8 at the headline shape
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for r in 1 2; do
  for v in head new; do
    if [ $v = head ]; then export BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so; else unset BNR_HIP_LIB; fi
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- default | tail -1
  done
done
} > gpurun_out/r6_headab.log 2>&1
cat gpurun_out/r6_headab.log
