#!/bin/bash
# issue priority of the factorization's update workgroups below the panel workgroups' (s_setprio 0 / 1 / 2 against everybody at 3): interleaved timings
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for r in 1 2 3; do
  for v in new up0 up1 up2; do
    if [ $v = new ]; then unset BNR_HIP_LIB; else export BNR_HIP_LIB=$R/tools/_ab/libbnr_$v.so; fi
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- default | tail -1
  done
done
} > gpurun_out/r6_prio.log 2>&1
cat gpurun_out/r6_prio.log
