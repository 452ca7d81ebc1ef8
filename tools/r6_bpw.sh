#!/bin/bash
# the back-projection with a register cap (more workgroups resident): 8 chains and one chain at the headline shape, interleaved with the uncapped build
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for v in new bp4 bp5; do
  if [ $v = new ]; then unset BNR_HIP_LIB; else export BNR_HIP_LIB=$R/tools/_ab/libbnr_$v.so; fi
  timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_$v.txt
done
unset BNR_HIP_LIB
diff gpurun_out/dig_new.txt gpurun_out/dig_bp4.txt && diff gpurun_out/dig_new.txt gpurun_out/dig_bp5.txt && echo "DIGESTS EQUAL"
for r in 1 2 3; do
  for v in new bp4 bp5; do
    if [ $v = new ]; then unset BNR_HIP_LIB; else export BNR_HIP_LIB=$R/tools/_ab/libbnr_$v.so; fi
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 4 640 500 100 7 -- default | tail -1
  done
done
} > gpurun_out/r6_bpw.log 2>&1
cat gpurun_out/r6_bpw.log
