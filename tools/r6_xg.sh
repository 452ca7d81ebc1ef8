#!/bin/bash
# k_xpass_group with its multipliers by DPP: digests against the last commit's build, the kernel's time under rocprofv3, interleaved sweep timings (headline group, Bool headline, cfg2 x 8)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_head.txt
timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_new.txt
diff gpurun_out/dig_head.txt gpurun_out/dig_new.txt && echo "DIGESTS EQUAL"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group or variants" 2>&1 | tail -2
for r in 1 2 3; do
  for v in head new; do
    if [ $v = head ]; then export BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so; else unset BNR_HIP_LIB; fi
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 4 640 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 8 640 200 50 5 -- default | tail -1
  done
done
unset BNR_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for v in head new; do
  if [ $v = head ]; then export BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so; else unset BNR_HIP_LIB; fi
  rm -rf $R/gpurun_out/prof_xg_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_xg_$v -o xg -- python3 $R/bench.py --steps 240 --warmup 24 --no-cpu-baseline > $R/gpurun_out/prof_xg_$v.log 2>&1
  echo "== $v"; python3 $R/tools/prof_summary.py $R/gpurun_out/prof_xg_$v | grep "xpass_group\|k_node<bnr_many\|k_tail<bnr_many"
  rm -rf $R/gpurun_out/prof_xg_$v/*/
done
} > gpurun_out/r6_xg.log 2>&1
cat gpurun_out/r6_xg.log
