#!/bin/bash
# bnr_tail_a with u staged in LDS: digests against the last commit's build, the untraced timeline's tail rows, interleaved timings at small and headline shapes
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_head.txt
timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_new.txt
diff gpurun_out/dig_head.txt gpurun_out/dig_new.txt && echo "DIGESTS EQUAL"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deconstructed or purge or hook or update_M or full_sweep" 2>&1 | tail -2
BNR_HIP_LIB=$R/_stamps/libbnr_hip.so timeout -k 5 100 python tools/stamps_timeline.py 1 100 30 5 | grep "tail\|node"
BNR_HIP_LIB=$R/_stamps/libbnr_hip.so timeout -k 5 100 python tools/stamps_timeline.py 8 | grep "tail\|node"
for r in 1 2 3; do
  for v in head new; do
    if [ $v = head ]; then export BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so; else unset BNR_HIP_LIB; fi
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 1 1000 100 30 5 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 1 1000 200 50 5 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- default | tail -1
    echo -n "$v: "; timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- default | tail -1
  done
done
} > gpurun_out/r6_ta.log 2>&1
cat gpurun_out/r6_ta.log
