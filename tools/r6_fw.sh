#!/bin/bash
# fuse_w: digests with and without (same library), a few parity tests, interleaved timings
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
BNR_HIP_LIB=$R/tools/_ab/libbnr_head.so timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_head.txt
timeout -k 10 300 python tools/table_digest.py > gpurun_out/dig_new.txt
diff gpurun_out/dig_head.txt gpurun_out/dig_new.txt && echo "DIGESTS EQUAL"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_sweep_matches_oracle or deconstructed or purge or hook or group" 2>&1 | tail -3
for r in 1 2 3; do
  for o in fuse_w=0 fuse_w=1; do
    timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- $o | tail -1
    timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- $o | tail -1
    timeout -k 10 200 python tools/ab_opt.py 1 1000 200 50 5 -- $o | tail -1
  done
done
} > gpurun_out/r6_fw.log 2>&1
cat gpurun_out/r6_fw.log
