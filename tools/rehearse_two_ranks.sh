#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a one-GPU box with every rank's own stderr kept (gpurun_out/r2_logs).
# usage: tools/rehearse_two_ranks.sh [repeats]   -- stops at the first failing repeat
set -o pipefail
N=${1:-1}
for i in $(seq 1 $N); do
  rm -rf gpurun_out/r2_logs; mkdir -p gpurun_out/r2_logs
  BNR_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29533 + i)) \
    --redirects 3 --log-dir gpurun_out/r2_logs bench.py --gpus 2 --steps 40 --warmup 8 --chains-per-gpu 2 --config cfg2 --no-cpu-baseline \
    > gpurun_out/r2_stdout.txt 2> gpurun_out/r2_stderr.txt
  rc=$?
  echo "repeat $i rc=$rc"
  if [ $rc -ne 0 ]; then
    find gpurun_out/r2_logs -type f | while read f; do echo "== $f"; tail -60 "$f"; done
    exit $rc
  fi
done
