#!/bin/bash
# ONE rehearsal of bench.py's multi-rank path on a one-GPU box (two ranks on device 0, exchanges over gloo), with every rank's own
# stdout/stderr kept under gpurun_out/r2_logs.  The same run is tests/test_zz_harness_gpu.py::test_bench_multi_rank_path_rehearsal.
# (History: round 2's first rehearsals failed at the rendezvous on boxes whose hostname does not resolve -- gloo and RCCL's bootstrap
#  socket picked the outward interface; bench.py now pins GLOO_SOCKET_IFNAME / NCCL_SOCKET_IFNAME to lo for a loopback master.)
set -o pipefail
rm -rf gpurun_out/r2_logs; mkdir -p gpurun_out/r2_logs
BNR_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
  --redirects 3 --log-dir gpurun_out/r2_logs bench.py --gpus 2 --steps 40 --warmup 8 --config cfg2 --no-cpu-baseline \
  > gpurun_out/r2_stdout.txt 2> gpurun_out/r2_stderr.txt
rc=$?
echo "rc=$rc"
find gpurun_out/r2_logs -type f -name "stdout.log" | while read f; do grep "^{" "$f"; done
if [ $rc -ne 0 ]; then find gpurun_out/r2_logs -type f | while read f; do echo "== $f"; tail -60 "$f"; done; fi
exit $rc
