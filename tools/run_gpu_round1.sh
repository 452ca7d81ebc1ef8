#!/bin/bash
# first GPU measurement pass: MFMA f64 peak, bench, rocprofv3 kernel stats
set -x
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_peak tools/mfma_f64_peak.hip && /tmp/mfma_f64_peak > gpurun_out/mfma_f64_peak.txt 2>&1
cat gpurun_out/mfma_f64_peak.txt
python bench.py --steps 500 --warmup 50 > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err; echo bench exit=$?
cat gpurun_out/bench_v1.json
tail -3 gpurun_out/bench_v1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_v1 -o v1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_v1.log 2>&1; echo prof exit=$?
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_v1 | head -20
