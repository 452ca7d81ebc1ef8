#!/bin/bash
# usage: tools/run_gpu.sh <tag> [pytest|nopytest] [bench args...]
tag=$1; shift
dopytest=$1; shift
mkdir -p gpurun_out
if [ "$dopytest" = "pytest" ]; then
  python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest exit=$?"
  tail -4 gpurun_out/pytest_gpu_$tag.log
fi
python bench.py "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench exit=$?"
cat gpurun_out/bench_$tag.json; tail -3 gpurun_out/bench_$tag.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --steps 200 --warmup 24 --no-cpu-baseline $PROF_ARGS > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof exit=$?"
cd $R && python tools/prof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_${tag}_summary.txt; cat gpurun_out/prof_${tag}_summary.txt

