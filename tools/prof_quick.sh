#!/bin/bash
# usage: tools/prof_quick.sh <tag> [bench args]   -- rocprofv3 kernel trace of a short bench run, per-kernel summary
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --steps 200 --warmup 24 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof exit=$?"
cd $R && python tools/prof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_${tag}_summary.txt; head -30 gpurun_out/prof_${tag}_summary.txt; tail -2 gpurun_out/prof_$tag.log
