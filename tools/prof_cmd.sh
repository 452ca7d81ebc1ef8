#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python script> [args]   -- rocprofv3 kernel trace of any python driver; leaves gpurun_out/prof_<tag>/
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
s=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/$s "$@" > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof exit=$?"
