#!/bin/bash
# usage: tools/prof_two_groups.sh <tag> G C steps [key=value ...] -- rocprofv3 kernel trace of tools/two_groups.py
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/tools/two_groups.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof exit=$?"
cd $R && python tools/prof_summary.py gpurun_out/prof_$tag | head -12
