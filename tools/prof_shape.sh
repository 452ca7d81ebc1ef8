#!/bin/bash
# usage: tools/prof_shape.sh <tag> <n,V,R> <sweeps> <groups> key=value ...  -- rocprofv3 kernel trace of tools/variant_time.py at a shape
tag=$1; shape=$2; sweeps=$3; groups=$4; shift 4
R=$GRAFT_REPO_ROOT
export BNR_SHAPE=$shape BNR_SWEEPS=$sweeps BNR_GROUPS=$groups
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/tools/variant_time.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof exit=$?"
cd $R && python tools/prof_summary.py gpurun_out/prof_$tag | head -14
