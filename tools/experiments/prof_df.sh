#!/bin/bash
# usage: tools/prof_df.sh <tag>   -- rocprofv3 kernel trace + timeline of the data-flow factorization (experiments build), 8 chains and 1 chain
tag=$1
R=$GRAFT_REPO_ROOT
export BNR_HIP_LIB=$R/bayesiannetworkregression.jl_amd/csrc/_var/${DF_LIB:-exp}.so
export BNR_SWEEPS=200
cd /tmp && export TMPDIR=/tmp
for nb in 8 1; do
  export BNR_GROUPS=$nb
  rm -rf $R/gpurun_out/prof_${tag}_$nb
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$nb -o df -- python3 $R/tools/variant_time.py factor_variant=4 > $R/gpurun_out/prof_${tag}_$nb.log 2>&1 || { echo "prof failed"; tail -5 $R/gpurun_out/prof_${tag}_$nb.log; exit 1; }
  python3 $R/tools/trace_timeline.py $R/gpurun_out/prof_${tag}_$nb 1500 30 > $R/gpurun_out/timeline_${tag}_$nb.txt
  python3 $R/tools/prof_summary.py $R/gpurun_out/prof_${tag}_$nb | head -12
done
