#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bytex$b -o bx$b -- python3 $R/tools/byte_x_prof.py $b > $R/gpurun_out/prof_bytex$b.log 2>&1
  echo "== byte_x=$b"; tail -1 $R/gpurun_out/prof_bytex$b.log
  python3 $R/tools/prof_summary.py $R/gpurun_out/prof_bytex$b | grep -E "kernel|k_xpass|k_backproj|k_gram|total"
done
