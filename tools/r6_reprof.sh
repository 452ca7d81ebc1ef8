#!/bin/bash
# the two kernel traces of the round-6 measurement pass again (after tools/prof_summary.py learnt to split the Gram rows by launch size)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_default $O/prof_binary
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o r6 -- python3 $R/bench.py --steps 480 --warmup 24 --no-cpu-baseline > $O/prof_default.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $O/prof_default.log; exit 1; }
python3 $R/tools/prof_summary.py $O/prof_default > $O/kernel_stats.txt; head -12 $O/kernel_stats.txt
python3 $R/tools/sweep_stats.py $O/prof_default > $O/sweep_stats.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_binary -o r6b -- python3 $R/bench.py --steps 480 --warmup 24 --no-cpu-baseline --binary-x > $O/prof_binary.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $O/prof_binary > $O/kernel_stats_binary_x.txt; head -8 $O/kernel_stats_binary_x.txt
rm -rf $O/prof_default $O/prof_binary
