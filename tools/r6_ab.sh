#!/bin/bash
# round 6: the built library against tools/_ab/libbnr_base.so (round-5 HEAD): table digests under both builds, interleaved timings, then the kernel trace of the bench command
# usage: tools/r6_ab.sh [noprof]
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
echo "== digests base"; BNR_HIP_LIB=$R/tools/_ab/libbnr_base.so python tools/table_digest.py > gpurun_out/dig_base.txt
echo "== digests new"; python tools/table_digest.py > gpurun_out/dig_new.txt; cat gpurun_out/dig_new.txt
diff gpurun_out/dig_base.txt gpurun_out/dig_new.txt && echo "DIGESTS EQUAL"
echo "== tail stamps headline"; BNR_HIP_LIB=_stamps/libbnr_hip.so python tools/stamps_tail.py 500 100 7
for r in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export BNR_HIP_LIB=$R/tools/_ab/libbnr_base.so; else unset BNR_HIP_LIB; fi
    echo -n "$v: "; tools/quick_bench.sh
  done
done
unset BNR_HIP_LIB
if [ "$1" != "noprof" ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r6ab -o r6ab -- python3 $R/bench.py --steps 480 --warmup 24 --no-cpu-baseline > $R/gpurun_out/prof_r6ab.log 2>&1; echo "prof exit=$?"
cd $R
python tools/prof_summary.py gpurun_out/prof_r6ab | head -22
python tools/sweep_stats.py gpurun_out/prof_r6ab
rm -rf gpurun_out/prof_r6ab
fi
} > gpurun_out/r6_ab.log 2>&1
cat gpurun_out/r6_ab.log
