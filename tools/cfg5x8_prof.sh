R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for kind in bool8 real; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5x8_$kind -o c5 -- python3 $R/tools/cfg5_passes.py $kind 0 8 > $R/gpurun_out/prof_c5x8_$kind.log 2>&1
echo "== $kind, 8 chains"; python3 $R/tools/prof_summary.py $R/gpurun_out/prof_c5x8_$kind | head -14
done
