#!/bin/bash
# Round-3 measurement pass on the GPU box: rocprofv3 kernel stats of the bench command (default schedule), one line per BASELINE
# config.  Everything lands in gpurun_out/ (copy what is to be judged into profiles/round3_*).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_round3 -o r3 -- python3 $R/bench.py --steps 200 --warmup 24 --no-cpu-baseline > $R/gpurun_out/prof_round3.log 2>&1 || exit 1
cd $R && python tools/prof_summary.py gpurun_out/prof_round3 > gpurun_out/round3_kernel_stats.txt
cp gpurun_out/prof_round3/r3_kernel_stats.csv gpurun_out/round3_rocprofv3_kernel_stats.csv
head -16 gpurun_out/round3_kernel_stats.txt
rm -f gpurun_out/round3_configs.txt
for cfg in cfg2 cfg4 cfg5; do
  python bench.py --config $cfg --chains-per-gpu 1 --steps 200 --warmup 24 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg one chain:', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; k_gram', round(r['avg_launch_us'],1), 'us', round(r['achieved'],1), 'TFLOP/s frac', round(r['frac'],3), 'sweep_frac', round(r['sweep_frac'],3))" | tee -a gpurun_out/round3_configs.txt || exit 1
done
python bench.py --config cfg3 --chains-per-gpu 16 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 16 chains:', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep')" | tee -a gpurun_out/round3_configs.txt
