#!/bin/bash
# rocprofv3 kernel stats of one BASELINE.json config, one chain (usage: tools/prof_config.sh cfg5 [steps])
cfg=$1; steps=${2:-100}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$cfg -o $cfg -- python3 $R/bench.py --config $cfg --chains-per-gpu 1 --steps $steps --warmup 8 --no-cpu-baseline > $R/gpurun_out/prof_$cfg.log 2>&1; echo "prof exit=$?"
cd $R && python tools/prof_summary.py gpurun_out/prof_$cfg > gpurun_out/prof_${cfg}_summary.txt; cat gpurun_out/prof_${cfg}_summary.txt | head -16; grep '"metric"' gpurun_out/prof_$cfg.log | cut -c1-300
