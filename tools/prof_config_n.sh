#!/bin/bash
# rocprofv3 kernel stats of one BASELINE.json config with C chains (usage: tools/prof_config_n.sh cfg4 8 [steps] [--binary-x])
cfg=$1; C=$2; steps=${3:-40}; extra=$4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${cfg}_$C -o $cfg -- python3 $R/bench.py --config $cfg --chains-per-gpu $C --steps $steps --warmup 8 --no-cpu-baseline $extra > $R/gpurun_out/prof_${cfg}_$C.log 2>&1; echo "prof exit=$?"
cd $R && python tools/prof_summary.py gpurun_out/prof_${cfg}_$C | head -16; grep '"metric"' gpurun_out/prof_${cfg}_$C.log | cut -c1-200
