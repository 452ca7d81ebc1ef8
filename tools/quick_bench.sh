#!/bin/bash
# usage: tools/quick_bench.sh [bench args]  -- value, ms/step, gram alone / beside the scalar branch, single chain
python bench.py --gpus 1 --steps 640 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step; gram', round(r['avg_launch_us'],1), round(r['avg_launch_us_two_branch_schedule'],1), 'frac', round(r['frac'],3), 'sweep_frac', round(r['sweep_frac'],3), 'single', round(d.get('single_chain', {'value': 0})['value']))"
