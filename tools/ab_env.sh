#!/bin/bash
# usage: tools/ab_env.sh VAR  -- quick_bench and the driver-style 20-step line with VAR unset and VAR=1, interleaved three times on the same box
v=$1
for r in 1 2 3; do
  for val in "" 1; do
    if [ -z "$val" ]; then unset $v; else export $v=$val; fi
    echo -n "$v=${val:-unset}: "; tools/quick_bench.sh
    echo -n "   20 steps: "; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,1))"
  done
done
