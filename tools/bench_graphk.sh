#!/bin/bash
for k in 8 16 32 64; do
  python bench.py --gpus 1 --steps 1920 --warmup 192 --no-cpu-baseline --graph-k $k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph_k $k', d['value'], d['ms_per_step'])" || exit 1
done
