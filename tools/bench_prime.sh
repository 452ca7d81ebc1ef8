#!/bin/bash
# driver-style short bench with different amounts of graph priming, next to the long run
for p in 0 1 4; do
  BNR_BENCH_PRIME=$p python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prime $p', d['value'], d['ms_per_step'], d['single_chain']['value'])" || exit 1
done
python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('long', d['value'], d['ms_per_step'], d['single_chain']['value'])"
