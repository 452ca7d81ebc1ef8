#!/bin/bash
# usage: tools/ab_base.sh [bench args]  -- the built library against tools/_ab/libbnr_base.so (built from another commit), interleaved three times on the same box through
# BNR_HIP_LIB: 640-sweep bench line, the driver-style 20-step line, one chain
for r in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export BNR_HIP_LIB=$GRAFT_REPO_ROOT/tools/_ab/libbnr_base.so; else unset BNR_HIP_LIB; fi
    echo -n "$v: "; tools/quick_bench.sh "$@"
    echo -n "   20 steps: "; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,1))"
  done
done
