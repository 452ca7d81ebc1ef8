#!/bin/bash
# usage: tools/ab_lib.sh  -- the built library against tools/_ab/libbnr_b.so (a variant build), interleaved on the same box: quick_bench + one chain + driver-style line
L=bayesiannetworkregression.jl_amd/libbnr_hip.so
cp $L /tmp/base.so
for r in 1 2 3; do
  for v in base b; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp tools/_ab/libbnr_b.so $L; fi
    echo -n "$v: "; tools/quick_bench.sh "$@"
    echo -n "   20 steps: "; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,1))"
  done
done
cp /tmp/base.so $L
