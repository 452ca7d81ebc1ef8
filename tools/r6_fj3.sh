#!/bin/bash
mkdir -p gpurun_out
{
for sh in "1 1000 70 19 5" "1 1000 100 30 5" "1 1000 140 40 5" "1 1000 300 60 5" "1 400 500 300 10" "8 640 100 30 5" "2 1000 100 30 5" "8 200 500 300 10"; do
  echo "== $sh"
  timeout -k 10 300 python tools/ab_opt.py $sh -- flag_join=0 flag_join=1 | sort
done
} > gpurun_out/r6_fj3.log 2>&1
cat gpurun_out/r6_fj3.log
