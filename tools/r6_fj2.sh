#!/bin/bash
mkdir -p gpurun_out
{
timeout -k 10 300 python tools/experiments/r6_fj.py
for r in 1 2; do
  for o in flag_join=0 flag_join=1; do
    timeout -k 10 200 python tools/ab_opt.py 1 1000 200 50 5 -- $o | tail -1
    timeout -k 10 200 python tools/ab_opt.py 1 1000 100 30 5 -- $o | tail -1
    timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- $o | tail -1
    timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- $o | tail -1
    timeout -k 10 200 python tools/ab_opt.py 8 640 200 50 5 -- $o | tail -1
  done
done
} > gpurun_out/r6_fj2.log 2>&1
cat gpurun_out/r6_fj2.log
