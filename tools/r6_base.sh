#!/bin/bash
# round 6 baseline pass: headline line, in-kernel phases of k_tail and of the panel sweep, config 2 alone
mkdir -p gpurun_out
{
echo "== quick bench"; tools/quick_bench.sh
echo "== tail stamps headline"; BNR_HIP_LIB=_stamps/libbnr_hip.so python tools/stamps_tail.py 500 100 7
echo "== panel stamps"; BNR_HIP_LIB=_stamps/libbnr_hip.so python tools/stamps_panel.py
echo "== cfg2 one chain"; python tools/ab_opt.py 1 2000 200 50 5 -- default
echo "== headline one chain"; python tools/ab_opt.py 1 1000 500 100 7 -- default
} > gpurun_out/r6_base.log 2>&1
tail -40 gpurun_out/r6_base.log
