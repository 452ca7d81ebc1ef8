#!/bin/bash
# untraced timelines of one sweep (one chain, the group of 8, config 2) from the -DBNR_STAMPS build
mkdir -p gpurun_out
export BNR_HIP_LIB=$GRAFT_REPO_ROOT/_stamps/libbnr_hip.so
{ timeout -k 10 200 python tools/stamps_timeline.py 1 && timeout -k 10 200 python tools/stamps_timeline.py 8 && timeout -k 10 200 python tools/stamps_timeline.py 1 200 50 5 ; } > gpurun_out/r6_timeline.log 2>&1
cat gpurun_out/r6_timeline.log
