#!/bin/bash
# Round 4: the experiments' own measurements, one file each under gpurun_out/r4x/ (copied to profiles/round4_exp_*.txt).
# Needs csrc/_var/exp.so and csrc/_var/stamps.so (tools/r4_build_variants.sh "exp:-DBNR_EXPERIMENTS" "stamps:-DBNR_EXPERIMENTS -DBNR_STAMPS").
set -e -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r4x; mkdir -p $O
timeout -k 10 300 python3 tools/r4_exp3.py > $O/chol_skip.txt 2>&1
echo "chol_skip done"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_sync_probe tools/xcd_sync_probe.hip
timeout -k 10 120 /tmp/xcd_sync_probe > $O/xcd_sync_probe.txt 2>&1
echo "xcd_sync_probe done"
timeout -k 10 300 python3 tools/r4_df.py > $O/dataflow.txt 2>&1
echo "dataflow A/B done"
timeout -k 10 120 python3 tools/stamps_df.py > $O/dataflow_stamps.txt 2>&1
echo "dataflow stamps done"
