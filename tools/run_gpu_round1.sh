#!/bin/bash
set -x
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_peak tools/mfma_f64_peak.hip && /tmp/mfma_f64_peak > gpurun_out/mfma_f64_peak.txt 2>&1
cat gpurun_out/mfma_f64_peak.txt
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest exit=$?
tail -15 gpurun_out/pytest_gpu.log
