#!/bin/bash
# round 6: GPU test-suite + digests + interleaved timings against the round-5 build
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r6.log 2>&1; echo "pytest exit=$?"; tail -5 gpurun_out/pytest_gpu_r6.log
tools/r6_ab.sh noprof | grep -v "^n="
