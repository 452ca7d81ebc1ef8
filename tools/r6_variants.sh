#!/bin/bash
# usage: tools/r6_variants.sh tag1 tag2 ...   -- tools/_ab/libbnr_<tag>.so, one chain and 8 chains at the headline shape, interleaved twice
R=$GRAFT_REPO_ROOT
{
for r in 1 2; do
for t in "$@"; do
  echo -n "$t: "; BNR_HIP_LIB=$R/tools/_ab/libbnr_$t.so python tools/ab_opt.py 1 1000 500 100 7 -- default | tail -1 | tr '\n' ' '
  BNR_HIP_LIB=$R/tools/_ab/libbnr_$t.so python tools/ab_opt.py 8 400 500 100 7 -- default | tail -1
done
done
} > gpurun_out/r6_variants.log 2>&1
cat gpurun_out/r6_variants.log
