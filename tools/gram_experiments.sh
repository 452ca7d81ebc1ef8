#!/bin/bash
# timing experiments on k_gram (results are WRONG for EXP != 0; timing only).  Runs on the GPU box.
cd bayesiannetworkregression.jl_amd/csrc
for e in 0 1 2 3; do
  make clean > /dev/null; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DBNR_GRAM_EXP=$e" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== BNR_GRAM_EXP=$e"
  (cd ../.. && timeout -k 10 120 python tools/time_gram.py)
done
make clean > /dev/null; make > /dev/null 2>&1
