#!/bin/bash
# timing experiments on k_gram (compile-time variants; some give WRONG results on purpose -- timing only).  Runs on the GPU box.
# usage: tools/gram_experiments.sh "<-D flags variant 1>" "<-D flags variant 2>" ...
cd bayesiannetworkregression.jl_amd/csrc
for e in "$@"; do
  make clean > /dev/null; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off $e" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== $e"
  (cd ../.. && timeout -k 10 120 python tools/time_gram.py && timeout -k 10 120 python tools/two_groups.py 1 8 400 overlap=0 | sed -n 1p)
done
make clean > /dev/null; make > /dev/null 2>&1
