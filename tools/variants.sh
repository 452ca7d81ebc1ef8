#!/bin/bash
# build variants (compile-time -D flags) and time one chain / a group of 8.  usage: tools/variants.sh "<flags 1>" "<flags 2>" ...
cd bayesiannetworkregression.jl_amd/csrc
for e in "$@"; do
  make clean > /dev/null; make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off $e" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== $e"
  (cd ../.. && timeout -k 10 120 python tools/two_groups.py 1 1 600 | sed -n 1p && timeout -k 10 120 python tools/two_groups.py 1 8 600 | sed -n 1p)
done
make clean > /dev/null; make > /dev/null 2>&1
