#!/bin/bash
# builds the library variants of a measurement pass into csrc/_var/ (cross-compiled here; the .so files travel to the GPU box)
set -e
cd "$(dirname "$0")/../bayesiannetworkregression.jl_amd/csrc"
mkdir -p _var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -shared"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"; [ "$defs" = "$spec" ] && defs=""
  echo "building _var/$name.so with '$defs'"
  /opt/rocm/bin/hipcc $FLAGS $defs -o _var/$name.so bnr_hip.hip &
done
wait
ls -la _var
