#!/bin/bash
# Round 4, notes H: an empty kernel captured as the FIRST forked branch of every sweep (nop_fork = 1) and the one-workgroup factorization of small
# problems (factor_variant 5), experiments build (csrc/_var/exp.so: tools/r4_build_variants.sh "exp:-DBNR_EXPERIMENTS")
cd "$(dirname "$0")/.."
export BNR_HIP_LIB=$PWD/bayesiannetworkregression.jl_amd/csrc/_var/exp.so
export BNR_SWEEPS=1500
python3 tools/variant_time.py factor_variant=-1 nop_fork=1 factor_variant=4 factor_variant=4,nop_fork=1
BNR_SHAPE=70,19,5 BNR_SWEEPS=3000 python3 tools/variant_time.py factor_variant=0 factor_variant=0,nop_fork=1 factor_variant=5 factor_variant=5,nop_fork=1
BNR_SHAPE=200,50,5 BNR_SWEEPS=3000 python3 tools/variant_time.py factor_variant=0 factor_variant=0,nop_fork=1
