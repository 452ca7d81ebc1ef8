#!/bin/bash
# graph-concurrency variants of tools/reserve_probe.hip (4 reserved CUs per XCD = one per shader engine)
P=./build_tools/reserve_probe
run() { echo "== variant $VARIANT $*"; env "$@" timeout -k 10 60 $P 4 2016 200 ${VARIANT:-1} 2>&1 | grep -E "graph|round|TIMED" ; }
for v in 7 8; do VARIANT=$v run A=1; done
