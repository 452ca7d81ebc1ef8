#!/bin/bash
for cfg in "1 0 9 0" "1 0 9 1" "1 1 0 0" "1 1 0 1"; do
  echo "== factor pipeline gram graph = $cfg"
  timeout -k 5 40 python tools/stage_test.py $cfg 2>&1 | tail -6
  rc=${PIPESTATUS[0]}; echo "rc=$rc"
  if [ "$rc" != "0" ]; then echo "stopping"; break; fi
done
