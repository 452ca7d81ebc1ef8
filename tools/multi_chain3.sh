#!/bin/bash
# lockstep groups: throughput vs chains per GPU (headline workload)
for c in 1 2 4 8; do
  timeout -k 10 300 python bench.py --steps 400 --warmup 48 --chains-per-gpu $c --no-cpu-baseline > gpurun_out/mc3_$c.json 2> gpurun_out/mc3_$c.err || { echo "c=$c failed"; tail -3 gpurun_out/mc3_$c.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/mc3_$c.json")); print("chains/GPU %d: %.0f it/s  ms/step %.3f  gram %.1f us/launch  %.1f TF/s" % ($c, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["achieved"]))
PY
done
