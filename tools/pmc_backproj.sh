#!/bin/bash
# PMC counters of the back-projection kernels at BASELINE configs[4]'s size (one chain) and at the headline shape (8 chains), separate --pmc passes -> gpurun_out/pmc_backproj.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tb.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 300, 10, seed=20240501)
import os
for pair in ((0, 1) if os.environ.get("BNR_BP_PATCH") else (0,)):      # BNR_BP_PATCH=1: a build with tools/experiments/backproj_pipelines.patch (k_backproj2 as well)
    ch = bnr_amd.Chain(X, y, 10, 12, 21, 1)
    if pair: ch.set_option("pair_backproj", pair)
    ch.set_option("graph", 0); ch.set_option("overlap", 0)
    ch.init_prior(); ch.run(2, 12, 12); ch.close()
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 12, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 12) for c in range(2, 9)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains); g.set_option("graph", 0); g.set_option("overlap", 0)
g.run(2, 12, 12)
PY
out=$R/gpurun_out/pmc_backproj.txt; : > $out
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-20)
  rm -rf /tmp/pb_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pb_$tag -o p -- python3 /tmp/tb.py > /tmp/pb_$tag.log 2>&1 || { echo "pass $pass failed" >> $out; tail -5 /tmp/pb_$tag.log >> $out; }
  python3 - <<PY >> $out
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pb_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_backproj" in k:
            agg[(k.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (who, c), v in sorted(agg.items()):
    print("%-28s %-24s %14.0f  (%d launches)" % (who, c, sum(v) / len(v), len(v)))
PY
done
cat $out
