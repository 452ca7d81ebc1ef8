#!/bin/bash
# PMC: vector instructions per launch of k_backproj against k_backproj64 at BASELINE configs[4]'s size, 8 chains (11 288 chunks of 32 edges) -> gpurun_out/pmc_backproj64.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tb64.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 300, 10, seed=20240501)
for wide in (0, 1):
    chains = [bnr_amd.Chain(X, y, 10, 10, 21, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 21, c, 10) for c in range(2, 9)]
    for ch in chains: ch.init_prior()
    g = bnr_amd.Group(chains); g.set_option("graph", 0); g.set_option("overlap", 0); g.set_option("wide_backproj", wide)
    g.run(2, 10, 10); g.close()
    for ch in chains: ch.close()
PY
out=$R/gpurun_out/pmc_backproj64.txt; : > $out
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-20)
  rm -rf /tmp/pb64_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pb64_$tag -o p -- python3 /tmp/tb64.py > /tmp/pb64_$tag.log 2>&1 || { echo "pass $pass failed" >> $out; tail -5 /tmp/pb64_$tag.log >> $out; }
  python3 - <<PY >> $out
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pb64_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_backproj" in k:
            agg[(k.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (who, c), v in sorted(agg.items()):
    print("%-28s %-24s %14.0f  (%d launches)" % (who, c, sum(v) / len(v), len(v)))
PY
done
cat $out
