#!/bin/bash
# PMC counters of k_gram, one chain and a lockstep group of 8, separate --pmc passes (guide: FETCH_SIZE/WRITE_SIZE in KB,
# FETCH_SIZE doubled on gfx950).  Writes gpurun_out/gram_pmc_round2.json.  Runs on the GPU box.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tg.py <<PY
import sys; sys.path.insert(0, "$R")
import bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 16, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 16) for c in range(2, 9)]
for ch in chains: ch.init_prior()
chains[0].set_option("graph", 0); chains[0].set_option("overlap", 0)
chains[0].run(2, 9, 9)                      # one chain alone: k_gram<bnr_one, 2>
g = bnr_amd.Group(chains); g.set_option("graph", 0); g.set_option("overlap", 0)
g.run(10, 16, 16)                           # the group: k_gram<bnr_many, 2>
PY
rm -f /tmp/pmc_all.txt
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-20)
  rm -rf /tmp/pm_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_$tag -o p -- python3 /tmp/tg.py > /tmp/pm_$tag.log 2>&1 || { echo "pass $pass failed"; tail -5 /tmp/pm_$tag.log; }
  python3 - <<PY >> /tmp/pmc_all.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gram" in k and "reduce" not in k:
            agg[("one" if "bnr_one" in k else "group8", r["Counter_Name"])].append(float(r["Counter_Value"]))
for (who, c), v in sorted(agg.items()):
    print(who, c, sum(v) / len(v), len(v))
PY
done
cat /tmp/pmc_all.txt
python3 - <<PY
import json
d = {"one": {}, "group8": {}}
for line in open("/tmp/pmc_all.txt"):
    who, c, v, n = line.split()
    d[who][c] = float(v)
out = {"kernel": "k_gram<.., 2>", "config": "n=500 V=100 q=5050 R=7; one chain and a lockstep group of 8 (tools/pmc_gram_group.sh, eager single-stream launches)"}
for who in d:
    c = d[who]
    o = {"counters": c}
    if "FETCH_SIZE" in c: o["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c: o["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: o["traffic_bytes_per_launch"] = o["fetch_bytes_per_launch"] + o["write_bytes_per_launch"]
    out[who] = o
json.dump(out, open("$R/gpurun_out/gram_pmc_round2.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
