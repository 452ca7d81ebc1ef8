#!/bin/bash
# PMC counters of the Gram kernels, separate --pmc passes (guide: FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled on gfx950):
# one chain (k_gram<bnr_one, 2>), a lockstep group of 8 (k_gram8<bnr_many>), and the same group with the persistent kernel
# (k_gram8p<bnr_many, nothing reserved).  Writes gpurun_out/gram_pmc_round3.json.  Runs on the GPU box.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tg.py <<PY
import sys; sys.path.insert(0, "$R")
import bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
solo = bnr_amd.Chain(X, y, 7, 12, 20240501, 1)
solo.init_prior(); solo.set_option("graph", 0); solo.set_option("overlap", 0)
solo.run(2, 9, 9)                           # one chain alone: k_gram<bnr_one, 2>
chains = [bnr_amd.Chain.like(solo, 20240501, c, 24) for c in range(1, 9)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains); g.set_option("graph", 0); g.set_option("overlap", 0)
g.run(2, 12, 12)                            # the group: k_gram8<bnr_many>
if len(sys.argv) > 1: g.set_option("gram_variant", 9)          # only on request (see notes round 3, E)
if len(sys.argv) > 1: g.run(13, 24, 24)
PY
rm -f /tmp/pmc_all.txt
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-20)
  rm -rf /tmp/pm_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_$tag -o p -- python3 /tmp/tg.py > /tmp/pm_$tag.log 2>&1 || { echo "pass $pass failed"; cp /tmp/pm_$tag.log $R/gpurun_out/pmc_fail_$tag.log; }
  python3 - <<PY >> /tmp/pmc_all.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gram" in k and "reduce" not in k and "gate" not in k:
            who = "k_gram<bnr_one,2>" if "bnr_one" in k else ("k_gram8p<bnr_many" if "k_gram8p" in k else "k_gram8<bnr_many>")
            agg[(who, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (who, c), v in sorted(agg.items()):
    print(who, c, sum(v) / len(v), len(v))
PY
done
cat /tmp/pmc_all.txt
python3 - <<PY
import json
d = {}
for line in open("/tmp/pmc_all.txt"):
    who, c, v, n = line.split()
    d.setdefault(who, {})[c] = float(v)
out = {"config": "n=500 V=100 q=5050 R=7; one chain alone / a lockstep group of 8 / the same group with the persistent kernel (tools/pmc_gram_round3.sh, eager single-stream launches, one --pmc pass per line of counters)"}
for who, c in d.items():
    o = {"counters": c}
    if "FETCH_SIZE" in c: o["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c: o["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: o["traffic_bytes_per_launch"] = o["fetch_bytes_per_launch"] + o["write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c: o["mfma_busy_frac_of_launch"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)   # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
    out[who] = o
json.dump(out, open("$R/gpurun_out/gram_pmc_round3.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
