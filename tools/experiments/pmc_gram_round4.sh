#!/bin/bash
# PMC counters of the Gram kernels, ONE rocprofv3 --pmc pass per group of counters (guide: FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled on
# gfx950; no --pmc together with the trace domains gpurun refuses): one chain (k_gram<bnr_one, 2>), a lockstep group of 8 (k_gram8<bnr_many>)
# on the DEFAULT library, and -- second half, experiments build -- the same group with round 3's tiles (no dead-block skip), the static resident
# kernel k_gram8s and the resident kernel with per-CU work lists k_gram8q.
# Every pass keeps its log and its return code under gpurun_out/pmc_round4/ whatever happens (round 3 lost the evidence of two early exits
# and one hang), and the driven script prints flushed phase markers, so that a pass that stops says where.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_round4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tg4.py <<PY
import sys, os; sys.path.insert(0, "$R")
def mark(s): print("PHASE", s, flush=True)
import bnr_amd
exp = len(sys.argv) > 1
mark("import done, experiments build: %s" % exp)
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
solo = bnr_amd.Chain(X, y, 7, 12, 20240501, 1)
solo.init_prior(); solo.set_option("graph", 0); solo.set_option("overlap", 0)
mark("one chain: k_gram<bnr_one, 2>")
solo.run(2, 9, 9)
chains = [bnr_amd.Chain.like(solo, 20240501, c, 40) for c in range(1, 9)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains); g.set_option("graph", 0); g.set_option("overlap", 0)
mark("group of 8: k_gram8<bnr_many>")
g.run(2, 12, 12)
if exp:
    mark("group of 8: k_gram8s (static resident)")
    g.set_option("gram_variant", 11); g.run(13, 22, 22)
    mark("group of 8: k_gram8q (per-CU lists, cu 7 reserved)")
    g.set_option("gram_variant", 13); g.set_option("resv_mask", 0x80); g.run(23, 32, 32)
mark("counters %s" % chains[0].counters())
mark("end")
PY
rm -f $O/pmc_all.txt
run_passes() {   # $1 = tag of the build, $2 = extra argument of tg4.py
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    tag=$1_$(echo $pass | tr ' ' '_' | cut -c1-20)
    rm -rf /tmp/pm_$tag
    echo "pass $tag: start $(date +%T)" | tee -a $O/passes.log
    timeout -k 5 75 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_$tag -o p -- python3 /tmp/tg4.py $2 > $O/pass_$tag.log 2>&1
    rc=$?
    echo "pass $tag: rc $rc, last phase: $(grep PHASE $O/pass_$tag.log | tail -1)" | tee -a $O/passes.log
    python3 - <<PY >> $O/pmc_all.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gram" in k and "reduce" not in k and "gate" not in k:
            base = k.split("(")[0].replace("void ", "")
            who = "$1:" + ("k_gram<bnr_one,2>" if "bnr_one" in k else base.split("<")[0] + "<bnr_many>")
            agg[(who, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (who, c), v in sorted(agg.items()):
    print(who, c, sum(v) / len(v), len(v))
PY
  done
}
unset BNR_HIP_LIB
run_passes default ""
export BNR_HIP_LIB=$R/bayesiannetworkregression.jl_amd/csrc/_var/r3like.so
run_passes r3like ""
export BNR_HIP_LIB=$R/bayesiannetworkregression.jl_amd/csrc/_var/exp.so
run_passes exp exp
unset BNR_HIP_LIB
cat $O/pmc_all.txt
python3 - <<PY
import json
d = {}
for line in open("$O/pmc_all.txt"):
    who, c, v, n = line.split()
    d.setdefault(who, {})[c] = float(v)
out = {"config": "n=500 V=100 q=5050 R=7; one chain alone / a lockstep group of 8; builds: default (dead blocks of the diagonal tiles skipped, no progress count), r3like (round 3's k_gram8: full diagonal tiles + per-workgroup progress atomic), exp (experiments: k_gram8s static resident, k_gram8q per-CU lists with cu id 7 of every shader engine left free); tools/pmc_gram_round4.sh, eager single-stream launches, one --pmc pass per line of counters"}
for who, c in d.items():
    o = {"counters": c}
    if "FETCH_SIZE" in c: o["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c: o["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: o["traffic_bytes_per_launch"] = o["fetch_bytes_per_launch"] + o["write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c: o["mfma_busy_frac_of_launch"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)   # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
    out[who] = o
json.dump(out, open("$O/gram_pmc_round4.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
echo "all passes:"; cat $O/passes.log
