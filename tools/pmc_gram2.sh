#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tg.py <<PY
import sys; sys.path.insert(0, "$R")
import bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
ch = bnr_amd.Chain(X, y, 7, 8, 20240501, 1)
ch.init_prior(); ch.run(2, 8, 4)
print(ch.debug_time_gram(20))
PY
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_$tag -o p -- python3 /tmp/tg.py > /tmp/pm_$tag.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_gram("):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(v) / len(v), 1) for c, v in agg.items()})
PY
done
