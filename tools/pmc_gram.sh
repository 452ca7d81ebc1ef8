#!/bin/bash
# PMC passes for k_gram (separate from tracing, as the guide prescribes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|MFMA|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|SQ_INSTS_VALU_MFMA|TCP_TCC_READ|SQ_INST_CYCLES_VMEM|GRBM_GUI_ACTIVE" | head -40 > $R/gpurun_out/pmc_list.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "pass $tag exit=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("k_gram", "k_chol_step", "k_backproj", "k_xpass"):
            if k in agg:
                print(k, {c: sum(v) / len(v) for c, v in agg[k].items()})
PY
