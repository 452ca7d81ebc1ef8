#!/bin/bash
# the two bandwidth-bound passes over X at BASELINE configs[4]'s size under rocprofv3, per image of X and back-projection kernel -> gpurun_out/cfg5_passes.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/cfg5_passes.txt; : > $out
for v in ${VARIANTS:-real:0 bool8:0}; do   # image:variant, variant = pair_backproj + 2 * cu_backproj
  set -- ${v/:/ }; tag=${1}_$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5_$tag -o c5 -- python3 $R/tools/cfg5_passes.py $1 $2 > $R/gpurun_out/prof_c5_$tag.log 2>&1
  echo "== X $1, back-projection variant $2: $(tail -1 $R/gpurun_out/prof_c5_$tag.log)" >> $out
  python3 $R/tools/prof_summary.py $R/gpurun_out/prof_c5_$tag | grep -E "kernel|k_xpass|k_backproj|k_gram|k_chol|k_tail|total" >> $out
done
python3 - >> $out <<PY
import re
txt = open("$out").read()
n_pad, q = 512, 45150
for sec in txt.split("== ")[1:]:
    head = sec.splitlines()[0]
    per = 8 if head.startswith("X real") or "bool64" in head else 1
    mb = n_pad * q * per / 1e6
    for line in sec.splitlines()[1:]:
        m = re.match(r"(?:void )?(k_xpass|k_backproj[23]?)\S*.*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m and float(m.group(3)) > 20:
            us = float(m.group(3))
            print("%-40s %-12s %6.1f us per launch, reads X once = %6.1f MB -> %5.2f TB/s = %.2f of the 8 TB/s HBM peak" % (head.split(":")[0], m.group(1), us, mb, mb / us, mb / us / 8.0))
PY
cat $out
