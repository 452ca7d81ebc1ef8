#!/bin/bash
# Round-6 measurement pass on the GPU box.  Everything lands in gpurun_out/r6prof/ (what is to be judged is copied into profiles/round6_*).
#   1. bench lines: default (2000 steps), driver-style (20 steps, 5 warm-up), --binary-x
#   2. rocprofv3 --kernel-trace --stats of the bench command (>= 400 k_gram8 launches), kernel resources from the trace
#   3. one line per BASELINE config with one chain and with eight
#   4. PMC passes of the group Gram (k_gram8) and of the one-chain Gram (k_gram), one --pmc pass per counter group
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6prof
mkdir -p $O
cd $R
echo "== 1. bench lines" | tee $O/progress.log
python bench.py --gpus 1 > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err || exit 1
python bench.py --gpus 1 --binary-x --no-cpu-baseline > $O/bench_binary_x.json 2> $O/bench_binary_x.err || exit 1
python - <<PY | tee -a $O/progress.log
import json
for f in ("bench_default", "bench_driver_style", "bench_binary_x"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"]), "it/s", round(d["ms_per_step"] * 1e3, 1), "us/sweep; Gram", round(r["avg_launch_us"], 1), "us frac", round(r["frac"], 3), "sweep_frac", round(r["sweep_frac"], 3), "single", round(d.get("single_chain", {"value": 0})["value"]), "cpu", round(d.get("cpu_baseline", {"value": 0})["value"], 1))
PY
echo "== 2. kernel stats of the bench command" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o r6 -- python3 $R/bench.py --steps 480 --warmup 24 --no-cpu-baseline > $O/prof_default.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $O/prof_default.log; exit 1; }
python3 $R/tools/prof_summary.py $O/prof_default > $O/kernel_stats.txt
head -8 $O/kernel_stats.txt | tee -a $O/progress.log
cp $O/prof_default/r6_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null
python3 - <<PY > $O/kernel_resources.txt
import csv, glob
seen = {}
for f in glob.glob("$O/prof_default/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        seen[k] = (r.get("Private_Segment_Size", r.get("Scratch_Size", "?")), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", r.get("Group_Segment_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
print("%-40s %8s %6s %6s %8s %6s" % ("kernel", "scratch", "vgpr", "sgpr", "lds", "wg"))
for k, v in sorted(seen.items()): print("%-40s %8s %6s %6s %8s %6s" % ((k[:40],) + v))
PY
rm -rf $O/prof_binary
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_binary -o r6b -- python3 $R/bench.py --steps 480 --warmup 24 --no-cpu-baseline --binary-x > $O/prof_binary.log 2>&1 || { echo "rocprofv3 (binary) failed"; exit 1; }
python3 $R/tools/prof_summary.py $O/prof_binary > $O/kernel_stats_binary_x.txt
head -8 $O/kernel_stats_binary_x.txt | tee -a $O/progress.log
cd $R
python3 tools/sweep_stats.py $O/prof_default > $O/sweep_stats.txt 2>&1; cat $O/sweep_stats.txt | tee -a $O/progress.log
rm -rf $O/prof_default/*/  $O/prof_binary/*/ 2>/dev/null      # (the traces are tens of MB: the summaries above are what is kept)
echo "== 3. configs" | tee -a $O/progress.log
rm -f $O/configs.txt
for cfg in cfg2 cfg4 cfg5; do
  for c in 1 8; do
    st=200; [ $cfg = cfg4 ] && [ $c = 8 ] && st=40
    python bench.py --config $cfg --chains-per-gpu $c --steps $st --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg $c chain(s):', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; k_gram', round(r['avg_launch_us'],1), 'us', round(r['achieved'],1), 'TFLOP/s frac', round(r['frac'],3), 'sweep_frac', round(r['sweep_frac'],3))" | tee -a $O/configs.txt || exit 1
  done
done
python bench.py --config cfg3 --chains-per-gpu 16 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg3 16 chains:', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; sweep_frac', round(r['sweep_frac'],3))" | tee -a $O/configs.txt
for cfg in cfg4 cfg5; do
  python bench.py --config $cfg --chains-per-gpu 1 --steps 200 --warmup 16 --no-cpu-baseline --binary-x 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg 1 chain, BINARY X (i8 Gram):', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; digits + Gram', round(r['avg_launch_us'],1), 'us')" | tee -a $O/configs.txt
done
[ -n "$BNR_SKIP_PMC" ] && { echo "(PMC passes skipped)" | tee -a $O/progress.log; exit 0; }
echo "== 4. PMC passes of the Gram kernels" | tee -a $O/progress.log
cd /tmp
cat > /tmp/tg6.py <<PY
import sys, os; sys.path.insert(0, "$R")
def mark(s): print("PHASE", s, flush=True)
import bnr_amd
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
mark("counters %s" % chains[0].counters())
mark("end")
PY
rm -f $O/pmc_all.txt $O/passes.log
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-20)
  rm -rf /tmp/pm_$tag
  echo "pass $tag: start $(date +%T)" | tee -a $O/passes.log
  timeout -k 5 75 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pm_$tag -o p -- python3 /tmp/tg6.py > $O/pass_$tag.log 2>&1
  rc=$?
  echo "pass $tag: rc $rc, last phase: $(grep PHASE $O/pass_$tag.log | tail -1)" | tee -a $O/passes.log
  python3 - <<PY >> $O/pmc_all.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gram" in k and "reduce" not in k:
            who = "k_gram<bnr_one,2>" if "bnr_one" in k else "k_gram8<bnr_many>"
            agg[(who, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (who, c), v in sorted(agg.items()):
    print(who, c, sum(v) / len(v), len(v))
PY
done
python3 - <<PY | tee -a $O/progress.log
import json, collections
d = collections.defaultdict(dict)
for line in open("$O/pmc_all.txt"):
    who, c, v, n = line.split()
    d[who][c] = float(v)
out = {"config": "n=500 V=100 q=5050 R=7; one chain alone / a lockstep group of 8; eager single-stream launches, one --pmc pass per group of counters (tools/round6_profiles.sh); round-5 Gram loop (unchanged in round 6) (no address VALU)"}
for who, c in d.items():
    e = {"counters": c}
    if "FETCH_SIZE" in c: e["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2          # KB, doubled on gfx950 (guide)
    if "WRITE_SIZE" in c: e["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: e["traffic_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c: e["mfma_busy_frac_of_launch"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
        mfma = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0                                        # a 16x16x4 f64 MFMA = 2048 flops = 4 MOPS of 512
        e["valu_instructions_per_mfma"] = (c["SQ_INSTS_VALU"] - mfma) / mfma
    out[who] = e
    print(who, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in e.items() if k != "counters"})
json.dump(out, open("$O/gram_pmc.json", "w"), indent=1)
PY
echo "== done" | tee -a $O/progress.log
