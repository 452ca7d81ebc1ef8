#!/bin/bash
# Round-4 measurement pass on the GPU box.  Everything lands in gpurun_out/r4prof/ (what is to be judged is copied into profiles/round4_*).
#   1. bench lines: default (2000 steps) and driver-style (20 steps, 5 warm-up)
#   2. rocprofv3 --kernel-trace --stats of the bench command (>= 400 k_gram8 launches) for three builds of the library:
#        default            (no Gram progress count, dead blocks of the diagonal tiles skipped)
#        noskip             (-DBNR_GRAM_SKIP_DEAD=0: round 3's tiles, without its count)
#        r3like             (-DBNR_EXPERIMENTS -DBNR_GRAM_SKIP_DEAD=0: round 3's k_gram8 incl. the per-workgroup progress atomic)
#      -> is the atomic what made k_gram8 3.6 % slower between rounds 2 and 3?  plus a timeline of the default
#   3. one line per BASELINE config with one chain and with eight
#   4. config 5 (the "HBM-bound large-q regime"): kernel stats with the byte image of a 0/1 model matrix on and off
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4prof
mkdir -p $O
cd $R
V=$R/bayesiannetworkregression.jl_amd/csrc/_var
echo "== 1. bench lines" | tee $O/progress.log
python bench.py --gpus 1 > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err || exit 1
python - <<PY | tee -a $O/progress.log
import json
for f in ("bench_default", "bench_driver_style"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"]), "it/s", round(d["ms_per_step"] * 1e3, 1), "us/sweep; k_gram8", round(r["avg_launch_us"], 1), "us frac", round(r["frac"], 3), "sweep_frac", round(r["sweep_frac"], 3), "single", round(d.get("single_chain", {"value": 0})["value"]), "cpu", round(d.get("cpu_baseline", {"value": 0})["value"], 1))
PY
echo "== 2. kernel stats of the bench command, three builds" | tee -a $O/progress.log
cd /tmp && export TMPDIR=/tmp
for b in default noskip r3like; do
  if [ $b = default ]; then unset BNR_HIP_LIB; else export BNR_HIP_LIB=$V/$b.so; fi
  rm -rf $O/prof_$b
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$b -o r4 -- python3 $R/bench.py --steps 480 --warmup 24 --no-cpu-baseline > $O/prof_$b.log 2>&1 || { echo "rocprofv3 ($b) failed"; tail -5 $O/prof_$b.log; exit 1; }
  python3 $R/tools/prof_summary.py $O/prof_$b > $O/kernel_stats_$b.txt
  echo "-- $b" | tee -a $O/progress.log; head -6 $O/kernel_stats_$b.txt | tee -a $O/progress.log
done
unset BNR_HIP_LIB
cp $O/prof_default/r4_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null
python3 $R/tools/trace_timeline.py $O/prof_default 20000 80 > $O/timeline_default.txt
# scratch / registers / LDS of every kernel as the trace reports them (the proof of "no scratch")
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
cd $R
echo "== 3. configs" | tee -a $O/progress.log
rm -f $O/configs.txt
for cfg in cfg2 cfg4 cfg5; do
  for c in 1 8; do
    st=200; [ $cfg = cfg4 ] && [ $c = 8 ] && st=40
    python bench.py --config $cfg --chains-per-gpu $c --steps $st --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg $c chain(s):', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; k_gram', round(r['avg_launch_us'],1), 'us', round(r['achieved'],1), 'TFLOP/s frac', round(r['frac'],3), 'sweep_frac', round(r['sweep_frac'],3))" | tee -a $O/configs.txt || exit 1
  done
done
python bench.py --config cfg3 --chains-per-gpu 16 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg3 16 chains:', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; sweep_frac', round(r['sweep_frac'],3))" | tee -a $O/configs.txt
echo "== 4. config 5, byte image on / off" | tee -a $O/progress.log
cd /tmp
for b in 0 1; do
  rm -rf $O/prof_bytex$b
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bytex$b -o bx$b -- python3 $R/tools/byte_x_prof.py $b > $O/prof_bytex$b.log 2>&1 || { echo "rocprofv3 byte_x=$b failed"; exit 1; }
  echo "== byte_x=$b: $(tail -1 $O/prof_bytex$b.log)" | tee -a $O/cfg5_roofline.txt
  python3 $R/tools/prof_summary.py $O/prof_bytex$b | grep -E "kernel|k_xpass|k_backproj|k_gram|k_chol|k_tail|k_node|total" | tee -a $O/cfg5_roofline.txt
done
python3 - <<PY | tee -a $O/cfg5_roofline.txt
# roofline of config 5's kernels from the two traces: algorithmic bytes per launch / average duration against 8 TB/s (HBM) and n^2 q flops against 78.6 TFLOP/s
import csv, glob
n, V, R = 500, 300, 10; q = V * (V + 1) // 2; n_pad = 512
for b in (0, 1):
    dur = {}
    for f in glob.glob("$O/prof_bytex%d/**/*kernel_trace.csv" % b, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("byte image %s:" % ("on" if b else "off"))
    bytes_x = n_pad * q * (1 if b else 8)
    for k, alg, unit in (("k_xpass", bytes_x, "B"), ("k_backproj", bytes_x, "B"), ("k_gram", float(n) * n * q, "F")):
        if k in dur:
            us = sum(dur[k]) / len(dur[k])
            if unit == "B": print("  %-11s %7.1f us per launch, reads X once = %6.1f MB -> %.2f TB/s = %.2f of the 8 TB/s HBM peak" % (k, us, alg / 1e6, alg / us / 1e6, alg / us / 1e6 / 8.0))
            else: print("  %-11s %7.1f us per launch, n^2 q = %.2f GFLOP -> %.1f TFLOP/s = %.2f of the 78.6 TFLOP/s f64 MFMA peak" % (k, us, alg / 1e9, alg / us / 1e6, alg / us / 1e6 / 78.6))
PY
echo "== done" | tee -a $O/progress.log
