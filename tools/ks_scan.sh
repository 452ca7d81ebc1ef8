#!/bin/bash
# usage: tools/ks_scan.sh <config> <chains> <steps> k1 k2 ...   (binary X; 0 = library default)
cfg=$1; c=$2; st=$3; shift 3
for k in "$@"; do
  if [ $k = 0 ]; then unset BNR_GRAM_KSPLIT; else export BNR_GRAM_KSPLIT=$k; fi
  python bench.py --config $cfg --chains-per-gpu $c --steps $st --warmup 16 --no-cpu-baseline --binary-x 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg $c chain(s) ksplit $k:', round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us/sweep; digits + Gram', round(r['avg_launch_us'],1), 'us')"
done
