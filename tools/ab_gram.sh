#!/bin/bash
# A/B of the two Gram kernels over the BASELINE configs (BNR_GRAM_VARIANT=16: k_gram, 8: k_gram8)
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,1), 'us; gram', round(r['avg_launch_us'],1), round(r['avg_launch_us_two_branch_schedule'],1), 'frac', round(r['frac'],3))"; }
for v in 16 8; do
  export BNR_GRAM_VARIANT=$v
  echo "variant $v cfg3 x8 : $(one --steps 640 --warmup 64)"
  echo "variant $v cfg3 x16: $(one --steps 320 --warmup 32 --chains-per-gpu 16)"
  echo "variant $v cfg4 x1 : $(one --config cfg4 --chains-per-gpu 1 --steps 100 --warmup 16)"
  echo "variant $v cfg5 x1 : $(one --config cfg5 --chains-per-gpu 1 --steps 200 --warmup 16)"
  echo "variant $v cfg5 x8 : $(one --config cfg5 --chains-per-gpu 8 --steps 40 --warmup 8)"
  echo "variant $v cfg2 x1 : $(one --config cfg2 --chains-per-gpu 1 --steps 400 --warmup 40)"
done
