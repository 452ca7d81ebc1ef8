#!/bin/bash
# usage: tools/ab_big_groups.sh  -- the built library against tools/_ab/libbnr_base.so at the large group configurations (BNR_HIP_LIB), interleaved twice
for v in base new base new; do if [ $v = base ]; then export BNR_HIP_LIB=$GRAFT_REPO_ROOT/tools/_ab/libbnr_base.so; else unset BNR_HIP_LIB; fi
echo -n "$v cfg5 x8 real: "; python bench.py --config cfg5 --chains-per-gpu 8 --steps 100 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value']))"
echo -n "$v cfg5 x8 bool: "; python bench.py --config cfg5 --chains-per-gpu 8 --steps 100 --warmup 16 --no-cpu-baseline --binary-x 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value']))"
echo -n "$v cfg4 x8 real: "; python bench.py --config cfg4 --chains-per-gpu 8 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value'],1))"
echo -n "$v cfg3 x16 real: "; python bench.py --config cfg3 --chains-per-gpu 16 --steps 200 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value'],1))"
done
