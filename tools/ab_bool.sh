for v in base new base new; do if [ $v = base ]; then export BNR_HIP_LIB=$GRAFT_REPO_ROOT/tools/_ab/libbnr_base.so; else unset BNR_HIP_LIB; fi
echo -n "$v headline bool x8: "; python bench.py --binary-x --steps 640 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value']), 'single', round(d['single_chain']['value']))"
echo -n "$v cfg5 bool x8: "; python bench.py --config cfg5 --chains-per-gpu 8 --steps 100 --warmup 16 --no-cpu-baseline --binary-x 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value']))"
echo -n "$v cfg5 bool x1: "; python bench.py --config cfg5 --chains-per-gpu 1 --steps 200 --warmup 16 --no-cpu-baseline --binary-x 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), round(d['value']))"
done
