for c in 1 2 4 8; do python bench.py --steps 600 --warmup 48 --no-cpu-baseline --chains-per-gpu $c | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chains/gpu', d['config']['chains_per_gpu'], 'it/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'gram us %.1f' % d['roofline']['avg_launch_us'])"; done
