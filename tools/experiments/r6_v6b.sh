mkdir -p gpurun_out
{ timeout -k 10 300 python tools/experiments/r6_v6.py && \
  timeout -k 10 200 python tools/ab_opt.py 1 1000 500 100 7 -- default factor_variant=6 && \
  timeout -k 10 200 python tools/ab_opt.py 8 640 500 100 7 -- default factor_variant=6 ; } > gpurun_out/r6_v6b.log 2>&1; tail -30 gpurun_out/r6_v6b.log
