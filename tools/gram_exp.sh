for e in 1 2 3; do
  cp bayesiannetworkregression.jl_amd/libbnr_exp$e.so bayesiannetworkregression.jl_amd/libbnr_hip.so
  echo "EXP $e (1 = no MFMA, 2 = no global loads in loop, 3 = one batch only)"
  python - <<'PY'
import sys; sys.path.insert(0,'.')
import bnr_amd, numpy as np
X,y,_=bnr_amd.make_synthetic(500,100,7,seed=20240501)
ch=bnr_amd.Chain(X,y,7,300,1,1); ch.init_prior()
try:
    ch.run(2,300,100)
except Exception as e: print('   (run error expected in experiments:', str(e)[:60], ')')
ch.set_profiling(True)
try:
    ch.run(101,300,200)
except Exception as e: pass
print('   gram us', ch.last_timing(1))
PY
done
