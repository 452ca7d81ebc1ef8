export BNR_HIP_LIB=$GRAFT_REPO_ROOT/_stamps/libbnr_hip.so
mkdir -p gpurun_out
{ timeout -k 10 120 python tools/stamps_steps.py 200 50 5 && timeout -k 10 120 python tools/stamps_steps.py 100 30 5 && timeout -k 10 120 python tools/stamps_steps.py 500 100 7 ; } > gpurun_out/r6_steps.log 2>&1
cat gpurun_out/r6_steps.log
