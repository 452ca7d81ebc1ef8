"""Build libbnr_hip.so (hipcc, gfx950) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("BNR_HIP_LIB") or os.path.join(HERE, "libbnr_hip.so")     # BNR_HIP_LIB: another build of the library (the sanitizer build of tools/sanitize_cpu.sh; julia/BNRHip.jl honours the same variable)


def build(force=False, verbose=False):
    if os.environ.get("BNR_HIP_LIB"):
        return LIB
    csrc = os.path.join(HERE, "csrc")
    cmd = ["make", "-C", csrc] + (["-B"] if force else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode:
        print(r.stdout)
    if r.returncode:
        raise RuntimeError("building libbnr_hip.so failed")
    return LIB
