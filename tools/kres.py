#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel: tools/kres.py [extra hipcc flags]  (cross-compiles on the CPU, no GPU needed)"""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "bayesiannetworkregression.jl_amd", "csrc", "bnr_hip.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",
       "-Rpass-analysis=kernel-resource-usage", "-shared", "-o", "/tmp/kres.so", src] + sys.argv[1:]
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^ ]* *(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (.*?) \[-Rpass", line)
    if not m:
        m2 = re.search(r"(Function Name|    VGPRs|    ScratchSize \[bytes/lane\]|    Occupancy \[waves/SIMD\]|    LDS Size \[bytes/block\]|    SGPRs): (\S+)", line)
        if not m2: continue
        k, v = m2.group(1).strip(), m2.group(2)
    else:
        k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
    print("%-70s vgpr %4s sgpr %4s scratch %4s occ %2s lds %6s" % (name[:70], c.get("VGPRs"), c.get("SGPRs"), c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
