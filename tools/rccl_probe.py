import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
import bnr_amd
if len(sys.argv) > 1 and sys.argv[1] == "torch_after":
    bnr_amd.device_count()
    import torch
    torch.cuda.init()
os.system("grep -E 'amdhip|rccl' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
print("devices", bnr_amd.device_count())
uid = bnr_amd.Comm.unique_id()
c = bnr_amd.Comm.rccl(uid, 0, 1, 0)
print(c.allgather(np.arange(4.0)))
c.close()
print("ok")
