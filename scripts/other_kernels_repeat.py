import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
torch.cuda.set_device(0)
for label, pre in (("cold", 0.0), ("again", 0.0), ("after 2 s idle", 2.0)):
    time.sleep(pre)
    o = bench.other_kernels()
    print(label, {k.split("_")[0] + k[-14:]: (v["ms"], v["ms_min"]) for k, v in o.items()}, flush=True)
print("copy", bench.copy_ceiling_gbs())
o = bench.other_kernels()
print("after copy", {k.split("_")[0] + k[-14:]: (v["ms"], v["ms_min"]) for k, v in o.items()}, flush=True)
