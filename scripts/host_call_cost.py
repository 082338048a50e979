import sys, time
import os; R=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..'); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import gtscript
import stencil_zoo as zoo
for name in ("laplacian","vertical_advection_dycore","horizontal_diffusion","two_stage_written_input","cross_column_recurrence"):
    defn, ext, scal, opts = zoo.ZOO[name]
    for use_lib in ((True, False) if name in ("laplacian", "horizontal_diffusion") else (False,)):
        obj = gtscript.stencil(backend="hip:mi300", definition=defn, externals=ext, device_sync=False, use_kernel_library=use_lib)
        domain=(16,16,8)
        arrays, origins = zoo.make_inputs(obj, domain)
        dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k,v in arrays.items()}
        frozen = obj.freeze(origin=origins, domain=domain)
        for _ in range(20): frozen(**dev, **scal)
        torch.cuda.synchronize()
        t=time.perf_counter(); n=2000
        for _ in range(n): frozen(**dev, **scal)
        t1=time.perf_counter(); torch.cuda.synchronize()
        print(f"{name:28s} {'library' if use_lib else 'generated':9s} frozen call host cost {(t1-t)/n*1e6:6.1f} us")
        t=time.perf_counter()
        for _ in range(n): obj(**dev, **scal, origin=origins, domain=domain)
        t1=time.perf_counter(); torch.cuda.synchronize()
        print(f"{name:28s} {'library' if use_lib else 'generated':9s} full   call host cost {(t1-t)/n*1e6:6.1f} us")

# launch-bound loop: 20 dependent small applies, eager vs one hipGraph replay
defn, ext, scal, opts = zoo.ZOO["laplacian"]
obj = gtscript.stencil(backend="hip:mi300", definition=defn, device_sync=False)
domain = (128, 128, 64)
shape = (130, 130, 64)
a = gt_storage.ones(shape, backend="hip:mi300", aligned_index=(1, 1, 0))
b = gt_storage.ones(shape, backend="hip:mi300", aligned_index=(1, 1, 0))
fr = obj.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=domain)


def loop():
    for _ in range(10):
        fr(inp=a, out=b)
        fr(inp=b, out=a)


loop(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): loop()
torch.cuda.synchronize()
eager = (time.perf_counter() - t) / 50
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loop()
g.replay(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t) / 50
print(f"20 dependent 128x128x64 Laplacian applies: eager {eager*1e6:7.1f} us, hipGraph replay {graph*1e6:7.1f} us")
