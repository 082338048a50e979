import sys, time
import os; R=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..'); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import gtscript
import stencil_zoo as zoo
for name in ("laplacian","vertical_advection_dycore","horizontal_diffusion"):
    defn, ext, scal, opts = zoo.ZOO[name]
    for use_lib in ((True, False) if name!="vertical_advection_dycore" else (False,)):
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
