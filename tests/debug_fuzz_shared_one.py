"""One program of tests/fuzz_stencils.make_shared_temporaries_stencil on the GPU against the oracle (debugging aid):
    python tests/debug_fuzz_shared_one.py <seed> [ni nj nk]
(lives under tests/ because it uses the oracle as the checker)"""
import os, sys, tempfile, pathlib, warnings
warnings.simplefilter("ignore")
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))  # noqa: E702
import numpy as np
import fuzz_stencils, stencil_zoo as zoo
import oracle.numpy_backend  # noqa: F401  (debugging script: the oracle is the checker)
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import gtscript

seed = int(sys.argv[1])
domain = tuple(int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (130, 11, 2)
tmp = pathlib.Path(tempfile.mkdtemp())
defn, scalars, text = fuzz_stencils.make_shared_temporaries_stencil(seed, tmp)
ref = gtscript.stencil(backend="numpy", definition=defn)
hip = gtscript.stencil(backend="hip:mi300", definition=defn)
prog = type(hip)._gt_program_
arrays, origins = zoo.make_inputs(ref, domain, seed)
expect = {k: v.copy() for k, v in arrays.items()}
ref(**expect, **scalars, origin=origins, domain=domain)
dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
hip(**dev, **scalars, origin=origins, domain=domain)
env = {k: v for k, v in os.environ.items() if k.startswith("GT4MI_")}
for k in arrays:
    got = dev[k].get()
    bad = (got != expect[k]) & ~(np.isnan(got) & np.isnan(expect[k]))
    where = np.argwhere(bad)
    print(f"seed {seed} {domain} {env} field {k}: {int(bad.sum())} mismatches", "" if not bad.any() else f"first at {where[0].tolist()} i in [{where[:,0].min()},{where[:,0].max()}] j in [{where[:,1].min()},{where[:,1].max()}] k in {sorted(set(where[:,2].tolist()))}")
    if bad.any():
        for w in where[:8]:
            print("      at", w.tolist(), "got", got[tuple(w)], "want", expect[k][tuple(w)])
print("   stages:", [(s.mapping, s.extent, k.vec, k.shared_vec, k.shared_halo) for s, k in zip(prog.plan.stages, prog.kernels)])
