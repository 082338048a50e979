"""The `compat/gt4py` shim: reference-style imports resolve to this repository's modules."""

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
import numpy as np
import gt4py.storage as gt_storage
from gt4py.cartesian import gtscript
from gt4py.cartesian.gtscript import Field, PARALLEL, computation, interval
import gt4py.cartesian.backend as gt_backend

assert "hip:mi300" in gt_backend.REGISTRY.names

@gtscript.stencil(backend=BACKEND)
def lap(inp: Field[np.float64], out: Field[np.float64]):
    with computation(PARALLEL), interval(...):
        out = -4.0 * inp[0, 0, 0] + inp[-1, 0, 0] + inp[1, 0, 0] + inp[0, -1, 0] + inp[0, 1, 0]

nx = ny = 32
inp = gt_storage.from_array(np.fromfunction(lambda x, y, z: x**2 + y**2, (nx, ny, 1)), backend=BACKEND,
                            aligned_index=(1, 1, 0))
out = gt_storage.zeros((nx, ny, 1), backend=BACKEND, aligned_index=(1, 1, 0))
lap(inp=inp, out=out, origin=(1, 1, 0), domain=(nx - 2, ny - 2, 1))
res = np.asarray(out.get() if hasattr(out, "get") else out)
assert (res[1:-1, 1:-1] == 4.0).all() and res.sum() == 4.0 * 30 * 30
print("ok", lap.backend)
'''


def _run(backend, prelude="", tmp_path=None):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "compat")]))
    script = tmp_path / "user_script.py"  # gtscript reads the definition's source: it has to be a file
    script.write_text(prelude + f"BACKEND = {backend!r}\n" + SNIPPET)
    return subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)


def test_reference_style_imports_on_the_cpu_oracle_backend(tmp_path):
    """The notebook's Laplacian (examples/lap_cartesian_vs_next.ipynb cells 5-9) with gt4py.* imports."""
    r = _run("numpy", "import sys; sys.path.insert(0, %r); import oracle.numpy_backend\n" % ROOT, tmp_path)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok numpy"), r.stderr[-2000:]


@pytest.mark.gpu
def test_reference_style_imports_on_hip_mi300(tmp_path):
    r = _run("hip:mi300", tmp_path=tmp_path)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok hip:mi300"), r.stderr[-2000:]
