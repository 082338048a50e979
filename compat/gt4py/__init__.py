"""Import-path shim: lets code written against the reference's module names run on this repository.

    PYTHONPATH=<repo>:<repo>/compat python user_script.py

    import gt4py.storage as gt_storage
    from gt4py.cartesian import gtscript
    from gt4py.cartesian.gtscript import Field, PARALLEL, computation, interval

    @gtscript.stencil(backend="hip:mi300")
    def lap(...): ...

Only the surface this repository implements is aliased (gt4py.cartesian.{gtscript, definitions, backend,
stencil_object}, gt4py.storage).  It is NOT gt4py: `gt4py.next`, `gt4py.eve` and the other backends do not
exist here.  Kept outside the `gt4py_amd` package on purpose, so that it can never shadow a real gt4py
installation unless the user puts `compat/` on the path.
"""
import sys as _sys

import gt4py_amd as _impl
from gt4py_amd import cartesian, storage  # noqa: F401
from gt4py_amd.cartesian import backend as _backend, definitions as _definitions, gtscript as _gtscript, \
    stencil_object as _stencil_object

__version__ = "0+gt4py_amd"

for _name, _module in {
    "cartesian": cartesian,
    "cartesian.gtscript": _gtscript,
    "cartesian.definitions": _definitions,
    "cartesian.backend": _backend,
    "cartesian.stencil_object": _stencil_object,
    "storage": storage,
}.items():
    _sys.modules[f"{__name__}.{_name}"] = _module
