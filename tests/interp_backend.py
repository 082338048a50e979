"""Test-only backend "interp": the call interface of this repository (argument validation, origins, domains) in front of the INDEPENDENT
interpreter `oracle/gtscript_interp.py`, so that test files written against a `backend` fixture can run their definitions through it.
What EXECUTES is the interpreter on the definition's source; the product's IR is not consulted (its `field_info` still sizes the check
of the arguments).  `Unsupported` constructs skip the test."""

import numpy as np
import pytest

import oracle.numpy_backend as numpy_oracle
from gt4py_amd.cartesian.backend import base
from gt4py_amd.cartesian.stencil_object import StencilObject
from oracle import gtscript_interp as gi


class InterpreterStencilObject(StencilObject):
    def _run_implementation(self, domain, origin, exec_info, arguments):
        cls = type(self)
        fields = {}
        for n in cls._gt_field_info_:
            a = arguments.get(n)
            if a is not None:
                fields[n] = (np.asarray(a), tuple(origin[n]))
        params = {n: arguments.get(n) for n in cls._gt_parameter_info_ if arguments.get(n) is not None}
        opts = cls._gt_options_
        try:
            gi.run(cls.definition_func, fields, params, domain, externals=cls._gt_constants_,
                   literal_int=opts.get("literal_int_precision", 64), literal_float=opts.get("literal_float_precision", 64),
                   while_semantics="pointwise" if opts.get("backend_opts", {}).get("while_loops") == "pointwise" else "numpy")
        except gi.Unsupported as ex:
            pytest.skip(f"the independent interpreter does not restate this: {ex}")


class InterpreterBackend(numpy_oracle.NumpyOracleBackend):
    name = "interp"

    def make_stencil_class(self):
        cls = super().make_stencil_class()
        return type(cls.__name__, (InterpreterStencilObject,), {k: v for k, v in vars(cls).items() if not k.startswith("__") or k == "__module__"})


if "interp" not in base.REGISTRY:
    base.register(InterpreterBackend)
