"""Build orchestration: definition function -> stencil object (singleton instance).

Counterpart of /root/reference/src/gt4py/cartesian/loader.py:30-69 and stencil_builder.py:27-120.
The reference fingerprints the canonical AST + externals + options into an on-disk ``.gt_cache``
(caching.py:300-326); nothing is generated or compiled here, so an in-process dict keyed by the
same ingredients is the whole cache.
"""

from __future__ import annotations

import hashlib
import inspect
import time
from dataclasses import dataclass
from typing import Any, Dict

from . import analysis, backend as gt_backend, definitions as gt_definitions, frontend, ir


@dataclass
class StencilBuilder:
    """What a backend needs to build an implementation (subset of stencil_builder.py:27-120)."""

    definition: Any
    backend_name: str
    options: gt_definitions.BuildOptions
    externals: Dict[str, Any]
    dtypes: Dict[Any, Any]
    stencil_ir: ir.Stencil
    args_data: analysis.ArgsData
    stencil_id: str
    source: str

    @property
    def class_name(self) -> str:
        return f"{self.options.name}__{self.backend_name.replace(':', '_')}_{self.stencil_id[:10]}"


_CLASS_CACHE: Dict[str, type] = {}


def _fingerprint(definition, backend: str, options: gt_definitions.BuildOptions, externals, dtypes, source: str) -> str:
    h = hashlib.sha256()
    h.update(source.encode())
    h.update(repr(sorted((str(k), repr(v)) for k, v in externals.items())).encode())
    h.update(repr(sorted((str(k), str(v)) for k, v in dtypes.items())).encode())
    h.update(repr((backend, options.qualified_name, options.literal_int_precision,
                   options.literal_float_precision, sorted(options.backend_opts.items()))).encode())
    # annotations may depend on globals that are invisible in the source text (e.g. `dtype`)
    h.update(repr(sorted((k, repr(v)) for k, v in getattr(definition, "__annotations__", {}).items())).encode())
    return h.hexdigest()


def load_stencil(frontend_name: str, backend_name: str, definition, externals, dtypes,
                 build_options: gt_definitions.BuildOptions):
    backend_cls = gt_backend.from_name(backend_name)
    try:
        source = inspect.getsource(definition)
    except (OSError, TypeError) as ex:
        raise gt_definitions.GTScriptDefinitionError(
            f"Cannot retrieve the source of '{getattr(definition, '__name__', definition)}'") from ex
    stencil_id = _fingerprint(definition, backend_name, build_options, externals, dtypes, source)
    if not build_options.rebuild and stencil_id in _CLASS_CACHE:
        return _CLASS_CACHE[stencil_id]

    t0 = time.perf_counter()
    stencil_ir = frontend.parse_stencil(definition, externals=externals, dtypes=dtypes, options=build_options)
    args_data = analysis.make_args_data(stencil_ir)
    if build_options.build_info is not None:
        build_options.build_info["parse_time"] = time.perf_counter() - t0
    builder = StencilBuilder(definition, backend_name, build_options, dict(externals), dict(dtypes),
                             stencil_ir, args_data, stencil_id, source)
    stencil_class = backend_cls(builder).generate()
    _CLASS_CACHE[stencil_id] = stencil_class
    return stencil_class


def gtscript_loader(definition_func, backend, build_options, externals, dtypes):
    if not isinstance(definition_func, type(lambda: None)):
        raise ValueError("Invalid stencil definition object ({obj})".format(obj=definition_func))
    if not build_options.name:
        build_options.name = f"{definition_func.__name__}"
    stencil_class = load_stencil("gtscript", backend, definition_func, externals, dtypes, build_options)
    return stencil_class()
