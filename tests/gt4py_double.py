"""A DOUBLE of the corner of gt4py that ``gt4py_amd.adapter.register_with_gt4py`` / ``_wrap_for_gt4py`` touch, so that those ~40 lines
EXECUTE somewhere (missing 2: "have never executed anywhere").

It is test infrastructure of this repository, written from the reference's INTERFACE, not from its code: every class and module below
offers exactly the names the pinned API surface lists for its counterpart (``tests/golden/gt4py_api_surface.json``, written from the
reference's sources by ``scripts/make_gt4py_api_surface.py``) -- ``check_against_surface`` refuses a double that offers a public name
the reference does not have -- with the little behaviour the glue relies on:

* ``backend.register`` / ``REGISTRY`` / ``from_name``, ``base.BaseBackend(builder)`` with ``check_options``;
* ``gtc.gtir_to_oir.GTIRToOIR().visit(gtir)`` -> the OIR (here: the hand-built, schema-checked trees of tests/test_adapter.py);
* ``backend.module_generator.make_args_data_from_gtir(pipeline)`` -> an object with ``field_info`` / ``parameter_info`` / ``domain_info``;
* ``stencil_object.StencilObject``: one instance per class, ``_call_run(field_args, parameter_args, domain, origin, *, validate_args,
  exec_info)`` that fills in origin and domain and calls ``self.run(_domain_, _origin_, exec_info, **fields, **params)``;
* a ``StencilBuilder`` with the properties the glue reads.

What it is NOT: gt4py.  It shows that the glue runs against objects shaped like gt4py's; the real thing needs Python >= 3.12."""

from __future__ import annotations

import dataclasses
import sys
import types
from typing import Any, Dict


def build(surface: Dict[str, Any], args_data_of_pipeline):
    """-> {module name: module}.  ``args_data_of_pipeline(pipeline)`` makes the ModuleData double (the test supplies this
    repository's own analysis of the translated IR)."""
    mods: Dict[str, types.ModuleType] = {}

    def module(name):
        mods[name] = types.ModuleType(name)
        return mods[name]

    # ---- gt4py.cartesian.backend.base ------------------------------------------------------------------------------------------
    base = module("gt4py.cartesian.backend.base")

    class Backend:
        name = None
        options: Dict[str, Any] = {}
        storage_info = None
        languages = None

        def __init__(self, builder):
            self.builder = builder

        def load(self):
            raise NotImplementedError

        def generate(self):
            raise NotImplementedError

        @classmethod
        def filter_options_for_id(cls, options):
            return options

    class BaseBackend(Backend):
        MODULE_GENERATOR_CLASS = None

        def check_options(self, options):
            unknown = [k for k in options.backend_opts if k not in self.options and not k.startswith("_")]
            if unknown:
                import warnings

                warnings.warn(f"Unknown option(s) {unknown} for backend '{self.name}'", RuntimeWarning)

        def _load(self):
            raise NotImplementedError

        def make_module(self):
            raise NotImplementedError

        def make_module_source(self, *, args_data=None):
            raise NotImplementedError

    REGISTRY: Dict[str, Any] = {}

    def register(backend_cls):
        assert issubclass(backend_cls, Backend) and isinstance(backend_cls.name, str)
        REGISTRY[backend_cls.name] = backend_cls
        return backend_cls

    def from_name(name):
        return REGISTRY[name]

    base.Backend, base.BaseBackend, base.REGISTRY, base.register, base.from_name = Backend, BaseBackend, REGISTRY, register, from_name
    backend_pkg = module("gt4py.cartesian.backend")
    backend_pkg.base = base
    for n in ("Backend", "BaseBackend", "REGISTRY", "register", "from_name"):
        setattr(backend_pkg, n, getattr(base, n))

    # ---- gt4py.cartesian.backend.module_generator ---------------------------------------------------------------------------------
    modgen = module("gt4py.cartesian.backend.module_generator")

    @dataclasses.dataclass
    class ModuleData:
        domain_info: Any = None
        field_info: Dict[str, Any] = dataclasses.field(default_factory=dict)
        parameter_info: Dict[str, Any] = dataclasses.field(default_factory=dict)
        unreferenced: list = dataclasses.field(default_factory=list)

    def make_args_data_from_gtir(pipeline):
        a = args_data_of_pipeline(pipeline)
        return ModuleData(domain_info=a.domain_info, field_info=dict(a.field_info), parameter_info=dict(a.parameter_info))

    modgen.ModuleData, modgen.make_args_data_from_gtir = ModuleData, make_args_data_from_gtir
    backend_pkg.module_generator = modgen

    # ---- gt4py.cartesian.gtc ------------------------------------------------------------------------------------------------------------
    gtc = module("gt4py.cartesian.gtc")
    passes = module("gt4py.cartesian.gtc.passes")

    class OirPipeline:  # (a Protocol in the reference: anything with run(oir))
        def run(self, oir):
            return oir

    passes.OirPipeline = OirPipeline
    gtir_to_oir = module("gt4py.cartesian.gtc.gtir_to_oir")

    class GTIRToOIR:
        def visit(self, gtir, **kwargs):
            return gtir.oir  # (the double's GTIR is a token that carries the OIR the reference would lower it to)

    gtir_to_oir.GTIRToOIR = GTIRToOIR
    gtc.passes, gtc.gtir_to_oir = passes, gtir_to_oir

    # ---- gt4py.cartesian.stencil_object -----------------------------------------------------------------------------------------------
    so = module("gt4py.cartesian.stencil_object")

    class StencilObject:
        _gt_id_: str
        definition_func: Any

        def __new__(cls, *args, **kwargs):
            if cls.__dict__.get("_instance") is None:
                cls._instance = object.__new__(cls)
            return cls._instance

        def _call_run(self, field_args, parameter_args, domain, origin, *, validate_args=True, exec_info=None):
            """Origin per field (a 3-tuple applies to every field, a dict may name fields), the largest domain that fits all fields.
            Every field goes through the stand-in of ``cp.asarray`` FIRST, as the reference's ``_call_run`` does before ``run()`` sees
            anything (stencil_object.py:69-93 -> storage/cartesian/utils.py:176-215, device "gpu"): what reaches the backend's
            ``run`` is a cupy array, not the caller's object."""
            field_args = {n: cupy_like_asarray(a) for n, a in field_args.items()}
            names = list(field_args)
            if origin is None:
                origin = {n: (0, 0, 0) for n in names}
            elif not isinstance(origin, dict):
                origin = {n: tuple(origin) for n in names}
            else:
                origin = {n: tuple(origin.get(n, origin.get("_all_", (0, 0, 0)))) for n in names}
            if domain is None:
                domain = tuple(min(field_args[n].shape[a] - origin[n][a] - self.field_info[n].boundary[a][1] for n in names) for a in range(3))
            self.run(_domain_=tuple(domain), _origin_=origin, exec_info=exec_info, **field_args, **parameter_args)

    so.StencilObject = StencilObject
    cartesian = module("gt4py.cartesian")
    cartesian.backend, cartesian.gtc, cartesian.stencil_object = backend_pkg, gtc, so
    root = module("gt4py")
    root.cartesian = cartesian
    return mods


class CupyLikeArray:
    """What ``cupy.asarray(x)`` hands on, as far as a backend's ``run()`` may rely on it: ``shape`` / ``dtype`` / ``strides`` and
    ``__cuda_array_interface__`` (version 3, byte strides always spelled out, ``stream=1`` as cupy sets it on ROCm) -- and NOTHING of
    the object it was made from: no ``.tensor``, no ``data_ptr()``, no ``__dlpack__``, no origin attributes.  Keeps the source alive,
    as a cupy view of foreign memory does."""

    def __init__(self, source):
        import numpy as _np

        cai = dict(source.__cuda_array_interface__)
        self.shape = tuple(int(n) for n in cai["shape"])
        self.dtype = _np.dtype(cai["typestr"])
        strides = cai.get("strides")
        if strides is None:  # C-contiguous
            strides, run = [], self.dtype.itemsize
            for n in reversed(self.shape):
                strides.insert(0, run)
                run *= n
        self.strides = tuple(int(b) for b in strides)
        self._interface = {"shape": self.shape, "typestr": self.dtype.str, "data": (int(cai["data"][0]), False), "version": 3,
                           "strides": self.strides, "stream": 1}
        self._keepalive = source

    @property
    def __cuda_array_interface__(self):
        return dict(self._interface)


def cupy_like_asarray(array):
    """``storage_utils.asarray(array, device="gpu")`` of the reference with the stand-in for cupy; an object that exports no
    ``__cuda_array_interface__`` is refused (the GPU door of gt4py NEEDS cupy: storage/cartesian/utils.py:186-188, 262-264)."""
    if array is None or isinstance(array, CupyLikeArray):
        return array
    if not hasattr(array, "__cuda_array_interface__"):
        raise TypeError(f"cp.asarray stand-in: {type(array).__name__} exports no __cuda_array_interface__")
    return CupyLikeArray(array)


@dataclasses.dataclass
class BuildOptions:
    name: str
    module: str
    backend_opts: Dict[str, Any] = dataclasses.field(default_factory=dict)
    rebuild: bool = False
    format_source: bool = True
    build_info: Any = None

    def as_dict(self):
        return dataclasses.asdict(self)


@dataclasses.dataclass
class StencilID:
    qualified_name: str
    version: str


class GTIRToken:
    def __init__(self, oir):
        self.oir = oir


class StencilBuilder:
    """The properties of the reference's StencilBuilder that the glue reads (stencil_builder.py:183-296), as plain attributes."""

    def __init__(self, definition, oir, *, name, backend_opts=None, externals=None):
        self.definition = definition
        self.options = BuildOptions(name=name, module="tests.gt4py_double", backend_opts=dict(backend_opts or {}))
        self.externals = dict(externals or {})
        self.gtir = GTIRToken(oir)
        self.gtir_pipeline = GTIRToken(oir)
        self.stencil_id = StencilID(f"tests.gt4py_double.{name}", "0123456789abcdef")
        self.module_qualname = f"tests.gt4py_double.m_{name}__double"
        self.class_name = f"{name}__double"
        self.backend = None  # set by the test: builder.backend = BackendClass(builder), as StencilBuilder.__init__ does


def check_against_surface(mods, surface) -> None:
    """The double offers no public name the reference does not have (it may offer fewer)."""
    pairs = [("gt4py.cartesian.backend.base", "backend/base.py", ("Backend", "BaseBackend")),
             ("gt4py.cartesian.backend.module_generator", "backend/module_generator.py", ("ModuleData",)),
             ("gt4py.cartesian.stencil_object", "stencil_object.py", ("StencilObject",)),
             ("gt4py.cartesian.gtc.gtir_to_oir", "gtc/gtir_to_oir.py", ("GTIRToOIR",))]
    for mod_name, file, classes in pairs:
        ref = surface[file]
        for cname in classes:
            cls = getattr(mods[mod_name], cname)
            rc = ref["classes"][cname]
            allowed = set(rc["attributes"]) | set(rc["properties"]) | set(rc["methods"])
            for base_name in rc["bases"]:
                b = base_name.split("[")[0].split(".")[-1]
                if b in ref["classes"]:
                    bc = ref["classes"][b]
                    allowed |= set(bc["attributes"]) | set(bc["properties"]) | set(bc["methods"])
            offered = {n for n in vars(cls) if not n.startswith("__") and n not in ("_instance", "_abc_impl")}
            extra = offered - allowed - ({"visit"} if cname == "GTIRToOIR" else set())  # (visit: inherited from eve.NodeTranslator)
            assert not extra, f"the double's {cname} offers {sorted(extra)}, which the reference's class does not have"
    sb = surface["stencil_builder.py"]["classes"]["StencilBuilder"]
    builder_names = set(sb["properties"]) | set(sb["attributes"])
    probe = StencilBuilder(lambda: None, None, name="probe")
    assert {n for n in vars(probe)} <= builder_names, sorted({n for n in vars(probe)} - builder_names)
    opts = surface["definitions.py"]["classes"]["BuildOptions"]
    assert {f.name for f in dataclasses.fields(BuildOptions)} <= set(opts["attributes"]), "BuildOptions double"
    assert {f.name for f in dataclasses.fields(StencilID)} <= set(surface["definitions.py"]["classes"]["StencilID"]["attributes"])


def install(monkeypatch, mods) -> None:
    for name, mod in mods.items():
        monkeypatch.setitem(sys.modules, name, mod)
