"""Backend plug-in interface and registry.

Same contract as /root/reference/src/gt4py/cartesian/backend/base.py:35-152: a backend is a class
with ``name``, ``options``, ``storage_info``, ``languages`` and ``generate()``; ``@register`` adds it
to ``REGISTRY`` and registers its ``storage_info`` with the storage layout registry so that
``storage.zeros(..., backend=name)`` works; ``from_name`` is what ``StencilObject`` calls at run
time.  There is no code generation here: ``generate()`` returns a ``StencilObject`` subclass built
in-process (the reference renders and imports a Python module, base.py:204-222).
"""

from __future__ import annotations

import abc
import time
import warnings
from typing import Any, ClassVar, Dict, Optional, Type

from .. import definitions as gt_definitions
from ...storage import layout as storage_layout


class Backend(abc.ABC):
    #: backend name, e.g. ``"hip:mi300"``
    name: ClassVar[str]
    #: accepted backend options: ``{name: {"versioning": bool, "type": type}}``
    options: ClassVar[Dict[str, Dict[str, Any]]]
    #: memory layout preset registered under the backend's name
    storage_info: ClassVar[storage_layout.LayoutInfo]
    #: languages of the computation / bindings (informational)
    languages: ClassVar[Optional[Dict[str, Any]]]

    def __init__(self, builder):
        self.builder = builder

    @abc.abstractmethod
    def generate(self) -> Type:
        """Build and return the StencilObject subclass for ``self.builder``."""

    def load(self) -> Optional[Type]:
        """Cached implementation, if any (in-process cache lives in the loader)."""
        return None

    @property
    def extra_cache_info(self) -> Dict[str, Any]:
        return {}

    @property
    def extra_cache_validation_keys(self):
        return []

    @classmethod
    def filter_options_for_id(cls, options: gt_definitions.BuildOptions) -> gt_definitions.BuildOptions:
        return options


class _Registry(dict):
    @property
    def names(self):
        return list(self.keys())

    def register(self, name: str, item):
        self[name] = item
        return item


REGISTRY = _Registry()


def from_name(name: str) -> Type[Backend]:
    backend = REGISTRY.get(name, None)
    if not backend:
        raise ValueError(f"Backend '{name}' is not registered. Valid options are: '{', '.join(REGISTRY.names)}'.")
    return backend


def register(backend_cls: Type[Backend]) -> Type[Backend]:
    assert issubclass(backend_cls, Backend) and backend_cls.name is not None
    if isinstance(backend_cls.name, str):
        storage_layout.register(backend_cls.name, backend_cls.storage_info)
        return REGISTRY.register(backend_cls.name, backend_cls)
    raise ValueError(f"Invalid 'name' attribute ('{backend_cls.name}') in backend class '{backend_cls}'")


class BaseBackend(Backend):
    """Shared option checking (base.py:194-202 of the reference: unknown options only warn)."""

    def check_options(self, options: gt_definitions.BuildOptions) -> None:
        assert self.options is not None
        unknown = set(options.backend_opts.keys()) - set(self.options.keys())
        if unknown:
            warnings.warn(
                f"Unknown options '{unknown}' for backend '{self.name}'", RuntimeWarning, stacklevel=2
            )

    def generate(self) -> Type:
        self.check_options(self.builder.options)
        start = time.perf_counter()
        cls = self.make_stencil_class()
        info = self.builder.options.build_info
        if info is not None:
            info["module_time"] = time.perf_counter() - start
        return cls

    @abc.abstractmethod
    def make_stencil_class(self) -> Type:
        ...
