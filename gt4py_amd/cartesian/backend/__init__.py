"""Backend registry of gt4py_amd.cartesian (mirrors gt4py.cartesian.backend's public names:
/root/reference/src/gt4py/cartesian/backend/__init__.py)."""

from .base import REGISTRY, Backend, BaseBackend, from_name, register
from .hip_backend import HipMI300Backend


__all__ = ["REGISTRY", "Backend", "BaseBackend", "HipMI300Backend", "from_name", "register"]
