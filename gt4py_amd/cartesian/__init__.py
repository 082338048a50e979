"""gt4py_amd.cartesian -- the gt4py.cartesian hot path (stencil call interface + hip:mi300 backend)."""

from . import backend, definitions, gtscript
from .stencil_object import FrozenStencil, StencilObject


__all__ = ["FrozenStencil", "StencilObject", "backend", "definitions", "gtscript"]
