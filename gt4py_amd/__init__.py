"""gt4py_amd -- MI355X-native stencil execution behind the gt4py.cartesian API.

Only the hot path of gt4py.cartesian is implemented here: hand-written gfx950
kernels for the 5-point Laplacian, horizontal diffusion and the vertical tridiagonal solve, the
``hip:mi300`` backend that dispatches to them through a C ABI, the storage allocators and the
stencil call interface.
"""

__version__ = "0.1.0"
