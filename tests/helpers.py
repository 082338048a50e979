"""Array wrappers used by the call-interface tests: objects that only expose the array-interface
protocols plus ``__gt_dims__`` / ``__gt_origin__`` (same role as the wrappers in
/root/reference/tests/cartesian_tests/utils.py:23-61)."""

from typing import Any, Tuple


class ArrayWrapper:
    def __init__(self, array, **_kwargs: Any) -> None:
        self.array = array

    @property
    def __array_interface__(self):
        return self.array.__array_interface__

    @property
    def __cuda_array_interface__(self):
        return self.array.__cuda_array_interface__


class DimensionsWrapper(ArrayWrapper):
    def __init__(self, dimensions: Tuple[str, ...], **kwargs: Any) -> None:
        super().__init__(**kwargs)
        if len(self.array.shape) != len(dimensions):
            raise ValueError(f"Non matching dimensions of array.shape {self.array.shape} and dimensions {dimensions}.")
        self.__gt_dims__ = dimensions


class OriginWrapper(ArrayWrapper):
    def __init__(self, *, origin: Tuple[int, ...], **kwargs: Any) -> None:
        super().__init__(**kwargs)
        if len(self.array.shape) != len(origin):
            raise ValueError(f"Non matching dimensions of array.shape {self.array.shape} and origin {origin}.")
        self.__gt_origin__ = origin


class HostOnlyWrapper:
    """Exposes ONLY __array_interface__ (no __cuda_array_interface__), for host-side tests."""

    def __init__(self, array, origin=None, dimensions=None):
        self.array = array
        if origin is not None:
            self.__gt_origin__ = origin
        if dimensions is not None:
            self.__gt_dims__ = dimensions

    @property
    def __array_interface__(self):
        return self.array.__array_interface__
