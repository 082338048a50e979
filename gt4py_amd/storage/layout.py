"""Memory-layout presets per backend name.

Counterpart of /root/reference/src/gt4py/storage/cartesian/layout.py:19-76 (LayoutInfo,
layout_maker_factory, layout checker) and layout_registry.py:16-122 (REGISTRY, from_name,
register).  A layout map assigns each dimension its rank in memory: the dimension holding the
largest value is the contiguous one (``(2, 1, 0)`` for I, J, K = I-contiguous).
"""

from __future__ import annotations

from typing import Any, Callable, Dict, Literal, Optional, Sequence, Tuple, TypedDict

import numpy as np


class LayoutInfo(TypedDict):
    alignment: int  # in ITEMS of the array dtype (byte alignment = alignment * itemsize)
    device: Literal["cpu", "gpu"]
    layout_map: Callable[[Tuple[str, ...]], Tuple[int, ...]]
    is_optimal_layout: Callable[[Any, Tuple[str, ...]], bool]


def layout_maker_factory(base_layout: Tuple[int, ...]) -> Callable[[Tuple[str, ...]], Tuple[int, ...]]:
    """Layout-map builder for a cartesian base layout given for (I, J, K).

    Missing cartesian axes are dropped; data dimensions ("0", "1", ...) always get the largest
    strides, in order (layout.py:28-57 of the reference; pinned by test_layout.py:16-131).
    """

    def layout_maker(dimensions: Tuple[str, ...]) -> Tuple[int, ...]:
        present = [axis in dimensions for axis in "IJK"]
        n_data = len(dimensions) - sum(present)
        # sort key per dimension, in the canonical order I, J, K, data0, data1, ...
        keys = [n_data + rank for rank, here in zip(base_layout, present) if here]
        keys += list(range(n_data))
        order = sorted(range(len(keys)), key=lambda d: keys[d])
        layout = [0] * len(keys)
        for position, dim in enumerate(order):
            layout[dim] = position
        return tuple(layout)

    return layout_maker


def check_layout(layout_map: Sequence[int], strides: Sequence[int]) -> bool:
    """True when ``strides`` do not increase while walking the dims from outermost to innermost."""
    if len(strides) != len(layout_map):
        return False
    previous = 0
    for dim in reversed(np.argsort(layout_map)):
        if strides[dim] < previous:
            return False
        previous = strides[dim]
    return True


def layout_checker_factory(layout_maker) -> Callable[[Any, Tuple[str, ...]], bool]:
    def layout_checker(field, dimensions: Tuple[str, ...]) -> bool:
        return check_layout(layout_maker(dimensions), field.strides)

    return layout_checker


REGISTRY: Dict[str, LayoutInfo] = {}


def from_name(name: str) -> LayoutInfo:
    info = REGISTRY.get(name, None)
    if info is None:
        raise ValueError(f"Layout '{name} is not registered. Valid options are: {REGISTRY.keys()}.")
    return info


def register(name: str, info: Optional[LayoutInfo]) -> None:
    if info is None:
        REGISTRY.pop(name, None)
        return
    assert isinstance(name, str) and isinstance(info, dict)
    REGISTRY[name] = info


def _preset(base_layout, *, alignment: int, device: str) -> LayoutInfo:
    maker = layout_maker_factory(base_layout)
    return LayoutInfo(alignment=alignment, device=device, layout_map=maker,
                      is_optimal_layout=layout_checker_factory(maker))


# Host presets the reference's tests allocate with (layout_registry.py:87-121).  They carry no
# compute backend here; they only describe memory.
register("numpy", _preset((0, 1, 2), alignment=1, device="cpu"))
register("gt:cpu_kfirst", _preset((0, 1, 2), alignment=1, device="cpu"))
register("gt:cpu_ifirst", _preset((2, 1, 0), alignment=1, device="cpu"))

# The MI355X backend: I-contiguous like gt:gpu (layout_registry.py:105-112), rows padded to -- and the aligned_index column
# placed on -- a 128-BYTE boundary, one L2 line, whatever the item size (16 fp64 / 32 fp32 items: `alignment` below is the fp64
# figure, `alignment_bytes` what the allocator uses).  Rounds 1-3 used gt:gpu's 32 items (256 B for fp64): a row of 130 items then
# occupies 160, and a fifth of every DRAM page of a 128-column local domain is padding.  Measured through bench.py on one box
# (profiles/r4_row_alignment.txt), 256 -> 128 bytes: the 128 x 256 x 512 share of a 4 x 2 decomposition 51.6 -> 48.7 us, 512^3
# +1 %, 512 x 512 x 128 +2 %, hdiff fp64 +1 %; the column kernels unchanged.  64 bytes is better still for the 16-byte-lane
# kernels (the narrow share: 45.6 us) but rows that straddle L2 lines cost the 8-byte-lane column kernels a fifth (generated
# vertical advection 93 -> 78 GLUPS).  GT4PY_AMD_ROW_ALIGN_BYTES overrides (256: the old preset), for A/B runs.
import os as _os

HIP_MI300_ROW_ALIGN_BYTES = int(_os.environ.get("GT4PY_AMD_ROW_ALIGN_BYTES", "128"))
HIP_MI300_LAYOUT = _preset((2, 1, 0), alignment=max(HIP_MI300_ROW_ALIGN_BYTES // 8, 1), device="gpu")
HIP_MI300_LAYOUT["alignment_bytes"] = HIP_MI300_ROW_ALIGN_BYTES  # type: ignore[typeddict-unknown-key]
register("hip:mi300", HIP_MI300_LAYOUT)
