"""Padded, aligned, layout-permuted buffer allocation.

Restates the allocation arithmetic of
/root/reference/src/gt4py/storage/allocators.py:187-273 (``_BaseNDArrayBufferAllocator.allocate``):

* the stride-1 dimension (the one with the largest layout value) is padded to a multiple of the
  alignment, in items;
* strides are C-order over the layout-permuted, padded shape;
* the view starts at a byte offset chosen so that element ``aligned_index`` sits on a
  ``byte_alignment`` boundary; ``byte_alignment - 1`` extra bytes are allocated to allow that.

CPU buffers are numpy arrays, GPU buffers are ``DeviceArray`` views of one PyTorch-ROCm allocation
(the reference uses ``cupy.empty`` + ``as_strided``, allocators.py:314-324, 368-370).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from .device_array import DeviceArray, torch, torch_dtype


@dataclass(frozen=True)
class BufferPlan:
    shape: Tuple[int, ...]
    padded_shape: Tuple[int, ...]
    strides: Tuple[int, ...]  # bytes
    total_bytes: int
    aligned_index_offset: int  # bytes, before correcting for the allocation's own misalignment
    byte_alignment: int
    itemsize: int

    def byte_offset(self, base_address: int) -> int:
        mismatch = (self.byte_alignment - base_address % self.byte_alignment) % self.byte_alignment
        return (self.aligned_index_offset + mismatch) % self.byte_alignment


def is_valid_layout_map(layout_map: Sequence[int]) -> bool:
    return sorted(layout_map) == list(range(len(layout_map)))


def plan_buffer(shape: Sequence[int], dtype: np.dtype, layout_map: Sequence[int], byte_alignment: int,
                aligned_index: Optional[Sequence[int]]) -> BufferPlan:
    shape = tuple(int(s) for s in shape)
    ndim = len(shape)
    if any(s < 0 for s in shape):
        raise ValueError(f"Invalid shape {shape}")
    if len(layout_map) != ndim or not is_valid_layout_map(layout_map):
        raise ValueError(f"Invalid layout_map {layout_map} for shape {shape}")
    itemsize = np.dtype(dtype).itemsize
    if math.gcd(byte_alignment, itemsize) not in (byte_alignment, itemsize):
        raise ValueError(f"Incompatible 'byte_alignment' ({byte_alignment}) and 'dtype' size ({itemsize})")
    items_per_block = (byte_alignment // itemsize) or 1

    # dims ordered from outermost (layout value 0) to innermost (layout value ndim-1)
    order = [list(layout_map).index(rank) for rank in range(ndim)]
    padded = list(shape)
    if ndim:
        inner = order[-1]
        padded[inner] = math.ceil(shape[inner] / items_per_block) * items_per_block
    strides = [itemsize] * ndim
    running = itemsize
    for pos in range(ndim - 2, -1, -1):
        running *= padded[order[pos + 1]]
        strides[order[pos]] = running
    total = itemsize * math.prod(padded) + (byte_alignment - 1)

    aligned_index = tuple(int(i) for i in (aligned_index or [0] * ndim))
    if ndim:
        ai = aligned_index[order[-1]]
        ai_offset = (items_per_block * math.ceil(ai / items_per_block) - ai) * itemsize
    else:
        ai_offset = 0
    return BufferPlan(shape, tuple(padded), tuple(strides), total, ai_offset, byte_alignment, itemsize)


def allocate_cpu(shape, layout_map, dtype, alignment_bytes, aligned_index) -> Tuple[np.ndarray, np.ndarray]:
    dtype = np.dtype(dtype)
    plan = plan_buffer(shape, dtype, layout_map, alignment_bytes, aligned_index)
    raw = np.empty((plan.total_bytes,), dtype=np.uint8)
    offset = plan.byte_offset(raw.ctypes.data)
    flat = raw[offset: offset + math.prod(plan.padded_shape) * plan.itemsize].view(dtype)
    view = np.lib.stride_tricks.as_strided(flat, shape=plan.padded_shape, strides=plan.strides)
    if plan.shape and plan.shape != plan.padded_shape:
        view = view[tuple(slice(0, s) for s in plan.shape)]
    return raw, view


#: Placement of big device buffers: period and step of the address pattern, smallest buffer it applies to.
PLACEMENT_PERIOD = 4 << 20
PLACEMENT_STEP = 1 << 20
PLACEMENT_MIN_BYTES = 64 << 20
_PLACEMENT_SLOT = [0]


def _placement_shift(base_address: int, alignment_bytes: int) -> int:
    """Bytes to skip at the start of a big device allocation so that the buffer begins at
    ``slot * 1 MiB (mod 4 MiB)``, ``slot`` = 1, 2, 3, 0, 1, ... for successive allocations.

    Equally shaped fields that a column kernel streams side by side (five 1.34 GB arrays in the tridiagonal
    solve, K planes exactly 8 MiB apart) are slowest when they sit at the same address modulo 4 MiB and
    fastest when address bits 20 and 21 differ between them: measured on MI355X with the fields at controlled
    offsets inside one allocation (profiles/r2_tridiag_placement_study.log) 79 GLUPS with offsets that are
    multiples of 2 MiB, 83-84 with 0.25-0.75 MiB steps, 85 with 1.5 MiB, 86.6 with 1 / 3 / 5 MiB steps; the
    pattern repeats with a period of 2 MiB in the step.  The shift is DERIVED FROM THE ADDRESS the allocator
    returned, so where the caching allocator happens to put a block does not matter (round 1 added a rotating
    offset to whatever address came back, which only works when all blocks start equally aligned).  The layout
    contract is untouched: strides, padding and the alignment of `aligned_index` are the reference's
    (storage/allocators.py:187-273); only where the buffer begins inside its (over-)allocation changes.
    ``GT4PY_AMD_ALLOC_SKEW_BYTES`` overrides the step (0 = off)."""
    import os

    step = int(os.environ.get("GT4PY_AMD_ALLOC_SKEW_BYTES", str(PLACEMENT_STEP)))
    if step <= 0:
        return 0
    step = -(-step // alignment_bytes) * alignment_bytes
    _PLACEMENT_SLOT[0] = (_PLACEMENT_SLOT[0] + 1) % 4
    want = (_PLACEMENT_SLOT[0] * step) % PLACEMENT_PERIOD
    return (want - base_address) % PLACEMENT_PERIOD


def allocate_gpu(shape, layout_map, dtype, alignment_bytes, aligned_index) -> Tuple["torch.Tensor", DeviceArray]:
    if torch is None or not torch.cuda.is_available():
        raise RuntimeError(
            "GPU allocation requested but no ROCm device is visible to PyTorch "
            "(gt4py_amd.storage(backend='hip:mi300') needs an MI355X)"
        )
    dtype = np.dtype(dtype)
    tdt = torch_dtype(dtype)
    plan = plan_buffer(shape, dtype, layout_map, alignment_bytes, aligned_index)
    if plan.total_bytes >= PLACEMENT_MIN_BYTES:
        # big fields are dealt over the device's memory groups (placement.py: two fields a kernel streams side by side are up to
        # 15 % faster when they do NOT share a group of memory channels); sizes the placer does not handle come back unclassified
        from . import placement

        placer = placement.device_placer()
        if placer is not None:
            raw, _ = placer.place(plan.total_bytes + PLACEMENT_PERIOD)
        else:
            raw = torch.empty((plan.total_bytes + PLACEMENT_PERIOD,), dtype=torch.uint8, device="cuda")
        skew = _placement_shift(raw.data_ptr(), alignment_bytes)
    else:
        raw = torch.empty((plan.total_bytes,), dtype=torch.uint8, device="cuda")
        skew = 0
    offset = skew + plan.byte_offset(raw.data_ptr() + skew)
    assert offset % plan.itemsize == 0, "device allocation is not item-aligned"
    n_items = math.prod(plan.padded_shape)
    flat = raw[offset: offset + n_items * plan.itemsize].view(tdt)
    view = torch.as_strided(flat, plan.padded_shape, tuple(s // plan.itemsize for s in plan.strides))
    if plan.shape and plan.shape != plan.padded_shape:
        view = view[tuple(slice(0, s) for s in plan.shape)]
    return raw, DeviceArray(view, owner=raw)
