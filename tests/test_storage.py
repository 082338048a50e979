"""gt4py_amd.storage on the CPU presets: layout tables, padding/alignment arithmetic, spec checks.

Modelled on /root/reference/tests/storage_tests/unit_tests/test_layout.py and test_interface.py
(``test_allocate_cpu`` :119-181, ``TestNormalizeStorageSpec`` :241-346).  The layout tables are
outputs of the reference's own layout.py (tests/golden/layout_tables.json, scripts/make_golden.py).
"""

import json
import pathlib

import numpy as np
import pytest

import gt4py_amd.storage as gt_storage
from gt4py_amd.storage import allocators, layout as gt_layout

GOLD = json.loads((pathlib.Path(__file__).parent / "golden" / "layout_tables.json").read_text())


@pytest.mark.parametrize("base", sorted(GOLD["layout_maker"]))
def test_layout_maker_matches_reference_tables(base):
    maker = gt_layout.layout_maker_factory(tuple(int(v) for v in base.split(",")))
    for dims, expected in GOLD["layout_maker"][base]:
        assert list(maker(tuple(dims))) == expected, (base, dims)


def test_layout_maker_spot_values():
    # hand-checked rows of the reference tables (test_layout.py:16-131)
    assert gt_layout.layout_maker_factory((0, 1, 2))(("I", "J", "K", "0", "1")) == (2, 3, 4, 0, 1)
    assert gt_layout.layout_maker_factory((2, 0, 1))(("I", "J", "K")) == (2, 0, 1)
    assert gt_layout.layout_maker_factory((2, 1, 0))(("I", "J", "K", "0")) == (3, 2, 1, 0)
    assert gt_layout.layout_maker_factory((2, 1, 0))(("J", "K")) == (1, 0)


def test_check_layout_matches_reference():
    for layout_map, strides, expected in GOLD["check_layout"]:
        assert gt_layout.check_layout(layout_map, strides) is expected


def test_registry_contents_and_errors():
    info = gt_layout.from_name("hip:mi300")
    # rows on a 128-byte boundary -- one L2 line -- whatever the item size (round 4: gt:gpu's 32 items left a fifth of a
    # 128-column local domain's DRAM pages to padding; profiles/r4_row_alignment.txt); `alignment` is the fp64 figure
    assert info["device"] == "gpu" and info["alignment"] == 16 and info["alignment_bytes"] == 128
    assert info["layout_map"](("I", "J", "K")) == (2, 1, 0)  # I contiguous, like gt:gpu
    assert gt_layout.from_name("numpy")["layout_map"](("I", "J", "K")) == (0, 1, 2)
    assert gt_layout.from_name("gt:cpu_ifirst")["layout_map"](("I", "J", "K")) == (2, 1, 0)
    with pytest.raises(ValueError):
        gt_layout.from_name("no-such-layout")
    with pytest.raises(RuntimeError, match="not registered"):
        gt_storage.empty((2, 2, 2), backend="no-such-backend")


def test_hip_mi300_rows_are_padded_to_whole_l2_lines_for_every_item_size():
    for dtype, items in ((np.float64, 16), (np.float32, 32)):
        a = gt_storage.empty((130, 7, 3), dtype, backend="numpy")  # (host preset: no padding at all)
        assert a.strides[0] == 7 * 3 * np.dtype(dtype).itemsize
        plan = allocators.plan_buffer((130, 7, 3), np.dtype(dtype), (2, 1, 0), 128, (1, 1, 0))
        assert plan.padded_shape[0] == -(-130 // items) * items and plan.strides[1] % 128 == 0 and plan.strides[2] % 128 == 0
        assert (plan.byte_offset(base_address=4096) + np.dtype(dtype).itemsize) % 128 == 0  # element [1, j, k] starts a line


def test_allocation_plan_worked_example():
    """zeros((516,516,128), float64, gt:gpu-like preset, aligned_index=(2,2,0))."""
    plan = allocators.plan_buffer((516, 516, 128), np.dtype("float64"), (2, 1, 0), 256, (2, 2, 0))
    assert plan.padded_shape == (544, 516, 128)
    assert plan.strides == (8, 4352, 2245632)
    assert plan.aligned_index_offset == 240
    assert plan.total_bytes == 8 * 544 * 516 * 128 + 255
    assert plan.byte_offset(base_address=4096) == 240
    assert plan.byte_offset(base_address=4096 + 8) == (240 + 248) % 256


@pytest.mark.parametrize("seed", range(25))
def test_allocate_cpu_properties(seed):
    """Buffer containment, alignment of the aligned_index column, shape (test_interface.py:119-181)."""
    rng = np.random.default_rng(seed)
    ndim = int(rng.integers(1, 5))
    shape = tuple(int(s) for s in rng.integers(1, 9, ndim))
    layout_map = tuple(int(v) for v in rng.permutation(ndim))
    dtype = np.dtype(rng.choice(["float64", "float32", "int32", "int8"]))
    align_items = int(rng.choice([1, 2, 4, 8, 32]))
    alignment_bytes = align_items * dtype.itemsize
    aligned_index = tuple(int(rng.integers(0, s)) for s in shape)
    raw, field = allocators.allocate_cpu(shape, layout_map, dtype, alignment_bytes, aligned_index)
    assert field.shape == shape and field.dtype == dtype
    lo, hi = np.lib.array_utils.byte_bounds(raw) if hasattr(np.lib, "array_utils") else np.byte_bounds(raw)
    flo, fhi = np.lib.array_utils.byte_bounds(field) if hasattr(np.lib, "array_utils") else np.byte_bounds(field)
    assert lo <= flo and fhi <= hi
    # every point that shares all coordinates with aligned_index except along the non-contiguous
    # axes is aligned: check the aligned_index element itself and a few columns
    inner = layout_map.index(max(layout_map))
    for _ in range(10):
        idx = [int(rng.integers(0, s)) for s in shape]
        idx[inner] = aligned_index[inner]
        addr = field.ctypes.data + sum(i * s for i, s in zip(idx, field.strides))
        assert addr % alignment_bytes == 0
    # strides follow the layout: contiguous axis has itemsize stride, order is monotone
    assert field.strides[inner] == dtype.itemsize
    assert gt_layout.check_layout(layout_map, field.strides)
    field[...] = 3  # writable everywhere without touching memory outside `raw`
    assert (field == 3).all()


@pytest.mark.parametrize("backend", ["numpy", "gt:cpu_ifirst", "gt:cpu_kfirst"])
def test_cpu_constructors(backend):
    z = gt_storage.zeros((3, 4, 5), np.float32, backend=backend, aligned_index=(1, 1, 0))
    assert isinstance(z, np.ndarray) and z.shape == (3, 4, 5) and z.dtype == np.float32 and (z == 0).all()
    o = gt_storage.ones((3, 4, 5), backend=backend)
    assert o.dtype == np.float64 and (o == 1).all()
    f = gt_storage.full((3, 4), 2.5, backend=backend, dimensions=["I", "K"])
    assert (f == 2.5).all()
    a = np.arange(24.0).reshape(2, 3, 4)
    c = gt_storage.from_array(a, backend=backend)
    assert np.array_equal(c, a) and c.ctypes.data != a.ctypes.data
    info = gt_layout.from_name(backend)
    assert info["is_optimal_layout"](c, ("I", "J", "K"))
    # numpy integer scalars are accepted for shape and aligned_index (test_call_interface.py:288-311)
    s = gt_storage.ones((np.int8(3), np.int16(4), np.int32(5)), backend=backend,
                        aligned_index=(np.int64(1), 1, 0))
    assert s.shape == (3, 4, 5)


def test_subarray_dtype_appends_data_dims():
    a = gt_storage.zeros((3, 4, 5), (np.float64, (2,)), backend="gt:cpu_ifirst")
    assert a.shape == (3, 4, 5, 2)
    # data dimensions get the largest strides whatever the cartesian layout (layout.py:43-55)
    assert a.strides[3] > max(a.strides[:3])


class TestNormalizeStorageSpec:
    def test_defaults(self):
        ai, shape, dtype, dims = gt_storage.normalize_storage_spec(None, (2, 3), float, None)
        assert ai == (0, 0) and shape == (2, 3) and dtype == np.float64 and dims == ("I", "J")
        assert gt_storage.normalize_storage_spec(None, (2, 3, 4, 5), float, None)[3] == ("I", "J", "K", "0")

    def test_type_errors(self):
        with pytest.raises(TypeError, match="shape"):
            gt_storage.normalize_storage_spec(None, None, float, None)
        with pytest.raises(TypeError, match="shape"):
            gt_storage.normalize_storage_spec(None, (1.5, 2), float, None)
        with pytest.raises(TypeError, match="aligned_index"):
            gt_storage.normalize_storage_spec((0.5, 1), (2, 2), float, None)

    def test_value_errors(self):
        with pytest.raises(ValueError, match="non-matching"):
            gt_storage.normalize_storage_spec((0,), (2, 2), float, None)
        with pytest.raises(ValueError, match="non-matching"):
            gt_storage.normalize_storage_spec(None, (2, 2), float, ("I", "J", "K"))
        with pytest.raises(ValueError, match="non-positive"):
            gt_storage.normalize_storage_spec(None, (2, 0), float, None)
        with pytest.raises(ValueError, match="negative"):
            gt_storage.normalize_storage_spec((0, -1), (2, 2), float, None)
        with pytest.raises(ValueError, match="Invalid dimensions"):
            gt_storage.normalize_storage_spec(None, (2, 2), float, ("I", "X"))


def test_gpu_preset_fails_loudly_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no ROCm device"):
        gt_storage.zeros((4, 4, 4), backend="hip:mi300")
