"""Memory-group-aware placement of big device fields (gt4py_amd/storage/placement.py): the POLICY on the CPU with models of the
allocator and of the probe; the probe and the classes on the device (-m gpu)."""

import numpy as np
import pytest

from gt4py_amd.storage.placement import MIN_BYTES, MemoryGroupPlacer

GB = 1 << 30


class _Block:
    """What `allocate` returns in the model: an address, and (hidden from the placer) the memory group it lives in."""

    def __init__(self, ptr, group):
        self.ptr, self.group = ptr, group

    def data_ptr(self):
        return self.ptr


class _Device:
    """A model of the driver + caching allocator: fresh blocks come from a scripted sequence of groups; a block that is dropped
    goes to a cache and is handed out again FIRST (LIFO) -- exactly why the placer must hold its rejected candidates."""

    same_gbs, other_gbs = 5300.0, 6900.0  # two buffers written side by side: same group / different groups

    def __init__(self, groups):
        self.groups, self.next_ptr, self.cache, self.live, self.probes = list(groups), 0x1000000, [], {}, 0
        self.extent = {}  # ptr -> bytes of every block ever handed out (the calibration probes INSIDE the reference)

    def allocate(self, nbytes):
        import weakref

        if self.cache:
            ptr, group = self.cache.pop()
        else:
            ptr, group = self.next_ptr, self.groups.pop(0) if self.groups else 0
            self.next_ptr += nbytes + (4 << 20)
        self.extent[ptr] = max(nbytes, self.extent.get(ptr, 0))
        block = _Block(ptr, group)
        self.live[ptr] = group
        weakref.finalize(block, self._freed, ptr, group)
        return block

    def _freed(self, ptr, group):
        self.live.pop(ptr, None)
        self.cache.append((ptr, group))

    def group_of(self, address):
        for ptr, group in self.live.items():
            if ptr <= address < ptr + self.extent.get(ptr, 1):
                return group
        raise KeyError(hex(address))

    def probe(self, a, b, nbytes):
        self.probes += 1
        assert nbytes >= MIN_BYTES
        return self.same_gbs if self.group_of(a) == self.group_of(b) else self.other_gbs


def test_big_fields_are_dealt_over_the_two_classes():
    # the reference lands in group 0; then: 0 0 0 1 0 0 1 1 ...
    dev = _Device([0] + [0, 0, 0, 1, 0, 0, 1, 1, 0, 0])
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=8)
    blocks = [placer.place(GB + (4 << 20), label=n) for n in ("inp0", "out0", "inp1", "out1")]
    classes = [c for _, c in blocks]
    assert classes == [0, 1, 0, 1]  # a stencil's input and output never share a group when a second group can be found
    assert [dev.live[b.data_ptr()] for b, _ in blocks] == [0, 1, 0, 1]
    # rejected candidates were HELD during the search (the cache would have handed the same block out again) and released after it;
    # every candidate is probed, also one that comes back from the cache (the same address may be other memory by then)
    assert placer.stats["wanted_class_not_found"] == 0 and placer.stats["searches"] == 4
    # (+ the calibration probe of the reference's two halves and the three samples that tell whether the reference sits in the common group)
    assert placer.stats["probes"] == placer.stats["candidates"] + 1 + 3
    assert placer.live == [2 * (GB + (4 << 20)), 2 * (GB + (4 << 20))]
    # a field that dies gives its bytes back: the next one goes where the room is
    del blocks[1]
    import gc

    gc.collect()
    assert placer.live[1] == GB + (4 << 20)
    _, cls = placer.place(GB + (4 << 20))
    assert cls == 1


def test_a_search_that_finds_nothing_takes_what_the_driver_gave():
    dev = _Device([0] * 40)  # one group only, as far as the search can see
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=5)
    got = [placer.place(GB) for _ in range(3)]
    assert [c for _, c in got] == [0, 0, 0] and placer.stats["wanted_class_not_found"] == 2  # (the 2nd and 3rd wanted class 1)
    assert placer.stats["candidates"] <= 1 + 5 + 5 and len({b.data_ptr() for b, _ in got}) == 3
    # bounded by bytes as well: nothing beyond max_held_bytes is ever held
    dev = _Device([0] * 40)
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=30, max_held_bytes=3 * GB)
    placer.place(GB)
    placer.place(GB)
    assert placer.stats["candidates"] <= 1 + 3


def test_sizes_the_probe_cannot_classify_are_left_alone():
    dev = _Device([0, 1, 0, 1])
    placer = MemoryGroupPlacer(dev.allocate, dev.probe)
    small, cls = placer.place(64 << 20)  # (the Infinity Cache would absorb the probe)
    huge, cls2 = placer.place(8 * GB)
    assert cls is None and cls2 is None and dev.probes == 0 and placer.reference is None and placer.stats["unclassified"] == 2
    # nothing is placed while the stream is being captured into a graph (the probe synchronises)
    busy = MemoryGroupPlacer(dev.allocate, dev.probe, capturing=lambda: True)
    assert busy.place(GB)[1] is None and dev.probes == 0 and busy.reference is None
    off = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=0)
    assert off.place(GB)[1] is None and dev.probes == 0


def test_the_threshold_is_calibrated_on_the_reference_itself():
    """The two halves of the reference share a group by construction: their pair rate is a same-group rate of THIS device.  Inside the
    MI355X band the measured constant stands; outside it (another device, a power cap, a profiler, a neighbour) 'other group' is
    RELATIVE to the reference's own rate -- the constant could never be reached and every search would fail."""
    from gt4py_amd.storage.placement import PAIR_GBS_OTHER_GROUP, RELATIVE_MARGIN

    dev = _Device([0, 0, 1])
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=4)
    placer.place(GB)
    assert placer.threshold_mode == "mi355x" and placer.threshold == PAIR_GBS_OTHER_GROUP and placer.self_pair_gbs == 5300.0
    assert placer.reference_bytes <= 1 << 30  # (capped: fields larger than the reference are probed on their first GiB)

    class SlowDevice(_Device):  # e.g. an MI300X, or an MI355X under a serialising profiler: nothing reaches 6.55 TB/s
        same_gbs, other_gbs = 3400.0, 4100.0

    slow = SlowDevice([0, 0, 0, 1, 0, 1])
    p2 = MemoryGroupPlacer(slow.allocate, slow.probe, max_candidates=6)
    kept = [p2.place(GB) for _ in range(3)]  # (kept alive: a field that dies gives its bytes back to the balance)
    got = [c for _, c in kept]
    assert p2.threshold_mode == "relative" and p2.threshold == pytest.approx(RELATIVE_MARGIN * 3400.0)
    assert got == [0, 1, 0] and p2.stats["wanted_class_not_found"] == 0  # the second group IS found, at 4.1 TB/s
    # a threshold given by the caller (or GT4PY_AMD_ALLOC_GROUP_PAIR_GBS) is never recalibrated
    p3 = MemoryGroupPlacer(slow.allocate, slow.probe, threshold_gbs=4000.0)
    p3.place(GB)
    assert p3.threshold_mode == "fixed" and p3.threshold == 4000.0


def test_class_0_is_the_group_the_driver_hands_out_most():
    """A reference that lands in the RARER group would make class 0 the scarce one (every search for it walks the whole budget);
    three plain samples after the calibration notice it and the reference moves to the common group."""
    dev = _Device([1] + [0, 0, 0] + [0, 0, 1, 0, 0, 0, 0])  # the first block -- the reference -- in group 1, nearly everything else in 0
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=6)
    kept = [placer.place(GB) for _ in range(3)]
    assert placer.stats.get("reference_moved_to_the_common_group") == 1 and dev.group_of(placer.reference.data_ptr()) == 0
    assert [c for _, c in kept] == [0, 1, 0] and [dev.group_of(b.data_ptr()) for b, _ in kept] == [0, 1, 0]
    assert placer.stats["wanted_class_not_found"] == 0
    # a reference in the common group stays
    dev2 = _Device([0] + [0, 1, 0] + [0, 1])
    p2 = MemoryGroupPlacer(dev2.allocate, dev2.probe, max_candidates=6)
    p2.place(GB)
    assert "reference_moved_to_the_common_group" not in p2.stats and dev2.group_of(p2.reference.data_ptr()) == 0


def test_a_placer_whose_searches_keep_failing_goes_dormant():
    dev = _Device([0] * 200)  # one group only
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=4, dormant_after=3)
    with pytest.warns(RuntimeWarning, match="dormant"):
        kept = [placer.place(GB) for _ in range(8)]  # wanted: 0 1 1 1 (three failures in a row) -> dormant
    assert placer.dormant and placer.stats["wanted_class_not_found"] == 3
    probes = dev.probes
    block, cls = placer.place(GB)
    assert cls is None and dev.probes == probes  # no more candidates, no more probes: the field lands where the driver puts it
    # a success in between resets the count
    dev2 = _Device([0, 0, 0, 0, 0, 0, 1] + [0] * 50)
    p2 = MemoryGroupPlacer(dev2.allocate, dev2.probe, max_candidates=2, dormant_after=3)
    import warnings as _w

    with _w.catch_warnings():
        _w.simplefilter("error")
        kept2 = [p2.place(GB), p2.place(GB), p2.place(GB)]  # 0, fail, fail
        assert p2.failed_in_a_row == 2 and not p2.dormant


def test_roles_decide_the_class_when_the_caller_knows_them():
    from gt4py_amd.storage.placement import deal_by_roles

    # what is written is dealt alternately starting with class 1; what is only read fills up the emptier class (ties: class 0)
    assert deal_by_roles([("inp", False, 1), ("out", True, 1)]) == {"inp": 0, "out": 1}
    assert deal_by_roles([("in_field", False, 1), ("out_field", True, 1), ("coeff", False, 1)]) == {"in_field": 0, "out_field": 1, "coeff": 0}
    assert deal_by_roles([("inf", False, 1), ("diag", False, 1), ("sup", True, 1), ("rhs", True, 1), ("out", True, 1)]) == \
        {"inf": 0, "diag": 0, "sup": 1, "rhs": 0, "out": 1}
    # sizes count: a big read-only field balances two small written ones
    assert deal_by_roles([("big", False, 10), ("a", True, 1), ("b", True, 1), ("small", False, 1)]) == {"big": 0, "a": 1, "b": 0, "small": 1}
    # ... and the stencil objects hand it out from their field_info
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    tri = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64})
    assert tri.placement_hint() == {"inf": 0, "diag": 0, "sup": 1, "rhs": 0, "out": 1}
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float32})
    assert hd.placement_hint() == {"in_field": 0, "out_field": 1, "coeff": 0}
    frozen = hd.freeze(origin={k: (2, 2, 0) for k in hd.field_info}, domain=(8, 8, 2))
    assert frozen.placement_hint() == hd.placement_hint()
    # the explicit class wins over the balance of live bytes
    dev = _Device([0, 0, 1, 1, 0, 1])
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=6)
    kept = [placer.place(GB, wanted=c) for c in (1, 1, 0)]
    assert [c for _, c in kept] == [1, 1, 0]


def test_memory_class_is_validated_and_ignored_on_the_host():
    import gt4py_amd.storage as gt_storage

    a = gt_storage.zeros((4, 4, 2), np.float64, backend="numpy", memory_class=1)  # host presets have no memory classes
    assert a.shape == (4, 4, 2)


@pytest.mark.gpu
def test_the_probe_and_the_placer_on_the_device():
    """The C entry measures; storages of 1 GB are classified, dealt over the classes when a second group is within reach of the
    search, and compute the same values wherever they live."""
    import ctypes

    import torch

    import gt4py_amd.storage as gt_storage
    from gt4py_amd import _lib
    from gt4py_amd.storage import placement

    lib = _lib.load()
    a = torch.empty(640 << 20, dtype=torch.uint8, device="cuda")
    b = torch.empty(640 << 20, dtype=torch.uint8, device="cuda")
    gbs = ctypes.c_double()
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check("probe", lib.gt4mi_memory_write_probe(a.data_ptr(), None, a.numel(), 4, stream, ctypes.byref(gbs)))
    alone = gbs.value
    _lib.check("probe", lib.gt4mi_memory_write_probe(a.data_ptr(), b.data_ptr(), a.numel(), 4, stream, ctypes.byref(gbs)))
    pair = gbs.value
    assert 2500.0 < alone < 8000.0 and 2500.0 < pair < 8000.0  # an HBM rate, not a cache rate
    assert lib.gt4mi_memory_write_probe(a.data_ptr(), None, 64 << 20, 4, stream, ctypes.byref(gbs)) == _lib.ERR_INVALID_ARGUMENT  # (the cache would absorb it)
    del a, b
    placer = placement.configure(max_candidates=4)
    if placer is None:
        pytest.skip("placement switched off in this environment")
    before = list(placer.stats["placed"])
    shape = (514, 514, 512)  # 1.08 GB: the headline's fields
    f = [gt_storage.zeros(shape, np.float64, backend="hip:mi300", aligned_index=(1, 1, 0)) for _ in range(2)]
    classes = [placement.class_of(x) for x in f]
    assert all(c in (0, 1) for c in classes) and sum(placer.stats["placed"]) == sum(before) + 2
    rep = placement.report()
    assert rep["enabled"] and rep["searches"] >= 2 and rep["fields"][-1]["bytes"] >= 514 * 514 * 512 * 8
    if rep["wanted_class_not_found"] == 0:
        assert sorted(classes) == [0, 1]  # dealt over the two classes
    assert float(f[0].tensor.abs().max()) == 0.0 and float(f[1].tensor.abs().max()) == 0.0  # zeros() filled AFTER the probe wrote
    # the threshold was calibrated on the placer's own reference (one per device, <= 1 GiB)
    assert rep["pair_threshold_mode"] in ("mi355x", "relative", "fixed") and rep["device"] == torch.cuda.current_device()
    assert rep["reference_bytes"] <= 1 << 30 and (rep["pair_threshold_mode"] == "fixed" or rep["reference_self_pair_gbs"] > 1000.0)
    # roles: the class a caller names wins over the balance of live bytes (when the search finds it)
    found_before = rep["wanted_class_not_found"]
    g = [gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=(1, 1, 0), memory_class=c) for c in (1, 1, 0)]
    if placement.report()["wanted_class_not_found"] == found_before and not placement.report()["dormant"]:
        assert [placement.class_of(x) for x in g] == [1, 1, 0]
    with pytest.raises(ValueError, match="memory_class"):
        gt_storage.empty(shape, np.float64, backend="hip:mi300", memory_class=2)
    with placement.disabled():  # what GT4PY_AMD_ALLOC_GROUPS=0 does for a whole process
        plain = gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=(1, 1, 0))
    assert placement.class_of(plain) is None


def test_the_wide_search_reaches_a_group_that_is_far_away_and_parks_its_neighbours():
    """Groups change along the physical address space, often tens of GB apart: behind `plain_candidates` plain candidates every
    further one follows a SPACER that is never touched; a find far away brings `park_extra` neighbours along, so that the next
    fields that want the class do not search again; spacers and rejected candidates go back to the DRIVER when a search ends."""
    class FarDevice(_Device):
        """Group by ADDRESS: everything below 40 GB is group 0, everything above group 1 (spacers move the frontier)."""

        def allocate(self, nbytes):
            import weakref

            if self.cache:
                ptr, group = self.cache.pop()
            else:  # the driver: first fit, lowest address first, among what is live or cached
                taken = sorted(self.sizes.items())
                ptr = 0
                for start, size in taken:
                    if ptr + nbytes <= start:
                        break
                    ptr = max(ptr, start + size)
                group = 0 if ptr < 40 * GB else 1
            self.sizes[ptr] = nbytes
            self.extent[ptr] = max(nbytes, self.extent.get(ptr, 0))
            block = _Block(ptr, group)
            self.live[ptr] = group
            weakref.finalize(block, self._freed, ptr, group)
            return block

        sizes = {}

        def drop_cache(self):  # (torch.cuda.empty_cache: cached blocks go back to the driver)
            for ptr, _ in self.cache:
                self.sizes.pop(ptr, None)
            self.cache.clear()

    FarDevice.sizes = {}
    dev = FarDevice([])
    released = []
    free = [200 * GB]
    placer = MemoryGroupPlacer(dev.allocate, dev.probe, max_candidates=24, spacer_bytes=8 * GB, plain_candidates=4, park_extra=3,
                               release=lambda: (released.append(len(dev.cache)), dev.drop_cache()), free_bytes=lambda: free[0])
    a, ca = placer.place(GB)           # class 0 at once
    b, cb = placer.place(GB)           # wants class 1: 4 plain candidates, then spacer + candidate until beyond 40 GB
    assert (ca, cb) == (0, 1) and dev.live[b.data_ptr()] == 1 and b.data_ptr() >= 40 * GB
    assert placer.stats["spacers"] >= 4 and released and placer.stats["wanted_class_not_found"] == 0
    assert sum(len(v) for v in placer.parked.values()) == 3  # its neighbours, parked
    searches = placer.stats["searches"]
    c, cc = placer.place(GB)           # class 0 again (the frontier is back at the start: the spacers were released)
    d, cd = placer.place(GB)           # class 1 from the parked blocks: no search
    assert (cc, cd) == (0, 1) and placer.stats["searches"] == searches + 1 and sum(len(v) for v in placer.parked.values()) == 2
    # a smaller field that wants the far class lives in one of the parked (bigger) blocks instead of starting a search that may fail
    searches = placer.stats["searches"]
    e, ce = placer.place(GB // 4, wanted=1)
    assert ce == 1 and placer.stats["searches"] == searches and placer.stats["served_from_a_bigger_parked_block"] == 1
    assert dev.live[e.data_ptr()] == 1 and sum(len(v) for v in placer.parked.values()) == 1
    # a device that is nearly full is not filled up with spacers
    FarDevice.sizes = {}
    dev2, free2 = FarDevice([]), [20 * GB]
    p2 = MemoryGroupPlacer(dev2.allocate, dev2.probe, max_candidates=24, spacer_bytes=8 * GB, free_bytes=lambda: free2[0], keep_free_bytes=16 * GB)
    first = p2.place(GB)
    second, cls = p2.place(GB)
    assert first[1] == 0 and cls == 0 and p2.stats.get("spacers", 0) == 0 and p2.stats["wanted_class_not_found"] == 1


def test_running_out_of_memory_in_the_middle_of_a_search_ends_the_search_not_the_program():
    dev = _Device([0] * 40)
    budget = {"left": 4 + 3}  # (+ the three samples taken once, right after the reference; they are released again)

    def allocate(nbytes):
        if budget["left"] <= 0:
            raise MemoryError("out of device memory")
        budget["left"] -= 1
        return dev.allocate(nbytes)

    placer = MemoryGroupPlacer(allocate, dev.probe, max_candidates=8)
    first = placer.place(GB)   # reference (+ its three samples) + one candidate
    second = placer.place(GB)  # wants class 1: two more candidates fit, then the device is full -> takes the first candidate
    assert first[1] == 0 and second[1] == 0 and placer.stats["search_ended_by_allocation_failure"] == 1
    budget["left"] = 0
    with pytest.raises(MemoryError):  # a field that does not fit at all is the caller's problem, as without a placer
        placer.place(GB)


@pytest.mark.gpu
@pytest.mark.perf
def test_the_tridiagonal_solve_is_faster_with_its_fields_dealt_over_two_memory_groups():
    """The effect the placer exists for, measured in ONE process (BASELINE configs[3], 1024 x 1024 x 160 fp64): the five fields of the
    solve dealt over the two memory classes against all five in class 0.  Rounds 2-4 saw "two speed modes by allocation set", 13 %
    apart, and could not steer them; with `placement.want` they can be chosen.  Skipped on a box where the wide search finds no
    second group; bit-identical results either way."""
    import torch

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates
    from gt4py_amd.storage import placement

    placer = placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=5)
    if placer is None:
        pytest.skip("placement switched off in this environment")
    dom = (1024, 1024, 160)
    tri = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64}, device_sync=False)
    gen = torch.Generator(device="cuda").manual_seed(7)
    ranges = {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (0, 0)}
    host = {n: (torch.rand(dom, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo) for n, (lo, hi) in ranges.items()}

    def solve_ms(classes):
        fields = {}
        for name, cls in zip(ranges, classes):
            with placement.want(cls):
                fields[name] = gt_storage.empty(dom, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0))
        got = tuple(placement.class_of(f) for f in fields.values())
        if got != tuple(classes):
            return None, None
        frozen = tri.freeze(origin={k: (0, 0, 0) for k in fields}, domain=dom)
        times = []
        for _ in range(6):
            for n in ("inf", "diag", "sup", "rhs", "out"):
                fields[n].tensor.copy_(host[n])  # pristine operands every launch (the solve overwrites sup and rhs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            frozen(**fields)
            b.record()
            b.synchronize()
            times.append(a.elapsed_time(b))
        return sorted(times)[len(times) // 2], fields["out"].tensor.clone()

    one, out_one = solve_ms((0, 0, 0, 0, 0))
    dealt, out_dealt = solve_ms((0, 1, 0, 1, 0))
    again, _ = solve_ms((0, 0, 0, 0, 0))
    try:
        if one is None or dealt is None or again is None:
            pytest.skip(f"no second memory group within reach of the wide search on this box ({placement.report()['wanted_class_not_found']} searches failed)")
        assert torch.equal(out_one, out_dealt)  # where a field lives never changes a bit of the result
        frac = lambda ms: 56.0 * np.prod(dom) / (ms * 1e-3) / 8e12  # noqa: E731
        print(f"tridiagonal solve, fraction of the HBM peak: one class {frac(one):.3f} / {frac(again):.3f}, dealt over two {frac(dealt):.3f}")
        # (measured on five boxes: 8-14 % apart.  A wall-clock ordering is a PERFORMANCE statement: it is asserted only under
        # GT4MI_PERF_ASSERT=1 -- a noisy or shared box must not end a `-x` correctness run -- the bit-identity above is
        # asserted always, the figures are printed above and recorded in profiles/)
        import os

        if os.environ.get("GT4MI_PERF_ASSERT") == "1":
            assert dealt < min(one, again), (one, dealt, again)
    finally:
        placement.configure(max_candidates=6, spacer_bytes=0, park_extra=0)  # (the defaults, for whatever runs after this test)
        placer.parked.clear()
        torch.cuda.empty_cache()
