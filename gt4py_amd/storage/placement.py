"""Memory-group-aware placement of big device fields (``hip:mi300`` storage preset) -- NEW relative to the reference, whose
allocators hand a byte count to cupy and know nothing of the device's memory system (storage/allocators.py:187-273).

What it is for (measured in round 5, profiles/r5_memory_groups.txt): MI355X's memory is not one uniformly interleaved pool.  Two
big allocations either share a GROUP of memory channels or they do not, nothing in the HIP API says which, and kernels feel it:

* two 1.3 GB fields written side by side: 5.0-6.4 TB/s when both live in one group, 6.8-7.0 TB/s in two;
* fp64 5-point Laplacian on 512^3: +2.3 % with ``in`` and ``out`` in different groups;
* tridiagonal solve 1024 x 1024 x 160: 0.70 of the HBM peak with its five fields dealt over two groups, 0.61 with all five in one
  -- the "two speed modes by allocation set" that rounds 2-4 could measure and not explain.

What it does: big allocations (``min_bytes`` <= size <= ``max_bytes``: 192 MiB .. 4 GiB) are CLASSIFIED with ``gt4mi_memory_write_probe`` against ONE
reference buffer the placer owns -- class 0: the reference's group, class 1: any other group -- and DEALT: a new field goes to the
class that holds fewer live bytes.  To get there the placer allocates candidates (held until the search ends: a block that went back
to the caching allocator would be handed out again at once), takes the first one of the wanted class, and releases the rest; the
search is bounded (``max_candidates``, ``max_held_bytes``) and when it finds nothing the field simply lands where the driver put it.
The layout contract is untouched: which raw block a storage lives in is no part of it.

The probe OVERWRITES a candidate: it only ever runs on fresh blocks (``empty`` semantics; ``zeros`` / ``ones`` / ``from_array`` fill
afterwards) and on the placer's own reference.  Every candidate is probed (~5 ms), also one whose address was seen before: the same
virtual address may come back on other physical memory; only live blocks keep their class.

ROLES (round 6): the deal by live bytes knows nothing of what a kernel does with a field.  ``deal_by_roles`` does -- what a stencil
WRITES is dealt alternately over the classes starting with class 1, what it only reads fills up the emptier class -- and
``StencilObject.placement_hint()`` / ``gt_storage.empty(..., memory_class=hint[name])`` carry it to the allocator
(profiles/r6_memory_roles.log: horizontal diffusion in / coeff / out = 0 / 0 / 1 is 3-4 % faster than 0 / 1 / 0; the tridiagonal solve
inf / diag / sup / rhs / out = 0 / 0 / 1 / 0 / 1 0.784-0.790 of the HBM peak, the best of all 16 assignments, 0 / 1 / 0 / 1 / 0 0.756-0.761).

The pair threshold is CALIBRATED per device when the reference buffer is made (``_calibrate``): the two halves of the reference share a
group by construction, and so their pair rate is a same-group rate of THIS device under THIS load; the MI355X constant (6 550 GB/s, in
the gap between 4.9-6.4 same group and 6.7-7.05 other group) stands as long as that rate lies in the MI355X band, otherwise -- another
device, a power cap, a profiler, a neighbour on the device -- the threshold becomes relative (1.06 x the reference's own rate).  A placer
whose searches fail ``dormant_after`` times in a row goes DORMANT: big fields land where the driver puts them, no more candidates, no more
probes.  Nothing is placed while the current stream is being captured into a graph (the probe synchronises).

What it costs a process that allocates big fields (all of it only from the first field of >= 192 MiB on): one reference buffer of
512 MiB .. 1 GiB per device for the life of the process; once, three sample blocks of the reference's size (released at once) that tell
whether the reference sits in the group the driver hands out most -- if not, one of them becomes the reference, so that class 0 is the
COMMON group and only class 1 needs the far search; per big allocation at most ``max_candidates`` (6) candidates of the field's size held
until the search ends (<= ``max_held_bytes``, 16 GiB) and ~5 ms of probe per candidate (the probe synchronises the stream); a smaller field
of the far class may be served from a bigger parked block.  ``report()`` says what happened, including a histogram of the probed rates.

One placer PER DEVICE (``device_placer()`` follows ``torch.cuda.current_device()``); the reference buffer is capped at 1 GiB;
``torch.cuda.empty_cache()`` after a search only when the search used spacers (the wide search) or ``release_cache=True`` was configured.

``GT4PY_AMD_ALLOC_GROUPS=0`` switches the placer off; ``GT4PY_AMD_ALLOC_GROUP_SEARCH`` = candidates per search (default 6);
``GT4PY_AMD_ALLOC_GROUP_PAIR_GBS`` fixes the threshold (no calibration).
"""

from __future__ import annotations

import os
import weakref
from typing import Any, Callable, Dict, List, Optional, Tuple

MIN_BYTES = 192 << 20  # below this the Infinity Cache (256 MB) absorbs the pair probe's writes (classification is clean from 192 MiB on:
#                        same group 5.0-5.7 TB/s, other group 6.7-6.95, profiles/r5_memory_groups.txt)
REFERENCE_BYTES = 512 << 20  # the placer's own buffer is at least this big
MAX_BYTES = 4 << 30    # above this a held candidate costs too much; such fields land where the driver puts them
MAX_REFERENCE_BYTES = 1 << 30  # ... and at most this big (fields larger than the reference are probed on their first GiB)
PAIR_GBS_OTHER_GROUP = 6550.0  # two buffers written side by side: >= this -> different groups (same group: 5.0-6.4, other: 6.8-7.0 TB/s)
SAME_GROUP_BAND_GBS = (4300.0, 6450.0)  # the reference's OWN halves written side by side on an unloaded MI355X (profiles/r5_memory_groups.txt)
RELATIVE_MARGIN = 1.06         # outside that band: other group = this much faster than the reference's own halves
DORMANT_AFTER = 3              # searches that may fail in a row before the placer stops searching


class MemoryGroupPlacer:
    """``allocate(nbytes) -> block`` (an object with ``data_ptr()``; it is kept alive by whoever holds it) and
    ``probe(ptr_a, ptr_b_or_0, nbytes) -> GB/s`` are injected: the tests run the policy on the CPU with models of both."""

    def __init__(self, allocate: Callable[[int], Any], probe: Callable[[int, int, int], float], *, min_bytes: int = MIN_BYTES,
                 max_bytes: int = MAX_BYTES, max_candidates: Optional[int] = None, max_held_bytes: int = 16 << 30,
                 threshold_gbs: Optional[float] = None, release: Optional[Callable[[], None]] = None,
                 free_bytes: Optional[Callable[[], int]] = None, spacer_bytes: int = 0, plain_candidates: int = 4, park_extra: int = 0,
                 keep_free_bytes: int = 16 << 30, dormant_after: int = DORMANT_AFTER, release_cache: bool = False,
                 capturing: Optional[Callable[[], bool]] = None):
        self.allocate, self.probe, self.release, self.free_bytes = allocate, probe, release, free_bytes
        self.min_bytes, self.max_bytes, self.max_held_bytes = int(min_bytes), int(max_bytes), int(max_held_bytes)
        # the WIDE search (off by default; `configure(spacer_bytes=...)`): groups change along the physical address space, often only
        # tens of GB apart, so after `plain_candidates` plain candidates every further candidate is preceded by a SPACER -- an
        # allocation of `spacer_bytes` that is never touched and only moves the driver's frontier -- as long as `keep_free_bytes`
        # of device memory stay free; everything is released when the search ends.  `park_extra`: when a block of a class is found
        # far away, up to that many more blocks of the same size are taken next to it (groups come in runs) and PARKED for the
        # next fields that want the class, so that the wide search runs once, not once per field.
        self.spacer_bytes, self.plain_candidates, self.park_extra = int(spacer_bytes), int(plain_candidates), int(park_extra)
        self.keep_free_bytes = int(keep_free_bytes)
        self.parked: Dict[Tuple[int, int], list] = {}  # (class, size) -> blocks
        self.forced: Optional[int] = None  # `want(cls)`: the class the next fields go to, whatever the balance says
        self.max_candidates = int(max_candidates if max_candidates is not None else os.environ.get("GT4PY_AMD_ALLOC_GROUP_SEARCH", "6"))
        fixed = threshold_gbs if threshold_gbs is not None else os.environ.get("GT4PY_AMD_ALLOC_GROUP_PAIR_GBS")
        self.threshold = float(fixed if fixed is not None else PAIR_GBS_OTHER_GROUP)
        self.threshold_mode = "fixed" if fixed is not None else "uncalibrated"  # -> "mi355x" / "relative" once the reference exists
        self.self_pair_gbs: Optional[float] = None  # the reference's own halves written side by side (same group by construction)
        self.dormant_after, self.failed_in_a_row, self.dormant = int(dormant_after), 0, False
        self.release_cache, self.capturing = bool(release_cache), capturing
        self.reference: Any = None          # the placer's own buffer: class 0 is ITS group
        self.reference_bytes = 0
        self.known: Dict[Tuple[int, int], int] = {}  # (address, size) -> class of the LIVE blocks the placer handed out or parked
        self.live = [0, 0]                  # bytes of live classified fields per class
        self.stats = {"placed": [0, 0], "searches": 0, "candidates": 0, "probes": 0, "wanted_class_not_found": 0, "unclassified": 0}
        self.log: List[Dict[str, Any]] = []  # one record per placed field (bench.py prints it)
        self.rates: List[float] = []  # the pair rate of every candidate probed (report(): a histogram -- is this box bimodal, where is the gap?)

    # ---- classification -----------------------------------------------------------------------------------------------------
    def _classify(self, block, nbytes: int) -> int:
        # Every candidate is PROBED, also one whose address was seen before: a block that went back to the driver (the caching
        # allocator's empty_cache, which the wide search itself calls) may come back at the same virtual address on other
        # physical memory.  Only blocks that are alive -- handed out or parked -- keep their class (`known`).
        span = min(nbytes, self.reference_bytes)
        try:
            gbs = self.probe(int(block.data_ptr()), int(self.reference.data_ptr()), span)
        except Exception as ex:  # noqa: BLE001 - a probe that fails switches the placer off; allocation itself goes on
            import warnings

            warnings.warn(f"gt4py_amd.storage.placement: the memory-group probe failed ({ex!r}); fields are no longer placed", RuntimeWarning)
            self.max_candidates = 0
            return 0
        self.stats["probes"] += 1
        self.rates.append(float(gbs))
        return 1 if gbs >= self.threshold else 0

    def _ensure_reference(self, nbytes: int) -> None:
        """The first big allocation's block becomes the reference (kept for the life of the placer: <= max_bytes)."""
        if self.reference is None:
            self.reference_bytes = max(min(nbytes, MAX_REFERENCE_BYTES), REFERENCE_BYTES, 2 * self.min_bytes)
            self.reference = self.allocate(self.reference_bytes)
            self._calibrate()

    def _calibrate(self) -> None:
        """The threshold of THIS device: the two halves of the reference share a group by construction, so their pair rate is a
        same-group rate under the conditions of this process.  Inside the MI355X band the measured constant stands (relative to the
        reference's own rate it would not: same-group pairs spread over 4.9-6.4 TB/s with the partner block); outside it the device is
        not an unloaded MI355X and 'other group' becomes RELATIVE_MARGIN x the reference's own rate."""
        if self.threshold_mode == "fixed":
            return
        half = (self.reference_bytes // 2) & ~0xFFF
        try:
            base = int(self.reference.data_ptr())
            self.self_pair_gbs = float(self.probe(base, base + half, half))
        except Exception as ex:  # noqa: BLE001
            import warnings

            warnings.warn(f"gt4py_amd.storage.placement: calibration probe failed ({ex!r}); fields are no longer placed", RuntimeWarning)
            self.max_candidates = 0
            return
        self.stats["probes"] += 1
        lo, hi = SAME_GROUP_BAND_GBS
        if lo <= self.self_pair_gbs <= hi:
            self.threshold, self.threshold_mode = PAIR_GBS_OTHER_GROUP, "mi355x"
        else:
            self.threshold, self.threshold_mode = RELATIVE_MARGIN * self.self_pair_gbs, "relative"
        self._prefer_the_common_group()

    def _prefer_the_common_group(self, samples: int = 3) -> None:
        """Class 0 is "the reference's group" -- and should be the group the driver hands out MOST of the time: a search for the
        common group succeeds at its first candidates, a search for the other one is what the spacers and parked blocks are for.
        A reference that happened to land in the rarer group turns that around (seen in round 6 under a profiler: 235 candidates,
        222 of them "other", six searches for class 0 failed).  So: `samples` plain candidates of the reference's size right after
        the calibration; if most of them are NOT in the reference's group, one of those becomes the reference."""
        try:
            blocks = [self.allocate(self.reference_bytes) for _ in range(samples)]
            other = []
            for b in blocks:
                gbs = float(self.probe(int(b.data_ptr()), int(self.reference.data_ptr()), self.reference_bytes))
                self.stats["probes"] += 1
                if gbs >= self.threshold:
                    other.append(b)
        except Exception:  # noqa: BLE001 - not enough memory for the samples, a probe that fails: the reference stays what it is
            return
        if 2 * len(other) > samples:
            self.reference = other[0]
            self.stats["reference_moved_to_the_common_group"] = 1

    # ---- placement ----------------------------------------------------------------------------------------------------------
    def place(self, nbytes: int, label: str = "", wanted: Optional[int] = None):
        """A block of ``nbytes`` in class ``wanted`` -- else the class ``want()`` forces, else the one that currently holds fewer live
        bytes -- if the search finds one; ``(block, cls)`` with ``cls`` None for sizes the placer does not handle, for a dormant
        placer and while the stream is being captured."""
        nbytes = int(nbytes)
        if not (self.min_bytes <= nbytes <= self.max_bytes) or self.max_candidates <= 0 or self.dormant or (self.capturing is not None and self.capturing()):
            self.stats["unclassified"] += 1
            return self.allocate(nbytes), None
        self._ensure_reference(nbytes)
        if self.max_candidates <= 0:  # (the calibration probe failed)
            self.stats["unclassified"] += 1
            return self.allocate(nbytes), None
        if wanted is None:
            wanted = self.forced if self.forced is not None else (0 if self.live[0] <= self.live[1] else 1)
        held, spacers, chosen, chosen_cls = [], [], None, None
        block_bytes = nbytes  # (what the chosen raw block really holds: more than asked for when it comes from a bigger parked block)
        parked = self.parked.get((wanted, nbytes))
        if not parked:
            # a BIGGER parked block of the wanted class (the smallest that fits, at most 8 x the request): where the other group is
            # far away every find is precious -- a field of 350 MB may live in a parked block of 1.1 GB rather than start a search
            # of its own that may fail (round 6: 5 of 26 searches failed on a box whose second group was rare within reach)
            fits = sorted(size for (cls, size), blocks in self.parked.items() if cls == wanted and blocks and nbytes < size <= 8 * nbytes)
            if fits:
                parked, block_bytes = self.parked[(wanted, fits[0])], fits[0]
                self.stats["served_from_a_bigger_parked_block"] = self.stats.get("served_from_a_bigger_parked_block", 0) + 1
        if parked:
            chosen, chosen_cls = parked.pop(), wanted
            self.known.pop((int(chosen.data_ptr()), block_bytes), None)
        else:
            self.stats["searches"] += 1
            held_bytes = 0
            for i in range(self.max_candidates):
                try:
                    if i >= self.plain_candidates and self.spacer_bytes > 0:
                        if self.free_bytes is not None and self.free_bytes() < self.spacer_bytes + nbytes + self.keep_free_bytes:
                            break
                        spacers.append(self.allocate(self.spacer_bytes))  # never touched: it only moves the driver's frontier
                        self.stats["spacers"] = self.stats.get("spacers", 0) + 1
                    block = self.allocate(nbytes)
                except Exception:  # noqa: BLE001 - out of memory in the middle of a SEARCH ends the search, not the program
                    if i == 0:
                        raise  # (the field itself does not fit: the caller's problem, as without a placer)
                    self.stats["search_ended_by_allocation_failure"] = self.stats.get("search_ended_by_allocation_failure", 0) + 1
                    break
                cls = self._classify(block, nbytes)
                self.stats["candidates"] += 1
                if cls == wanted:
                    chosen, chosen_cls = block, cls
                    for _ in range(self.park_extra if i >= self.plain_candidates else 0):  # found far away: take its neighbours too
                        try:
                            extra = self.allocate(nbytes)
                        except Exception:  # noqa: BLE001
                            break
                        if self._classify(extra, nbytes) == wanted:
                            self.parked.setdefault((wanted, nbytes), []).append(extra)
                            self.known[(int(extra.data_ptr()), nbytes)] = wanted
                        else:
                            held.append((extra, 1 - wanted))
                    break
                held.append((block, cls))
                held_bytes += nbytes
                if not spacers and held_bytes + nbytes > self.max_held_bytes:
                    break
            if chosen is None:  # nothing of the wanted class within the budget: the first candidate, whatever it is
                self.stats["wanted_class_not_found"] += 1
                chosen, chosen_cls = held.pop(0)
                self.failed_in_a_row += 1
                if self.dormant_after > 0 and self.failed_in_a_row >= self.dormant_after:
                    # a second group is not within reach (or the probe cannot tell on this device / under this load): stop paying
                    # candidates and probes for every big allocation
                    self.dormant = True
                    import warnings

                    warnings.warn(f"gt4py_amd.storage.placement: {self.failed_in_a_row} searches in a row found no block of the wanted memory class "
                                  f"(pair threshold {self.threshold:.0f} GB/s, {self.threshold_mode}); the placer is dormant from here on -- "
                                  "placement.configure(...) re-arms it", RuntimeWarning)
            else:
                self.failed_in_a_row = 0
        n_rejected = len(held)
        used_spacers = len(spacers)
        del held, spacers  # rejected candidates and spacers go back to the allocator NOW, not before: it would have handed them out again
        if self.release is not None and (used_spacers or (self.release_cache and n_rejected > 4)):
            self.release()  # (back to the DRIVER -- the user's caching allocator is emptied: only after a WIDE search, which parks tens
            #                 of GB in it, or when configured)
        self.live[chosen_cls] += nbytes
        self.stats["placed"][chosen_cls] += 1
        self.log.append({"label": label, "bytes": nbytes, "class": chosen_cls, "wanted": wanted, "candidates_rejected": n_rejected})
        key = (int(chosen.data_ptr()), block_bytes)
        self.known[key] = chosen_cls
        weakref.finalize(chosen, self._gone, chosen_cls, nbytes, key)
        return chosen, chosen_cls

    def _gone(self, cls: int, nbytes: int, key) -> None:
        self.live[cls] -= nbytes
        self.known.pop(key, None)


_PLACERS: Dict[int, MemoryGroupPlacer] = {}  # one per device: the reference buffer, the live balance and the classes are the device's
_DISABLED_REASON: Optional[str] = None
_SUSPENDED = 0  # depth of `disabled()` contexts


def _current_device() -> Optional[int]:
    import torch

    return int(torch.cuda.current_device()) if torch.cuda.is_available() else None


def device_placer() -> Optional[MemoryGroupPlacer]:
    """The placer of the CURRENT device's ``hip:mi300`` storages (``torch.cuda.current_device()``), or None when it is switched off
    (``GT4PY_AMD_ALLOC_GROUPS=0``, inside ``disabled()``) or there is no device."""
    global _DISABLED_REASON
    if _DISABLED_REASON is not None or _SUSPENDED:
        return None
    if os.environ.get("GT4PY_AMD_ALLOC_GROUPS", "1") == "0":
        _DISABLED_REASON = "GT4PY_AMD_ALLOC_GROUPS=0"
        return None
    device = _current_device()
    if device is None:
        return None
    placer = _PLACERS.get(device)
    if placer is not None:
        return placer
    import ctypes

    import torch

    from .. import _lib

    lib = _lib.load()

    def allocate(nbytes: int):
        return torch.empty((int(nbytes),), dtype=torch.uint8, device=f"cuda:{device}")

    def free_bytes() -> int:
        return int(torch.cuda.mem_get_info(device)[0])

    def probe(a: int, b: int, nbytes: int) -> float:
        gbs = ctypes.c_double()
        with torch.cuda.device(device):  # (the probe refuses buffers of another device than its stream's)
            _lib.check("gt4mi_memory_write_probe",
                       lib.gt4mi_memory_write_probe(a, b or None, int(nbytes), 6, torch.cuda.current_stream().cuda_stream, ctypes.byref(gbs)))
        return float(gbs.value)

    def capturing() -> bool:
        return bool(torch.cuda.is_current_stream_capturing())

    placer = _PLACERS[device] = MemoryGroupPlacer(allocate, probe, release=torch.cuda.empty_cache, free_bytes=free_bytes, capturing=capturing)
    placer.device = device
    return placer


class disabled:
    """``with placement.disabled(): f = gt_storage.empty(...)`` -- big fields allocated inside land where the driver puts them (what
    ``GT4PY_AMD_ALLOC_GROUPS=0`` does for a whole process)."""

    def __enter__(self):
        global _SUSPENDED
        _SUSPENDED += 1
        return self

    def __exit__(self, *exc):
        global _SUSPENDED
        _SUSPENDED -= 1
        return False


def configure(*, max_candidates: Optional[int] = None, max_held_bytes: Optional[int] = None, spacer_bytes: Optional[int] = None,
              park_extra: Optional[int] = None, keep_free_bytes: Optional[int] = None, release_cache: Optional[bool] = None,
              dormant_after: Optional[int] = None) -> Optional[MemoryGroupPlacer]:
    """An application that knows it is about to allocate the fields of a bandwidth-bound stencil may widen the search
    (``bench.py`` does): candidates per search, bytes of rejected candidates held at once, and the WIDE search -- spacers of
    ``spacer_bytes`` between the candidates (the groups change along the physical address space, often only tens of GB apart),
    ``park_extra`` neighbours of a far find kept for the next fields, ``keep_free_bytes`` of device memory never touched;
    ``release_cache``: ``torch.cuda.empty_cache()`` also after a plain search that rejected more than 4 candidates.  Re-arms a
    dormant placer.  Applies to the CURRENT device's placer."""
    placer = device_placer()
    if placer is not None:
        for name, value in (("max_candidates", max_candidates), ("max_held_bytes", max_held_bytes), ("spacer_bytes", spacer_bytes),
                            ("park_extra", park_extra), ("keep_free_bytes", keep_free_bytes), ("dormant_after", dormant_after)):
            if value is not None:
                setattr(placer, name, int(value))
        if release_cache is not None:
            placer.release_cache = bool(release_cache)
        placer.dormant, placer.failed_in_a_row = False, 0
    return placer


def deal_by_roles(fields) -> Dict[str, int]:
    """``fields``: (name, is_written, nbytes) in the order of the stencil's signature -> {name: memory class}.  What the stencil
    WRITES (also what it reads and writes) is dealt alternately over the two classes starting with class 1, so that the written
    streams never all share a group; what it only reads goes to the class that holds fewer of the stencil's bytes so far (ties: class
    0, the reference's group).  Measured, profiles/r6_memory_roles.log: Laplacian in / out = 0 / 1; horizontal diffusion in / coeff /
    out = 0 / 0 / 1 (3-4 % faster than 0 / 1 / 0); tridiagonal solve inf / diag / sup / rhs / out = 0 / 0 / 1 / 0 / 1 (the best of all 16
    assignments, 0.784-0.790 of the HBM peak against 0.756-0.761 for the alternate deal and 0.67-0.68 for one class)."""
    fields = [(str(n), bool(w), int(b)) for n, w, b in fields]
    load, out, next_written = [0, 0], {}, 1
    for name, written, nbytes in fields:
        if written:
            out[name] = next_written
            load[next_written] += nbytes
            next_written = 1 - next_written
    for name, written, nbytes in fields:
        if not written:
            cls = 0 if load[0] <= load[1] else 1
            out[name] = cls
            load[cls] += nbytes
    return {name: out[name] for name, _, _ in fields}


class want:
    """``with placement.want(cls): out = gt_storage.empty(...)`` -- the big fields allocated inside go to memory class ``cls`` (0 / 1)
    instead of wherever the balance of live bytes points.  The placer deals fields in the order they are allocated; a program that
    knows the ROLES can do better: what a stencil writes belongs in the other class than what it reads (horizontal diffusion,
    512 x 1024 x 80 fp64: in / coeff / out in classes 0 / 0 / 1 is 4.7 % faster than 0 / 0 / 0 and 2 % faster than 0 / 1 / 0,
    profiles/r5_memory_groups.txt).  ``cls`` may be None (no preference); a no-op when the placer is off."""

    def __init__(self, cls: Optional[int]):
        self.cls, self.before = cls, None

    def __enter__(self):
        placer = device_placer()
        if placer is not None:
            self.before, placer.forced = placer.forced, self.cls
        return self

    def __exit__(self, *exc):
        placer = device_placer()
        if placer is not None:
            placer.forced = self.before
        return False


def class_of(array) -> Optional[int]:
    """0 / 1: the memory class the placer found for the raw block behind a storage; None: not classified (small, huge, placer off)."""
    raw = getattr(array, "_owner", None)
    if raw is None or not _PLACERS:
        return None
    device = getattr(getattr(raw, "device", None), "index", None)
    placer = _PLACERS.get(device if device is not None else _current_device())
    return None if placer is None else placer.known.get((int(raw.data_ptr()), int(raw.numel())))


def report() -> Optional[Dict[str, Any]]:
    """What the placer did so far (for a benchmark line): classes of the placed fields, searches, candidates, probes."""
    p = _PLACERS.get(_current_device()) if _PLACERS else None
    if p is None:
        return {"enabled": False, "why": _DISABLED_REASON} if _DISABLED_REASON else None
    histogram: Dict[str, int] = {}
    for r in p.rates:
        key = str(int(r // 100) * 100)
        histogram[key] = histogram.get(key, 0) + 1
    return {"enabled": True, "device": getattr(p, "device", None), "dormant": p.dormant, "pair_threshold_mode": p.threshold_mode,
            "pair_rate_histogram_100gbs": dict(sorted(histogram.items(), key=lambda kv: int(kv[0]))),
            "reference_self_pair_gbs": p.self_pair_gbs, "reference_bytes": p.reference_bytes, "fields_placed_per_class": list(p.stats["placed"]), "live_bytes_per_class": list(p.live), "searches": p.stats["searches"],
            "candidates": p.stats["candidates"], "probes": p.stats["probes"], "wanted_class_not_found": p.stats["wanted_class_not_found"],
            "spacers": p.stats.get("spacers", 0), "spacer_bytes": p.spacer_bytes, "parked_blocks": sum(len(v) for v in p.parked.values()),
            "pair_threshold_gbs": p.threshold, "max_candidates": p.max_candidates, "fields": list(p.log)[-32:]}
