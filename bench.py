#!/usr/bin/env python3
"""Headline benchmark: GLUPS of the fp64 5-point Laplacian on a 512^3 grid through the full user
path (gt4py_amd.storage -> @gtscript.stencil(backend="hip:mi300") -> FrozenStencil -> C ABI -> HIP).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one apply of the stencil over the whole 512^3 grid on synthetic input that is already
resident in HBM.  With N > 1 the SAME 512^3 grid is split over the ranks (strong scaling); a step is
then halo exchange (RCCL send/recv) + kernel(s), see DESIGN.md section 6.  Rank 0 prints ONE JSON line.

Extra objects in the line (see DESIGN.md "Measurement"):
  roofline      HBM roofline of the dominant kernel: algorithmic bytes (16 B per lattice update) / average launch
                duration measured with HIP events on the launch stream over K back-to-back launches (median and
                minimum of a second, per-launch-event pass are reported next to it), against the 8.0 TB/s nominal peak; `traffic` = HBM bytes per launch from the
                committed rocprofv3 PMC measurement IF it was taken on the kernel sources of this tree
                (profiles/hbm_traffic.json carries their hash), else null.
  cpu_baseline  the oracle's C/OpenMP restatement of gt:cpu_ifirst semantics (kind "port") timed on this
                host's cores, threads pinned, in a child process (N == 1 only).

`--workload hdiff2048` runs BASELINE.json configs[4] instead (horizontal diffusion fp64, 512 x 1024 x 80
per rank = 2048 x 2048 x 80 on the 4 x 2 grid of 8 ranks, ghost depth 2; weak scaling); the default and the
headline metric stay the Laplacian.

Every phase of an N > 1 run has a deadline: a rank that is stuck in a collective prints where and exits with
status 3 (and the launcher takes the other ranks down) instead of hanging the node.
"""

from __future__ import annotations

import argparse
import datetime
import hashlib
import json
import os
import pathlib
import statistics
import subprocess
import sys
import threading
import time

import numpy as np  # noqa: F401 - also resolves the annotations of the stencil definition below

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

BYTES_PER_LUP = 16.0  # fp64: one read + one write per lattice update
PEAK_GBS = 8000.0  # MI355X HBM3E nominal (MI355X_MICROARCH.md)
_LAP_SOURCES = ("gt4py_amd/csrc/lap5.hip.h", "gt4py_amd/csrc/lane_shift.hip.h", "gt4py_amd/csrc/common.hip.h", "gt4py_amd/csrc/Makefile")
_HDIFF_SOURCES = ("gt4py_amd/csrc/hdiff.hip.h", "gt4py_amd/csrc/hdiff_share.hip.h", "gt4py_amd/csrc/hdiff_jmarch.hip.h", "gt4py_amd/csrc/lane_shift.hip.h",
                  "gt4py_amd/csrc/common.hip.h", "gt4py_amd/csrc/Makefile")
_TRIDIAG_SOURCES = ("gt4py_amd/csrc/tridiag.hip.h", "gt4py_amd/csrc/tridiag_stack.hip.h", "gt4py_amd/csrc/common.hip.h", "gt4py_amd/csrc/Makefile")
_GENERATED_SOURCES = ("gt4py_amd/cartesian/backend/hip_codegen.py", "gt4py_amd/cartesian/backend/stage_planner.py",
                      "gt4py_amd/cartesian/backend/hip_generic.py", "gt4py_amd/csrc/rtc.hip.h")
KERNEL_SOURCES = {  # the files whose contents decide what a kernel does, per profiled workload (the keys of profiles/hbm_traffic.json)
    "lap5_f64_512": _LAP_SOURCES,
    "laplacian_f64_512x512x128_config1": _LAP_SOURCES,
    "hdiff_limiter_f32_1024x1024x80": _HDIFF_SOURCES,
    "hdiff_limiter_f32_literal32_1024x1024x80": _HDIFF_SOURCES,
    "hdiff_limiter_f64_512x1024x80": _HDIFF_SOURCES,
    "tridiagonal_f64_1024x1024x160": _TRIDIAG_SOURCES,
    "generated_vertical_advection_f64_1024x1024x160": _GENERATED_SOURCES,
    "generated_laplacian_f64_512x512x512": _GENERATED_SOURCES,
    "generated_hdiff_limiter_f64_512x1024x80": _GENERATED_SOURCES,
}
# which kernel of a rocprofv3 trace belongs to which workload: a substring of the demangled name (the two Laplacian workloads run
# the same instantiation: scripts/profile_all_kernels.sh profiles them in separate processes)
KERNEL_NEEDLES = {
    "lap5_f64_512": "lap5_strip_kernel<double, double",
    "laplacian_f64_512x512x128_config1": "lap5_strip_kernel<double, double",
    "hdiff_limiter_f32_1024x1024x80": "hdiff_share_kernel<float, double, double",
    "hdiff_limiter_f32_literal32_1024x1024x80": "hdiff_share_kernel<float, float, float",
    "hdiff_limiter_f64_512x1024x80": "hdiff_share_kernel<double, double, double",
    "tridiagonal_f64_1024x1024x160": "tridiag_",
    "generated_vertical_advection_f64_1024x1024x160": "gt4mi__vertical_advection_dycore_stage",
    "generated_laplacian_f64_512x512x512": "gt4mi_lap_notebook_stage",
    "generated_hdiff_limiter_f64_512x1024x80": "gt4mi_hdiff_limiter_field_stage",
}


def _build_flags(makefile_text: str) -> bytes:
    """The lines of csrc/Makefile that decide what the compiler makes of a kernel: the HIPCC / ARCH / FLAGS assignments with their
    continuation lines -- not its comments or the targets of tooling builds, which change without changing any kernel."""
    keep, continued = [], False
    for line in makefile_text.splitlines():
        if continued or line.split("?=")[0].split(":=")[0].strip() in ("HIPCC", "ARCH", "FLAGS"):
            keep.append(line.strip())
            continued = line.rstrip().endswith("\\")
    return "\n".join(keep).encode()


def kernel_source_hash(workload: str, read=None) -> str:
    """16 hex digits over the files that decide what the workload's kernel does (KERNEL_SOURCES); of the Makefile only the
    compiler, architecture and flags.  `read(rel) -> bytes`: another tree's files (scripts/rehash_traffic.py: `git show`)."""
    read = read or (lambda rel: (ROOT / rel).read_bytes())
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[workload]:
        h.update(rel.encode())
        data = read(rel)
        h.update(_build_flags(data.decode()) if rel.endswith("Makefile") else data)
    return h.hexdigest()[:16]


class Watchdog:
    """Deadline per phase: when one expires the process says where it was and exits with status 3.

    A rank that hangs inside a collective (communicator set-up, a send without its receive, a peer that died)
    would otherwise sit there until the node's own limit; nothing is re-executed, the process just ends."""

    def __init__(self, rank: int):
        self.rank, self._timer = rank, None
        # Once a contract-complete measurement exists (the provisional sequential form before the calibration of the
        # overlapped forms, later the headline itself) a deadline no longer costs the whole line: `safe(reason)` prints
        # what was measured (rank 0), marked "deadline_exceeded", and the process ends with status 0.
        self.safe = None
        self.scale = float(os.environ.get("GT4MI_BENCH_DEADLINE_SCALE", "1"))  # tests shorten the deadlines

    def arm(self, seconds: float, what: str) -> None:
        self.disarm()
        seconds = seconds * self.scale
        self._timer = threading.Timer(seconds, self._fire, (seconds, what))
        self._timer.daemon = True
        self._timer.start()

    def disarm(self) -> None:
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def _fire(self, seconds, what):
        status = 3
        try:
            if self.safe is not None:
                os.write(2, f"bench.py: rank {self.rank} exceeded the {seconds:.0f} s deadline of phase '{what}'; the line "
                            f"measured before that phase is printed instead (\"deadline_exceeded\")\n".encode())
                self.safe(f"phase '{what}' ran past its {seconds:.0f} s deadline")
                status = 0
            else:
                os.write(2, f"bench.py: rank {self.rank} exceeded the {seconds:.0f} s deadline of phase '{what}'; "
                            f"exiting with status 3\n".encode())
        finally:
            os._exit(status)


def copy_ceiling_gbs(steps: int = 10, nbytes: int = 1 << 30, classes=None) -> float:
    """Streaming device copy (gt4mi_stream_copy, 16-byte lanes) of 1 GiB, read + write bytes per second:
    the achievable-HBM yardstick SURVEY.md section 8d asks to report from the same run.  `classes` = (class of the source, class
    of the destination): both buffers placed by the storage allocator's memory-group placer (storage/placement.py) -- the copy
    rate with source and destination in ONE group of memory channels next to the rate with them in TWO."""
    import torch

    from gt4py_amd import _lib

    lib = _lib.load()
    if classes is None:
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    else:
        from gt4py_amd.storage import placement

        placer = placement.device_placer()
        if placer is None:
            return None
        with placement.want(classes[0]):
            src, got_src = placer.place(nbytes)
        with placement.want(classes[1]):
            dst, got_dst = placer.place(nbytes)
        if (got_src, got_dst) != tuple(classes):
            return None  # (no second group within reach on this device)
    src.fill_(1)

    def call(i):
        _lib.check("gt4mi_stream_copy", lib.gt4mi_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes,
                                                              torch.cuda.current_stream().cuda_stream))

    for i in range(2):
        call(i)
    torch.cuda.synchronize()
    ms = _time_launches(call, steps)["mean"]
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def allocator_variants(lap, frozen, shape, steps: int):
    """`value_default_allocator`, `value_allocator_off` (GLUPS of the headline kernel, whole 512^3 grid) and the memory classes the
    fields got.  The bench's own fields come from `placement.configure(...)`'s wide search plus the stencil's role hint; a program
    that only calls `gt_storage.empty(...)` gets the first of these two numbers, one that sets GT4PY_AMD_ALLOC_GROUPS=0 the second."""
    import numpy as np
    import torch

    from gt4py_amd.storage import placement

    placer = placement.device_placer()
    saved = None
    if placer is not None:
        saved = {k: getattr(placer, k) for k in ("max_candidates", "spacer_bytes", "park_extra")}
        parked, placer.parked = placer.parked, {}
    out = {}
    try:
        for key, off in (("default_allocator", False), ("allocator_off", True)):
            if placer is not None:
                placement.configure(max_candidates=6, spacer_bytes=0, park_extra=0)
            if off:
                with placement.disabled():
                    pairs = _device_fields(shape, n_pairs=2, seed=4242)
            else:
                pairs = _device_fields(shape, n_pairs=2, seed=4242)

            def call(i):
                inp, o = pairs[i % len(pairs)]
                frozen(inp=inp, out=o)

            for i in range(5):
                call(i)
            torch.cuda.synchronize()
            ms = _time_launches(call, max(int(steps), 20))["mean"]
            out[f"value_{key}"] = round(float(np.prod(GRID)) / ms / 1e6, 2)
            out[f"memory_classes_{key}"] = [[placement.class_of(i), placement.class_of(o)] for i, o in pairs]
            del pairs
            torch.cuda.empty_cache()
    finally:
        if placer is not None:
            placement.configure(**saved)
            placer.parked = parked
    out["allocator_note"] = ("value / value_by_events = fields from the wide search (placement.configure: 24 candidates, 8 GB spacers) + the "
                             "stencil's placement_hint(); value_default_allocator = plain gt_storage.empty at the allocator's defaults; "
                             "value_allocator_off = GT4PY_AMD_ALLOC_GROUPS=0; the last two are timed with HIP events around the launches: "
                             "compare them with value_by_events")
    return out


def host_cost_per_call(lap, n: int = 300):
    """Microseconds of host time per call (SURVEY.md section 8d asks for the end-to-end Python cost next to the kernel
    time): `FrozenStencil.__call__` and the validating `StencilObject.__call__` on a small domain (the cost does not
    depend on the domain; a small one keeps the device queue from filling up), launches left asynchronous."""
    import numpy as np
    import torch

    import gt4py_amd.storage as gt_storage

    inp = gt_storage.ones((34, 34, 8), np.float64, backend="hip:mi300", aligned_index=(1, 1, 0))
    out = gt_storage.zeros((34, 34, 8), np.float64, backend="hip:mi300", aligned_index=(1, 1, 0))
    origin = {"inp": (1, 1, 0), "out": (1, 1, 0)}
    frozen = lap.freeze(origin=origin, domain=(32, 32, 8))
    res = {}
    for name, fn in (("frozen_call", lambda: frozen(inp=inp, out=out)),
                     ("validated_call", lambda: lap(inp, out, origin=(1, 1, 0), domain=(32, 32, 8)))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        res[name + "_us"] = round(host / n * 1e6, 2)
    return res


def other_kernels(steps: int = 20, only=None):
    """The other kernels of the north star at their BASELINE.json sizes, through the same call path
    (storage -> stencil -> FrozenStencil), HIP-event timed.  Informational: `value` stays the Laplacian.  Every entry carries
    its own `roofline` object: `traffic` = HBM bytes per launch from the committed rocprofv3 PMC measurement of THIS tree's
    kernel sources (profiles/hbm_traffic.json, written by scripts/profile_all_kernels.sh; null when the sources have changed
    since).  `only`: a collection of entry names (the profiling script runs subsets in separate processes)."""
    import numpy as np
    import torch

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    from gt4py_amd.storage import placement

    gen = torch.Generator(device="cuda").manual_seed(2024)

    def field(shape, dtype, origin, lo=-1.0, hi=1.0, cls=None):
        # `cls`: the memory class the stencil's placement_hint() names for this field (what a stencil writes is dealt alternately
        # over the two classes, what it only reads fills up the emptier one: gt4py_amd/storage/placement.py deal_by_roles)
        f = gt_storage.empty(shape, dtype, backend="hip:mi300", aligned_index=origin, memory_class=cls)
        f.tensor.copy_(torch.rand(shape, dtype=f.tensor.dtype, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    out = {}

    def wanted(name):
        return only is None or name in only

    def run(name, obj, fields, origin, domain, bytes_per_lup, scalars=None, note=None):
        frozen = obj.freeze(origin=origin, domain=domain)
        if isinstance(fields, (list, tuple)):  # several sets of fields, rotated launch by launch
            call = lambda i: frozen(**fields[i % len(fields)], **(scalars or {}))  # noqa: E731
        else:
            call = lambda i: frozen(**fields, **(scalars or {}))  # noqa: E731
        # Warm up and time for a fixed amount of device time, not a fixed count: the clocks need a few milliseconds of
        # load to settle, and 10 + 20 launches of a 0.18 ms kernel measured it 7-12 % slow (scripts/hdiff_bench_context.py
        # next to scripts/hdiff_api_timing.py on one box: 0.206 vs 0.182 ms)
        t0 = time.perf_counter()
        n_warm = 0
        while n_warm < 10 or time.perf_counter() - t0 < 0.05:
            call(n_warm)
            n_warm += 1
            if n_warm % 16 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t0) / n_warm, 1e-5)
        n = int(min(400, max(steps, 0.04 / est)))
        t = _time_launches(call, n)
        ms = t["mean"]
        lups = float(np.prod(domain))
        gbs = bytes_per_lup * lups / (ms * 1e-3) / 1e9
        traffic, traffic_source = _committed_traffic(name)
        out[name] = {"domain": list(domain), "ms": round(ms, 4), "ms_median": round(t["median"], 4), "ms_min": round(t["min"], 4),
                     "launches_timed": t["n"], "glups": round(lups / ms / 1e6, 1), "algorithmic_bytes_per_lup": bytes_per_lup,
                     "achieved_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_GBS, 4),
                     "roofline": {"bound": "hbm", "kernel": KERNEL_NEEDLES.get(name), "achieved": round(gbs, 1), "peak": PEAK_GBS,
                                  "unit": "GB/s", "frac": round(gbs / PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                                  "algorithmic_bytes_per_launch": bytes_per_lup * lups,
                                  "traffic_over_algorithmic": round(traffic / (bytes_per_lup * lups), 4) if traffic else None}}
        sets = fields if isinstance(fields, (list, tuple)) else [fields]
        out[name]["memory_classes"] = [{k: placement.class_of(v) for k, v in fs.items()} for fs in sets]  # (None: too small to classify)
        if note:
            out[name]["note"] = note

    # BASELINE.json configs[1] as named: 512 x 512 x 128 fp64.  One field is 285 MB with its halo -- about the size of the
    # 256 MB Infinity Cache -- so the launches rotate over FOUR (inp, out) pairs (2.3 GB; SURVEY.md section 8d asks for >= 3):
    # nothing a launch reads or writes can still be cache-resident from its previous turn.
    if wanted("laplacian_f64_512x512x128_config1"):
        dom = (512, 512, 128)
        lap_obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64},
                                   device_sync=False)
        shape = (dom[0] + 2, dom[1] + 2, dom[2])
        hint = lap_obj.placement_hint()
        sets = [{"inp": field(shape, np.float64, (1, 1, 0), cls=hint["inp"]), "out": field(shape, np.float64, (1, 1, 0), cls=hint["out"])}
                for _ in range(4)]
        run("laplacian_f64_512x512x128_config1", lap_obj, sets, {k: (1, 1, 0) for k in ("inp", "out")}, dom, 16.0,
            note="BASELINE.json configs[1] at its own size, four rotating (inp, out) pairs = 2.3 GB so that the 256 MB Infinity "
                 "Cache cannot serve repeats; the headline `value` is the same kernel on 512^3")
        out["laplacian_f64_512x512x128_config1"]["rotating_pairs"] = len(sets)
        del sets
        torch.cuda.empty_cache()

    for tag, dt, dom, lit, note in (
            ("hdiff_limiter_f32_1024x1024x80", np.float32, (1024, 1024, 80), 64,
             "BASELINE.json configs[2] with the reference's default float64 literals: lap / flx / fly are float64 "
             "temporaries, every float32 operand is widened where it meets one (gtir_upcaster.py:43-143)"),
            ("hdiff_limiter_f32_literal32_1024x1024x80", np.float32, (1024, 1024, 80), 32,
             "the same stencil built with literal_float_precision=32: float32 throughout (what a model that runs in "
             "single precision sets); different arithmetic, so a different stencil -- shown next to the default"),
            ("hdiff_limiter_f64_512x1024x80", np.float64, HDIFF_SHARE, 64, "the per-rank share of BASELINE.json configs[4]")):
        if not wanted(tag):
            continue
        obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt},
                               device_sync=False, literal_float_precision=lit)
        shape = (dom[0] + 4, dom[1] + 4, dom[2])
        hint = obj.placement_hint()
        fields = {"in_field": hdiff_input(shape, dt, gen, cls=hint["in_field"]), "coeff": field(shape, dt, (2, 2, 0), 0.025, 0.025, cls=hint["coeff"]),
                  "out_field": field(shape, dt, (2, 2, 0), cls=hint["out_field"])}
        run(tag, obj, fields, {k: (2, 2, 0) for k in fields}, dom, 3.0 * np.dtype(dt).itemsize, note=note)
        del fields
    if wanted("tridiagonal_f64_1024x1024x160"):
        out.update(_tridiagonal_entry(field, steps))
        torch.cuda.empty_cache()

    # the generic executor (stencils outside the three kernel families are compiled, not rejected): the reference's
    # vertical advection and the Laplacian again, this time through the code generator
    if wanted("generated_vertical_advection_f64_1024x1024x160"):
        dom = (1024, 1024, 160)
        obj = gtscript.stencil(backend="hip:mi300", definition=_vertical_advection_dycore, externals={"BET_M": 0.5, "BET_P": 0.5},
                               device_sync=False)
        shape = (dom[0] + 1, dom[1], dom[2] + 1)
        hint = obj.placement_hint()
        fields = {n: field(shape, np.float64, (0, 0, 0), cls=hint[n]) for n in ("utens_stage", "u_stage", "wcon", "u_pos", "utens")}
        run("generated_vertical_advection_f64_1024x1024x160", obj, fields, {k: (0, 0, 0) for k in fields}, dom, 48.0,
            scalars={"dtr_stage": 3.0 / 20.0},
            note="one generated column kernel (forward + backward sweep); 5 fields read, 1 written; the forward sweep's "
                 "ccol / dcol are read back by the backward sweep: the top 144 of 160 levels stay in registers + LDS "
                 "(stage_planner.TopCache, 104 + 40), the rest makes a round trip through scratch and u_pos is read by both "
                 "sweeps.  Byte budget per lattice update: 48 algorithmic + 8 (u_pos twice) + 3.2 (16 of 160 levels of ccol / dcol "
                 "through scratch) + ~2.2 (wcon's i + 1 column, alignment) = 61.4 = the 1.28 x the counters show; at the 6.4 TB/s the "
                 "kernel moves that is 0.63 of the peak: this design's ceiling at K = 160 (registers and LDS are full; three on-chip "
                 "values for 96 levels would move 64 B) -- DESIGN.md section 4b")
        del fields
        torch.cuda.empty_cache()
    if wanted("generated_laplacian_f64_512x512x512"):
        dom = (512, 512, 512)
        obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64},
                               device_sync=False, use_kernel_library=False)
        shape = (dom[0] + 2, dom[1] + 2, dom[2])
        hint = obj.placement_hint()
        fields = {"inp": field(shape, np.float64, (1, 1, 0), cls=hint["inp"]), "out": field(shape, np.float64, (1, 1, 0), cls=hint["out"])}
        run("generated_laplacian_f64_512x512x512", obj, fields, {k: (1, 1, 0) for k in fields}, dom, 16.0,
            note="the headline stencil through the code generator instead of the hand-written kernel")
        del fields
        torch.cuda.empty_cache()
    if wanted("generated_hdiff_limiter_f64_512x1024x80"):
        dom = HDIFF_SHARE
        obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64},
                               device_sync=False, use_kernel_library=False)
        shape = (dom[0] + 4, dom[1] + 4, dom[2])
        hint = obj.placement_hint()
        fields = {"in_field": hdiff_input(shape, np.float64, gen, cls=hint["in_field"]), "coeff": field(shape, np.float64, (2, 2, 0), 0.025, 0.025, cls=hint["coeff"]),
                  "out_field": field(shape, np.float64, (2, 2, 0), cls=hint["out_field"])}
        run("generated_hdiff_limiter_f64_512x1024x80", obj, fields, {k: (2, 2, 0) for k in fields}, dom, 24.0,
            note="the flux-limited horizontal diffusion through the code generator: one strip kernel, lap / flx / fly computed "
                 "once per point and passed between lanes with DPP shifts (hip_codegen._emit_shared_kernel)")
        del fields
        torch.cuda.empty_cache()
    return out


def _placement_class(array):
    from gt4py_amd.storage import placement

    return placement.class_of(array)


def _tridiagonal_entry(field, steps: int):
    """BASELINE.json configs[3]: the vertical tridiagonal solve on 1024 x 1024 x 160 fp64, timed on the SURVEY section 8d inputs
    EVERY launch: the solve overwrites `sup` and `rhs` in place, so each launch is preceded by a restore of both from pristine
    copies -- outside the timed interval (an event pair around every launch; the kernel runs 1.6 ms, an event costs microseconds)
    -- and the launches rotate over TWO sets of the five fields.  (Rounds 1-4 timed it on whatever the previous launch had left in
    sup / rhs: after hundreds of forward sweeps no longer the specified operands)  The speed of this
    kernel depends on the allocation set (0.60-0.73 of the peak, profiles/r4_tridiag_translation.txt): both sets are reported."""
    import numpy as np
    import torch

    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    name, dom = "tridiagonal_f64_1024x1024x160", (1024, 1024, 160)
    obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64},
                           device_sync=False)
    hint = obj.placement_hint()  # inf / diag / sup / rhs / out -> 0 / 0 / 1 / 0 / 1: BOTH sets get the same arrangement
    sets = [{"inf": field(dom, np.float64, (0, 0, 0), cls=hint["inf"]), "diag": field(dom, np.float64, (0, 0, 0), 4.0, 5.0, cls=hint["diag"]),
             "sup": field(dom, np.float64, (0, 0, 0), cls=hint["sup"]), "rhs": field(dom, np.float64, (0, 0, 0), -10.0, 10.0, cls=hint["rhs"]),
             "out": field(dom, np.float64, (0, 0, 0), cls=hint["out"])} for _ in range(2)]
    pristine = [{k: fs[k].tensor.clone() for k in ("sup", "rhs")} for fs in sets]
    frozen = obj.freeze(origin={k: (0, 0, 0) for k in sets[0]}, domain=dom)

    def restore(i):
        fs, keep = sets[i % 2], pristine[i % 2]
        fs["sup"].tensor.copy_(keep["sup"])
        fs["rhs"].tensor.copy_(keep["rhs"])

    for i in range(4):  # warm-up: both sets, clocks
        restore(i)
        frozen(**sets[i % 2])
    torch.cuda.synchronize()
    n = max(int(steps), 20)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i, (a, b) in enumerate(pairs):
        restore(i)
        a.record()
        frozen(**sets[i % 2])
        b.record()
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) for a, b in pairs]
    ms = sum(per) / len(per)
    by_set = [sum(per[s::2]) / len(per[s::2]) for s in (0, 1)]
    lups = float(np.prod(dom))
    gbs = 56.0 * lups / (ms * 1e-3) / 1e9
    traffic, traffic_source = _committed_traffic(name)
    entry = {"domain": list(dom), "ms": round(ms, 4), "ms_median": round(statistics.median(per), 4), "ms_min": round(min(per), 4),
             "launches_timed": n, "glups": round(lups / ms / 1e6, 1), "algorithmic_bytes_per_lup": 56.0,
             "achieved_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_GBS, 4),
             "ms_by_allocation_set": [round(v, 4) for v in by_set],
             "frac_of_hbm_peak_by_allocation_set": [round(56.0 * lups / (v * 1e-3) / 1e9 / PEAK_GBS, 4) for v in by_set],
             "field_addresses_mod_4MiB": [[int(f.ptr % (4 << 20)) for f in fs.values()] for fs in sets],
             "memory_classes_by_allocation_set": [{k: _placement_class(v) for k, v in fs.items()} for fs in sets],
             "memory_classes_wanted": hint,
             "inputs": "SURVEY.md section 8d (diag ~ U[4, 5), inf, sup ~ U[-1, 1), rhs ~ U[-10, 10)), sup and rhs restored from "
                       "pristine copies before EVERY launch, outside the timed interval; two rotating sets of the five fields",
             "roofline": {"bound": "hbm", "kernel": KERNEL_NEEDLES.get(name), "achieved": round(gbs, 1), "peak": PEAK_GBS, "unit": "GB/s",
                          "frac": round(gbs / PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                          "algorithmic_bytes_per_launch": 56.0 * lups,
                          "traffic_over_algorithmic": round(traffic / (56.0 * lups), 4) if traffic else None},
             "note": "the speed of the K-strided column kernels depends on which memory groups their fields live in (0.67 of the HBM peak "
                     "with all five in one group, 0.78-0.79 with the written streams split 2 + 1 over two: profiles/r6_memory_roles.log); both "
                     "allocation sets are allocated by the stencil's placement_hint().  The backward sweep re-reads the part of sup', rhs' "
                     "that does not fit on chip (144 of 160 levels stay in registers + LDS)"}
    return {name: entry}


def _vertical_advection_dycore(utens_stage: Field[np.float64], u_stage: Field[np.float64], wcon: Field[np.float64],  # noqa: F821
                               u_pos: Field[np.float64], utens: Field[np.float64], *, dtr_stage: float):  # noqa: F821
    """/root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/stencil_definitions.py:235-313."""
    from __externals__ import BET_M, BET_P

    with computation(FORWARD):  # noqa: F821
        with interval(0, 1):  # noqa: F821
            gcv = 0.25 * (wcon[1, 0, 1] + wcon[0, 0, 1])
            cs = gcv * BET_M
            ccol = gcv * BET_P
            bcol = dtr_stage - ccol[0, 0, 0]
            correction_term = -cs * (u_stage[0, 0, 1] - u_stage[0, 0, 0])
            dcol = dtr_stage * u_pos[0, 0, 0] + utens[0, 0, 0] + utens_stage[0, 0, 0] + correction_term
            divided = 1.0 / bcol[0, 0, 0]
            ccol = ccol[0, 0, 0] * divided
            dcol = dcol[0, 0, 0] * divided
        with interval(1, -1):  # noqa: F821
            gav = -0.25 * (wcon[1, 0, 0] + wcon[0, 0, 0])
            gcv = 0.25 * (wcon[1, 0, 1] + wcon[0, 0, 1])
            as_ = gav * BET_M
            cs = gcv * BET_M
            acol = gav * BET_P
            ccol = gcv * BET_P
            bcol = dtr_stage - acol[0, 0, 0] - ccol[0, 0, 0]
            correction_term = -as_ * (u_stage[0, 0, -1] - u_stage[0, 0, 0]) - cs * (u_stage[0, 0, 1] - u_stage[0, 0, 0])
            dcol = dtr_stage * u_pos[0, 0, 0] + utens[0, 0, 0] + utens_stage[0, 0, 0] + correction_term
            divided = 1.0 / (bcol[0, 0, 0] - ccol[0, 0, -1] * acol[0, 0, 0])
            ccol = ccol[0, 0, 0] * divided
            dcol = (dcol[0, 0, 0] - (dcol[0, 0, -1]) * acol[0, 0, 0]) * divided
        with interval(-1, None):  # noqa: F821
            gav = -0.25 * (wcon[1, 0, 0] + wcon[0, 0, 0])
            as_ = gav * BET_M
            acol = gav * BET_P
            bcol = dtr_stage - acol[0, 0, 0]
            correction_term = -as_ * (u_stage[0, 0, -1] - u_stage[0, 0, 0])
            dcol = dtr_stage * u_pos[0, 0, 0] + utens[0, 0, 0] + utens_stage[0, 0, 0] + correction_term
            divided = 1.0 / (bcol[0, 0, 0] - ccol[0, 0, -1] * acol[0, 0, 0])
            dcol = (dcol[0, 0, 0] - (dcol[0, 0, -1]) * acol[0, 0, 0]) * divided
    with computation(BACKWARD):  # noqa: F821
        with interval(-1, None):  # noqa: F821
            datacol = dcol[0, 0, 0]
            utens_stage = dtr_stage * (datacol - u_pos[0, 0, 0])
        with interval(0, -1):  # noqa: F821
            datacol = dcol[0, 0, 0] - ccol[0, 0, 0] * datacol[0, 0, 1]
            utens_stage = dtr_stage * (datacol - u_pos[0, 0, 0])


# ---- CPU baseline: a child process with pinned OpenMP threads -------------------------------------------
def usable_cores() -> int:
    """min(affinity mask, cgroup CPU quota): the cores this container may really keep busy."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    try:  # CFS bandwidth quota of the container, e.g. "1600000 100000" = 16 cores
        quota, period = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def host_model() -> str:
    try:
        for line in pathlib.Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline_child(seconds_budget: float) -> None:
    """Runs in its own process (no GPU, no torch: the OpenMP runtime starts with the binding set by the
    parent): the oracle's C/OpenMP port on the full 512^3 grid in batches; prints one JSON object.

    The whole grid is used on purpose: a 512x512x64 slab (2 x 135 MB) stays resident in the 512 MB of
    L3 of a dual EPYC 9575F host and reports a cache bandwidth, not the workload's."""
    import numpy as np

    from oracle import cpu_ifirst

    lib = None
    try:  # rebuild for this host's ISA when a compiler is around; else use the prebuilt library
        path = cpu_ifirst.build(march="native", out=pathlib.Path("/tmp") / f"libcpu_ifirst_{os.getpid()}.so")
        lib = cpu_ifirst.load(path)
    except Exception:
        if cpu_ifirst.available():
            lib = cpu_ifirst.load()
    if lib is None:
        print(json.dumps(None))
        return
    cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    lib.oracle_set_threads(cores)
    dom = GRID
    rng = np.random.default_rng(1337)
    inp = np.asfortranarray(rng.uniform(-1, 1, (dom[0] + 2, dom[1] + 2, dom[2])))
    out = np.asfortranarray(np.zeros_like(inp))
    for _ in range(3):  # page touch + warm-up
        cpu_ifirst.lap5_f64(inp, out, (1, 1, 0), (1, 1, 0), dom, lib=lib)
    batch, rates, t_start = 10, [], time.perf_counter()
    while True:
        t0 = time.perf_counter()
        for _ in range(batch):
            cpu_ifirst.lap5_f64(inp, out, (1, 1, 0), (1, 1, 0), dom, lib=lib)
        rates.append(dom[0] * dom[1] * dom[2] * batch / (time.perf_counter() - t0) / 1e9)
        if time.perf_counter() - t_start > seconds_budget or len(rates) >= 60:
            break
    dt = time.perf_counter() - t_start
    med = statistics.median(rates)
    # SURVEY.md section 8d also asks for the numpy restatement, single thread: what the reference's `numpy` backend does for
    # this stencil (five ufunc passes with full-size temporaries), on the same grid in numpy's own (K-contiguous) layout
    numpy_line = None
    try:
        from oracle import ref_numpy

        nk = min(128, dom[2])  # a 512 x 512 x 128 slab (BASELINE configs[1]: 268 MB per array, far beyond one core's L3 share)
        a = np.ascontiguousarray(inp[:, :, :nk])
        b = np.zeros_like(a)
        ref_numpy.laplacian(a, b)  # page touch + warm-up
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            ref_numpy.laplacian(a, b)
            times.append(time.perf_counter() - t0)
        numpy_line = {"value": round(dom[0] * dom[1] * nk / min(times) / 1e9, 4), "unit": "GLUPS", "cores": 1,
                      "sample": f"oracle/ref_numpy.laplacian (the numpy backend's statement-by-statement evaluation: five ufunc passes "
                                f"with full-size temporaries) on a {dom[0]}x{dom[1]}x{nk} slab, K-contiguous, best of 3 applies "
                                f"({min(times):.2f} s each)"}
        del a, b
    except Exception as ex:  # the second baseline must not take the first one down
        numpy_line = {"error": repr(ex)}
    print(json.dumps({
        "value": round(med, 4),
        "unit": "GLUPS",
        "cores": lib.oracle_max_threads(),
        "kind": "port",
        "sample": f"fp64 5-pt Laplacian on the full {dom[0]}x{dom[1]}x{dom[2]} grid, {len(rates)} batches of {batch} "
                  f"applies in {dt:.1f} s (median batch), C/OpenMP restatement of gt:cpu_ifirst semantics "
                  f"(oracle/cpu_ifirst.c), I-contiguous, threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores); "
                  f"{lib.oracle_max_threads()} threads = what the container may keep busy (min of affinity mask and cgroup CPU "
                  f"quota) of the host's {os.cpu_count()} logical CPUs",
        "host_logical_cpus": os.cpu_count(),
        "gb_per_s": round(med * BYTES_PER_LUP, 2),
        "batch_glups_min_max": [round(min(rates), 3), round(max(rates), 3)],
        "spread_pct": round(100.0 * (max(rates) - min(rates)) / med, 1),
        # (the host is shared: a neighbour's burst shows in a few batches; the value is the MEDIAN batch, and the spread of the
        # middle 80 % of the batches says how stable that is)
        "spread_p10_p90_pct": round(100.0 * (sorted(rates)[int(0.9 * (len(rates) - 1))] - sorted(rates)[int(0.1 * (len(rates) - 1))]) / med, 1),
        "host": host_model(),
        "numpy_single_thread": numpy_line,
    }))


def cpu_baseline(seconds_budget: float = 12.0):
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_DYNAMIC="false")
    proc = subprocess.run([sys.executable, str(pathlib.Path(__file__).resolve()), "--cpu-baseline-child", str(seconds_budget)],
                          env=env, capture_output=True, text=True, timeout=seconds_budget * 6 + 120)
    if proc.returncode != 0:
        raise RuntimeError(f"cpu baseline child failed: {proc.stderr[-500:]}")
    return json.loads(proc.stdout.strip().splitlines()[-1])


def _committed_traffic(workload: str):
    """HBM bytes per launch from the committed PMC measurement -- only when it was taken on these kernel sources."""
    f = ROOT / "profiles" / "hbm_traffic.json"
    try:
        entry = json.loads(f.read_text()).get(workload)
    except Exception:
        return None, "no committed measurement"
    if not isinstance(entry, dict):
        return None, "committed measurement carries no kernel-source hash"
    if entry.get("kernel_source_sha") != kernel_source_hash(workload):
        return None, (f"committed measurement was taken on other kernel sources ({entry.get('kernel_source_sha')} at "
                      f"{entry.get('git_sha')}, tree has {kernel_source_hash(workload)}): re-run scripts/profile_bench.sh")
    return entry.get("bytes_per_launch"), f"rocprofv3 PMC, {entry.get('source')}, git {entry.get('git_sha')}"


# ---- N > 1 -----------------------------------------------------------------------------------------------
# Agreement between ranks, slowest-rank timing, wall-clock budgets, the best-first calibration order and the fall-back ladder of
# the halo transport (direct -> direct-fenced -> rccl) live in gt4py_amd/distributed/calibrate.py (pure Python, torch-free at
# import); the set-up of the two decomposed workloads -- fields, exchangers, canary, calibration, form checks, the timed step --
# in gt4py_amd/distributed/workloads.py (moved there in round 6: importable and unit-testable without a subprocess).  The names
# stay reachable as bench.<name> for the tests and scripts that grew up with them here.
from gt4py_amd.distributed.calibrate import (FailedOnSomeRank, WallBudget, _agree, _slowest_rank_ms, best_of, calibrate_laplacian,  # noqa: E402,F401
                                             calibrate_transports, calibration_line_keys, calibration_seconds, direct_fenced,
                                             direct_mode, direct_step_down, hdiff_calibration_order, ladder_line_keys,
                                             lap_calibration_order, lap_candidate_of, lap_key, measure_candidate, run_calibration)
from gt4py_amd.distributed.workloads import (CANARY_STRESS_EPOCHS, CHECK_EPOCHS, DIRECT_TIMEOUT_MS, GRID, HDIFF_GLOBAL, HDIFF_SHARE,  # noqa: E402,F401
                                             _device_fields, _lap_definition, _native_comm, _setup_distributed_laplacian,
                                             _setup_hdiff2048, _test_hang, _time_launches, direct_canary, gather_rank_proof,
                                             hdiff_input, transport_fallback_banner)


def decomposed_line_keys(proof, transport_fallback: bool, n_gpus: int, timestep) -> dict:
    """Top-level keys every N > 1 (or self-loop) line carries: how many ranks RCCL itself reports and on which devices,
    whether the run fell back from the native transport, and the communication-avoiding steppers beside the headline."""
    proof = proof or {}
    out = {"rccl_nranks": proof.get("rccl_nranks"), "rank_devices": proof.get("rank_devices"),
           "rccl_matches_n_gpus": bool(proof) and proof.get("rccl_nranks") == n_gpus and bool(proof.get("rccl_ranks_agree"))
           and len(proof.get("rank_devices") or []) == n_gpus,
           "transport_fallback": bool(transport_fallback)}
    if timestep is not None:
        out["extra"] = timestep
    return out


def main() -> None:
    if len(sys.argv) >= 2 and sys.argv[1] == "--cpu-baseline-child":
        cpu_baseline_child(float(sys.argv[2]) if len(sys.argv) > 2 else 12.0)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("lap512", "hdiff2048"), default="lap512",
                    help="lap512 (default, the headline metric) or hdiff2048 = BASELINE.json configs[4] (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allocator-variants", action="store_true",
                    help="skip value_default_allocator / value_allocator_off (profiling runs: every launch of the headline kernel in the "
                         "trace is then one of the timed workload's)")
    ap.add_argument("--no-other-kernels", action="store_true",
                    help="skip the informational hdiff / tridiagonal lines (N=1 only)")
    ap.add_argument("--dist-selfloop", action="store_true",
                    help="1-GPU rehearsal of the N>1 step: periodic domain whose halo messages go to the "
                         "rank itself through RCCL (not the headline metric)")
    ap.add_argument("--selfloop-grid", default="",
                    help="with --dist-selfloop and lap512: PIxPJ, e.g. 4x2 -- the share of one rank of that process grid "
                         "(512/PI x 512/PJ x 512), periodic along every cut axis (W / E neighbours too)")
    ap.add_argument("--selfloop-ranks", type=int, default=1,
                    help="with --dist-selfloop and lap512: shrink J to 512/N, the per-rank share of an N-GPU run")
    args = ap.parse_args()

    import numpy as np
    import torch

    # Native libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on
    # stdout, so everything else is routed to stderr until the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dog = Watchdog(rank)
    _ACTIVE["dog"] = dog
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # GT4MI_BENCH_ONE_DEVICE=1 (rehearsals only, never a measurement): the N ranks of `torchrun --nproc-per-node N` all on device 0
    # -- RCCL refuses that, so the process group is gloo, the native communicator has no RCCL behind it and the faces travel through
    # the direct transport only (hipIpc between the processes).  What it is for: the N > 1 control flow of this program -- every
    # collective, the budgeted calibration, the agreement on failures, the keys of the line -- with N REAL ranks before a node
    # with N devices ever runs it.  The times it prints are those of N processes sharing one device.
    one_device = os.environ.get("GT4MI_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
        os.environ["GT4MI_BENCH_TRANSPORTS"] = "direct"
        # N kernels that wait for each other share ONE device's wave slots: the units of a one-launch step that wait for a face
        # hold theirs, and with shares of 512 levels three ranks that run a kernel ahead of the fourth fill every slot of the chip
        # with waiting units -- the fourth never gets to push (seen at N = 4 on the 4 x 1 grid: a resource deadlock that a device
        # per rank cannot have).  A slab of 32 levels keeps all ranks' units together below the chip's 1 280 workgroup slots.
        global GRID, HDIFF_SHARE, HDIFF_GLOBAL
        levels = int(os.environ.get("GT4MI_BENCH_ONE_DEVICE_LEVELS", "32"))
        # (configs[4]'s share likewise: eight full-size shares on one device took 17-96 s per candidate in round 4)
        from gt4py_amd.distributed import workloads as _workloads

        _workloads.set_levels(levels)
        GRID, HDIFF_SHARE, HDIFF_GLOBAL = _workloads.GRID, _workloads.HDIFF_SHARE, _workloads.HDIFF_GLOBAL
    torch.cuda.set_device(local_rank)
    # GT4MI_BENCH_FORCE_DISTRIBUTED=1: take the N > 1 code path with a world of ONE rank (process group, collectives,
    # communicator through the broadcast, calibration, line keys) -- the rehearsal a 1-GPU box allows of everything in that
    # path except a message to another device (scripts / tests only; never the headline)
    distributed = world > 1 or os.environ.get("GT4MI_BENCH_FORCE_DISTRIBUTED", "0") == "1"
    ctx = {"world": world, "rank": rank, "local_rank": local_rank, "distributed": distributed, "dog": dog, "one_device": one_device}
    if one_device:
        ctx["collective_device"] = "cpu"
    if os.environ.get("GT4MI_BENCH_DIRECT_MODE") in ("direct-fenced", "rccl"):
        # start further down the ladder (scripts: what the fenced mode costs on the self-loop)
        ctx["direct_mode"] = os.environ["GT4MI_BENCH_DIRECT_MODE"]
        if ctx["direct_mode"] == "rccl":
            ctx["direct_dropped"] = "GT4MI_BENCH_DIRECT_MODE"
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dog.arm(300, "torch.distributed rendezvous (init_process_group)")
        if one_device:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=240))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=240))
        ctx["dist"] = dist
        dog.arm(180, "first collective (barrier)")
        dist.barrier()
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; running on {n_gpus} GPU(s)", file=sys.stderr)

    from gt4py_amd import _lib
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.distributed import NativeHaloExchanger

    decomposed = distributed or args.dist_selfloop
    extras = {"exchangers": []}
    if args.workload == "hdiff2048":
        bytes_per_lup, kernel_name = 24.0, "hdiff_jmarch_kernel<double,...>"
        metric = "GLUPS (lattice updates/s) fp64 horizontal diffusion 2048x2048x80 on the 4x2 grid (BASELINE.json configs[4])"
        scaling = "weak"
    else:
        bytes_per_lup, kernel_name = BYTES_PER_LUP, "lap5_strip_kernel<double,double,0,2,8,*>"
        metric, scaling = "GLUPS (lattice updates/s) fp64 5-pt Laplacian 512^3", "strong"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(step, kernel_step, local_domain, total_lups, config, phase=""):
        """Warm-up, EXACTLY --steps timed steps between barriers (slowest rank), the dominant kernel's launch durations from
        HIP events -> the contract's line (without the informational sections)."""
        dog.arm(120 + 2.0 * args.warmup, "warm-up steps" + phase)
        for i in range(args.warmup):
            step(i)
        barrier()
        dog.arm(120 + 2.0 * args.steps, "timed steps" + phase)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.get("collective_device", "cuda"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # dominant kernel: launch durations from HIP events on the launch stream (>= 50 launches when --steps allows)
        dog.arm(300, "kernel timing" + phase)
        torch.cuda.synchronize()
        for i in range(3):
            kernel_step(i)
        torch.cuda.synchronize()
        kt = _time_launches(kernel_step, max(args.steps, 50))  # (>= 50 timed launches)
        kernel_ms = kt["mean"]
        local_lups = float(np.prod(local_domain))
        achieved = bytes_per_lup * local_lups / (kernel_ms * 1e-3) / 1e9
        glups = total_lups * args.steps / elapsed / 1e9
        return {
            "metric": metric,
            "value": round(glups, 2),
            "unit": "GLUPS",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": config,
            "pct_hbm_roofline": round(100.0 * glups * bytes_per_lup / (PEAK_GBS * n_gpus), 2),
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 1),
                "peak": PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / PEAK_GBS, 4),
                "traffic": None,
                "traffic_source": "not the profiled workload",
                "kernel_ms": round(kernel_ms, 5),
                "kernel_ms_stats": {k: (round(v, 5) if k != "n" else v) for k, v in kt.items()},
                "algorithmic_bytes_per_launch": bytes_per_lup * local_lups,
                "measured_copy_gbs": None,
            },
            "device": _lib.device_info(),
        }

    def emit(line) -> None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)  # anything native code prints while tearing down goes to stderr again

    def keep_safe(line, note: str) -> None:
        """From here on a phase that overruns its deadline (a rank stuck in a collective of a never-rehearsed form, an
        informational section that hangs) costs that phase, not the measurement: Watchdog._fire prints `line` and exits 0."""
        snapshot = json.dumps(line) if rank == 0 else None

        def safe(reason):
            if snapshot is not None:
                out = json.loads(snapshot)
                out["deadline_exceeded"] = f"{reason}; {note}"
                os.write(saved_stdout, (json.dumps(out) + "\n").encode())

        dog.safe = safe

    def provisional(step, kernel_step, local_domain, total_lups, config, proof):
        """Called by the set-up of a decomposed workload BEFORE it calibrates the overlapped forms: the plain sequence
        (exchange on the caller's stream, then one launch over the whole local domain) measured by the contract."""
        line = measure(step, kernel_step, local_domain, total_lups, config, " (provisional: sequential form)")
        line["provisional"] = ("the sequential form (exchange, then one launch over the whole local domain), measured before the "
                               "calibration of the overlapped forms; printed only because a later phase overran its deadline")
        line.update(decomposed_line_keys(proof, False, n_gpus, None))
        keep_safe(line, "the overlapped forms were not measured")

    if decomposed and os.environ.get("GT4MI_BENCH_PROVISIONAL", "1") == "1":
        ctx["provisional"] = provisional
    if args.workload == "hdiff2048":
        step, kernel_step, local_domain, config, extras = _setup_hdiff2048(args, ctx)
        total_lups = extras["total_lups"]
    elif not decomposed:
        # The storage allocator deals big fields over the device's memory groups (gt4py_amd/storage/placement.py: `in` and `out` of a
        # stencil in different groups of memory channels are worth 2 % on this kernel, 13 % on the tridiagonal solve).  A program that
        # is about to allocate the fields of bandwidth-bound kernels may widen the allocator's search for a second group: up to 24
        # candidates per field instead of 6, from the fifth on each behind an 8 GB spacer that is never touched (the groups change
        # along the physical address space; everything but the chosen block and a few parked neighbours is released when a search ends).
        from gt4py_amd.storage import placement

        placement.configure(max_candidates=int(os.environ.get("GT4MI_BENCH_GROUP_SEARCH", "24")),
                            spacer_bytes=int(os.environ.get("GT4MI_BENCH_GROUP_SPACER_GB", "8")) << 30, park_extra=8)
        lap = gtscript.stencil(backend="hip:mi300", definition=_lap_definition(), dtypes={"T": np.float64},
                               device_sync=False)
        origin = {"inp": (1, 1, 0), "out": (1, 1, 0)}
        shape = (GRID[0] + 2, GRID[1] + 2, GRID[2])
        pairs = _device_fields(shape, n_pairs=2, seed=1337, hint=lap.placement_hint())  # rotate pairs: nothing survives in MALL/L2
        frozen = lap.freeze(origin=origin, domain=GRID)

        def step(i):
            inp, out = pairs[i % len(pairs)]
            frozen(inp=inp, out=out)

        local_domain = GRID
        kernel_step = step
        config = {"workload": "fp64 5-point Laplacian 512x512x512 (examples/lap_cartesian_vs_next.ipynb cell 7), "
                              "origin (1,1,0), hip:mi300 storage layout", "grid": list(GRID), "decomposition": "1x1",
                  "call_path": "FrozenStencil",
                  "memory_classes_of_the_fields": [[placement.class_of(i), placement.class_of(o)] for i, o in pairs]}
        total_lups = float(np.prod(GRID))
    else:
        step, kernel_step, local_domain, config, extras = _setup_distributed_laplacian(args, ctx)
        total_lups = extras["total_lups"]

    line = measure(step, kernel_step, local_domain, total_lups, config)
    headline = args.workload == "lap512" and not decomposed
    if headline:
        line["roofline"]["traffic"], line["roofline"]["traffic_source"] = _committed_traffic("lap5_f64_512")
    if decomposed:
        line.update(decomposed_line_keys(extras.get("proof"), bool(extras.get("transport_fallback")), n_gpus, None))
        line.update(extras.get("calibration") or {})
    keep_safe(line, "the sections after the headline measurement are missing")

    exchangers = extras.get("exchangers") or []
    if exchangers and isinstance(exchangers[0], NativeHaloExchanger):
        config["side_stream_concurrent"] = exchangers[0].concurrent
        if getattr(exchangers[0], "transport", "rccl") == "direct":  # (a wait that ran out of time means garbage was timed)
            status = [ex.direct_status() for ex in exchangers]
            config["direct_transport_status"] = {"timed_out": any(st["timed_out"] for st in status),
                                                 "exchanges": sum(st["exchanges"] for st in status)}
    timestep = None
    _test_hang(dog, "informational")
    if callable(extras.get("timestep")):  # collective: every rank runs it
        dog.arm(420, "communication-avoiding time steppers (informational)")
        try:
            timestep = extras["timestep"]()
        except Exception as ex:
            print(f"rank {rank}: time-stepper measurement failed ({ex!r})", file=sys.stderr)
    if rank == 0:
        dog.arm(900, "informational kernels and CPU baseline")
        if not decomposed:
            # what a plain streaming copy reaches on this device in this run (not the bar, the context)
            line["roofline"]["measured_copy_gbs"] = round(copy_ceiling_gbs(), 1)
            if headline:  # the same copy with its two buffers in one memory group and in two (None: no second group in reach)
                try:
                    same, across = copy_ceiling_gbs(classes=(0, 0)), copy_ceiling_gbs(classes=(0, 1))
                    line["roofline"]["measured_copy_gbs_by_memory_groups"] = {"one_group": same and round(same, 1), "two_groups": across and round(across, 1)}
                except Exception as ex:
                    print(f"copy ceiling by memory groups failed: {ex!r}", file=sys.stderr)
        if decomposed:
            line.update(decomposed_line_keys(extras.get("proof"), bool(extras.get("transport_fallback")), n_gpus, timestep))
        if headline:
            try:
                line["host_cost_per_call"] = host_cost_per_call(lap)
            except Exception as ex:
                print(f"host_cost_per_call failed: {ex!r}", file=sys.stderr)
        if headline and not args.no_allocator_variants:
            # What a drop-in user gets: the headline kernel on fields from the storage allocator AT ITS
            # DEFAULTS (6 plain candidates per search, no spacers, no parked neighbours, no role hints: plain gt_storage.empty) and
            # with the memory-group placer OFF -- 5 + 20 launches each, outside the contract's timed steps.
            try:
                # (event-timed like the two figures below -- `value` itself is wall-clock over the K contract steps, barriers and
                # launch gaps included, and runs 1 % under it)
                line["value_by_events"] = round(float(np.prod(GRID)) / line["roofline"]["kernel_ms"] / 1e6, 2)
                line.update(allocator_variants(lap, frozen, shape, args.steps))
            except Exception as ex:
                print(f"allocator_variants failed: {ex!r}", file=sys.stderr)
        if headline and not args.no_other_kernels:
            try:
                line["other_kernels"] = other_kernels()
            except Exception as ex:
                line["other_kernels"] = None
                print(f"other_kernels failed: {ex!r}", file=sys.stderr)
        if headline:
            from gt4py_amd.storage import placement as _placement

            groups = _placement.report() or {}  # what the allocator's placer did for the big fields of this run (per-field records:
            groups.pop("fields", None)          # 3 KB -- left out of the line; placement.report() has them)
            line["memory_groups"] = groups
        if headline and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = None
                print(f"cpu_baseline failed: {ex!r}", file=sys.stderr)
        if headline:
            # LAST key of the line (a record that keeps only the tail of a long line keeps this): every BASELINE config's figure
            ok = line.get("other_kernels") or {}
            line["summary"] = {"lap5_f64_512": {"glups": line["value"], "frac": line["roofline"]["frac"],
                                                "traffic_over_algorithmic": round(line["roofline"]["traffic"] / line["roofline"]["algorithmic_bytes_per_launch"], 4)
                                                if line["roofline"].get("traffic") else None,
                                                "glups_by_events": line.get("value_by_events"),
                                                "glups_default_allocator": line.get("value_default_allocator"),
                                                "glups_allocator_off": line.get("value_allocator_off")},
                               **{name: {"glups": e.get("glups"), "frac": e.get("frac_of_hbm_peak"),
                                         "traffic_over_algorithmic": (e.get("roofline") or {}).get("traffic_over_algorithmic"),
                                         **({"frac_by_allocation_set": e["frac_of_hbm_peak_by_allocation_set"]}
                                            if "frac_of_hbm_peak_by_allocation_set" in e else {})}
                                  for name, e in ok.items() if isinstance(e, dict)},
                               "cpu_baseline_glups": (line.get("cpu_baseline") or {}).get("value")}
        dog.safe = None  # the complete line is about to be printed: a later deadline must not print a second one
        emit(line)
    dog.safe = lambda reason: None  # the line is out: trouble while tearing down no longer turns into status 3
    if distributed:
        dog.arm(120, "final barrier and process-group teardown")
        dist.barrier()
        dist.destroy_process_group()
    dog.disarm()


_ACTIVE = {"dog": None}


def _guarded_main() -> None:
    """An exception after a contract-complete measurement exists prints that measurement (rank 0) instead of losing it; the
    other ranks then run into their deadlines and end the same way (Watchdog.safe)."""
    try:
        main()
    except Exception:
        dog = _ACTIVE["dog"]
        if dog is None or dog.safe is None:
            raise
        import traceback

        traceback.print_exc()
        sys.stderr.flush()
        dog.disarm()
        dog.safe("an exception ended the run (traceback on stderr)")
        os._exit(0)


if __name__ == "__main__":
    _guarded_main()
