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
GRID = (512, 512, 512)
HDIFF_SHARE = (512, 1024, 80)  # per-rank share of BASELINE.json configs[4]
HDIFF_GLOBAL = (2048, 2048, 80)
# how long a device-side wait of the direct transport may take in THIS program before its plan fails (the library's default is
# 30 s): a broken direct transport costs the calibration this much per wait, then its forms are dropped
DIRECT_TIMEOUT_MS = int(os.environ.get("GT4MI_BENCH_DIRECT_TIMEOUT_MS", "2000"))
_LAP_SOURCES = ("gt4py_amd/csrc/lap5.hip.h", "gt4py_amd/csrc/lane_shift.hip.h", "gt4py_amd/csrc/common.hip.h", "gt4py_amd/csrc/Makefile")
_HDIFF_SOURCES = ("gt4py_amd/csrc/hdiff.hip.h", "gt4py_amd/csrc/hdiff_jmarch.hip.h", "gt4py_amd/csrc/lane_shift.hip.h",
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


# ---------------------------------------------------------------------------------------------------------
def _lap_definition():
    from gt4py_amd.cartesian.backend import hip_templates

    return hip_templates.lap_notebook


def _device_fields(shape, n_pairs, seed, origin=(1, 1, 0), hint=None):
    """`n_pairs` (inp, out) pairs in HBM with the hip:mi300 layout; inp ~ U[-1, 1), seeded on device.  `hint`: the stencil's
    `placement_hint()` -- {"inp": class, "out": class} -- for the storage allocator (None: its own deal by live bytes)."""
    import numpy as np
    import torch

    import gt4py_amd.storage as gt_storage

    pairs = []
    gen = torch.Generator(device="cuda").manual_seed(seed)
    hint = hint or {}
    for _ in range(n_pairs):
        inp = gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=origin, memory_class=hint.get("inp"))
        out = gt_storage.zeros(shape, np.float64, backend="hip:mi300", aligned_index=origin, memory_class=hint.get("out"))
        inp.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1)
        pairs.append((inp, out))
    return pairs


def _time_launches(fn, steps):
    """Launch durations (ms) from HIP events on the launch stream.

    "mean": `steps` launches back to back between ONE event pair -- the average launch duration the roofline is
    computed from (what `rocprofv3 --kernel-trace --stats` reports as the kernel's average).  "median" / "min" /
    "max": a second pass with an event between every two launches (SURVEY.md section 8d asks for median and
    minimum); an event is a barrier packet, so consecutive kernels cannot overlap their tail and head there and
    these run 1-3 % above the back-to-back mean for kernels of 0.2-0.4 ms."""
    import torch

    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps):
        fn(i)
    b.record()
    b.synchronize()
    mean = a.elapsed_time(b) / steps
    events = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    events[0].record()
    for i in range(steps):
        fn(i)
        events[i + 1].record()
    events[-1].synchronize()
    per = [events[i].elapsed_time(events[i + 1]) for i in range(steps)]
    return {"mean": mean, "median": statistics.median(per), "min": min(per), "max": max(per), "n": steps}


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
    out["allocator_note"] = ("value = fields from the wide search (placement.configure: 24 candidates, 8 GB spacers) + the stencil's "
                             "placement_hint(); value_default_allocator = plain gt_storage.empty at the allocator's defaults; "
                             "value_allocator_off = GT4PY_AMD_ALLOC_GROUPS=0")
    return out


def hdiff_input(shape, dtype, gen, origin=(2, 2, 0), cls=None):
    """the demo notebook's field (docs/.../demo_horizontal_diffusion.ipynb cell 9) plus noise, so that
    the flux limiter fires on a non-trivial subset: 5 + 8 (2 + cos(pi (x + 1.5 y)) + sin(2 pi (x + 1.5 y))) / 4 + 0.1 U[-1, 1),
    x = i / N, y = j / N; the same on every level."""
    import math

    import torch

    import gt4py_amd.storage as gt_storage

    f = gt_storage.empty(shape, dtype, backend="hip:mi300", aligned_index=origin, memory_class=cls)
    x = torch.arange(shape[0], dtype=torch.float64, device="cuda")[:, None, None] / shape[0]
    y = torch.arange(shape[1], dtype=torch.float64, device="cuda")[None, :, None] / shape[1]
    s = x + 1.5 * y
    base = 5.0 + 8.0 * (2.0 + torch.cos(math.pi * s) + torch.sin(2.0 * math.pi * s)) / 4.0
    noise = 0.1 * (torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 2.0 - 1.0)
    f.tensor.copy_((base + noise).to(f.tensor.dtype))
    return f


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
                 "sweeps (8 of the excess bytes per lattice update, by construction: DESIGN.md section 4b)")
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
# import); the names stay reachable as bench.<name> for the tests and scripts that grew up with them here.
from gt4py_amd.distributed.calibrate import (FailedOnSomeRank, WallBudget, _agree, _slowest_rank_ms, best_of, calibrate_laplacian,  # noqa: E402,F401
                                             calibrate_transports, calibration_line_keys, calibration_seconds, direct_fenced,
                                             direct_mode, direct_step_down, hdiff_calibration_order, ladder_line_keys,
                                             lap_calibration_order, lap_candidate_of, lap_key, measure_candidate, run_calibration)

# how many consecutive epochs of the probe every form of the distributed apply is checked on before it is timed (the last one next
# to an HBM-saturating background: selfcheck.FormCheck.check), and how many the canary of the direct transport runs under load
CHECK_EPOCHS = int(os.environ.get("GT4MI_BENCH_CHECK_EPOCHS", "3"))
CANARY_STRESS_EPOCHS = int(os.environ.get("GT4MI_BENCH_CANARY_EPOCHS", "200"))


def gather_rank_proof(ctx, info) -> dict:
    """What RCCL itself reports on every rank (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), gathered: the line's
    evidence that the communicator really spans N ranks on N devices."""
    mine = (int(info["rank"]), int(info["device"]), int(info["nranks"]))
    if ctx["distributed"]:
        everyone = [None] * ctx["world"]
        ctx["dist"].all_gather_object(everyone, mine)
    else:
        everyone = [mine]
    return {"rccl_nranks": int(info["nranks"]), "rccl_ranks_agree": len({e[2] for e in everyone}) == 1,
            "rank_devices": [[e[0], e[1]] for e in sorted(everyone)]}


def transport_fallback_banner(rank: int, why: str) -> None:
    """A run that silently changed transport would put a Python-driven exchange (~250 us per step) into the scaling
    curve without anyone noticing: say it loudly (and the JSON line carries "transport_fallback": true)."""
    if rank == 0:
        bar = "!" * 100
        print(f"{bar}\nbench.py: NATIVE RCCL TRANSPORT UNAVAILABLE ({why}); FALLING BACK TO torch.distributed P2P DRIVEN FROM "
              f"PYTHON.\nThe numbers of this run are NOT those of the product path (libgt4py_amd's native RCCL plan).\n{bar}",
              file=sys.stderr, flush=True)


def _test_hang(dog, phase: str) -> None:
    """tests/test_gpu_distributed.py: GT4MI_BENCH_TEST_HANG=<phase> makes the process sit in that phase until a deadline."""
    if os.environ.get("GT4MI_BENCH_TEST_HANG") == phase:
        dog.arm(8.0 / dog.scale, f"{phase} (a hang simulated for the tests)")  # 8 s whatever the scale of the real deadlines
        time.sleep(10 ** 6)


def direct_canary(ctx) -> bool:
    """Before THIS process maps another device's memory and lets its kernels store into it: a child process per rank does exactly
    that on a small problem -- `python -m gt4py_amd.distributed --transport direct`, the self-check of the direct transport (no RCCL,
    its own gloo group) -- and all ranks agree on the outcome.  A memory fault or a hang between real devices then ends a child, not
    the run.  Every form runs CHECK_EPOCHS consecutive epochs of the probe and the one-stream forms CANARY_STRESS_EPOCHS more, every
    other one next to an HBM-saturating background, with the ranks launching together (selfcheck.py: every round is sensitive to a receive
    buffer read too early).  DOWN THE LADDER: should the default mode fail on any rank, the children run once more in the fenced
    mode; if that passes the calibration uses the direct transport FENCED (ctx["direct_mode"]), else it stays on RCCL."""
    dog, rank, world = ctx["dog"], ctx["rank"], ctx["world"]
    import shutil
    import socket
    import tempfile

    def children(fenced: bool, attempt: int) -> bool:
        dog.arm(420, f"canary of the direct transport (child processes{', fenced' if fenced else ''})")
        # the children's own rendezvous.  One node: a file in a fresh directory (no port that was free a moment ago, no second
        # listener on the launcher's address).  Ranks on several hosts cannot share a file in /tmp: then a TCP store on rank 0's
        # address, on a port rank 0 found free.
        hosts = [socket.gethostname()]
        if ctx["distributed"]:
            hosts = [None] * world
            ctx["dist"].all_gather_object(hosts, socket.gethostname())
        one_node = len(set(hosts)) == 1
        where, tmpdir = [None], None
        if rank == 0:
            if one_node:
                tmpdir = tempfile.mkdtemp(prefix="gt4mi_canary_")
                where[0] = ("file", os.path.join(tmpdir, "rendezvous"))
            else:
                with socket.socket() as sock:
                    sock.bind(("", 0))
                    where[0] = ("tcp", os.environ.get("MASTER_ADDR", hosts[0]), sock.getsockname()[1])
        if ctx["distributed"]:
            ctx["dist"].broadcast_object_list(where, src=0)
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(ctx["local_rank"]),
                   PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("GT4MI_RENDEZVOUS_FILE", None)
        if where[0][0] == "file":
            env["GT4MI_RENDEZVOUS_FILE"] = str(where[0][1])
        else:
            env["MASTER_ADDR"], env["MASTER_PORT"] = str(where[0][1]), str(where[0][2])
        # (the launcher's own variables would send the child to the launcher's store -- TORCHELASTIC_USE_AGENT_STORE)
        for key in [k for k in env if k.startswith("TORCHELASTIC_")] + ["GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE",
                                                                        "GT4MI_BENCH_TEST_HANG"]:
            env.pop(key, None)
        ok = 0
        try:
            fails = os.environ.get("GT4MI_BENCH_TEST_CANARY_FAILS", "")  # (tests: "1" = both modes fail, "unfenced" = only the default mode)
            if fails == "1" or (fails == "unfenced" and not fenced):
                raise RuntimeError("simulated for the tests")
            cmd = [sys.executable, "-m", "gt4py_amd.distributed", "--transport", "direct", "--domain", "256", "192", "8",
                   "--epochs", str(CHECK_EPOCHS), "--stress-epochs", str(CANARY_STRESS_EPOCHS)] + (["--fenced"] if fenced else [])
            proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
            ok = int(proc.returncode == 0)
            if not ok:
                print(f"rank {rank}: the canary of the direct transport{' (fenced)' if fenced else ''} ended with status "
                      f"{proc.returncode}: {(proc.stdout + proc.stderr)[-900:]}", file=sys.stderr)
        except Exception as ex:  # (a timeout: the child is killed)
            print(f"rank {rank}: the canary of the direct transport{' (fenced)' if fenced else ''} failed ({ex!r})", file=sys.stderr)
        good = bool(_agree(ctx, ok))  # (every child has ended on every rank: the directory is no longer needed)
        if tmpdir is not None:
            shutil.rmtree(tmpdir, ignore_errors=True)
        return good

    if children(fenced=direct_fenced(ctx), attempt=0):
        return True
    if not direct_fenced(ctx):
        direct_step_down(ctx, "canary", "the self-check of the direct transport failed in child processes on some rank")
        ctx.pop("direct_retry", None)  # (nothing has been calibrated on the transport yet)
        if children(fenced=True, attempt=1):
            return True
    direct_step_down(ctx, "canary (fenced)", "the self-check of the fenced direct transport failed in child processes on some rank")
    if rank == 0:
        print("bench.py: the direct halo transport did not pass its canary on every rank, with or without fences: the calibration "
              "stays on RCCL", file=sys.stderr)
    return False


def _native_comm(ctx, selfloop: bool):
    """(NativeComm or None, proof) -- creating the communicator is collective; should it fail on any rank, every rank
    falls back to the torch transport together.  proof = what RCCL itself reports (ncclCommCount) + the rank -> device map."""
    import torch

    from gt4py_amd.distributed import NativeComm

    dog, rank = ctx["dog"], ctx["rank"]
    dog.arm(180, "native RCCL communicator (ncclCommInitRank)")
    ok, comm, info = 1, None, None
    try:
        if ctx.get("one_device"):  # (no RCCL between ranks that share a device: the direct transport only)
            comm = NativeComm(rank=ctx["rank"], world_size=ctx["world"], rccl=False)
        else:
            comm = NativeComm() if not selfloop else NativeComm(rank=0, world_size=1)
        info = comm.info()
    except Exception as ex:
        ok = 0
        print(f"rank {rank}: native RCCL communicator failed ({ex!r})", file=sys.stderr)
    if not _agree(ctx, ok):
        transport_fallback_banner(rank, "ncclCommInitRank failed on at least one rank")
        return None, None
    return comm, gather_rank_proof(ctx, info)


def _setup_distributed_laplacian(args, ctx):
    """Returns (step, kernel_step, local_domain, config, extras) for the decomposed headline workload.

    Headline (like-for-like with N = 1 and with the north star): INDEPENDENT applies on fixed inputs, ghost depth 1, the
    input's ghost cells exchanged on EVERY apply next to the interior kernel (gt4mi_dist_lap5_f64: pack, interior ||
    send/recv/unpack, one ring kernel).  The communication-avoiding time steppers are a different workload (u <- lap(u),
    one exchange per H steps) and are reported beside it (extras["timestep"] -> line["extra"])."""
    import numpy as np
    import torch

    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.distributed import (Decomposition, HaloExchanger, NativeHaloExchanger, choose_process_grid, overlapped_apply,
                                       process_grid_candidates)

    world, rank, local_rank, distributed, dog = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["distributed"], ctx["dog"]
    selfloop = args.dist_selfloop and world == 1
    total = (GRID[0], GRID[1] // max(args.selfloop_ranks, 1), GRID[2]) if selfloop else GRID
    selfloop_grid = None
    if selfloop and args.selfloop_grid:  # the share of ONE rank of a PI x PJ grid, every neighbour the rank itself
        pi, pj = (int(v) for v in args.selfloop_grid.split("x"))
        total, selfloop_grid = (GRID[0] // pi, GRID[1] // pj, GRID[2]), (pi, pj)
    lap = gtscript.stencil(backend="hip:mi300", definition=_lap_definition(), dtypes={"T": np.float64}, device_sync=False)
    transport = os.environ.get("GT4MI_BENCH_COMM", "native")
    mode = os.environ.get("GT4MI_BENCH_MODE", "apply")
    fallback = False
    comm, proof = None, None
    if transport == "native":
        comm, proof = _native_comm(ctx, selfloop)
        if comm is None:
            transport, fallback = "torch", True

    def grid_of(name):
        pi, pj = name.split("x")
        return int(pi), int(pj)

    periodic = (False, True) if selfloop else (False, False)
    if selfloop_grid is not None:
        periodic = (selfloop_grid[0] > 1, selfloop_grid[1] > 1)
    grid = (1, 1) if selfloop else choose_process_grid(world, total)
    pinned_grid = "GT4MI_BENCH_GRID" in os.environ
    if pinned_grid:
        grid = grid_of(os.environ["GT4MI_BENCH_GRID"])
    single_phase = os.environ.get("GT4MI_BENCH_SINGLE_PHASE", "0") == "1"
    schedule, wg_per_cu = os.environ.get("GT4MI_BENCH_SCHEDULE", "join"), int(os.environ.get("GT4MI_BENCH_WG_PER_CU", "0"))
    # how the faces travel: RCCL send/recv, or peer stores from the pack kernel (csrc/direct.hip.h); both are calibrated
    transports = tuple(os.environ.get("GT4MI_BENCH_TRANSPORTS", "rccl,direct").split(","))
    halo_transport = transports[0]
    calibration = None

    def apply_candidate(cand_grid, cand_single, cand_schedule="join", cand_wg=0, cand_transport="rccl"):
        """(step(i), keepalive) of the headline form on one process grid / message table / schedule / throttle / transport."""
        cdec = Decomposition(total, cand_grid, rank, halo=1, periodic=periodic)
        cpairs = _device_fields(cdec.local_shape, n_pairs=2, seed=1337 + rank, origin=cdec.origin)
        # (direct_timeout_ms: a broken direct transport costs the calibration 2 s per wait, then its forms are dropped)
        cex = [NativeHaloExchanger(cdec, np.float64, comm, single_phase=cand_single).tune(cand_schedule, cand_wg, direct_timeout_ms=DIRECT_TIMEOUT_MS)
               for _ in cpairs]
        if cand_transport == "direct":  # peer stores from the pack kernel instead of RCCL send/recv (collective; raises on EVERY rank
            for ex in cex:              # when it is not available on some rank: measure_candidate then drops the candidate)
                ex.use_direct_transport().tune(direct_fenced=direct_fenced(ctx))  # (the ladder's current rung: calibrate.py)
        bound = [ex.make_dist_lap5(inp, out, cdec.origin, cdec.origin) for ex, (inp, out) in zip(cex, cpairs)]
        state = {"i": 0}

        def call():
            bound[state["i"] % len(bound)]()
            state["i"] += 1

        chk = form_check(cdec)
        probe_apply = cex[0].make_dist_lap5(chk.probe, chk.out, cdec.origin, cdec.origin)

        def probe_run():
            probe_apply()
            cex[0].end()

        def check():
            # CHECK_EPOCHS consecutive epochs (each round's correct values differ from the last round's in every cell), the last
            # one next to an HBM-saturating background
            good, found = chk.check(probe_run, CHECK_EPOCHS, 1)
            if cand_transport == "direct" and cex[0].direct_status()["timed_out"]:
                good, found = False, "a wait of the direct transport ran out of time; " + found
            return good, found

        return call, (cdec, cpairs, cex, bound, check)

    checks = {}

    def form_check(cdec):
        """distributed.FormCheck of one process grid: fields whose correct outcome every rank knows exactly."""
        key = cdec.grid
        if key not in checks:
            import gt4py_amd.storage as gt_storage
            from gt4py_amd.distributed import FormCheck

            cfrozen = lap.freeze(origin={"inp": cdec.origin, "out": cdec.origin}, domain=cdec.local_domain)
            checks[key] = FormCheck(cdec, (lambda: gt_storage.zeros(cdec.local_shape, np.float64, backend="hip:mi300",
                                                                    aligned_index=cdec.origin)),
                                    (lambda a, b: cfrozen(inp=a, out=b)))
        return checks[key]

    if transport == "native" and callable(ctx.get("provisional")):
        # Before anything that has never run between two devices is tried (the fused applies, their schedules, the other
        # process grids): the plain sequence on the default grid -- exchange on the caller's stream (pack, one RCCL group of
        # sends and receives, unpack), then ONE launch over the whole local domain -- measured by the contract and kept as
        # the line to print should a later phase hang (Watchdog.safe).
        pdec = Decomposition(total, grid, rank, halo=1, periodic=periodic)
        ok = 1
        try:
            ppairs = _device_fields(pdec.local_shape, n_pairs=2, seed=1337 + rank, origin=pdec.origin)
            pex = [NativeHaloExchanger(pdec, np.float64, comm).tune(direct_timeout_ms=DIRECT_TIMEOUT_MS) for _ in ppairs]
            if "rccl" not in transports:  # (GT4MI_BENCH_TRANSPORTS=direct, GT4MI_BENCH_ONE_DEVICE: no send/recv at all)
                for ex in pex:
                    ex.use_direct_transport().tune(direct_fenced=direct_fenced(ctx))
            pfrozen = lap.freeze(origin={"inp": pdec.origin, "out": pdec.origin}, domain=pdec.local_domain)

            def pstep(i):
                inp, out = ppairs[i % len(ppairs)]
                pex[i % len(ppairs)].exchange(inp)
                pfrozen(inp=inp, out=out)

            def pkernel(i):
                inp, out = ppairs[i % len(ppairs)]
                pfrozen(inp=inp, out=out)

            pconfig = {"workload": "fp64 5-point Laplacian 512x512x512 split over the ranks (strong scaling); independent applies "
                                   "on fixed inputs (two rotating pairs), ghost depth 1, the input's ghost cells exchanged before "
                                   "EVERY apply (sequential form: exchange, then one launch over the whole local domain)",
                       "grid": list(total), "decomposition": f"{grid[0]}x{grid[1]}", "local_domain": list(pdec.local_domain),
                       "halo_depth": 1, "halo_bytes_per_rank_per_exchange": pex[0].bytes_per_exchange,
                       "message_table": "two-phase (I faces, then J faces with the fresh I-halo columns)", "transport": "native",
                       "mode": "apply", "selfloop": bool(selfloop), "exchange_overlapped_with_interior": False}
            chk = form_check(pdec)  # first of all: is what it computes right?  (fields whose correct outcome is known exactly)

            def provisional_run():
                pex[0].exchange(chk.probe)
                pfrozen(inp=chk.probe, out=chk.out)

            good, found = chk.check(provisional_run, CHECK_EPOCHS, 1)
            ctx["forms_checked"] = ctx.get("forms_checked", 0) + 1
            if not good:
                raise RuntimeError("wrong results: " + found)
            ctx["provisional"](pstep, pkernel, pdec.local_domain, float(np.prod(pdec.global_domain)), pconfig, proof)
            for ex in pex:
                ex.close()
            del ppairs, pex, pfrozen
            torch.cuda.empty_cache()
        except Exception as ex:
            ok = 0
            print(f"rank {rank}: the sequential form of the native halo exchange failed ({ex!r})", file=sys.stderr)
        if not _agree(ctx, ok):
            transport, comm, fallback = "torch", None, True
            transport_fallback_banner(rank, "the native halo exchange failed in its plainest form (exchange, then one launch)")
    canary = None
    stats = {"run": 0, "skipped_for_time": 0, "failed": []}
    _test_hang(dog, "calibration")
    if transport == "native" and mode == "apply" and not ("GT4MI_BENCH_SINGLE_PHASE" in os.environ and pinned_grid):
        # Measured before the warm-up, all ranks agreeing on the slowest rank's time: the process grid (xGMI is point-to-point:
        # what costs is the LARGEST message of a round, 1x8 sends 2.1 MB faces, 4x2 and 2x4 at most 1.05 MB), the message table
        # (two rounds with 4 neighbours, or one round with faces + corners to 8), schedule, throttle, transport -- BEST FIRST
        # under a wall-clock budget (lap_calibration_order): the RCCL forms the north star names come first, the direct
        # transport after its canary; what the budget does not reach is counted (calibration_candidates_skipped_for_time).
        rccl_seconds = calibration_seconds("GT4MI_BENCH_CALIBRATION_SECONDS", 90)
        direct_seconds = calibration_seconds("GT4MI_BENCH_DIRECT_CALIBRATION_SECONDS", 60)
        dog.arm(rccl_seconds + direct_seconds + 600, "calibration of process grid x message table x schedule x transport")
        ok, table = 1, {}
        try:
            grids = [grid] if (selfloop or pinned_grid) else process_grid_candidates(world, total, 1)
            phases = (single_phase,) if "GT4MI_BENCH_SINGLE_PHASE" in os.environ else (False, True)

            def measure(cand):
                def make(cand=cand):
                    call, keep = apply_candidate(*cand)
                    return call, (lambda: [ex.close(collective=False) for ex in keep[2]]), keep[4]

                return measure_candidate(ctx, make, 24)

            def canary_of_the_direct_transport():
                if not distributed or ctx.get("one_device"):  # (the self-loop: every peer is this process itself; the rehearsal on one device IS a canary)
                    return None
                good = direct_canary(ctx)  # before THIS process maps another device's memory: a child process per rank tries it
                dog.arm(2 * direct_seconds + 600, "calibration of the direct transport")  # (the stage may run twice: unfenced, fenced)
                return good

            canary, transports = calibrate_laplacian(ctx, grid, grids, phases, transports, measure, canary_of_the_direct_transport,
                                                     rccl_seconds, direct_seconds, table, stats)
            torch.cuda.empty_cache()
        except Exception as ex:
            ok = 0
            print(f"rank {rank}: native RCCL halo exchange failed during calibration ({ex!r})", file=sys.stderr)
        if not _agree(ctx, ok and bool(table)):
            transport, comm, fallback = "torch", None, True
            transport_fallback_banner(rank, "the native halo exchange failed during calibration (no candidate ran on every rank)")
        else:
            calibration = table
            grid, single_phase, schedule, wg_per_cu, halo_transport = lap_candidate_of(min(table, key=table.get))
    elif transport == "native" and "direct" in transports and distributed and not ctx.get("one_device"):
        canary = direct_canary(ctx)
        if not canary:
            transports = tuple(t for t in transports if t != "direct") or ("rccl",)
            halo_transport = transports[0]
    dog.arm(180, "set-up of the decomposed fields and exchangers")
    dec = Decomposition(total, grid, rank, halo=1, periodic=periodic)
    origin = {"inp": dec.origin, "out": dec.origin}
    local_domain = dec.local_domain
    frozen = lap.freeze(origin=origin, domain=local_domain)
    stepper_state = {}
    headline_verdict = None
    if transport == "native" and mode == "apply":
        call, keep = apply_candidate(grid, single_phase, schedule, wg_per_cu, halo_transport)
        pairs, exchangers = keep[1], keep[2]
        headline_verdict = keep[4]()  # the form that is about to be timed, checked once more
        ctx["forms_checked"] = ctx.get("forms_checked", 0) + 1

        def step(i):
            call()
    elif transport == "native":  # GT4MI_BENCH_MODE=timestep: the communication-avoiding stepper as the timed workload
        halo = max(1, int(os.environ.get("GT4MI_BENCH_HALO", "2")))
        dec = Decomposition(total, grid, rank, halo=halo, periodic=periodic)
        origin = {"inp": dec.origin, "out": dec.origin}
        frozen = lap.freeze(origin=origin, domain=dec.local_domain)
        pairs = _device_fields(dec.local_shape, n_pairs=2, seed=1337 + rank, origin=dec.origin)
        exchangers = [NativeHaloExchanger(dec, np.float64, comm, single_phase=single_phase).tune(schedule, wg_per_cu)]
        a, b = pairs[0][0], pairs[1][0]
        a.tensor.mul_(1e-150)  # ~8x growth per step stays finite for 600 steps
        b.tensor.copy_(a.tensor)
        cycle = exchangers[0].make_time_skewed_lap5(a, b, dec.origin)
        stepper_state["halo"] = halo

        def step(i):
            if i % halo == 0:
                cycle()
        keep = (cycle,)
    else:  # torch.distributed point-to-point ops driven from Python
        pairs = _device_fields(dec.local_shape, n_pairs=2, seed=1337 + rank, origin=dec.origin)
        exchangers = [HaloExchanger(dec, torch.float64, torch.device("cuda", local_rank)) for _ in pairs]
        keep = ()

        def step(i):
            inp, out = pairs[i % len(pairs)]
            overlapped_apply(lap, dec, origin, {"inp": inp, "out": out}, {"inp": exchangers[i % len(pairs)]})

        chk = form_check(dec)
        headline_verdict = chk.check((lambda: overlapped_apply(lap, dec, origin, {"inp": chk.probe, "out": chk.out}, {"inp": exchangers[0]})),
                                     CHECK_EPOCHS, 1)
        ctx["forms_checked"] = ctx.get("forms_checked", 0) + 1

    verified = None
    if headline_verdict is not None:
        everywhere = bool(_agree(ctx, 1 if headline_verdict[0] else 0))
        if not everywhere:
            print(f"rank {rank}: THE TIMED FORM GIVES WRONG RESULTS on at least one rank (here: {headline_verdict[1]})", file=sys.stderr)
        verified = {"headline_form_correct_on_every_rank": everywhere, "forms_checked": ctx.get("forms_checked", 0),
                    "forms_rejected": ctx.get("forms_rejected", 0), "ghost_cells_checked_on_rank_0": form_check(dec).ghost_cells_to_fill,
                    "epochs_per_form": CHECK_EPOCHS, "rounds_checked_on_rank_0": sum(c.rounds_checked for c in checks.values()),
                    "how": "every form is run on a field whose own points hold an exact function of the GLOBAL coordinates + 65536 x "
                           "EPOCH and whose ghost cells hold a sentinel, for epochs_per_form consecutive epochs (so that what the "
                           "previous round left in any receive buffer is wrong in every cell), the last one next to an HBM-saturating "
                           "background: afterwards every cell must equal that function (or still the sentinel beyond a physical "
                           "boundary) and the result must equal the local kernel applied to the exactly known input, bit for bit, on "
                           "every rank (gt4py_amd/distributed/selfcheck.py); wrong forms are dropped from the calibration, a wrong "
                           "form of the direct transport moves every rank down the ladder direct -> direct-fenced -> rccl"}

    def kernel_step(i):  # the local kernel alone, for the per-GPU roofline figure
        inp, out = pairs[i % len(pairs)]
        frozen(inp=inp, out=out)

    def pipelined_applies():
        """The same independent applies WITHOUT the per-apply join (GT4MI_PLAN_DEFER_JOIN): each (inp, out) pair has its own
        plan and side stream, so the interior of apply i + 1 runs next to the exchange and ring of apply i; every apply
        still exchanges its own input's ghost cells.  ms per apply, slowest rank."""
        table = {}
        for cand_wg, cand_transport in [(w, t) for w in (0, 4, 2) for t in transports]:
            if cand_transport == "direct" and ctx.get("direct_dropped"):
                continue
            if not ctx["informational_budget"].more():  # (collective: every rank stops at the same candidate)
                break

            def make(cand_wg=cand_wg, cand_transport=cand_transport):
                call, keep = apply_candidate(grid, single_phase, "chain", cand_wg, cand_transport)
                for ex in keep[2]:
                    ex.tune(defer_join=True)
                return call, (lambda: [(ex.end(), ex.close(collective=False)) for ex in keep[2]])

            ms = measure_candidate(ctx, make, 48)
            if ms is not None:
                table[f"chain_wg{cand_wg}_{cand_transport}"] = ms
        torch.cuda.empty_cache()
        return table

    def timestep_extras():
        """The communication-avoiding time steppers (u <- lap(u), ghost regions H deep, one exchange per H steps) on the
        chosen grid: ms per STEP of every schedule x depth, slowest rank; collective, so every rank runs it."""
        if transport != "native" or os.environ.get("GT4MI_BENCH_TIMESTEP", "1") == "0":
            return None
        ctx["informational_budget"] = WallBudget(ctx, calibration_seconds("GT4MI_BENCH_INFORMATIONAL_SECONDS", 60))
        pipelined = pipelined_applies() if mode == "apply" else None
        table = {}
        for cand_halo in (1, 2, 3, 4):
            if cand_halo > 1 and ((grid[1] > 1 or selfloop) and total[1] // grid[1] < 2 * (2 * cand_halo - 1)
                                  or grid[0] > 1 and total[0] // grid[0] < 2 * (2 * cand_halo - 1)):
                continue
            cdec = Decomposition(total, grid, rank, halo=cand_halo, periodic=periodic)
            for stepper, cand_transport in [(st, t) for st in ("skewed_join", "skewed_chain", "skewed_chain_wg4", "wide_overlap",
                                                               "wide_sequential") for t in transports]:
                if not stepper.startswith("skewed") and cand_halo == 3:
                    continue
                if cand_transport == "direct" and ctx.get("direct_dropped"):
                    continue
                if not ctx["informational_budget"].more():
                    continue
                per_call = cand_halo if stepper.startswith("skewed") else 1

                def make(stepper=stepper, cdec=cdec, cand_transport=cand_transport):
                    cpairs = _device_fields(cdec.local_shape, n_pairs=2, seed=7 + rank, origin=cdec.origin)
                    ca, cb = cpairs[0][0], cpairs[1][0]
                    ca.tensor.mul_(1e-150)
                    cb.tensor.copy_(ca.tensor)
                    cex = NativeHaloExchanger(cdec, np.float64, comm, single_phase=single_phase)
                    if cand_transport == "direct":
                        cex.use_direct_transport().tune(direct_fenced=direct_fenced(ctx))
                    if stepper.startswith("skewed"):
                        cex.tune("chain" if "chain" in stepper else "join", 4 if stepper.endswith("wg4") else 0)
                        fn = cex.make_time_skewed_lap5(ca, cb, cdec.origin)
                    else:
                        fn = cex.make_time_stepper_lap5(ca, cb, cdec.origin, overlap=stepper == "wide_overlap")
                    return fn, (lambda: cex.close(collective=False))

                calls = max(24 // per_call, 6) if per_call > 1 else 24
                ms = measure_candidate(ctx, make, calls, warm=2 * (cand_halo if per_call == 1 else 1))
                if ms is not None:
                    table[f"{stepper}_halo{cand_halo}_{cand_transport}"] = round(ms / per_call, 5)
        torch.cuda.empty_cache()
        lups = float(np.prod(dec.global_domain))
        out = {}
        if pipelined:
            pbest = min(pipelined, key=pipelined.get)
            out = {"pipelined_apply_glups": round(lups / pipelined[pbest] / 1e6, 2), "pipelined_apply_best": pbest,
                   "pipelined_apply_ms": pipelined,
                   "pipelined_apply_workload": "the applies of `value` without the join after each one: the applies are independent "
                                               "(two rotating pairs, a plan and side stream each), so apply i + 1's interior kernel "
                                               "runs next to apply i's exchange and ring; every apply still exchanges its own ghost cells"}
        if not table:
            return out or None
        best = min(table, key=table.get)
        return {**out, "timestep_glups": round(lups / table[best] / 1e6, 2), "timestep_best": best, "timestep_ms_per_step": table,
                "timestep_workload": "time stepping u <- lap(u) on the same decomposed grid, ghost regions H deep, ONE exchange "
                                     "per H steps (skewed: boundary bands first, the faces travel next to H interior kernels; "
                                     "wide: grown launches, exchange next to one interior kernel / after a full-domain kernel) "
                                     "-- a different workload from `value`, reported beside it"}

    what = {"apply": "independent applies on fixed inputs (two rotating pairs), ghost depth 1, the input's ghost cells "
                     "exchanged on EVERY apply next to the interior kernel (RCCL send/recv on a side stream)",
            "timestep": "time stepping u <- lap(u), time-skewed schedule, ghost regions %d deep: one RCCL exchange per %d steps"
                        % (stepper_state.get("halo", 1), stepper_state.get("halo", 1))}[mode if transport == "native" else "apply"]
    config = {"workload": "fp64 5-point Laplacian 512x512x512 split over the ranks (strong scaling); " + what,
              "grid": list(total), "decomposition": f"{grid[0]}x{grid[1]}", "local_domain": list(dec.local_domain),
              "halo_depth": stepper_state.get("halo", 1),
              "halo_bytes_per_rank_per_exchange": exchangers[0].bytes_per_exchange,
              "message_table": ("single-phase (faces + corners, up to 8 neighbours)" if single_phase else
                                "two-phase (I faces, then J faces with the fresh I-halo columns)") if transport == "native" else "two-phase",
              "transport": transport, "mode": mode if transport == "native" else "apply", "selfloop": bool(selfloop),
              "exchange_overlapped_with_interior": True,
              "schedule": schedule if transport == "native" else "join", "interior_workgroups_per_cu": wg_per_cu,
              "halo_transport": (halo_transport + (" (peer stores from the pack kernel, flags in the receiver's memory; no send/recv kernel)"
                                                   if halo_transport == "direct" else " (send/recv)")) if transport == "native" else "torch",
              "calibration_ms_per_apply": calibration, "verified": verified, "direct_transport_dropped_at": ctx.get("direct_dropped"),
              "direct_transport_canary": canary, **ladder_line_keys(ctx)}
    extras = {"exchangers": exchangers, "total_lups": float(np.prod(dec.global_domain)), "keep": (pairs, comm, frozen, keep),
              "timestep": timestep_extras, "proof": proof, "transport_fallback": fallback,
              "calibration": calibration_line_keys(calibration, stats, ctx, {"total": total, "halo": 1, "itemsize": 8, "grid": grid})
              if calibration is not None else None}
    return step, kernel_step, dec.local_domain, config, extras


# ---- BASELINE.json configs[4]: horizontal diffusion, 512 x 1024 x 80 per rank, ghost depth 2 --------------
def _setup_hdiff2048(args, ctx):
    import numpy as np
    import torch

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates
    from gt4py_amd.distributed import (Decomposition, HaloExchanger, NativeHaloExchanger, choose_process_grid, overlapped_apply,
                                       sequential_apply)

    world, rank, local_rank, distributed, dog = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["distributed"], ctx["dog"]
    selfloop = args.dist_selfloop and world == 1
    halo = 2
    grid = choose_process_grid(world, HDIFF_GLOBAL, halo)  # 8 ranks -> 4 x 2
    if "GT4MI_BENCH_GRID" in os.environ:
        pi, pj = os.environ["GT4MI_BENCH_GRID"].split("x")
        grid = (int(pi), int(pj))
    total = (HDIFF_SHARE[0] * grid[0], HDIFF_SHARE[1] * grid[1], HDIFF_SHARE[2])  # weak scaling: fixed share per rank
    periodic = (True, True) if selfloop else (False, False)
    dec = Decomposition(total, grid, rank, halo=halo, periodic=periodic)
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64},
                          device_sync=False)
    gen = torch.Generator(device="cuda").manual_seed(4242 + rank)

    def field(lo, hi):
        f = gt_storage.empty(dec.local_shape, np.float64, backend="hip:mi300", aligned_index=dec.origin)
        f.tensor.copy_(torch.rand(dec.local_shape, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    fields = {"in_field": hdiff_input(dec.local_shape, np.float64, gen, dec.origin), "coeff": field(0.025, 0.025),
              "out_field": field(-1.0, 1.0)}
    origin = {k: dec.origin for k in fields}
    frozen = hd.freeze(origin=origin, domain=dec.local_domain)
    decomposed = distributed or selfloop
    transport, comm, proof, exchangers, fallback = "none", None, None, [], False
    timings, choice = None, "single launch"
    headline_verdict, verified, ghost_cells = None, None, 0
    hd_transports, canary = tuple(os.environ.get("GT4MI_BENCH_TRANSPORTS", "rccl,direct").split(",")), None
    stats = {"run": 0, "skipped_for_time": 0, "failed": []}
    if decomposed:
        transport = os.environ.get("GT4MI_BENCH_COMM", "native")
        if transport == "native":
            comm, proof = _native_comm(ctx, selfloop)
            if comm is None:
                transport, fallback = "torch", True
        if transport == "native":
            # one C call per apply (gt4mi_dist_hdiff_f64: pack, interior || exchange, ONE ring kernel) with either message
            # table, and the plain sequence (exchange, then one full-domain launch): measured, slowest rank decides
            flags = type(hd)._gt_binding_.flags
            edge_candidates = tuple(int(v) for v in os.environ.get("GT4MI_BENCH_EDGE_CANDIDATES", "2,16,32").split(","))

            def make_form(name):
                """(callable, exchanger) of one apply form.  One plan (side stream, staging buffers) per form, created when
                it is measured and closed right after: with dozens of plans alive the runtime maps some side streams onto
                the caller's hardware queue and those forms run serialised (0.27 ms instead of 0.21, seen with 72 plans)."""
                parts = name.split("_")
                single = parts[1] == "single"
                chk = form_check()
                if parts[0] == "sequential":
                    ex = NativeHaloExchanger(dec, np.float64, comm, single_phase=single).tune(direct_timeout_ms=DIRECT_TIMEOUT_MS)
                    if "rccl" not in hd_transports:  # (GT4MI_BENCH_TRANSPORTS=direct, GT4MI_BENCH_ONE_DEVICE)
                        ex.use_direct_transport().tune(direct_fenced=direct_fenced(ctx))
                    probe_fields = {"in_field": chk.probe, "out_field": chk.out, "coeff": fields["coeff"]}
                    probe_apply = lambda: sequential_apply(hd, dec, origin, probe_fields, {"in_field": ex})  # noqa: E731
                    fn = lambda: sequential_apply(hd, dec, origin, fields, {"in_field": ex})  # noqa: E731
                else:
                    ex = NativeHaloExchanger(dec, np.float64, comm, single_phase=single).tune(parts[3], int(parts[4][2:]),
                                                                                              edge_columns=int(parts[5][4:]),
                                                                                              direct_timeout_ms=DIRECT_TIMEOUT_MS)
                    if parts[6:] == ["direct"]:  # peer stores from the pack kernel instead of RCCL send/recv (collective; raises on
                        # EVERY rank when some rank cannot: measure_candidate then drops the form); the ladder's current rung
                        ex.use_direct_transport().tune(direct_fenced=direct_fenced(ctx))
                    probe_apply = ex.make_dist_hdiff(chk.probe, chk.out, fields["coeff"], dec.origin, flags)
                    fn = ex.make_dist_hdiff(fields["in_field"], fields["out_field"], fields["coeff"], dec.origin, flags)

                def probe_run():
                    probe_apply()
                    ex.end()

                def check():
                    good, found = chk.check(probe_run, CHECK_EPOCHS, 1)  # (consecutive epochs, the last one under HBM load)
                    if parts[6:] == ["direct"] and ex.direct_status()["timed_out"]:
                        good, found = False, "a wait of the direct transport ran out of time; " + found
                    return good, found

                return fn, ex, check

            checks = []

            def form_check():
                """distributed.FormCheck on this rank's share: fields whose correct outcome every rank knows exactly."""
                if not checks:
                    from gt4py_amd.distributed import FormCheck

                    checks.append(FormCheck(dec, (lambda: gt_storage.zeros(dec.local_shape, np.float64, backend="hip:mi300",
                                                                           aligned_index=dec.origin)),
                                            (lambda a, b: frozen(in_field=a, out_field=b, coeff=fields["coeff"]))))
                return checks[0]

            if callable(ctx.get("provisional")):
                # the plainest form first, measured by the contract and kept as the line to print should a later phase hang
                # (see _setup_distributed_laplacian)
                ok = 1
                try:
                    pfn, pex, pcheck = make_form("sequential_two_phase")
                    good, found = pcheck()  # first of all: is what it computes right?
                    ctx["forms_checked"] = ctx.get("forms_checked", 0) + 1
                    if not good:
                        raise RuntimeError("wrong results: " + found)
                    pconfig = {"workload": "BASELINE.json configs[4]: fp64 horizontal diffusion (lap-of-lap + flux limiter), "
                                           f"{HDIFF_SHARE[0]}x{HDIFF_SHARE[1]}x{HDIFF_SHARE[2]} per rank (weak scaling; 8 ranks = "
                                           "2048x2048x80 on the 4x2 grid), ghost depth 2, in_field's ghost cells exchanged before "
                                           "every apply (sequential form: exchange, then one launch over the whole local domain)",
                               "grid": list(total), "decomposition": f"{grid[0]}x{grid[1]}", "local_domain": list(dec.local_domain),
                               "halo_depth": halo, "halo_bytes_per_rank_per_exchange": pex.bytes_per_exchange,
                               "transport": "native", "selfloop": bool(selfloop), "apply_form": "sequential_two_phase"}
                    ctx["provisional"]((lambda i: pfn()), (lambda i: frozen(**fields)), dec.local_domain, float(np.prod(total)),
                                       pconfig, proof)
                    pex.close()
                except Exception as exn:
                    ok = 0
                    print(f"rank {rank}: the sequential form of the native halo exchange failed ({exn!r})", file=sys.stderr)
                if not _agree(ctx, ok):
                    transport, comm, fallback = "torch", None, True
                    transport_fallback_banner(rank, "the native halo exchange failed in its plainest form (exchange, then one launch)")
            _test_hang(dog, "calibration")
            if transport == "native":  # (still: the plainest form ran on every rank)
                # (the "swap" schedules exist for this step too and are 4-6 % slower than "chain" on the self-loop: here the
                # interior kernel, not the chain, is the critical path -- GT4MI_BENCH_HDIFF_SCHEDULES adds them)
                schedules = tuple(os.environ.get("GT4MI_BENCH_HDIFF_SCHEDULES", "join,chain,inline").split(","))
                pinned = os.environ.get("GT4MI_BENCH_FORM")
                rccl_seconds = calibration_seconds("GT4MI_BENCH_CALIBRATION_SECONDS", 90)
                direct_seconds = calibration_seconds("GT4MI_BENCH_DIRECT_CALIBRATION_SECONDS", 60)
                dog.arm(rccl_seconds + direct_seconds + 600, "calibration of the apply forms")
                ok, timings = 1, {}
                try:
                    first, refine, direct_stage = hdiff_calibration_order(schedules, edge_candidates, hd_transports)

                    def measure(name):
                        def make(name=name):
                            fn, ex, check = make_form(name)
                            return fn, (lambda: ex.close(collective=False)), check

                        return measure_candidate(ctx, make, 16)

                    def wanted(names):
                        return [n for n in names if pinned is None or pinned == n]

                    def best_rccl():
                        mine = {k: v for k, v in timings.items() if not k.endswith("_direct")}
                        return min(mine, key=mine.get) if mine else None

                    def canary_of_the_direct_transport():
                        if not distributed or ctx.get("one_device"):
                            return None
                        good = direct_canary(ctx)  # (see _setup_distributed_laplacian)
                        dog.arm(2 * direct_seconds + 600, "calibration of the direct transport")
                        return good

                    canary, hd_transports = calibrate_transports(ctx, first, refine, direct_stage, str, (lambda name: name.endswith("_direct")),
                                                                 best_rccl, measure, canary_of_the_direct_transport, rccl_seconds,
                                                                 direct_seconds, timings, stats, wanted, hd_transports)
                except Exception as exn:
                    ok = 0
                    print(f"rank {rank}: native RCCL halo exchange failed during calibration ({exn!r})", file=sys.stderr)
                if not _agree(ctx, ok and bool(timings)):
                    transport, comm, fallback = "torch", None, True
                    transport_fallback_banner(rank, "the native halo exchange failed during calibration")
                else:
                    choice = min(timings, key=timings.get)
                    chosen, ex, check = make_form(choice)
                    headline_verdict = check()  # the form that is about to be timed, checked once more
                    ctx["forms_checked"] = ctx.get("forms_checked", 0) + 1
                    exchangers = [ex]  # the one the line describes

                    def step(i):
                        chosen()
        if transport != "native":
            ex = HaloExchanger(dec, torch.float64, torch.device("cuda", local_rank))
            exchangers, choice = [ex], "overlapped (torch transport)"

            def step(i):
                overlapped_apply(hd, dec, origin, fields, {"in_field": ex})

            from gt4py_amd.distributed import FormCheck

            chk = FormCheck(dec, (lambda: gt_storage.zeros(dec.local_shape, np.float64, backend="hip:mi300", aligned_index=dec.origin)),
                            (lambda a, b: frozen(in_field=a, out_field=b, coeff=fields["coeff"])))
            headline_verdict = chk.check((lambda: overlapped_apply(hd, dec, origin, {"in_field": chk.probe, "out_field": chk.out,
                                                                                       "coeff": fields["coeff"]}, {"in_field": ex})),
                                         CHECK_EPOCHS, 1)
            ctx["forms_checked"] = ctx.get("forms_checked", 0) + 1
            ghost_cells = chk.ghost_cells_to_fill
            del chk
        elif checks:
            ghost_cells = checks[0].ghost_cells_to_fill
        if headline_verdict is not None:
            everywhere = bool(_agree(ctx, 1 if headline_verdict[0] else 0))
            if not everywhere:
                print(f"rank {rank}: THE TIMED FORM GIVES WRONG RESULTS on at least one rank (here: {headline_verdict[1]})", file=sys.stderr)
            verified = {"headline_form_correct_on_every_rank": everywhere, "forms_checked": ctx.get("forms_checked", 0),
                        "forms_rejected": ctx.get("forms_rejected", 0), "ghost_cells_checked_on_rank_0": ghost_cells,
                        "epochs_per_form": CHECK_EPOCHS,
                        "how": "see gt4py_amd/distributed/selfcheck.py: every form is run on a field that holds an exact function "
                               "of the GLOBAL coordinates + 65536 x EPOCH (ghost cells: a sentinel) for epochs_per_form consecutive "
                               "epochs, the last one under HBM load; every cell and every point of the result must then be the known "
                               "one, bit for bit, on every rank; wrong forms are dropped from the calibration, a wrong form of the "
                               "direct transport moves every rank down the ladder direct -> direct-fenced -> rccl"}
    else:
        def step(i):
            frozen(**fields)

    def kernel_step(i):
        frozen(**fields)

    def pipelined_applies():
        """Back-to-back applies without the join after each one (GT4MI_PLAN_DEFER_JOIN; the bench's applies are independent):
        the interior of apply i + 1 runs next to the exchange and ring of apply i."""
        if transport != "native":
            return None
        table = {}
        flags = type(hd)._gt_binding_.flags
        budget = WallBudget(ctx, calibration_seconds("GT4MI_BENCH_INFORMATIONAL_SECONDS", 60))
        for single in (False, True):
            for cand_wg in (0, 3, 2):
                for cand_edge, cand_transport in [(e, t) for e in (2, 16, 32)
                                                  for t in hd_transports]:
                    if cand_transport == "direct" and ctx.get("direct_dropped"):
                        continue
                    if not budget.more():
                        continue

                    def make(single=single, cand_wg=cand_wg, cand_edge=cand_edge, cand_transport=cand_transport):
                        ex = NativeHaloExchanger(dec, np.float64, comm, single_phase=single).tune("chain", cand_wg, defer_join=True,
                                                                                                  edge_columns=cand_edge)
                        if cand_transport == "direct":
                            ex.use_direct_transport().tune(direct_fenced=direct_fenced(ctx))
                        fn = ex.make_dist_hdiff(fields["in_field"], fields["out_field"], fields["coeff"], dec.origin, flags)
                        return fn, (lambda: (ex.end(), ex.close(collective=False)))

                    ms = measure_candidate(ctx, make, 32)
                    if ms is not None:
                        table[f"{'single' if single else 'two'}_phase_chain_wg{cand_wg}_edge{cand_edge}_{cand_transport}"] = ms
        if not table:
            return None
        best = min(table, key=table.get)
        return {"pipelined_apply_glups": round(float(np.prod(total)) / table[best] / 1e6, 2), "pipelined_apply_best": best,
                "pipelined_apply_ms": table,
                "pipelined_apply_workload": "the applies of `value` without the join after each one (they are independent): apply "
                                            "i + 1's interior kernel runs next to apply i's exchange and ring"}

    config = {"workload": "BASELINE.json configs[4]: fp64 horizontal diffusion (lap-of-lap + flux limiter), "
                          f"{HDIFF_SHARE[0]}x{HDIFF_SHARE[1]}x{HDIFF_SHARE[2]} per rank (weak scaling; 8 ranks = 2048x2048x80 on the "
                          "4x2 grid), ghost depth 2, in_field's ghost cells exchanged every apply",
              "grid": list(total), "decomposition": f"{grid[0]}x{grid[1]}", "local_domain": list(dec.local_domain),
              "halo_depth": halo, "halo_bytes_per_rank_per_exchange": exchangers[0].bytes_per_exchange if exchangers else 0,
              "transport": transport, "selfloop": bool(selfloop), "apply_form": choice,
              "calibration_ms_per_apply": timings, "verified": verified, "direct_transport_dropped_at": ctx.get("direct_dropped"),
              "direct_transport_canary": canary, **ladder_line_keys(ctx)}
    extras = {"exchangers": exchangers, "total_lups": float(np.prod(total)), "keep": (fields, comm, frozen),
              "proof": proof, "transport_fallback": fallback,
              "timestep": pipelined_applies if decomposed and os.environ.get("GT4MI_BENCH_TIMESTEP", "1") != "0" else None,
              "calibration": calibration_line_keys(timings, stats, ctx, {"total": total, "halo": halo, "itemsize": 8, "grid": grid})
              if timings else None}
    return step, kernel_step, dec.local_domain, config, extras


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
        GRID = (GRID[0], GRID[1], levels)
        # (configs[4]'s share likewise: eight full-size shares on one device took 17-96 s per candidate in round 4)
        HDIFF_SHARE, HDIFF_GLOBAL = (HDIFF_SHARE[0], HDIFF_SHARE[1], levels), (HDIFF_GLOBAL[0], HDIFF_GLOBAL[1], levels)
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
                            spacer_bytes=int(os.environ.get("GT4MI_BENCH_GROUP_SPACER_GB", "8")) << 30, park_extra=5)
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
        if headline:
            # What a drop-in user gets: the headline kernel on fields from the storage allocator AT ITS
            # DEFAULTS (6 plain candidates per search, no spacers, no parked neighbours, no role hints: plain gt_storage.empty) and
            # with the memory-group placer OFF -- 5 + 20 launches each, outside the contract's timed steps.
            try:
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

            line["memory_groups"] = _placement.report()  # what the allocator's placer did for every big field of this run
        if headline and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = None
                print(f"cpu_baseline failed: {ex!r}", file=sys.stderr)
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
