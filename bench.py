#!/usr/bin/env python3
"""Headline benchmark: GLUPS of the fp64 5-point Laplacian on a 512^3 grid through the full user
path (gt4py_amd.storage -> @gtscript.stencil(backend="hip:mi300") -> FrozenStencil -> C ABI -> HIP).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one apply of the stencil over the whole 512^3 grid on synthetic input that is already
resident in HBM.  With N > 1 the SAME 512^3 grid is split over the ranks along J (strong scaling);
a step is then halo exchange (RCCL send/recv on a side stream) overlapped with the interior kernel,
followed by the boundary-strip kernels.  Rank 0 prints ONE JSON line.

Extra objects in the line (see DESIGN.md "Measurement"):
  roofline      HBM roofline of the dominant kernel (lap5_strip_kernel): algorithmic bytes
                (16 B per lattice update) / mean launch duration measured with HIP events on the
                launch stream, against the 8.0 TB/s nominal peak; `traffic` = HBM bytes per launch
                from rocprofv3 PMC counters when a committed measurement exists, else null.
  cpu_baseline  the oracle's C/OpenMP restatement of gt:cpu_ifirst semantics (kind "port") timed on
                this host's cores on a bounded sample (N == 1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

BYTES_PER_LUP = 16.0  # fp64: one read + one write per lattice update
PEAK_GBS = 8000.0  # MI355X HBM3E nominal (MI355X_MICROARCH.md)
GRID = (512, 512, 512)


def _lap_definition():
    from gt4py_amd.cartesian.backend import hip_templates

    return hip_templates.lap_notebook


def _device_fields(shape, n_pairs, seed, origin=(1, 1, 0)):
    """`n_pairs` (inp, out) pairs in HBM with the hip:mi300 layout; inp ~ U[-1, 1), seeded on device."""
    import gt4py_amd.storage as gt_storage

    pairs = []
    gen = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(n_pairs):
        inp = gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=origin)
        out = gt_storage.zeros(shape, np.float64, backend="hip:mi300", aligned_index=origin)
        inp.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1)
        pairs.append((inp, out))
    return pairs


def _time_launches(fn, steps):
    """Mean duration (ms) of `steps` back-to-back launches, from HIP events on the launch stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for i in range(steps):
        fn(i)
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / steps


def copy_ceiling_gbs(steps: int = 10, nbytes: int = 1 << 30) -> float:
    """Streaming device copy (gt4mi_stream_copy, 16-byte lanes) of 1 GiB, read + write bytes per second:
    the achievable-HBM yardstick SURVEY.md section 8d asks to report from the same run."""
    from gt4py_amd import _lib

    lib = _lib.load()
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    src.fill_(1)

    def call(i):
        _lib.check("gt4mi_stream_copy", lib.gt4mi_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes,
                                                              torch.cuda.current_stream().cuda_stream))

    for i in range(2):
        call(i)
    torch.cuda.synchronize()
    ms = _time_launches(call, steps)
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def other_kernels(steps: int = 10):
    """The other kernels of the north star at their BASELINE.json sizes, through the same call path
    (storage -> stencil -> FrozenStencil), HIP-event timed.  Informational: `value` stays the Laplacian."""
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    gen = torch.Generator(device="cuda").manual_seed(2024)

    def field(shape, dtype, origin, lo=-1.0, hi=1.0):
        f = gt_storage.empty(shape, dtype, backend="hip:mi300", aligned_index=origin)
        f.tensor.copy_(torch.rand(shape, dtype=f.tensor.dtype, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    out = {}

    def run(name, obj, fields, origin, domain, bytes_per_lup, scalars=None, note=None):
        frozen = obj.freeze(origin=origin, domain=domain)
        call = lambda i: frozen(**fields, **(scalars or {}))  # noqa: E731
        for i in range(2):
            call(i)
        torch.cuda.synchronize()
        ms = _time_launches(call, steps)
        lups = float(np.prod(domain))
        gbs = bytes_per_lup * lups / (ms * 1e-3) / 1e9
        out[name] = {"domain": list(domain), "ms": round(ms, 4), "glups": round(lups / ms / 1e6, 1),
                     "algorithmic_bytes_per_lup": bytes_per_lup, "achieved_gbs": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / PEAK_GBS, 4)}
        if note:
            out[name]["note"] = note

    for tag, dt, dom in (("hdiff_limiter_f32_1024x1024x80", np.float32, (1024, 1024, 80)),
                         ("hdiff_limiter_f64_512x1024x80", np.float64, (512, 1024, 80))):
        obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt},
                               device_sync=False)
        shape = (dom[0] + 4, dom[1] + 4, dom[2])
        fields = {"in_field": field(shape, dt, (2, 2, 0), 0.0, 10.0), "coeff": field(shape, dt, (2, 2, 0), 0.0, 0.05),
                  "out_field": field(shape, dt, (2, 2, 0))}
        run(tag, obj, fields, {k: (2, 2, 0) for k in fields}, dom, 3.0 * np.dtype(dt).itemsize)
        del fields
    dom = (1024, 1024, 160)
    obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64},
                           device_sync=False)
    fields = {"inf": field(dom, np.float64, (0, 0, 0)), "diag": field(dom, np.float64, (0, 0, 0), 4.0, 5.0),
              "sup": field(dom, np.float64, (0, 0, 0)), "rhs": field(dom, np.float64, (0, 0, 0), -10.0, 10.0),
              "out": field(dom, np.float64, (0, 0, 0))}
    run("tridiagonal_f64_1024x1024x160", obj, fields, {k: (0, 0, 0) for k in fields}, dom, 56.0,
        note="the backward sweep re-reads the part of sup', rhs' that does not fit on chip (72 of 160 levels stay in "
             "registers + LDS): 64.8 B/LUP moved; inputs are whatever the previous launch left in sup/rhs "
             "(timing only, values are checked in tests/)")
    del fields
    torch.cuda.empty_cache()

    # the generic executor (stencils outside the three kernel families are compiled, not rejected): the reference's
    # vertical advection and the Laplacian again, this time through the code generator
    dom = (1024, 1024, 160)
    obj = gtscript.stencil(backend="hip:mi300", definition=_vertical_advection_dycore, externals={"BET_M": 0.5, "BET_P": 0.5},
                           device_sync=False)
    shape = (dom[0] + 1, dom[1], dom[2] + 1)
    fields = {n: field(shape, np.float64, (0, 0, 0)) for n in ("utens_stage", "u_stage", "wcon", "u_pos", "utens")}
    run("generated_vertical_advection_f64_1024x1024x160", obj, fields, {k: (0, 0, 0) for k in fields}, dom, 48.0,
        scalars={"dtr_stage": 3.0 / 20.0},
        note="one generated column kernel (forward + backward sweep); 5 fields read, 1 written; the forward sweep's "
             "ccol / dcol make a round trip through scratch on top of the 48 algorithmic B/LUP")
    del fields
    torch.cuda.empty_cache()
    dom = (512, 512, 512)
    obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64},
                           device_sync=False, use_kernel_library=False)
    shape = (dom[0] + 2, dom[1] + 2, dom[2])
    fields = {"inp": field(shape, np.float64, (1, 1, 0)), "out": field(shape, np.float64, (1, 1, 0))}
    run("generated_laplacian_f64_512x512x512", obj, fields, {k: (1, 1, 0) for k in fields}, dom, 16.0,
        note="the headline stencil through the code generator instead of the hand-written kernel")
    del fields
    torch.cuda.empty_cache()
    return out


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


def cpu_baseline(seconds_budget: float = 12.0):
    """Time the oracle's C/OpenMP port on the same 512^3 workload for a bounded number of applies.

    The whole grid is used on purpose: a 512x512x64 slab (2 x 135 MB) stays resident in the 512 MB of
    L3 of a dual EPYC 9575F host and reports a cache bandwidth, not the workload's."""
    from oracle import cpu_ifirst

    lib = None
    try:  # rebuild for this host's ISA when a compiler is around; else use the prebuilt library
        path = cpu_ifirst.build(march="native", out=pathlib.Path("/tmp") / f"libcpu_ifirst_{os.getpid()}.so")
        lib = cpu_ifirst.load(path)
    except Exception:
        if cpu_ifirst.available():
            lib = cpu_ifirst.load()
    if lib is None:
        return None
    try:  # the cores this process may actually run on (cgroup / affinity), not the machine total
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    try:  # CFS bandwidth quota of the container, e.g. "1600000 100000" = 16 cores
        quota, period = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    lib.oracle_set_threads(cores)
    dom = GRID
    rng = np.random.default_rng(1337)
    inp = np.asfortranarray(rng.uniform(-1, 1, (dom[0] + 2, dom[1] + 2, dom[2])))
    out = np.asfortranarray(np.zeros_like(inp))
    cpu_ifirst.lap5_f64(inp, out, (1, 1, 0), (1, 1, 0), dom, lib=lib)  # warm-up / page touch
    reps, t0 = 0, time.perf_counter()
    while True:
        cpu_ifirst.lap5_f64(inp, out, (1, 1, 0), (1, 1, 0), dom, lib=lib)
        reps += 1
        if time.perf_counter() - t0 > seconds_budget or reps >= 400:
            break
    dt = time.perf_counter() - t0
    glups = dom[0] * dom[1] * dom[2] * reps / dt / 1e9
    return {
        "value": round(glups, 4),
        "unit": "GLUPS",
        "cores": lib.oracle_max_threads(),
        "kind": "port",
        "sample": f"fp64 5-pt Laplacian on the full {dom[0]}x{dom[1]}x{dom[2]} grid, {reps} applies in "
                  f"{dt:.1f} s, C/OpenMP restatement of gt:cpu_ifirst semantics (oracle/cpu_ifirst.c), I-contiguous",
        "gb_per_s": round(glups * BYTES_PER_LUP, 2),
    }


def _committed_traffic(workload: str):
    f = ROOT / "profiles" / "hbm_traffic.json"
    if f.exists():
        try:
            return json.loads(f.read_text()).get(workload)
        except Exception:
            return None
    return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-kernels", action="store_true",
                    help="skip the informational hdiff / tridiagonal lines (N=1 only)")
    ap.add_argument("--dist-selfloop", action="store_true",
                    help="1-GPU rehearsal of the N>1 step: periodic-in-J domain whose halo messages go to the "
                         "rank itself through RCCL (not the headline metric)")
    ap.add_argument("--selfloop-ranks", type=int, default=1,
                    help="with --dist-selfloop: shrink J to 512/N, the per-rank share of an N-GPU run")
    args = ap.parse_args()

    # Native libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on
    # stdout, so everything else is routed to stderr until the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; running on {n_gpus} GPU(s)", file=sys.stderr)

    from gt4py_amd import _lib
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.distributed import (Decomposition, HaloExchanger, NativeComm, NativeHaloExchanger,
                                       choose_process_grid, overlapped_apply)

    decomposed = distributed or args.dist_selfloop

    lap = gtscript.stencil(backend="hip:mi300", definition=_lap_definition(), dtypes={"T": np.float64},
                           device_sync=False)
    origin = {"inp": (1, 1, 0), "out": (1, 1, 0)}

    if not decomposed:
        shape = (GRID[0] + 2, GRID[1] + 2, GRID[2])
        pairs = _device_fields(shape, n_pairs=2, seed=1337)  # rotate pairs: nothing survives in MALL/L2
        frozen = lap.freeze(origin=origin, domain=GRID)

        def step(i):
            inp, out = pairs[i % len(pairs)]
            frozen(inp=inp, out=out)

        local_domain = GRID
        kernel_step = step
        config = {"workload": "fp64 5-point Laplacian 512x512x512 (examples/lap_cartesian_vs_next.ipynb cell 7), "
                              "origin (1,1,0), hip:mi300 storage layout", "grid": list(GRID), "decomposition": "1x1",
                  "call_path": "FrozenStencil"}
    else:
        selfloop = args.dist_selfloop and world == 1
        grid = (1, 1) if selfloop else choose_process_grid(world, GRID)
        total = (GRID[0], GRID[1] // max(args.selfloop_ranks, 1), GRID[2]) if selfloop else GRID
        transport = os.environ.get("GT4MI_BENCH_COMM", "native")
        mode = os.environ.get("GT4MI_BENCH_MODE", "timestep")
        comm = None
        if transport == "native":
            # RCCL communicator owned by libgt4py_amd.  Creating it is collective; should it fail on any
            # rank, every rank falls back to the torch.distributed transport together.
            ok = 1
            try:
                comm = NativeComm() if not selfloop else NativeComm(rank=0, world_size=1)
            except Exception as ex:
                ok = 0
                print(f"rank {rank}: native RCCL communicator failed ({ex!r})", file=sys.stderr)
            if distributed:
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                transport, comm = "torch", None
                print("falling back to GT4MI_BENCH_COMM=torch", file=sys.stderr)
        # ghost depth = steps served by one exchange (communication-avoiding time stepping) and whether that
        # exchange runs next to the last step's interior kernel or after a full-domain kernel.  Which
        # combination wins depends on how long the links take: in the 1-GPU rehearsal (on-device self-copy)
        # "depth 4, not overlapped" is fastest (58.9 us per 512x64x512 step vs 70.4 for depth 2 overlapped,
        # profiles/r1_dist_selfloop_seq_vs_overlap.log), on slow links overlap and a smaller depth should win.
        # So unless pinned through the environment, a short calibration BEFORE the warm-up picks it, with all
        # ranks agreeing on the slowest rank's timings.
        calibration = None
        overlap = os.environ.get("GT4MI_BENCH_OVERLAP", "1") != "0"
        halo = 1
        if transport == "native" and mode == "timestep":
            pinned = "GT4MI_BENCH_HALO" in os.environ or "GT4MI_BENCH_OVERLAP" in os.environ
            halo = max(1, int(os.environ.get("GT4MI_BENCH_HALO", "2")))
            if not pinned:
                def calibrate():
                    calibration = {}
                    for cand_halo in (1, 2, 4):
                        if (grid[1] > 1 and total[1] // grid[1] < 2 * cand_halo) or (grid[0] > 1 and total[0] // grid[0] < 2 * cand_halo):
                            continue
                        cdec = Decomposition(total, grid, rank, halo=cand_halo,
                                             periodic=(False, True) if selfloop else (False, False))
                        cpairs = _device_fields(cdec.local_shape, n_pairs=2, seed=7 + rank, origin=cdec.origin)
                        for cand_overlap in (True, False):
                            ca, cb = cpairs[0][0], cpairs[1][0]
                            ca.tensor.mul_(1e-150)
                            cex = NativeHaloExchanger(cdec, np.float64, comm)
                            cstep = cex.make_time_stepper_lap5(ca, cb, cdec.origin, overlap=cand_overlap)
                            for _ in range(2 * cand_halo):
                                cstep()
                            torch.cuda.synchronize()
                            if distributed:
                                dist.barrier()
                            t0 = time.perf_counter()
                            for _ in range(24):
                                cstep()
                            torch.cuda.synchronize()
                            dt = torch.tensor([(time.perf_counter() - t0) / 24], dtype=torch.float64, device="cuda")
                            if distributed:
                                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                            calibration[f"halo{cand_halo}_{'overlap' if cand_overlap else 'sequential'}"] = round(float(dt.item()) * 1e3, 5)
                            cex.close()
                        del cpairs
                    torch.cuda.empty_cache()
                    best = min(calibration, key=calibration.get)
                    return calibration, int(best.split("_")[0][4:]), best.endswith("overlap")

                # A transport that creates its communicator but cannot move data must not take the run down:
                # every rank reports whether its calibration went through, and all fall back together.
                ok = 1
                try:
                    calibration, halo, overlap = calibrate()
                except Exception as ex:
                    ok, calibration = 0, None
                    print(f"rank {rank}: native RCCL halo exchange failed during calibration ({ex!r})", file=sys.stderr)
                if distributed:
                    flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    ok = int(flag.item())
                if not ok:
                    transport, comm, halo, overlap = "torch", None, 1, True
                    print("falling back to GT4MI_BENCH_COMM=torch", file=sys.stderr)
        dec = Decomposition(total, grid, rank, halo=halo, periodic=(False, True) if selfloop else (False, False))
        origin = {"inp": dec.origin, "out": dec.origin}
        pairs = _device_fields(dec.local_shape, n_pairs=2, seed=1337 + rank, origin=dec.origin)
        local_domain = dec.local_domain
        frozen = lap.freeze(origin=origin, domain=local_domain)
        if transport == "native":
            # whole step (pack, RCCL send/recv, unpack, interior, strips, 2 streams) = one C call
            exchangers = [NativeHaloExchanger(dec, np.float64, comm) for _ in pairs]
            if mode == "timestep":
                # time stepping u <- lap(u) between two buffers.  Ghost regions are `halo` deep and one
                # exchange serves `halo` steps (the steps in between grow their domain into the ghost
                # region instead of communicating); the exchange of the freshly written field travels
                # next to that step's interior kernel and is joined `halo` steps later.
                # The amplitude starts at 1e-150 so that ~8x growth per step stays finite for 600 steps.
                a, b = pairs[0][0], pairs[1][0]
                a.tensor.mul_(1e-150)
                stepper = exchangers[0].make_time_stepper_lap5(a, b, origin["inp"], overlap=overlap)

                def step(i):
                    stepper()
            else:  # independent applies on fixed inputs: exchange the input, then apply
                steps_bound = [ex.make_dist_lap5(inp, out, origin["inp"], origin["out"]) for ex, (inp, out) in
                               zip(exchangers, pairs)]

                def step(i):
                    steps_bound[i % len(pairs)]()
        else:  # torch.distributed point-to-point ops driven from Python
            exchangers = [HaloExchanger(dec, torch.float64, torch.device("cuda", local_rank)) for _ in pairs]

            def step(i):
                inp, out = pairs[i % len(pairs)]
                overlapped_apply(lap, dec, origin, {"inp": inp, "out": out}, {"inp": exchangers[i % len(pairs)]})

        def kernel_step(i):  # the local kernel alone, for the per-GPU roofline figure
            inp, out = pairs[i % len(pairs)]
            frozen(inp=inp, out=out)

        config = {"workload": "fp64 5-point Laplacian 512x512x512 split over ranks along J (strong scaling); "
                              + ("time stepping u <- lap(u), ghost regions %d deep: one RCCL send/recv exchange per %d "
                                 "steps, overlapped with the interior kernel" % (halo, halo) if mode == "timestep"
                                 and transport == "native" else "independent applies, ghost cells exchanged every step"),
                  "grid": list(GRID), "decomposition": f"{grid[0]}x{grid[1]}", "local_domain": list(local_domain),
                  "halo_depth": halo, "halo_bytes_per_rank_per_exchange": exchangers[0].bytes_per_exchange,
                  "transport": transport, "mode": mode, "selfloop": bool(selfloop),
                  "exchange_overlapped_with_interior": bool(overlap) if mode == "timestep" and transport == "native" else None,
                  "calibration_ms_per_step": calibration}

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: mean launch duration from HIP events on the launch stream
    torch.cuda.synchronize()
    kernel_ms = _time_launches(kernel_step, args.steps)
    local_lups = float(np.prod(local_domain))
    achieved = BYTES_PER_LUP * local_lups / (kernel_ms * 1e-3) / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    total_lups = float(np.prod(dec.global_domain)) if decomposed else float(np.prod(GRID))
    glups = total_lups * args.steps / elapsed / 1e9

    if decomposed and isinstance(exchangers[0], NativeHaloExchanger):
        config["side_stream_concurrent"] = exchangers[0].concurrent
    if rank == 0:
        traffic =_committed_traffic("lap5_f64_512") if not decomposed else None
        line = {
            "metric": "GLUPS (lattice updates/s) fp64 5-pt Laplacian 512^3",
            "value": round(glups, 2),
            "unit": "GLUPS",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": config,
            "pct_hbm_roofline": round(100.0 * glups * BYTES_PER_LUP / (PEAK_GBS * n_gpus), 2),
            "roofline": {
                "bound": "hbm",
                "kernel": "lap5_strip_kernel<double,double,0,2,8,*>",
                "achieved": round(achieved, 1),
                "peak": PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / PEAK_GBS, 4),
                "traffic": traffic,
                "kernel_ms": round(kernel_ms, 5),
                "algorithmic_bytes_per_launch": BYTES_PER_LUP * local_lups,
                # what a plain streaming copy reaches on this device in this run (not the bar, the context)
                "measured_copy_gbs": round(copy_ceiling_gbs(), 1) if not decomposed else None,
            },
            "device": _lib.device_info(),
        }
        if not decomposed and not args.no_other_kernels:
            try:
                line["other_kernels"] = other_kernels()
            except Exception as ex:
                line["other_kernels"] = None
                print(f"other_kernels failed: {ex!r}", file=sys.stderr)
        if not decomposed and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = None
                print(f"cpu_baseline failed: {ex!r}", file=sys.stderr)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)  # anything native code prints while tearing down goes to stderr again
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
