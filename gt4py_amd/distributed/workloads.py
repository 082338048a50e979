"""The decomposed workloads of ``bench.py`` -- the 512^3 Laplacian split over the ranks and BASELINE.json configs[4] (horizontal
diffusion 2048 x 2048 x 80 over 4 x 2) -- as a module of the package: set-up of the local fields and exchangers, the canary of the
direct transport, the calibration of process grid x message table x schedule x transport, the form checks, the step the bench
times.  Moved out of bench.py in round 6 (it is the program that meets 8 devices first: here its parts can be imported and
unit-tested without a subprocess); ``bench.py`` re-exports every name.  Imports torch lazily, like bench.py."""

from __future__ import annotations

import os
import pathlib
import statistics
import subprocess
import sys
import time

import numpy as np  # noqa: F401

from .calibrate import (FailedOnSomeRank, WallBudget, _agree, _slowest_rank_ms, best_of, calibrate_laplacian,  # noqa: F401
                        calibrate_transports, calibration_line_keys, calibration_seconds, direct_fenced, direct_mode,
                        direct_step_down, hdiff_calibration_order, ladder_line_keys, lap_calibration_order, lap_candidate_of,
                        lap_key, measure_candidate, run_calibration)

ROOT = pathlib.Path(__file__).resolve().parents[2]
GRID = (512, 512, 512)
HDIFF_SHARE = (512, 1024, 80)  # per-rank share of BASELINE.json configs[4]
HDIFF_GLOBAL = (2048, 2048, 80)
# how long a device-side wait of the direct transport may take in THIS program before its plan fails (the library's default is
# 30 s): a broken direct transport costs the calibration this much per wait, then its forms are dropped
DIRECT_TIMEOUT_MS = int(os.environ.get("GT4MI_BENCH_DIRECT_TIMEOUT_MS", "2000"))


def set_levels(levels: int) -> None:
    """The one-device rehearsal (N processes on ONE device, GT4MI_BENCH_ONE_DEVICE=1) runs slabs of `levels` levels instead of the
    full K extent: N kernels that wait for each other share one device's wave slots (bench.py main)."""
    global GRID, HDIFF_SHARE, HDIFF_GLOBAL
    GRID = (GRID[0], GRID[1], int(levels))
    HDIFF_SHARE, HDIFF_GLOBAL = (HDIFF_SHARE[0], HDIFF_SHARE[1], int(levels)), (HDIFF_GLOBAL[0], HDIFF_GLOBAL[1], int(levels))

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
        ok, port_taken = 0, 0
        try:
            fails = os.environ.get("GT4MI_BENCH_TEST_CANARY_FAILS", "")  # (tests: "1" = both modes fail, "unfenced" = only the default mode)
            if fails == "1" or (fails == "unfenced" and not fenced):
                raise RuntimeError("simulated for the tests")
            cmd = [sys.executable, "-m", "gt4py_amd.distributed", "--transport", "direct", "--domain", "256", "192", "8",
                   "--epochs", str(CHECK_EPOCHS), "--stress-epochs", str(CANARY_STRESS_EPOCHS)] + (["--fenced"] if fenced else [])
            proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
            ok = int(proc.returncode == 0)
            # (several hosts only: the port rank 0 found free a moment ago may have been taken before the child's store bound it)
            port_taken = int(not ok and where[0][0] == "tcp" and ("EADDRINUSE" in proc.stderr or "ddress already in use" in proc.stderr))
            if not ok:
                print(f"rank {rank}: the canary of the direct transport{' (fenced)' if fenced else ''} ended with status "
                      f"{proc.returncode}: {(proc.stdout + proc.stderr)[-900:]}", file=sys.stderr)
        except Exception as ex:  # (a timeout: the child is killed)
            print(f"rank {rank}: the canary of the direct transport{' (fenced)' if fenced else ''} failed ({ex!r})", file=sys.stderr)
        good = bool(_agree(ctx, ok))  # (every child has ended on every rank: the directory is no longer needed)
        if tmpdir is not None:
            shutil.rmtree(tmpdir, ignore_errors=True)
        if not good and attempt == 0 and not _agree(ctx, int(not port_taken)):  # some rank lost the race for the port: once more, fresh port
            return children(fenced, attempt=1)
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
              "calibration": calibration_line_keys(calibration, stats, ctx, {"total": total, "halo": 1, "itemsize": 8, "grid": grid,
                                                                             "selfloop_grid": selfloop_grid})
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

    hint = hd.placement_hint()  # in / coeff / out -> memory classes 0 / 0 / 1 (storage/placement.py: what is written goes to the other group)

    def field(lo, hi, cls=None):
        f = gt_storage.empty(dec.local_shape, np.float64, backend="hip:mi300", aligned_index=dec.origin, memory_class=cls)
        f.tensor.copy_(torch.rand(dec.local_shape, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    fields = {"in_field": hdiff_input(dec.local_shape, np.float64, gen, dec.origin, cls=hint["in_field"]),
              "coeff": field(0.025, 0.025, hint["coeff"]), "out_field": field(-1.0, 1.0, hint["out_field"])}
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
