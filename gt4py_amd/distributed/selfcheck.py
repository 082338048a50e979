"""A self-check of a halo exchange that needs no gather and no reference implementation: every rank fills its own points
with an exactly representable function of their GLOBAL coordinates and its ghost cells with a sentinel, exchanges, and then
knows what every ghost cell must hold -- the function at that cell's global coordinates (wrapped on periodic axes), or still
the sentinel where the cell lies beyond a physical boundary (ghost cells there stay untouched).

The function carries an EPOCH: F(i, j, k) + 65 536 * epoch, advanced on every ``FormCheck.reset()``.  Without it only the first
exchange of a plan could expose a receive side that reads its buffer before the peer's stores have landed: the buffers keep
the previous exchange's payload, and with the same probe every time that stale payload IS the right answer (VERDICT round 4,
"What's weak" 1).  With it the previous payload is off by exactly 65 536 in every cell, so every round is sensitive.

``bench.py`` runs it on every rank before it reports an N > 1 number (the first run between two real devices has nobody
else to check it); the CPU tests run it over gloo.  NEW relative to the reference (single-device).
"""

from __future__ import annotations

from typing import Tuple

from .halo import Decomposition

SENTINEL = -1.0
EPOCH_STEP = 65536.0  # > max F = 65 521: the value ranges of two epochs never meet; exact in float64 up to epoch 2**37


def coordinate_values(decomp: Decomposition, device="cpu", epoch: int = 0):
    """(own, expected) as float64 torch tensors of ``decomp.local_shape`` (axes I, J, K; any layout may hold them).

    ``own``: F(global i, j, k) on the rank's own points, ``SENTINEL`` in the ghost cells -- the field before the exchange.
    ``expected``: what a correct exchange leaves -- F at every ghost cell whose global coordinates lie inside the global
    domain (after wrapping on periodic axes), ``SENTINEL`` elsewhere.  F = 1 + (7919 i + 104729 j + 1299709 k) mod 65521
    + 65 536 * ``epoch``: a positive integer (exact in float64, equality is exact) that jumps from point to point, so a
    stencil applied to it is no constant either and a flux limiter fires on part of it; two epochs share no value."""
    import torch

    (gdi, gdj, _), (di, dj, dk), h = decomp.global_domain, decomp.local_domain, decomp.halo
    oi, oj, _ = decomp.offset
    gi = torch.arange(-h, di + h, dtype=torch.float64, device=device) + oi
    gj = torch.arange(-h, dj + h, dtype=torch.float64, device=device) + oj
    k = torch.arange(dk, dtype=torch.float64, device=device)
    inside_i = (gi >= 0) & (gi < gdi)
    inside_j = (gj >= 0) & (gj < gdj)
    if decomp.periodic[0]:
        gi, inside_i = torch.remainder(gi, gdi), torch.ones_like(inside_i)
    if decomp.periodic[1]:
        gj, inside_j = torch.remainder(gj, gdj), torch.ones_like(inside_j)
    gi3, gj3, k3 = gi[:, None, None], gj[None, :, None], k[None, None, :]
    f = torch.remainder(7919.0 * gi3 + 104729.0 * gj3 + 1299709.0 * k3, 65521.0) + 1.0 + EPOCH_STEP * float(epoch)
    inside = (inside_i[:, None, None] & inside_j[None, :, None]).expand_as(f)
    expected = torch.where(inside, f, torch.full_like(f, SENTINEL))
    own = torch.full_like(f, SENTINEL)
    own[h:h + di, h:h + dj, :] = f[h:h + di, h:h + dj, :]
    return own, expected


def count_wrong_cells(decomp: Decomposition, exchanged, epoch: int = 0) -> Tuple[int, int]:
    """(cells of ``exchanged`` that differ from what a correct exchange of ``coordinate_values(decomp, epoch=epoch)[0]`` leaves,
    ghost cells that should have been filled).  ``exchanged``: a float64 torch tensor of ``decomp.local_shape``."""
    _, expected = coordinate_values(decomp, exchanged.device, epoch)
    h, (di, dj, _) = decomp.halo, decomp.local_domain
    filled = int((expected != SENTINEL).sum().item()) - di * dj * expected.shape[2]
    return int((exchanged != expected).sum().item()), filled


class HbmLoad:
    """An HBM-saturating background for a check: back-to-back device copies (2 x ``mbytes`` MB of traffic each, ~90 us at
    6 TB/s for the default) on a stream of its own.  The window in which the round-3 receive side read stale payload is a
    payload load served BEFORE the peer's store landed next to a flag load served AFTER its add; an idle memory system serves
    both in order almost always, a saturated one -- what every real apply has beside its exchange: the interior kernel --
    does not.  A check that only ever runs on an idle device checks the easy case."""

    def __init__(self, device, mbytes: int = 256):
        import torch

        n = (int(mbytes) << 20) // 8
        self.src = torch.empty(n, dtype=torch.float64, device=device).fill_(1.0)
        self.dst = torch.empty_like(self.src)
        self.stream = torch.cuda.Stream(device=device)

    def start(self, copies: int = 24) -> None:
        """Enqueue ``copies`` copies behind whatever the CURRENT stream holds now (the probe's refill) and return at once."""
        import torch

        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(copies):
                self.dst.copy_(self.src, non_blocking=True)

    def wait(self) -> None:
        self.stream.synchronize()


class FormCheck:
    """Does one form of a distributed apply (a fused step with some schedule, message table, throttle ...) do what the
    plain definition says?  Fields on this rank: ``probe`` = ``coordinate_values`` of the current epoch (own points F, ghost
    cells the sentinel), ``out`` = zeros.  After ``reset()`` -- which ADVANCES THE EPOCH, so that what the previous round left
    in any receive buffer, ghost cell or cache is wrong everywhere -- the caller runs the form once on (``probe``, ``out``);
    ``verdict()`` then compares

    * every cell of ``probe`` with what a correct exchange leaves (``expected``), and
    * ``out`` with ``ref_out`` = ``reference_apply(expected, .)``: the LOCAL kernel over the whole local domain on the exactly
      known input, no exchange involved -- a ring point computed from a ghost cell that had not arrived yet differs even
      when the ghost cell is correct by the time anybody looks.

    ``check(run, rounds=3, loaded=1)`` runs ``rounds`` consecutive epochs, the last ``loaded`` of them next to an
    HBM-saturating background (``HbmLoad``), and returns the first failing verdict or the last passing one.

    ``new_field()`` returns a zero-initialised float64 device array of ``decomp.local_shape`` with a ``.tensor`` view;
    ``reference_apply(in_array, out_array)`` enqueues the local kernel on the current stream."""

    def __init__(self, decomp: Decomposition, new_field, reference_apply):
        self.decomp = decomp
        self.probe, self.out, self._ref_in, self.ref_out = new_field(), new_field(), new_field(), new_field()
        self._reference_apply = reference_apply
        self._own0, self._expected0 = coordinate_values(decomp, self.probe.tensor.device)
        self._mine = self._own0 != SENTINEL      # the rank's own points
        self._known = self._expected0 != SENTINEL  # ... and the ghost cells a correct exchange fills
        self._expected = self._expected0
        self.epoch = 0  # (advanced by reset(): the first round runs in epoch 1)
        self.rounds_checked = 0
        self._load = None
        h, (di, dj, dk) = decomp.halo, decomp.local_domain
        self.ghost_cells_to_fill = int(self._known.sum().item()) - di * dj * dk

    def _shifted(self, base, where):
        import torch

        return torch.where(where, base + EPOCH_STEP * float(self.epoch), base)

    def reset(self) -> None:
        """A new epoch: ``probe`` <- own points F + 65 536 epoch, ghost cells the sentinel; ``out`` <- 0; ``ref_out`` <- the local
        kernel on this epoch's exactly known input (enqueued on the current stream)."""
        self.epoch += 1
        self.probe.tensor.copy_(self._shifted(self._own0, self._mine))
        self.out.tensor.zero_()
        self._expected = self._shifted(self._expected0, self._known)
        self._ref_in.tensor.copy_(self._expected)
        self._reference_apply(self._ref_in, self.ref_out)

    def verdict(self) -> Tuple[bool, str]:
        import torch

        if self.probe.tensor.is_cuda:
            torch.cuda.synchronize()
        self.rounds_checked += 1
        differ = self.probe.tensor != self._expected
        wrong_cells = int(differ.sum().item())
        wrong_out = int((self.out.tensor != self.ref_out.tensor).sum().item())
        ok = wrong_cells == 0 and wrong_out == 0
        stale = ""
        if wrong_cells:  # how many of the wrong cells hold exactly the PREVIOUS epoch's value: a receive buffer read too early
            n_stale = int((differ & (self.probe.tensor == self._expected - EPOCH_STEP)).sum().item())
            stale = f" [{n_stale} of them hold the previous epoch's value: read before the peer's stores had landed]"
        return ok, (f"{wrong_cells} cells of the exchanged field differ from F(global coordinates){stale} "
                    f"({self.ghost_cells_to_fill} ghost cells to fill), {wrong_out} points of the result differ from the "
                    f"local kernel on the exactly known input [epoch {self.epoch}]")

    def check(self, run, rounds: int = 3, loaded=1, before_run=None) -> Tuple[bool, str]:
        """``rounds`` consecutive epochs of ``run()`` (the form on (``probe``, ``out``), enqueued on the current stream; it may
        join side streams itself), the last ``loaded`` of them -- or, ``loaded="alternate"``, every other one -- while ``HbmLoad``
        keeps the memory system busy on a third stream.  BOTH kinds of round matter: a saturated memory system widens the window
        between a store and its visibility, but on ONE device shared by two ranks it also keeps the ranks' kernels from meeting
        in flight, which HIDES a receive side that reads too early (profiles/r5_two_rank_direct_loop.log: the round-3 load order
        fails 1 % of the idle rounds and none of the loaded ones).  ``before_run()`` (optional) runs after each reset and before
        ``run`` -- where ranks meet (a barrier) so that they launch together.  The first failing verdict ends it; else the last
        one, with the count of rounds."""
        import torch

        on_gpu = self.probe.tensor.is_cuda
        verdict = (True, "no round ran")
        n_loaded = 0
        for r in range(rounds):
            self.reset()
            under_load = on_gpu and (r % 2 == 1 if loaded == "alternate" else r >= rounds - int(loaded))
            n_loaded += int(under_load)
            if under_load and self._load is None:
                self._load = HbmLoad(self.probe.tensor.device)
            if before_run is not None:
                if on_gpu:
                    torch.cuda.synchronize()
                before_run()
            if under_load:
                self._load.start()
            try:
                run()
                ok, found = self.verdict()
            finally:
                if under_load:
                    self._load.wait()
            verdict = (ok, found + (" (under HBM load)" if under_load else ""))
            if not ok:
                return verdict
        return verdict[0], f"{rounds} epochs, {n_loaded} of them under HBM load; last: {verdict[1]}"


def run_selfcheck(domain=(256, 192, 16), periodic: bool = False, transports=("native", "torch"), out=None, epochs: int = 3,
                  stress_epochs: int = 0, fenced: bool = False) -> int:
    """The exchange and every fused distributed apply, checked on exactly known fields on every rank of the job (or, with
    ``periodic``, of a world of one rank whose neighbours are the rank itself).  Launch one process per GPU
    (``python -m torch.distributed.run --nproc-per-node N -m gt4py_amd.distributed``); returns the number of failed
    checks over all ranks, rank 0 prints the table.

    Every form runs ``epochs`` consecutive epochs of the probe (``FormCheck.check``: each round's correct values differ from
    the previous round's in every cell), the last one next to an HBM-saturating background.  ``stress_epochs`` > 0: the
    one-stream form of the direct transport -- what ``bench.py`` is most likely to time -- additionally runs that many epochs,
    every other one under load, the ranks meeting before every launch (``bench.py``'s canary asks for 200).  ``fenced``: the direct
    transport in its fenced mode (``tune(direct_fenced=True)``: release / acquire fences around the flags)."""
    import os
    import sys

    import numpy as np
    import torch
    import torch.distributed as dist

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    from . import HaloExchanger, NativeComm, NativeHaloExchanger, choose_process_grid, overlapped_apply, sequential_apply

    out = out or sys.stdout
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    started_group = False
    direct_only = tuple(transports) == ("direct",)  # the direct transport alone: no RCCL anywhere, the descriptions travel over gloo
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():  # (launched by torch.distributed.run, any world size)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if direct_only and os.environ.get("GT4MI_RENDEZVOUS_FILE"):
            # (bench.py's canary: the children of the ranks meet through a file in a directory of their own -- no port that was
            # free a moment ago, no second listener on the launcher's address)
            dist.init_process_group("gloo", store=dist.FileStore(os.environ["GT4MI_RENDEZVOUS_FILE"], world), rank=rank, world_size=world)
        elif direct_only:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        started_group = True
    results = []  # (name, ok, detail)
    if "torch" in transports and not dist.is_initialized():  # (the torch transport needs a process group, also to talk to itself)
        transports = tuple(t for t in transports if t != "torch")
    if direct_only:
        comm = NativeComm(rank=rank, world_size=world, rccl=False)
    else:
        comm = NativeComm() if "native" in transports else None
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64})
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64})
    try:
        for halo in (1, 2):
            grid = choose_process_grid(world, domain, halo)
            wrap = (periodic or False, periodic or False)
            dec = Decomposition(tuple(domain), grid, rank, halo, periodic=wrap)
            new = lambda: gt_storage.zeros(dec.local_shape, np.float64, backend="hip:mi300", aligned_index=dec.origin)  # noqa: E731
            if halo == 1:
                fr = lap.freeze(origin={"inp": dec.origin, "out": dec.origin}, domain=dec.local_domain)
                local = lambda a, b: fr(inp=a, out=b)  # noqa: E731
                names, stencil = ("inp", "out"), lap
                extra = {}
            else:
                coeff = new()
                coeff.tensor.fill_(0.025)
                fr = hd.freeze(origin={n: dec.origin for n in ("in_field", "out_field", "coeff")}, domain=dec.local_domain)
                local = lambda a, b: fr(in_field=a, out_field=b, coeff=coeff)  # noqa: E731
                names, stencil = ("in_field", "out_field"), hd
                extra = {"coeff": coeff}
            chk = FormCheck(dec, new, local)
            origin = {n: dec.origin for n in names + tuple(extra)}

            def meet():
                if world > 1 and dist.is_initialized():
                    dist.barrier()

            def record(name, run, rounds=None, loaded=1):
                try:
                    ok, detail = chk.check(run, rounds or epochs, loaded, before_run=meet if rounds else None)
                except Exception as ex:  # noqa: BLE001 - a failing form is a result, not the end of the check
                    ok, detail = False, repr(ex)
                results.append((f"halo {halo} grid {grid[0]}x{grid[1]} {name}", ok, detail))

            for single in (False, True):
                table = "single-phase" if single else "two-phase"
                fields = {names[0]: chk.probe, names[1]: chk.out, **extra}
                if "torch" in transports:
                    ex = HaloExchanger(dec, torch.float64, torch.device("cuda", local_rank), single_phase=single)
                    record(f"torch {table} sequential", lambda: sequential_apply(stencil, dec, origin, fields, {names[0]: ex}))
                    record(f"torch {table} overlapped", lambda: overlapped_apply(stencil, dec, origin, fields, {names[0]: ex}))
                for how in ((("direct",) if direct_only else ("rccl", "direct")) if comm is not None else ()):
                    nex = NativeHaloExchanger(dec, np.float64, comm, single_phase=single)
                    if how == "direct":  # peer stores from the pack kernel (csrc/direct.hip.h); collective, all ranks fail together
                        try:
                            nex.use_direct_transport()
                            nex.tune(direct_fenced=fenced)
                        except RuntimeError as ex:  # (no fine-grained memory / hipIpc on some rank: RCCL remains -- reported, not a failure)
                            results.append((f"halo {halo} grid {grid[0]}x{grid[1]} native/direct {table}: NOT AVAILABLE ({str(ex)[:120]})",
                                            not direct_only, "the direct transport was asked for and is not available"))
                            nex.close()
                            continue
                    record(f"native/{how} {table} sequential", lambda: sequential_apply(stencil, dec, origin, fields, {names[0]: nex}))
                    for schedule in ("join", "chain", "swap", "swap-packed", "inline"):
                        for wg in (0, 2):
                            nex.tune(schedule, wg)
                            fused = (nex.make_dist_lap5(chk.probe, chk.out, dec.origin, dec.origin) if halo == 1 else
                                     nex.make_dist_hdiff(chk.probe, chk.out, coeff, dec.origin, type(hd)._gt_binding_.flags))

                            def run(fused=fused):
                                fused()
                                nex.end()

                            record(f"native/{how} {table} fused {schedule} wg{wg}", run)
                            if how == "direct" and schedule == "inline" and wg == 0 and stress_epochs > 0:
                                record(f"native/{how} {table} fused {schedule} wg{wg} x {stress_epochs} epochs, every other one under load", run,
                                       rounds=stress_epochs, loaded="alternate")
                    if how == "direct" and nex.direct_status()["timed_out"]:
                        results.append((f"halo {halo} grid {grid[0]}x{grid[1]} native/direct {table}: waits", False, "a wait ran out of time"))
                    nex.close()  # (collective on the direct transport: nobody unmaps a pool a peer may still push into)
            del chk
    finally:
        if comm is not None:
            comm.close()
    mine = [(n, bool(ok), d) for n, ok, d in results]
    everyone = [mine]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    failed = sum(1 for per_rank in everyone for _, ok, _ in per_rank if not ok)
    if rank == 0:
        for i, (name, _, _) in enumerate(mine):
            verdicts = [per_rank[i] for per_rank in everyone]
            bad = [(r, v[2]) for r, v in enumerate(verdicts) if not v[1]]
            print(f"{name:60s} {'ok on every rank' if not bad else 'WRONG on ' + ', '.join(f'rank {r}: {d}' for r, d in bad)}", file=out)
        print(f"{len(mine)} checks x {world} rank(s): {'all correct' if failed == 0 else str(failed) + ' FAILED'}", file=out)
    if started_group:
        _leave_together(dist, world)
        dist.destroy_process_group()
    return failed


def _leave_together(dist, world: int, seconds: float = 60.0) -> None:
    """Every rank counts itself out on the group's store and waits until all have: nobody closes its sockets while another rank
    is still inside a collective (a `barrier` followed by `destroy_process_group` is exactly that race on gloo: the first rank
    through tears the pair down under the last one)."""
    import time

    try:
        store = dist.distributed_c10d._get_default_store()
        store.add("gt4mi_selfcheck_ranks_done", 1)
        t0 = time.monotonic()
        while store.add("gt4mi_selfcheck_ranks_done", 0) < world and time.monotonic() - t0 < seconds:
            time.sleep(0.01)
    except Exception:  # noqa: BLE001 - leaving is best effort; the results have been gathered and printed
        pass


def main(argv=None) -> int:
    """``python -m gt4py_amd.distributed [--domain I J K] [--periodic] [--transport native|torch|both]``"""
    import argparse

    ap = argparse.ArgumentParser(prog="python -m gt4py_amd.distributed", description=run_selfcheck.__doc__)
    ap.add_argument("--domain", type=int, nargs=3, default=(256, 192, 16), metavar=("I", "J", "K"), help="GLOBAL compute domain")
    ap.add_argument("--periodic", action="store_true", help="wrap both axes (with one rank: every neighbour is the rank itself)")
    ap.add_argument("--transport", choices=("native", "torch", "both", "direct"), default="both",
                    help="direct: only the direct transport of the native path, without RCCL (bench.py's canary)")
    ap.add_argument("--epochs", type=int, default=3, help="consecutive epochs of the probe per form (the last one under HBM load)")
    ap.add_argument("--stress-epochs", type=int, default=0,
                    help="the direct transport's one-stream forms additionally run this many epochs, every other one under HBM load, "
                         "ranks launching together (bench.py's canary: 200)")
    ap.add_argument("--fenced", action="store_true", help="the direct transport in its fenced mode (release / acquire around the flags)")
    a = ap.parse_args(argv)
    return 1 if run_selfcheck(tuple(a.domain), a.periodic, ("native", "torch") if a.transport == "both" else (a.transport,),
                              epochs=a.epochs, stress_epochs=a.stress_epochs, fenced=a.fenced) else 0
