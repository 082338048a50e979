"""Halo exchange driven from native code: RCCL send/recv + pack/unpack + stream choreography inside
``libgt4py_amd.so`` (``gt4mi_comm_*``, ``gt4mi_halo_plan_*``, ``gt4mi_halo_exchange*``,
``gt4mi_dist_lap5_f64`` in include/gt4py_amd.h).

The Python classes below only build the box tables once and then make ONE ctypes call per exchange
(or per distributed stencil apply), which matters when a step is tens of microseconds of GPU time.
NEW relative to the reference: gt4py.cartesian has no multi-device path.
"""

from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from .. import _lib
from .halo import Decomposition, halo_sides, receive_order


def _stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


class NativeComm:
    """An RCCL communicator owned by libgt4py_amd (one rank per process/GPU).

    The 128-byte unique id is created on rank 0 and broadcast through ``torch.distributed`` when a
    process group is initialised (any launcher would do); a single-process run needs no group.
    """

    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None, group=None, rccl: bool = True):
        lib = _lib.load()
        if not rccl:
            # no RCCL communicator behind it (gt4mi_comm_create_local): for ranks RCCL cannot join -- two processes on ONE device,
            # the rehearsal a 1-GPU box allows -- whose exchangers then use the direct transport only
            handle = ctypes.c_void_p()
            _lib.check("gt4mi_comm_create_local", lib.gt4mi_comm_create_local(int(world_size), int(rank), ctypes.byref(handle)))
            self.rank, self.world_size, self._handle, self._lib = int(rank), int(world_size), handle, lib
            return
        if rank is None or world_size is None:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                rank, world_size = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world_size = 0, 1
        self.rank, self.world_size = int(rank), int(world_size)
        uid = ctypes.create_string_buffer(128)
        failure = None
        if self.rank == 0:
            try:
                _lib.check("gt4mi_comm_unique_id", lib.gt4mi_comm_unique_id(uid))
            except Exception as ex:  # the other ranks are waiting for the broadcast: tell them instead of leaving them there
                failure = f"rank 0 could not create the RCCL unique id: {ex}"
        if self.world_size > 1:
            import torch.distributed as dist

            box = [(uid.raw, failure) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            raw, failure = box[0]
            uid = ctypes.create_string_buffer(raw, 128)
        if failure is not None:
            raise RuntimeError(failure)
        handle = ctypes.c_void_p()
        _lib.check("gt4mi_comm_create", lib.gt4mi_comm_create(uid, self.world_size, self.rank, ctypes.byref(handle)))
        self._handle = handle
        self._lib = lib

    @property
    def handle(self) -> ctypes.c_void_p:
        return self._handle

    def info(self) -> dict:
        """What RCCL itself reports for this communicator: {"nranks": ncclCommCount, "rank": ncclCommUserRank,
        "device": ncclCommCuDevice} -- the proof a benchmark line can carry that RCCL joined N ranks."""
        n, r, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check("gt4mi_comm_info", self._lib.gt4mi_comm_info(self._handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)))
        return {"nranks": n.value, "rank": r.value, "device": d.value}

    def close(self) -> None:
        if self._handle is not None and self._handle.value:
            self._lib.gt4mi_comm_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass


def _field_struct(array, origin=(0, 0, 0)) -> _lib.Field:
    """gt4mi_field of a DeviceArray or torch tensor."""
    if hasattr(array, "ptr"):
        return _lib.Field.make(array.ptr, array.shape, array.strides, origin)
    isz = array.element_size()
    return _lib.Field.make(array.data_ptr(), tuple(array.shape), tuple(s * isz for s in array.stride()), origin)


_POISON = {"why": None}


def poison_collectives(why: str) -> None:
    """A helper thread of this process is stuck inside a collective of the job's process group (an unmet closing round): from now on
    this rank must not issue another collective on that group."""
    if _POISON["why"] is None:
        _POISON["why"] = str(why)


def collectives_poisoned():
    """None, or why this rank may no longer issue collectives on the job's process group."""
    return _POISON["why"]


class CollectivesPoisoned(RuntimeError):
    """Raised instead of issuing a collective next to one that is still pending in a helper thread of this process."""


class NativeHaloExchanger:
    """Two-phase ghost-cell exchange of one field shape through a native plan (see halo.halo_boxes)."""

    @staticmethod
    def message_tables(decomp: Decomposition, single_phase: bool = False):
        """(sends, recvs) of one rank as plain tuples (peer, phase, lo, extent), in the order RCCL sees them.

        RCCL matches the k-th send to a peer with the k-th receive posted for that peer inside one group, so
        the ORDER is part of the protocol: sends go out in ``halo.SIDES`` order, receives in the order of the opposite
        sides (``halo.receive_order``) -- with a periodic axis of 1 or 2 ranks several messages of a phase go to the same
        peer, and my low-side face must land in the peer's HIGH-side ghost zone.  ``single_phase``: one round with
        faces and corner boxes to up to 8 neighbours instead of two rounds with 4 (``halo.halo_sides``).  Pure Python
        (checked for whole process grids on the CPU in tests/test_distributed.py: every send has a receive of the same
        size waiting for it)."""
        sends, recvs = [], []
        for p, phase in enumerate(halo_sides(decomp, single_phase)):
            for _, peer, send_lo, _, ext in phase:
                sends.append((int(peer), p, tuple(int(v) for v in send_lo), tuple(int(v) for v in ext)))
            for m in receive_order(phase):
                _, peer, _, recv_lo, ext = phase[m]
                recvs.append((int(peer), p, tuple(int(v) for v in recv_lo), tuple(int(v) for v in ext)))
        return sends, recvs

    def __init__(self, decomp: Decomposition, dtype, comm: NativeComm, single_phase: bool = False):
        self.decomp = decomp
        self.comm = comm
        self.single_phase = bool(single_phase)
        self.itemsize = np.dtype(dtype).itemsize
        send_table, recv_table = self.message_tables(decomp, single_phase)
        sends = [_lib.HaloMsg.make(*m) for m in send_table]
        recvs = [_lib.HaloMsg.make(*m) for m in recv_table]
        self.bytes_per_exchange = sum(int(np.prod(tuple(m.extent))) for m in sends) * self.itemsize
        self.largest_message_bytes = max([int(np.prod(tuple(m.extent))) for m in sends] or [0]) * self.itemsize
        SendArr, RecvArr = _lib.HaloMsg * max(len(sends), 1), _lib.HaloMsg * max(len(recvs), 1)
        plan = ctypes.c_void_p()
        lib = _lib.load()
        _lib.check("gt4mi_halo_plan_create",
                   lib.gt4mi_halo_plan_create(comm.handle, self.itemsize, SendArr(*sends), len(sends),
                                              RecvArr(*recvs), len(recvs), ctypes.byref(plan)))
        self._plan = plan
        self._lib = lib
        self.transport = "rccl"
        self.direct_fenced = False
        self._failed = False  # a timeout of the direct transport was seen (synchronize / close)
        self.failed_close_seconds = 20.0  # how long close() of a FAILED plan waits for the other ranks before it leaks the pool
        self._close_round = None  # direct transport with other ranks: the collective that precedes the release of the pool
        nb = decomp.neighbours
        self.sides = ((1 if nb["W"] is not None else 0) | (2 if nb["E"] is not None else 0)
                      | (4 if nb["S"] is not None else 0) | (8 if nb["N"] is not None else 0))

    def tune(self, schedule: Optional[str] = None, interior_wg_per_cu: Optional[int] = None,
             defer_join: Optional[bool] = None, edge_columns: Optional[int] = None,
             direct_timeout_ms: Optional[int] = None, direct_fenced: Optional[bool] = None) -> "NativeHaloExchanger":
        """How the fused distributed steps built on this exchanger are scheduled (gt4mi_halo_plan_set_option):
        ``schedule`` "join" (pack and interior on the caller's stream, send/recv/unpack beside it, join, ring), "chain"
        (the caller's stream carries the interior only; pack, send/recv, unpack and ring in order on the side stream) or
        "swap" (the Laplacian step: that chain on the caller's stream, the interior kernel on the side stream; "swap-packed":
        the interior forks off after the pack);
        ``interior_wg_per_cu`` limits the occupancy of the interior kernel while the exchange runs next to it (0 = no
        limit); ``defer_join`` (chain schedule) lets a fused step return without joining the side stream -- for INDEPENDENT
        applies, whose results the caller consumes only after ``end()``; ``direct_fenced``: the direct transport's fenced mode
        (GT4MI_PLAN_DIRECT_FENCED: a system-scope release before every flag is raised, an acquire behind every flag load --
        the fall-back between the default direct transport and RCCL).  ``None`` leaves an option as it is."""
        if schedule is not None:
            value = {"join": _lib.SCHEDULE_JOIN, "chain": _lib.SCHEDULE_CHAIN, "swap": _lib.SCHEDULE_SWAP, "swap-packed": _lib.SCHEDULE_SWAP_PACKED, "inline": _lib.SCHEDULE_INLINE,
                     "default": -1}[schedule]
            _lib.check("gt4mi_halo_plan_set_option", self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_SCHEDULE, value))
        if interior_wg_per_cu is not None:
            _lib.check("gt4mi_halo_plan_set_option",
                       self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_INTERIOR_WG_PER_CU, int(interior_wg_per_cu)))
        if edge_columns is not None:  # fused hdiff: width of the W / E boxes the ring kernel takes off the interior
            _lib.check("gt4mi_halo_plan_set_option",
                       self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_EDGE_COLUMNS, int(edge_columns)))
        if defer_join is not None:  # independent applies: the caller joins with end() before consuming results
            _lib.check("gt4mi_halo_plan_set_option",
                       self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_DEFER_JOIN, int(bool(defer_join))))
        if direct_timeout_ms is not None:  # direct transport: how long a device-side wait may take before the plan fails (0: default)
            _lib.check("gt4mi_halo_plan_set_option",
                       self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_DIRECT_TIMEOUT_MS, int(direct_timeout_ms)))
        if direct_fenced is not None:
            _lib.check("gt4mi_halo_plan_set_option",
                       self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_DIRECT_FENCED, int(bool(direct_fenced))))
            self.direct_fenced = bool(direct_fenced)
        return self

    @staticmethod
    def direct_wiring(rank: int, peers_of: dict):
        """Which receive every send of ``rank`` lands in, and which send fills every receive -- RCCL's matching rule, which the
        direct transport keeps: the k-th send to a peer pairs with the k-th receive that peer posted for this rank, per phase.
        ``peers_of[r] = (send_peers, recv_peers)`` with the peers per phase in plan order (``message_tables``).  Returns
        ``(send_to, recv_from)``: ``send_to[p][m] = (peer, index of the peer's receive)``, ``recv_from[p][m] = (peer, index of the
        peer's send)``.  Pure Python: checked for whole process grids on the CPU (tests/test_distributed.py)."""

        def kth(peers_list, who, k):
            seen = -1
            for i, q in enumerate(peers_list):
                seen += q == who
                if q == who and seen == k:
                    return i
            raise RuntimeError(f"rank {who} has no message number {k} for rank {rank}: the message tables do not pair up")

        send_peers, recv_peers = peers_of[rank]
        send_to = [[(q, kth(peers_of[q][1][p], rank, send_peers[p][:m].count(q))) for m, q in enumerate(send_peers[p])] for p in (0, 1)]
        recv_from = [[(q, kth(peers_of[q][0][p], rank, recv_peers[p][:m].count(q))) for m, q in enumerate(recv_peers[p])] for p in (0, 1)]
        return send_to, recv_from

    # ---- the direct transport: peer stores from the pack kernel instead of RCCL send/recv (csrc/direct.hip.h) ----
    def use_direct_transport(self, group=None, all_gather=None) -> "NativeHaloExchanger":
        """Switch this exchanger -- and every fused step built on it -- to the direct transport.  COLLECTIVE over the ranks of
        the decomposition: every rank exports its pool of flag words and receive buffers (hipIpc), the descriptions travel
        through ``torch.distributed.all_gather_object`` (``all_gather(obj) -> list``: any other channel; not needed when every
        neighbour is the rank itself), and every message is connected to its counterpart by RCCL's matching rule: the k-th
        send to a peer lands in the buffer of the k-th receive that peer posted for this rank.

        A rank on which a step fails (no fine-grained device memory, a handle that cannot be opened) still takes part in
        both rounds of the gather, so that EVERY rank raises ``RuntimeError`` together and stays on RCCL.

        From here on ``close()`` is COLLECTIVE too (one more round over the same channel): the neighbours' kernels write into
        this rank's pool, which may only be released once every rank's last exchange has finished on its device.

        Failure is hard: a device-side wait for a neighbour that runs out of time (``tune(direct_timeout_ms=...)``, default
        30 s) leaves the ghost cells of that exchange incomplete and fails the plan -- the next ``exchange`` / fused step /
        ``end()`` raises ``NativeError`` (status ``ERR_TIMEOUT``), and so does every call after it."""
        lib, plan = self._lib, self._plan
        sends, recvs = self.message_tables(self.decomp, self.single_phase)
        rank = self.decomp.rank

        def per_phase(table):
            return [[m[0] for m in table if m[1] == p] for p in (0, 1)]  # peers, in plan order

        def layout(is_send, phase, index):
            off, flag = ctypes.c_int64(), ctypes.c_int()
            _lib.check("gt4mi_halo_plan_direct_layout",
                       lib.gt4mi_halo_plan_direct_layout(plan, phase, int(is_send), index, ctypes.byref(off), ctypes.byref(flag)))
            return int(off.value), int(flag.value)

        mine = {"rank": rank, "send_peers": per_phase(sends), "recv_peers": per_phase(recvs), "failure": None}
        # (decided by the process grid, which every rank knows alike -- not by this rank's own neighbours: the gathers below are
        # collective)
        alone = self.decomp.grid[0] * self.decomp.grid[1] == 1
        if not alone and all_gather is None:
            import torch.distributed as dist

            def all_gather(obj):
                out = [None] * dist.get_world_size(group)
                dist.all_gather_object(out, obj, group=group)
                return out

        try:
            info = _lib.DirectInfo()
            _lib.check("gt4mi_halo_plan_direct_prepare", lib.gt4mi_halo_plan_direct_prepare(plan, ctypes.byref(info)))
            mine["info"] = bytes(info)
            mine["recv_layout"] = [[layout(False, p, m) for m in range(len(mine["recv_peers"][p]))] for p in (0, 1)]
            mine["send_flags"] = [[layout(True, p, m)[1] for m in range(len(mine["send_peers"][p]))] for p in (0, 1)]
        except Exception as ex:  # noqa: BLE001 - reported to every rank below
            mine["failure"] = f"rank {rank}: {ex}"
        everyone = {rank: mine} if alone else {e["rank"]: e for e in all_gather(mine)}
        failures = [e["failure"] for e in everyone.values() if e["failure"]]
        keep = []  # the DirectInfo structures must outlive the connect calls only

        def info_of(q):
            if q == rank:
                return None
            peer = _lib.DirectInfo.from_buffer_copy(everyone[q]["info"])
            keep.append(peer)
            return ctypes.byref(peer)

        mine_failure = None
        if not failures:
            try:
                send_to, recv_from = self.direct_wiring(rank, {r: (e["send_peers"], e["recv_peers"]) for r, e in everyone.items()})
                for p in (0, 1):
                    for m, (q, j) in enumerate(send_to[p]):
                        off, flag = everyone[q]["recv_layout"][p][j]
                        _lib.check("gt4mi_halo_plan_direct_connect",
                                   lib.gt4mi_halo_plan_direct_connect(plan, p, 1, m, info_of(q), off, flag))
                    for m, (q, j) in enumerate(recv_from[p]):
                        _lib.check("gt4mi_halo_plan_direct_connect",
                                   lib.gt4mi_halo_plan_direct_connect(plan, p, 0, m, info_of(q), 0, everyone[q]["send_flags"][p][j]))
            except Exception as ex:  # noqa: BLE001
                mine_failure = f"rank {rank}: {ex}"
        # (second round: nobody starts pushing before every rank has mapped its peers -- or everybody gives up together)
        failures += [f for f in ([mine_failure] if alone else all_gather(mine_failure)) if f]
        if failures:
            raise RuntimeError("the direct halo transport is not available on every rank: " + "; ".join(sorted(set(failures))))
        _lib.check("gt4mi_halo_plan_set_option", lib.gt4mi_halo_plan_set_option(plan, _lib.PLAN_TRANSPORT, _lib.TRANSPORT_DIRECT))
        self.transport = "direct"
        self._close_round = None if alone else all_gather
        return self

    def use_rccl_transport(self) -> "NativeHaloExchanger":
        _lib.check("gt4mi_halo_plan_set_option", self._lib.gt4mi_halo_plan_set_option(self._plan, _lib.PLAN_TRANSPORT, _lib.TRANSPORT_RCCL))
        self.transport = "rccl"
        return self

    def direct_status(self) -> dict:
        """{"timed_out": a wait of the direct transport ever ran out of time (synchronises), "exchanges": started so far}."""
        t, n = ctypes.c_int(), ctypes.c_uint()
        _lib.check("gt4mi_halo_plan_direct_status", self._lib.gt4mi_halo_plan_direct_status(self._plan, ctypes.byref(t), ctypes.byref(n)))
        return {"timed_out": bool(t.value), "exchanges": int(n.value)}

    def synchronize(self) -> None:
        """Wait for everything enqueued so far and FAIL if a wait of the direct transport ran out of time in any exchange
        started on this plan (``NativeError``, status ``ERR_TIMEOUT``).  Timeouts are otherwise detected lazily -- by the NEXT call
        on the plan, and the call that enqueued the failing exchange has long returned OK --, so a caller whose last use of the
        plan is "exchange, synchronise, read the ghost cells" calls this instead of a bare device synchronise: the CONSUMING
        call fails, not some later producing one.  (``end()`` itself stays asynchronous: it is on the path of every timed
        apply.)  On the RCCL transport: a device synchronise."""
        if self.transport == "direct":
            if self.direct_status()["timed_out"]:  # (synchronises the device first)
                self._failed = True
                raise _lib.NativeError("gt4mi_halo_plan_direct_status", _lib.ERR_TIMEOUT,
                                       "direct transport: a wait for a neighbour ran out of time in an exchange of this plan; the ghost "
                                       "cells of that exchange are incomplete and the plan stays failed: close it on every rank "
                                       "(close(collective=False) if the neighbour may be gone)")
        else:
            import torch

            torch.cuda.synchronize()

    def exchange(self, array) -> None:
        """Enqueue the exchange of ``array``'s ghost cells on the current stream."""
        f = _field_struct(array)
        _lib.check("gt4mi_halo_exchange", self._lib.gt4mi_halo_exchange(self._plan, ctypes.byref(f), _stream_ptr()))

    def fork(self) -> None:
        """Mark the fork point on the current stream; the next ``begin`` waits only for earlier work."""
        _lib.check("gt4mi_halo_exchange_fork", self._lib.gt4mi_halo_exchange_fork(self._plan, _stream_ptr()))

    def begin(self, array) -> None:
        f = _field_struct(array)
        _lib.check("gt4mi_halo_exchange_begin",
                   self._lib.gt4mi_halo_exchange_begin(self._plan, ctypes.byref(f), _stream_ptr()))

    def end(self) -> None:
        _lib.check("gt4mi_halo_exchange_end", self._lib.gt4mi_halo_exchange_end(self._plan, _stream_ptr()))

    def make_dist_lap5(self, inp, out, origin_inp: Sequence[int], origin_out: Sequence[int], variant: int = 0,
                       hold_arrays: bool = True, flags: int = 0):
        """Pre-bind one distributed apply of a 5-point stencil (halo 1; gt4mi_dist_lap5_f64 / _f32 by the exchanger's item
        size, ``flags`` as for gt4mi_lap5_f32) -> a zero-argument callable.  ``hold_arrays=False``: the callable does not
        keep the arrays alive (a caller that checks their identity itself, ``distributed.fused_apply``)."""
        if self.decomp.halo != 1 or self.itemsize not in (4, 8):
            raise ValueError("gt4mi_dist_lap5_* needs fp64 / fp32 fields and a halo of 1")
        fi, fo = _field_struct(inp, origin_inp), _field_struct(out, origin_out)
        dom = _lib.domain3(self.decomp.local_domain)
        sides = self.sides
        ri, ro = ctypes.byref(fi), ctypes.byref(fo)
        if self.itemsize == 8:
            name, fn, head = "gt4mi_dist_lap5_f64", self._lib.gt4mi_dist_lap5_f64, (variant,)
        else:
            name, fn, head = "gt4mi_dist_lap5_f32", self._lib.gt4mi_dist_lap5_f32, (variant, int(flags))

        def apply():
            rc = fn(self._plan, dom, ri, ro, *head, sides, _stream_ptr())  # (a closed exchanger's plan is NULL: an error, not a crash)
            if rc:
                _lib.check(name, rc)

        apply._keepalive = (fi, fo, dom) + ((inp, out) if hold_arrays else ())  # type: ignore[attr-defined]
        return apply

    def lap5_uses_edge_units(self, inp, out, origin_inp: Sequence[int], origin_out: Sequence[int]) -> bool:
        """Would ``make_dist_lap5`` on these arrays run the unpack and the boundary strips as the edge units of
        csrc/lap5_edge.hip.h (on the direct transport's one-stream schedule: the whole apply as ONE launch)?  Nothing is
        enqueued (gt4mi_dist_lap5_query)."""
        fi, fo = _field_struct(inp, origin_inp), _field_struct(out, origin_out)
        answer = ctypes.c_int()
        _lib.check("gt4mi_dist_lap5_query",
                   self._lib.gt4mi_dist_lap5_query(self._plan, _lib.domain3(self.decomp.local_domain), ctypes.byref(fi), ctypes.byref(fo),
                                                   self.sides, ctypes.byref(answer)))
        return bool(answer.value)

    def make_dist_hdiff(self, in_field, out_field, coeff, origin: Sequence[int], flags: int, coeff_scalar: float = 0.0,
                        hold_arrays: bool = True):
        """Pre-bind one distributed apply of horizontal diffusion (gt4mi_dist_hdiff_f64 / _f32: pack, interior next to the
        exchange, one ring kernel) -> a zero-argument callable.  ``coeff`` is a device array or None (then
        ``coeff_scalar``); ``flags`` as for gt4mi_hdiff_* (``KernelBinding.flags`` of a recognised stencil);
        ``hold_arrays`` as for ``make_dist_lap5``."""
        if self.decomp.halo != 2:
            raise ValueError("gt4mi_dist_hdiff needs ghost regions exactly 2 deep")
        fi, fo = _field_struct(in_field, origin), _field_struct(out_field, origin)
        fc = _field_struct(coeff, origin) if coeff is not None else None
        dom = _lib.domain3(self.decomp.local_domain)
        name = "gt4mi_dist_hdiff_f64" if self.itemsize == 8 else "gt4mi_dist_hdiff_f32"
        fn, sides = getattr(self._lib, name), self.sides
        ri, ro, rc_ = ctypes.byref(fi), ctypes.byref(fo), (ctypes.byref(fc) if fc is not None else None)
        cs, fl = float(coeff_scalar), int(flags)

        def apply():
            rc = fn(self._plan, dom, ri, ro, rc_, cs, fl, sides, _stream_ptr())
            if rc:
                _lib.check(name, rc)

        apply._keepalive = (fi, fo, fc, dom) + ((in_field, out_field, coeff) if hold_arrays else ())  # type: ignore[attr-defined]
        return apply

    def make_time_skewed_lap5(self, field_a, field_b, origin: Sequence[int], variant: int = 0):
        """Pre-bind the time-skewed Laplacian stepper (gt4mi_dist_lap5_f64_skewed): ONE call advances a whole cycle of
        H = ``decomp.halo`` steps, boundary bands first, then the faces travel next to all H interior kernels.  Bit-identical
        to H undecomposed steps.  Primes the pipeline with one exchange of ``field_a``.  Returns a zero-argument callable;
        ``callable.steps_per_call`` is H and ``callable.result()`` the field that holds the newest values."""
        if self.itemsize != 8:
            raise ValueError("the native Laplacian time stepper needs fp64 fields")
        halo = int(self.decomp.halo)
        fa, fb = _field_struct(field_a, origin), _field_struct(field_b, origin)
        dom = _lib.domain3(self.decomp.local_domain)
        fn, plan, sides = self._lib.gt4mi_dist_lap5_f64_skewed, self._plan, self.sides
        ra, rb = ctypes.byref(fa), ctypes.byref(fb)
        self.begin(field_a)
        state = {"cycles": 0}
        swaps = halo % 2 == 1  # an odd number of steps leaves the result in the other buffer

        def cycle():
            flipped = swaps and state["cycles"] % 2 == 1
            src, dst = (rb, ra) if flipped else (ra, rb)
            rc = fn(plan, dom, src, dst, variant, sides, halo, _stream_ptr())
            if rc:
                _lib.check("gt4mi_dist_lap5_f64_skewed", rc)
            state["cycles"] += 1

        def result():
            steps = state["cycles"] * halo
            return field_b if steps % 2 == 1 else field_a

        cycle.steps_per_call = halo  # type: ignore[attr-defined]
        cycle.result = result  # type: ignore[attr-defined]
        cycle._keepalive = (fa, fb, dom, field_a, field_b)  # type: ignore[attr-defined]
        return cycle

    @property
    def concurrent(self):
        """True / False once the side stream has been probed against the caller's stream, else None."""
        v = self._lib.gt4mi_halo_plan_concurrent(self._plan)
        return None if v == 2 else bool(v)

    def make_time_stepper_lap5(self, field_a, field_b, origin: Sequence[int], variant: int = 0, overlap: bool = True):
        """Pre-bind the pipelined time-stepping Laplacian: call n computes b = lap(a) for even n and
        a = lap(b) for odd n (gt4mi_dist_lap5_f64_wide).  With a ghost depth H = ``decomp.halo`` one
        exchange serves H steps: step n of a cycle (phase n % H) computes the local domain grown by
        H-1-phase rows towards every neighbour -- redundantly recomputing what the neighbour also
        computes -- and only the last phase exchanges the freshly written field's H-deep faces, next
        to its interior kernel.  H == 1 is the plain pipelined exchange-every-step scheme.  Results
        are bit-identical to the undecomposed run for every H (the redundant rows evaluate the same
        expression on the same values).  Primes the pipeline with one exchange of ``field_a``.
        ``overlap=False`` keeps everything on the caller's stream: the last phase is one full-domain kernel
        followed by the exchange (pack, send/recv, unpack) -- no boundary strips, no side stream, nothing for a
        send/recv kernel to compete with; which form wins depends on how long the links take.
        Returns a zero-argument callable; ``callable.result()`` is the field written last."""
        if self.itemsize != 8:
            raise ValueError("the native Laplacian time stepper needs fp64 fields")
        halo = int(self.decomp.halo)  # ghost depth == number of steps one exchange serves
        fa, fb = _field_struct(field_a, origin), _field_struct(field_b, origin)
        dom = _lib.domain3(self.decomp.local_domain)
        fn, plan, sides = self._lib.gt4mi_dist_lap5_f64_wide, self._plan, self.sides
        ra, rb = ctypes.byref(fa), ctypes.byref(fb)
        self.begin(field_a)  # ghost cells of the first input
        state = {"n": 0}

        lap, exchange = self._lib.gt4mi_lap5_f64, self._lib.gt4mi_halo_exchange
        if not overlap:
            self.end()  # the priming exchange is joined here; later ones run on the caller's stream

        def step():
            n = state["n"]
            src, dst = (ra, rb) if n % 2 == 0 else (rb, ra)
            stream = _stream_ptr()
            if overlap or n % halo < halo - 1:
                rc = fn(plan, dom, src, dst, variant, sides, halo, n % halo, stream)
                if rc:
                    _lib.check("gt4mi_dist_lap5_f64_wide", rc)
            else:
                rc = lap(dom, src, dst, variant, 0, stream, None) or exchange(plan, dst, stream)
                if rc:
                    _lib.check("gt4mi_lap5_f64 / gt4mi_halo_exchange", rc)
            state["n"] = n + 1

        step.result = lambda: field_b if state["n"] % 2 == 1 else field_a  # type: ignore[attr-defined]
        step._keepalive = (fa, fb, dom, field_a, field_b)  # type: ignore[attr-defined]
        return step

    def close(self, collective: bool = True) -> None:
        """Destroy the native plan.  Once ``use_direct_transport`` has connected this exchanger to OTHER ranks this is a
        COLLECTIVE call: every rank synchronises its device (its last pushes and "consumed" signals have landed), the ranks
        meet once on the channel that carried the pools' descriptions, and only then is the pool unmapped and freed --
        a neighbour's kernel never writes into memory that is gone.  ``collective=False`` is for a caller that has established
        exactly that itself: since this plan's last exchange every rank has synchronised its device AND met the others in a
        collective (``bench.py``'s candidates: the all-reduce of the timings)."""
        if self._plan is not None and self._plan.value:
            if self._close_round is not None and collective:
                import torch

                torch.cuda.synchronize()  # the whole device: the plan's side stream too
                # A plan whose wait ran out of time has a neighbour that never arrived -- it may be gone, and a collective with a
                # rank that is gone never completes.  The failed plan still OFFERS the round (live neighbours are waiting in it),
                # but from a helper thread and for a bounded time; if it does not complete the plan is released without it: its
                # pool is leaked rather than freed under a peer that may still push (a leak is harmless, a store into freed
                # memory is a fault on the peer's device).
                failed = self._failed
                try:
                    failed = failed or self.direct_status()["timed_out"]
                except Exception:  # noqa: BLE001
                    failed = True
                if failed:
                    import threading
                    import warnings

                    meet = threading.Thread(target=lambda: self._close_round(("closing", self.decomp.rank)), daemon=True)
                    meet.start()
                    meet.join(timeout=self.failed_close_seconds)
                    if meet.is_alive():
                        # the helper thread stays blocked INSIDE a collective of the job's process group: another collective from
                        # the main thread would run concurrently with it on the same group (mismatched or out-of-order collectives,
                        # a hang on the ranks that did meet).  The group is POISONED from here on: `collectives_poisoned()` is what
                        # `calibrate._agree` / `_slowest_rank_ms` consult before they issue one.
                        poison_collectives(f"rank {self.decomp.rank}: the closing round of a failed direct plan did not complete within "
                                           f"{self.failed_close_seconds:.0f} s")
                        warnings.warn("NativeHaloExchanger.close(): the plan's direct transport has FAILED (a wait for a neighbour ran "
                                      f"out of time) and the ranks did not meet within {self.failed_close_seconds:.0f} s: the pool is "
                                      "leaked instead of waiting for a neighbour that may be gone; no further collective will be "
                                      "issued on the job's process group by this rank", RuntimeWarning, stacklevel=2)
                        self._close_round = None
                        self._plan = ctypes.c_void_p()  # (the native plan and its pool stay allocated: see above)
                        return
                    self._close_round = None
                if self._close_round is not None:
                    self._close_round(("closing", self.decomp.rank))
            self._close_round = None
            self._lib.gt4mi_halo_plan_destroy(self._plan)
            self._plan = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        # (never collective: a garbage-collected exchanger that is still connected to other ranks keeps its pool -- a leak is
        # harmless, a neighbour's store into freed memory is a fault on ITS device)
        try:
            if self._close_round is None:
                self.close()
        except Exception:
            pass
