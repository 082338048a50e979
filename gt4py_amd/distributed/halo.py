"""IJ domain decomposition and ghost-cell (halo) exchange -- NEW relative to the reference.

gt4py.cartesian has no communication layer at all (no NCCL/MPI/GHEX call site);
a stencil call is single-process, single-device.  This module adds what an 8-GPU MI355X node
needs: one process per GPU, a Cartesian (PI x PJ) split of the I and J axes (K is never split: the
K-sequential sweeps stay on one GPU), and a per-step exchange of the read fields' ghost cells with
the 4 face neighbours as RCCL point-to-point messages (``torch.distributed`` batch_isend_irecv =
ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd) issued on a side stream so that it overlaps
the interior compute.  Messages are small (<= 2 MB for 512^3) and therefore latency-bound;
there is no all-reduce anywhere on the path.

Two-phase exchange: I faces first, then J faces *including* the freshly received I-halo columns,
which delivers the corner cells horizontal diffusion needs (lap[+-1, 0] reads in[+-1, +-1]) without
diagonal messages.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np



class _LazyModule:
    """``torch`` / ``torch.distributed``, imported on first use: the pure-Python parts of this package (message tables, process
    grids, ``calibrate``) are imported by programs that must start without paying for -- or having -- torch (``import bench``
    on the profiling path, tests/test_bench_infrastructure.py)."""

    def __init__(self, name: str):
        self.__dict__["_name"] = name

    def __getattr__(self, attr):
        import importlib

        return getattr(importlib.import_module(self.__dict__["_name"]), attr)


torch = _LazyModule("torch")
dist = _LazyModule("torch.distributed")


def process_grid_candidates(n: int, domain: Sequence[int], halo: int = 1, min_rows: int = 8):
    """Every (PI, PJ) with PI * PJ == n whose smallest local block is at least ``2 * halo`` points wide
    along a cut axis (so that a ghost zone is filled by ONE neighbour) and at least ``min_rows`` rows high."""
    out = []
    for pi in range(1, n + 1):
        if n % pi:
            continue
        pj = n // pi
        li, lj = domain[0] // pi, domain[1] // pj  # the smallest block of an even split
        if li < 1 or lj < 1:
            continue
        if (pi > 1 and li < 2 * halo) or (pj > 1 and lj < max(2 * halo, min_rows)):
            continue
        out.append((pi, pj))
    return out


def exchange_cost(grid: Tuple[int, int], domain: Sequence[int], halo: int = 1):
    """(largest message in points, number of sequential phases) of one ghost-cell exchange on ``grid``.

    xGMI is point-to-point: every neighbour has its own link, so an exchange takes as long as its LARGEST
    message per phase (not the sum), plus one latency per phase (I faces, then J faces)."""
    pi, pj = grid
    li, lj = -(-domain[0] // pi), -(-domain[1] // pj)
    # I faces also carry the ghost rows on sides without a J neighbour (halo_boxes)
    i_face = halo * (lj + (2 * halo if pj == 1 else halo)) * domain[2] if pi > 1 else 0
    j_face = halo * (li + 2 * halo) * domain[2] if pj > 1 else 0
    return max(i_face, j_face), (pi > 1) + (pj > 1)


def choose_process_grid(n: int, domain: Sequence[int], halo: int = 1) -> Tuple[int, int]:
    """(PI, PJ) with PI * PJ == n: the static default; ``bench.py`` measures the candidates instead.

    Link-aware: minimise the largest message of an exchange (``exchange_cost``), then the number of phases.
    1 x 8 on a 512^3 grid sends two 2.1 MB faces per rank, 4 x 2 and 2 x 4 send at most 1.05 MB.  Exact ties
    (a square domain cut along both axes) go to the grid with more cuts along I, which is the 4 x 2 that
    BASELINE.json configs[4] names for the 2048 x 2048 x 80 horizontal diffusion; with a single cut axis the
    cut goes along J, whose faces are contiguous rows of the I-contiguous storage.
    """
    candidates = process_grid_candidates(n, domain, halo)
    if not candidates:
        raise ValueError(f"cannot split domain {tuple(domain)} over {n} ranks")

    def key(grid):
        largest, phases = exchange_cost(grid, domain, halo)
        pi, pj = grid
        one_axis = pi == 1 or pj == 1
        return (largest, phases, pi if one_axis else -pi)

    return min(candidates, key=key)


def _split(n: int, parts: int, index: int) -> Tuple[int, int]:
    """(start, size) of block ``index`` when n points are dealt to ``parts`` blocks as evenly as possible."""
    base, rem = divmod(n, parts)
    start = index * base + min(index, rem)
    return start, base + (1 if index < rem else 0)


@dataclass(frozen=True)
class Decomposition:
    """Placement of one rank in the (PI x PJ) process grid over a global (dI, dJ, dK) domain."""

    global_domain: Tuple[int, int, int]
    grid: Tuple[int, int]
    rank: int
    halo: int
    #: periodic wrap-around in (I, J); default False = physical boundary (ghost cells untouched,
    #: as the reference leaves halos untouched -- SURVEY.md section 8a N4)
    periodic: Tuple[bool, bool] = (False, False)

    @property
    def coords(self) -> Tuple[int, int]:
        return self.rank % self.grid[0], self.rank // self.grid[0]  # I fastest

    def rank_of(self, ci: int, cj: int) -> Optional[int]:
        if self.periodic[0]:
            ci %= self.grid[0]
        if self.periodic[1]:
            cj %= self.grid[1]
        if 0 <= ci < self.grid[0] and 0 <= cj < self.grid[1]:
            return cj * self.grid[0] + ci
        return None

    @property
    def offset(self) -> Tuple[int, int, int]:
        ci, cj = self.coords
        return _split(self.global_domain[0], self.grid[0], ci)[0], _split(self.global_domain[1], self.grid[1], cj)[0], 0

    @property
    def local_domain(self) -> Tuple[int, int, int]:
        ci, cj = self.coords
        return (_split(self.global_domain[0], self.grid[0], ci)[1], _split(self.global_domain[1], self.grid[1], cj)[1],
                self.global_domain[2])

    @property
    def local_shape(self) -> Tuple[int, int, int]:
        d, h = self.local_domain, self.halo
        return d[0] + 2 * h, d[1] + 2 * h, d[2]

    @property
    def origin(self) -> Tuple[int, int, int]:
        return self.halo, self.halo, 0

    @property
    def neighbours(self) -> Dict[str, Optional[int]]:
        ci, cj = self.coords
        return {"W": self.rank_of(ci - 1, cj), "E": self.rank_of(ci + 1, cj),
                "S": self.rank_of(ci, cj - 1), "N": self.rank_of(ci, cj + 1)}

    def global_slices(self, with_halo: bool = True) -> Tuple[slice, slice, slice]:
        """Index of this rank's block (optionally with halo) inside the halo-padded GLOBAL array."""
        (oi, oj, _), (di, dj, dk), h = self.offset, self.local_domain, self.halo
        if with_halo:
            return slice(oi, oi + di + 2 * h), slice(oj, oj + dj + 2 * h), slice(0, dk)
        return slice(oi + h, oi + h + di), slice(oj + h, oj + h + dj), slice(0, dk)

    def interior_and_strips(self):
        """Split the local compute domain into the part that needs no remote halo and the strips
        that do: returns (interior, [strips]) as (origin_shift(3), domain(3)) pairs relative to the
        local compute-domain origin.  Only sides with a neighbour produce a strip."""
        (di, dj, dk), h, nb = self.local_domain, self.halo, self.neighbours
        lo_i = h if nb["W"] is not None else 0
        hi_i = h if nb["E"] is not None else 0
        lo_j = h if nb["S"] is not None else 0
        hi_j = h if nb["N"] is not None else 0
        lo_i, hi_i = min(lo_i, di), min(hi_i, max(di - lo_i, 0))
        lo_j, hi_j = min(lo_j, dj), min(hi_j, max(dj - lo_j, 0))
        interior = ((lo_i, lo_j, 0), (di - lo_i - hi_i, dj - lo_j - hi_j, dk))
        strips = []
        if lo_j:
            strips.append(((0, 0, 0), (di, lo_j, dk)))
        if hi_j:
            strips.append(((0, dj - hi_j, 0), (di, hi_j, dk)))
        if lo_i:
            strips.append(((0, lo_j, 0), (lo_i, dj - lo_j - hi_j, dk)))
        if hi_i:
            strips.append(((di - hi_i, lo_j, 0), (hi_i, dj - lo_j - hi_j, dk)))
        strips = [s for s in strips if all(x > 0 for x in s[1])]
        return interior, strips


class HipPacker:
    """Pack/unpack boxes of a strided device field with the gfx950 kernels of libgt4py_amd."""

    def __init__(self):
        from .. import _lib

        self._lib_mod = _lib
        self._lib = _lib.load()

    def _field(self, tensor) -> "ctypes.Structure":
        isz = tensor.element_size()
        return self._lib_mod.Field.make(tensor.data_ptr(), tuple(tensor.shape), tuple(s * isz for s in tensor.stride()),
                                        (0, 0, 0))

    def pack(self, tensor, lo, ext, buffer) -> None:
        stream = torch.cuda.current_stream().cuda_stream
        rc = self._lib.gt4mi_halo_pack(ctypes.byref(self._field(tensor)), self._lib_mod.domain3(lo),
                                       self._lib_mod.domain3(ext), buffer.data_ptr(), tensor.element_size(), stream)
        self._lib_mod.check("gt4mi_halo_pack", rc)

    def unpack(self, tensor, lo, ext, buffer) -> None:
        stream = torch.cuda.current_stream().cuda_stream
        rc = self._lib.gt4mi_halo_unpack(ctypes.byref(self._field(tensor)), self._lib_mod.domain3(lo),
                                         self._lib_mod.domain3(ext), buffer.data_ptr(), tensor.element_size(), stream)
        self._lib_mod.check("gt4mi_halo_unpack", rc)


Box = Tuple[int, Tuple[int, int, int], Tuple[int, int, int], Tuple[int, int, int]]


def halo_boxes(decomp: Decomposition) -> List[List[Box]]:
    """Per phase, the (peer, send_lo, recv_lo, extent) boxes of this rank in LOCAL array indices,
    low side first.

    Phase 0, I faces: they span the owned rows plus the halo rows on sides WITHOUT a J neighbour --
    those rows hold physical-boundary data the neighbour's corner reads need; halo rows on sides with
    a J neighbour are not valid yet and arrive (I-halo columns included) in phase 1.
    Phase 1, J faces over the full I extent including the halo columns -> corners travel with them.
    """
    h = decomp.halo
    di, dj, dk = decomp.local_domain
    si = decomp.local_shape[0]
    nb = decomp.neighbours
    phases: List[List[Box]] = [[], []]
    j_lo = 0 if nb["S"] is None else h
    j_hi = dj + 2 * h if nb["N"] is None else dj + h
    if nb["W"] is not None:
        phases[0].append((nb["W"], (h, j_lo, 0), (0, j_lo, 0), (h, j_hi - j_lo, dk)))
    if nb["E"] is not None:
        phases[0].append((nb["E"], (di, j_lo, 0), (di + h, j_lo, 0), (h, j_hi - j_lo, dk)))
    if nb["S"] is not None:
        phases[1].append((nb["S"], (0, h, 0), (0, 0, 0), (si, h, dk)))
    if nb["N"] is not None:
        phases[1].append((nb["N"], (0, dj, 0), (0, dj + h, 0), (si, h, dk)))
    return phases


#: sides of a block in message order, and the side a message sent towards each of them arrives on
SIDES = ("W", "E", "S", "N", "SW", "SE", "NW", "NE")
OPPOSITE = {"W": "E", "E": "W", "S": "N", "N": "S", "SW": "NE", "NE": "SW", "SE": "NW", "NW": "SE"}
_SIDE_STEP = {"W": (-1, 0), "E": (1, 0), "S": (0, -1), "N": (0, 1), "SW": (-1, -1), "SE": (1, -1), "NW": (-1, 1), "NE": (1, 1)}

SideBox = Tuple[str, int, Tuple[int, int, int], Tuple[int, int, int], Tuple[int, int, int]]


def halo_sides(decomp: Decomposition, single_phase: bool = False) -> List[List[SideBox]]:
    """Per phase, the (side, peer, send_lo, recv_lo, extent) boxes of this rank in LOCAL array indices, in ``SIDES`` order.

    Two-phase (default): ``halo_boxes`` with the side named.  Single-phase: ONE round with up to 8 neighbours -- the four
    faces (owned cells only along the other axis, plus the ghost cells on a side WITHOUT a neighbour there: physical
    boundary data the neighbour's corner reads need) and the four h x h corners to the diagonal neighbours.  Half the
    dependent rounds of pack -> send/recv -> unpack; on a fully connected xGMI node every neighbour has a link of its own.
    """
    if not single_phase:
        names = (("W", "E"), ("S", "N"))
        out = []
        for p, phase in enumerate(halo_boxes(decomp)):
            nb = decomp.neighbours
            sides = [sd for sd in names[p] if nb[sd] is not None]
            out.append([(sd, *box) for sd, box in zip(sides, phase)])
        return out
    h = decomp.halo
    di, dj, dk = decomp.local_domain
    ci, cj = decomp.coords
    nb = {sd: decomp.rank_of(ci + st[0], cj + st[1]) for sd, st in _SIDE_STEP.items()}
    # along an axis: (send range, receive range) of the low / high side, and the range of a face along the OTHER axis
    lo_send, lo_recv = (h, 2 * h), (0, h)

    def hi(d):
        return (d, d + h), (d + h, d + 2 * h)

    def across(d, low_nb, high_nb):  # owned cells + ghost cells on sides without a neighbour
        return (0 if low_nb is None else h), (d + 2 * h if high_nb is None else d + h)

    boxes: List[SideBox] = []
    for sd in SIDES:
        peer = nb[sd]
        if peer is None:
            continue
        si, sj = _SIDE_STEP[sd]
        if si == 0:
            i_send = i_recv = across(di, nb["W"], nb["E"])
        else:
            i_send, i_recv = (lo_send, lo_recv) if si < 0 else hi(di)
        if sj == 0:
            j_send = j_recv = across(dj, nb["S"], nb["N"])
        else:
            j_send, j_recv = (lo_send, lo_recv) if sj < 0 else hi(dj)
        ext = (i_send[1] - i_send[0], j_send[1] - j_send[0], dk)
        boxes.append((sd, peer, (i_send[0], j_send[0], 0), (i_recv[0], j_recv[0], 0), ext))
    return [boxes, []]


def receive_order(phase: Sequence[SideBox]) -> List[int]:
    """Indices into ``phase`` in the order the receives must be posted.  RCCL / gloo pair the k-th send to a peer with the
    k-th receive posted for that peer; sends go out in ``SIDES`` order, so the receive that matches the peer's send
    towards side s is the one into my ghost zone on the OPPOSITE side: post receives in the order of the opposite sides.
    (With a periodic axis of 1 or 2 ranks several messages of a phase go to the same peer -- the case that needs it.)"""
    by_side = {box[0]: n for n, box in enumerate(phase)}
    return [by_side[OPPOSITE[sd]] for sd in SIDES if OPPOSITE[sd] in by_side]


class HaloExchanger:
    """Halo exchange of one field shape/dtype for one rank through ``torch.distributed`` point-to-point
    operations (RCCL on GPUs, gloo in the CPU tests), with persistent staging buffers."""

    def __init__(self, decomp: Decomposition, dtype, device, packer=None, group=None, single_phase: bool = False,
                 stage_on_host: bool = False):
        self.decomp = decomp
        self.group = group
        self.device = torch.device(device)
        #: device buffers are copied to pinned host twins for the transfer (a process group whose backend cannot move device
        #: memory, e.g. gloo between two processes that share ONE GPU -- how the decomposed GPU path runs with two real
        #: ranks on a 1-GPU box, tests/test_gpu_distributed.py)
        self.stage_on_host = bool(stage_on_host) and self.device.type == "cuda"
        self.packer = packer if packer is not None else HipPacker()
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.sides = halo_sides(decomp, single_phase)
        self.phases = [[box[1:] for box in phase] for phase in self.sides]
        self.buffers = {}
        for p, phase in enumerate(self.phases):
            for m, (_, _, _, ext) in enumerate(phase):
                n = int(np.prod(ext))
                self.buffers[(p, m, "send")] = torch.empty(n, dtype=dtype, device=self.device)
                self.buffers[(p, m, "recv")] = torch.empty(n, dtype=dtype, device=self.device)
                if self.stage_on_host:
                    self.buffers[(p, m, "send", "host")] = torch.empty(n, dtype=dtype, pin_memory=True)
                    self.buffers[(p, m, "recv", "host")] = torch.empty(n, dtype=dtype, pin_memory=True)

    @property
    def bytes_per_exchange(self) -> int:
        return sum(b.numel() * b.element_size() for key, b in self.buffers.items() if key[2] == "send" and len(key) == 3)

    def _run_phase(self, tensor, p: int) -> None:
        phase = self.phases[p]
        if not phase:
            return
        ops = []
        for m, (_, send_lo, _, ext) in enumerate(phase):
            self.packer.pack(tensor, send_lo, ext, self.buffers[(p, m, "send")])
        # NCCL / gloo match the k-th send to a peer with the k-th receive posted for that peer, so the ORDER is
        # part of the protocol: sends go out in SIDES order, receives in the order of the opposite sides
        # (receive_order).  On a periodic axis with 1 or 2 ranks several faces of a phase go to the same peer, and
        # my low-side face must land in the peer's high-side ghost zone (NativeHaloExchanger.message_tables does
        # the same).
        wire = ("host",) if self.stage_on_host else ()
        if self.stage_on_host:
            for m in range(len(phase)):
                self.buffers[(p, m, "send", "host")].copy_(self.buffers[(p, m, "send")], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        for m, (peer, _, _, _) in enumerate(phase):
            ops.append(dist.P2POp(dist.isend, self.buffers[(p, m, "send") + wire], peer, self.group))
        for m in receive_order(self.sides[p]):
            ops.append(dist.P2POp(dist.irecv, self.buffers[(p, m, "recv") + wire], phase[m][0], self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if self.stage_on_host:
            for m in range(len(phase)):
                self.buffers[(p, m, "recv")].copy_(self.buffers[(p, m, "recv", "host")], non_blocking=True)
        for m, (_, _, recv_lo, ext) in enumerate(phase):
            self.packer.unpack(tensor, recv_lo, ext, self.buffers[(p, m, "recv")])

    def exchange(self, tensor) -> None:
        """Blocking (stream-ordered on GPU) exchange of ``tensor``'s halo on the CURRENT stream."""
        for p in range(len(self.phases)):
            self._run_phase(tensor, p)

    def start(self, tensor):
        """Launch the exchange on the side stream; returns an event to wait on.  On a CPU device (gloo, the
        tests) there are no streams: the exchange completes here and the handle is None."""
        if self.stream is None:
            self.exchange(tensor)
            return None
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.exchange(tensor)
            done = torch.cuda.Event()
            done.record(self.stream)
        return done

    def finish(self, done) -> None:
        if done is not None:
            torch.cuda.current_stream(self.device).wait_event(done)


def scatter_global(global_array: np.ndarray, decomp: Decomposition) -> np.ndarray:
    """This rank's halo-padded block of a halo-padded global host array (test/bench helper)."""
    return np.ascontiguousarray(global_array[decomp.global_slices(with_halo=True)])
