"""A self-check of a halo exchange that needs no gather and no reference implementation: every rank fills its own points
with an exactly representable function of their GLOBAL coordinates and its ghost cells with a sentinel, exchanges, and then
knows what every ghost cell must hold -- the function at that cell's global coordinates (wrapped on periodic axes), or still
the sentinel where the cell lies beyond a physical boundary (ghost cells there stay untouched).

``bench.py`` runs it on every rank before it reports an N > 1 number (the first run between two real devices has nobody
else to check it); the CPU tests run it over gloo.  NEW relative to the reference (single-device).
"""

from __future__ import annotations

from typing import Tuple

from .halo import Decomposition

SENTINEL = -1.0


def coordinate_values(decomp: Decomposition, device="cpu"):
    """(own, expected) as float64 torch tensors of ``decomp.local_shape`` (axes I, J, K; any layout may hold them).

    ``own``: F(global i, j, k) on the rank's own points, ``SENTINEL`` in the ghost cells -- the field before the exchange.
    ``expected``: what a correct exchange leaves -- F at every ghost cell whose global coordinates lie inside the global
    domain (after wrapping on periodic axes), ``SENTINEL`` elsewhere.  F = 1 + (7919 i + 104729 j + 1299709 k) mod 65521: a
    positive integer (exact in float64, equality is exact) that jumps from point to point, so a stencil applied to it is
    no constant either and a flux limiter fires on part of it."""
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
    f = torch.remainder(7919.0 * gi3 + 104729.0 * gj3 + 1299709.0 * k3, 65521.0) + 1.0
    inside = (inside_i[:, None, None] & inside_j[None, :, None]).expand_as(f)
    expected = torch.where(inside, f, torch.full_like(f, SENTINEL))
    own = torch.full_like(f, SENTINEL)
    own[h:h + di, h:h + dj, :] = f[h:h + di, h:h + dj, :]
    return own, expected


def count_wrong_cells(decomp: Decomposition, exchanged) -> Tuple[int, int]:
    """(cells of ``exchanged`` that differ from what a correct exchange of ``coordinate_values(decomp)[0]`` leaves, ghost
    cells that should have been filled).  ``exchanged``: a float64 torch tensor of ``decomp.local_shape``."""
    _, expected = coordinate_values(decomp, exchanged.device)
    h, (di, dj, _) = decomp.halo, decomp.local_domain
    filled = int((expected != SENTINEL).sum().item()) - di * dj * expected.shape[2]
    return int((exchanged != expected).sum().item()), filled


class FormCheck:
    """Does one form of a distributed apply (a fused step with some schedule, message table, throttle ...) do what the
    plain definition says?  Fields on this rank: ``probe`` = ``coordinate_values`` (own points F, ghost cells the sentinel),
    ``out`` = zeros.  After ``reset()`` the caller runs the form once on (``probe``, ``out``); ``verdict()`` then compares

    * every cell of ``probe`` with what a correct exchange leaves (``expected``), and
    * ``out`` with ``ref_out`` = ``reference_apply(expected, .)``: the LOCAL kernel over the whole local domain on the exactly
      known input, no exchange involved -- a ring point computed from a ghost cell that had not arrived yet differs even
      when the ghost cell is correct by the time anybody looks.

    ``new_field()`` returns a zero-initialised float64 device array of ``decomp.local_shape`` with a ``.tensor`` view;
    ``reference_apply(in_array, out_array)`` enqueues the local kernel on the current stream."""

    def __init__(self, decomp: Decomposition, new_field, reference_apply):
        import torch

        self.decomp = decomp
        self.probe, self.out, ref_in, self.ref_out = new_field(), new_field(), new_field(), new_field()
        self._own, self._expected = coordinate_values(decomp, self.probe.tensor.device)
        ref_in.tensor.copy_(self._expected)
        reference_apply(ref_in, self.ref_out)
        torch.cuda.synchronize() if self.probe.tensor.is_cuda else None
        h, (di, dj, dk) = decomp.halo, decomp.local_domain
        self.ghost_cells_to_fill = int((self._expected != SENTINEL).sum().item()) - di * dj * dk

    def reset(self) -> None:
        self.probe.tensor.copy_(self._own)
        self.out.tensor.zero_()

    def verdict(self) -> Tuple[bool, str]:
        import torch

        if self.probe.tensor.is_cuda:
            torch.cuda.synchronize()
        wrong_cells = int((self.probe.tensor != self._expected).sum().item())
        wrong_out = int((self.out.tensor != self.ref_out.tensor).sum().item())
        ok = wrong_cells == 0 and wrong_out == 0
        return ok, (f"{wrong_cells} cells of the exchanged field differ from F(global coordinates) "
                    f"({self.ghost_cells_to_fill} ghost cells to fill), {wrong_out} points of the result differ from the "
                    f"local kernel on the exactly known input")
