"""N > 1 path on the CPU: decomposition geometry, and a world_size-2 gloo run of
halo exchange + interior/strip execution that must reproduce the single-domain oracle bit-exactly.

On the GPU the same ``HaloExchanger`` runs with the HIP pack kernels and the RCCL backend; here the
transport is gloo on CPU tensors, the packer is a torch-slicing stand-in defined in this file, and
the compute is the oracle (tests may use it).
"""

import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from gt4py_amd.distributed import Decomposition, HaloExchanger, choose_process_grid, scatter_global
from mp_util import run_ranks
from oracle import ref_numpy as R


class TorchSlicePacker:
    """CPU stand-in for the HIP pack kernels: dense buffer is I-fastest, then J, then K."""

    @staticmethod
    def _box(t, lo, ext):
        return t[lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]]

    def pack(self, tensor, lo, ext, buffer):
        buffer.copy_(self._box(tensor, lo, ext).permute(2, 1, 0).reshape(-1))

    def unpack(self, tensor, lo, ext, buffer):
        self._box(tensor, lo, ext).copy_(buffer.reshape(ext[2], ext[1], ext[0]).permute(2, 1, 0))


def test_choose_process_grid_is_link_aware():
    """Smallest largest message first (xGMI is point-to-point: neighbours do not share a link), then fewer
    phases; a single cut goes along J (contiguous faces); BASELINE.json configs[4] gets its 4 x 2."""
    from gt4py_amd.distributed import exchange_cost, process_grid_candidates

    assert choose_process_grid(1, (512, 512, 512)) == (1, 1)
    assert choose_process_grid(2, (512, 512, 512)) == (1, 2)
    assert choose_process_grid(4, (512, 512, 512)) == (2, 2)
    assert choose_process_grid(8, (512, 512, 512)) == (4, 2)
    assert choose_process_grid(8, (2048, 2048, 80), halo=2) == (4, 2)
    assert choose_process_grid(8, (2048, 64, 80)) == (8, 1)  # J slabs of 8 rows would send 2 x 2050 x 80 points
    assert choose_process_grid(8, (64, 2048, 80)) == (1, 8)
    # the largest message is what the choice halves: 1 x 8 sends 514 x 512 points, 4 x 2 at most 258 x 512
    assert exchange_cost((1, 8), (512, 512, 512))[0] == 514 * 512 and exchange_cost((4, 2), (512, 512, 512))[0] == 257 * 512
    assert exchange_cost((1, 8), (512, 512, 512))[1] == 1 and exchange_cost((4, 2), (512, 512, 512))[1] == 2
    assert process_grid_candidates(8, (512, 512, 512)) == [(1, 8), (2, 4), (4, 2), (8, 1)]
    assert process_grid_candidates(8, (512, 16, 8), halo=2) == [(4, 2), (8, 1)]  # J blocks need >= max(2 halo, 8) rows
    with pytest.raises(ValueError):
        choose_process_grid(8, (3, 3, 3))


@pytest.mark.parametrize("grid", [(1, 1), (2, 1), (1, 2), (4, 2), (3, 3)])
def test_decomposition_tiles_the_domain(grid):
    gd, h = (37, 29, 3), 2
    seen = np.zeros(gd[:2], int)
    for rank in range(grid[0] * grid[1]):
        d = Decomposition(gd, grid, rank, h)
        (oi, oj, _), (di, dj, dk) = d.offset, d.local_domain
        seen[oi:oi + di, oj:oj + dj] += 1
        assert dk == gd[2] and d.local_shape == (di + 2 * h, dj + 2 * h, dk)
        nb = d.neighbours
        ci, cj = d.coords
        assert (nb["W"] is None) == (ci == 0) and (nb["E"] is None) == (ci == grid[0] - 1)
        assert (nb["S"] is None) == (cj == 0) and (nb["N"] is None) == (cj == grid[1] - 1)
        # interior + strips tile the local domain exactly once
        cover = np.zeros((di, dj), int)
        (si, sj, _), (ei, ej, _) = d.interior_and_strips()[0]
        if ei > 0 and ej > 0:
            cover[si:si + ei, sj:sj + ej] += 1
        for (si, sj, _), (ei, ej, _) in d.interior_and_strips()[1]:
            cover[si:si + ei, sj:sj + ej] += 1
        assert (cover == 1).all()
    assert (seen == 1).all()


def test_in_process_4x2_exchange_and_hdiff():
    """All 8 ranks simulated in one process (no transport): after the two-phase exchange every
    rank's halo -- corners included -- equals the global field, and decomposed hdiff == global."""
    rng = np.random.default_rng(4)
    gd, h, grid = (20, 14, 3), 2, (4, 2)
    glob = rng.uniform(-10, 10, (gd[0] + 2 * h, gd[1] + 2 * h, gd[2]))
    coeff = rng.uniform(0, 0.5, glob.shape)
    want = np.zeros_like(glob)
    R.hdiff(glob, want, coeff, domain=gd)
    decs = [Decomposition(gd, grid, r, h) for r in range(8)]
    local = []
    for d in decs:
        blk = scatter_global(glob, d).copy()
        # wipe the ghost cells that belong to neighbours: they must come from the exchange
        nb = d.neighbours
        if nb["W"] is not None:
            blk[:h] = np.nan
        if nb["E"] is not None:
            blk[-h:] = np.nan
        if nb["S"] is not None:
            blk[:, :h] = np.nan
        if nb["N"] is not None:
            blk[:, -h:] = np.nan
        local.append(torch.from_numpy(blk))
    p = TorchSlicePacker()
    exchangers = [HaloExchanger(d, torch.float64, "cpu", packer=p) for d in decs]
    for phase in (0, 1):  # play the exchanger's own box tables, delivering the buffers by hand
        mail = {}
        for ex, t in zip(exchangers, local):
            for peer, send_lo, _, ext in ex.phases[phase]:
                buf = torch.empty(int(np.prod(ext)), dtype=t.dtype)
                p.pack(t, send_lo, ext, buf)
                mail[(ex.decomp.rank, peer)] = (buf, ext)
        for ex, t in zip(exchangers, local):
            for peer, _, recv_lo, ext in ex.phases[phase]:
                buf, sent_ext = mail[(peer, ex.decomp.rank)]
                assert tuple(sent_ext) == tuple(ext)
                p.unpack(t, recv_lo, ext, buf)
    got = np.zeros_like(glob)
    for d, t in zip(decs, local):
        blk = t.numpy()
        assert np.array_equal(blk, scatter_global(glob, d))  # halos (with corners) restored exactly
        out = np.zeros_like(blk)
        cf = scatter_global(coeff, d)
        (shift, sub), strips = d.interior_and_strips()
        for sh, dom in [(shift, sub)] + strips:
            if all(x > 0 for x in dom):
                org = tuple(o + s for o, s in zip(d.origin, sh))
                R.hdiff(blk, out, cf, origin_in=org, origin_out=org, origin_coeff=org, domain=dom)
        got[d.global_slices(with_halo=False)] = out[h:-h, h:-h]
    assert np.array_equal(got[h:-h, h:-h], want[h:-h, h:-h])


@pytest.mark.parametrize("grid,periodic", [((1, 1), (True, True)), ((1, 2), (False, True)), ((2, 1), (True, False)),
                                           ((2, 2), (True, True)), ((1, 2), (True, True)), ((3, 2), (True, True))])
@pytest.mark.parametrize("halo", [1, 2])
@pytest.mark.parametrize("single_phase", [False, True])
def test_torch_transport_message_order_on_periodic_axes(grid, periodic, halo, single_phase):
    """HaloExchanger posts sends low side first and receives HIGH side first.  With 1 or 2 ranks on a periodic
    axis both faces of a phase go to the same peer and the k-th send is matched with the k-th receive (NCCL and
    gloo alike): replay every rank's operation list with that rule -- no transport needed -- and compare the
    ghost cells with the periodic global array."""
    gd = (6 * grid[0] + 1, 5 * grid[1] + 2, 2)
    n = grid[0] * grid[1]
    rng = np.random.default_rng(8)
    full = rng.uniform(-1, 1, (gd[0] + 2 * halo, gd[1] + 2 * halo, gd[2]))
    if periodic[0]:
        full[:halo], full[-halo:] = full[-2 * halo:-halo].copy(), full[halo:2 * halo].copy()
    if periodic[1]:
        full[:, :halo], full[:, -halo:] = full[:, -2 * halo:-halo].copy(), full[:, halo:2 * halo].copy()
    decs = [Decomposition(gd, grid, r, halo, periodic=periodic) for r in range(n)]
    have, want = [], []
    for d in decs:
        w = scatter_global(full, d).copy()
        h = w.copy()
        nb = d.neighbours
        if nb["W"] is not None:
            h[:halo] = np.nan
        if nb["E"] is not None:
            h[-halo:] = np.nan
        if nb["S"] is not None:
            h[:, :halo] = np.nan
        if nb["N"] is not None:
            h[:, -halo:] = np.nan
        have.append(torch.from_numpy(h))
        want.append(w)

    class Recorder:
        """stands in for torch.distributed: records the P2P operations of one batch"""

        def __init__(self):
            self.ops = []

    recorded = {}
    real_batch, real_p2p = dist.batch_isend_irecv, dist.P2POp

    class FakeOp:
        def __init__(self, op, tensor, peer, group=None):
            self.kind = "send" if op is dist.isend else "recv"
            self.tensor, self.peer = tensor, peer

    class Done:
        def wait(self):
            return None

    exchangers = [HaloExchanger(d, torch.float64, "cpu", packer=TorchSlicePacker(), single_phase=single_phase) for d in decs]
    try:
        dist.P2POp = FakeOp
        for phase in (0, 1):
            # every rank packs and "posts" its batch; unpack is deferred until the mail has been delivered
            batches = {}
            for r, ex in enumerate(exchangers):
                def fake_batch(ops, r=r):
                    batches[r] = list(ops)
                    return [Done() for _ in ops]

                dist.batch_isend_irecv = fake_batch
                box = ex.phases[phase]
                if not box:
                    continue
                for m, (_, send_lo, _, ext) in enumerate(box):
                    ex.packer.pack(have[r], send_lo, ext, ex.buffers[(phase, m, "send")])
                # the product's own posting order
                ops = []
                real_unpack = ex.packer.unpack
                ex.packer.unpack = lambda *a, **k: None
                ex._run_phase(have[r], phase)
                ex.packer.unpack = real_unpack
            mailbox = {}
            for r, ops in batches.items():
                for op in ops:
                    if op.kind == "send":
                        mailbox.setdefault((r, op.peer), []).append(op.tensor.clone())
            for r, ops in batches.items():
                for op in ops:
                    if op.kind == "recv":
                        op.tensor.copy_(mailbox[(op.peer, r)].pop(0))
            assert all(not q for q in mailbox.values())
            for r, ex in enumerate(exchangers):
                for m, (_, _, recv_lo, ext) in enumerate(ex.phases[phase]):
                    ex.packer.unpack(have[r], recv_lo, ext, ex.buffers[(phase, m, "recv")])
    finally:
        dist.batch_isend_irecv, dist.P2POp = real_batch, real_p2p
    for r in range(n):
        np.testing.assert_array_equal(have[r].numpy(), want[r], err_msg=f"rank {r}")


def _last_attempt(tmp_path):
    """The directory of the attempt that passed (tests/mp_util.run_ranks gives every attempt a directory of its own)."""
    return sorted(tmp_path.glob("attempt*"))[-1]


def _worker(rank: int, world: int, tmpdir: str, grid, single_phase: bool = False):
    rng = np.random.default_rng(99)  # same stream on every rank -> same global field
    gd, h = (24, 18, 4), 1
    glob = rng.uniform(-1, 1, (gd[0] + 2, gd[1] + 2, gd[2]))
    dec = Decomposition(gd, grid, rank, h)
    blk = scatter_global(glob, dec).copy()
    nb = dec.neighbours
    if nb["W"] is not None:
        blk[:h] = 0
    if nb["E"] is not None:
        blk[-h:] = 0
    if nb["S"] is not None:
        blk[:, :h] = 0
    if nb["N"] is not None:
        blk[:, -h:] = 0
    t = torch.from_numpy(blk)
    ex = HaloExchanger(dec, torch.float64, "cpu", packer=TorchSlicePacker(), single_phase=single_phase)
    ex.exchange(t)
    out = np.zeros_like(blk)
    (shift, sub), strips = dec.interior_and_strips()
    for sh, dom in [(shift, sub)] + strips:
        org = tuple(o + s for o, s in zip(dec.origin, sh))
        R.laplacian(blk, out, origin_inp=org, origin_out=org, domain=dom)
    # the gather-free self-check bench.py runs before it reports an N > 1 number (distributed/selfcheck.py)
    from gt4py_amd.distributed.selfcheck import coordinate_values, count_wrong_cells

    cut = (grid[0] > 1, grid[1] > 1)  # (gloo cannot send to the rank itself: periodic along the cut axes only)
    for pdec in (dec, Decomposition(gd, grid, rank, 2, periodic=cut), Decomposition(gd, grid, rank, 1, periodic=cut)):
        own, _ = coordinate_values(pdec)
        before = count_wrong_cells(pdec, own)
        HaloExchanger(pdec, torch.float64, "cpu", packer=TorchSlicePacker(), single_phase=single_phase).exchange(own)
        after = count_wrong_cells(pdec, own)
        assert before[0] == before[1] > 0 and after == (0, before[1]), (rank, pdec.periodic, before, after)
        own[0, 0, 0] += 1.0  # a single wrong ghost cell is seen
        assert count_wrong_cells(pdec, own)[0] == 1
    gathered = [None] * world
    dist.all_gather_object(gathered, (dec.global_slices(with_halo=False), out[h:-h, h:-h], ex.bytes_per_exchange))
    if rank == 0:
        got = np.zeros_like(glob)
        for sl, o, _ in gathered:
            got[sl] = o
        want = np.zeros_like(glob)
        R.laplacian(glob, want)
        np.save(os.path.join(tmpdir, "ok.npy"), np.array([np.array_equal(got, want), gathered[0][2]]))


@pytest.mark.multiprocess
@pytest.mark.parametrize("grid,single_phase", [((1, 2), False), ((2, 1), False), ((2, 1), True)])
def test_gloo_world_size_2_laplacian(grid, single_phase, tmp_path):
    run_ranks(_worker, 2, tmp_path, args=(grid, single_phase))
    ok, nbytes = np.load(_last_attempt(tmp_path) / "ok.npy")
    assert ok == 1
    assert nbytes > 0


@pytest.mark.multiprocess
def test_gloo_world_size_8_laplacian_on_the_4x2_process_grid_single_phase(tmp_path):
    """The north star's 512^3 decomposition (4 x 2) with eight real processes and the one-round message table (faces and
    corners to every neighbour at once)."""
    assert choose_process_grid(8, (512, 512, 512)) == (4, 2)
    run_ranks(_worker, 8, tmp_path, args=((4, 2), True))
    ok, nbytes = np.load(_last_attempt(tmp_path) / "ok.npy")
    assert ok == 1 and nbytes > 0


@pytest.mark.parametrize("grid,periodic", [((1, 8), (False, False)), ((4, 2), (False, False)), ((2, 2), (True, True)),
                                           ((1, 2), (False, True)), ((1, 1), (True, True)), ((2, 4), (True, False)),
                                           ((1, 4), (False, True))])
@pytest.mark.parametrize("halo", [1, 2])
@pytest.mark.parametrize("single_phase", [False, True])
def test_native_message_tables_pair_up_across_ranks(grid, periodic, halo, single_phase):
    """The native (RCCL) exchanger cannot run on more than one rank here, but its protocol can be checked:
    simulate every rank's message table and deliver the k-th send of rank r to peer p into p's k-th receive
    from r (RCCL's matching rule inside one group).  Every message must find a receive of the same extent,
    nothing may be left over, and after both phases every rank's ghost cells must hold the values of the
    periodic / bounded global array -- exactly what the torch transport is tested for above."""
    from gt4py_amd.distributed import Decomposition, NativeHaloExchanger

    global_domain = (16 * grid[0], 12 * grid[1], 3)
    n = grid[0] * grid[1]
    decs = [Decomposition(global_domain, grid, r, halo, periodic=periodic) for r in range(n)]
    rng = np.random.default_rng(0)
    gi, gj, gk = global_domain
    full = rng.uniform(-1, 1, (gi + 2 * halo, gj + 2 * halo, gk))
    # periodic axes: ghost cells of the global array wrap around; bounded axes keep their own values
    if periodic[0]:
        full[:halo], full[-halo:] = full[-2 * halo:-halo].copy(), full[halo:2 * halo].copy()
    if periodic[1]:
        full[:, :halo], full[:, -halo:] = full[:, -2 * halo:-halo].copy(), full[:, halo:2 * halo].copy()
    locals_ = []
    for d in decs:
        i0, j0 = d.offset[0], d.offset[1]
        li, lj, _ = d.local_domain
        want = full[i0:i0 + li + 2 * halo, j0:j0 + lj + 2 * halo].copy()
        have = want.copy()
        nb = d.neighbours
        # wipe the ghost cells that a neighbour is going to deliver
        if nb["W"] is not None:
            have[:halo] = np.nan
        if nb["E"] is not None:
            have[-halo:] = np.nan
        if nb["S"] is not None:
            have[:, :halo] = np.nan
        if nb["N"] is not None:
            have[:, -halo:] = np.nan
        locals_.append((have, want))
    tables = [NativeHaloExchanger.message_tables(d, single_phase) for d in decs]
    if single_phase:  # one round: faces and corners, at most 8 boxes per rank
        assert all(ph == 0 for sends, recvs in tables for _, ph, _, _ in (*sends, *recvs))
        assert max(len(sends) for sends, _ in tables) <= 8
    for phase in (0, 1):
        mailbox = {}
        for r, (sends, _) in enumerate(tables):
            for peer, ph, lo, ext in sends:
                if ph != phase:
                    continue
                box = locals_[r][0][lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]].copy()
                mailbox.setdefault((r, peer), []).append(box)
        for r, (_, recvs) in enumerate(tables):
            for peer, ph, lo, ext in recvs:
                if ph != phase:
                    continue
                queue = mailbox.get((peer, r))
                assert queue, f"rank {r} waits for a message from {peer} that is never sent (phase {phase})"
                box = queue.pop(0)
                assert box.shape == tuple(ext), (r, peer, phase, box.shape, ext)
                locals_[r][0][lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]] = box
        assert all(not q for q in mailbox.values()), f"unmatched sends in phase {phase}"
    for r, (have, want) in enumerate(locals_):
        np.testing.assert_array_equal(have, want, err_msg=f"rank {r}")


@pytest.mark.parametrize("grid", [(1, 4), (2, 2), (1, 8)])
@pytest.mark.parametrize("halo,nsteps", [(1, 3), (2, 5), (3, 7), (4, 8)])
def test_wide_halo_scheme_model_on_bounded_grids(grid, halo, nsteps):
    """Numpy model of gt4mi_dist_lap5_f64_wide over a whole NON-periodic process grid (ranks at the physical
    boundary have neighbours on one side only -- a case the 1-GPU self-loop tests cannot reach): phase p of a
    cycle computes the local domain grown by halo-1-p cells towards every side that HAS a neighbour, the last
    phase exchanges `halo`-deep faces through the native message tables.  After n steps the assembled field
    must equal n steps on the undecomposed array (whose own ghost ring is a fixed boundary condition)."""
    from gt4py_amd.distributed import Decomposition, NativeHaloExchanger
    from oracle import ref_numpy as R

    H = halo
    gd = (12 * grid[0] + 0, 10 * grid[1], 2)
    n = grid[0] * grid[1]
    decs = [Decomposition(gd, grid, r, H) for r in range(n)]
    rng = np.random.default_rng(3)
    full = rng.uniform(-1, 1, (gd[0] + 2 * H, gd[1] + 2 * H, gd[2])) * 1e-3

    # reference: n steps on the global array; only its compute domain changes
    u, v = full.copy(), full.copy()
    core = (slice(H - 1, -(H - 1)) if H > 1 else slice(None),) * 2 + (slice(None),)
    for _ in range(nsteps):
        R.laplacian(u[core], v[core])  # origin (1,1,0) of the depth-1 view = compute domain of the global array
        u, v = v, u
    want = u

    def exchange(fields):
        tables = [NativeHaloExchanger.message_tables(d) for d in decs]
        for phase in (0, 1):
            mailbox = {}
            for r, (sends, _) in enumerate(tables):
                for peer, ph, lo, ext in sends:
                    if ph == phase:
                        mailbox.setdefault((r, peer), []).append(
                            fields[r][lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]].copy())
            for r, (_, recvs) in enumerate(tables):
                for peer, ph, lo, ext in recvs:
                    if ph == phase:
                        fields[r][lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]] = mailbox[(peer, r)].pop(0)

    a, b = [], []
    for d in decs:
        i0, j0 = d.offset[0], d.offset[1]
        li, lj, _ = d.local_domain
        a.append(full[i0:i0 + li + 2 * H, j0:j0 + lj + 2 * H].copy())
        b.append(full[i0:i0 + li + 2 * H, j0:j0 + lj + 2 * H].copy())
    # (the initial ghost cells are already right: they were cut from the global array)
    src, dst = a, b
    for step in range(nsteps):
        p = step % H
        ext = H - 1 - p
        for r, d in enumerate(decs):
            nb = d.neighbours
            li, lj, _ = d.local_domain
            lo_i = H - (ext if nb["W"] is not None else 0)
            hi_i = H + li + (ext if nb["E"] is not None else 0)
            lo_j = H - (ext if nb["S"] is not None else 0)
            hi_j = H + lj + (ext if nb["N"] is not None else 0)
            view = (slice(lo_i - 1, hi_i + 1), slice(lo_j - 1, hi_j + 1), slice(None))
            R.laplacian(src[r][view], dst[r][view])
        if ext == 0:
            exchange(dst)
        src, dst = dst, src
    for r, d in enumerate(decs):
        i0, j0 = d.offset[0], d.offset[1]
        li, lj, _ = d.local_domain
        np.testing.assert_array_equal(src[r][H:H + li, H:H + lj], want[H + i0:H + i0 + li, H + j0:H + j0 + lj],
                                      err_msg=f"rank {r}")


def ring_boxes(domain, outer, inner):
    """The boxes of csrc/lap5_ring.hip.h: (domain grown by outer[W, E, S, N]) minus (domain shrunk by inner[...]) as
    [(i0, i1, j0, j1)] relative to the compute-domain origin."""
    di, dj = domain[0], domain[1]
    (ow, oe, os_, on), (iw, ie, is_, in_) = outer, inner
    boxes = []
    if os_ + is_ > 0:
        boxes.append((-ow, di + oe, -os_, is_))
    if on + in_ > 0:
        boxes.append((-ow, di + oe, dj - in_, dj + on))
    if ow + iw > 0 and dj - is_ - in_ > 0:
        boxes.append((-ow, iw, is_, dj - in_))
    if oe + ie > 0 and dj - is_ - in_ > 0:
        boxes.append((di - ie, di + oe, is_, dj - in_))
    return boxes


@pytest.mark.parametrize("grid", [(1, 4), (2, 2), (4, 2), (1, 8)])
@pytest.mark.parametrize("halo,ncycles", [(1, 3), (2, 3), (3, 2)])
@pytest.mark.parametrize("single_phase", [False, True])
@pytest.mark.parametrize("ghosts_arrive", ["at once", "at the end of the cycle"])
def test_time_skewed_scheme_model_on_bounded_grids(grid, halo, ncycles, single_phase, ghosts_arrive):
    """Numpy model of gt4mi_dist_lap5_f64_skewed over a whole non-periodic process grid, with exactly the two buffers and
    the launch order of the C code: the bands of steps 1 .. H (from H - s points outside the domain to 2H - s inside it),
    pack of the result's faces, then the interiors of steps 1 .. H.  The ghost cells may land anywhere between the pack and
    the next cycle: both extremes are played.  After every cycle the assembled field must equal H more steps on the
    undecomposed array."""
    from gt4py_amd.distributed import Decomposition, NativeHaloExchanger
    from oracle import ref_numpy as R

    H = halo
    gd = (12 * grid[0], 10 * grid[1], 2)
    n = grid[0] * grid[1]
    decs = [Decomposition(gd, grid, r, H) for r in range(n)]
    rng = np.random.default_rng(5)
    full = rng.uniform(-1, 1, (gd[0] + 2 * H, gd[1] + 2 * H, gd[2])) * 1e-3
    u, v = full.copy(), full.copy()
    core = (slice(H - 1, -(H - 1)) if H > 1 else slice(None),) * 2 + (slice(None),)
    tables = [NativeHaloExchanger.message_tables(d, single_phase) for d in decs]

    def pack(fields):  # what the ranks send, copied at pack time
        mail = [{}, {}]
        for phase in (0, 1):
            for r, (sends, _) in enumerate(tables):
                for peer, ph, lo, ext in sends:
                    if ph == phase:
                        mail[phase].setdefault((r, peer), []).append((lo, ext))
        return mail

    def deliver(fields):  # a two-phase plan packs its second phase after the first has been unpacked
        for phase in (0, 1):
            mailbox = {}
            for r, (sends, _) in enumerate(tables):
                for peer, ph, lo, ext in sends:
                    if ph == phase:
                        mailbox.setdefault((r, peer), []).append(
                            fields[r][lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]].copy())
            for r, (_, recvs) in enumerate(tables):
                for peer, ph, lo, ext in recvs:
                    if ph == phase:
                        fields[r][lo[0]:lo[0] + ext[0], lo[1]:lo[1] + ext[1], lo[2]:lo[2] + ext[2]] = mailbox[(peer, r)].pop(0)

    def apply(src, dst, box):
        i0, i1, j0, j1 = box
        if i1 <= i0 or j1 <= j0:
            return
        view = (slice(H + i0 - 1, H + i1 + 1), slice(H + j0 - 1, H + j1 + 1), slice(None))
        R.laplacian(src[view], dst[view])

    a, b = [], []
    for d in decs:
        i0, j0 = d.offset[0], d.offset[1]
        li, lj, _ = d.local_domain
        a.append(full[i0:i0 + li + 2 * H, j0:j0 + lj + 2 * H].copy())
        # both buffers carry the fixed physical-boundary ring -- also where it crosses a neighbour's ghost rows, which the
        # grown bands read and no exchange of the OTHER buffer ever fills (same precondition as the _wide form)
        b.append(a[-1].copy())
    for cycle in range(ncycles):
        def src_of(st):
            return a if st % 2 == 1 else b

        def dst_of(st):
            return b if st % 2 == 1 else a

        sides = []
        for d in decs:
            nb = d.neighbours
            sides.append([nb[k] is not None for k in ("W", "E", "S", "N")])
        for st in range(1, H + 1):  # bands
            for r, d in enumerate(decs):
                outer = [H - st if f else 0 for f in sides[r]]
                inner = [2 * H - st if f else 0 for f in sides[r]]
                for box in ring_boxes(d.local_domain, outer, inner):
                    apply(src_of(st)[r], dst_of(st)[r], box)
        result = dst_of(H)
        snapshot = [f.copy() for f in result]  # the faces are final here: what travels is what is packed now
        if ghosts_arrive == "at once":
            deliver(result)
        for st in range(1, H + 1):  # interiors
            for r, d in enumerate(decs):
                li, lj, _ = d.local_domain
                w, e, s_, n_ = [2 * H - st if f else 0 for f in sides[r]]
                apply(src_of(st)[r], dst_of(st)[r], (w, li - e, s_, lj - n_))
        if ghosts_arrive != "at once":
            for r in range(n):  # the faces must not have changed since they were packed
                (sends, _) = tables[r]
                for peer, ph, lo, ext in sends:
                    if ph == 0:
                        sl = (slice(lo[0], lo[0] + ext[0]), slice(lo[1], lo[1] + ext[1]), slice(lo[2], lo[2] + ext[2]))
                        np.testing.assert_array_equal(result[r][sl], snapshot[r][sl])
            deliver(result)
        if H % 2 == 1:
            a, b = b, a  # the caller swaps the roles for an odd number of steps per cycle
        for _ in range(H):
            R.laplacian(u[core], v[core])
            u, v = v, u
        for r, d in enumerate(decs):
            i0, j0 = d.offset[0], d.offset[1]
            li, lj, _ = d.local_domain
            np.testing.assert_array_equal(a[r][H:H + li, H:H + lj], u[H + i0:H + i0 + li, H + j0:H + j0 + lj],
                                          err_msg=f"cycle {cycle}, rank {r}")


# ---- world_size-2 gloo runs of the two drivers an N > 1 GPU job uses ------------------------------------------
class HostField(np.ndarray):
    """numpy array that also answers ``.tensor`` (what the torch transport packs from): a CPU stand-in for DeviceArray."""

    @property
    def tensor(self):
        return torch.from_numpy(np.asarray(self))


def _wipe_neighbour_ghosts(blk, dec, value=np.nan):
    h, nb = dec.halo, dec.neighbours
    if nb["W"] is not None:
        blk[:h] = value
    if nb["E"] is not None:
        blk[-h:] = value
    if nb["S"] is not None:
        blk[:, :h] = value
    if nb["N"] is not None:
        blk[:, -h:] = value


def _worker_hdiff_driver(rank: int, world: int, tmpdir: str, grid):
    """The product's distributed drivers (overlapped_apply, sequential_apply, TunedApply's scratch rule) with the torch
    transport on gloo and the oracle's numpy backend as the stencil: BASELINE.json configs[4] in miniature."""
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates
    from gt4py_amd.distributed import TunedApply, overlapped_apply, sequential_apply

    hd = gtscript.stencil(backend="numpy", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64})
    rng = np.random.default_rng(2048)  # same stream on every rank -> same global fields
    gd, h = (22, 26, 3), 2
    glob = rng.uniform(-10, 10, (gd[0] + 2 * h, gd[1] + 2 * h, gd[2]))
    coeff = rng.uniform(0, 0.5, glob.shape)
    want = np.zeros_like(glob)
    R.hdiff(glob, want, coeff, domain=gd)
    dec = Decomposition(gd, grid, rank, h)
    origin = {n: dec.origin for n in ("in_field", "out_field", "coeff")}
    results = {}
    for form, apply in (("overlapped", overlapped_apply), ("sequential", sequential_apply)):
        blk = scatter_global(glob, dec).copy()
        _wipe_neighbour_ghosts(blk, dec)  # must come from the exchange
        args = {"in_field": blk.view(HostField), "coeff": scatter_global(coeff, dec).copy().view(HostField),
                "out_field": np.zeros_like(blk).view(HostField)}
        ex = HaloExchanger(dec, torch.float64, "cpu", packer=TorchSlicePacker())
        apply(hd, dec, origin, args, {"in_field": ex})
        results[form] = np.asarray(args["out_field"])[h:-h, h:-h].copy()
        assert np.array_equal(np.asarray(args["in_field"]), scatter_global(glob, dec))  # ghost cells incl. corners
    # calibration works on clones of the written fields only
    tuned = TunedApply(hd, dec, origin, {"in_field": ex})
    scratch = tuned._scratch_arguments(args)
    assert scratch["out_field"] is not args["out_field"] and scratch["in_field"] is args["in_field"]
    gathered = [None] * world
    dist.all_gather_object(gathered, (dec.global_slices(with_halo=False), results))
    if rank == 0:
        ok = True
        for form in ("overlapped", "sequential"):
            got = np.zeros_like(glob)
            for sl, res in gathered:
                got[sl] = res[form]
            ok = ok and np.array_equal(got[h:-h, h:-h], want[h:-h, h:-h])
        np.save(os.path.join(tmpdir, "ok.npy"), np.array([ok]))


@pytest.mark.multiprocess
@pytest.mark.parametrize("grid", [(1, 2), (2, 1)])
def test_gloo_world_size_2_hdiff_drivers(grid, tmp_path):
    run_ranks(_worker_hdiff_driver, 2, tmp_path, args=(grid,))
    assert np.load(_last_attempt(tmp_path) / "ok.npy")[0] == 1


@pytest.mark.multiprocess
def test_gloo_world_size_8_hdiff_on_the_4x2_process_grid(tmp_path):
    """BASELINE.json configs[4]'s process grid with EIGHT real processes (gloo, CPU): every rank scatters its share of one
    global field, exchanges faces and corners with its neighbours (interior ranks of the 4 x 2 grid have five of them), runs
    the drivers, and the assembled result equals the oracle on the undecomposed field."""
    assert choose_process_grid(8, (2048, 2048, 80)) == (4, 2)
    run_ranks(_worker_hdiff_driver, 8, tmp_path, args=((4, 2),))
    assert np.load(_last_attempt(tmp_path) / "ok.npy")[0] == 1


def _worker_wide_halo(rank: int, world: int, tmpdir: str, grid, halo: int, periodic):
    """The communication-avoiding time stepper (gt4mi_dist_lap5_f64_wide) as a model on real transport: phase p of a
    cycle computes the local domain grown by halo-1-p cells towards every side that has a neighbour, the last phase
    exchanges halo-deep faces.  n steps must equal n steps on the undecomposed array."""
    H, nsteps = halo, 2 * halo + 1
    gd = (8 * grid[0] + 3, 9 * grid[1] + 1, 2)
    rng = np.random.default_rng(77)
    full = rng.uniform(-1, 1, (gd[0] + 2 * H, gd[1] + 2 * H, gd[2])) * 1e-3

    def wrap(a):
        if periodic[0]:
            a[:H], a[-H:] = a[-2 * H:-H].copy(), a[H:2 * H].copy()
        if periodic[1]:
            a[:, :H], a[:, -H:] = a[:, -2 * H:-H].copy(), a[:, H:2 * H].copy()
        return a

    # reference: n steps on the global array (periodic axes re-wrapped before every step)
    u, v = wrap(full.copy()), full.copy()
    core = (slice(H - 1, -(H - 1)) if H > 1 else slice(None),) * 2 + (slice(None),)
    for _ in range(nsteps):
        R.laplacian(u[core], v[core])
        u, v = wrap(v), u
    dec = Decomposition(gd, grid, rank, H, periodic=periodic)
    ex = HaloExchanger(dec, torch.float64, "cpu", packer=TorchSlicePacker())
    src = scatter_global(wrap(full.copy()), dec).copy()
    dst = src.copy()
    nb = dec.neighbours
    li, lj, _ = dec.local_domain
    for step in range(nsteps):
        ext = H - 1 - step % H
        lo_i, hi_i = H - (ext if nb["W"] is not None else 0), H + li + (ext if nb["E"] is not None else 0)
        lo_j, hi_j = H - (ext if nb["S"] is not None else 0), H + lj + (ext if nb["N"] is not None else 0)
        view = (slice(lo_i - 1, hi_i + 1), slice(lo_j - 1, hi_j + 1), slice(None))
        R.laplacian(src[view], dst[view])
        if ext == 0:
            ex.exchange(torch.from_numpy(dst))
        src, dst = dst, src
    i0, j0 = dec.offset[0], dec.offset[1]
    ok = np.array_equal(src[H:H + li, H:H + lj], u[H + i0:H + i0 + li, H + j0:H + j0 + lj])
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        np.save(os.path.join(tmpdir, "ok.npy"), np.array([all(flags)]))


@pytest.mark.multiprocess
@pytest.mark.parametrize("grid,periodic", [((1, 2), (False, False)), ((2, 1), (False, False)), ((1, 2), (False, True)),
                                           ((2, 1), (True, False))])  # (gloo cannot send to the sending rank itself)
@pytest.mark.parametrize("halo", [1, 2, 3])
def test_gloo_world_size_2_wide_halo_time_stepping(grid, periodic, halo, tmp_path):
    run_ranks(_worker_wide_halo, 2, tmp_path, args=(grid, halo, periodic))
    assert np.load(_last_attempt(tmp_path) / "ok.npy")[0] == 1


@pytest.mark.parametrize("grid", [(4, 2), (1, 8), (3, 2)])
def test_tridiagonal_solve_decomposes_without_any_exchange(grid):
    """K is never split and the vertical solve reads no horizontal neighbour, so an IJ-decomposed
    tridiagonal solve is the local solve on every rank's share -- ghost depth 0, no message.  The drivers with an empty
    exchange table on every rank of the grid == the oracle on the undecomposed fields (``sup`` / ``rhs`` are updated in
    place on every share exactly as on the whole)."""
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates
    from gt4py_amd.distributed import overlapped_apply, sequential_apply

    tri = gtscript.stencil(backend="numpy", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64})
    rng = np.random.default_rng(7)
    gd = (13, 10, 9)
    glob = {"inf": rng.uniform(-1, 1, gd), "diag": rng.uniform(4, 5, gd), "sup": rng.uniform(-1, 1, gd), "rhs": rng.uniform(-10, 10, gd),
            "out": np.zeros(gd)}
    want = {k: v.copy() for k, v in glob.items()}
    R.tridiag(want["inf"], want["diag"], want["sup"], want["rhs"], want["out"])
    for apply in (overlapped_apply, sequential_apply):
        got = {k: np.full(gd, np.nan) for k in ("sup", "rhs", "out")}
        for rank in range(grid[0] * grid[1]):
            dec = Decomposition(gd, grid, rank, 0)
            assert dec.local_shape == dec.local_domain and dec.interior_and_strips()[1] == []
            sl = dec.global_slices(with_halo=False)
            args = {k: np.ascontiguousarray(v[sl]).view(HostField) for k, v in glob.items()}
            apply(tri, dec, {k: (0, 0, 0) for k in args}, args, {})
            for k in got:
                got[k][sl] = np.asarray(args[k])
        for k in got:
            assert np.array_equal(got[k], want[k]), (apply.__name__, k)


@pytest.mark.parametrize("grid,periodic", [((1, 8), (False, False)), ((4, 2), (False, False)), ((2, 2), (True, True)),
                                           ((1, 2), (True, True)), ((1, 1), (True, True)), ((2, 4), (True, False)),
                                           ((1, 4), (False, True)), ((2, 1), (True, True)), ((3, 2), (True, True))])
@pytest.mark.parametrize("halo", [1, 2])
@pytest.mark.parametrize("single_phase", [False, True])
def test_direct_transport_wiring_on_whole_process_grids(grid, periodic, halo, single_phase):
    """The direct transport (peer stores from the pack kernel) runs between two processes on the GPU box; how its messages are
    WIRED on any process grid can be checked here: ``NativeHaloExchanger.direct_wiring`` pairs every send with a receive of
    the same extent at its peer, no receive is fed twice or left out, ``recv_from`` is the inverse of ``send_to`` -- and
    delivering every face straight into the receive it is wired to (what the pack kernel does), phase by phase, leaves every
    rank's ghost cells with the values of the periodic / bounded global array, corners included."""
    from gt4py_amd.distributed import Decomposition, NativeHaloExchanger

    global_domain = (16 * grid[0], 12 * grid[1], 3)
    n = grid[0] * grid[1]
    decs = [Decomposition(global_domain, grid, r, halo, periodic=periodic) for r in range(n)]
    tables = [NativeHaloExchanger.message_tables(d, single_phase) for d in decs]

    def per_phase(table):
        return [[m for m in table if m[1] == p] for p in (0, 1)]

    msgs = [(per_phase(sends), per_phase(recvs)) for sends, recvs in tables]
    peers_of = {r: ([[m[0] for m in ph] for ph in msgs[r][0]], [[m[0] for m in ph] for ph in msgs[r][1]]) for r in range(n)}
    wiring = [NativeHaloExchanger.direct_wiring(r, peers_of) for r in range(n)]
    fed = set()
    for r in range(n):
        send_to, recv_from = wiring[r]
        for p in (0, 1):
            assert len(send_to[p]) == len(msgs[r][0][p]) and len(recv_from[p]) == len(msgs[r][1][p])
            for m, (q, j) in enumerate(send_to[p]):
                assert msgs[q][1][p][j][0] == r and msgs[q][1][p][j][3] == msgs[r][0][p][m][3]  # my peer expects ME, the same extent
                assert (q, p, j) not in fed
                fed.add((q, p, j))
                assert wiring[q][1][p][j] == (r, m)  # ... and knows that this send of mine fills that receive
    assert fed == {(q, p, j) for q in range(n) for p in (0, 1) for j in range(len(msgs[q][1][p]))}
    # deliver
    rng = np.random.default_rng(0)
    gi, gj, gk = global_domain
    full = rng.uniform(-1, 1, (gi + 2 * halo, gj + 2 * halo, gk))
    if periodic[0]:
        full[:halo], full[-halo:] = full[-2 * halo:-halo].copy(), full[halo:2 * halo].copy()
    if periodic[1]:
        full[:, :halo], full[:, -halo:] = full[:, -2 * halo:-halo].copy(), full[:, halo:2 * halo].copy()
    have, want = [], []
    for d in decs:
        i0, j0 = d.offset[0], d.offset[1]
        li, lj, _ = d.local_domain
        w = full[i0:i0 + li + 2 * halo, j0:j0 + lj + 2 * halo].copy()
        h = w.copy()
        nb = d.neighbours
        if nb["W"] is not None:
            h[:halo] = np.nan
        if nb["E"] is not None:
            h[-halo:] = np.nan
        if nb["S"] is not None:
            h[:, :halo] = np.nan
        if nb["N"] is not None:
            h[:, -halo:] = np.nan
        have.append(h)
        want.append(w)

    def box(lo, ext):
        return tuple(slice(a, a + e) for a, e in zip(lo, ext))

    for p in (0, 1):
        pool = {}  # (rank, receive index) -> what a pack kernel stored into that receive buffer
        for r in range(n):
            for m, (q, j) in enumerate(wiring[r][0][p]):
                pool[(q, j)] = have[r][box(msgs[r][0][p][m][2], msgs[r][0][p][m][3])].copy()
        for r in range(n):
            for j, (_, _, lo, ext) in enumerate(msgs[r][1][p]):
                have[r][box(lo, ext)] = pool[(r, j)]
    for r in range(n):
        assert np.array_equal(have[r], want[r]), r


# ---- the epoch-stamped self-check ------------------------------------------------------------------
class _HostField:
    """What FormCheck needs of a field: a `.tensor`."""

    def __init__(self, shape):
        import torch

        self.tensor = torch.zeros(shape, dtype=torch.float64)


def _wrap_ghost_cells(t, h):
    """A correct exchange of a periodic world of ONE rank, in place: every ghost cell takes the value of the point it wraps to."""
    t[:h], t[-h:] = t[-2 * h:-h].clone(), t[h:2 * h].clone()
    t[:, :h], t[:, -h:] = t[:, -2 * h:-h].clone(), t[:, h:2 * h].clone()


@pytest.mark.parametrize("halo", [1, 2])
def test_every_round_of_the_form_check_can_see_a_receive_buffer_read_too_early(halo):
    """`FormCheck` advances an EPOCH on every reset: F(global i, j, k) + 65 536 x epoch.  A transport whose receive side hands out
    what the PREVIOUS exchange left in its buffer -- the round-3 defect of the direct transport -- produced exactly the right
    values round after round as long as the probe never changed; only the first exchange of a plan (empty buffers) could show
    it.  Modelled here on the CPU: an "exchange" that delivers the ghost values of one round earlier passes the epoch-less check
    from the second round on and fails the epoch-stamped one in EVERY round, the verdict naming the stale values."""
    import torch

    from gt4py_amd.distributed import Decomposition
    from gt4py_amd.distributed.selfcheck import EPOCH_STEP, SENTINEL, FormCheck, coordinate_values

    dec = Decomposition((12, 10, 3), (1, 1), 0, halo=halo, periodic=(True, True))
    h = halo

    def local(a, b):  # a stand-in stencil that reads every ghost cell a 5-point (or wider) stencil reads
        t, o = a.tensor, b.tensor
        o.zero_()
        o[h:-h, h:-h] = t[h:-h, h:-h] + 0.5 * (t[:-2 * h, h:-h] + t[2 * h:, h:-h] + t[h:-h, :-2 * h] + t[h:-h, 2 * h:])

    chk = FormCheck(dec, (lambda: _HostField(dec.local_shape)), local)
    assert chk.ghost_cells_to_fill == (12 + 2 * h) * (10 + 2 * h) * 3 - 12 * 10 * 3

    def correct():
        _wrap_ghost_cells(chk.probe.tensor, h)
        local(chk.probe, chk.out)

    ok, found = chk.check(correct, rounds=4, loaded=1)
    assert ok and "4 epochs" in found and chk.epoch == 4 and chk.rounds_checked == 4
    # two epochs share no value: whatever a round leaves anywhere is wrong in the next one, everywhere
    own1, exp1 = coordinate_values(dec, epoch=1)
    own2, exp2 = coordinate_values(dec, epoch=2)
    assert torch.equal(exp2 - exp1, torch.full_like(exp1, EPOCH_STEP)) and not set(exp1.flatten().tolist()) & set(exp2.flatten().tolist())
    assert torch.equal(own2[h:-h, h:-h], exp2[h:-h, h:-h]) and int((own2 == SENTINEL).sum()) == chk.ghost_cells_to_fill

    # the stale transport: ghost cells of THIS round come from the buffer the previous round filled
    buffers = {"held": None}

    def stale():
        t = chk.probe.tensor
        fresh = t.clone()
        _wrap_ghost_cells(fresh, h)
        if buffers["held"] is None:  # the first exchange of the plan: nothing stale to hand out yet -- it waits and is correct
            t.copy_(fresh)
        else:
            mine = torch.zeros_like(t, dtype=torch.bool)
            mine[h:-h, h:-h] = True
            t.copy_(torch.where(mine, t, buffers["held"]))
        buffers["held"] = fresh
        local(chk.probe, chk.out)

    verdicts = [chk.check(stale, rounds=1, loaded=0) for _ in range(4)]
    assert verdicts[0][0] is True  # (only round 0 of the epoch-less check could ever see it -- and it did not have to)
    for ok, found in verdicts[1:]:
        assert not ok and f"{chk.ghost_cells_to_fill} cells of the exchanged field differ" in found
        assert f"[{chk.ghost_cells_to_fill} of them hold the previous epoch's value" in found and ", 0 points of the result" not in found
    # ... which the same check WITHOUT the epoch (the round-4 form) accepts: the stale payload is the right answer
    own0, exp0 = coordinate_values(dec)
    t = own0.clone()
    _wrap_ghost_cells(t, h)
    assert torch.equal(t, exp0)  # round n's buffer content == round n + 1's expectation when the probe never changes


def _worker_epoch_check(rank: int, world: int, tmpdir: str, grid, single_phase: bool):
    """Two real ranks (gloo): `FormCheck.check` over the torch transport -- consecutive epochs pass -- and over a transport that
    delivers every message ONE EXCHANGE LATE (the previous exchange's payload, as a receive buffer read before the neighbour's
    stores have landed would): every round after the plan's first fails on every rank, naming the stale values."""
    import torch
    import torch.distributed as dist

    from gt4py_amd.distributed import Decomposition, HaloExchanger
    from gt4py_amd.distributed.selfcheck import FormCheck

    h = 1
    dec = Decomposition((16, 12, 3), grid, rank, h, periodic=(grid[0] > 1, grid[1] > 1))  # (gloo cannot send to the rank itself)

    def local(a, b):
        t, o = a.tensor, b.tensor
        o.zero_()
        o[h:-h, h:-h] = -4.0 * t[h:-h, h:-h] + t[:-2 * h, h:-h] + t[2 * h:, h:-h] + t[h:-h, :-2 * h] + t[h:-h, 2 * h:]

    chk = FormCheck(dec, (lambda: _HostField(dec.local_shape)), local)
    ex = HaloExchanger(dec, torch.float64, "cpu", packer=TorchSlicePacker(), single_phase=single_phase)

    def correct():
        ex.exchange(chk.probe.tensor)
        local(chk.probe, chk.out)

    ok, found = chk.check(correct, rounds=3, loaded=1, before_run=dist.barrier)
    assert ok and chk.epoch == 3, found

    held = {"ghosts": None}

    def one_exchange_late():
        t = chk.probe.tensor
        fresh = t.clone()
        ex.exchange(fresh)  # what a correct transport delivers now ...
        mine = torch.zeros_like(t, dtype=torch.bool)
        mine[h:-h, h:-h] = True
        if held["ghosts"] is not None:  # ... is handed out only at the NEXT exchange
            t.copy_(torch.where(mine, t, held["ghosts"]))
        else:
            t.copy_(fresh)
        held["ghosts"] = fresh
        local(chk.probe, chk.out)

    verdicts = [list(chk.check(one_exchange_late, rounds=1, loaded=0, before_run=dist.barrier)) for _ in range(3)]
    return {"ghost_cells": chk.ghost_cells_to_fill, "verdicts": verdicts}


@pytest.mark.multiprocess
@pytest.mark.parametrize("grid,single_phase", [((1, 2), False), ((2, 1), True)])
def test_gloo_world_size_2_epoch_stamped_check_sees_a_late_transport_in_every_round(grid, single_phase, tmp_path):
    got = run_ranks(_worker_epoch_check, 2, tmp_path, args=(grid, single_phase))
    for r in (0, 1):
        n = got[r]["ghost_cells"]
        assert n > 0 and got[r]["verdicts"][0][0] is True  # (the plan's first exchange has nothing stale to hand out)
        for ok, found in got[r]["verdicts"][1:]:
            assert ok is False and f"[{n} of them hold the previous epoch's value" in found, (r, found)


def test_a_rank_with_a_pending_closing_round_issues_no_further_collectives():
    """close() of a failed direct plan offers the closing round from a helper thread for a bounded time; if the ranks
    do not meet, that thread stays inside a collective of the job's process group -- and this rank must not start another one next
    to it.  The group is marked poisoned; the agreement helpers refuse loudly instead of issuing a concurrent collective."""
    from gt4py_amd.distributed import calibrate, native

    class NeverCalled:
        class ReduceOp:
            MIN = MAX = None

        def all_reduce(self, *a, **k):  # pragma: no cover - the point of the test
            raise AssertionError("a collective was issued on a poisoned group")

        barrier = all_reduce

    ctx = {"distributed": True, "dist": NeverCalled(), "rank": 0, "collective_device": "cpu"}
    before = dict(native._POISON)
    try:
        native._POISON["why"] = None
        assert native.collectives_poisoned() is None
        native.poison_collectives("rank 0: the closing round of a failed direct plan did not complete within 20 s")
        native.poison_collectives("a second reason is not recorded")
        assert "closing round" in native.collectives_poisoned()
        with pytest.raises(native.CollectivesPoisoned, match="closing round"):
            calibrate._agree(ctx, 1)
        assert calibrate._agree({"distributed": False}, 1) == 1  # a single process has no group to poison
    finally:
        native._POISON.update(before)
