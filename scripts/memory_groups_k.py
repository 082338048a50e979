#!/usr/bin/env python3
"""Run on the GPU box: HOW MANY memory groups are within reach, and does a kernel with three written streams (the tridiagonal solve:
sup, rhs, out) gain from THREE groups over two?  (The placer deals over two classes: the reference's group and "any other".)

N raw blocks of the solve's field size, from the fifth on each behind an untouched spacer (the groups change along the physical
address space); the full pair matrix of `gt4mi_memory_write_probe`; groups = connected "written side by side no faster than 6.55
TB/s"; then the solve (BASELINE configs[3], pristine operands every launch) with its five fields on chosen blocks.

    python3 scripts/memory_groups_k.py [--blocks 28] [--spacer-gib 6] > profiles/r5_memory_groups_k.log"""
import argparse
import ctypes
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd import _lib  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=28)
    ap.add_argument("--spacer-gib", type=float, default=6.0)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    lib = _lib.load()
    placer = placement.device_placer()
    dom = (1024, 1024, 160)
    first = gt_storage.empty(dom, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0))  # (also makes the placer's reference)
    nbytes = int(first._owner.numel())
    del first
    torch.cuda.empty_cache()

    def probe(a, b, span):
        gbs = ctypes.c_double()
        _lib.check("gt4mi_memory_write_probe", lib.gt4mi_memory_write_probe(a, b or None, int(span), 6, torch.cuda.current_stream().cuda_stream, ctypes.byref(gbs)))
        return float(gbs.value)

    blocks, spacers = [], []
    for i in range(args.blocks):
        if i >= 4 and args.spacer_gib > 0:
            if torch.cuda.mem_get_info()[0] < int(args.spacer_gib * 2 ** 30) + nbytes + (24 << 30):
                break
            spacers.append(torch.empty((int(args.spacer_gib * 2 ** 30),), dtype=torch.uint8, device="cuda"))
        blocks.append(torch.empty((nbytes,), dtype=torch.uint8, device="cuda"))
    n = len(blocks)
    print(f"{n} blocks of {nbytes / 2 ** 20:.0f} MiB, {len(spacers)} untouched spacers of {args.spacer_gib} GiB; free now {torch.cuda.mem_get_info()[0] / 2 ** 30:.0f} GiB")
    pair = np.zeros((n, n))
    for a in range(n):
        for b in range(a, n):
            pair[a, b] = pair[b, a] = probe(blocks[a].data_ptr(), 0 if a == b else blocks[b].data_ptr(), nbytes)
    group = [-1] * n
    leaders = []
    for a in range(n):
        for g, lead in enumerate(leaders):
            if pair[a, lead] < placer.threshold:
                group[a] = g
                break
        else:
            group[a] = len(leaders)
            leaders.append(a)
    print("GB/s written: [alone] on the diagonal, (a, b) side by side elsewhere")
    for a in range(n):
        print(f"{a:3d} {chr(65 + group[a])} | " + " ".join((f"[{pair[a, b]:4.0f}]" if a == b else f" {pair[a, b]:4.0f} ") if b >= a else "   .  " for b in range(n)))
    print("groups: " + " ".join(f"{a}:{chr(65 + group[a])}" for a in range(n)) + f"   ({len(leaders)} group(s))")
    wrong = [(a, b) for a in range(n) for b in range(a + 1, n) if (pair[a, b] < placer.threshold) != (group[a] == group[b])]
    print(f"pairs whose rate contradicts the grouping by leaders: {len(wrong)} of {n * (n - 1) // 2} {wrong[:12]}")
    same = [pair[a, b] for a in range(n) for b in range(a + 1, n) if group[a] == group[b]]
    other = [pair[a, b] for a in range(n) for b in range(a + 1, n) if group[a] != group[b]]
    if same and other:
        print(f"same group {min(same):.0f} .. {max(same):.0f} GB/s, different groups {min(other):.0f} .. {max(other):.0f} GB/s")
    sys.stdout.flush()
    del spacers
    members = {g: [a for a in range(n) if group[a] == g] for g in range(len(leaders))}

    # ---- the solve on chosen blocks: a block is handed to the storage layer through the placer's parking lot ---------------------------
    tri = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64}, device_sync=False)
    gen = torch.Generator(device="cuda").manual_seed(7)
    ranges = {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (0, 0)}
    host = {k: (torch.rand(dom, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo) for k, (lo, hi) in ranges.items()}

    def solve(groups_of_fields, rounds=8):
        """groups_of_fields: a group letter index per field (inf diag sup rhs out); None when a group has too few blocks."""
        taken, use = {}, []
        for g in groups_of_fields:
            idx = taken.get(g, 0)
            if g not in members or idx >= len(members[g]):
                return None
            use.append(members[g][idx])
            taken[g] = idx + 1
        fields = {}
        for name, b in zip(ranges, use):
            placer.parked[(0, nbytes)] = [blocks[b]]
            with placement.want(0):
                fields[name] = gt_storage.empty(dom, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0))
            assert fields[name]._owner.data_ptr() == blocks[b].data_ptr()
        frozen = tri.freeze(origin={k: (0, 0, 0) for k in fields}, domain=dom)
        times = []
        for _ in range(rounds):
            for k in ranges:
                fields[k].tensor.copy_(host[k])  # pristine operands every launch (the solve overwrites sup and rhs)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            frozen(**fields)
            b.record()
            b.synchronize()
            times.append(a.elapsed_time(b))
        ms = sorted(times)[len(times) // 2]
        return ms, use, fields["out"].tensor.clone()

    cases = [("all five in one group", (0, 0, 0, 0, 0)), ("two groups, dealt in order (the placer's deal)", (0, 1, 0, 1, 0)),
             ("two groups, writers 2 + 1", (0, 1, 1, 0, 0)), ("two groups, all three writers in one", (0, 0, 1, 1, 1)),
             ("three groups, one writer each, readers in A and B", (0, 1, 2, 0, 1)), ("three groups, one writer each, readers with the out's", (2, 2, 0, 1, 2)),
             ("three groups dealt in order", (0, 1, 2, 0, 1)), ("three groups: readers in C, writers A B A", (2, 2, 0, 1, 0)),
             ("four groups", (0, 1, 2, 3, 0)), ("five groups", (0, 1, 2, 3, 4)),
             ("all five in one group (again)", (0, 0, 0, 0, 0)), ("two groups, dealt in order (again)", (0, 1, 0, 1, 0))]
    # groups ordered by size so that "group 0" is the one with most blocks
    order = sorted(members, key=lambda g: -len(members[g]))
    members = {i: members[g] for i, g in enumerate(order)}
    print("groups by size: " + "  ".join(f"{chr(65 + i)}' = {chr(65 + g)} ({len(members[i])} blocks)" for i, g in enumerate(order)))
    ref_out = None
    for what, gs in cases:
        got = solve(gs)
        if got is None:
            print(f"tridiagonal 1024x1024x160  {what:58s} groups {gs}: (not enough groups / blocks on this box)")
            continue
        ms, use, out = got
        if ref_out is None:
            ref_out = out
        same_bits = bool(torch.equal(out, ref_out))
        print(f"tridiagonal 1024x1024x160  {what:58s} groups {''.join(chr(65 + g) + chr(39) for g in gs)} blocks {use}: {ms:.4f} ms  "
              f"{56.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f} of 8 TB/s   bit-identical: {same_bits}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
