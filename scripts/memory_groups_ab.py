#!/usr/bin/env python3
"""Run on the GPU box: do MID-SIZE fields (the 339 MB fields of horizontal diffusion, the 285 MB fields of BASELINE configs[1]) gain from
living in different memory groups, like the 1.1-1.3 GB fields do (gt4py_amd/storage/placement.py)?

    python3 scripts/memory_groups_ab.py > profiles/r5_memory_groups_midsize_ab.log

The placer's wide search finds one raw block of 1.3 GB in each class; fields of the stencil's shape are laid out INSIDE those blocks (the
hip:mi300 layout: I-contiguous, rows of whole 128-byte lines), so that every combination of classes can be timed with the same blocks."""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402
from gt4py_amd.storage.device_array import DeviceArray  # noqa: E402


def field_in(block, slot, shape, dtype, fill):
    """A (I, J, K) tensor with I-contiguous strides and 128-byte aligned rows inside `block`, `slot` fields of that size from its start."""
    isz = np.dtype(dtype).itemsize
    row = -(-shape[0] * isz // 128) * 128 // isz
    n = row * shape[1] * shape[2]
    tdt = {np.dtype("float64"): torch.float64, np.dtype("float32"): torch.float32}[np.dtype(dtype)]
    flat = block[slot * n * isz:(slot + 1) * n * isz].view(tdt)
    t = torch.as_strided(flat, shape, (1, row, row * shape[1]))
    t.copy_(fill(shape).to(tdt))
    return DeviceArray(t, owner=block)


def time_ms(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def main() -> int:
    torch.cuda.set_device(0)
    placer = placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=0)
    nbytes = 1300 << 20
    blocks = {}
    for _ in range(4):
        block, cls = placer.place(nbytes)
        blocks.setdefault(cls, []).append(block)
        if len(blocks.get(0, [])) >= 2 and len(blocks.get(1, [])) >= 2:
            break
    print("classes found:", {k: len(v) for k, v in blocks.items()}, "report:", {k: v for k, v in placement.report().items() if k != "fields"})
    if len(blocks.get(0, [])) < 2 or len(blocks.get(1, [])) < 2:
        print("no second memory group within reach on this box: nothing to compare")
        return 0
    gen = torch.Generator(device="cuda").manual_seed(7)
    rnd = lambda shape: torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1  # noqa: E731
    # ---- horizontal diffusion, the configs[4] share: in, coeff read, out written (3 x 339 MB) -----------------------------------------
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64}, device_sync=False)
    dom, shape = (512, 1024, 80), (516, 1028, 80)
    for name, (ci, cc, co) in (("in coeff out all in class 0", (0, 0, 0)), ("coeff in class 1", (0, 1, 0)), ("out in class 1", (0, 0, 1)),
                               ("coeff and out in class 1", (0, 1, 1)), ("all in class 1", (1, 1, 1)), ("in coeff out all in class 0", (0, 0, 0))):
        use = {0: iter(blocks[0]), 1: iter(blocks[1])}
        slots = {0: 0, 1: 0}
        fields = {}
        for fname, cls, fill in (("in_field", ci, lambda s: 5.0 + rnd(s)), ("coeff", cc, lambda s: torch.full(s, 0.025, dtype=torch.float64, device="cuda")),
                                 ("out_field", co, rnd)):
            fields[fname] = field_in(blocks[cls][0], slots[cls], shape, np.float64, fill)  # (up to three fields side by side in the first block of the class)
            slots[cls] += 1
        frozen = hd.freeze(origin={k: (2, 2, 0) for k in fields}, domain=dom)
        ms = time_ms(lambda: frozen(**fields))
        print(f"hdiff fp64 512x1024x80   {name:32s} {ms:.4f} ms  {24.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f} of 8 TB/s")
    # ---- BASELINE configs[1]: Laplacian 512 x 512 x 128 (2 x 285 MB), four rotating pairs inside the blocks -----------------------------
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
    dom, shape = (512, 512, 128), (514, 514, 128)
    for name, (ci, co) in (("in and out in class 0", (0, 0)), ("out in class 1", (0, 1)), ("in and out in class 1", (1, 1)), ("in and out in class 0", (0, 0))):
        pairs = []
        for p in range(2):  # two rotating pairs per block pair (4 fields of 285 MB fit a 1.3 GB block)
            if ci == co:
                a = field_in(blocks[ci][p], 0, shape, np.float64, rnd)
                b = field_in(blocks[ci][p], 1, shape, np.float64, rnd)
                c = field_in(blocks[ci][p], 2, shape, np.float64, rnd)
                d = field_in(blocks[ci][p], 3, shape, np.float64, rnd)
                pairs += [(a, b), (c, d)]
            else:
                pairs += [(field_in(blocks[ci][p], 0, shape, np.float64, rnd), field_in(blocks[co][p], 0, shape, np.float64, rnd)),
                          (field_in(blocks[ci][p], 1, shape, np.float64, rnd), field_in(blocks[co][p], 1, shape, np.float64, rnd))]
        frozen = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=dom)
        state = {"i": 0}

        def step():
            a, b = pairs[state["i"] % len(pairs)]
            state["i"] += 1
            frozen(inp=a, out=b)

        ms = time_ms(step, 400)
        print(f"Laplacian fp64 512x512x128 (four rotating pairs)   {name:24s} {ms:.4f} ms  {16.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f} of 8 TB/s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
