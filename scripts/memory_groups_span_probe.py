#!/usr/bin/env python3
"""Run on the GPU box: how small may a block be and still be classified?  Blocks of known class (1.3 GB, found by the placer's wide
search) are probed against the reference on spans of 192 ... 1024 MiB, at several offsets inside the block."""
import ctypes
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gt4py_amd import _lib  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402


def main() -> int:
    torch.cuda.set_device(0)
    placer = placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=0)
    blocks = {}
    held = []
    for _ in range(4):
        block, cls = placer.place(1300 << 20)
        held.append(block)
        blocks.setdefault(cls, block)
    print("classes found:", sorted(blocks))
    if len(blocks) < 2:
        print("no second memory group within reach on this box")
        return 0
    lib = _lib.load()
    ref = placer.reference
    stream = torch.cuda.current_stream().cuda_stream

    def probe(a, b, nbytes):
        g = ctypes.c_double()
        _lib.check("probe", lib.gt4mi_memory_write_probe(a, b, nbytes, 6, stream, ctypes.byref(g)))
        return g.value

    for span_mib in (192, 256, 288, 336, 512, 1024):
        span = span_mib << 20
        row = []
        for cls in (0, 1):
            vals = []
            for off_mib in (0, 300, 600):
                if (off_mib << 20) + span > blocks[cls].numel():
                    continue
                vals.append(probe(blocks[cls].data_ptr() + (off_mib << 20), ref.data_ptr(), span))
            row.append(vals)
        print(f"span {span_mib:5d} MiB   class-0 block vs reference: {[round(v) for v in row[0]]}   class-1 block vs reference: {[round(v) for v in row[1]]} GB/s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
