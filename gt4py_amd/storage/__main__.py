"""``python -m gt4py_amd.storage [--fields 12] [--mib 1024]``: a map of the device's MEMORY GROUPS as this process sees them.

N separately allocated buffers; the diagonal is the write bandwidth of a buffer alone, entry (a, b) that of both written side by
side (``gt4mi_memory_write_probe``), GB/s; then the groups: a and b share one when the pair is no faster than 6.55 TB/s.  What the
storage allocator's placer (``placement.py``) works with -- a deployment check for a new box, driver or partition mode."""
import argparse
import ctypes
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m gt4py_amd.storage", description=__doc__)
    ap.add_argument("--fields", type=int, default=12)
    ap.add_argument("--mib", type=int, default=1024, help="size of each buffer (>= 192)")
    a = ap.parse_args(argv)
    import torch

    from .. import _lib
    from .placement import PAIR_GBS_OTHER_GROUP

    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (torch.cuda.is_available() is False)")
    lib, nbytes, n = _lib.load(), a.mib << 20, a.fields
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(n)]
    stream = torch.cuda.current_stream().cuda_stream

    def probe(x, y):
        g = ctypes.c_double()
        _lib.check("gt4mi_memory_write_probe", lib.gt4mi_memory_write_probe(x.data_ptr(), y.data_ptr() if y is not None else None, nbytes, 6, stream, ctypes.byref(g)))
        return g.value

    pair = [[0.0] * n for _ in range(n)]
    print(f"{n} buffers of {a.mib} MiB; GB/s written: [alone] on the diagonal, (a, b) together elsewhere")
    for i in range(n):
        row = []
        for j in range(n):
            if j < i:
                row.append("   .   ")
            elif j == i:
                row.append(f"[{probe(bufs[i], None):5.0f}]")
            else:
                pair[i][j] = pair[j][i] = probe(bufs[i], bufs[j])
                row.append(f" {pair[i][j]:5.0f} ")
        print(f"{i:3d} | " + " ".join(row), flush=True)
    group, k = [-1] * n, 0
    for i in range(n):
        if group[i] < 0:
            group[i] = k
            for j in range(i + 1, n):
                if group[j] < 0 and pair[i][j] < PAIR_GBS_OTHER_GROUP:
                    group[j] = k
            k += 1
    print("groups: " + " ".join(f"{i}:{chr(ord('A') + g)}" for i, g in enumerate(group)) + f"   ({k} group(s) among {n} x {a.mib} MiB)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
