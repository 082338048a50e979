#!/usr/bin/env python3
"""Timeline of one distributed step from a `rocprofv3 --kernel-trace --output-format csv` run.

    python3 scripts/trace_timeline.py <kernel_trace.csv> [anchor substring = ring_kernel] [steps to print = 2]

Finds the last launches of the anchor kernel (the ring kernel ends a fused distributed apply), and prints every kernel
that started in the window of each of those steps: start offset (us) from the first kernel of the step, duration (us),
queue, name -- so that what overlaps what, and what the step's critical path is, can be read off."""
import csv
import sys


def short(name: str) -> str:
    name = name.replace("gt4mi::", "").replace("void ", "")
    cut = name.find("<")
    return (name[:cut] if cut > 0 else name)[:44]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "ring_kernel"
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if anchor in r[3]]
    if len(ends) < nsteps + 1:
        print(f"only {len(ends)} launches of '{anchor}' in {path}")
        return
    for n in range(nsteps, 0, -1):
        first, last = ends[-n - 1] + 1, ends[-n]
        t0 = rows[first][0]
        print(f"--- step ending with launch #{len(ends) - n} of {anchor}: {(rows[last][1] - t0) / 1e3:.1f} us from its first kernel "
              f"to the end of the ring")
        for s, e, q, name in rows[first:last + 1]:
            print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  q{q:>3}  {short(name)}")
    # mean duration per kernel over the whole trace
    acc = {}
    for s, e, q, name in rows:
        k = short(name)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print("--- mean duration per kernel over the trace (us)")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t / n:9.1f} us x {n:5d}  {k}")


if __name__ == "__main__":
    main()
