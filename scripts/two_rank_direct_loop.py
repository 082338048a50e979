"""Run on the GPU box: the two-process direct-transport test in a loop, every rank's report kept.

    python3 scripts/two_rank_direct_loop.py [--runs 50] [--rounds 64] [--lib gt4py_amd/lib/libgt4py_amd_r3order.so] > log

Each run starts TWO fresh processes on the one device (tests/mp_util.py: a report per rank, no barrier-then-destroy race); each
rank runs ``tests/test_gpu_distributed._two_rank_direct_worker`` on the one-stream ("inline") schedule -- the parity cases
(``--cases full``: every case of the test; ``small``: one small Laplacian, so that a run is mostly self-check), three applies
each, ghost cells and results against the oracle -- and then ``--rounds`` rounds of the gather-free self-check
(``FormCheck.check``: a NEW EPOCH of the probe every round -- the payload the previous round left in the receive buffers is wrong
in every cell, so EVERY round can see a buffer read too early, not only a plan's first --, both ranks launch together behind a
barrier, one fused apply, every ghost cell and every ring point compared with exactly known values; ``--load``: which rounds run
next to an HBM-saturating background).  ``--lib`` selects another build of the library (``make r3order``: the receive side with
the load order of rounds 2-3) -- the A/B behind profiles/r5_two_rank_direct_loop.log; ``--fenced``: the transport's fenced mode.
A failing round does not end a run: the worker collects every verdict, the tally counts SENSITIVE ROUNDS and failed ones.
"""

from __future__ import annotations

import argparse
import collections
import os
import pathlib
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=64)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--grids", default="1x2,2x1")
    ap.add_argument("--load", choices=("all", "alternate", "none"), default="alternate")
    ap.add_argument("--cases", choices=("full", "small"), default="small")
    ap.add_argument("--fenced", action="store_true")
    args = ap.parse_args()
    if args.lib:
        os.environ["GT4PY_AMD_LIB"] = str((ROOT / args.lib).resolve())
    os.environ["GT4MI_TEST_VERDICT_ROUNDS"] = str(args.rounds)
    os.environ["GT4MI_TEST_VERDICT_LOAD"] = args.load
    os.environ["GT4MI_TEST_VERDICT_KEEP_GOING"] = "1"  # (the worker reports failed rounds instead of asserting on the first)
    if args.fenced:
        os.environ["GT4MI_TEST_DIRECT_FENCED"] = "1"
    cases = None if args.cases == "full" else [["lap5", 1, [66, 34, 2]]]
    import mp_util
    import test_gpu_distributed as T

    grids = [tuple(int(v) for v in g.split("x")) for g in args.grids.split(",")]
    print(f"library: {os.environ.get('GT4PY_AMD_LIB', 'gt4py_amd/lib/libgt4py_amd.so')}{' (FENCED mode)' if args.fenced else ''}; "
          f"{args.runs} runs x {len(grids)} grids x (bounded, periodic); {args.rounds} epoch-stamped self-check rounds per run and rank "
          f"(load: {args.load}; parity cases: {args.cases})", flush=True)
    tally = collections.Counter()
    t0 = time.time()
    for run in range(args.runs):
        for grid in grids:
            for periodic in ((False, False), (True, True)):
                with tempfile.TemporaryDirectory() as tmp:
                    results, problems = mp_util._attempt(T._two_rank_direct_worker, 2, tmp, (grid, periodic, "inline", cases), "gloo", 600.0, 120.0)
                what = f"run {run:3d} grid {grid[0]}x{grid[1]} periodic {int(periodic[0])}{int(periodic[1])}"
                if not problems:
                    bad = {r: sum(1 for v in rep["verdicts"] if not v[0]) for r, rep in results.items()}
                    stale = {r: sum(1 for v in rep["verdicts"] if not v[0] and "previous epoch" in v[1]) for r, rep in results.items()}
                    tally["sensitive rounds (per rank)"] += len(results[0]["verdicts"])
                    tally["failed rounds (rank 0 + rank 1)"] += sum(bad.values())
                    tally["... of them with the previous epoch's values in ghost cells"] += sum(stale.values())
                    tally["runs with a failed round" if sum(bad.values()) else "runs without a failed round"] += 1
                    first = next((v[1][:300] for rep in results.values() for v in rep["verdicts"] if not v[0]), "")
                    print(f"{what}: {'ok  ' if not sum(bad.values()) else 'BAD '} {len(results[0]['verdicts'])} rounds, timed_out "
                          f"{[results[r]['status']['timed_out'] for r in (0, 1)]}, failed rounds per rank {bad}"
                          + (f"; first: {first}" if first else ""), flush=True)
                    continue
                tally["FAILED"] += 1
                print(f"{what}: FAILED", flush=True)
                for text in problems:
                    lines = [ln for ln in text.splitlines() if ln.strip()]
                    head = lines[0]
                    # the assertion line (rank that found wrong values) or the last line of a collective's error (its peer)
                    key = next((ln.strip()[:600] for ln in reversed(lines) if "AssertionError" in ln or "Error" in ln), lines[-1][:600])
                    kind = "wrong values" if "AssertionError" in key else "collective ended (the peer had left)"
                    tally[f"  ranks with {kind}"] += 1
                    print(f"    {head}  {kind}: {key}", flush=True)
    print(f"---- {dict(tally)} in {time.time() - t0:.0f} s", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
