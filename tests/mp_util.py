"""Multi-process tests: one harness for every test that starts ranks of its own.

What it guarantees (round 3's driver run lost 1 799 tests to a two-rank test that could not say which rank had failed):

* EVERY rank reports: ``rank{r}.ok`` (the worker's return value as JSON) or ``rank{r}.err`` (the full traceback) in the test's
  temporary directory; the parent asserts on all of them and prints every traceback, not the first dead rank's;
* no rank tears the process group down while another is still inside a collective: a rank that is done writes its file and
  then WAITS for the files of the others before it leaves (a file handshake, no ``barrier`` + ``destroy_process_group`` race); a
  rank that failed leaves at once, so that the others' collectives end with an error instead of waiting for it;
* the rendezvous is a ``FileStore`` inside the temporary directory -- no port that was free a moment ago;
* one retry, and a retry is never silent NOR free: the failed attempt's tracebacks go into a warning that the pytest summary
  prints AND into gpurun_out/retries.jsonl, which fails the run's last test (tests/retry_log.py, tests/test_zz_retries.py).
"""

from __future__ import annotations

import json
import os
import time
import traceback
import warnings

HANDSHAKE_SECONDS = 120.0


def _reports(out_dir: str, world: int):
    done = {}
    for r in range(world):
        for kind in ("ok", "err"):
            path = os.path.join(out_dir, f"rank{r}.{kind}")
            if os.path.exists(path):
                done[r] = (kind, path)
    return done


def _entry(rank: int, world: int, out_dir: str, worker, args, backend: str, collective_seconds: float):
    import datetime

    import torch.distributed as dist

    status = 1
    try:
        store = dist.FileStore(os.path.join(out_dir, "rendezvous"), world)
        dist.init_process_group(backend, store=store, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=collective_seconds))
        result = worker(rank, world, out_dir, *args)
        tmp = os.path.join(out_dir, f"rank{rank}.ok.tmp")
        with open(tmp, "w") as fh:
            json.dump(result, fh)
        os.replace(tmp, os.path.join(out_dir, f"rank{rank}.ok"))
        # stay (process group, device context, exported memory and all) until every rank has reported
        t0 = time.monotonic()
        while len(_reports(out_dir, world)) < world and time.monotonic() - t0 < HANDSHAKE_SECONDS:
            time.sleep(0.02)
        status = 0
    except BaseException:  # noqa: BLE001 - whatever it was, it goes into the rank's report
        tmp = os.path.join(out_dir, f"rank{rank}.err.tmp")
        with open(tmp, "w") as fh:
            fh.write(traceback.format_exc())
        os.replace(tmp, os.path.join(out_dir, f"rank{rank}.err"))
    finally:
        # no destroy_process_group: the peers may be anywhere; the sockets close with the process
        os._exit(status)


def _attempt(worker, world: int, out_dir: str, args, backend: str, timeout: float, collective_seconds: float):
    import torch.multiprocessing as mp

    os.makedirs(out_dir, exist_ok=True)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_entry, args=(r, world, out_dir, worker, args, backend, collective_seconds), daemon=False)
             for r in range(world)]
    for p in procs:
        p.start()
    deadline = time.monotonic() + timeout
    for p in procs:
        p.join(max(0.0, deadline - time.monotonic()))
    hung = [r for r, p in enumerate(procs) if p.is_alive()]
    for r in hung:  # exactly the processes started above
        procs[r].kill()
        procs[r].join(10)
    reports, results, problems = _reports(out_dir, world), {}, []
    for r in range(world):
        if r in reports and reports[r][0] == "ok":
            with open(reports[r][1]) as fh:
                results[r] = json.load(fh)
        elif r in reports:
            with open(reports[r][1]) as fh:
                problems.append(f"---- rank {r} raised ----\n{fh.read()}")
        else:
            how = f"was killed after {timeout:.0f} s" if r in hung else f"ended with exit code {procs[r].exitcode}"
            problems.append(f"---- rank {r} left no report: it {how} ----")
    return results, problems


def run_ranks(worker, world: int, tmp_path, args=(), backend: str = "gloo", timeout: float = 600.0, attempts: int = 2,
              collective_seconds: float = 180.0):
    """Run ``worker(rank, world, out_dir, *args)`` on ``world`` fresh processes joined in a ``backend`` process group; returns
    ``{rank: what the worker returned}`` (JSON-able) or fails the test with every rank's traceback."""
    import pytest

    first = None
    for attempt in range(attempts):
        results, problems = _attempt(worker, world, os.path.join(str(tmp_path), f"attempt{attempt}"), tuple(args), backend, timeout,
                                     collective_seconds)
        if not problems:
            if first is not None:
                import retry_log

                retry_log.record(retry_log.current_test_id(f"{getattr(worker, '__name__', worker)}{tuple(args)!r}"), first)
                warnings.warn(f"{getattr(worker, '__name__', worker)}{tuple(args)!r}: PASSED ONLY ON ITS RETRY; the first attempt:\n{first}")
            return results
        if first is None:
            first = "\n".join(problems)
    pytest.fail(f"{getattr(worker, '__name__', worker)}: {len(problems)} of {world} ranks failed (attempt {attempts} of {attempts})\n"
                + "\n".join(problems) + ("\n==== first attempt ====\n" + first if attempts > 1 else ""), pytrace=False)
