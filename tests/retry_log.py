"""A retry is never silent AND never free: every first-attempt failure that a retry wrapper absorbed is
appended to ``gpurun_out/retries.jsonl`` (test id, what the first attempt said), and the always-collected last test
``tests/test_zz_retries.py::test_zz_no_retry_fired*`` -- plus the session-finish hook in conftest.py, for runs under xdist, whose
workers finish in any order -- FAILS the run when that file is non-empty, unless ``GT4MI_ALLOW_RETRY=1``.  A green driver run
therefore means: no retry fired.
"""

from __future__ import annotations

import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def path() -> str:
    return os.environ.get("GT4MI_RETRY_LOG") or os.path.join(ROOT, "gpurun_out", "retries.jsonl")


def start_session() -> None:
    """(the controller process of a pytest run, once) the previous run's log is kept as ``.prev``, this run starts empty."""
    p = path()
    os.makedirs(os.path.dirname(p), exist_ok=True)
    if os.path.exists(p):
        os.replace(p, p + ".prev")


def record(test_id: str, first_failure: str) -> None:
    p = path()
    os.makedirs(os.path.dirname(p), exist_ok=True)
    with open(p, "a") as fh:  # (O_APPEND: whole lines from concurrent workers do not interleave at these sizes)
        fh.write(json.dumps({"test": test_id, "time": time.time(), "pid": os.getpid(), "first_attempt": first_failure[-6000:]}) + "\n")


def fired() -> list:
    try:
        with open(path()) as fh:
            return [json.loads(ln) for ln in fh if ln.strip()]
    except FileNotFoundError:
        return []


def allowed() -> bool:
    return os.environ.get("GT4MI_ALLOW_RETRY", "0") == "1"


def current_test_id(default: str) -> str:
    return os.environ.get("PYTEST_CURRENT_TEST", default).split(" (")[0]
