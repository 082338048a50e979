"""The multi-process harness itself (tests/mp_util.py): every rank reports, a failure on ANY rank is shown with its own
traceback, nobody waits for a rank that has left, and a retry is loud."""

import warnings

import pytest

import mp_util

pytestmark = pytest.mark.multiprocess


def _rank_1_asserts(rank: int, world: int, out_dir: str, fail_rank: int):
    import torch
    import torch.distributed as dist

    t = torch.ones(1)
    dist.all_reduce(t)
    assert rank != fail_rank, f"rank {rank} was told to fail after the first collective"
    dist.barrier()  # (the surviving ranks sit here when the failing rank leaves)
    return {"sum": float(t[0])}


def test_a_failure_on_rank_1_is_reported_with_rank_1s_traceback(tmp_path):
    results, problems = mp_util._attempt(_rank_1_asserts, 2, str(tmp_path / "a"), (1,), "gloo", 120.0, 60.0)
    text = "\n".join(problems)
    assert "---- rank 1 raised ----" in text and "rank 1 was told to fail after the first collective" in text
    # rank 0 did not wait for ever: its barrier ended with an error, which is reported as what it is
    assert "---- rank 0 raised ----" in text and 0 not in results


def test_every_rank_reports_and_the_reports_come_back(tmp_path):
    assert mp_util.run_ranks(_rank_1_asserts, 2, tmp_path, args=(-1,)) == {0: {"sum": 2.0}, 1: {"sum": 2.0}}


def _fails_once(rank: int, world: int, out_dir: str, marker: str):
    import os

    import torch.distributed as dist

    dist.barrier()
    first = not os.path.exists(marker)
    dist.barrier()
    if rank == 0 and first:
        open(marker, "w").close()
    assert not (first and rank == 1), "the first attempt fails on rank 1"
    return rank


def test_a_pass_on_the_retry_is_never_silent_nor_free(tmp_path, monkeypatch):
    """... it is a warning in the summary AND an entry in the retry log, which fails the run's last test (tests/retry_log.py).
    (This test's own, deliberate retry goes to a log of its own.)"""
    import retry_log
    import test_zz_retries

    monkeypatch.setenv("GT4MI_RETRY_LOG", str(tmp_path / "retries.jsonl"))
    monkeypatch.delenv("GT4MI_ALLOW_RETRY", raising=False)
    assert retry_log.fired() == []
    test_zz_retries._check()  # nothing fired: passes
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert mp_util.run_ranks(_fails_once, 2, tmp_path, args=(str(tmp_path / "marker"),)) == {0: 0, 1: 1}
    text = "\n".join(str(w.message) for w in seen)
    assert "PASSED ONLY ON ITS RETRY" in text and "the first attempt fails on rank 1" in text
    fired = retry_log.fired()
    assert len(fired) == 1 and "test_a_pass_on_the_retry_is_never_silent_nor_free" in fired[0]["test"]
    assert "the first attempt fails on rank 1" in fired[0]["first_attempt"]
    with pytest.raises(pytest.fail.Exception, match="passed only on their retry"):
        test_zz_retries._check()
    monkeypatch.setenv("GT4MI_ALLOW_RETRY", "1")
    test_zz_retries._check()  # accepted explicitly


def test_the_last_failure_fails_the_test_with_every_ranks_traceback(tmp_path):
    with pytest.raises(pytest.fail.Exception) as info:
        mp_util.run_ranks(_rank_1_asserts, 2, tmp_path, args=(0,), attempts=1)
    assert "rank 0 was told to fail" in str(info.value) and "1 of 2 ranks failed" not in str(info.value)
