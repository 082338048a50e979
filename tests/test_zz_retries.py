"""Collected LAST (tests/conftest.py): did any retry wrapper absorb a first-attempt failure in this run?  See tests/retry_log.py."""

import pytest

import retry_log


def _check():
    fired = retry_log.fired()
    if fired and not retry_log.allowed():
        lines = [f"{e['test']}:\n{e['first_attempt'][-1500:]}" for e in fired]
        pytest.fail(f"{len(fired)} test(s) passed only on their retry (gpurun_out/retries.jsonl; GT4MI_ALLOW_RETRY=1 to accept):\n"
                    + "\n----\n".join(lines), pytrace=False)


def test_zz_no_retry_fired():
    _check()


@pytest.mark.gpu
def test_zz_no_retry_fired_gpu():
    _check()
