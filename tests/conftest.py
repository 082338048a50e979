import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "multiprocess: starts processes of its own (ranks, bench.py, torchrun): collected LAST")
    config.addinivalue_line("markers", "perf: holds a wall-clock comparison; the ordering is asserted only under GT4MI_PERF_ASSERT=1")


def pytest_sessionstart(session):
    if not hasattr(session.config, "workerinput"):  # (the controller, not an xdist worker)
        import retry_log

        retry_log.start_session()


def pytest_sessionfinish(session, exitstatus):
    """Under xdist the workers finish in any order and `test_zz_no_retry_fired` may run before a retry fires elsewhere: the
    controller looks once more when everything is over."""
    if hasattr(session.config, "workerinput"):
        return
    import retry_log

    fired = retry_log.fired()
    if fired and not retry_log.allowed() and session.exitstatus == 0:
        print(f"\n{len(fired)} test(s) passed only on their retry (gpurun_out/retries.jsonl): the run FAILS (GT4MI_ALLOW_RETRY=1 to accept)")
        session.exitstatus = 1


def _gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


# The suite runs with -x.  What must not be hidden by somebody else's failure comes first: the parity suites proper -- the
# full-size BASELINE configs, the golden fixtures, the kernel matrix, the generic executor, the reference's own suites (its
# ALL_BACKENDS matrix idea, /root/reference/tests/cartesian_tests/definitions.py:31-46) --, then the differential fuzzer, then
# the single-process tests of the distributed path, and LAST every test that starts processes of its own (a rendezvous, a second
# rank on the same device, a torchrun child: the tests most exposed to the box they run on).  Round 3's driver run stopped at
# such a test in position 683 of 2 482 and never reached one kernel parity test.
_FILE_ORDER = (
    "test_gpu_stencils", "test_gpu_golden_and_alias", "test_gpu_kernels", "test_gpu_generic", "test_reference_suites",
    "test_reference_codegen_cases", "test_reference_definitions", "test_reference_feature_cases", "test_math_builtins",
)
_LATE_FILES = ("test_fuzz_codegen", "test_gpu_distributed", "test_distributed", "test_bench_infrastructure")


def _bucket(item) -> int:
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name == "test_zz_retries":
        return 2000  # the very last: did any retry fire?
    if item.get_closest_marker("multiprocess") is not None:
        return 1000
    if name in _FILE_ORDER:
        return _FILE_ORDER.index(name)
    if name in _LATE_FILES:
        return 500 + _LATE_FILES.index(name)
    return 100  # everything else, in collection order, between the parity suites and the late files


def pytest_collection_modifyitems(config, items):
    items.sort(key=_bucket)  # (stable: the order inside a bucket is the collection order)
    # `-m gpu` on a machine without a GPU: skip instead of erroring deep inside HIP.
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
