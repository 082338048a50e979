"""CPU: the parts of bench.py that must work on the first multi-GPU run nobody can rehearse -- the per-phase deadline
(a rank stuck in a collective exits with status 3 instead of hanging the node), the tie between a committed PMC traffic
figure and the kernel sources it was measured on, the pinned CPU-baseline child, and the HIP-event statistics contract."""

import json
import os
import pathlib
import subprocess
import sys
import time

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_watchdog_ends_a_stuck_process_with_status_3():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "dog = bench.Watchdog(5); dog.arm(0.5, 'a collective that never returns')\n"
            "time.sleep(60)\n") % str(ROOT)
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50)
    assert proc.returncode == 3 and time.perf_counter() - t0 < 30
    assert "rank 5" in proc.stderr and "a collective that never returns" in proc.stderr and "status 3" in proc.stderr


def test_watchdog_rearm_and_disarm():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "dog = bench.Watchdog(0)\n"
            "dog.arm(0.3, 'first'); dog.arm(30, 'second'); time.sleep(0.8)\n"  # re-arming cancels the first deadline
            "dog.arm(0.3, 'third'); dog.disarm(); time.sleep(0.8)\n"
            "print('alive')\n") % str(ROOT)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50)
    assert proc.returncode == 0 and "alive" in proc.stdout


def test_watchdog_prints_the_safe_line_and_exits_0_once_a_measurement_exists():
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench\n"
            "dog = bench.Watchdog(0)\n"
            "dog.safe = lambda reason: os.write(1, ('SAFE ' + reason + chr(10)).encode())\n"
            "dog.arm(0.5, 'calibration of the overlapped forms')\n"
            "time.sleep(60)\n") % str(ROOT)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50)
    assert proc.returncode == 0
    assert proc.stdout.startswith("SAFE phase 'calibration of the overlapped forms' ran past its")
    assert "printed instead" in proc.stderr and "status 3" not in proc.stderr
    # GT4MI_BENCH_DEADLINE_SCALE (tests only) shortens every deadline
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "dog = bench.Watchdog(0); dog.arm(50, 'x'); time.sleep(60)\n") % str(ROOT)
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50,
                          env=dict(os.environ, GT4MI_BENCH_DEADLINE_SCALE="0.01"))
    assert proc.returncode == 3 and time.perf_counter() - t0 < 30


def test_committed_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    real = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())["lap5_f64_512"]
    assert set(real) >= {"bytes_per_launch", "kernel", "git_sha", "kernel_source_sha", "source"}
    assert any("lap5_strip_kernel" in k for k in real["kernel"])
    # the committed figure is reported exactly when it belongs to the kernel sources in the tree (after touching
    # gt4py_amd/csrc/{lap5,lane_shift,common}.hip.h or the Makefile: scripts/profile_bench.sh on the GPU)
    value, why = bench._committed_traffic("lap5_f64_512")
    if real["kernel_source_sha"] == bench.kernel_source_hash("lap5_f64_512"):
        assert value == real["bytes_per_launch"] and 1.0 < value / (16.0 * 512**3) < 1.2 and "git" in why
    else:
        assert value is None and "other kernel sources" in why
    # a stale or hash-less record is not reported
    fake_root = tmp_path / "repo"
    (fake_root / "profiles").mkdir(parents=True)
    for rel in bench.KERNEL_SOURCES["lap5_f64_512"]:
        (fake_root / rel).parent.mkdir(parents=True, exist_ok=True)
        (fake_root / rel).write_bytes((ROOT / rel).read_bytes())
    monkeypatch.setattr(bench, "ROOT", fake_root)
    current = dict(real, kernel_source_sha=bench.kernel_source_hash("lap5_f64_512"))  # a record of exactly these sources
    (fake_root / "profiles" / "hbm_traffic.json").write_text(json.dumps({"lap5_f64_512": current}))
    assert bench._committed_traffic("lap5_f64_512")[0] == real["bytes_per_launch"]
    (fake_root / bench.KERNEL_SOURCES["lap5_f64_512"][0]).write_text("// a different kernel\n")
    value, why = bench._committed_traffic("lap5_f64_512")
    assert value is None and "other kernel sources" in why
    (fake_root / "profiles" / "hbm_traffic.json").write_text(json.dumps({"lap5_f64_512": 123}))
    assert bench._committed_traffic("lap5_f64_512") == (None, "committed measurement carries no kernel-source hash")


def test_cpu_baseline_child_reports_pinned_threads_and_spread():
    env = dict(os.environ, OMP_NUM_THREADS="2", OMP_PROC_BIND="close", OMP_PLACES="cores")
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-child", "1.0"], env=env, capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-500:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["kind"] == "port" and line["unit"] == "GLUPS" and line["cores"] == 2 and line["value"] > 0
    assert {"sample", "spread_pct", "batch_glups_min_max", "host", "gb_per_s"} <= set(line)
    assert "512x512x512" in line["sample"] and "pinned" in line["sample"]
    numpy_line = line["numpy_single_thread"]  # the numpy restatement, single thread, beside it
    assert numpy_line["cores"] == 1 and 0 < numpy_line["value"] < line["value"] * 4 and "512x512x128" in numpy_line["sample"]
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)


def test_bench_module_has_no_gpu_side_effects_on_import():
    """`import bench` (scripts/summarize_profile.py does it on the GPU box for the source hash) must not touch torch."""
    proc = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print('torch' in sys.modules)" % str(ROOT)],
                          capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0 and proc.stdout.strip() == "False"


def _proof_worker(rank: int, world: int, port: int, tmpdir: str):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = {"world": world, "rank": rank, "distributed": True, "dist": dist, "device": "cpu"}
        # what NativeComm.info() returns on an N-rank communicator (RCCL itself cannot run here)
        proof = bench.gather_rank_proof(ctx, {"rank": rank, "device": rank, "nranks": world})
        everyone_ok = bench._agree(ctx, 1)
        one_failed = bench._agree(ctx, 0 if rank == 1 else 1)
        ms = bench._slowest_rank_ms(ctx, (lambda: time.sleep(0.002 * (rank + 1))), calls=3, warm=1)  # rank 1 is the slow one
        # a calibration candidate that fails on ONE rank is dropped on every rank without anyone waiting in a barrier
        def good():
            return (lambda: time.sleep(0.001)), (lambda: None)

        def bad_on_rank_1():
            if rank == 1:
                raise RuntimeError("this device refuses the option")
            return (lambda: None), (lambda: None)

        candidates = [bench.measure_candidate(ctx, good, 3, warm=1), bench.measure_candidate(ctx, bad_on_rank_1, 3, warm=1),
                      bench.measure_candidate(ctx, good, 3, warm=1)]
        keys = bench.decomposed_line_keys(proof, False, world, {"timestep_glups": 1.0})
        fallen = bench.decomposed_line_keys(None, True, world, None)
        with open(os.path.join(tmpdir, f"rank{rank}.json"), "w") as fh:
            json.dump({"keys": keys, "fallen": fallen, "ok": [everyone_ok, one_failed], "ms": ms, "candidates": candidates}, fh)
    finally:
        dist.destroy_process_group()


def test_the_n_gpu_line_says_how_many_ranks_rccl_saw_and_whether_the_transport_fell_back(tmp_path, capfd):
    """the first SCALE line must be interpretable -- `rccl_nranks` / `rank_devices` gathered from
    every rank, `transport_fallback` as a top-level key (plus a banner on stderr), the time steppers under `extra`.  The
    collective helpers run here on gloo, world_size 2."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_proof_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        got = json.loads((tmp_path / f"rank{rank}.json").read_text())
        keys = got["keys"]
        assert keys["rccl_nranks"] == 2 and keys["rank_devices"] == [[0, 0], [1, 1]] and keys["rccl_matches_n_gpus"] is True
        assert keys["transport_fallback"] is False and keys["extra"] == {"timestep_glups": 1.0}
        assert got["fallen"] == {"rccl_nranks": None, "rank_devices": None, "rccl_matches_n_gpus": False, "transport_fallback": True}
        assert got["ok"] == [1, 0]  # one failing rank makes every rank fall back together
        assert got["ms"] >= 3.5  # the slowest rank's time on every rank
        c = got["candidates"]
        assert c[0] is not None and c[0] >= 1.0 and c[1] is None and c[2] is not None
    bench.transport_fallback_banner(0, "testing")
    err = capfd.readouterr().err
    assert "NATIVE RCCL TRANSPORT UNAVAILABLE (testing)" in err and "NOT those of the product path" in err
    bench.transport_fallback_banner(1, "testing")  # only rank 0 shouts
    assert capfd.readouterr().err == ""
