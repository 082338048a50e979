"""CPU: the parts of bench.py that must work on the first multi-GPU run nobody can rehearse -- the per-phase deadline
(a rank stuck in a collective exits with status 3 instead of hanging the node), the tie between a committed PMC traffic
figure and the kernel sources it was measured on, the pinned CPU-baseline child, and the HIP-event statistics contract."""

import json
import os
import pathlib
import subprocess
import sys
import time
import zlib

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


def _proof_worker(rank: int, world: int, tmpdir: str):
    import torch.distributed as dist

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


@pytest.mark.multiprocess
def test_the_n_gpu_line_says_how_many_ranks_rccl_saw_and_whether_the_transport_fell_back(tmp_path, capfd):
    """the first SCALE line must be interpretable -- `rccl_nranks` / `rank_devices` gathered from
    every rank, `transport_fallback` as a top-level key (plus a banner on stderr), the time steppers under `extra`.  The
    collective helpers run here on gloo, world_size 2."""
    from mp_util import run_ranks

    run_ranks(_proof_worker, 2, tmp_path)
    for rank in (0, 1):
        got = json.loads((sorted(tmp_path.glob("attempt*"))[-1] / f"rank{rank}.json").read_text())
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


# ---- the bounded, ordered calibration of an N > 1 run ---------------------------------------------
def test_the_calibration_order_puts_the_north_stars_rccl_forms_first():
    grids = [(4, 2), (2, 4), (1, 8), (8, 1)]
    first, refine, direct = bench.lap_calibration_order((4, 2), grids, (False, True), ("rccl", "direct"))
    # an overlapped RCCL headline on the grid choose_process_grid returns after at most four candidates
    assert first[:4] == [((4, 2), False, "swap", 0, "rccl"), ((4, 2), False, "join", 0, "rccl"),
                         ((4, 2), True, "swap", 0, "rccl"), ((4, 2), True, "join", 0, "rccl")]
    assert all(c[4] == "rccl" for c in first) and {c[0] for c in first[4:]} == {(2, 4), (1, 8), (8, 1)}
    assert all(c[0] == (2, 4) and c[4] == "rccl" for c in refine(((2, 4), True, "swap", 0, "rccl")))
    d = direct(((2, 4), True, "swap", 0, "rccl"))
    assert all(c[4] == "direct" for c in d) and d[0] == ((2, 4), False, "inline", 0, "direct") and d[1][:3] == ((2, 4), True, "inline")
    keys = [bench.lap_key(c) for c in first + refine(first[0]) + d]
    assert [bench.lap_key(bench.lap_candidate_of(k)) for k in keys] == keys
    assert bench.lap_calibration_order((4, 2), grids, (False,), ("rccl",))[2](None) == []  # no direct transport asked for
    h_first, h_refine, h_direct = bench.hdiff_calibration_order(("join", "chain", "inline"), (2, 16, 32), ("rccl", "direct"))
    assert h_first == ["fused_two_phase_chain_wg2_edge16", "fused_two_phase_join_wg2_edge16", "fused_single_phase_chain_wg2_edge16",
                       "fused_single_phase_join_wg2_edge16"]
    assert all("single_phase" in n for n in h_refine("fused_single_phase_chain_wg2_edge16")[:-1]) and h_refine(h_first[0])[-2:] == [
        "sequential_two_phase", "sequential_single_phase"]
    assert h_direct(h_first[0])[:2] == ["fused_two_phase_chain_wg3_edge16_direct", "fused_single_phase_chain_wg3_edge16_direct"]
    assert all(n.endswith("_direct") for n in h_direct(None)) and "fused_two_phase_inline_wg0_edge32_direct" in h_direct(None)


def _calibration_worker(rank: int, world: int, tmpdir: str, rccl_seconds: float, canary_says, join_fails_when="unfenced", direct_seconds=1.0):
    import torch.distributed as dist

    ctx = {"world": world, "rank": rank, "distributed": True, "dist": dist, "device": "cpu"}
    grids = [(1, 2), (2, 1)]
    measured = []

    def measure(cand):  # a candidate costs 0.25 s on the slow rank; its "time" is a function of the candidate alone
        time.sleep(0.25 if rank == 1 else 0.02)
        measured.append(bench.lap_key(cand) + ("/fenced" if cand[4] == "direct" and bench.direct_fenced(ctx) else ""))
        if cand[4] == "direct" and cand[2] == "join" and (join_fails_when == "always" or not bench.direct_fenced(ctx)):
            return None  # (a form that fails, e.g. wrong results under the epoch-stamped check: one rung down the ladder)
        return round(0.1 + 0.01 * (zlib.crc32(bench.lap_key(cand).encode()) % 7) - (0.05 if cand[4] == "direct" and cand[2] == "inline" else 0.0), 5)

    table, stats = {}, {"run": 0, "skipped_for_time": 0, "failed": []}
    canary, transports = bench.calibrate_laplacian(ctx, (1, 2), grids, (False, True), ("rccl", "direct"), measure, (lambda: canary_says),
                                                   rccl_seconds, direct_seconds, table, stats)
    return {"table": table, "stats": stats, "measured": measured, "canary": canary, "transports": list(transports),
            "keys": bench.calibration_line_keys(table, stats, ctx), "dropped": ctx.get("direct_dropped")}


@pytest.mark.multiprocess
def test_a_small_budget_still_yields_an_overlapped_rccl_headline_on_every_rank_alike(tmp_path):
    """gloo, world size 2, rank 1 four times slower per candidate: with a budget of 1.2 s the calibration stops at the SAME
    candidate on both ranks (the budget is a collective), the head of the order -- RCCL "swap" / "join" on the default grid --
    has been measured, the tail is counted as skipped, and the line says what RCCL and the direct transport each achieved."""
    from mp_util import run_ranks

    got = run_ranks(_calibration_worker, 2, tmp_path, args=(1.2, True))
    assert got[0]["measured"] == got[1]["measured"] and got[0]["table"] == got[1]["table"]
    m, keys = got[0]["measured"], got[0]["keys"]
    assert m[:2] == ["1x2_twophase_swap_wg0_rccl", "1x2_twophase_join_wg0_rccl"]
    rccl_run = [k for k in m if k.endswith("_rccl")]
    assert 3 <= len(rccl_run) <= 7  # 1.2 s at 0.25 s per candidate on the slow rank
    assert keys["rccl_best_ms_per_apply"] is not None and keys["rccl_best_form"].split("_")[2] in ("swap", "join")
    assert keys["calibration_candidates_skipped_for_time"] > 0 and keys["calibration_candidates_run"] == len(m)
    # the direct transport had its own second: "inline" first
    direct_run = [k for k in m if "_direct" in k]
    assert direct_run and direct_run[0].split("_")[2] == "inline" and keys["direct_best_ms_per_apply"] < keys["rccl_best_ms_per_apply"]
    assert got[0]["canary"] is True and got[0]["transports"] == ["rccl", "direct"]


def test_the_line_names_every_measured_process_grid_with_its_face_sizes():
    """SURVEY.md section 8e's message sizes, per process grid the calibration measured, beside the best per-apply time on each
    transport: 512^3 over 4 x 2 -> I faces 1 x 256 x 512 x 8 B = 1.05 MB, J faces 1 x (128 + 2) x 512 x 8 B = 0.53 MB; over 1 x 8 only
    N / S faces of 2.1 MB; horizontal diffusion 2048 x 2048 x 80 over 4 x 2 (ghost depth 2) -> 1.31 MB and 0.66 MB."""
    from gt4py_amd.distributed.calibrate import calibration_line_keys, per_process_grid_keys

    table = {"4x2_twophase_swap_wg0_rccl": 0.102, "4x2_singlephase_swap_wg0_rccl": 0.110, "4x2_twophase_inline_wg0_direct": 0.066,
             "1x8_twophase_inline_wg0_direct": 0.052, "1x8_twophase_swap_wg0_rccl": 0.090}
    grids = per_process_grid_keys(table, (512, 512, 512), 1, 8)
    assert set(grids) == {"4x2", "1x8"}
    assert grids["4x2"]["local_domain"] == [128, 256, 512] and grids["4x2"]["face_bytes_per_neighbour"] == {"west_east": 1048576, "north_south": 532480}
    assert grids["1x8"]["face_bytes_per_neighbour"] == {"west_east": 0, "north_south": 2097152} and grids["1x8"]["neighbours"] == 2
    assert grids["4x2"]["best_ms_per_apply"] == {"rccl": 0.102, "direct": 0.066} and grids["4x2"]["best_form"]["rccl"] == "4x2_twophase_swap_wg0_rccl"
    hd = per_process_grid_keys({"overlapped": 0.2, "inline_direct": 0.19}, (2048, 2048, 80), 2, 8, (4, 2))
    assert hd["4x2"]["face_bytes_per_neighbour"] == {"west_east": 1310720, "north_south": 660480} and hd["4x2"]["best_ms_per_apply"] == {"rccl": 0.2, "direct": 0.19}
    # the one-rank self-loop rehearsal of one share of 4 x 2: its "1x1" forms are reported under the grid they rehearse
    loop = per_process_grid_keys({"1x1_twophase_swap_wg4_rccl": 0.0977, "1x1_singlephase_inline_wg0_direct": 0.0718}, (128, 256, 512), 1, 8, (1, 1), (4, 2))
    assert set(loop) == {"4x2"} and loop["4x2"]["selfloop"] is True and loop["4x2"]["local_domain"] == [128, 256, 512]
    assert loop["4x2"]["face_bytes_per_neighbour"] == grids["4x2"]["face_bytes_per_neighbour"]
    stats = {"run": 5, "skipped_for_time": 0, "failed": []}
    keys = calibration_line_keys(table, stats, None, {"total": (512, 512, 512), "halo": 1, "itemsize": 8, "grid": (4, 2)})
    assert keys["per_process_grid"] == grids and "per_process_grid" not in calibration_line_keys(table, stats)


@pytest.mark.multiprocess
def test_a_direct_form_that_fails_moves_every_rank_down_the_ladder(tmp_path):
    """direct -> direct-fenced -> rccl.  A form of the direct transport that fails on some rank -- wrong
    results under the epoch-stamped check, a timeout, a set-up error: `measure` returns None on every rank alike -- ends the direct
    stage, switches every rank to the fenced mode, throws away what was measured without fences and runs the stage again; the line
    says so.  A form that fails WITH fences drops the transport, and nothing it measured can become the headline."""
    from mp_util import run_ranks

    got = run_ranks(_calibration_worker, 2, tmp_path, args=(0.3, True, "unfenced", 30.0))
    assert got[0]["measured"] == got[1]["measured"] and got[0]["table"] == got[1]["table"]
    m, keys = got[0]["measured"], got[0]["keys"]
    direct = [k for k in m if "_direct" in k]
    first_fenced = next(i for i, k in enumerate(direct) if k.endswith("/fenced"))
    # unfenced up to the first "join" (which failed), then the WHOLE stage again with fences -- the failed form included, and passing
    assert direct[first_fenced - 1].split("_")[2] == "join" and not any(k.endswith("/fenced") for k in direct[:first_fenced])
    assert all(k.endswith("/fenced") for k in direct[first_fenced:]) and direct[first_fenced].split("_")[2] == "inline"
    assert any(k.split("_")[2] == "join" for k in direct[first_fenced:]) or keys["calibration_candidates_skipped_for_time"] > 0
    assert keys["direct_transport_mode"] == "direct-fenced" and got[0]["dropped"] is None and got[0]["transports"] == ["rccl", "direct"]
    assert [(s["from"], s["to"]) for s in keys["direct_transport_ladder"]] == [("direct", "direct-fenced")]
    assert keys["direct_transport_ladder"][0]["at"].endswith("_join_wg0_direct") and keys["calibration_candidates_failed"] == []
    assert keys["calibration_candidates_failed_unfenced"] == [keys["direct_transport_ladder"][0]["at"]]
    assert keys["direct_best_ms_per_apply"] is not None  # (measured with fences)

    got = run_ranks(_calibration_worker, 2, tmp_path / "always", args=(0.3, True, "always", 30.0))
    keys = got[0]["keys"]
    assert got[0]["table"] == got[1]["table"] and all(k.endswith("_rccl") for k in got[0]["table"])
    assert keys["direct_transport_mode"] == "rccl" and got[0]["dropped"].endswith("_join_wg0_direct") and got[0]["transports"] == ["rccl"]
    assert [(s["from"], s["to"]) for s in keys["direct_transport_ladder"]] == [("direct", "direct-fenced"), ("direct-fenced", "rccl")]
    assert keys["direct_best_ms_per_apply"] is None and got[0]["stats"]["measured_before_the_drop"]


@pytest.mark.multiprocess
def test_a_failed_canary_keeps_the_calibration_on_rccl(tmp_path):
    from mp_util import run_ranks

    got = run_ranks(_calibration_worker, 2, tmp_path, args=(30.0, False))
    assert got[0]["canary"] is False and got[0]["transports"] == ["rccl"] and got[0]["keys"]["direct_best_ms_per_apply"] is None
    assert all(k.endswith("_rccl") for k in got[0]["measured"]) and got[0]["keys"]["calibration_candidates_skipped_for_time"] == 0
    assert got[0]["keys"]["direct_transport_mode"] == "direct" and got[0]["keys"]["direct_transport_ladder"] == []  # (the canary is the caller's: bench.direct_canary records its own steps)
    # with time to spare every RCCL candidate of the order ran: both grids, both tables, the throttles on the best grid
    assert {k.split("_")[0] for k in got[0]["measured"]} == {"1x2", "2x1"} and any("_wg4_" in k for k in got[0]["measured"])


def _timed_call_fails_on_one_rank(rank: int, world: int, tmpdir: str):
    import torch.distributed as dist

    ctx = {"world": world, "rank": rank, "distributed": True, "dist": dist, "device": "cpu"}
    calls = {"n": 0}

    def fn():  # fails on rank 1 only, in the middle of the timed loop
        calls["n"] += 1
        if rank == 1 and calls["n"] == 4:
            raise RuntimeError("a wait for a neighbour ran out of time")

    raised = None
    try:
        bench._slowest_rank_ms(ctx, fn, calls=6, warm=1)
    except bench.FailedOnSomeRank as ex:
        raised = str(ex)
    still_in_step = bench._agree(ctx, 1)  # the next collective finds every rank in the same place
    dropped = bench.measure_candidate(ctx, (lambda: (fn, (lambda: None))), 4, warm=0)  # (rank 1's callable has failed once; now it works)
    return {"raised": raised, "still_in_step": still_in_step, "calls": calls["n"], "dropped": dropped}


@pytest.mark.multiprocess
def test_a_timed_call_that_fails_on_one_rank_fails_on_every_rank_together(tmp_path):
    """Found by the rehearsal of `bench.py` with real ranks on one device: the direct transport fails hard, so a timed apply can
    raise on the ranks that waited for a neighbour and not on the others -- which must not leave the ranks in different
    collectives.  Every rank raises `FailedOnSomeRank` after the same barrier and reductions; the next collective is in step."""
    from mp_util import run_ranks

    got = run_ranks(_timed_call_fails_on_one_rank, 2, tmp_path)
    assert got[0]["raised"] == "on another rank" and "ran out of time" in got[1]["raised"]
    assert got[0]["still_in_step"] == got[1]["still_in_step"] == 1
    assert got[0]["calls"] >= 7 and got[1]["calls"] >= 4 and got[0]["dropped"] is not None and got[0]["dropped"] == got[1]["dropped"]


def test_the_decomposed_workloads_live_in_the_package_and_bench_re_exports_them():
    """bench.py is the program that meets 8 devices first; the set-up of its two decomposed workloads is a module of the package
    (importable without running the bench) and bench.<name> is the same object."""
    import bench
    from gt4py_amd.distributed import workloads

    for name in ("_setup_distributed_laplacian", "_setup_hdiff2048", "direct_canary", "_native_comm", "gather_rank_proof", "_device_fields",
                 "_time_launches", "hdiff_input", "_lap_definition"):
        assert getattr(bench, name) is getattr(workloads, name)
    before = (workloads.GRID, workloads.HDIFF_SHARE, workloads.HDIFF_GLOBAL)
    try:
        workloads.set_levels(32)  # the one-device rehearsal's slab
        assert workloads.GRID == (512, 512, 32) and workloads.HDIFF_SHARE == (512, 1024, 32) and workloads.HDIFF_GLOBAL == (2048, 2048, 32)
    finally:
        workloads.GRID, workloads.HDIFF_SHARE, workloads.HDIFF_GLOBAL = before
    assert bench.GRID == (512, 512, 512)
