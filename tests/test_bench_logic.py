"""CPU tests of the DRIVER logic of bench.py and atlas_amd/dist.py -- argument handling, the one-JSON-line contract,
the choice of the multi-GPU decomposition with its run-time self-check and fallback, the all-to-all pipeline with its
cross-check, barriers and max-over-ranks timing -- in real processes over gloo, with tests/fake_trans.py standing in for
the device transform (the transform itself is tested on the GPU; nothing here measures anything)."""
import json
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench_worker(rank, world, port, argv, grid, corrupt_mirror, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    os.environ.pop("ATLAS_AMD_BENCH_CROSSCHECK", None)
    import atlas_amd
    import atlas_amd.dist_torch as aadist   # what bench.py selects off the GPU (--dist-impl auto)
    import bench
    from fake_trans import FakeTrans
    FakeTrans.corrupt_mirror = corrupt_mirror
    atlas_amd.Trans = FakeTrans
    aadist.Trans = FakeTrans
    aadist.DEVICE = "cpu"
    bench.DEVICE = "cpu"
    bench.GRID, bench.TRUNC, bench.NLEV = grid, 15, 3
    sys.argv = ["bench.py"] + argv
    sys.stdout = open(os.path.join(outdir, f"stdout_{rank}.txt"), "w")
    try:
        bench.main()
    finally:
        sys.stdout.flush()


def run_bench(tmp_path, world, argv, grid="O16", corrupt_mirror=False, expect_value=True):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, argv, grid, corrupt_mirror, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, f"bench rank exited with {p.exitcode}"
    lines = [ln for ln in open(tmp_path / "stdout_0.txt").read().splitlines() if ln.strip()]
    out = json.loads(lines[-1])                       # the JSON line is the LAST line of rank 0's stdout
    for r in range(1, world):
        assert open(tmp_path / f"stdout_{r}.txt").read().strip() == ""     # only rank 0 prints
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["higher_is_better"] is True and out["dtype"] == "f64" and out["data"] == "synthetic"
    if not expect_value:
        return out
    assert out["value"] > 0 and out["ms_per_step"] > 0 and "workload" in out["config"]
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    return out


def test_single_gpu_line_has_roofline_and_cpu_baseline(tmp_path):
    out = run_bench(tmp_path, 1, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-sample-fields", "2"])
    assert out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "single GPU"
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "transforms/s"
    # the per-stage split and the one-thread figure beside the all-core one
    assert cb["legendre_s"] > 0 and cb["fourier_s"] > 0 and cb["layout_s"] >= 0
    assert cb["one_thread"]["cores"] == 1 and cb["one_thread"]["value"] > 0 and cb["one_thread"]["legendre_s"] > 0
    # three timed blocks of exactly --steps steps; value = the median block; device state (null fields off the GPU)
    rp = out["repeats"]
    assert rp["blocks"] == 3 and rp["steps_per_block"] == 2 and len(rp["values"]) == 3
    assert rp["min"] <= out["value"] <= rp["max"] and sorted(rp["values"])[1] == pytest.approx(out["value"])
    assert out["ms_per_step"] == pytest.approx(sorted(rp["ms_per_step"])[1])
    assert "clocks" in out
    assert "multi_gpu_crosscheck" not in out and "alt_decomposition" not in out
    # the full-field parity block: every grid point of the CPU baseline's transform against the device result
    # of the same spectra (here the stand-in "device" computes something else: only the bookkeeping is checked)
    par = out["parity"]
    assert par["fields"] == 2 and par["rows"] == 32 and par["points"] > 0 and par["tolerance_rel_rms"] == 1e-12
    assert set(par) >= {"rel_rms", "max_abs", "ok", "against"}


def test_gpus_n_without_a_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` the way the driver launches N = 1 (no torch.distributed.run): the script re-executes itself
    under the launcher, rendezvous on 127.0.0.1, and rank 0 prints the one JSON line.  Stand-in transform
    over gloo (BENCH_TEST_STANDIN=1), as everywhere in this file."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"BENCH_TEST_STANDIN": "1", "BENCH_TEST_GRID": "O16", "BENCH_WATCHDOG_S": "200"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "no launcher (WORLD_SIZE unset)" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["test_standin"] is True
    assert out["metric"].startswith("STAND-IN")
    assert out["config"]["parallelism"].startswith(("m-sharded", "mirror-band"))       # --dist-mode auto: the trial's winner
    assert out["multi_gpu_crosscheck"]["bitwise_equal_on_all_ranks"] is True


def test_a_launcher_with_the_wrong_rank_count_is_an_error(tmp_path):
    import subprocess
    env = dict(os.environ)
    env.update({"BENCH_TEST_STANDIN": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode != 0 and "--nproc-per-node 4" in (r.stderr + r.stdout)


PREFIX = {"alltoall": "m-sharded", "mirror": "mirror-band"}


def check_auto_line(out, world):
    """--dist-mode auto: both decompositions pass their bitwise checks, an untimed trial of both decides which
    one is timed, the line says which and why, the other one is `alt_decomposition`, and every rank's stage times are listed"""
    ac = out["auto_choice"]
    assert ac["timed"] in PREFIX and set(ac["trial_ms_per_step"]) == {"alltoall", "mirror"}
    faster = min(ac["trial_ms_per_step"], key=ac["trial_ms_per_step"].get)
    assert ac["timed"] == faster
    assert out["config"]["parallelism"].startswith(PREFIX[ac["timed"]]) and "chosen by --dist-mode auto" in out["config"]["parallelism"]
    assert {"alltoall_period_ms", "mirror_period_ms", "exchange_hidden", "favours", "why"} <= set(ac["model"])
    alt = out["alt_decomposition"]
    assert alt["mode"] == ({"alltoall", "mirror"} - {ac["timed"]}).pop() and alt["value"] > 0
    assert out["multi_gpu_crosscheck"]["bitwise_equal_on_all_ranks"] is True
    pr = out["per_rank"]
    assert [r["rank"] for r in pr] == list(range(world)) and all(r["legendre_ms"] >= 0 and r["fourier_ms"] >= 0 for r in pr)


def test_two_ranks_auto_times_the_winner_of_a_trial_and_reports_the_other_beside_it(tmp_path):
    out = run_bench(tmp_path, 2, ["--gpus", "2", "--steps", "2", "--warmup", "1"])
    check_auto_line(out, 2)
    assert out["dist_impl"].startswith("torch")                           # off the GPU; "native" on MI355X
    assert "cpu_baseline" not in out                                      # rank 0 at N = 1 only
    assert out["repeats"]["blocks"] == 3 and len(out["repeats"]["values"]) == 3


def test_the_model_of_auto_picks_mirror_bands_where_the_exchange_does_not_hide():
    """the committed per-rank cost record (profiles/r05_scaling_model.json) through bench.predict_decomposition: at 50 GB/s per
    link the transposition is exposed at P = 2 and 4; at 150 GB/s it hides from P = 4 on (where the two decompositions are within
    a few per cent of each other and the live trial decides)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "r05_scaling_model.json")) as f:
        sm = json.load(f)
    single = sm["single_gpu"]["ms_per_transform"]
    pair_bytes = {2: 1268e6, 4: 370e6, 8: 100e6}
    picks = {}
    for P in (2, 4, 8):
        ranks = sm["P"][str(P)]["ranks"]
        slow = max(ranks, key=lambda r: r["legendre_ms"] + r["fourier_ms"])
        for rate in (50.0, 150.0):
            m = bench.predict_decomposition(P, slow["legendre_ms"], slow["fourier_ms"], max(r["pack_ms"] for r in ranks),
                                            pair_bytes[P], rate, single_ms=single)
            picks[(P, rate)] = (m["favours"], m["exchange_hidden"])
            assert m["exchange_hidden"] == (m["alltoall_comm_ms"] <= m["alltoall_compute_ms"])
            want = "alltoall" if m["alltoall_period_ms"] <= m["mirror_period_ms"] else "mirror"
            assert m["favours"] == want and (str(int(rate)) in m["why"] or m["exchange_hidden"])
    # exposed at 50 GB/s for P = 2 and 4 (1.27 / 0.37 GB over one link per transform): mirror bands; hidden at 150 GB/s from P = 4 on
    assert picks[(2, 50.0)] == ("mirror", False) and picks[(4, 50.0)] == ("mirror", False)
    assert picks[(4, 150.0)][1] is True and picks[(8, 150.0)][1] is True


def test_mirror_bands_are_not_reported_when_their_self_check_fails(tmp_path):
    out = run_bench(tmp_path, 2, ["--gpus", "2", "--steps", "1", "--warmup", "0"], corrupt_mirror=True)
    assert out["config"]["parallelism"].startswith("m-sharded")
    assert out["alt_decomposition"]["selfcheck_bitwise_equal_on_all_ranks"] is False
    assert "value" not in out["alt_decomposition"]


def test_all_to_all_decomposition_with_cross_check(tmp_path):
    """what the 8-GPU run does (mode alltoall: Legendre stage by wavenumber, all_to_all_single, pipelined transforms,
    then the bitwise cross-check against the band decomposition), here with 3 ranks"""
    out = run_bench(tmp_path, 3, ["--gpus", "3", "--steps", "2", "--warmup", "1", "--dist-mode", "alltoall"])
    assert out["config"]["parallelism"].startswith("m-sharded")
    assert out["multi_gpu_crosscheck"]["bitwise_equal_on_all_ranks"] is True


def test_a_failed_native_driver_withholds_the_value(tmp_path):
    """--gpus N > 1 must measure csrc/dist_trans.hip or nothing: here the library's driver cannot start
    (no HIP device in this container), every rank falls back to the torch.distributed stand-in together -- the line keeps
    that number under `fallback_value`, prints "value": null and says why"""
    out = run_bench(tmp_path, 2, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-impl", "native", "--no-alt"],
                    expect_value=False)
    assert out["native_failed"] is True and out["dist_impl"].startswith("torch (fallback)")
    assert out["value"] is None and out["fallback_value"] > 0 and "error" in out and "dist_trans.hip" in out["error"]


def test_explicit_modes(tmp_path):
    out = run_bench(tmp_path, 2, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-mode", "band"])
    assert out["config"]["parallelism"].startswith("latitude-band") and "alt_decomposition" not in out
    out = run_bench(tmp_path, 2, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-mode", "mirror"])
    assert out["config"]["parallelism"].startswith("mirror-band") and "alt_decomposition" not in out
    out = run_bench(tmp_path, 2, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-alt"])
    assert out["config"]["parallelism"].startswith("m-sharded") and "alt_decomposition" not in out


def test_eight_ranks_auto(tmp_path):
    out = run_bench(tmp_path, 8, ["--gpus", "8", "--steps", "1", "--warmup", "1"], grid="O32")
    assert out["n_gpus"] == 8
    check_auto_line(out, 8)
    assert "8 transform(s) per step" in out["config"]["workload"]


@pytest.mark.parametrize("ngpus", [2, 8])
def test_bench_dry_run_checks_the_message_plan_without_a_device(ngpus):
    """`bench.py --gpus N --dry-run`: the library's host code builds bands and messages of all N ranks
    at the headline resolution; the plan must cover both buffers exactly once, the two ends of every pair must agree, and
    the totals must be the kept part of the intermediate"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ngpus), "--dry-run", "--xgmi-gbs", "40"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["dry_run"] and d["n_gpus"] == ngpus and d["problems"] == []
    assert all(d["plan_checks"].values())
    assert d["kept_intermediate_bytes"] == 2312346 * 274 * 8
    assert sum(d["latitude_bands_rows"]) == 2560 and sum(d["band_points"]) == 6599680
    lo, hi = d["bytes_leaving_a_gpu_per_transform"]["min"], d["bytes_leaving_a_gpu_per_transform"]["max"]
    share = d["kept_intermediate_bytes"] / ngpus * (ngpus - 1) / ngpus
    assert 0.8 * share < lo <= hi < 1.2 * share
    assert d["expected_exchange_ms_per_transform"]["assumed_link_GBs"] == 40


def _watchdog_worker(outdir):
    sys.path.insert(0, ROOT)
    import time
    import bench
    os.environ["BENCH_WATCHDOG_S"] = "0.5"
    sys.stdout = open(os.path.join(outdir, "wd_stdout.txt"), "w")
    wd = bench.Watchdog(rank=0, world=2, enabled=True)
    wd.phase("a phase that finishes")
    wd.phase("a phase that hangs")
    time.sleep(30)


def test_watchdog_reports_the_hung_phase_of_a_multi_rank_run(tmp_path):
    """bench.py, N > 1: a phase that does not finish ends the process with status 3 and a JSON line whose value is null"""
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_watchdog_worker, args=(str(tmp_path),))
    p.start()
    p.join(20)
    assert p.exitcode == 3
    out = json.loads(open(tmp_path / "wd_stdout.txt").read().strip().splitlines()[-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "a phase that hangs" in out["error"]
    import bench
    wd = bench.Watchdog(rank=0, world=1, enabled=False)     # single GPU: never armed
    wd.phase("x")
    assert wd.timer is None
    wd.done()
