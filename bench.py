"""Headline benchmark: inverse spherical-harmonics transforms per second, TL1279 -> O1280, 137 levels.

One *transform* = one TransLocal::invtrans of one field on all 137 levels (nb_scalar_fields = 137); a *step* is
one such transform per GPU with spectra and grid-point arrays resident in HBM.
    python bench.py --gpus N --steps K --warmup W
N = 1: single MI355X (the whole transform on one device).
N > 1: launched by torch.distributed.run, one rank per GPU; every step processes N transforms (weak scaling), each
       transform distributed over all N GPUs and returned in Atlas's latitude bands (atlas_amd/dist.py).  The timed
       decomposition is the one BASELINE configs C3 / C4 name: Legendre stage sharded by zonal wavenumber, m ->
       latitude-band transposition of the Fourier intermediate over RCCL (ncclSend / ncclRecv groups issued by the library,
       csrc/comm.hip, on a second HIP stream), pipelined against the neighbouring transforms, Fourier stage on the local
       band.  torch.distributed only launches the ranks, carries RCCL's 128-byte unique id and the barriers.  After the
       timed region (a) one transform is re-run with the exchange-free latitude-band decomposition and compared bit for
       bit ("multi_gpu_crosscheck"), (b) the exchange-free mirror-band decomposition (a northern band of rows and its
       mirror image per GPU, work 1/N) is timed too and reported as "alt_decomposition" if it first reproduces, bit for
       bit on every rank, the same rows computed through the zonal-band crop path.
       --dist-mode band|mirror times one of the exchange-free decompositions instead; --dist-impl torch uses the
       torch.distributed implementation of the driver (atlas_amd/dist_torch.py; also what the CPU tests of this file
       run over gloo).
Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

# torch.distributed.run exports OMP_NUM_THREADS=1 to every rank; the (untimed) set-up computes the Legendre tables on
# the host with OpenMP -- about 80 s on one thread at T1279 -- so give every rank its share of the cores.  Must happen
# before any OpenMP runtime is loaded (the library, torch).
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode); the GPU
# box exports this already -- keep it if some launcher dropped it.  Must be set before the HIP runtime starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
_lws = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
if _lws > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1":
    os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // _lws))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRID, TRUNC, NLEV = "O1280", 1279, 137
DEVICE = "cuda"                # tests/test_bench_logic.py runs this file's control flow on "cpu" with a stand-in transform
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X spec, dense fp64 matrix = 1024 SIMDs x 32 flop/cycle x 2.4 GHz.  A kernel of nothing
                               # but v_mfma_f64_16x16x4_f64 sustains 67.3 TF/s (shader clock 2.05 GHz under fp64 matrix
                               # load, profiles/r02_mfma_sustained.txt); bench.py measures that in-run as
                               # roofline.peak_sustained_measured -- `frac` stays against the spec number
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
PARITY_TOL = 1e-12             # rel-RMS of the whole field against the CPU restatement (the reference's tests: 1e-13 against analytic
                               # fields at T63; at T1279 the two fp64 summation orders differ by ~1e-15)


KERNEL_SOURCES = ("legendre_kernel.hip", "fft_kernel.hip", "fft_kernel_pairs.hip", "fft_ct_rows.h", "fft_pair.h", "fft_device.h", "fft_core.h", "fft_plan.cpp",
                  "trans.hip", "trans_plan.cpp", "Makefile")   # the Makefile carries per-file code-generation flags


def kernel_source_digest():
    """SHA-256 over the sources that determine the two stages' memory traffic: a PMC measurement is attached to a bench
    line only if it was taken on exactly these sources (tools/prof_round_summary.py stamps the profile with it)"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "atlas_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_valu_instructions():
    """vector-ALU wave-instructions per transform of the two stages from the same committed, digest-stamped PMC profile
    (SQ_INSTS_VALU): a stage that is bound by its own instruction stream has the floor
    instructions x 4 cycles (fp64, one wavefront-instruction per 4 cycles and SIMD) / 1024 SIMDs / clock."""
    import glob
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_sha256") == kernel_source_digest() and d.get("valu_wave_instructions_per_transform"):
            best = dict(d["valu_wave_instructions_per_transform"])
            best["_profile"] = os.path.basename(path)
    return best


def measured_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; tools/prof_round.sh).  PMC counters cannot be read from
    inside the timed process, so the figures come from the newest committed profile -- and only if it carries the digest
    of the kernel sources this run was built from; a stale profile yields null, not an old number."""
    import glob
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_sha256") == kernel_source_digest():
            best = dict(d.get("traffic_bytes_per_launch", {}))
            best["_profile"] = os.path.basename(path)
    return best


def cpu_baseline(sample_fields=8):
    """the oracle (CPU restatement of TransLocal: per-m GEMM pair + per-row c2r FFT) timed on the host cores on a
    bounded sample: `sample_fields` of the 137 levels, full TL1279 -> O1280 geometry."""
    import numpy as np
    import atlas_amd
    import oracle
    from helpers import red_spectra
    g = atlas_amd.Grid(GRID)
    op = oracle.OraclePlan(TRUNC, g.nx(), g.y(), with_tables=True)   # setup (tables) is not timed, as on the GPU
    sp = red_spectra(TRUNC, sample_fields)
    # untimed warm-up with one field: builds the per-row-length FFT plans (tables; serial), like the GPU setup
    op.invtrans(1, np.ascontiguousarray(sp.reshape(-1, sample_fields)[:, :1]).reshape(-1), use_fft=True)
    t0 = time.perf_counter()
    op.invtrans(sample_fields, sp, use_fft=True)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": (sample_fields / NLEV) / dt, "unit": "transforms/s", "cores": cores, "kind": "port",
            "sample": f"{sample_fields} of {NLEV} levels of one TL{TRUNC}->{GRID} transform in {dt:.2f} s "
                      f"(oracle/translocal_oracle.c: plain restatement, OpenMP over m and over (field,row), tables and FFT plans "
                      f"built beforehand); scaled by {NLEV}/{sample_fields}"}


def cpu_baseline_blas(sample_fields=8, keep_field=False, one_thread_fields=4):
    """CPU baseline: the reference's algorithm with library kernels, BLAS dgemm (numpy) and pocketfft (scipy.fft) -- the
    reference's eckit "lapack" + pocketfft configuration (oracle/translocal_blas.py) -- on a pool of threads over wavenumbers /
    row groups, with the wall-clock split SURVEY 8(d) asks for (legendre_s / fourier_s / layout_s) and a one-thread figure from
    a `one_thread_fields`-level sample scaled to 137 levels.  keep_field: also return the grid-point field it computed and its
    spectra (the checker of the bench line's `parity` block)."""
    import numpy as np
    import atlas_amd
    import oracle
    from oracle.translocal_blas import invtrans_blas
    from helpers import red_spectra
    g = atlas_amd.Grid(GRID)
    op = oracle.OraclePlan(TRUNC, g.nx(), g.y(), with_tables=True)
    sp = red_spectra(TRUNC, sample_fields)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    workers = max(1, min(avail, 64))   # one thread per core of a 64-core host; more threads than that only fight over memory
    invtrans_blas(op, 1, np.ascontiguousarray(sp.reshape(-1, sample_fields)[:, :1]).reshape(-1), workers=workers)   # warm-up
    # three timed calls (each the whole sample): the MEDIAN is `value`, all three are listed (repeated calls
    # after a warm-up; one call is 1.5 - 2 s at 137 levels on 64 threads)
    runs = []
    for _ in range(3):
        tm_i = {}
        t0 = time.perf_counter()
        field = invtrans_blas(op, sample_fields, sp, workers=workers, timings=tm_i)
        runs.append((time.perf_counter() - t0, tm_i))
    runs.sort(key=lambda r_: r_[0])
    dt, tm = runs[1]
    line = {"value": (sample_fields / NLEV) / dt, "unit": "transforms/s", "cores": workers, "kind": "port",
            "legendre_s": tm.get("legendre_s"), "fourier_s": tm.get("fourier_s"), "layout_s": tm.get("layout_s"),
            "calls_s": [r_[0] for r_ in runs], "value_is": "median of three timed calls",
            "sample": f"{sample_fields} of {NLEV} levels of one TL{TRUNC}->{GRID} transform in {dt:.2f} s on {workers} threads "
                      f"(of {avail} available): the reference's algorithm with library kernels -- per-m dgemm pairs (numpy/OpenBLAS, "
                      f"one wavenumber per thread) + pocketfft c2r batched per row length (scipy.fft) -- i.e. TransLocal with eckit "
                      f"'lapack' + pocketfft (oracle/translocal_blas.py; tables built beforehand); legendre_s / fourier_s: wall clock "
                      f"of the two stages, layout_s: the part of it spent in the split / merge / transpose copies (thread-time / "
                      f"threads); scaled by {NLEV}/{sample_fields}"}
    if one_thread_fields and one_thread_fields > 0:
        k = int(min(one_thread_fields, sample_fields))
        sp1 = np.ascontiguousarray(sp.reshape(-1, sample_fields)[:, :k]).reshape(-1)
        tm1 = {}
        t0 = time.perf_counter()
        invtrans_blas(op, k, sp1, workers=1, timings=tm1)
        dt1 = time.perf_counter() - t0
        line["one_thread"] = {"value": (k / NLEV) / dt1, "unit": "transforms/s", "cores": 1,
                              "legendre_s": tm1.get("legendre_s"), "fourier_s": tm1.get("fourier_s"), "layout_s": tm1.get("layout_s"),
                              "sample": f"{k} of {NLEV} levels in {dt1:.2f} s on one thread (BLAS limited to one thread), scaled by {NLEV}/{k}"}
    return (line, field, sp) if keep_field else line


def full_field_parity(tr, grid, sample_fields, field_ref, sp_ref, device):
    """The whole grid-point field of the CPU baseline's transform (every row, every sampled level) against the device result
    of the SAME spectra through the timed entry point, after the timed region: rel_rms = RMS(gpu - cpu) / max|cpu|, the
    metric of the reference's own tests (compute_rms, src/tests/trans/test_transgeneral.cc:472-489).  The oracle is the
    checker here, never the thing measured."""
    import numpy as np
    import torch
    nf = int(sample_fields)
    gp = torch.full((nf * grid.size(),), float("nan"), dtype=torch.float64, device=device)
    tr.invtrans(nf, torch.from_numpy(np.ascontiguousarray(sp_ref)).to(device), gp)
    tr.synchronize()
    got = gp.cpu().numpy().reshape(nf, -1)
    del gp
    ref = np.asarray(field_ref, dtype=np.float64).reshape(nf, -1)
    # field by field: no temporaries of the size of the whole field (7.2 GB each at 137 levels)
    mx, worst, ssq, finite = 0.0, 0.0, 0.0, True
    for k in range(nf):
        d = got[k] - ref[k]
        finite = finite and bool(np.isfinite(d).all())
        mx = max(mx, float(np.abs(ref[k]).max()))
        worst = max(worst, float(np.abs(d).max()))
        ssq += float(np.dot(d, d))
    rms = float(np.sqrt(ssq / ref.size) / mx) if mx > 0 and finite else float("nan")
    if not finite:
        worst = float("nan")
    return {"rel_rms": rms, "max_abs": worst, "max_abs_ref": mx, "rows": int(grid.ny()), "fields": nf,
            "points": int(grid.size()), "tolerance_rel_rms": PARITY_TOL, "ok": bool(rms <= PARITY_TOL),
            "against": "cpu_baseline's field (oracle/translocal_blas.py: per-m dgemm + per-row pocketfft c2r), same spectra, "
                       "every grid point of every level; metric compute_rms of test_transgeneral.cc:472-489"}


def dry_run(args):
    """`bench.py --gpus N --dry-run`: everything of the N-GPU run that can be decided without a device.  Runs in one process
    on the host (CI, no GPU): the same C++ code that DistributedTrans::ensure() runs on every rank builds the latitude bands
    and the message lists of ALL N ranks; the checks below are the ones whose failure would otherwise show up as a hang or
    as wrong rows on first contact with real hardware."""
    import ctypes as C
    import numpy as np
    import atlas_amd
    from atlas_amd import _lib
    from atlas_amd.dist import packed_transpose_messages
    P, nf = int(args.gpus), NLEV
    g = atlas_amd.Grid(GRID)
    nlat0 = np.zeros(TRUNC + 1, dtype=np.int32)
    mm = np.zeros(g.ny(), dtype=np.int32)
    _lib.check(_lib.trans_geometry_probe(g._h, TRUNC, 0, nlat0.ctypes.data, mm.ctypes.data))
    mm = np.minimum(mm, TRUNC)
    bands = np.zeros(P + 1, dtype=np.int32)
    _lib.check(_lib.latitude_bands(g._h, TRUNC, P, bands.ctypes.data))
    cols = 2 * nf
    plans = [packed_transpose_messages(mm, cols, bands, P, part) for part in range(P)]
    problems = []
    link = np.zeros((P, P), dtype=np.int64)   # bytes rank a -> rank b
    for a, (msgs, (stot, rtot)) in enumerate(plans):
        send = np.zeros(stot, dtype=np.int8)
        recv = np.zeros(rtot, dtype=np.int8)
        for peer, sb, se, rb, re in msgs:
            send[sb:se] += 1
            recv[rb:re] += 1
            link[a, peer] += (se - sb) * 8
        if not (send == 1).all():
            problems.append(f"rank {a}: send buffer not covered exactly once")
        if not (recv == 1).all():
            problems.append(f"rank {a}: receive buffer not covered exactly once")
    for a in range(P):
        for b in range(P):
            sa = [se - sb for peer, sb, se, _, _ in plans[a][0] if peer == b]
            rb_ = [re - rb for peer, _, _, rb, re in plans[b][0] if peer == a]
            if sa != rb_:
                problems.append(f"ranks {a} -> {b}: piece sizes differ between the two ends")
    kept = int((mm.astype(np.int64) + 1).sum()) * cols * 8
    off_dev = link.copy()
    np.fill_diagonal(off_dev, 0)
    if int(link.sum()) != kept:
        problems.append("the messages do not add up to the kept part of the intermediate")
    worst_link = int(off_dev.max()) if P > 1 else 0
    per_gpu_out = off_dev.sum(axis=1)
    args.xgmi_gbs = 50.0 if args.xgmi_gbs is None else args.xgmi_gbs
    rate = args.xgmi_gbs * 1e9
    rows = [int(bands[q + 1] - bands[q]) for q in range(P)]
    pts = np.concatenate([[0], np.cumsum(np.asarray(g.nx(), dtype=np.int64))])
    out = {
        "dry_run": True, "n_gpus": P, "workload": f"TL{TRUNC} -> {GRID}, {nf} fields per transform",
        "decomposition": "Legendre stage: wavenumbers m % P == rank; transposition: packed runs (kept wavenumbers x 2 nf columns "
                         "per row) rank -> owner of the row's latitude band; Fourier stage on the band",
        "latitude_bands_rows": rows,
        "band_points": [int(pts[bands[q + 1]] - pts[bands[q]]) for q in range(P)],
        "kept_intermediate_bytes": kept,
        "bytes_leaving_a_gpu_per_transform": {"min": int(per_gpu_out.min()), "max": int(per_gpu_out.max())},
        "largest_pair_message_bytes": worst_link,
        "messages_per_rank": len(plans[0][0]),
        "expected_exchange_ms_per_transform": {
            "assumed_link_GBs": args.xgmi_gbs,
            "all_links_concurrent (bound by the busiest pair)": worst_link / rate * 1e3,
            "one_link_at_a_time (sum over peers of the busiest gpu)": float(per_gpu_out.max()) / rate * 1e3},
        "single_gpu_stage_ms_for_scale": "Legendre ~8.5 / P (m-sharded), Fourier ~7.8 / P (banded) -- profiles/r03_*",
        "plan_checks": {"send_and_receive_buffers_covered_exactly_once": not any("covered" in p for p in problems),
                        "both_ends_of_every_pair_agree": not any("differ" in p for p in problems),
                        "total_equals_kept_intermediate": not any("add up" in p for p in problems)},
        "problems": problems,
    }
    # what `--dist-mode auto` would be expected to pick at this link rate, from the newest committed per-rank cost record
    # (tools/scaling_model.py: every rank's share alone on one device); a live run decides by an untimed trial of both
    import glob
    models = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scaling_model.json")))
    if models and P > 1:
        with open(models[-1]) as f:
            sm = json.load(f)
        ranks = (sm.get("P", {}).get(str(P)) or {}).get("ranks")
        if ranks:
            slow = max(ranks, key=lambda r: r["legendre_ms"] + r["fourier_ms"])
            out["auto_choice_predicted"] = predict_decomposition(
                P, slow["legendre_ms"], slow["fourier_ms"], max(r["pack_ms"] for r in ranks), worst_link, args.xgmi_gbs,
                single_ms=sm["single_gpu"]["ms_per_transform"])
            out["auto_choice_predicted"]["costs_from"] = "profiles/" + os.path.basename(models[-1])
    sys.stdout.write(json.dumps(out) + "\n")
    if problems:
        raise SystemExit(1)


def self_launch(ngpus):
    """re-executes this command line under torch.distributed.run, one rank per GPU of this node"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(ngpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] no launcher (WORLD_SIZE unset): " + " ".join(cmd) + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def standin_transform():
    """BENCH_TEST_STANDIN=1 (CPU tests of the driver logic only): tests/fake_trans.py stands in for the device transform on a
    tiny grid, over gloo.  The JSON line then says so in `metric` and carries "test_standin": true -- it is not a measurement."""
    global DEVICE, GRID, TRUNC, NLEV
    if os.environ.get("BENCH_TEST_STANDIN") != "1":
        return False
    import atlas_amd
    import atlas_amd.dist_torch as aadist
    from fake_trans import FakeTrans
    atlas_amd.Trans = FakeTrans
    aadist.Trans = FakeTrans
    aadist.DEVICE = "cpu"
    DEVICE = "cpu"
    GRID, TRUNC, NLEV = os.environ.get("BENCH_TEST_GRID", "O16"), 15, 3
    return True


class Watchdog:
    """N > 1 only: a multi-rank run that hangs (a collective one rank never enters, an RCCL group whose two ends disagree)
    would sit until the caller's time limit with nothing on stdout.  Every phase of the run re-arms a timer; if a phase does not
    finish in `limit_s` seconds the rank says which one on stderr, rank 0 prints a JSON line with "error" (value null:
    unmeasured, never a number) and the process exits with status 3.  BENCH_WATCHDOG_S overrides (0: off)."""

    def __init__(self, rank, world, enabled, limit_s=None):
        self.rank, self.world = rank, world
        self.limit_s = float(os.environ.get("BENCH_WATCHDOG_S", "900" if limit_s is None else str(limit_s)))
        self.enabled = bool(enabled) and self.limit_s > 0
        self.timer, self.name = None, None

    def _expired(self, name):
        sys.stderr.write(f"[bench] rank {self.rank}: watchdog: phase '{name}' did not finish in {self.limit_s:.0f} s\n")
        sys.stderr.flush()
        if self.rank == 0:
            sys.stdout.write(json.dumps({
                "metric": "inverse SH transforms/sec (TL1279, O1280, 137 lev)", "value": None, "unit": "transforms/s",
                "n_gpus": self.world, "error": f"watchdog: phase '{name}' did not finish in {self.limit_s:.0f} s on rank 0 "
                                               f"(multi-rank run hung; see stderr of the other ranks)"}) + "\n")
            sys.stdout.flush()
        os._exit(3)

    def phase(self, name):
        if not self.enabled:
            return
        import threading
        self.done()
        self.name = name
        self.timer = threading.Timer(self.limit_s, self._expired, args=(name,))
        self.timer.daemon = True
        self.timer.start()

    def done(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


class DeviceSampler:
    """Clocks / power / temperature of the device WHILE the timed blocks run (a bench line must let a reader
    tell a slow box from a slow build).  A thread polls amdsmi's gpu_metrics (0.2 ms per call, no subprocess) every few
    milliseconds: shader clock per XCD, memory clock, socket power, hot-spot temperature, and the firmware's throttle-residency
    accumulators (power / thermal / PROCHOT), whose increase over the timed region divided by the increase of the accumulation
    counter is the share of the time the part was held back.  Fallback: the hwmon files of the card; neither: every field null.
    A measurement aid: it never fails the bench."""
    FIELDS = ("prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc")

    def __init__(self, device_index=0, period_s=0.004):
        self.period, self.samples, self.thread, self.stop_flag = period_s, [], None, False
        self.read, self.source, self.static = None, None, {}
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[device_index]
            amdsmi.amdsmi_get_gpu_metrics_info(h)
            self.read, self.source = (lambda: amdsmi.amdsmi_get_gpu_metrics_info(h)), "amdsmi gpu_metrics"
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(h)
                self.static["power_cap_w"] = float(cap["power_cap"]) / 1e6
                clk = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                self.static["sclk_max_mhz"] = clk.get("max_clk")
            except Exception:
                pass
        except Exception:
            import glob
            hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
            if hw:
                d = os.path.dirname(hw[min(device_index, len(hw) - 1)])

                def rd(name, scale):
                    try:
                        with open(os.path.join(d, name)) as f:
                            return float(f.read()) / scale
                    except Exception:
                        return None
                self.read = lambda: {"current_gfxclk": rd("freq1_input", 1e6), "current_uclk": rd("freq2_input", 1e6),
                                     "current_socket_power": rd("power1_input", 1e6), "temperature_hotspot": rd("temp2_input", 1e3),
                                     "temperature_mem": rd("temp3_input", 1e3)}
                self.source = "hwmon sysfs"
                self.static["power_cap_w"] = rd("power1_cap", 1e6)

    @staticmethod
    def _num(v):
        return float(v) if isinstance(v, (int, float)) else None

    def snapshot(self):
        if self.read is None:
            return None
        try:
            m = self.read()
        except Exception:
            return None
        clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float))] or \
               [c for c in [m.get("current_gfxclk")] if isinstance(c, (int, float))]
        out = {"t": time.perf_counter(), "sclk": (sum(clks) / len(clks)) if clks else None, "sclk_min_xcd": min(clks) if clks else None,
               "mclk": self._num(m.get("current_uclk")), "power": self._num(m.get("current_socket_power")),
               "temp_hotspot": self._num(m.get("temperature_hotspot")), "temp_mem": self._num(m.get("temperature_mem")),
               "acc": self._num(m.get("accumulation_counter"))}
        for k in self.FIELDS:
            out[k] = self._num(m.get(k))
        return out

    def start(self):
        if self.read is None:
            return
        import threading
        self.samples, self.stop_flag = [], False

        def loop():
            while not self.stop_flag:
                s_ = self.snapshot()
                if s_ is not None:
                    self.samples.append(s_)
                time.sleep(self.period)
        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()

    def stop(self):
        if self.thread is not None:
            self.stop_flag = True
            self.thread.join()
            self.thread = None

    def summary(self, before, after):
        """before / after: snapshots taken outside the timed blocks (idle device); samples: inside them"""
        if self.read is None:
            return {"source": None, "note": "neither amdsmi nor the card's hwmon files are readable here"}

        def stat(key):
            v = [s_[key] for s_ in self.samples if s_.get(key) is not None]
            return {"mean": sum(v) / len(v), "min": min(v), "max": max(v)} if v else None
        out = {"source": self.source, "samples_in_timed_blocks": len(self.samples), "period_ms": self.period * 1e3,
               "sclk_mhz": stat("sclk"), "sclk_slowest_xcd_mhz": stat("sclk_min_xcd"), "mclk_mhz": stat("mclk"),
               "socket_power_w": stat("power"),
               "temp_hotspot_c": {"before": (before or {}).get("temp_hotspot"), "after": (after or {}).get("temp_hotspot")},
               "temp_mem_c": {"before": (before or {}).get("temp_mem"), "after": (after or {}).get("temp_mem")}}
        out.update(self.static)
        if before and after and before.get("acc") is not None and after.get("acc") is not None and after["acc"] > before["acc"]:
            span = after["acc"] - before["acc"]
            out["throttle_residency"] = {k.replace("_residency_acc", ""): ((after[k] - before[k]) / span
                                                                          if before.get(k) is not None and after.get(k) is not None else None)
                                         for k in self.FIELDS}
            out["throttle_residency"]["note"] = ("share of the firmware's accumulation ticks between the snapshot before the first and "
                                                 "after the last timed block in which the limiter was active (ppt = socket power)")
        return out


def measured_hbm_copy_rate(device, nbytes=2 << 30, repeats=5):
    """what this box's HBM delivers on a plain device-to-device copy (read + write bytes per second), measured with the same
    events as the stages right after the timed blocks: a box-speed reference for the bandwidth side, like
    peak_sustained_measured for the matrix side"""
    import torch
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=device).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(repeats):
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def predict_decomposition(world, leg_ms, fft_ms, pack_ms, busiest_pair_bytes, link_gbs, mirror_ms=None, single_ms=None):
    """The model behind `--dist-mode auto` (also what --dry-run prints): per transform and rank, in the
    pipelined driver the Trans stream carries L + F and the communication stream pack + exchange, the exchange bound by the busiest
    pair's bytes over one xGMI link; the period is the larger of the two.  The exchange-free mirror bands cost (L1 + F1) / P plus
    the polar-band penalty (measured when available).  Returns the prediction and which decomposition it favours and why."""
    exch_ms = busiest_pair_bytes / (link_gbs * 1e9) * 1e3 if link_gbs and link_gbs > 0 else float("inf")
    compute = leg_ms + fft_ms
    comm = pack_ms + exch_ms
    a2a = max(compute, comm)
    if mirror_ms is None and single_ms is not None:
        mirror_ms = 1.10 * single_ms / world          # polar bands cost 10 % more in the Fourier stage (profiles/r05_scaling_model.json)
    out = {"alltoall_period_ms": a2a, "alltoall_compute_ms": compute, "alltoall_comm_ms": comm, "exchange_ms_at_link_rate": exch_ms,
           "link_gbs": link_gbs, "busiest_pair_bytes": int(busiest_pair_bytes), "exchange_hidden": bool(comm <= compute),
           "mirror_period_ms": mirror_ms}
    if mirror_ms is None or a2a <= mirror_ms:
        out["favours"] = "alltoall"
        out["why"] = (f"the exchange hides: pack + exchange {comm:.2f} ms <= Legendre + Fourier {compute:.2f} ms per transform and rank"
                      if comm <= compute else
                      f"the exchange is exposed ({comm:.2f} ms against {compute:.2f} ms of compute) but still beats the mirror bands")
    else:
        out["favours"] = "mirror"
        out["why"] = (f"the exchange does not hide at {link_gbs:.0f} GB/s per link: pack + exchange {comm:.2f} ms against Legendre + "
                      f"Fourier {compute:.2f} ms per transform and rank; mirror bands need no exchange ({mirror_ms:.2f} ms)")
    return out


def link_probe(world, device, mib_per_pair=32, reps=3):
    """what a pair of GPUs moves while ALL pairs exchange at once (the pattern of the transposition): torch.distributed
    all_to_all_single over RCCL, `mib_per_pair` MiB per pair and direction; GB/s per pair and direction"""
    import torch
    import torch.distributed as dist
    n = (mib_per_pair << 20) // 8
    a = torch.ones(world * n, dtype=torch.float64, device=device)
    b = torch.empty_like(a)
    dist.all_to_all_single(b, a)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.all_to_all_single(b, a)
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = min(best, float(t.item()))
    del a, b
    return {"pattern": f"all_to_all_single, {mib_per_pair} MiB per pair and direction, all pairs at once (RCCL via torch.distributed)",
            "gbs_per_pair_and_direction": (mib_per_pair << 20) / best / 1e9, "seconds": best}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=3,
                    help="timed blocks of exactly --steps steps each (barrier + synchronize on both sides of every block); `value` is "
                         "the MEDIAN block, min / max are reported beside it under `repeats`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-fields", type=int, default=NLEV,
                    help="levels of the transform the CPU baseline runs (default: all 137 -- the dgemm shapes of the real call)")
    ap.add_argument("--cpu-baseline-naive", action="store_true",
                    help="also time the plain-loop restatement (oracle/translocal_oracle.c, OpenMP) as cpu_baseline_naive")
    ap.add_argument("--cpu-baseline-blas", action="store_true", help=argparse.SUPPRESS)   # the default now
    ap.add_argument("--force-dist", action="store_true", help="exercise the distributed driver even with one rank")
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "alltoall", "band", "mirror"],
                    help="N > 1: wavenumber sharding + RCCL transposition (auto, alltoall), or one of the exchange-free "
                         "decompositions (see atlas_amd/dist.py)")
    ap.add_argument("--dist-impl", default="auto", choices=["auto", "native", "torch"],
                    help="N > 1: driver inside the library (RCCL from C++) or the torch.distributed one (auto: native on GPUs)")
    ap.add_argument("--no-alt", action="store_true", help="N > 1: do not time the alternative decomposition (auto then times the "
                                                         "wavenumber-sharded one without a trial)")
    ap.add_argument("--trial-steps", type=int, default=3,
                    help="N > 1, --dist-mode auto: steps of the untimed trial of each decomposition that decides which one is timed")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device work: build the message plan of the distributed transform for --gpus N ranks (host code of "
                         "the library), check that it covers the send and receive buffers exactly once and that both ends of "
                         "every pair agree, print per-link bytes and the expected exchange time; one JSON line")
    ap.add_argument("--xgmi-gbs", type=float, default=None,
                    help="--dry-run: assumed one-directional rate of one xGMI link in GB/s (MI355X: 7 links x ~153 GB/s "
                         "aggregate per GPU on paper; RCCL send/recv between two GPUs has been seen at ~50 GB/s)")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    import atlas_amd
    from helpers import red_spectra

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the ranks ourselves (one per GPU, rendezvous on 127.0.0.1 --
        # the container's host name may not resolve); the ranks inherit stdout, so the contract (ONE JSON line, from rank 0)
        # and the per-phase watchdog are those of a launched run
        return self_launch(args.gpus)
    standin_transform()          # BENCH_TEST_STANDIN=1 only (tests/test_bench_logic.py)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher)")
    on_gpu = DEVICE == "cuda"
    if on_gpu:
        torch.cuda.set_device(local_rank)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
    nf = NLEV
    g = atlas_amd.Grid(GRID)

    use_dist = world > 1 or args.force_dist
    wd = Watchdog(rank, world, enabled=world > 1)
    wd.phase("set-up (process group, tables, communicator)")
    crosscheck = None
    dm = None
    if not use_dist:
        tr = atlas_amd.Trans(g, TRUNC, profile=True)
        tr.use_torch_stream()
        sp = torch.from_numpy(red_spectra(TRUNC, nf)).to(DEVICE)
        gp = torch.zeros(nf * g.size(), dtype=torch.float64, device=DEVICE)

        def step():
            tr.invtrans(nf, sp, gp)

        def barrier():
            sync()
    else:
        import torch.distributed as dist
        if "RANK" not in os.environ:   # --force-dist in a plain process: a one-rank group of its own
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": str(sk.getsockname()[1])})
            sk.close()
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        impl = args.dist_impl if args.dist_impl != "auto" else ("native" if on_gpu else "torch")
        impl_note = None
        if impl == "native":
            from atlas_amd.dist import DistributedTrans
        else:
            from atlas_amd.dist_torch import DistributedTrans
        # every rank holds the spectra of the transforms of a step (replicated input, as TransLocal's callers hold it;
        # each rank reads only the wavenumbers it owns)
        sps = [torch.from_numpy(red_spectra(TRUNC, nf, seed=20251114 + i)).to(DEVICE) for i in range(min(world, 2))]
        mode = "alltoall" if args.dist_mode == "auto" else args.dist_mode
        try:
            dtr = DistributedTrans(g, TRUNC, profile=True, mode=mode)
            ok = 1
        except Exception as e:   # e.g. RCCL not loadable from the library: every rank falls back together
            sys.stderr.write(f"[bench] rank {rank}: native distributed driver unavailable: {type(e).__name__}: {e}\n")
            ok, dtr = 0, None
            impl_note = f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=DEVICE)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if impl != "native":
                raise SystemExit("distributed driver failed to start")
            from atlas_amd.dist_torch import DistributedTrans
            impl, dtr = "torch (fallback)", DistributedTrans(g, TRUNC, profile=True, mode=mode)
        tr = dtr.trans
        gp = torch.zeros(nf * tr.nb_gridpoints(), dtype=torch.float64, device=DEVICE)

        # multi-GPU parity evidence, BEFORE the timed region so that a wrong transport cannot produce a timed number: the
        # transposed decomposition must reproduce, bit for bit, what the exchange-free latitude-band decomposition
        # computes for this rank's rows (tests/test_gpu_trans.py and tests/test_gpu_dist_native.py show both equal the
        # single-device result).  If the library's own driver fails it, every rank switches to the torch.distributed
        # driver together and the JSON line says so.
        def crosscheck_against_bands(d, cls):
            ok, err = 0, None
            try:
                dto = cls(g, TRUNC, mode="band")
                gp_a, gp_b = torch.empty_like(gp), torch.empty_like(gp)
                d.invtrans(nf, sps[0], gp_a)
                dto.invtrans(nf, sps[0], gp_b)
                sync()
                ok = int(bool(torch.equal(gp_a, gp_b)) and bool(torch.isfinite(gp_a).all())
                         and float(gp_a.abs().max()) > 0.0)
                del dto, gp_a, gp_b
            except Exception as e:  # the check must never cost the measurement
                err = f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device=DEVICE)   # every rank takes part, failed or not
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            out = {"compared": f"alltoall vs band decomposition, {nf} fields, every rank's latitude band, before the "
                               f"timed region", "bitwise_equal_on_all_ranks": bool(int(flag.item()))}
            if err:
                out["error_on_rank0"] = err
            if on_gpu:
                torch.cuda.empty_cache()
            return out

        if dtr.mode == "alltoall":
            wd.phase("cross-check of the transposed decomposition against the band decomposition")
            crosscheck = crosscheck_against_bands(dtr, DistributedTrans)
            if not crosscheck["bitwise_equal_on_all_ranks"] and impl == "native":
                from atlas_amd.dist_torch import DistributedTrans
                impl_note = "the library's driver failed the cross-check against the band decomposition: " + \
                            crosscheck.get("error_on_rank0", "results differ")
                impl, dtr = "torch (fallback)", DistributedTrans(g, TRUNC, profile=True, mode=mode)
                tr = dtr.trans
                crosscheck = crosscheck_against_bands(dtr, DistributedTrans)
        gps = [gp] * world
        # the library's driver takes the spectra scattered by m (every rank holds only the blocks of its own
        # wavenumbers, 1/P of the 1.8 GB): that is the timed input; the replicated arrays above feed the cross-checks
        shards, sharded_note = None, None
        if impl == "native" and dtr.mode == "alltoall" and hasattr(dtr, "invtrans_many_sharded"):
            wd.phase("m-sharded spectra: bitwise check against the replicated input")
            try:
                shards = [torch.from_numpy(dtr.shard_spectra(nf, s_.cpu().numpy())).to(DEVICE) for s_ in sps]
                gp_r, gp_s = torch.empty_like(gp), torch.empty_like(gp)
                dtr.invtrans(nf, sps[0], gp_r)
                dtr.invtrans_many_sharded(nf, [shards[0]], [gp_s])
                sync()
                ok = int(bool(torch.equal(gp_r, gp_s)) and bool(torch.isfinite(gp_s).all()))
                del gp_r, gp_s
            except Exception as e:
                sys.stderr.write(f"[bench] rank {rank}: sharded-input entry point failed: {type(e).__name__}: {e}\n")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=DEVICE)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                shards, sharded_note = None, "sharded-input entry point disagreed with the replicated one: replicated input timed"
            else:
                sharded_note = "input spectra scattered by m (1/P of the coefficients per rank), bitwise equal to the replicated call"

        def step_alltoall():
            # `world` transforms per step, software-pipelined: the exchange of transform i overlaps the Legendre
            # stage of transform i+1 and the Fourier stage of transform i-1
            if shards is not None:
                dtr.invtrans_many_sharded(nf, [shards[i % len(shards)] for i in range(world)], gps)
            else:
                dtr.invtrans_many(nf, [sps[i % len(sps)] for i in range(world)], gps)

        def barrier():
            sync()
            dist.barrier()
            sync()

        def timed_steps(fn, n):
            """n steps of fn between barriers; seconds, MAX over ranks"""
            barrier()
            t_ = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            t_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=DEVICE)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())

        # ---- the exchange-free mirror bands (a northern band of rows and its mirror image per GPU), usable only if every
        # rank reproduces, bit for bit, the same rows through the GPU-tested zonal-band crop path
        dm, gp_m, mirror_ok = None, None, False
        if world > 1 and dtr.mode == "alltoall" and not args.no_alt:
            wd.phase("mirror-band decomposition: set-up and bitwise self-check")
            ok = 0
            try:
                dm = DistributedTrans(g, TRUNC, profile=True, mode="mirror")
                b0, b1 = dm.trans.mirror_rows()
                gp_m = torch.empty(nf * dm.trans.nb_gridpoints(), dtype=torch.float64, device=DEVICE)
                dm.invtrans(nf, sps[0], gp_m)
                ok, first = 1, 0
                for j0, j1 in ((b0, b1), (g.ny() - b1, g.ny() - b0)):
                    tc = atlas_amd.Trans(g, TRUNC, rows=(j0, j1))
                    n = tc.nb_gridpoints()
                    gp_c = torch.empty(nf * n, dtype=torch.float64, device=DEVICE)
                    tc.invtrans(nf, sps[0], gp_c)
                    tc.synchronize()
                    sync()
                    same = torch.equal(gp_m.view(nf, -1)[:, first:first + n], gp_c.view(nf, -1))
                    ok = ok if (same and bool(torch.isfinite(gp_c).all()) and float(gp_c.abs().max()) > 0.0) else 0
                    first += n
                    del tc, gp_c
                ok = ok if first * nf == gp_m.numel() else 0
            except Exception as e:   # any failure means: not used, not reported
                sys.stderr.write(f"[bench] rank {rank}: mirror-band self-check failed: {type(e).__name__}: {e}\n")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=DEVICE)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            mirror_ok = bool(int(flag.item()))

        def step_mirror():
            dm.invtrans_many(nf, [sps[i % len(sps)] for i in range(world)], [gp_m] * world)

        # ---- `--dist-mode auto`: which decomposition is TIMED is decided by an untimed trial of both on this
        # node -- the wavenumber-sharded one where its exchange hides behind the two stages, the mirror bands where it does not
        # -- and the line says which and why; the other one is reported as `alt_decomposition`.  The model (per-rank stage costs
        # measured in the trial, busiest pair's bytes, link rate from --xgmi-gbs or the all-pairs probe) is printed beside it.
        timed_mode, auto_choice, links = dtr.mode, None, None
        if on_gpu and world > 1:
            try:
                wd.phase("link probe (all pairs at once)")
                links = link_probe(world, DEVICE)
            except Exception as e:
                sys.stderr.write(f"[bench] rank {rank}: link probe failed: {type(e).__name__}: {e}\n")
        if args.dist_mode == "auto" and world > 1 and dtr.mode == "alltoall" and mirror_ok:
            wd.phase("trial of the two decompositions")
            nt = max(1, args.trial_steps)
            step_alltoall()
            step_mirror()
            tr.timings(reset=True)
            if hasattr(dtr, "exchange_timings"):
                dtr.exchange_timings(reset=True)
            t_a2a = timed_steps(step_alltoall, nt) / nt * 1e3
            tma = tr.timings()
            xta = dtr.exchange_timings(reset=True) if hasattr(dtr, "exchange_timings") else None
            t_mir = timed_steps(step_mirror, nt) / nt * 1e3
            # per transform and rank, slowest rank (a step is `world` transforms)
            mine = torch.tensor([tma["legendre_ms"] / max(tma["legendre_calls"], 1), tma["fourier_ms"] / max(tma["fourier_calls"], 1),
                                 (xta or {}).get("pack_ms", 0.0), float((xta or {}).get("bytes_to_busiest_peer", 0))],
                                dtype=torch.float64, device=DEVICE)
            dist.all_reduce(mine, op=dist.ReduceOp.MAX)
            rate = args.xgmi_gbs if args.xgmi_gbs is not None else ((links or {}).get("gbs_per_pair_and_direction") or 50.0)
            model = predict_decomposition(world, float(mine[0]), float(mine[1]), float(mine[2]), float(mine[3]), rate,
                                          mirror_ms=t_mir / world)
            timed_mode = "alltoall" if t_a2a <= t_mir else "mirror"
            auto_choice = {"timed": timed_mode, "decided_by": f"untimed trial, {nt} steps of each decomposition after one warm-up step",
                           "trial_ms_per_step": {"alltoall": t_a2a, "mirror": t_mir},
                           "why": (f"trial: m-sharded Legendre + RCCL transposition {t_a2a:.2f} ms per step, mirror bands {t_mir:.2f} ms; "
                                   f"model: {model['why']}"),
                           "model": model,
                           "link_rate_source": "--xgmi-gbs" if args.xgmi_gbs is not None else "all-pairs probe of this run"}
        if auto_choice is not None and timed_mode == "mirror":
            tr, gp = dm.trans, gp_m
            step = step_mirror
        else:
            step = step_alltoall

    wd.phase("warm-up steps")
    for _ in range(args.warmup):
        step()
    barrier()
    wd.phase("timed steps")
    # `--blocks` blocks of exactly `--steps` steps, each bracketed by barrier + synchronize; the block with the MEDIAN time is
    # the one `value`, `ms_per_step` and the per-stage times describe (reference benchmark: min / max / iterations,
    # atlas-benchmark-trans.cc:257-289).  The device state is sampled while the blocks run.
    sampler = DeviceSampler(local_rank) if (on_gpu and rank == 0) else None
    state_before = sampler.snapshot() if sampler else None
    blocks = []
    if sampler:
        sampler.start()
    xt_blocks = []
    for _ in range(max(1, args.blocks)):
        tr.timings(reset=True)
        if use_dist and timed_mode == "alltoall" and hasattr(dtr, "exchange_timings"):
            dtr.exchange_timings(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        bdt = time.perf_counter() - t0
        if use_dist:
            import torch.distributed as dist
            tmax = torch.tensor([bdt], dtype=torch.float64, device=DEVICE)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            bdt = float(tmax.item())
        blocks.append((bdt, tr.timings()))
        xt_blocks.append(dtr.exchange_timings(reset=True) if (use_dist and timed_mode == "alltoall" and hasattr(dtr, "exchange_timings"))
                         else None)
    if sampler:
        sampler.stop()
    state_after = sampler.snapshot() if sampler else None
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    dt, tm = blocks[order[(len(blocks) - 1) // 2]]   # the median block (the lower one of an even count)
    xt = xt_blocks[order[(len(blocks) - 1) // 2]]
    wd.phase("after the timed region (alternative decomposition, report)")
    transforms = args.steps * world
    ms_per_step = dt / args.steps * 1e3

    alt = None
    per_rank = None
    if use_dist and world > 1:
        import torch.distributed as dist
        # the decomposition that was NOT timed, for comparison (same steps, barriers, MAX over ranks), in a short run
        if dtr.mode == "alltoall" and not args.no_alt:
            other = "mirror" if timed_mode == "alltoall" else "alltoall"
            alt = {"mode": other}
            if other == "mirror":
                alt["selfcheck_bitwise_equal_on_all_ranks"] = mirror_ok
            if other == "alltoall" or mirror_ok:
                asteps = max(1, min(args.steps, 5))
                fn = step_mirror if other == "mirror" else step_alltoall
                fn()
                adt = timed_steps(fn, asteps)
                alt.update({"value": asteps * world / adt, "unit": "transforms/s", "steps": asteps, "ms_per_step": adt / asteps * 1e3,
                            "note": ("exchange-free: every GPU transforms a northern band of rows and its mirror image; output "
                                     "is two row ranges per GPU, not Atlas's contiguous bands") if other == "mirror" else
                                    "m-sharded Legendre + RCCL m -> latitude transposition + latitude-band FFT: output in Atlas's bands"})
        # where the curve bends: every rank's stage times of the median block, per transform
        mine = {"rank": rank, "legendre_ms": tm["legendre_ms"] / max(tm["legendre_calls"], 1),
                "fourier_ms": tm["fourier_ms"] / max(tm["fourier_calls"], 1)}
        if xt:
            mine.update({"pack_ms": xt["pack_ms"], "exchange_ms": xt["exchange_ms"], "bytes_sent": xt["bytes_sent"],
                         "bytes_received": xt["bytes_received"], "bytes_to_busiest_peer": xt["bytes_to_busiest_peer"],
                         "peers": xt["peers"],
                         "achieved_gbs_busiest_link": (xt["bytes_to_busiest_peer"] / (xt["exchange_ms"] * 1e-3) / 1e9
                                                       if xt["exchange_ms"] > 0 else None),
                         "achieved_gbs_all_links_out": (xt["bytes_sent"] / (xt["exchange_ms"] * 1e-3) / 1e9
                                                        if xt["exchange_ms"] > 0 else None)})
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    if dm is not None:
        del dm

    if rank == 0:
        leg_ms = tm["legendre_ms"] / max(tm["legendre_calls"], 1)
        fft_ms = tm["fourier_ms"] / max(tm["fourier_calls"], 1)
        # algorithmic work per launch (DESIGN.md "Kernels"): Legendre flops of SURVEY 8(d) / world (m-sharding);
        # Fourier bytes = kept part of the intermediate read once + grid-point output written once
        if use_dist and timed_mode == "mirror":
            # rank 0 owns rows [0, b1) and their mirror images: the geometry of its object is exactly its share
            b1 = tr.mirror_rows()[1]
            leg_flops = tr.legendre_flops(nf)
            kept_modes = float(sum(2 * max(0, b1 - int(v)) for v in tr.nlat0()))
            fft_bytes = kept_modes * nf * 16 + nf * tr.nb_gridpoints() * 8
        else:
            leg_flops = tr.legendre_flops(nf) / world
            kept_modes = float(sum(int(tr.nlat0()[m] < g.ny() // 2) * 2 * (g.ny() // 2 - int(tr.nlat0()[m]))
                                   for m in range(TRUNC + 1)))          # (lat, m) pairs with data
            fft_bytes = (kept_modes * nf * 16 + nf * g.size() * 8) / world
        leg_tf = leg_flops / (leg_ms * 1e-3) / 1e12 if leg_ms > 0 else 0.0
        fft_gbs = fft_bytes / (fft_ms * 1e-3) / 1e9 if fft_ms > 0 else 0.0
        traffic = measured_traffic() if world == 1 and not use_dist else {}   # PMC passes: the single-GPU workload
        kernels = [
            # one launch per transform: the single largest kernel of the path (rocprofv3 --stats agrees, profiles/)
            {"kernel": "legendre_kernel_lean" if NLEV == 137 and os.environ.get("ATLAS_AMD_LEG_KERNEL", "lean") == "lean"
                       else "legendre_kernel", "launches_per_transform": 1, "bound": "mfma", "achieved": leg_tf,
             "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": leg_tf / FP64_MFMA_PEAK_TFLOPS,
             "avg_ms": leg_ms, "traffic": traffic.get("legendre_kernel")},
            # the Fourier stage is one launch per row-length class (fft_rows_ct_kernel<CtShape<F,K>> /
            # fft_rows_kernel); bytes and time are those of the whole stage of one transform
            {"kernel": "fft_rows_ct_kernel<*> (Fourier stage, all row classes)",
             "launches_per_transform": int(tm.get("fourier_launches", 0)) or None, "bound": "hbm",
             "achieved": fft_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fft_gbs / HBM_PEAK_GBS,
             "avg_ms": fft_ms, "traffic": traffic.get("fourier_stage")},
        ]
        if world == 1 and not use_dist:
            # what the part sustains on fp64 MFMA alone, measured right after the timed region (same thermal state)
            try:
                from atlas_amd import _lib
                sustained = _lib.diag_mfma_f64_rate(25.0, 3)
                kernels[0]["peak_sustained_measured"] = sustained
                kernels[0]["frac_of_sustained"] = leg_tf / sustained if sustained > 0 else None
            except Exception as e:   # a measurement aid: never fails the bench
                sys.stderr.write(f"[bench] sustained MFMA rate not measured: {type(e).__name__}: {e}\n")
            if on_gpu:
                try:
                    kernels[1]["hbm_copy_measured"] = measured_hbm_copy_rate(DEVICE)
                    kernels[1]["frac_of_hbm_copy_measured"] = fft_gbs / kernels[1]["hbm_copy_measured"]
                except Exception as e:
                    sys.stderr.write(f"[bench] HBM copy rate not measured: {type(e).__name__}: {e}\n")
        # the bound that actually applies to the Fourier stage is its own vector-ALU instruction stream (counter traffic
        # 1.04 x algorithmic, VALU the busiest unit): floor = wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz
        valu = measured_valu_instructions() if world == 1 and not use_dist else {}
        if valu.get("fourier_stage"):
            floor_ms = valu["fourier_stage"] * 4.0 / 1024.0 / 2.4e9 * 1e3
            kernels[1]["valu_wave_instructions"] = valu["fourier_stage"]
            kernels[1]["valu_issue_floor_ms"] = floor_ms
            kernels[1]["frac_of_valu_issue_floor"] = floor_ms / fft_ms if fft_ms > 0 else None
            kernels[1]["valu_source"] = f"profiles/{valu['_profile']} (rocprofv3 --pmc SQ_INSTS_VALU, same kernel sources)"
        # headline roofline: of the kernels / stages that take at least a quarter of the step, the one FURTHEST from its
        # bound -- not the most flattering one
        stage_ms = leg_ms + fft_ms
        candidates = [k for k in kernels if stage_ms > 0 and k["avg_ms"] >= 0.25 * stage_ms] or kernels
        dominant = min(candidates, key=lambda k: k["frac"])
        out = {
            "metric": "inverse SH transforms/sec (TL1279, O1280, 137 lev)",
            "value": transforms / dt, "unit": "transforms/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"TransLocal invtrans TL{TRUNC} -> {GRID}, {NLEV} levels (nb_scalar_fields={NLEV}) "
                                   f"per transform, {world} transform(s) per step",
                       "grid": GRID, "truncation": TRUNC, "levels": NLEV,
                       "parallelism": "single GPU" if not use_dist else (
                           f"m-sharded Legendre + RCCL m->latitude transposition + latitude-band FFT over {world} GPUs"
                           if timed_mode == "alltoall" else
                           f"mirror-band sharding of both stages over {world} GPUs (a northern band of rows and its mirror "
                           f"image per GPU: no exchange, hemisphere symmetry kept)" if timed_mode == "mirror" else
                           f"latitude-band sharding of both stages over {world} GPUs (no exchange; Legendre rows of a "
                           f"band are computed without their mirror hemisphere: 2/P of the single-GPU Legendre work)")},
            "roofline": {k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
            "roofline_kernels": kernels,
            "repeats": {"blocks": len(blocks), "steps_per_block": args.steps, "value_is": "median block",
                        "values": [transforms / b[0] for b in blocks],
                        "ms_per_step": [b[0] / args.steps * 1e3 for b in blocks],
                        "min": transforms / max(b[0] for b in blocks), "max": transforms / min(b[0] for b in blocks),
                        "legendre_ms": [b[1]["legendre_ms"] / max(b[1]["legendre_calls"], 1) for b in blocks],
                        "fourier_ms": [b[1]["fourier_ms"] / max(b[1]["fourier_calls"], 1) for b in blocks]},
            "clocks": sampler.summary(state_before, state_after) if sampler else None,
        }
        # what THIS box delivers on the two bare resources, measured in this run (boxes of the pool differ by up to 10 % on the
        # same build: profiles/r06_ab_r4_vs_head.txt, r06_ab_prefetch.txt)
        out["box"] = {"mfma_f64_sustained_tflops": kernels[0].get("peak_sustained_measured"),
                      "hbm_copy_gbs": kernels[1].get("hbm_copy_measured")}
        out["roofline"]["kernel"] = dominant["kernel"]
        out["roofline"]["avg_ms"] = dominant["avg_ms"]
        for k in ("peak_sustained_measured", "frac_of_sustained", "valu_issue_floor_ms", "frac_of_valu_issue_floor",
                  "valu_wave_instructions", "hbm_copy_measured", "frac_of_hbm_copy_measured"):
            if k in dominant:
                out["roofline"][k] = dominant[k]
        out["roofline"]["traffic_source"] = (f"profiles/{traffic['_profile']} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same "
                                             f"kernel sources)" if traffic.get("_profile") else
                                             ("none: the committed PMC passes profile the single-GPU workload, not the distributed one"
                                              if (world > 1 or use_dist) else
                                              "none: no committed PMC profile matches these kernel sources"))
        if use_dist and world > 1:
            out["per_rank"] = per_rank
            out["per_rank_note"] = ("stage times per transform of the median block on every rank: legendre / fourier on the Trans "
                                    "stream, pack / exchange on the communication stream (exchange = the send / receive group incl. "
                                    "waiting for the peers); achieved_gbs_busiest_link = bytes to the busiest peer / exchange_ms")
            out["link_probe"] = links
            if auto_choice is not None:
                out["auto_choice"] = auto_choice
                out["config"]["parallelism"] += "; chosen by --dist-mode auto: " + auto_choice["why"]
        if crosscheck is not None:
            out["multi_gpu_crosscheck"] = crosscheck
        if alt is not None:
            out["alt_decomposition"] = alt
        if use_dist:
            # which driver was TIMED: "native" = the library's (C++ / RCCL, csrc/dist_trans.hip); anything else means the
            # library path did not run and this line must not be read as its measurement
            out["dist_impl"] = impl
            out["native_failed"] = bool(impl.startswith("torch (fallback)"))
            if impl_note is not None:
                out["dist_impl_note"] = impl_note
            if out["native_failed"]:
                # a multi-GPU line measures csrc/dist_trans.hip or nothing: what the torch.distributed
                # stand-in reached is kept beside it for diagnosis, never as `value`
                out["fallback_value"], out["value"] = out["value"], None
                out["fallback_ms_per_step"] = out["ms_per_step"]
                out["error"] = ("the library's distributed driver (csrc/dist_trans.hip over RCCL) did not run or failed its "
                                "cross-check; the timed region ran atlas_amd/dist_torch.py instead -- value withheld"
                                + (": " + impl_note if impl_note else ""))
        if use_dist and sharded_note is not None:
            out["config"]["input_spectra"] = sharded_note
        if os.environ.get("BENCH_TEST_STANDIN") == "1":
            out["metric"] = "STAND-IN TRANSFORM (driver-logic test over gloo, not a measurement): " + out["metric"]
            out["test_standin"] = True
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            out["cpu_baseline"], field_ref, sp_ref = cpu_baseline_blas(args.cpu_sample_fields, keep_field=True)
            # full-field parity of the transform that was just timed against the field the CPU baseline computed (default:
            # all 137 levels of the same synthetic spectra): a fast kernel whose results differ is not a result
            out["parity"] = full_field_parity(tr, g, args.cpu_sample_fields, field_ref, sp_ref, DEVICE)
            del field_ref, sp_ref
            if on_gpu and not out["parity"]["ok"]:
                out["error"] = (f"parity: rel_rms {out['parity']['rel_rms']:.3e} against the CPU restatement exceeds "
                                f"{PARITY_TOL:g}: the timed value is withheld")
                out["value_withheld"], out["value"] = out["value"], None
            if args.cpu_baseline_naive:
                out["cpu_baseline_naive"] = cpu_baseline(min(args.cpu_sample_fields, 24))
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    wd.done()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio, which would
        # otherwise be flushed after Python's line at exit
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
