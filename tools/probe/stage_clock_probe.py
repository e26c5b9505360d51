"""Probe (GPU box) [r6]: the shader clock and socket power the firmware reports WHILE one kind of work runs alone for about a second --
the bare fp64 MFMA loop (csrc/diag.hip), the Legendre stage alone, the Fourier stage alone, the whole transform, an HBM copy -- sampled from
amdsmi gpu_metrics every 2 ms (bench.DeviceSampler).  Which unit the power limiter holds back, and by how much.
    python tools/probe/stage_clock_probe.py [seconds per phase]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
import bench  # noqa: E402
from atlas_amd import _lib  # noqa: E402
from helpers import red_spectra  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
T, nf = 1279, 137
g = atlas_amd.Grid("O1280")
tr = atlas_amd.Trans(g, T)
tr.use_torch_stream()
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
RP = tr.fourier_row_pitch(nf)
F = torch.zeros(g.ny() * (T + 1) * RP, dtype=torch.float64, device="cuda")
tr.invtrans(nf, sp, gp)
tr.legendre_device(T, nf, sp, F)
torch.cuda.synchronize()
a = torch.empty(1 << 28, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
flops = tr.legendre_flops(nf)


def phase(name, fn, unit_of):
    s = bench.DeviceSampler(0, period_s=0.002)
    fn()
    torch.cuda.synchronize()
    time.sleep(0.5)                      # let the clocks recover from the phase before
    s.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        n += 4
    dt = time.perf_counter() - t0
    s.stop()
    smp = s.samples[len(s.samples) // 4:]      # the settled part
    f = lambda k: float(np.mean([x[k] for x in smp if x.get(k) is not None])) if smp else float("nan")
    print(f"{name:34s} {dt / n * 1e3:8.3f} ms per call   {unit_of(dt / n):>22s}   sclk {f('sclk'):6.0f} MHz (slowest XCD {f('sclk_min_xcd'):6.0f})   "
          f"{f('power'):6.0f} W   mclk {f('mclk'):5.0f}", flush=True)


phase("bare fp64 MFMA loop (25 ms)", lambda: _lib.diag_mfma_f64_rate(25.0, 1), lambda t: "")
print("   (its own figure: %.1f TFLOP/s)" % _lib.diag_mfma_f64_rate(25.0, 3))
phase("Legendre stage alone", lambda: tr.legendre_device(T, nf, sp, F), lambda t: f"{flops / t / 1e12:.1f} TFLOP/s")
phase("Fourier stage alone", lambda: tr.fourier_device(nf, 0, [F], [T + 1], gp), lambda t: "")
phase("whole transform", lambda: tr.invtrans(nf, sp, gp), lambda t: f"{1 / t:.1f} transforms/s")
phase("device copy of 2 GiB", lambda: b.copy_(a), lambda t: f"{2 * a.numel() * 8 / t / 1e12:.2f} TB/s")
