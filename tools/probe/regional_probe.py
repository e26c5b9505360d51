"""Probe (GPU box) [r6]: the no_nest / unstructured branch (atlas_amd__RegionalTrans__*): time per call and the rate of its Fourier part
(a dense sum over wavenumbers per point: 4 flops per (point, field, wavenumber))   python tools/probe/regional_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402

for T, nlon, nlat, nf in ((159, 200, 100, 10), (639, 500, 300, 20), (639, 1000, 500, 137), (1279, 1000, 500, 137)):
    lats = np.linspace(60.0, 30.0, nlat)
    t0 = time.perf_counter()
    rt = atlas_amd.RegionalTrans(nlon, -10.0, 0.05, lats, T)
    setup = time.perf_counter() - t0
    sp = torch.from_numpy(red_spectra(T, nf)).cuda()
    gp = torch.zeros(nf * nlon * nlat, dtype=torch.float64, device="cuda")
    rt.invtrans(nf, sp, gp)
    rt.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        rt.invtrans(nf, sp, gp)
    rt.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flops_f = 4.0 * nlon * nlat * nf * (T + 1)
    flops_l = 2.0 * nlat * (T + 1) * (T + 2) / 2 * 2 * nf
    print(f"T{T} {nlon} x {nlat} points, {nf} fields: set-up {setup:.2f} s, {dt * 1e3:.2f} ms per call "
          f"(Fourier part {flops_f / 1e9:.1f} GFLOP, Legendre part {flops_l / 1e9:.1f} GFLOP -> {(flops_f + flops_l) / dt / 1e12:.1f} TFLOP/s overall)", flush=True)
