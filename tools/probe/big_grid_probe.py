"""Probe (GPU box) [r6]: resolutions beyond the benchmark's -- TL1999 -> O2000 (16 M points; the longest row 8016 points: Bluestein length 8192,
half of a CU's LDS ... ) and what the library says to O2560 (rows beyond 10 240 complex LDS elements).  Set-up time, time per call, sampled rows
of every Fourier class against the oracle.   python tools/probe/big_grid_probe.py [O2000 1999 nf]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
import oracle  # noqa: E402
from helpers import compute_rms, red_spectra, rows_of_every_fft_class  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "O2000"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1999
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 16
g = atlas_amd.Grid(name)
print(f"{name}: {g.size()} points, {g.ny()} rows, longest {int(g.nx().max())}", flush=True)
t0 = time.time()
try:
    tr = atlas_amd.Trans(g, T)
except Exception as e:   # noqa: BLE001
    print("Trans(...) failed:", e)
    sys.exit(0)
print(f"set-up {time.time() - t0:.1f} s; Legendre table {tr.legendre_table_bytes() / 1e9:.1f} GB", flush=True)
sp_h = red_spectra(T, nf, seed=3)
sp = torch.from_numpy(sp_h).cuda()
gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
tr.invtrans(nf, sp, gp)
tr.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    tr.invtrans(nf, sp, gp)
tr.synchronize()
print(f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per call of {nf} fields; finite: {bool(torch.isfinite(gp).all())}", flush=True)
rows, _ = rows_of_every_fft_class(tr, extra=[0, g.ny() // 2 - 1, g.ny() - 1])
op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
off = np.concatenate([[0], np.cumsum(g.nx())])
v = gp.view(nf, -1)
worst = 0.0
t0 = time.time()
for r, ref in zip(rows, op.invtrans_rows(nf, sp_h, rows, use_fft=True)):
    worst = max(worst, compute_rms(v[:, off[r]:off[r + 1]].cpu().numpy(), ref))
print(f"{len(rows)} rows of every Fourier class against the oracle ({time.time() - t0:.0f} s): worst rel-RMS {worst:.2e}")
