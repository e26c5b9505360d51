"""Probe (GPU box) [r6]: stage times over the field count at one grid -- looks for field counts whose tiling falls off the curve.
   python tools/probe/nf_sweep.py [grid=O1280] [T=1279] [f32]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402

grid = sys.argv[1] if len(sys.argv) > 1 else "O1280"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1279
f32 = "f32" in sys.argv
g = atlas_amd.Grid(grid)
tr = atlas_amd.Trans(g, T, profile=True)
tr.use_torch_stream()
blk = torch.from_numpy(red_spectra(T, 48)).cuda().reshape(-1, 48)
for nf in (1, 2, 4, 8, 9, 16, 17, 24, 25, 32, 33, 40, 48, 49, 64, 72, 96, 97, 120, 137, 144, 145, 192, 274):
    sp = blk.repeat(1, (nf + 47) // 48)[:, :nf].contiguous().reshape(-1)
    gp = torch.zeros(nf * g.size(), dtype=torch.float32 if f32 else torch.float64, device="cuda")
    if f32:
        sp = sp.float()
    for _ in range(2):
        tr.invtrans(nf, sp, gp)
    torch.cuda.synchronize()
    tr.timings(reset=True)
    reps = 5
    for _ in range(reps):
        tr.invtrans(nf, sp, gp)
    torch.cuda.synchronize()
    tm = tr.timings()
    L, F = tm["legendre_ms"] / reps, tm["fourier_ms"] / reps
    print(f"nf {nf:4d}: legendre {L:7.3f} ms ({L / nf * 1e3:7.1f} us/field)  fourier {F:7.3f} ms ({F / nf * 1e3:7.1f} us/field)", flush=True)
    del sp, gp
