"""Probe (GPU box) [r6]: the vor/div preparation kernel's two forms over the number of vor/div fields (ATLAS_AMD_PREPARE=rows | stream),
TL1279: is the switch-over (>= 48 fields) where it should be?   python tools/probe/prepare_sweep.py [T=1279] [grid=O1280]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1279
grid = sys.argv[2] if len(sys.argv) > 2 else "O1280"
g = atlas_amd.Grid(grid)
blk = torch.from_numpy(red_spectra(T, 32)).cuda().reshape(-1, 32)
for nvd in (1, 4, 8, 16, 24, 32, 40, 47, 48, 56, 64, 65, 96, 137):
    row = []
    for form in ("rows", "stream"):
        os.environ["ATLAS_AMD_PREPARE"] = form
        tr = atlas_amd.Trans(g, T, profile=True)
        tr.use_torch_stream()
        vor = blk.repeat(1, (nvd + 31) // 32)[:, :nvd].contiguous().reshape(-1)
        div = vor.clone()
        gp = torch.zeros(2 * nvd * g.size(), dtype=torch.float64, device="cuda")
        for _ in range(2):
            tr.invtrans(0, None, nvd, vor, div, gp)
        torch.cuda.synchronize()
        tr.timings(reset=True)
        for _ in range(5):
            tr.invtrans(0, None, nvd, vor, div, gp)
        torch.cuda.synchronize()
        tm = tr.timings()
        row.append(tm["prepare_ms"] / max(tm["prepare_calls"], 1))
        del tr, gp, vor, div
    print(f"nvd {nvd:4d}: rows {row[0]:7.4f} ms   stream {row[1]:7.4f} ms   -> {'stream' if row[1] < row[0] else 'rows'}", flush=True)
