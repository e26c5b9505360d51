#!/bin/bash
# Dev tool (GPU box): store side of Fourier-intermediate layouts with several wavenumbers of one field per 128-byte line
# (dev build -DAA_LEG_LAYOUT_PROBE; the Fourier results are unusable): Legendre ms per call for p.abl = 0 (product layout), 1, 2, 3
export ATLAS_AMD_LIB=$PWD/atlas_amd/lib/dev/libatlas_amd_lp.so
for rep in 1 2; do for v in 0 1 2 3; do
ATLAS_AMD_LEG_LAYOUT_PROBE=$v python - <<PY
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, atlas_amd
from helpers import red_spectra
g = atlas_amd.Grid("O1280"); T, nf = 1279, 137
tr = atlas_amd.Trans(g, T, profile=True)
sp = torch.from_numpy(red_spectra(T, nf)).cuda(); gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
for _ in range(3): tr.invtrans(nf, sp, gp)
tr.synchronize(); tr.timings(reset=True)
for _ in range(6): tr.invtrans(nf, sp, gp)
tr.synchronize(); tm = tr.timings()
print("layout probe", os.environ["ATLAS_AMD_LEG_LAYOUT_PROBE"], "legendre ms", round(tm["legendre_ms"] / tm["legendre_calls"], 3), "fourier ms", round(tm["fourier_ms"] / tm["fourier_calls"], 3))
PY
done; done
