#!/bin/bash
# Dev tool (GPU box; dev build -DAA_FFT_ABLATE, results wrong by construction): Fourier stage with the gather's wavenumber index
# shifted right by 0 / 1 / 2 / 3 bits (1 / 2 / 4 / 8 lanes per 128-byte line of the intermediate) and with all lanes on one line
export ATLAS_AMD_LIB=$PWD/atlas_amd/lib/dev/libatlas_amd_abl.so
export ATLAS_AMD_FFT_NATIVE=0
for rep in 1 2; do for a in 0 64 128 256 1 32; do
ATLAS_AMD_FFT_ABLATE=$a python - <<PY
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
print("ablate", os.environ["ATLAS_AMD_FFT_ABLATE"], "fourier ms", round(tm["fourier_ms"] / tm["fourier_calls"], 3), "legendre ms", round(tm["legendre_ms"] / tm["legendre_calls"], 3))
PY
done; done
