"""Dev tool (GPU box): Fourier-stage time and result difference between two settings of an environment switch
(read when a Trans is constructed), e.g.  python tools/fft_ab.py ATLAS_AMD_FFT_HYBRID 0 1 [grid T nf]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
var, v0, v1 = sys.argv[1], sys.argv[2], sys.argv[3]
grid, T, nf = (sys.argv[4], int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else ("O1280", 1279, 137)
g = atlas_amd.Grid(grid)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
outs = []
for v in (v0, v1):
    os.environ[var] = v
    tr = atlas_amd.Trans(g, T, profile=True, tables="device")
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    for _ in range(3):
        tr.invtrans(nf, sp, gp)
    tr.synchronize()
    tr.timings(reset=True)
    for _ in range(5):
        tr.invtrans(nf, sp, gp)
    tr.synchronize()
    tm = tr.timings()
    fc, lc = max(tm["fourier_calls"], 1), max(tm["legendre_calls"], 1)
    print(f"{var}={v}: fourier {tm['fourier_ms'] / fc:.3f} ms  legendre {tm['legendre_ms'] / lc:.3f} ms per call ({fc} calls)", flush=True)
    outs.append(gp.clone())
    del tr
d = (outs[0] - outs[1]).abs().max().item()
s = outs[0].abs().max().item()
print(f"max |diff| {d:.3e}  (max |value| {s:.3e})  rel {d / s:.3e}")
