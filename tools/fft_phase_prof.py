"""Dev tool (GPU box): per-phase cycle breakdown of the FFT kernel at TL1279/O1280."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from atlas_amd import _lib
from helpers import red_spectra
grid, T, nf = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("O1280", 1279, 137)
g = atlas_amd.Grid(grid)
tr = atlas_amd.Trans(g, T, profile=True)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
tr.invtrans(nf, sp, gp); tr.synchronize()
_lib.check(_lib.Trans_fft_phase_profile(tr._h, 1, None))
tr.timings(reset=True)
tr.invtrans(nf, sp, gp); tr.synchronize()
out = (C.c_ulonglong * 64)()
_lib.check(_lib.Trans_fft_phase_profile(tr._h, 0, out))
tm = tr.timings()
v = np.array(out[:], dtype=np.float64)
print("fourier ms", tm["fourier_ms"], "legendre ms", tm["legendre_ms"])
print("ATLAS_AMD_FFT_ONLY_M =", os.environ.get("ATLAS_AMD_FFT_ONLY_M"))
for name, sl in (("bluestein", slice(0, 32)), ("direct", slice(32, 64))):
    tot = v[sl].sum()
    print(name, "total Gcycles(thread0 sum; clock64 = 100 MHz ticks)", tot / 1e9)
    for i, x in enumerate(v[sl]):
        if x: print("   phase %2d: %6.2f %%" % (i, 100 * x / tot))
