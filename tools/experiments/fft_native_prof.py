"""Dev tool (GPU box): the native mixed-radix rows alone (ATLAS_AMD_FFT_ONLY_NATIVE=1) -- stage time and the per-phase shader-clock
breakdown of worker 0 (fft_native_impl.h: stamp): python tools/fft_native_prof.py [grid T nf]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ATLAS_AMD_FFT_ONLY_NATIVE"] = "1"
import numpy as np, torch, atlas_amd
from atlas_amd import _lib
from helpers import red_spectra
grid, T, nf = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("O1280", 1279, 137)
g = atlas_amd.Grid(grid)
tr = atlas_amd.Trans(g, T, profile=True)
cls = tr.fft_row_classes()
nat = cls[:, 2] == 4
nx = np.asarray(g.nx())
print(f"native rows {int(nat.sum())} of {len(cls)}, points {nx[nat].sum() / nx.sum():.4f} of the grid")
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
for _ in range(3):
    tr.invtrans(nf, sp, gp)
tr.synchronize()
tr.timings(reset=True)
for _ in range(5):
    tr.invtrans(nf, sp, gp)
tr.synchronize()
tm = tr.timings()
print("native rows only: fourier ms per call", tm["fourier_ms"] / max(tm["fourier_calls"], 1))
_lib.check(_lib.Trans_fft_phase_profile(tr._h, 1, None))
tr.invtrans(nf, sp, gp); tr.synchronize()
out = (C.c_ulonglong * 64)()
_lib.check(_lib.Trans_fft_phase_profile(tr._h, 0, out))
v = np.array(out[16:26], dtype=np.float64)
names = ["requests issued (gather, tables)", "wait: gather + tables landed", "fold: reads + arithmetic", "barrier", "fold: writes + twiddle requests",
         "barrier", "stage 0 (radix RL, no twiddles)", "barrier", "middle stages (+ barriers)", "last stage + store"]
jobs = int(nat.sum()) * nf
print(f"worker 0, mean per job ({jobs} jobs), clock64 ticks of 10 ns:")
for n_, x in zip(names, v):
    print(f"   {n_:36s} {x / jobs * 10:8.1f} ns  {100 * x / v.sum():5.1f} %")
print(f"   {'job':36s} {v.sum() / jobs * 10:8.1f} ns")
