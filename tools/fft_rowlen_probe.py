"""Dev tool (GPU box): Fourier-stage cost per grid point for one row length at a time -- regular grids of 256 Gaussian latitudes
whose rows all have the same length n: which of the row kernels (specialised direct, generic direct, specialised / generic
Bluestein) the length takes, and ps per point.   python tools/fft_rowlen_probe.py n1 n2 ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
T, nf, N = 639, 137, 128
y = atlas_amd.gaussian_latitudes(N)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
for n in [int(v) for v in sys.argv[1:]]:
    g = atlas_amd.StructuredGrid(nx=np.full(2 * N, n, dtype=np.int32), y=y)
    os.environ["ATLAS_AMD_FFT_DEBUG"] = "1"
    tr = atlas_amd.Trans(g, T, profile=True)
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    for _ in range(2):
        tr.invtrans(nf, sp, gp)
    tr.synchronize(); tr.timings(reset=True)
    for _ in range(5):
        tr.invtrans(nf, sp, gp)
    tr.synchronize()
    tm = tr.timings()
    ms = tm["fourier_ms"] / tm["fourier_calls"]
    print(f"n={n} h={n//2}: fourier {ms:.3f} ms = {ms * 1e9 / (nf * g.size()):.2f} ps/point", flush=True)
    del tr
