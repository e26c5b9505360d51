"""Dev tool (GPU box): per-rank cost of the multi-GPU decompositions at TL1279 -> O1280, 137 levels, emulated on one
device (one rank's share at a time): latitude-band and mirror-band sharding (both stages local) and wavenumber sharding
(Legendre stage only; its Fourier stage equals the band one)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
T, nf = 1279, 137
g = atlas_amd.Grid("O1280")
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
for P, parts in ((2, (0,)), (4, (0, 1)), (8, (0, 3))):
    for part in parts:
        for shard in ("band", "mirror", "m"):
            tr = atlas_amd.Trans(g, T, profile=True, nparts=P, part=part, shard=shard)
            tr.use_torch_stream()
            if shard in ("band", "mirror"):
                gp = torch.zeros(nf * tr.nb_gridpoints(), dtype=torch.float64, device="cuda")
                run = lambda: tr.invtrans(nf, sp, gp)
            else:
                F = torch.zeros(tr.fourier_size(nf), dtype=torch.float64, device="cuda")
                run = lambda: tr.legendre_device(T, nf, sp, F)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            tr.timings(reset=True)
            t0 = time.perf_counter()
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5 * 1e3
            tm = tr.timings()
            print(f"P={P} part={part} shard={shard}: {dt:.2f} ms/transform  legendre {tm['legendre_ms']/max(tm['legendre_calls'],1):.2f} ms"
                  f"  fourier {tm['fourier_ms']/max(tm['fourier_calls'],1):.2f} ms", flush=True)
            del tr
