"""Dev tool (GPU box): the host-pointer entry point (host arrays in / out, what atlas__Trans__invtrans_scalar callers pass)
at TL1279 -> O1280, 137 levels: wall time and PCIe rate with and without the pinned staging pipeline, result compared bit
for bit with the device-pointer path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
grid, T, nf = "O1280", 1279, 137
g = atlas_amd.Grid(grid)
sp = red_spectra(T, nf)
ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
tr = atlas_amd.Trans(g, T)
tr.invtrans(nf, torch.from_numpy(sp).cuda(), ref)
tr.synchronize()
ref = ref.cpu().numpy()
del tr
nbytes = sp.nbytes + ref.nbytes
SWEEP = [("0", "16", "8"), ("1", "16", "8")]
if "--sweep" in sys.argv:
    SWEEP += [("1", c, t) for t in ("8", "12", "16", "24") for c in ("16", "24", "32", "40")]
for pipe, chunk, threads in SWEEP:
    os.environ["ATLAS_AMD_HOST_PIPELINE"] = pipe      # (all three are read per call by the library)
    os.environ["ATLAS_AMD_HOST_CHUNK"] = chunk
    os.environ["ATLAS_AMD_HOST_THREADS"] = threads
    tr = atlas_amd.Trans(g, T)
    gp = np.zeros(nf * g.size())
    tr.invtrans(nf, sp, gp)
    ts = []
    for _ in range(4):
        gp[:] = 0
        t0 = time.perf_counter()
        tr.invtrans(nf, sp, gp)
        ts.append(time.perf_counter() - t0)
    print(f"ATLAS_AMD_HOST_PIPELINE={pipe} chunk={chunk if pipe == '1' else '-'} threads={threads if pipe == '1' else '-'}: {min(ts) * 1e3:.1f} ms per transform (host arrays; {sorted(round(t * 1e3, 1) for t in ts)}), "
          f"{nbytes / min(ts) / 1e9:.1f} GB/s over PCIe (1.8 GB up + 7.2 GB down), bitwise equal to the device path: "
          f"{np.array_equal(gp, ref)}", flush=True)
    del tr

# the vor/div call shape (atlas__Trans__invtrans: nb_scalar + nb_vordiv, TransInterface.h:74-79) through host arrays
if "--vordiv" in sys.argv:
    ns = nvd = 137
    vor, div = red_spectra(T, nvd, seed=2), red_spectra(T, nvd, seed=3)
    nb = sp.nbytes + vor.nbytes + div.nbytes + (ns + 2 * nvd) * g.size() * 8
    for pipe in ("0", "1"):
        os.environ.update({"ATLAS_AMD_HOST_PIPELINE": pipe, "ATLAS_AMD_HOST_CHUNK": "16", "ATLAS_AMD_HOST_THREADS": "8"})
        tr = atlas_amd.Trans(g, T)
        gpv = np.zeros((ns + 2 * nvd) * g.size())
        tr.invtrans(ns, sp, nvd, vor, div, gpv)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            tr.invtrans(ns, sp, nvd, vor, div, gpv)
            ts.append(time.perf_counter() - t0)
        print(f"vor/div call, nscalar 137 + nvordiv 137 (411 output fields), ATLAS_AMD_HOST_PIPELINE={pipe}: {min(ts) * 1e3:.1f} ms "
              f"({sorted(round(t * 1e3, 1) for t in ts)}), {nb / min(ts) / 1e9:.1f} GB/s over PCIe (5.4 GB up + 21.7 GB down)", flush=True)
        del tr
