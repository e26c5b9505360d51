"""Dev tool (GPU box): stage times for an arbitrary (grid, truncation, levels): python tools/bench_grid.py F1280 1279 137 [f32]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
grid, T, nf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"   # fp32 variant (BASELINE config C5)
g = atlas_amd.Grid(grid)
t0 = time.time()
tr = atlas_amd.Trans(g, T, profile=True)
print(f"{grid} T{T} nf={nf}: setup {time.time() - t0:.1f} s, npts {g.size()}", flush=True)
tr.use_torch_stream()
if nf <= 200:
    sp = torch.from_numpy(red_spectra(T, nf)).cuda()
else:   # many fields: tile a 137-field block on the device (host generation of 18 GB would take minutes)
    blk = torch.from_numpy(red_spectra(T, 137)).cuda().reshape(-1, 137)
    reps = (nf + 136) // 137
    sp = blk.repeat(1, reps)[:, :nf].contiguous().reshape(-1)
    del blk
gp = torch.zeros(nf * g.size(), dtype=torch.float32 if f32 else torch.float64, device="cuda")
if f32:
    sp = sp.to(torch.float32)
for _ in range(2):
    tr.invtrans(nf, sp, gp)
torch.cuda.synchronize()
tr.timings(reset=True)
t0 = time.perf_counter()
for _ in range(5):
    tr.invtrans(nf, sp, gp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5 * 1e3
tm = tr.timings()
if nf > 200:   # field k and field k + 137 carry the same spectra: results must be identical
    v = gp.reshape(nf, -1)
    print("tiled fields identical:", bool(torch.equal(v[3], v[3 + 137])), bool(torch.equal(v[136], v[nf - 1 - ((nf - 1) % 137) + 136 - 137] if nf % 137 == 0 else v[136 + 137])))
print(f"{dt:.2f} ms/transform  legendre {tm['legendre_ms']/tm['legendre_calls']:.2f} ms  fourier {tm['fourier_ms']/tm['fourier_calls']:.2f} ms")
