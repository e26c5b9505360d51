"""Dev probe (GPU box): throughput of two Trans objects on their own streams (Legendre stage of one next to the Fourier
stage of the other) against one object run back to back, TL1279 -> O1280, 137 levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
grid, T, nf = "O1280", 1279, 137
g = atlas_amd.Grid(grid)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
trs = [atlas_amd.Trans(g, T) for _ in range(2)]
gps = [torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
def run(n, two):
    for i in range(n):
        k = i % 2 if two else 0
        trs[k].invtrans(nf, sp, gps[k])
    for t in trs:
        t.synchronize()
    torch.cuda.synchronize()
for two in (False, True, False, True):
    run(4, two)
    t0 = time.perf_counter()
    n = 20
    run(n, two)
    dt = time.perf_counter() - t0
    print(f"{'two objects, two streams' if two else 'one object':26s}: {dt / n * 1e3:.2f} ms per transform, {n / dt:.1f} transforms/s", flush=True)
assert torch.equal(gps[0], gps[1])
