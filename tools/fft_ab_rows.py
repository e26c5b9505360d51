"""Dev tool (GPU box): per-row difference between two settings of an environment switch (see fft_ab.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
var, v0, v1 = sys.argv[1], sys.argv[2], sys.argv[3]
grid, T, nf = (sys.argv[4], int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else ("O1280", 1279, 3)
g = atlas_amd.Grid(grid)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
outs = []
for v in (v0, v1):
    os.environ[var] = v
    tr = atlas_amd.Trans(g, T, tables="device")
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp, gp)
    tr.synchronize()
    outs.append(gp.cpu().numpy().reshape(nf, -1))
    del tr
nx = np.asarray(g.nx())
off = np.concatenate([[0], np.cumsum(nx)])
def awk(h):
    for p in (2, 3, 5):
        while h % p == 0:
            h //= p
    return h
bad = 0
for j in range(len(nx) // 2):
    d = np.abs(outs[0][:, off[j]:off[j + 1]] - outs[1][:, off[j]:off[j + 1]]).max()
    s = np.abs(outs[0][:, off[j]:off[j + 1]]).max()
    h = nx[j] // 2
    if d > 1e-11 * max(s, 1):
        bad += 1
        if bad < 40:
            print(f"row {j} nx {nx[j]} h {h} A {awk(h)} B {h // awk(h)} diff {d:.2e} of {s:.2e}")
print("bad rows (NH):", bad, "of", len(nx) // 2)
