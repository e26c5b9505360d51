"""Dev tool (GPU box): per-row comparison of the native mixed-radix rows with the Bluestein / direct plan of the same rows
(ATLAS_AMD_FFT_NATIVE=1 against 0): python tools/fft_native_rows.py [grid T nf]  -- prints the rows that differ and their stage lists"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, atlas_amd
from atlas_amd import _lib
from helpers import red_spectra
grid, T, nf = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("O320", 319, 19)
g = atlas_amd.Grid(grid)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
outs = []
for v in ("0", "1"):
    os.environ["ATLAS_AMD_FFT_NATIVE"] = v
    tr = atlas_amd.Trans(g, T)
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp, gp); tr.synchronize()
    outs.append(gp.cpu().numpy().reshape(nf, -1))
    cls = tr.fft_row_classes()
    del tr
nx = np.asarray(g.nx()); off = np.concatenate([[0], np.cumsum(nx)])
scale = np.abs(outs[0]).max()
bad = 0
for j in range(len(nx)):
    if cls[j, 2] != 4:
        continue
    d = np.abs(outs[0][:, off[j]:off[j + 1]] - outs[1][:, off[j]:off[j + 1]]).max(axis=1) / scale
    if d.max() > 1e-12:
        info = np.zeros(16, dtype=np.int32); _lib.check(_lib.fft_plan_info(int(nx[j]), 1, info.ctypes.data))
        bad += 1
        if bad <= 40:
            print(f"row {j} n={nx[j]} h={nx[j] // 2} radices(DIF)={list(info[4:4 + info[3]])} pitch={info[13]} worst field {int(d.argmax())} err {d.max():.2e} fields bad {(d > 1e-12).sum()}/{nf}")
print("native rows", int((cls[:, 2] == 4).sum()), "bad", bad)
