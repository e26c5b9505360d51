"""Dev probe (GPU box): the Legendre stage of transform i+1 next to the Fourier stage of transform i (two streams, two Fourier
intermediates), TL1279 -> O1280, 137 levels.  Run once per library (ATLAS_AMD_LIB=<dev build>): the product library lets either
kernel fill a CU (2 + 0 or 0 + 2 workgroups); the -DAA_COEX build compiles the Legendre kernel to 136 registers (one workgroup per
CU) and the 256-register Fourier kernels to 240, so that every CU holds one workgroup of each (matrix cores and vector ALU of
the same CU busy at the same time).  Prints ms per transform of the serial order and of the overlapped order; checks the
overlapped results bitwise against invtrans()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from atlas_amd import _lib
from atlas_amd.trans import _ptr
import ctypes as C
from helpers import red_spectra

grid, T, nf = os.environ.get("COEX_GRID", "O1280"), int(os.environ.get("COEX_T", "1279")), int(os.environ.get("COEX_NF", "137"))
n_rep = int(os.environ.get("COEX_N", "16"))
g = atlas_amd.Grid(grid)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
trL, trF = atlas_amd.Trans(g, T), atlas_amd.Trans(g, T)
F = [torch.zeros(trL.fourier_size(nf), dtype=torch.float64, device="cuda") for _ in range(2)]
gp = [torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda") for _ in range(2)]
ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
trL.invtrans(nf, sp, ref)
trL.synchronize()
torch.cuda.synchronize()


def legendre(tr, Fb):
    _lib.check(_lib.Trans_legendre_device(tr._h, T, nf, _ptr(sp), _ptr(Fb)))


def fourier(tr, Fb, out):
    bases = (C.c_void_p * 1)(_ptr(Fb))
    cnts = (C.c_int * 1)(T + 1)
    _lib.check(_lib.Trans_fourier_device(tr._h, nf, 0, bases, cnts, _ptr(out)))


def run(n, overlap, prio):
    pl, pf = {"none": (0, 0), "leg": (-1, 0), "fft": (0, -1)}[prio]
    SL, SF = torch.cuda.Stream(priority=pl), torch.cuda.Stream(priority=pf)
    trL.set_stream(SL.cuda_stream)
    trF.set_stream(SF.cuda_stream)
    evL = [torch.cuda.Event() for _ in range(n)]
    evF = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        b = i % 2
        back = 2 if overlap else 1
        if i >= back:
            SL.wait_event(evF[i - back])      # overlap: the Fourier stage of transform i-2 has read F[b]; serial: transform i-1 is complete
        legendre(trL, F[b])
        evL[i].record(SL)
        SF.wait_event(evL[i])
        fourier(trF, F[b], gp[b])
        evF[i].record(SF)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    for overlap, prio in [(False, "none"), (True, "none"), (True, "leg"), (True, "fft")]:
        run(3, overlap, prio)
        ms = run(n_rep, overlap, prio)
        ok = torch.equal(gp[0], ref) and torch.equal(gp[1], ref)
        print(f"{os.environ.get('ATLAS_AMD_LIB', 'product'):40s} {'overlap' if overlap else 'serial ':8s} prio={prio:5s}: {ms:7.3f} ms per transform = {1e3 / ms:6.1f} transforms/s  bitwise={ok}", flush=True)
