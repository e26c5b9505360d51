"""Experiment (round 5, experiments build): two jobs per workgroup in sequence (tools/experiments/fft_ct_rows_seq.inc) against the
product rows: bitwise comparison of the whole field and Fourier stage times.
    ATLAS_AMD_LIB=atlas_amd/lib/dev/libatlas_amd_exp.so python tools/experiments/fft_seq_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, atlas_amd
from helpers import red_spectra
T, nf = 1279, 137
g = atlas_amd.Grid("O1280")
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
def run(env, steps=10):
    for k, v in env.items():
        os.environ[k] = v
    tr = atlas_amd.Trans(g, T, profile=True)
    tr.use_torch_stream()
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    for _ in range(3):
        tr.invtrans(nf, sp, gp)
    torch.cuda.synchronize()
    tr.timings(reset=True)
    for _ in range(steps):
        tr.invtrans(nf, sp, gp)
    torch.cuda.synchronize()
    tm = tr.timings()
    for k in env:
        del os.environ[k]
    return gp, tm["fourier_ms"] / tm["fourier_calls"], tm["legendre_ms"] / tm["legendre_calls"]
ref, f0, l0 = run({})
print(f"product: Fourier {f0:.3f} ms, Legendre {l0:.3f}")
for env in ({"ATLAS_AMD_FFT_SEQ": "1", "ATLAS_AMD_FFT_PREFETCH": "0"}, {"ATLAS_AMD_FFT_PREFETCH": "0"}, {"ATLAS_AMD_FFT_SEQ": "1"}, {}):
    for rep in range(2):
        gp, f1, l1 = run(env)
        print(env, f"Fourier {f1:.3f} ms, bitwise equal to the product: {bool(torch.equal(gp, ref))}", flush=True)
