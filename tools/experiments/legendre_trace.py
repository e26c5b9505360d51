"""Dev tool (GPU box) [r4]: per-wavefront phase times of the lean Legendre kernels.
Build (the trace is a patch, not part of the product sources):
    mkdir -p /tmp/tr && cp -r atlas_amd include tools /tmp/tr/ 2>/dev/null; cd /tmp/tr/atlas_amd/csrc && rm -rf build*
    patch legendre_kernel.hip < <repo>/tools/experiments/legendre_trace.patch
    make -j8 BUILD=build_lt LIBNAME=dev/libatlas_amd_legtrace.so && cp ../lib/dev/libatlas_amd_legtrace.so <repo>/atlas_amd/lib/dev/
Run:  ATLAS_AMD_LIB=$PWD/atlas_amd/lib/dev/libatlas_amd_legtrace.so python tools/experiments/legendre_trace.py [grid T nf f32]
Every wavefront of the first 131 072 records: prologue (start -> first stage in LDS, barrier passed), then per stage the time to the end
of its MFMA steps, the wait for the operands of the next stage (vmcnt), their LDS writes, the barrier; the epilogue up to the completion of
its stores.  Stamps are s_memtime + s_waitcnt lgkmcnt(0) at points where the hand-counted LDS counter is zero (each costs a scalar-memory
round trip, and the accumulators cost registers -- the fp32 kernel loses its third workgroup per CU: the traced kernel is slower than the
product; read the shares, not the totals)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402

argv = sys.argv[1:5] + ["F1280", "1279", "137", "1"][len(sys.argv) - 1:]
grid, T, nf, f32 = argv[0], int(argv[1]), int(argv[2]), int(argv[3]) != 0
lib = C.CDLL(os.environ["ATLAS_AMD_LIB"])
dump = lib.atlas_amd__leg_trace_dump
dump.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_int]
g = atlas_amd.Grid(grid)
tr = atlas_amd.Trans(g, T, profile=True)
tr.use_torch_stream()
dt = torch.float32 if f32 else torch.float64
sp = torch.from_numpy(red_spectra(T, nf)).cuda().to(dt)
gp = torch.zeros(nf * g.size(), dtype=dt, device="cuda")
for _ in range(2):
    tr.invtrans(nf, sp, gp)
torch.cuda.synchronize()
n = C.c_uint(0)
assert dump(None, C.byref(n), 1) == 0          # reset
tr.timings(reset=True)
tr.invtrans(nf, sp, gp)
torch.cuda.synchronize()
tm = tr.timings()
W = 12
buf = np.zeros((1 << 17) * W, dtype=np.uint64)
assert dump(buf.ctypes.data, C.byref(n), 0) == 0
rec = buf[: n.value * W].reshape(-1, W).astype(np.int64)
info = rec[:, 0]
active = (info & 1) == 1
nstage = (info >> 16) & 0xFFFF
m = info >> 32
life = (rec[:, 8] - rec[:, 1]) & 0xFFFFFFFF
print(f"{grid} T{T} nf={nf} {'fp32' if f32 else 'fp64'}: Legendre stage {tm['legendre_ms'] / max(tm['legendre_calls'], 1):.3f} ms (traced build); "
      f"{n.value} wavefront records, {int(active.sum())} with latitudes")
sel = active & (nstage >= 8)
names = ["prologue", "MFMA steps", "operand wait (vmcnt)", "LDS writes", "barrier", "epilogue"]
cols = [2, 3, 4, 5, 6, 7]
tot = life[sel].sum()
print(f"  wavefronts with >= 8 stages: {int(sel.sum())}, mean lifetime {life[sel].mean() / 100.0:.1f} us (100 MHz counter), mean stages {nstage[sel].mean():.1f}")
for nme, c in zip(names, cols):
    print(f"    {nme:24s} {100.0 * rec[sel, c].sum() / tot:5.1f} % of wavefront time   mean {rec[sel, c].mean() / 100.0:8.2f} us"
          + (f"   per stage {rec[sel, c].sum() / nstage[sel].sum() * 10.0:7.1f} ns" if c in (3, 4, 5, 6) else ""))
for lo, hi in ((0, 64), (64, 320), (320, 640), (640, 960), (960, 1280)):
    s2 = sel & (m >= lo) & (m < hi)
    if s2.sum():
        t2 = life[s2].sum()
        print(f"  m in [{lo},{hi}): {int(s2.sum())} wavefronts, lifetime {life[s2].mean() / 100.0:.1f} us, stages {nstage[s2].mean():.0f}: "
              + ", ".join(f"{nme.split()[0]} {100.0 * rec[s2, c].sum() / t2:.0f} %" for nme, c in zip(names, cols)))
