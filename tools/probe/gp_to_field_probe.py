"""Dev tool (GPU box): the [nf][npts] -> [npts][nf] transposition in front of the halo exchange of the distributed transform
(csrc/vd2uv_kernel.hip: gp_to_field_rows_kernel; ATLAS_AMD_GP_TO_FIELD=tiles: the 32 x 32-tile form) -- bitwise check against
torch and the rate, for the band of one rank of 4 on O640 (C3) and of 8 on O1280 (C4), 137 levels.
    python tools/probe/gp_to_field_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atlas_amd import _lib

fn = _lib._sig("atlas_amd__diag_gp_to_field", C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_double))
for name, npts, nf in (("C3 band (O640 / 4)", 415360, 137), ("C4 band (O1280 / 8)", 824960, 137), ("ragged", 100003, 61),
                       ("few fields", 50001, 3), ("many fields", 20000, 1370)):
    gp = torch.randn(nf, npts, dtype=torch.float64, device="cuda")
    want = gp.t().contiguous()
    for form in ("tiles", "rows"):
        os.environ["ATLAS_AMD_GP_TO_FIELD"] = form
        out = torch.full((npts, nf), float("nan"), dtype=torch.float64, device="cuda")
        ms = C.c_double(0.0)
        _lib.check(fn(gp.data_ptr(), out.data_ptr(), npts, nf, 20, C.byref(ms)))
        torch.cuda.synchronize()
        ok = torch.equal(out, want)
        print(f"{name:22s} {form:6s} {ms.value * 1e3:9.1f} us  {2 * npts * nf * 8 / ms.value / 1e6:8.0f} GB/s (read + write)  bitwise {'ok' if ok else 'WRONG'}",
              flush=True)
