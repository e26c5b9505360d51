"""Probe (GPU box) [r6]: one regional target under rocprofv3 -- kernel split of the no_nest branch
python tools/probe/regional_one.py T nlon nlat nf"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402

T, nlon, nlat, nf = (int(v) for v in sys.argv[1:5])
rt = atlas_amd.RegionalTrans(nlon, -10.0, 0.05, np.linspace(60.0, 30.0, nlat), T)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
gp = torch.zeros(nf * nlon * nlat, dtype=torch.float64, device="cuda")
for _ in range(6):
    rt.invtrans(nf, sp, gp)
rt.synchronize()
torch.cuda.synchronize()
