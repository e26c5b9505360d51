"""Seeded random cases of the whole transform against the oracle: hand-made reduced grids (random symmetric latitudes, random row
lengths -- even, odd, prime, tiny, up to a few hundred points -- or one length for all rows), random truncations below, at and
above what the rows resolve, random field counts, scalar and vor/div calls, fp64 and fp32.  The fixed configurations of the other
files exercise the shapes the benchmark grids launch; this file asks for the ones nobody thought of (round 5's comparison of
library and oracle over every region count found the one disagreement of the equal_regions partitioner that way)."""
import os

import numpy as np
import pytest

import atlas_amd
from helpers import compute_rms, red_spectra
from oracle import translocal as oracle

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _random_grid(rng):
    nh = int(rng.integers(1, 33))
    lats = np.sort(rng.uniform(0.5, 89.5, nh))[::-1]
    kind = int(rng.integers(0, 4))
    if kind == 0:      # regular: one length
        nxh = np.full(nh, int(rng.integers(1, 200)) * 2)
    elif kind == 1:    # reduced, even lengths growing towards the equator
        nxh = np.sort(rng.integers(2, 200, nh)) * 2
    elif kind == 2:    # anything goes: odd, prime, tiny
        nxh = rng.integers(1, 300, nh)
    else:              # octahedral-like 20 + 4 j with a random start
        nxh = int(rng.integers(4, 40)) + 4 * np.arange(nh)
    y = np.concatenate([lats, -lats[::-1]])
    nx = np.concatenate([nxh, nxh[::-1]]).astype(np.int32)
    return nx, y


@pytest.mark.parametrize("seed", range(int(os.environ.get("ATLAS_AMD_FUZZ_CASES", "40"))))   # more: a one-off soak run
def test_random_grid_truncation_and_fields_against_the_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    nx, y = _random_grid(rng)
    T = int(rng.integers(1, 96))
    g = atlas_amd.StructuredGrid(nx=nx, y=y)
    tr = atlas_amd.Trans(g, T)
    plan = oracle.OraclePlan(T, nx, y)
    npts = g.size()
    nf = int(rng.choice([1, 2, 3, 5, 8, 17, 48, 49, 100]))
    sp = red_spectra(T, nf, seed=seed)
    gp = torch.full((nf * npts,), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp).cuda(), gp)
    tr.synchronize()
    got = gp.cpu().numpy()
    ref = plan.invtrans(nf, sp, use_fft=False)
    assert np.isfinite(got).all()
    assert compute_rms(got, ref) < 1e-13, ("scalar fp64", list(nx[:len(nx) // 2]), T, nf)
    # fp32 variant of the same call
    gp32 = torch.full((nf * npts,), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp).cuda().float(), gp32)
    tr.synchronize()
    assert compute_rms(gp32.cpu().numpy().astype(np.float64), ref) < 2e-6, ("scalar fp32", list(nx[:len(nx) // 2]), T, nf)
    # the vor/div call: ns scalars beside nvd pairs
    ns, nvd = int(rng.integers(0, 4)), int(rng.integers(1, 6))
    s, vor, div = red_spectra(T, max(ns, 1), seed + 1), red_spectra(T, nvd, seed + 2), red_spectra(T, nvd, seed + 3)
    w = torch.full(((ns + 2 * nvd) * npts,), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(ns, torch.from_numpy(s).cuda() if ns else None, nvd, torch.from_numpy(vor).cuda(), torch.from_numpy(div).cuda(), w)
    tr.synchronize()
    wref = plan.invtrans_vordiv(ns, s if ns else None, nvd, vor, div, use_fft=False).reshape(ns + 2 * nvd, -1)
    wgot = w.cpu().numpy().reshape(ns + 2 * nvd, -1)
    assert np.isfinite(wgot).all()
    for f in range(ns + 2 * nvd):
        assert compute_rms(wgot[f], wref[f]) < 1e-12, ("vor/div", f, list(nx[:len(nx) // 2]), T, ns, nvd)
