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


@pytest.mark.parametrize("seed", range(int(os.environ.get("ATLAS_AMD_FUZZ_DIST_CASES", "12"))))
def test_random_distributed_transform_equals_the_single_device_one(seed):
    """[r6] the same for the distributed transform inside the library (csrc/dist_trans.hip over the "local" communicator: P ranks as host
    threads on one device): random reduced grids, rank counts 2 .. 6 (more ranks than some grids have rows per hemisphere: empty
    bands), field counts, message limits that cut the runs into one .. many pieces, replicated and m-sharded input.  Every rank's
    latitude band must be, bit for bit, the rows of the single-device transform -- which must meet the oracle."""
    import threading
    from atlas_amd.comm import CommHub
    from atlas_amd.dist import DistributedTrans
    rng = np.random.default_rng(7000 + seed)
    nx, y = _random_grid(rng)
    while len(nx) < 8:                       # a few rows at least, so that most ranks own some
        nx, y = _random_grid(rng)
    T = int(rng.integers(2, 64))
    nf = int(rng.choice([1, 2, 5, 9, 48, 49]))
    P = int(rng.integers(2, 7))
    maxmsg = int(rng.choice([0, 8, 4096, 1 << 16]))
    sharded = bool(rng.integers(0, 2))
    g = atlas_amd.StructuredGrid(nx=nx, y=y)
    sp = red_spectra(T, nf, seed=seed)
    sp_d = torch.from_numpy(sp).cuda()
    tr = atlas_amd.Trans(g, T)
    gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp_d, gp)
    tr.synchronize()
    ref = gp.cpu().numpy().reshape(nf, -1)
    assert compute_rms(ref.reshape(-1), oracle.OraclePlan(T, nx, y).invtrans(nf, sp, use_fft=False)) < 1e-13
    off = np.concatenate([[0], np.cumsum(nx)])
    hub = CommHub(P)
    comms = [hub.comm(r) for r in range(P)]
    out, err = [None] * P, [None] * P

    def rank(r):
        try:
            torch.cuda.set_device(0)
            d = DistributedTrans(g, T, comm=comms[r], mode="alltoall")
            if maxmsg:
                d.set_max_message_bytes(max(maxmsg, 8))
            b = d.trans.bands()
            n = d.trans.nb_gridpoints()
            gpr = torch.full((max(nf * n, 1),), float("nan"), dtype=torch.float64, device="cuda")
            if sharded:
                sh = torch.from_numpy(d.shard_spectra(nf, sp)).cuda()
                d.invtrans_many_sharded(nf, [sh], [gpr])
            else:
                d.invtrans(nf, sp_d, gpr)
            d.trans.synchronize()
            out[r] = (int(b[r]), int(b[r + 1]), gpr.cpu().numpy()[:nf * n].reshape(nf, -1) if n else np.zeros((nf, 0)))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=rank, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for e in err:
        if e is not None:
            raise e
    rows = 0
    for b0, b1, band in out:
        assert band.shape[1] == off[b1] - off[b0]
        assert np.array_equal(band, ref[:, off[b0]:off[b1]]), (seed, P, T, nf, maxmsg, sharded, b0, b1)
        rows += b1 - b0
    assert rows == len(nx)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ATLAS_AMD_FUZZ_REGIONAL_CASES", "12"))))
def test_random_regional_targets_against_the_oracle(seed):
    """[r6] the no_nest branch (atlas_amd__RegionalTrans__*, TransLocal.cc:394-406,719-738,1139-1148): random targets -- latitudes in any
    order with repeats, mirror pairs and the equator, random west / spacing (wrapping past 360 degrees or not), row lengths 1 .. 400,
    1 .. 60 rows, truncations 1 .. 120, field counts that leave the last tile of the matrix product part-filled; scalar and vor/div
    calls against the oracle's restatement of the branch."""
    rng = np.random.default_rng(7000 + seed)
    nlat = int(rng.integers(1, 61))
    lats = rng.uniform(-85.0, 85.0, nlat)
    if nlat > 4:
        lats[1] = -lats[0]                 # a mirror pair
        lats[2] = lats[3]                  # a repeated latitude
        if seed % 3 == 0:
            lats[4] = 0.0                  # the equator
    nlon = int(rng.integers(1, 401))
    west, dlon = float(rng.uniform(-180.0, 360.0)), float(rng.uniform(0.01, 3.0))
    lons = west + dlon * np.arange(nlon)
    T = int(rng.integers(1, 121))
    nf = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100]))
    sp = red_spectra(T, nf, seed=seed)
    rt = atlas_amd.RegionalTrans(nlon, west, dlon, lats, T)
    gp = torch.full((nf * nlon * nlat,), float("nan"), dtype=torch.float64, device="cuda")
    rt.invtrans(nf, torch.from_numpy(sp).cuda(), gp)
    rt.synchronize()
    got = gp.cpu().numpy()
    assert np.isfinite(got).all()
    want = oracle.invtrans_regional(T, lats, lons, nf, sp)
    assert compute_rms(got, want.ravel()) < 1e-13, ("scalar", nlat, nlon, T, nf)
    ns, nvd = int(rng.integers(0, 3)), int(rng.integers(1, 5))
    s, vor, div = red_spectra(T, max(ns, 1), seed + 1), red_spectra(T, nvd, seed + 2), red_spectra(T, nvd, seed + 3)
    w = np.full((ns + 2 * nvd) * nlon * nlat, np.nan)
    rt.invtrans_vordiv(ns, s if ns else None, nvd, vor, div, w)
    wref = oracle.invtrans_regional_vordiv(T, lats, lons, ns, s if ns else None, nvd, vor, div)
    assert compute_rms(w, wref.ravel()) < 1e-12, ("vor/div", nlat, nlon, T, ns, nvd)
