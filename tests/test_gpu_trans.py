"""GPU parity tests of the TransLocal inverse transform (through the C ABI) against the CPU oracle.

Tolerance (fp64): rel-RMS (RMS diff / max|ref|, the reference's compute_rms) <= 1e-13 against the oracle on the
same inputs and <= 1e-13 against the analytic known answers at T <= 63 (reference: 1e-13, test_transgeneral.cc:534);
wind: 2e-6 vs analytic (reference :538), 1e-12 vs oracle.  Full-size (TL1279 -> O1280, 137 levels) is checked on
sampled latitude rows against the oracle and through size-independent properties (linearity, zero input,
dropped m=T wavenumber, host/device entry points agree bitwise)."""
import math
import os

import numpy as np
import pytest

import atlas_amd
import oracle
from atlas_amd import _lib
from helpers import (CLOSED_FORMS, analytic_scalar, compute_rms, red_spectra, rows_of_every_fft_class, unit_spectrum, wind_kat)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-13


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_device(tr, nf, sp):
    gp = torch.zeros(nf * tr.nb_gridpoints_global(), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, dev(sp), gp)
    tr.synchronize()
    return gp.cpu().numpy()


_trans_cache = {}


def get_trans(gridname, T):
    key = (gridname, T)
    if key not in _trans_cache:
        g = atlas_amd.Grid(gridname)
        _trans_cache[key] = (g, atlas_amd.Trans(g, T))
    return _trans_cache[key]


@pytest.mark.parametrize("gridname,T,nf", [
    ("O64", 63, 1),      # BASELINE config C1
    ("O64", 63, 3), ("F64", 63, 2), ("O32", 31, 20), ("O32", 31, 8), ("O32", 31, 9),
    ("F32", 31, 137),    # 137 columns -> 9 r-tiles per wave
    ("O32", 31, 150),    # more than 144 fields -> two column chunks
    ("O48", 95, 5),      # linear truncation branch (T >= ndgl-1)
    ("O80", 100, 4),     # quadratic branch
    ("O160", 159, 60),   # BASELINE config C2
    ("F320", 319, 3),    # regular Gaussian grid, h = 640 = 5*2^7: specialised direct (no Bluestein) kernel
    ("F256", 255, 2),    # h = 512 = 2^9
])
def test_invtrans_parity_with_oracle(gridname, T, nf):
    g, tr = get_trans(gridname, T)
    sp = red_spectra(T, nf)
    gp = run_device(tr, nf, sp)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=(T > 100))
    assert np.isfinite(gp).all()
    assert compute_rms(gp, ref) < TOL


def test_nlat0_matches_oracle():
    for gridname, T in [("O64", 63), ("F64", 63), ("O160", 159), ("O80", 100)]:
        g, tr = get_trans(gridname, T)
        assert np.array_equal(tr.nlat0(), oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False).nlat0)


def test_legendre_stage_parity():
    gridname, T, nf = "O64", 63, 5
    g, tr = get_trans(gridname, T)
    sp = red_spectra(T, nf)
    RP = tr.fourier_row_pitch(nf)
    F = torch.zeros(tr.fourier_size(nf), dtype=torch.float64, device="cuda")
    tr.legendre_device(T, nf, dev(sp), F)
    tr.synchronize()
    F = F.cpu().numpy().reshape(g.ny(), T + 1, RP)
    op = oracle.OraclePlan(T, g.nx(), g.y())
    ref = op.legendre(nf, sp)  # [fld][lat][m][re,im]  (TransLocal.h:177-180)
    mine = F[:, :, :2 * nf].reshape(g.ny(), T + 1, nf, 2).transpose(2, 0, 1, 3)
    nlat0, nl = op.nlat0, g.ny()
    for m in range(T + 1):
        rows = [j for j in range(nl) if min(j, nl - 1 - j) >= nlat0[m]]
        assert compute_rms(mine[:, rows, m, :], ref[:, rows, m, :]) < TOL, m


@pytest.mark.parametrize("gridname", ["F64", "O64"])
def test_analytic_known_answers(gridname):
    """reference KATs (test_transgeneral.cc:80-272, 433-449): closed forms n<=3 and sectoral harmonics"""
    T = 63
    g, tr = get_trans(gridname, T)
    nx, lat = g.nx(), g.y()
    cases = [(n, m, im) for (n, m) in CLOSED_FORMS for im in (0, 1) if not (m == 0 and im == 1)]
    cases += [(n, n, im) for n in (5, 17, 45) for im in (0, 1)]
    nf = len(cases)
    sp = np.zeros((T + 1) * (T + 2) * nf)
    for f, (n, m, im) in enumerate(cases):
        sp += unit_spectrum(T, nf, n, m, im, fld=f)
    gp = run_device(tr, nf, sp).reshape(nf, -1)
    for f, (n, m, im) in enumerate(cases):
        mask = [oracle.fourier_truncation(T, int(nx[j]), int(nx.max()), len(nx), lat[j] * math.pi / 180.0,
                                          g.regular()) > m for j in range(len(nx))]
        assert compute_rms(gp[f], analytic_scalar(nx, lat, n, m, im, mask)) < TOL, (n, m, im)


def test_wavenumber_T_is_dropped_and_zero_maps_to_zero():
    # TransLocal.cc:982: entries are used only if m < truncation -> the single (T,T) coefficient has no effect
    T = 31
    g, tr = get_trans("O32", T)
    gp = run_device(tr, 1, unit_spectrum(T, 1, T, T, 0))
    assert np.array_equal(gp, np.zeros_like(gp))
    gp = run_device(tr, 2, np.zeros((T + 1) * (T + 2) * 2))
    assert np.array_equal(gp, np.zeros_like(gp))


def test_host_and_device_entry_points_agree_bitwise():
    T, nf = 63, 4
    g, tr = get_trans("O64", T)
    sp = red_spectra(T, nf, seed=7)
    a = run_device(tr, nf, sp)
    b = np.zeros(nf * g.size())
    tr.invtrans(nf, sp, b)
    assert np.array_equal(a, b)


def test_vordiv_path_against_oracle_and_analytic():
    T = 63
    g, tr = get_trans("F64", T)
    nx, lat = g.nx(), g.y()
    op = oracle.OraclePlan(T, nx, lat)
    ns, nvd = 2, 3
    sp, vor, div = red_spectra(T, ns, 1), red_spectra(T, nvd, 2), red_spectra(T, nvd, 3)
    gp = np.zeros((ns + 2 * nvd) * g.size())
    tr.invtrans(ns, sp, nvd, vor, div, gp)
    ref = op.invtrans_vordiv(ns, sp, nvd, vor, div)
    assert compute_rms(gp, ref) < 1e-12
    # device entry point
    gp_d = torch.zeros(gp.size, dtype=torch.float64, device="cuda")
    tr.invtrans(ns, dev(sp), nvd, dev(vor), dev(div), gp_d)
    tr.synchronize()
    assert np.array_equal(gp_d.cpu().numpy(), gp)
    # analytic wind known answers (test_transgeneral.cc:286-371), tolerance 2e-6 (:538)
    for ivar_in in (0, 1):
        for (n, m, imag) in [(1, 0, 0), (1, 1, 0), (1, 1, 1)]:
            coef, zero = unit_spectrum(T, 1, n, m, imag), np.zeros((T + 1) * (T + 2))
            v, d = (coef, zero) if ivar_in == 0 else (zero, coef)
            wind = np.zeros(2 * g.size())
            tr.invtrans_vordiv2wind(1, v, d, wind)
            wind = wind.reshape(2, -1)
            for ivar_out in (0, 1):
                ana = np.concatenate([wind_kat(ivar_in, ivar_out, n, m, imag, np.arange(k) * (2 * math.pi / k),
                                               y * math.pi / 180.0) for k, y in zip(nx, lat)])
                assert compute_rms(wind[ivar_out], ana) < 2e-6


def test_legendre_cache_round_trip():
    # TransLocal.cc:608-647: write_legendre / read: a Trans built from the exported blob gives identical results
    T, nf = 31, 3
    g, tr = get_trans("O32", T)
    blob = tr.legendre_cache()
    op = oracle.OraclePlan(T, g.nx(), g.y())
    sym, asym = op.tables()
    assert blob.tobytes() == sym.tobytes() + asym.tobytes()   # byte-compatible with the reference layout
    tr2 = atlas_amd.Trans(g, T, legendre_cache=blob)
    sp = red_spectra(T, nf, seed=3)
    assert np.array_equal(run_device(tr, nf, sp), run_device(tr2, nf, sp))
    with pytest.raises(_lib.AtlasAmdError):
        atlas_amd.Trans(g, T, legendre_cache=blob[:-8])
    # LegendreCacheCreator::create() / create(path) (LegendreCacheCreatorLocal.cc:150-160)
    creator = atlas_amd.LegendreCacheCreator(g, T)
    assert creator.supported() and creator.create().tobytes() == blob.tobytes()
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        path = creator.create(os.path.join(d, creator.uid() + ".leg"))
        assert np.fromfile(path, dtype=np.uint8).tobytes() == blob.tobytes()


def test_regular_lonlat_grid_with_poles_and_equator():
    # L-type grid: 17 latitudes 90..-90 (poles clamped to +-89.9999999, equator row shared by both hemispheres)
    ny, nxl, T = 17, 32, 7
    lat = np.linspace(90.0, -90.0, ny)
    g = atlas_amd.StructuredGrid(nx=np.full(ny, nxl), y=lat)
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, 3)
    gp = run_device(tr, 3, sp)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(3, sp)
    assert compute_rms(gp, ref) < TOL


DIRECT_SHAPES = [(f, k) for f, ks in ((1, range(8, 14)), (3, range(7, 12)), (5, range(6, 11)), (9, (8, 9)), (15, (8,))) for k in ks]


@pytest.mark.parametrize("f,k", DIRECT_SHAPES)
def test_every_direct_row_shape_fp64_and_fp32(f, k):
    """One regular grid per specialised direct-row shape (h = n/2 = F * 2^K itself a length of the family, fft_core.h:
    ct_supported): the staged form of the direct rows (modes gathered once into LDS, twiddles requested up front, one butterfly
    per worker and stage; M = 8192 = [16,16,16,2] keeps the phase loop), its 256-register instances (first radix >= 15) and
    the fp32-arithmetic instances of the fp32 variant.  Oracle: the fp64 CPU restatement; fp32: 2e-6 rel-RMS against the
    fp64 device result of the float-rounded spectra."""
    M, T, nf = f << k, 21, 3
    lat = np.array([70.0, 25.0, -25.0, -70.0])
    g = atlas_amd.StructuredGrid(nx=np.full(4, 2 * M), y=lat)
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, nf, seed=300 + M % 97)
    gp = run_device(tr, nf, sp)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True)
    assert compute_rms(gp, ref) < TOL
    sp32 = sp.astype(np.float32)
    ref32 = run_device(tr, nf, sp32.astype(np.float64))
    gp32 = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp32)
    tr.synchronize()
    assert compute_rms(gp32.cpu().numpy().astype(np.float64), ref32) < 2e-6


def _largest_prime_at_most(n):
    while True:
        if n > 1 and all(n % d for d in range(2, int(n ** 0.5) + 1)):
            return n
        n -= 1


@pytest.mark.parametrize("f,k", DIRECT_SHAPES)
def test_every_bluestein_row_shape_fp64_and_fp32(f, k):
    """One grid per specialised Bluestein shape M = F * 2^K: rows whose half length h is the largest prime with 2h - 1 <= M (and a
    second, smaller prime half length of the same class), full truncation up to the row's Nyquist limit for the short shapes.
    Covers the plain and the row_ct3 (M >= 3840) kernels, the zero-filled staging / zero chirp padding, and the fp32-arithmetic
    instances of both (the BASELINE-sized fp32 tests only reach M <= 768 through Bluestein)."""
    M = f << k
    h1 = _largest_prime_at_most((M + 1) // 2)
    h2 = _largest_prime_at_most(h1 - 1)
    nx = np.array([2 * h1, 2 * h2, 2 * h2, 2 * h1])
    lat = np.array([70.0, 25.0, -25.0, -70.0])
    T, nf = min(191, h2 - 1), 9                     # nine fields: a full field group and a one-field group
    g = atlas_amd.StructuredGrid(nx=nx, y=lat)
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, nf, seed=500 + M % 89)
    gp = run_device(tr, nf, sp)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True)
    assert compute_rms(gp, ref) < TOL
    sp32 = sp.astype(np.float32)
    ref32 = run_device(tr, nf, sp32.astype(np.float64))
    gp32 = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp32)
    tr.synchronize()
    assert bool(torch.isfinite(gp32).all())
    assert compute_rms(gp32.cpu().numpy().astype(np.float64), ref32) < 2e-6


def test_development_switches_do_not_exist_in_the_product_library(monkeypatch, capfd):
    """ATLAS_AMD_FFT_NATIVE=1 (the native mixed-radix rows, tools/experiments/ since round 5) and the other switches of class
    "dev" (csrc/env.cpp) are compiled out of the product library: a process that happens to carry one in its environment gets the
    default plan -- not an exception out of the Trans constructor (rounds 3 - 5), not another kernel -- and one line on stderr
    """
    if "exp" in os.environ.get("ATLAS_AMD_LIB", "") or "dev" in os.environ.get("ATLAS_AMD_LIB", ""):
        pytest.skip("experiments / dev build")
    g = atlas_amd.Grid("O320")
    plan0 = atlas_amd.Trans(g, 319).fourier_launch_plan()
    for k2, v in (("ATLAS_AMD_FFT_NATIVE", "1"), ("ATLAS_AMD_FFT_HYBRID", "1"), ("ATLAS_AMD_FFT_ONLY_M", "4096"), ("ATLAS_AMD_FFT_LDS_PAD", "40000")):
        monkeypatch.setenv(k2, v)
    tr = atlas_amd.Trans(g, 319)
    assert tr.fourier_launch_plan() == plan0
    cfg = atlas_amd._lib.effective_config()
    assert cfg["ATLAS_AMD_FFT_NATIVE"]["source"] == "compiled out" and cfg["ATLAS_AMD_FFT_NATIVE"]["value"] == "0"
    assert "development switch" in capfd.readouterr().err


def test_coarse_row_classes_in_one_launch_are_bitwise_equal_to_one_launch_per_class(monkeypatch):
    """[r4] small reduced grids (BASELINE C2: TL159 -> O160): the coarse Bluestein classes 256 / 512 / 1024 share ONE launch whose
    workgroups switch into the instantiated body of their row's class (fft_rows_coarse_kernel); same row code, same bits as one
    launch per class (ATLAS_AMD_FFT_COARSE_FUSED=0), fp64 and fp32, and the oracle agrees"""
    g = atlas_amd.Grid("O160")
    T, nf = 159, 11
    sp = red_spectra(T, nf, seed=43)
    sp32 = torch.from_numpy(sp.astype(np.float32)).cuda()
    outs, plans = {}, {}
    for fused in ("1", "0"):
        monkeypatch.setenv("ATLAS_AMD_FFT_COARSE_FUSED", fused)
        tr = atlas_amd.Trans(g, T)
        cls = tr.fft_row_classes()
        assert set(cls[:, 1]) <= {256, 512, 1024} and (cls[:, 2] == 1).all()
        plans[fused] = tr.fourier_launch_plan()        # the switch is read per object: the other path really ran
        assert plans[fused]["coarse_fused"] == int(fused), plans
        gp32 = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
        tr.invtrans(nf, sp32, gp32)
        tr.synchronize()
        outs[fused] = (run_device(tr, nf, sp), gp32.cpu().numpy())
    assert plans["0"]["launches"] >= plans["1"]["launches"] + 2, plans      # three classes in one launch / one launch each
    assert np.array_equal(outs["1"][0], outs["0"][0]), "fp64: fused and per-class launches differ"
    # fp32: the same source instantiated in two kernels is contracted into FMAs differently by the compiler (1 ulp here and there;
    # round 4's "bitwise" claim compared a setting with itself -- ADVICE r4): within 1e-6 of the largest value, and both within
    # the fp32 tolerance of the fp64 result
    assert np.abs(outs["1"][1] - outs["0"][1]).max() <= 1e-6 * np.abs(outs["1"][0]).max()
    for k in ("1", "0"):
        assert compute_rms(outs[k][1].astype(np.float64), outs["1"][0]) < 2e-6
    assert compute_rms(outs["1"][0], oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True)) < TOL


@pytest.mark.parametrize("nf", [1, 2, 4, 6, 13, 60])
def test_several_fields_of_a_short_row_per_wavefront_are_bitwise_equal_to_one_field_per_workgroup(nf, monkeypatch):
    """[r4] the coarse classes of small reduced grids: four (Bluestein length 256) / two (512) consecutive fields of a row share a
    wavefront (fft_kernel.hip: fft_rows_coarse_multi_kernel -- every field its own work array and 16 / 32 workers, the fields' modes
    gathered together and read through a strided view; jobs at the end of the fields that are not full run their fields one after the
    other).  Same arithmetic per (row, field): the same bits as ATLAS_AMD_FFT_COARSE_MULTI=0 (BASELINE C2's grid; field counts with full
    and ragged last groups), and the vor/div call with its scaled wind fields."""
    g = atlas_amd.Grid("O160")
    T = 159
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, nf, seed=90 + nf)
    outs = {}
    for multi in ("1", "0"):
        monkeypatch.setenv("ATLAS_AMD_FFT_COARSE_MULTI", multi)
        outs[multi] = run_device(tr, nf, sp)
    assert np.array_equal(outs["1"], outs["0"])
    assert compute_rms(outs["1"], oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True)) < TOL
    if nf in (2, 13):
        nvd = 3
        vor, div = red_spectra(T, nvd, seed=7), red_spectra(T, nvd, seed=8)
        res = {}
        for multi in ("1", "0"):
            monkeypatch.setenv("ATLAS_AMD_FFT_COARSE_MULTI", multi)
            gp = torch.full(((nf + 2 * nvd) * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
            tr.invtrans(nf, dev(sp), nvd, dev(vor), dev(div), gp)
            tr.synchronize()
            res[multi] = gp.cpu().numpy()
        assert np.isfinite(res["1"]).all() and np.array_equal(res["1"], res["0"])


def test_fourier_scheduling_switches_do_not_change_results(monkeypatch):
    """L2 prefetch of a later job's modes, row -> XCD affinity and the number of class streams only move work around:
    bit-identical grid points with them off (fft_kernel.hip: PrefetchJob, fft_device.h: fft_unit_to_job)."""
    g, tr = get_trans("O320", 319)
    nf = 19                                   # three field groups, the last one with three fields
    sp = red_spectra(319, nf, seed=77)
    ref = run_device(tr, nf, sp)
    for env in ({"ATLAS_AMD_FFT_PREFETCH": "0"}, {"ATLAS_AMD_FFT_ROW_AFFINITY": "0"}, {"ATLAS_AMD_FFT_PREFETCH": "7"},
                {"ATLAS_AMD_FFT_STREAMS": "1"}, {"ATLAS_AMD_FFT_PREFETCH": "0", "ATLAS_AMD_FFT_ROW_AFFINITY": "0"}):
        for k2, v in env.items():
            monkeypatch.setenv(k2, v)
        got = run_device(tr, nf, sp)
        for k2 in env:
            monkeypatch.delenv(k2)
        assert np.array_equal(got, ref), env


def test_translocal_option_keys_are_accepted_and_change_nothing():
    """trans::Trans(grid, T, option::type("local") | option::fft("FFTW") | option::matrix_multiply(...) | option::warning(0)):
    keys every existing TransLocal caller passes (TransLocal.cc:61-110,326-335) -- accepted, noted, same bits"""
    g, tr = get_trans("O32", 31)
    sp = red_spectra(31, 3, seed=5)
    ref = run_device(tr, 3, sp)
    tr2 = atlas_amd.Trans(g, 31, type="local", fft="FFTW", matrix_multiply="lapack", precompute=True, warning=0,
                          write_fft="/tmp/unused.fft")
    for k in ("fft=FFTW", "matrix_multiply=lapack", "precompute=1", "warning=0", "write_fft=/tmp/unused.fft"):
        assert k in tr2.notes, tr2.notes
    assert np.array_equal(run_device(tr2, 3, sp), ref)
    assert atlas_amd.Trans(g, 31).notes == ""
    with pytest.raises(_lib.AtlasAmdError, match="FFT backend"):
        atlas_amd.Trans(g, 31, fft="FFT992")


def test_not_implemented_like_translocal():
    g, tr = get_trans("O32", 31)
    with pytest.raises(NotImplementedError):
        tr.dirtrans(1, None, None)
    with pytest.raises(NotImplementedError):
        tr.invtrans_adj(1, None, None)


# ---------------------------------------------------------------- BASELINE full size: TL1279 -> O1280, 137 levels
@pytest.fixture(scope="module")
def trans_full():
    g = atlas_amd.Grid("O1280")
    return g, atlas_amd.Trans(g, 1279)


def test_full_size_sampled_rows_against_oracle(trans_full):
    """TL1279 -> O1280, all 137 fields: one northern and one southern row of EVERY Fourier kernel class that is launched (the
    library reports the class of each row), at the full mode count of the row -- the 1280-piece LDS-DMA gather, the zero-filled
    staging above mmax, the L2 prefetch and the packed mode offsets only run at this size (Bluestein
    M = 5120 / 4096 / 2048 / 2560 ... were never sampled) -- plus the rows of rounds 1 - 3."""
    g, tr = trans_full
    T, nf = 1279, 137
    sp = red_spectra(T, nf)
    gp = run_device(tr, nf, sp).reshape(nf, -1)
    # 1275: n = 5120, h = 2560 -> specialised direct kernel; 540 / 900 / 1100: Bluestein lengths 2304 / 3840 / 4608
    # ([9,16,16], [15,16,16], [18,16,16]); 1147: h = 2304 itself -> direct kernel of that shape
    rows, classes = rows_of_every_fft_class(tr, extra=[0, 1, 540, 639, 900, 1100, 1147, 1275, 1279, 1280, 2000, 2559])
    launched_M = {c[1] for c in classes if c[2] == 1}
    # ([r6] the rows short enough for a coarse length -- 2h - 1 <= 2048 -- take 256 / 512 / 1024 in one launch and 2048 on every reduced grid
    # now: the tight classes 1280 / 1536 of rounds 2 - 5 are no longer launched on this grid)
    assert {256, 512, 1024, 2048, 2560, 3072, 4096, 5120, 6144} <= launched_M, launched_M
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    worst = 0.0
    for r, ref in zip(rows, op.invtrans_rows(nf, sp, rows, use_fft=True)):
        err = compute_rms(gp[:, off[r]:off[r + 1]], ref)
        worst = max(worst, err)
        assert err < 1e-12, (r, tuple(tr.fft_row_classes()[r]), err)
    print(f"full size: {len(rows)} rows of {len(classes)} classes, worst rel-rms {worst:.2e}")


def test_full_size_rotated_second_round_is_bitwise_equal(trans_full, monkeypatch):
    """ATLAS_AMD_FFT_MIDROT: in the row_ct3 rows with more than 256 middle butterflies (M = 4608 / 5120 / 6144) the wavefront that
    takes the second round rotates with the job -- other workers, the same butterflies: identical bits (fft_ct_rows.h: trot)"""
    g, tr = trans_full
    T, nf = 1279, 11
    sp = red_spectra(T, nf, seed=91)
    outs = {}
    for rot in ("0", "1"):
        monkeypatch.setenv("ATLAS_AMD_FFT_MIDROT", rot)
        outs[rot] = run_device(tr, nf, sp)
    monkeypatch.delenv("ATLAS_AMD_FFT_MIDROT")
    assert np.array_equal(outs["0"], outs["1"])


def test_full_size_linearity(trans_full):
    g, tr = trans_full
    T, nf = 1279, 16
    x, y = red_spectra(T, nf, seed=11), red_spectra(T, nf, seed=12)
    a, b = 0.75, -1.5
    gx, gy, gxy = run_device(tr, nf, x), run_device(tr, nf, y), run_device(tr, nf, a * x + b * y)
    assert compute_rms(gxy, a * gx + b * gy) < 1e-13
    # field independence: field k of a multi-field call == the single-field call
    g1 = run_device(tr, 1, x.reshape(-1, nf)[:, 5].copy())
    assert compute_rms(gx.reshape(nf, -1)[5], g1) < 1e-14


# ---------------------------------------------------------------- multi-GPU decomposition, emulated on one device
@pytest.mark.parametrize("gridname,T,nf,nparts", [("O64", 63, 5, 2), ("O64", 63, 3, 3), ("F32", 31, 4, 4),
                                                  ("O160", 159, 9, 8)])
def test_sharded_stages_reproduce_single_device_result(gridname, T, nf, nparts):
    """P objects with (nparts=P, part=p) in one process: m-sharded Legendre stage, the all-to-all replaced by device
    copies that follow atlas_amd.dist_torch.transpose_plan, latitude-band Fourier stage.  Must equal the single-object
    result bit for bit (same arithmetic per (m, latitude) and per row)."""
    from atlas_amd.dist_torch import transpose_plan
    g, tr1 = get_trans(gridname, T)
    sp = red_spectra(T, nf, seed=5)
    ref = run_device(tr1, nf, sp).reshape(nf, -1)
    sp_d = dev(sp)
    trs = [atlas_amd.Trans(g, T, nparts=nparts, part=p) for p in range(nparts)]
    bands = trs[0].bands()
    RP = trs[0].fourier_row_pitch(nf)
    plans = [transpose_plan(g.ny(), T, RP, bands, nparts, p) for p in range(nparts)]
    F = []
    for p, tr in enumerate(trs):
        assert tr.owned_wavenumbers() == plans[p]["cnt"][p]
        f = torch.zeros(tr.fourier_size(nf), dtype=torch.float64, device="cuda")
        tr.legendre_device(T, nf, sp_d, f)
        tr.synchronize()
        F.append(f)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for q, tr in enumerate(trs):
        R = torch.zeros(sum(plans[q]["out_splits"]), dtype=torch.float64, device="cuda")
        for p in range(nparts):   # what all_to_all_single delivers to rank q from rank p
            src0 = sum(plans[p]["in_splits"][:q])
            n = plans[p]["in_splits"][q]
            assert n == plans[q]["out_splits"][p]
            R[plans[q]["out_offsets"][p]:plans[q]["out_offsets"][p] + n] = F[p][src0:src0 + n]
        gp = torch.zeros(nf * tr.nb_gridpoints(), dtype=torch.float64, device="cuda")
        tr.fourier_device(nf, 0, [R[o:] for o in plans[q]["out_offsets"]], plans[q]["cnt"], gp)
        tr.synchronize()
        lo, hi = off[bands[q]], off[bands[q + 1]]
        assert tr.nb_gridpoints() == hi - lo
        assert np.array_equal(gp.cpu().numpy().reshape(nf, -1), ref[:, lo:hi]), q


@pytest.mark.parametrize("gridname,T,nf,nparts", [("O64", 63, 5, 2), ("O64", 63, 3, 3), ("F32", 31, 4, 4),
                                                  ("O160", 159, 9, 8), ("O160", 159, 20, 5)])
def test_latitude_band_sharding_reproduces_single_device_result(gridname, T, nf, nparts):
    """shard="band": every device computes all wavenumbers for the rows of its own latitude band (Atlas bands rule)
    and transforms them without any exchange.  The per-(m, latitude) and per-row arithmetic is that of the single
    device, so the concatenated bands must equal its result bit for bit; also on the vor/div path."""
    g, tr1 = get_trans(gridname, T)
    sp = red_spectra(T, nf, seed=9)
    ref = run_device(tr1, nf, sp).reshape(nf, -1)
    sp_d = dev(sp)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    nvd = 2
    vor, div = red_spectra(T, nvd, seed=10), red_spectra(T, nvd, seed=11)
    ref_uv = torch.zeros((nf + 2 * nvd) * g.size(), dtype=torch.float64, device="cuda")
    tr1.invtrans(nf, sp_d, nvd, dev(vor), dev(div), ref_uv)
    tr1.synchronize()
    ref_uv = ref_uv.cpu().numpy().reshape(nf + 2 * nvd, -1)
    total = 0
    for p in range(nparts):
        tr = atlas_amd.Trans(g, T, nparts=nparts, part=p, shard="band")
        bands = tr.bands()
        lo, hi = off[bands[p]], off[bands[p + 1]]
        assert tr.nb_gridpoints() == hi - lo
        total += tr.nb_gridpoints()
        gp = torch.full((nf * tr.nb_gridpoints(),), np.nan, dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp_d, gp)
        tr.synchronize()
        assert np.array_equal(gp.cpu().numpy().reshape(nf, -1), ref[:, lo:hi]), p
        gp = torch.full(((nf + 2 * nvd) * tr.nb_gridpoints(),), np.nan, dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp_d, nvd, dev(vor), dev(div), gp)
        tr.synchronize()
        assert np.array_equal(gp.cpu().numpy().reshape(nf + 2 * nvd, -1), ref_uv[:, lo:hi]), p
    assert total == g.size()


def test_field_and_fieldset_overloads_like_translocal():
    """TransLocal.cc:818-897: rank-1 Field / FieldSet wrappers over the pointer API, and the wind field of
    invtrans_vordiv2wind in both accepted shapes -- the (npts, 2) branch goes through gp_transpose exactly as written in
    the reference (out[f*npts + g] = tmp[g*2 + f], tmp = the (2, npts) result)."""
    from atlas_amd._lib import AtlasAmdError
    g, tr = get_trans("O32", 31)
    npts, nspec = g.size(), tr.nb_spectral_coefficients()
    sp = red_spectra(31, 1, seed=3)
    want = np.zeros(npts)
    tr.invtrans(1, sp, want)
    got = tr.invtrans_field(sp.copy(), np.zeros(npts))
    assert np.array_equal(got, want)
    longer = tr.invtrans_field(sp.copy(), np.full(npts + 7, -1.0))     # "hopefully the halo is appended"
    assert np.array_equal(longer[:npts], want) and np.all(longer[npts:] == -1.0)
    sps = [red_spectra(31, 1, seed=s) for s in (4, 5, 6)]
    gps = tr.invtrans_fieldset(sps, [np.zeros(npts) for _ in sps])
    for s, gp in zip(sps, gps):
        ref = np.zeros(npts)
        tr.invtrans(1, s, ref)
        assert np.array_equal(gp, ref)
    with pytest.raises(AtlasAmdError):
        tr.invtrans_fieldset(sps, [np.zeros(npts)])                    # sizes differ
    with pytest.raises(AtlasAmdError):
        tr.invtrans_field(np.zeros((nspec, 1)), np.zeros(npts))        # rank-1 only
    vor, div = red_spectra(31, 1, seed=7), red_spectra(31, 1, seed=8)
    uv = np.zeros(2 * npts)
    tr.invtrans_vordiv2wind(1, vor, div, uv)
    w2n = tr.invtrans_vordiv2wind_field(vor, div, np.zeros((2, npts)))
    assert np.array_equal(w2n.ravel(), uv)
    wn2 = tr.invtrans_vordiv2wind_field(vor, div, np.zeros((npts, 2)))
    lit = np.empty(2 * npts)
    for f in range(2):                                                  # gp_transpose, TransLocal.cc:861-867
        lit[f * npts:(f + 1) * npts] = uv[np.arange(npts) * 2 + f]
    assert np.array_equal(wn2.ravel(), lit)
    with pytest.raises(NotImplementedError):
        tr.invtrans_vordiv2wind_field(vor, div, np.zeros((3, npts)))
    for fn in (tr.invtrans_grad_field, tr.invtrans_adj_field, tr.dirtrans_field):
        with pytest.raises(NotImplementedError):
            fn(None, None)


@pytest.mark.parametrize("T,nf", [(31, 3), (63, 1), (159, 5)])
def test_vordivtouv_execute_against_oracle(T, nf):
    """VorDivToUV::execute (VorDivToUVLocal.cc:62-189), host and device pointers; tolerance 1e-14 relative to max |U|
    (the oracle tabulates eps/lap as the reference does, the kernel evaluates them per element)."""
    vor, div = red_spectra(T, nf, seed=31), red_spectra(T, nf, seed=32)
    Uo, Vo = oracle.vd2uv(T, nf, vor, div)
    vd = atlas_amd.VorDivToUV(T)
    ncoef = (T + 1) * (T + 2)
    U, V = vd.execute(ncoef, nf, vor, div, np.zeros_like(vor), np.zeros_like(vor))
    for got, want in ((U, Uo), (V, Vo)):
        assert np.abs(got - want).max() <= 1e-14 * np.abs(want).max()
    Ud, Vd = vd.execute(ncoef, nf, dev(vor), dev(div), torch.zeros(vor.size, dtype=torch.float64, device="cuda"),
                        torch.zeros(vor.size, dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(Ud.cpu().numpy(), U) and np.array_equal(Vd.cpu().numpy(), V)
    from atlas_amd._lib import AtlasAmdError
    with pytest.raises(AtlasAmdError):
        vd.execute(ncoef - 2, nf, vor, div, U, V)


def test_regular_gaussian_full_size_sampled_rows():
    """BASELINE config C5's grid in fp64: TL1279 -> F1280 (every row n = 5120, h = 2560 = 5*2^9: specialised direct
    kernel), sampled rows against the oracle."""
    g = atlas_amd.Grid("F1280")
    tr = atlas_amd.Trans(g, 1279)
    T, nf = 1279, 6
    sp = red_spectra(T, nf, seed=17)
    gp = run_device(tr, nf, sp).reshape(nf, -1)
    rows = [0, 7, 1279, 1280, 2559]
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for r, ref in zip(rows, op.invtrans_rows(nf, sp, rows, use_fft=True)):
        assert compute_rms(gp[:, off[r]:off[r + 1]], ref) < 1e-12, r


@pytest.mark.parametrize("gridname,T,nf,rows", [("O64", 63, 4, (10, 40)), ("F32", 31, 3, (0, 9)), ("O160", 159, 7, (200, 320)),
                                                ("O64", 63, 2, (60, 70))])
def test_zonal_band_crop_equals_rows_of_the_global_transform(gridname, T, nf, rows):
    """rows=(j0, j1): the nested regional case of TransLocal for domains that keep whole latitude rows
    (TransLocal.cc:394-470): the result is rows j0..j1-1 of the global transform, bit for bit, on the scalar and the
    vor/div path and through the host-pointer API."""
    g, tr1 = get_trans(gridname, T)
    sp = red_spectra(T, nf, seed=41)
    ref = run_device(tr1, nf, sp).reshape(nf, -1)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    lo, hi = off[rows[0]], off[rows[1]]
    tr = atlas_amd.Trans(g, T, rows=rows)
    assert tr.nb_gridpoints() == hi - lo and list(tr.bands()) == list(rows)
    gp = torch.full((nf * (hi - lo),), np.nan, dtype=torch.float64, device="cuda")
    tr.invtrans(nf, dev(sp), gp)
    tr.synchronize()
    assert np.array_equal(gp.cpu().numpy().reshape(nf, -1), ref[:, lo:hi])
    host = np.full(nf * (hi - lo), np.nan)
    tr.invtrans(nf, sp, host)
    assert np.array_equal(host.reshape(nf, -1), ref[:, lo:hi])
    vor, div = red_spectra(T, 1, seed=42), red_spectra(T, 1, seed=43)
    full = np.zeros((nf + 2) * g.size())
    tr1.invtrans(nf, sp, 1, vor, div, full)
    part = np.full((nf + 2) * (hi - lo), np.nan)
    tr.invtrans(nf, sp, 1, vor, div, part)
    assert np.array_equal(part.reshape(nf + 2, -1), full.reshape(nf + 2, -1)[:, lo:hi])
    with pytest.raises(_lib.AtlasAmdError):
        atlas_amd.Trans(g, T, rows=(5, g.ny() + 1))


@pytest.mark.parametrize("gridname,T,nf", [("O64", 63, 5), ("F64", 63, 3), ("O160", 159, 20), ("F320", 319, 9)])
def test_fp32_variant_against_fp64(gridname, T, nf):
    """BASELINE config C5's precision: float spectra / table / grid points, fp32 MFMA in the Legendre stage.  Tolerance:
    rel-RMS <= 2e-6 against the fp64 result of the same (float-rounded) spectra; fields stay independent."""
    g, tr = get_trans(gridname, T)
    sp32 = red_spectra(T, nf, seed=51).astype(np.float32)
    ref = run_device(tr, nf, sp32.astype(np.float64))
    gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp)
    tr.synchronize()
    got = gp.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    assert compute_rms(got, ref) < 2e-6
    one = torch.zeros(g.size(), dtype=torch.float32, device="cuda")
    tr.invtrans(1, torch.from_numpy(np.ascontiguousarray(sp32.reshape(-1, nf)[:, 2])).cuda(), one)
    tr.synchronize()
    assert np.array_equal(one.cpu().numpy(), gp.cpu().numpy().reshape(nf, -1)[2])
    host = np.full(nf * g.size(), np.nan, dtype=np.float32)       # host-pointer entry point
    tr.invtrans(nf, sp32, host)
    assert np.array_equal(host, gp.cpu().numpy())


@pytest.mark.parametrize("gridname,T,nf", [("F320", 319, 9), ("F64", 63, 1), ("F160", 159, 34), ("O160", 159, 7), ("O320", 319, 19),
                                           ("N320", 319, 4), ("N64", 63, 7), ("N160", 159, 1)])
def test_fp32_two_field_jobs_against_the_one_field_form(gridname, T, nf, monkeypatch):
    """[r4] The fp32 variant's direct and specialised Bluestein rows take TWO fields per job (csrc/fft_pair.h, fft_kernel_pairs.hip:
    lane x / lane y of packed fp32 instructions, the pair's 16 bytes gathered by one LDS-DMA request): odd field counts (the last
    job's second lane is not stored), a single field, reduced grids (direct smooth rows, Bluestein rows plain and row_ct3; [r6] the
    run-time shaped rows of the classic N grids: odd {3,5}-smooth row lengths).  Against the fp64 device result of the float-rounded
    spectra (2e-6), against the one-field form (ATLAS_AMD_FFT_F32_PAIRS=0; same arithmetic, other instruction selection: 1e-6 of
    the largest value), and the array next to the last field stays untouched."""
    g, tr = get_trans(gridname, T)
    sp32 = red_spectra(T, nf, seed=77).astype(np.float32)
    ref = run_device(tr, nf, sp32.astype(np.float64))
    guard = g.size()
    out = {}
    for pairs in ("1", "0"):
        monkeypatch.setenv("ATLAS_AMD_FFT_F32_PAIRS", pairs)
        gp = torch.full(((nf + 1) * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
        tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp[:nf * g.size()])
        tr.synchronize()
        assert bool(torch.isnan(gp[nf * g.size():]).all()), "wrote past the last field"
        out[pairs] = gp[:nf * g.size()].cpu().numpy().astype(np.float64)
        assert compute_rms(out[pairs], ref) < 2e-6, pairs
    assert np.abs(out["1"] - out["0"]).max() <= 1e-6 * np.abs(ref).max()
    assert guard > 0


# ---------------------------------------------------------------- BASELINE configs at their own sizes
def test_config_C4_batch_of_ten_transforms_sampled_rows(trans_full):
    """BASELINE config C4's call: TL1279 -> O1280, 10 x 137 = 1370 fields in ONE invtrans (18 GB of spectra, 72 GB of
    grid points, all resident): sampled rows of sampled fields against the oracle, and field independence against the
    137-field call."""
    g, tr = trans_full
    T, nf = 1279, 1370
    rng = np.random.default_rng(4)
    sp = torch.from_numpy(red_spectra(T, 137, seed=21)).cuda().reshape(-1, 137).repeat(1, 10).contiguous().reshape(-1)
    gp = torch.empty(nf * g.size(), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp, gp)
    tr.synchronize()
    gp137 = torch.empty(137 * g.size(), dtype=torch.float64, device="cuda")
    tr.invtrans(137, dev(red_spectra(T, 137, seed=21)), gp137)
    tr.synchronize()
    v, v137 = gp.view(nf, -1), gp137.view(137, -1)
    for f in (0, 136, 137, 700, 1369):                 # field f of the batch == field f % 137 of the single transform
        assert torch.equal(v[f], v137[f % 137]), f
    rows, _ = rows_of_every_fft_class(tr, extra=[3, 1279, 2100])     # a northern and a southern row of every kernel class
    fields = [5, 640, 1368]
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    sub = np.ascontiguousarray(sp.view(-1, nf)[:, fields].cpu().numpy()).reshape(-1)
    for r, ref in zip(rows, op.invtrans_rows(len(fields), sub, rows, use_fft=True)):
        got = v[fields][:, off[r]:off[r + 1]].cpu().numpy()
        assert compute_rms(got, ref) < 1e-12, r
    del gp, sp, gp137
    torch.cuda.empty_cache()


def test_config_C5_fp32_on_F1280_137_levels_against_the_oracle():
    """BASELINE config C5: TL1279 -> F1280 (N1280's latitudes, full grid), 137 levels, fp32 storage + fp32 MFMA Legendre
    stage.  Oracle: the fp64 CPU restatement on the float-rounded spectra, sampled rows.  Tolerance 2e-6 rel-RMS (fp32
    products accumulated over up to 640 terms of O(1) magnitude: eps_32 * sqrt(K) ~ 1.5e-6)."""
    g = atlas_amd.Grid("F1280")
    T, nf = 1279, 137
    tr = atlas_amd.Trans(g, T)
    sp32 = red_spectra(T, nf, seed=61).astype(np.float32)
    gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp)
    tr.synchronize()
    assert bool(torch.isfinite(gp).all())
    rows, _ = rows_of_every_fft_class(tr, extra=[0, 11, 1279, 1280, 2559])
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    v = gp.view(nf, -1)
    for r, ref in zip(rows, op.invtrans_rows(nf, sp32.astype(np.float64), rows, use_fft=True)):
        got = v[:, off[r]:off[r + 1]].cpu().numpy().astype(np.float64)
        assert compute_rms(got, ref) < 2e-6, r


def test_classic_reduced_gaussian_grid_against_oracle():
    """N<n> by name (BASELINE C5 writes "N1280"): TL63 -> N64 and sampled rows of TL1279 -> N1280"""
    g = atlas_amd.Grid("N64")
    T, nf = 63, 3
    sp = red_spectra(T, nf, seed=71)
    gp = run_device(atlas_amd.Trans(g, T), nf, sp)
    assert compute_rms(gp, oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True)) < 1e-13
    g = atlas_amd.Grid("N1280")
    T, nf = 1279, 5
    sp = red_spectra(T, nf, seed=72)
    tr = atlas_amd.Trans(g, T)
    gp = run_device(tr, nf, sp).reshape(nf, -1)
    rows, _ = rows_of_every_fft_class(tr, extra=[0, 300, 1279, 2559])
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for r, ref in zip(rows, op.invtrans_rows(nf, sp, rows, use_fft=True)):
        assert compute_rms(gp[:, off[r]:off[r + 1]], ref) < 1e-12, r
    # [r6] the same rows in the fp32 variant: the odd {3,5}-smooth row lengths of the classic grids (run-time shaped rows) take two
    # fields per job there (fft_kernel_pairs.hip: fft_rows_pair_kernel); five fields: the last job's second lane is not stored
    assert any(int(g.nx()[r]) % 2 == 1 for r in rows)
    sp32 = sp.astype(np.float32)
    gp32 = torch.full((nf * g.size() + 8,), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp32[:nf * g.size()])
    tr.synchronize()
    assert bool(torch.isnan(gp32[nf * g.size():]).all()) and bool(torch.isfinite(gp32[:nf * g.size()]).all())
    v = gp32[:nf * g.size()].view(nf, -1)
    for r, ref in zip(rows, op.invtrans_rows(nf, sp32.astype(np.float64), rows, use_fft=True)):
        assert compute_rms(v[:, off[r]:off[r + 1]].cpu().numpy().astype(np.float64), ref) < 2e-6, r


LEG_KERNELS = ("classic", "lean") + (("stream",) if "exp" in os.environ.get("ATLAS_AMD_LIB", "") else ())   # "stream" [r6]: experiments build


@pytest.mark.parametrize("case", ["scalar_O160_nf40", "vordiv_F64", "sharded_O160_nf44", "band_O160_nf42", "scalar_O64_nf137",
                                  "f32_O160_nf40", "f32_O64_nf137", "f32_F64_nf44",
                                  "scalar_O160_nf60", "scalar_O64_nf25", "scalar_O64_nf10", "f32_O160_nf60", "f32_O64_nf10",
                                  "scalar_O64_nf1", "scalar_O160_nf5", "scalar_O320_nf70", "scalar_O320_nf100", "f32_O320_nf100"])
def test_legendre_kernel_variants_are_bitwise_equal(case, monkeypatch):
    """The 96-column workgroup of the Legendre stage (field counts whose 16-column tiles come in sixes: nf 33..48, 81..96,
    129..144, ...) has two implementations of the same arithmetic in the same order: the generic template ("classic") and
    the default without vector-ALU work in its stage loop ("lean"), both in fp64 and in the fp32 variant.  (Three more that lost -- "split", "dma", "lean2" -- live in
    tools/experiments and are compared the same way by tools/experiments/test_experiments.py on an experiments build.)
    ATLAS_AMD_LEG_KERNEL is read at every launch; every entry point that reaches the stage must give identical bits."""
    outs = {}
    for kernel in LEG_KERNELS:
        monkeypatch.setenv("ATLAS_AMD_LEG_KERNEL", kernel)
        if case.startswith("f32"):
            # the fp32 variant [r3]: legendre_kernel<3, 2, float> ("classic") against the float instantiation of the lean body
            gridname, T, nf = {"f32_O160_nf40": ("O160", 159, 40), "f32_O64_nf137": ("O64", 63, 137), "f32_F64_nf44": ("F64", 63, 44),
                               "f32_O160_nf60": ("O160", 159, 60), "f32_O64_nf10": ("O64", 63, 10), "f32_O320_nf100": ("O320", 319, 100)}[case]
            g, tr = get_trans(gridname, T)
            sp32 = red_spectra(T, nf, seed=14).astype(np.float32)
            gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
            tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp)
            tr.synchronize()
            outs[kernel] = gp.cpu().numpy()
            assert np.isfinite(outs[kernel]).all()
        elif case.startswith("scalar"):
            # nf = 60: 8 tiles in two column chunks of four (two per wavefront); 25: four tiles in one chunk; 10: two tiles (one per
            # wavefront) -- the narrower instances of the lean body [r3]
            gridname, T, nf = {"scalar_O160_nf40": ("O160", 159, 40), "scalar_O64_nf137": ("O64", 63, 137), "scalar_O160_nf60": ("O160", 159, 60),
                               "scalar_O64_nf25": ("O64", 63, 25), "scalar_O64_nf10": ("O64", 63, 10),
                               # [r6] one tile (1 .. 8 fields): the lean two-group workgroup with its second group on padding columns
                               "scalar_O64_nf1": ("O64", 63, 1), "scalar_O160_nf5": ("O160", 159, 5),
                               # [r6] tiles that do not come in sixes: full 96-column chunks + one narrower launch (9 = 6 + 3, 13 = 12 + 1;
                               # from T = 256 on)
                               "scalar_O320_nf70": ("O320", 319, 70), "scalar_O320_nf100": ("O320", 319, 100)}[case]
            g, tr = get_trans(gridname, T)
            outs[kernel] = run_device(tr, nf, red_spectra(T, nf, seed=11))
        elif case == "vordiv_F64":
            T, ns, nvd = 63, 6, 17          # 2 * 17 wind fields + 6 scalars = 40 fields in one Legendre launch
            g, tr = get_trans("F64", T)
            sp, vor, div = red_spectra(T, ns, 1), red_spectra(T, nvd, 2), red_spectra(T, nvd, 3)
            gp = torch.zeros((ns + 2 * nvd) * g.size(), dtype=torch.float64, device="cuda")
            tr.invtrans(ns, dev(sp), nvd, dev(vor), dev(div), gp)
            tr.synchronize()
            outs[kernel] = gp.cpu().numpy()
        elif case == "sharded_O160_nf44":
            T, nf, nparts = 159, 44, 3
            g = atlas_amd.Grid("O160")
            sp_d, parts = dev(red_spectra(T, nf, seed=12)), []
            for part in range(nparts):
                tr = atlas_amd.Trans(g, T, nparts=nparts, part=part)
                f = torch.zeros(tr.fourier_size(nf), dtype=torch.float64, device="cuda")
                tr.legendre_device(T, nf, sp_d, f)
                tr.synchronize()
                parts.append(f.cpu().numpy())
            outs[kernel] = np.concatenate(parts)
        else:
            T, nf, rows = 159, 42, (200, 320)
            g = atlas_amd.Grid("O160")
            tr = atlas_amd.Trans(g, T, rows=rows)
            gp = torch.zeros(nf * tr.nb_gridpoints(), dtype=torch.float64, device="cuda")
            tr.invtrans(nf, dev(red_spectra(T, nf, seed=13)), gp)
            tr.synchronize()
            outs[kernel] = gp.cpu().numpy()
    assert float(np.abs(outs["classic"]).max()) > 0
    for kernel in LEG_KERNELS[1:]:
        assert np.array_equal(outs["classic"], outs[kernel]), kernel
    if case.startswith("f32"):
        # [r6] the fp32 lean kernel on PAIRS of latitude tiles (128 latitudes x 96 columns per workgroup, 24 MFMAs per wavefront and
        # stage): the same products in the same order per (latitude, column) -- bit-identical, for field counts that take the
        # 96-column workgroup (the others do not have the form and must be unaffected by the switch)
        monkeypatch.setenv("ATLAS_AMD_LEG_KERNEL", "lean")
        monkeypatch.setenv("ATLAS_AMD_LEG_F32_TILES", "2")
        gridname, T, nf = {"f32_O160_nf40": ("O160", 159, 40), "f32_O64_nf137": ("O64", 63, 137), "f32_F64_nf44": ("F64", 63, 44),
                           "f32_O160_nf60": ("O160", 159, 60), "f32_O64_nf10": ("O64", 63, 10), "f32_O320_nf100": ("O320", 319, 100)}[case]
        g, tr = get_trans(gridname, T)
        sp32 = red_spectra(T, nf, seed=14).astype(np.float32)
        gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
        tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp)
        tr.synchronize()
        assert np.array_equal(outs["classic"], gp.cpu().numpy()), "lean, pairs of latitude tiles"


def test_a_constructor_that_throws_releases_its_device_memory():
    """TransLocal's constructor can fail after allocating (a Legendre cache of the wrong size is detected in upload()):
    the stream and the buffers allocated so far must be released, or a caller probing cache files exhausts HBM."""
    T = 159
    g, tr = get_trans("O160", T)
    blob = tr.legendre_cache()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        with pytest.raises(_lib.AtlasAmdError):
            atlas_amd.Trans(g, T, legendre_cache=blob[:-8])
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)   # plans and row tables of one object are a few MB; 40 leaks would be far more


@pytest.mark.parametrize("gridname,T,domain", [("O64", 63, (-5., 5., -2.5, 0.)),      # test_transgeneral.cc:760-767
                                               ("F32", 31, (0., 90., 0., 90.)),        # :1343
                                               ("O160", 159, (350., 372., -20., 35.)),
                                               ("O64", 63, (0., 10., -90., -10.))])   # :1154 (southern hemisphere)
def test_rectangular_domain_crop_is_the_window_of_the_global_transform(gridname, T, domain):
    """trans::Trans(global_grid, domain, truncation) (TransLocal.cc:394-470, 1120-1135): the rows inside the domain are
    transformed at full length and the longitude window is kept, wrapping around.  Must equal, bit for bit, the same
    points of the global transform; the point list is rebuilt here from the grid definition, not from the library."""
    g, tr_global = get_trans(gridname, T)
    nf = 3
    sp = red_spectra(T, nf, seed=21)
    ref = run_device(tr_global, nf, sp).reshape(nf, -1)
    west, east, south, north = domain
    nx, y = g.nx(), g.y()
    off = np.concatenate([[0], np.cumsum(nx)])
    idx = []
    for j in range(len(nx)):
        if south - 1e-6 <= y[j] <= north + 1e-6:
            dx = 360.0 / nx[j]
            lon = np.arange(-2 * nx[j], 2 * nx[j] + 1) * dx        # more than one period either side
            k = np.arange(-2 * nx[j], 2 * nx[j] + 1)[(lon >= west - 1e-6) & (lon <= east + 1e-6)][:nx[j]]
            assert len(k) > 0 and np.all(np.diff(k) == 1)
            idx.extend(off[j] + (k % nx[j]))
    idx = np.array(idx)
    tr = atlas_amd.Trans(g, T, domain=domain)
    assert tr.nb_gridpoints() == len(idx) and int(tr.window_count.sum()) == len(idx)
    gp = torch.zeros(nf * len(idx), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, dev(sp), gp)
    tr.synchronize()
    assert np.array_equal(gp.cpu().numpy().reshape(nf, -1), ref[:, idx])
    # host entry point and the vor/div path on the same crop
    gp_h = np.zeros(nf * len(idx))
    tr.invtrans(nf, sp, gp_h)
    assert np.array_equal(gp_h.reshape(nf, -1), ref[:, idx])
    ns, nvd = 1, 1
    s1, vor, div = red_spectra(T, ns, 31), red_spectra(T, nvd, 32), red_spectra(T, nvd, 33)
    full = np.zeros((ns + 2 * nvd) * g.size())
    tr_global.invtrans(ns, s1, nvd, vor, div, full)
    crop = np.zeros((ns + 2 * nvd) * len(idx))
    tr.invtrans(ns, s1, nvd, vor, div, crop)
    assert np.array_equal(crop.reshape(ns + 2 * nvd, -1), full.reshape(ns + 2 * nvd, -1)[:, idx])


def test_host_pointer_pipeline_is_bitwise_equal_to_the_device_path(monkeypatch):
    """the host-pointer entry point (what atlas__Trans__invtrans_scalar callers pass, TransInterface.h:74-79) runs large calls
    as a full-duplex pipeline over chunks of fields (csrc/trans.hip: invtrans_host_pipelined): upload of chunk c+1, transform
    of chunk c and download of chunk c-1 at the same time, pinned staging buffers reused two chunks later.  Same bits as the
    one-call device path and as the serial host path, for chunk sizes that give 2, 3 and 5 chunks incl. a ragged last one."""
    g, tr = get_trans("O640", 639)
    T, nf = 639, 40                                   # 40 x 1.66 M points x 8 B = 531 MB: above the pipeline's threshold
    sp = red_spectra(T, nf, seed=93)
    ref = run_device(tr, nf, sp)
    for env in ({"ATLAS_AMD_HOST_PIPELINE": "0"}, {}, {"ATLAS_AMD_HOST_CHUNK": "16"}, {"ATLAS_AMD_HOST_CHUNK": "8"},
                {"ATLAS_AMD_HOST_CHUNK": "24"}):
        for k2, v in env.items():
            monkeypatch.setenv(k2, v)
        for rep in range(2):                          # the second call reuses the staging buffers
            gp = np.full(nf * g.size(), np.nan)
            tr.invtrans(nf, sp, gp)
            assert np.array_equal(gp, ref), (env, rep)
        for k2 in env:
            monkeypatch.delenv(k2)


def test_host_pointer_pipeline_falls_back_to_the_serial_path_when_its_buffers_cannot_be_allocated(monkeypatch):
    """a failed hipHostMalloc of the pipeline's pinned staging buffers (memlock limit, memory pressure) used
    to leave raised capacities beside freed pointers -- the next call would have copied into freed memory, release() freed them
    twice.  Now the set is committed only when all eight allocations succeeded; on failure the object holds no staging buffers, the
    call runs the serial pageable path (same bits), and a later call allocates again."""
    g, tr = get_trans("O640", 639)
    T, nf = 639, 40
    sp = red_spectra(T, nf, seed=94)
    ref = run_device(tr, nf, sp)
    # fail with no buffers; succeed (chunks of 16); fail while growing to chunks of 32 (the old set is dropped); succeed again
    for fail, chunk in (("1", "16"), (None, "16"), ("1", "32"), (None, "32")):
        monkeypatch.setenv("ATLAS_AMD_HOST_CHUNK", chunk)
        if fail:
            monkeypatch.setenv("ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC", fail)
        else:
            monkeypatch.delenv("ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC", raising=False)
        gp = np.full(nf * g.size(), np.nan)
        tr.invtrans(nf, sp, gp)
        assert np.array_equal(gp, ref), (fail, chunk)
    monkeypatch.delenv("ATLAS_AMD_HOST_CHUNK", raising=False)


def test_reference_poles_switch_reproduces_the_reference_at_the_south_pole(monkeypatch):
    """INTEGRATION.md "Deviations": by default a row at latitude -90 of a no_nest target is the mirror image of the north-pole
    row; ATLAS_AMD_REFERENCE_POLES=1 reproduces what the reference computes there -- its Legendre routine sets cos(colatitude) =
    +1, sin = 0 for its recurrences within a metre of either pole while its series keep the true colatitude
    (LegendrePolynomials.cc:58-76): a row that is neither pole's polynomials -- which is what the oracle's restatement of that
    routine returns unpatched."""
    T, nf = 63, 4
    sp = red_spectra(T, nf, seed=45)
    lats, west, dlon, nlon = np.array([90.0, 30.0, -89.99999995, -90.0, -30.0]), 0.0, 11.25, 32
    lons = west + dlon * np.arange(nlon)
    want = oracle.invtrans_regional(T, lats, lons, nf, sp)          # the reference's arithmetic, wrong sign at -90 included
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ATLAS_AMD_REFERENCE_POLES", flag)
        rt = atlas_amd.RegionalTrans(nlon, west, dlon, lats, T)
        gp = torch.zeros(nf * nlon * len(lats), dtype=torch.float64, device="cuda")
        rt.invtrans(nf, dev(sp), gp)
        rt.synchronize()
        outs[flag] = gp.cpu().numpy().reshape(nf, len(lats), nlon)
    assert compute_rms(outs["1"].ravel(), want.ravel()) < 1e-13
    assert np.array_equal(outs["1"][:, [0, 1, 4], :], outs["0"][:, [0, 1, 4], :])      # only the polar rows differ
    assert compute_rms(outs["0"][:, 3, :].ravel(), want[:, 3, :].ravel()) > 1e-3          # the default is the mirror image


@pytest.mark.parametrize("case", ["northern_box", "across_equator_unsorted", "with_equator_and_poles"])
def test_regional_target_that_is_not_a_crop_of_a_global_grid(case):
    """TransLocal's no_nest branch (TransLocal.cc:394-406,535-557,719-738,1139-1148) through atlas_amd__RegionalTrans__*:
    arbitrary latitudes, longitudes west + i * dlon.  Against the oracle's restatement of that branch (rel-RMS 1e-13), and
    on points that are points of a global regular grid against the global transform."""
    T, nf = 63, 5
    sp = red_spectra(T, nf, seed=41)
    if case == "northern_box":
        lats, west, dlon, nlon = 61.25 - 0.37 * np.arange(23), -12.5, 0.61, 41
    elif case == "across_equator_unsorted":
        lats, west, dlon, nlon = np.array([-7.1, 12.0, 3.3, -3.3, 25.5, -40.25, 0.4]), 170.0, 1.7, 30   # wraps past 180
    else:
        lats, west, dlon, nlon = np.array([90.0, 45.0, 0.0, -45.0, -90.0, 10.0, -10.0]), 0.0, 11.25, 32
    rt = atlas_amd.RegionalTrans(nlon, west, dlon, lats, T)
    assert rt.nb_gridpoints() == nlon * len(lats)
    lons = west + dlon * np.arange(nlon)
    want = oracle.invtrans_regional(T, lats, lons, nf, sp)
    # Latitude -90: the reference's Legendre routine replaces cos(colatitude) by +1 within a metre of EITHER pole
    # (LegendrePolynomials.cc:74-77), which is wrong at the south pole when the latitude is not mirrored first (this branch
    # does not mirror).  The library returns the mirror image of the north-pole polynomials, Pbar_n^m(-x) = (-1)^(n+m)
    # Pbar_n^m(x): expected value = the oracle at +90 with the spectra's signs flipped accordingly.
    for j in np.flatnonzero(lats < -89.99):
        flip = np.concatenate([[(-1.0) ** (n + m) for n in range(m, T + 1)] for m in range(T + 1)])
        spf = (sp.reshape(-1, 2, nf) * flip[:, None, None]).ravel()
        want[:, j, :] = oracle.invtrans_regional(T, [-lats[j]], lons, nf, spf)[:, 0, :]
    gp = torch.zeros(nf * nlon * len(lats), dtype=torch.float64, device="cuda")
    rt.invtrans(nf, dev(sp), gp)
    rt.synchronize()
    got = gp.cpu().numpy().reshape(nf, len(lats), nlon)
    assert compute_rms(got.ravel(), want.ravel()) < 1e-13
    gp_h = np.zeros(nf * nlon * len(lats))
    rt.invtrans(nf, sp, gp_h)
    assert np.array_equal(gp_h.reshape(got.shape), got)
    # vor/div path (TransLocal.cc:1523-1597) on the same target; the poles are left out (1 / cos(lat) of the clamped latitude
    # amplifies the rounding of u and v there by 6e8)
    if case != "with_equator_and_poles":
        ns, nvd = 2, 2
        s1, vor, div = red_spectra(T, ns, 51), red_spectra(T, nvd, 52), red_spectra(T, nvd, 53)
        want_vd = oracle.invtrans_regional_vordiv(T, lats, lons, ns, s1, nvd, vor, div)
        gp_vd = np.zeros((ns + 2 * nvd) * nlon * len(lats))
        rt.invtrans_vordiv(ns, s1, nvd, vor, div, gp_vd)
        assert compute_rms(gp_vd, want_vd.ravel()) < 1e-12
    if case == "with_equator_and_poles":
        return
    # points of the global regular grid F64: the global transform there (FFT path) within rounding
    g, tr = get_trans("F64", T)
    ref = run_device(tr, nf, sp).reshape(nf, g.ny(), -1)
    rows, i0, step, n = [5, 31, 64, 100, 127], 7, 3, 40
    rt2 = atlas_amd.RegionalTrans(n, i0 * 360.0 / 256, step * 360.0 / 256, g.y()[rows], T)
    gp2 = torch.zeros(nf * n * len(rows), dtype=torch.float64, device="cuda")
    rt2.invtrans(nf, dev(sp), gp2)
    rt2.synchronize()
    sel = ref[:, rows][:, :, i0 + step * np.arange(n)]
    assert compute_rms(gp2.cpu().numpy(), sel.ravel()) < 1e-13


@pytest.mark.parametrize("T,nlat,nlon,nf", [(42, 31, 300, 7), (21, 130, 129, 1), (106, 9, 257, 30)])
def test_regional_target_over_several_tiles_of_the_matrix_product(T, nlat, nlon, nf):
    """the Fourier part of the no_nest branch is one fp64-MFMA matrix product over (target row, field) pairs x longitudes in 128 x 128
    tiles, the contraction in stages of 8 wavenumbers (csrc/dft_gemm.hip): pair counts, row lengths and truncations that are
    not multiples of the tile / stage sizes, several tiles each way -- against the oracle's restatement (TransLocal.cc:719-738,
    1139-1148), scalar and vor/div calls; every element outside the target stays untouched."""
    rng = np.random.default_rng(T)
    lats = np.sort(rng.uniform(-80.0, 80.0, nlat))[::-1].copy()
    west, dlon = -33.0, 360.0 / (nlon + 3)
    lons = west + dlon * np.arange(nlon)
    sp = red_spectra(T, nf, seed=T + 1)
    rt = atlas_amd.RegionalTrans(nlon, west, dlon, lats, T)
    gp = torch.full((nf * nlon * nlat + 64,), float("nan"), dtype=torch.float64, device="cuda")
    rt.invtrans(nf, dev(sp), gp[:-64])
    rt.synchronize()
    got = gp.cpu().numpy()
    assert np.all(np.isnan(got[-64:]))
    want = oracle.invtrans_regional(T, lats, lons, nf, sp)
    assert compute_rms(got[:-64], want.ravel()) < 1e-13
    ns, nvd = 1, max(1, nf // 3)
    s1, vor, div = red_spectra(T, ns, 7), red_spectra(T, nvd, 8), red_spectra(T, nvd, 9)
    want_vd = oracle.invtrans_regional_vordiv(T, lats, lons, ns, s1, nvd, vor, div)
    gp_vd = np.zeros((ns + 2 * nvd) * nlon * nlat)
    rt.invtrans_vordiv(ns, s1, nvd, vor, div, gp_vd)
    assert compute_rms(gp_vd, want_vd.ravel()) < 1e-12


@pytest.mark.parametrize("gridname,T,nf,limit", [("O64", 63, 5, 150), ("N160", 159, 8, 256), ("F160", 159, 3, 256), ("O160", 159, 33, 300),
                                                 ("O640", 639, 40, 1200)])
def test_rows_beyond_the_lds_are_a_matrix_product(gridname, T, nf, limit, monkeypatch):
    """[r6] A row whose transform needs more than 10 080 complex LDS elements (the four longest row lengths of O2560; regular grids beyond
    F5040) is evaluated as a matrix product with a cos / sin table on fp64 MFMA (csrc/dft_gemm.hip, the kernel of the no_nest branch) instead
    of failing the set-up.  The test hook ATLAS_AMD_FFT_LDS_ELEMS lowers that limit (LDS footprints come in blocks of 256 elements) so that the longer rows of
    small grids take the path:
    against the oracle, against the same object without the hook (1e-13 of each other: an FFT and a direct sum), fp32 variant, the
    vor/div call (u, v scaled by 1 / cos(lat) in the product's epilogue), a field subset through the host-pointer pipeline, and what lies
    behind the last field untouched."""
    g = atlas_amd.Grid(gridname)
    sp = red_spectra(T, nf, seed=31)
    small = g.size() < 200000                              # O640: against the default path only (which the other tests pin)
    ref = run_device(atlas_amd.Trans(g, T), nf, sp)
    want = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True) if small else ref
    monkeypatch.setenv("ATLAS_AMD_FFT_LDS_ELEMS", str(limit))
    tr = atlas_amd.Trans(g, T)
    monkeypatch.delenv("ATLAS_AMD_FFT_LDS_ELEMS")          # read when the object is built
    gp = torch.full((nf * g.size() + 16,), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, dev(sp), gp[:-16])
    tr.synchronize()
    got = gp.cpu().numpy()
    assert np.all(np.isnan(got[-16:])) and np.isfinite(got[:-16]).all()
    assert compute_rms(got[:-16], want) < 1e-13
    assert compute_rms(got[:-16], ref) < 1e-13 and not np.array_equal(got[:-16], ref)     # another algorithm on the long rows
    # host pointers (O640 / 40 fields: the full-duplex pipeline over chunks of 16 fields -- later chunks start at a field > 0)
    host = np.full(nf * g.size(), np.nan)
    tr.invtrans(nf, sp, host)
    assert np.array_equal(host, got[:-16])
    if not small:
        return
    # fp32 variant
    sp32 = sp.astype(np.float32)
    gp32 = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp32)
    tr.synchronize()
    assert compute_rms(gp32.cpu().numpy().astype(np.float64), run_device(tr, nf, sp32.astype(np.float64))) < 2e-6
    # vor/div call
    ns, nvd = 1, 2
    s1, vor, div = red_spectra(T, ns, 7), red_spectra(T, nvd, 8), red_spectra(T, nvd, 9)
    w = torch.full(((ns + 2 * nvd) * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(ns, dev(s1), nvd, dev(vor), dev(div), w)
    tr.synchronize()
    wref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans_vordiv(ns, s1, nvd, vor, div, use_fft=True)
    assert compute_rms(w.cpu().numpy(), wref.ravel()) < 1e-12


def test_unstructured_target_points():
    """TransLocal's unstructured path (TransLocal.cc:741-790, 1200-1420; compared there with the structured result in
    test_transgeneral.cc:1336-1490): a list of (lon, lat) points.  Against the oracle's per-point restatement, and the points
    of a structured grid must reproduce the structured transform within rounding."""
    T, nf = 63, 11
    sp = red_spectra(T, nf, seed=61)
    rng = np.random.default_rng(5)
    lons, lats = rng.uniform(-180.0, 540.0, 300), rng.uniform(-89.0, 89.0, 300)
    lats[:7] = [0.0, 45.0, -45.0, 45.0, 12.5, -12.5, 0.0]        # repeated and mirrored latitudes, the equator
    ut = atlas_amd.RegionalTrans.unstructured(lons, lats, T)
    assert ut.nb_gridpoints() == 300
    gp = np.zeros(nf * 300)
    ut.invtrans(nf, sp, gp)
    want = oracle.invtrans_unstructured(T, lons, lats, nf, sp)
    assert compute_rms(gp, want.ravel()) < 1e-13
    gp_d = torch.zeros(nf * 300, dtype=torch.float64, device="cuda")
    ut.invtrans(nf, dev(sp), gp_d)
    ut.synchronize()
    assert np.array_equal(gp_d.cpu().numpy(), gp)
    # every point of O32 as an unstructured target (test_transgeneral.cc:1336-1490 does this with a cropped grid)
    g, tr = get_trans("O32", T)
    nx, y = g.nx(), g.y()
    plon = np.concatenate([np.arange(n) * (360.0 / n) for n in nx])
    plat = np.concatenate([np.full(n, v) for n, v in zip(nx, y)])
    ref = run_device(tr, nf, sp)
    ut2 = atlas_amd.RegionalTrans.unstructured(plon, plat, T)
    gp2 = np.zeros(nf * len(plon))
    ut2.invtrans(nf, sp, gp2)
    # the structured path truncates the Fourier sum towards the poles (nlat0), the unstructured one does not: compare where
    # every wavenumber is kept
    full = np.array([oracle.fourier_truncation(T, int(nx[j]), int(nx.max()), len(nx), math.radians(y[j]), False) >= T
                     for j in range(len(nx))])
    keep = np.concatenate([np.full(n, k) for n, k in zip(nx, full)])
    assert keep.sum() > 1000
    assert compute_rms(gp2.reshape(nf, -1)[:, keep].ravel(), ref.reshape(nf, -1)[:, keep].ravel()) < 1e-13
    # vor/div -> u, v
    ns, nvd = 1, 2
    s1, vor, div = red_spectra(T, ns, 71), red_spectra(T, nvd, 72), red_spectra(T, nvd, 73)
    parts = oracle.invtrans_regional_vordiv   # per point through the regional restatement
    want_vd = np.stack([parts(T, [lats[i]], [lons[i]], ns, s1, nvd, vor, div)[:, 0, 0] for i in range(40)], axis=1)
    ut3 = atlas_amd.RegionalTrans.unstructured(lons[:40], lats[:40], T)
    gp3 = np.zeros((ns + 2 * nvd) * 40)
    ut3.invtrans_vordiv(ns, s1, nvd, vor, div, gp3)
    assert compute_rms(gp3, want_vd.ravel()) < 1e-12
