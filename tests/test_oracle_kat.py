"""Pins the ORACLE (oracle/translocal_oracle.c) against the reference's own known-answer tests and against
independent evaluations, on the CPU:
  * analytic spherical harmonics of src/tests/trans/test_transgeneral.cc:80-374 (closed forms n<=3, sectoral n=m),
    with the reference's row mask `fourier_truncation > m` (:433-449) and tolerance 1e-13 (:534)
  * wind known answers (:286-371), tolerance 2e-6 (:538)
  * c2r contract of linalg::FFT (FFT.h:27-72) against numpy.fft.irfft (pocketfft)
  * the 2x2 GEMM known answer of src/tests/linalg/test_linalg_dense.cc:117-136 through the oracle's GEMM order
"""
import math

import numpy as np
import pytest

import atlas_amd
import oracle
from helpers import (CLOSED_FORMS, analytic_scalar, compute_rms, pbar, red_spectra, unit_spectrum, wind_kat)


def grid_arrays(name):
    g = atlas_amd.Grid(name)
    return g, g.nx(), g.y()


def row_mask(T, nx, lat, regular, m):
    """rows on which wavenumber m survives in the reference test: ftrc > m (test_transgeneral.cc:445)"""
    ny = len(nx)
    return [oracle.fourier_truncation(T, int(nx[j]), int(nx.max()), ny, lat[j] * math.pi / 180.0, regular) > m
            for j in range(ny)]


def test_closed_forms_match_general_formula():
    x = np.linspace(-0.99, 0.99, 41)
    for (n, m), f in CLOSED_FORMS.items():
        ref = f(x, np.sqrt(1 - x * x))
        assert np.allclose(pbar(n, m, x), ref, rtol=1e-13, atol=1e-14), (n, m)


def test_legendre_lat_against_closed_forms():
    trc = 64
    for lat in [0.0, 0.3, -1.1, 1.5]:
        lp = oracle.legendre_lat(trc, lat)
        s, c = math.sin(lat), math.cos(lat)
        for (n, m), f in CLOSED_FORMS.items():
            idx = (2 * trc + 3 - m) * m // 2 + n - m
            assert abs(lp[idx] - f(s, c)) < 2e-14, (n, m, lat)
        # sectoral harmonics up to 45 (test_transgeneral.cc:258-272): Pbar_n^n = sqrt((2n+1)!!/(2n)!!) cos^n
        for n in range(1, 46):
            idx = (2 * trc + 3 - n) * n // 2
            ref = math.sqrt(math.prod((2 * k + 1) / (2 * k) for k in range(1, n + 1))) * c ** n
            assert abs(lp[idx] - ref) <= 3e-13 * max(1.0, abs(ref)), (n, lat)


@pytest.mark.parametrize("gridname,T", [("F64", 63), ("O64", 63)])
def test_oracle_scalar_analytic(gridname, T):
    """test_trans_vordiv_with_translib / test_trans_domain style: every unit coefficient with a closed form"""
    g, nx, lat = grid_arrays(gridname)
    op = oracle.OraclePlan(T, nx, lat)
    cases = [(n, m) for (n, m) in CLOSED_FORMS] + [(n, n) for n in (4, 7, 13, 21, 33, 45)]
    worst = 0.0
    for n, m in cases:
        for imag in (0, 1):
            if m == 0 and imag == 1:
                continue
            sp = unit_spectrum(T, 1, n, m, imag)
            gp = op.invtrans(1, sp)
            ref = analytic_scalar(nx, lat, n, m, imag, row_mask(T, nx, lat, g.regular(), m))
            worst = max(worst, compute_rms(gp, ref))
    assert worst < 1e-13, worst


def test_oracle_fft_equals_direct_on_transform():
    g, nx, lat = grid_arrays("O32")
    op = oracle.OraclePlan(31, nx, lat)
    sp = red_spectra(31, 3)
    assert compute_rms(op.invtrans(3, sp, use_fft=True), op.invtrans(3, sp, use_fft=False)) < 1e-15


@pytest.mark.parametrize("gridname,T,nf", [("O32", 31, 3), ("F32", 31, 2), ("O48", 95, 2), ("O160", 159, 4)])
def test_blas_pocketfft_variant_equals_plain_oracle(gridname, T, nf):
    """oracle/translocal_blas.py (BLAS dgemm + pocketfft, the tuned CPU baseline of bench.py) against the plain
    restatement: same algorithm and tables, only the summation order of the two library kernels differs"""
    from oracle.translocal_blas import invtrans_blas
    g, nx, lat = grid_arrays(gridname)
    op = oracle.OraclePlan(T, nx, lat)
    sp = red_spectra(T, nf, seed=4)
    ref = op.invtrans(nf, sp, use_fft=True)
    assert compute_rms(invtrans_blas(op, nf, sp), ref) < 1e-15
    # the threaded form (bench.py's cpu_baseline: a pool over wavenumbers / row groups) computes the same numbers and reports
    # where its wall clock went
    tm = {}
    assert np.array_equal(invtrans_blas(op, nf, sp, workers=4, timings=tm), invtrans_blas(op, nf, sp))
    assert tm["threads"] == 4 and tm["legendre_s"] > 0 and tm["fourier_s"] > 0 and 0 <= tm["layout_s"] <= tm["legendre_s"] + tm["fourier_s"]


def test_oracle_rows_equals_table_path():
    g, nx, lat = grid_arrays("O32")
    T, nf = 31, 4
    op = oracle.OraclePlan(T, nx, lat)
    sp = red_spectra(T, nf)
    gp = op.invtrans(nf, sp).reshape(nf, -1)
    off = np.concatenate([[0], np.cumsum(nx)])
    rows = [0, 5, 31, 32, 50, 63]
    for r, blk in zip(rows, op.invtrans_rows(nf, sp, rows)):
        assert np.array_equal(blk, gp[:, off[r]:off[r + 1]])


def test_oracle_wind_analytic():
    T = 63
    g, nx, lat = grid_arrays("F64")
    op = oracle.OraclePlan(T, nx, lat)
    worst = 0.0
    for ivar_in in (0, 1):
        for (n, m) in [(1, 0), (1, 1)]:
            for imag in (0, 1):
                if m == 0 and imag == 1:
                    continue
                coef = unit_spectrum(T, 1, n, m, imag)
                zero = np.zeros_like(coef)
                vor, div = (coef, zero) if ivar_in == 0 else (zero, coef)
                gp = op.invtrans_vordiv(0, None, 1, vor, div).reshape(2, -1)
                for ivar_out in (0, 1):
                    ref = np.concatenate([
                        wind_kat(ivar_in, ivar_out, n, m, imag, np.arange(k) * (2 * math.pi / k), y * math.pi / 180.0)
                        for k, y in zip(nx, lat)])
                    worst = max(worst, compute_rms(gp[ivar_out], ref))
    assert worst < 2e-6, worst


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 12, 20, 36, 44, 60, 76, 124, 148, 200, 212, 1283 * 4, 641 * 2, 2048, 5136])
def test_c2r_contract_against_pocketfft(n):
    rng = np.random.default_rng(n)
    nc = n // 2 + 1
    x = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
    xx = x.copy()
    xx[0] = xx[0].real
    if n % 2 == 0:
        xx[-1] = xx[-1].real
    ref = np.fft.irfft(xx, n) * n  # unnormalised c2r
    assert compute_rms(oracle.c2r_direct(n, x), ref) < 5e-15
    assert compute_rms(oracle.c2r_fft(n, x), ref) < 5e-15


def test_fourier_truncation_known_values():
    # linear / quadratic / cubic branches of TransLocal.cc:272-300, hand-evaluated
    ft = oracle.fourier_truncation
    assert ft(63, 256, 256, 128, 0.3, True) == 63       # full grid: (nx-1)/2 = 127 -> min(T, .)
    assert ft(127, 20, 256, 128, 1.5, False) == 9       # linear (T >= ndgl-1): (20-1)/2
    assert ft(63, 20, 272, 128, 88.9277 * math.pi / 180, False) == 8   # cubic: 19/(2+cos^2) - 1 -> 8
    assert ft(100, 200, 400, 128, 0.0, False) == 99     # quadratic: weight = 3*(127-100)/128 = 0 (INTEGER division) -> 199/2


def test_gemm_known_answer():
    # src/tests/linalg/test_linalg_dense.cc:117-136: [[1,-2],[-4,2]]^2 = [[9,-6],[-12,12]] (column-major C = A*B)
    # through the oracle's own GEMM (column-major operands, the contraction of the Legendre stage)
    A = np.array([[1., -2.], [-4., 2.]])
    C = oracle.gemm(np.asfortranarray(A), np.asfortranarray(A))
    assert np.array_equal(C, np.array([[9., -6.], [-12., 12.]]))
    rng = np.random.default_rng(5)
    A, B = rng.standard_normal((7, 5)), rng.standard_normal((5, 9))
    assert np.allclose(oracle.gemm(A, B), A @ B, rtol=1e-14, atol=1e-14)


def test_legendre_functions_at_high_degree_against_mpmath():
    """Independent pin of the Legendre tables beyond the degrees the reference's own tests reach (closed forms n <= 3,
    sectoral n <= 45): 240 values of the normalised P^m_n(sin lat) with n up to 1280 computed by mpmath in 60-digit
    arithmetic with the standard three-term recurrence in n (tests/golden/legendre_mpmath.json, written by
    tests/golden/make_legendre_mpmath_fixture.py) against orc_legendre_lat, the restatement of
    LegendrePolynomials.cc:47-151 (Fourier-series start + Belousov recurrence in fp64).  Measured worst difference
    2.1e-12 absolute (n = 1280, m = 0 at 89.9 degrees, value 4.6): the rounding of the fp64 recurrence itself."""
    import json
    import os
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legendre_mpmath.json")))
    trc, cache, worst = 1280, {}, 0.0
    assert len(fix["samples"]) >= 200 and max(s["n"] for s in fix["samples"]) == 1280
    for s in fix["samples"]:
        lat = s["lat_rad"]
        if lat not in cache:
            cache[lat] = oracle.legendre_lat(trc, lat)
        n, m = s["n"], s["m"]
        got = cache[lat][(2 * trc + 3 - m) * m // 2 + n - m]
        worst = max(worst, abs(got - s["value"]))
        assert abs(got - s["value"]) <= 5e-12 * max(1.0, abs(s["value"])), (n, m, lat, got, s["value"])
    assert worst > 0.0      # two different computations, not the same number twice


def test_regional_no_fft_branch_restatement_is_pinned_by_the_global_oracle():
    """oracle.invtrans_regional restates TransLocal's no_nest branch (Legendre polynomials at the target's latitudes + DFT as a
    matrix product, TransLocal.cc:394-406,535-557,719-738,1139-1148).  On a target whose points are points of a global
    regular Gaussian grid it must give what the pinned global oracle gives there -- and the closed forms quoted by
    test_transgeneral.cc:117-272 at arbitrary points."""
    import math
    import oracle
    from helpers import CLOSED_FORMS, red_spectra
    T, N, nf = 31, 32, 3
    x, _ = np.polynomial.legendre.leggauss(2 * N)
    lat = np.degrees(np.arcsin(x[::-1]))
    nx = np.full(2 * N, 4 * N, dtype=np.int32)
    op = oracle.OraclePlan(T, nx, lat)
    sp = red_spectra(T, nf, seed=7)
    ref = op.invtrans(nf, sp).reshape(nf, 2 * N, 4 * N)
    rows, cols = [3, 17, 40, 63], np.arange(5, 50, 3)
    got = oracle.invtrans_regional(T, lat[rows], cols * (360.0 / (4 * N)), nf, sp)
    want = ref[:, rows][:, :, cols]
    assert np.sqrt(((got - want) ** 2).sum() / (want ** 2).sum()) < 1e-13
    # the vor/div path of the same branch against the global oracle's
    ns, nvd = 2, 2
    s1, vor, div = red_spectra(T, ns, 1), red_spectra(T, nvd, 2), red_spectra(T, nvd, 3)
    ref = op.invtrans_vordiv(ns, s1, nvd, vor, div).reshape(ns + 2 * nvd, 2 * N, 4 * N)
    got = oracle.invtrans_regional_vordiv(T, lat[rows], cols * (360.0 / (4 * N)), ns, s1, nvd, vor, div)
    want = ref[:, rows][:, :, cols]
    assert np.sqrt(((got - want) ** 2).sum() / (want ** 2).sum()) < 1e-13
    lats, lons = np.array([71.3, 12.0, -33.33, -80.5]), 10.0 + 0.7 * np.arange(9)
    for (n, m), form in CLOSED_FORMS.items():
        for imag in ((0, 1) if m > 0 else (0,)):
            sp1 = np.zeros(((T + 1) * (T + 2) // 2, 2, 1))
            sp1[(2 * T + 3 - m) * m // 2 + n - m, imag, 0] = 1.0
            got = oracle.invtrans_regional(T, lats, lons, 1, sp1)[0]
            P = np.array([form(math.sin(math.radians(v)), math.cos(math.radians(v))) for v in lats])
            lam = np.radians(lons)
            want = P[:, None] * ((2.0 if m > 0 else 1.0) * (np.cos(m * lam) if imag == 0 else -np.sin(m * lam)))[None, :]
            assert np.abs(got - want).max() < 1e-13 * max(1.0, np.abs(want).max()), (n, m, imag)
