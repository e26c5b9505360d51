"""Edges of the transform the reference runs through the same code path as everything else (FFTW plans any nx, the GEMMs take any
shape: TransLocal.cc:652-686,1101-1196) and that are their own kernels, plans or branches here: odd / prime / tiny row lengths of
a hand-made reduced grid, truncations 0 and 1, a truncation far above what the grid resolves, zero fields, field counts either
side of the 96-column chunk of the Legendre workgroup."""
import numpy as np
import pytest

import atlas_amd
from helpers import compute_rms, red_spectra
from oracle import translocal as oracle

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
TOL = 1e-13


def _device_field(tr, nf, sp, npts):
    gp = torch.full((nf * npts,), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp).cuda(), gp)
    tr.synchronize()
    return gp.cpu().numpy()


@pytest.mark.parametrize("T", [15, 31])
def test_ragged_grid_with_odd_prime_and_tiny_rows(T):
    """rows of 2 .. 49 points: even lengths below every specialised shape, odd {3,5}-smooth lengths (9, 15, 25, 27: complex transform
    of the row itself), odd lengths with other factors (7, 11, 13, 17, 19, 23, 49: plain DFT rows)"""
    y = atlas_amd.Grid("F16").y()
    half = [2, 3, 4, 5, 6, 7, 9, 11, 13, 15, 17, 19, 23, 25, 27, 49]
    nx = np.array(half + half[::-1], dtype=np.int32)
    g = atlas_amd.StructuredGrid(nx=nx, y=y)
    assert g.size() == int(nx.sum())
    tr = atlas_amd.Trans(g, T)
    plan = oracle.OraclePlan(T, nx, y)
    for nf in (1, 3):
        sp = red_spectra(T, nf, seed=T + nf)
        got = _device_field(tr, nf, sp, g.size())
        assert np.isfinite(got).all()
        assert compute_rms(got, plan.invtrans(nf, sp, use_fft=False)) < TOL
    # the host-pointer entry point on the same grid
    sp = red_spectra(T, 2, seed=3)
    out = np.full(2 * g.size(), np.nan)
    tr.invtrans(2, sp, out)
    assert compute_rms(out, plan.invtrans(2, sp, use_fft=False)) < TOL


@pytest.mark.parametrize("gridname,T", [("O8", 0), ("O8", 1), ("F8", 0), ("O8", 63), ("F16", 95), ("N16", 40)])
def test_truncations_at_both_ends(gridname, T):
    """T = 0 (one coefficient, which the reference drops), T = 1, and truncations far above what the rows resolve (fourier_truncation
    cuts the zonal wavenumbers per row, TransLocal.cc:272-300; the Legendre sums still run to n = T)"""
    g = atlas_amd.Grid(gridname)
    tr = atlas_amd.Trans(g, T)
    assert tr.nb_spectral_coefficients() == (T + 1) * (T + 2)
    nf = 2
    sp = red_spectra(T, nf, seed=9)
    got = _device_field(tr, nf, sp, g.size())
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=False)
    assert np.isfinite(got).all() and compute_rms(got, ref) < TOL
    if T == 0:
        # TransLocal keeps the zonal wavenumbers jm < truncation only (TransLocal.cc:966-1003; tests/test_gpu_trans.py:110): at
        # T = 0 nothing is left and the field is zero, not the constant -- the reference's behaviour, reproduced
        assert not ref.any() and not got.any()


def test_zero_fields_is_a_no_op():
    g = atlas_amd.Grid("O16")
    tr = atlas_amd.Trans(g, 15)
    gp = torch.full((g.size(),), 7.0, dtype=torch.float64, device="cuda")
    sp = torch.zeros(tr.nb_spectral_coefficients(), dtype=torch.float64, device="cuda")
    tr.invtrans(0, sp, gp)
    tr.synchronize()
    assert bool((gp == 7.0).all())
    out = np.full(g.size(), 7.0)
    tr.invtrans(0, np.zeros(tr.nb_spectral_coefficients()), out)
    assert (out == 7.0).all()
    # a vor/div call with no scalars, and one with no vor/div pairs
    wind = np.full(2 * g.size(), np.nan)
    vor, div = red_spectra(15, 1, seed=1), red_spectra(15, 1, seed=2)
    tr.invtrans(0, None, 1, vor, div, wind)
    ref = np.full(2 * g.size(), np.nan)
    tr.invtrans_vordiv2wind(1, vor, div, ref)
    assert np.array_equal(wind, ref)
    sc = np.full(g.size(), np.nan)
    tr.invtrans(1, vor, 0, None, None, sc)
    ref = np.full(g.size(), np.nan)
    tr.invtrans(1, vor, ref)
    assert np.array_equal(sc, ref)


@pytest.mark.parametrize("nf", [47, 48, 49, 95, 96, 97, 143, 144, 145])
def test_field_counts_around_the_column_chunk(nf):
    """the Legendre workgroup takes 96 columns = 48 fields (real, imaginary); the last chunk is partly filled, its columns beyond the
    last field must not reach the output; every field is checked on its own (a column landing in the wrong field would show)"""
    T = 31
    g = atlas_amd.Grid("O32")
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, nf, seed=nf)
    got = _device_field(tr, nf, sp, g.size()).reshape(nf, -1)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=False).reshape(nf, -1)
    worst = max(compute_rms(got[f], ref[f]) for f in range(nf))
    assert worst < TOL, worst


def test_twenty_thousand_fields_in_one_call():
    """far more fields than any launch dimension was sized for by the benchmark configurations (1370): 20 000 fields on O32 in one
    call, seven distinct spectra repeated -- every copy equals the seven-field call bit for bit (fields are independent columns of
    the two GEMMs, TransLocal.cc:1040-1070, and independent transforms of the FFT)"""
    T, ndist, nf = 31, 7, 20000
    g = atlas_amd.Grid("O32")
    tr = atlas_amd.Trans(g, T)
    base = red_spectra(T, ndist, seed=77).reshape(-1, ndist)
    want = _device_field(tr, ndist, base.reshape(-1).copy(), g.size()).reshape(ndist, -1)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(ndist, base.reshape(-1).copy(), use_fft=False).reshape(ndist, -1)
    assert compute_rms(want, ref) < TOL
    sp = torch.from_numpy(base).cuda()[:, torch.arange(nf, device="cuda") % ndist].contiguous()
    gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp.view(-1), gp)
    tr.synchronize()
    gp = gp.view(nf, -1)
    wd = torch.from_numpy(want).cuda()
    for k in range(ndist):
        assert bool((gp[k::ndist] == wd[k]).all()), k


def test_creating_and_destroying_transforms_does_not_leak_device_memory():
    g = atlas_amd.Grid("O64")
    sp = torch.from_numpy(red_spectra(63, 3)).cuda()
    gp = torch.zeros(3 * g.size(), dtype=torch.float64, device="cuda")

    def cycle():
        tr = atlas_amd.Trans(g, 63)
        tr.invtrans(3, sp, gp)
        tr.synchronize()
        del tr

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0, free1)     # 40 objects of ~10 MB each would show
