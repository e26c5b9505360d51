"""GPU parity tests of the experiments that are NOT in the product library (tools/experiments/*.inc).
Build the experiments library and point the loader at it:
    make -C atlas_amd/csrc experiments
    ATLAS_AMD_LIB=atlas_amd/lib/dev/libatlas_amd_exp.so python -m pytest tools/experiments/test_experiments.py -q
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import atlas_amd  # noqa: E402
import oracle  # noqa: E402
import test_gpu_trans as product_tests  # noqa: E402
from helpers import compute_rms, red_spectra  # noqa: E402
from test_gpu_trans import run_device  # noqa: E402

pytestmark = pytest.mark.skipif("exp" not in os.environ.get("ATLAS_AMD_LIB", ""),
                                reason="needs ATLAS_AMD_LIB=.../libatlas_amd_exp.so (make -C atlas_amd/csrc experiments)")


# ---- host run of the dense-stage rows (tools/experiments/fft_hybrid_core.h): no GPU needed, but an experiments build [moved here r4]
import math  # noqa: E402

from atlas_amd import _lib  # noqa: E402

HYBRID_LENGTHS = [28, 44, 52, 68, 76, 132, 140, 148, 244, 260, 404, 1004, 2 * 514, 4 * 61 * 10, 4 * 7 * 183, 4 * 1285,
                  2 * 2 * 3 * 7 * 61, 4 * 1283]   # the last two: no dense-stage plan (A > 257), usual plan


@pytest.mark.parametrize("n", HYBRID_LENGTHS)
def test_fft_hybrid_phase_code_against_pocketfft(n):
    """host run of the dense-stage rows of fft_core.h (fold + symmetric split, cos / sin matrix stage, native stages)"""
    rng = np.random.default_rng(n)
    nc = n // 2 + 1
    for mmax in (nc - 1, max(0, n // 3), 0):
        x = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
        x[mmax + 1:] = 0
        out = np.zeros(n)
        _lib.check(_lib.fft_host_row_hybrid(n, np.ascontiguousarray(x).ctypes.data, mmax, out.ctypes.data))
        xx = x.copy()
        xx[0] = xx[0].real
        xx[-1] = xx[-1].real
        assert compute_rms(out, np.fft.irfft(xx, n) * n) < 2e-15, (n, mmax)


def test_fft_hybrid_phase_code_on_every_row_length_of_O1280():
    N, T = 1280, 1279
    g = atlas_amd.Grid(f"O{N}")
    rng = np.random.default_rng(7)
    nx, y = g.nx(), g.y()
    worst = 0.0
    for j in range(N):
        n = int(nx[j])
        nc = n // 2 + 1
        mmax = _lib.fourier_truncation(T, n, g.nxmax(), 2 * N, math.radians(y[j]), 0)
        x = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
        x[mmax + 1:] = 0
        out = np.zeros(n)
        _lib.check(_lib.fft_host_row_hybrid(n, np.ascontiguousarray(x).ctypes.data, mmax, out.ctypes.data))
        xx = x.copy()
        xx[0] = xx[0].real
        xx[-1] = xx[-1].real
        worst = max(worst, compute_rms(out, np.fft.irfft(xx, n) * n))
    assert worst < 2e-15, worst


def test_hybrid_fourier_rows_equal_the_bluestein_rows(monkeypatch):
    """the dense-stage ("hybrid", opt-in) Fourier kernel -- radix-A DFT on the fp64 matrix cores + radix-{2..9} stages --
    against the Bluestein kernels on every row of O320 / TL319 and against the oracle on sampled rows"""
    g = atlas_amd.Grid("O320")
    T, nf = 319, 9
    sp = red_spectra(T, nf, seed=81)
    monkeypatch.setenv("ATLAS_AMD_FFT_HYBRID", "0")
    a = run_device(atlas_amd.Trans(g, T), nf, sp)
    monkeypatch.setenv("ATLAS_AMD_FFT_HYBRID", "1")
    b = run_device(atlas_amd.Trans(g, T), nf, sp)
    assert compute_rms(b, a) < 1e-14 and not np.array_equal(a, b)      # different arithmetic, same transform
    rows = [40, 63, 200, 319, 500]
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for r, ref in zip(rows, op.invtrans_rows(nf, sp, rows, use_fft=True)):
        assert compute_rms(b.reshape(nf, -1)[:, off[r]:off[r + 1]], ref) < 1e-13, r



@pytest.mark.parametrize("case", ["scalar_O160_nf40", "vordiv_F64", "sharded_O160_nf44", "band_O160_nf42", "scalar_O64_nf137"])
def test_legendre_experiment_kernels_are_bitwise_equal(case, monkeypatch):
    """"lean2" (pairs of latitude tiles), "split" (role-split) and "dma" (both operands by LDS-DMA) against "classic" / "lean": the product test with two more
    kernel names"""
    monkeypatch.setattr(product_tests, "LEG_KERNELS", ("classic", "lean", "lean2", "split", "dma"))
    product_tests.test_legendre_kernel_variants_are_bitwise_equal.__wrapped__(case, monkeypatch) \
        if hasattr(product_tests.test_legendre_kernel_variants_are_bitwise_equal, "__wrapped__") \
        else product_tests.test_legendre_kernel_variants_are_bitwise_equal(case, monkeypatch)


def test_half_window_bluestein_rows_are_bitwise_equal_to_the_product_rows(monkeypatch):
    """[r4] the [R0,16,16] rows with LDS as a half-row exchange window (tools/experiments/fft_halfwin_rows.inc,
    ATLAS_AMD_FFT_HALFWIN=1): same arithmetic in the same order as row_phase_ct / row_ct3 -- bit-identical grid points on TL1279 ->
    O1280 (every class M >= 3840 occurs), 9 fields"""
    g = atlas_amd.Grid("O1280")
    T, nf = 1279, 9
    sp = red_spectra(T, nf, seed=19)
    ref = run_device(atlas_amd.Trans(g, T), nf, sp)
    monkeypatch.setenv("ATLAS_AMD_FFT_HALFWIN", "1")
    got = run_device(atlas_amd.Trans(g, T), nf, sp)
    assert np.array_equal(got, ref)
