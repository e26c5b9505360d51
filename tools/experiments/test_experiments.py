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


# ---- native mixed-radix rows (tools/experiments/fft_native*.{h,hip}, fft_native_plan.inc): in the product library in round 4 (opt-in),
# ---- moved here in round 5 -- at parity with the Bluestein rows they replace, never the default
import test_host_logic as _host_tests  # noqa: E402,F401  (helpers of the CPU test below)
from helpers import rows_of_every_fft_class  # noqa: E402

TOL = 1e-13
torch = pytest.importorskip("torch")


def test_native_mixed_radix_rows_on_every_row_length_of_O1280_that_has_a_plan():
    """[r4] the native mixed-radix rows (tools/experiments/fft_native.h; opt-in on the device): every distinct row length of O1280 for which the
    planner finds a stage list -- half lengths with any number of primes 2..13 and at most one prime 17..31 that are not a length
    of the specialised direct family: 428 of the 1280 lengths, 25 % of the grid points -- through the host run of the kernel's own
    tables (fold permutation, per-stage butterfly tables) and butterflies (dense odd-prime radices included), with the row's own
    Fourier truncation, against pocketfft.  What FFTW / pocketfft do natively for the reference (linalg/fft/FFTW.cc:38-61)."""
    g = atlas_amd.Grid("O1280")
    T, N = 1279, 1280
    rng = np.random.default_rng(4)
    nx, y = g.nx(), g.y()
    worst, nat, pts = 0.0, 0, 0
    firsts = set()
    for j in range(N):
        n = int(nx[j])
        info = np.zeros(16, dtype=np.int32)
        _lib.check(_lib.fft_plan_info(n, 1, info.ctypes.data))
        if info[0] != 5:                       # FFT_NATIVE
            continue
        ns, rad = int(info[3]), [int(v) for v in info[4:4 + int(info[3])]]
        assert 2 <= ns <= 4 and int(np.prod(rad)) == n // 2 == info[1]
        assert rad[-1] % 2 == 1 and rad[-1] <= 31 and all(r <= 16 for r in rad[:-1])     # DIF order: the first executed stage is last
        assert info[13] % 2 == 1 and info[13] >= n // 2 // rad[0]                         # odd LDS pitch of the top-level blocks
        assert info[2] >= n // 2 + 1
        firsts.add(rad[-1])
        nat += 1
        pts += n
        nc = n // 2 + 1
        mmax = _lib.fourier_truncation(T, n, g.nxmax(), 2 * N, math.radians(y[j]), 0)
        x = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
        x[mmax + 1:] = 0
        out = np.zeros(n)
        _lib.check(_lib.fft_host_row_native(n, np.ascontiguousarray(x).ctypes.data, mmax, out.ctypes.data))
        xx = x.copy()
        xx[0] = xx[0].real
        xx[-1] = xx[-1].real
        worst = max(worst, compute_rms(out, np.fft.irfft(xx, n) * n))
    assert nat == 428 and abs(pts / float(np.sum(nx[:N])) - 0.2523) < 1e-3
    assert firsts == {3, 5, 7, 9, 11, 13, 15, 17, 19, 23, 29, 31}
    assert worst < 1e-15, worst


def test_native_plan_is_off_by_default_and_refuses_lengths_without_a_stage_list():
    info = np.zeros(16, dtype=np.int32)
    _lib.check(_lib.fft_plan_info(2 * 1190, 0, info.ctypes.data))          # h = 2 * 5 * 7 * 17
    assert info[0] == 1                                                    # Bluestein unless asked for
    _lib.check(_lib.fft_plan_info(2 * 1190, 1, info.ctypes.data))
    assert info[0] == 5 and sorted(info[4:4 + info[3]]) == [7, 10, 17]
    for n in (2 * 2 * 641, 2 * 17 * 19 * 4, 2 * 37 * 32, 2 * 1024, 2 * 2560):   # big prime, two primes > 13, prime > 31, family lengths
        _lib.check(_lib.fft_plan_info(n, 1, info.ctypes.data))
        assert info[0] != 5, n
    out = np.zeros(2 * 1282)
    assert _lib.fft_host_row_native(2 * 1282, np.zeros(1283, dtype=np.complex128).ctypes.data, 10, out.ctypes.data) != 0



def test_native_mixed_radix_rows_against_the_oracle(monkeypatch):
    """[r4] ATLAS_AMD_FFT_NATIVE=1: rows whose half length has a native stage list take fft_rows_nat_kernel (one kernel for every
    shape, tools/experiments/fft_native_impl.h) instead of a Bluestein row.  The whole O320 / T319 field in fp64 against the oracle (19 fields:
    two full field groups and one of three), repeated calls bit-identical (the kernel's L2 prefetch requests must not land in
    anybody's registers), and the fp32 variant against the fp64 device result of the float-rounded spectra."""
    monkeypatch.setenv("ATLAS_AMD_FFT_NATIVE", "1")
    g = atlas_amd.Grid("O320")
    T, nf = 319, 19
    tr = atlas_amd.Trans(g, T)
    cls = tr.fft_row_classes()
    assert (cls[:, 2] == 4).sum() >= 300                      # most rows of this grid have a native plan
    sp = red_spectra(T, nf, seed=41)
    gp = run_device(tr, nf, sp)
    ref = oracle.OraclePlan(T, g.nx(), g.y()).invtrans(nf, sp, use_fft=True)
    assert compute_rms(gp, ref) < TOL
    for _ in range(4):
        assert np.array_equal(run_device(tr, nf, sp), gp)
    sp32 = sp.astype(np.float32)
    ref32 = run_device(tr, nf, sp32.astype(np.float64))
    gp32 = torch.full((nf * g.size(),), float("nan"), dtype=torch.float32, device="cuda")
    tr.invtrans(nf, torch.from_numpy(sp32).cuda(), gp32)
    tr.synchronize()
    assert bool(torch.isfinite(gp32).all())
    assert compute_rms(gp32.cpu().numpy().astype(np.float64), ref32) < 2e-6
    assert tr.fourier_launch_plan()["native_two_fields"] == 0
    monkeypatch.setenv("ATLAS_AMD_FFT_NATIVE_FPJ", "2")                         # two fields per workgroup: same arithmetic per field
    tr2 = atlas_amd.Trans(g, T)
    assert tr2.fourier_launch_plan()["native_two_fields"] >= 1                  # read per object: the other path really ran
    assert np.array_equal(run_device(tr2, nf, sp), gp)
    monkeypatch.delenv("ATLAS_AMD_FFT_NATIVE_FPJ")
    monkeypatch.delenv("ATLAS_AMD_FFT_NATIVE")
    assert (atlas_amd.Trans(g, T).fft_row_classes()[:, 2] != 4).all()          # opt-in: off by default


def test_native_mixed_radix_rows_at_full_size_one_row_pair_per_shape_class(monkeypatch):
    """TL1279 -> O1280, 137 fields with the native rows on: a northern and a southern row of every (first radix, number of stages)
    class of native rows -- dense radix-17 .. 31 first stages, two to four stages, one and two rounds of butterflies per stage --
    and of every Bluestein class that remains, against the oracle"""
    monkeypatch.setenv("ATLAS_AMD_FFT_NATIVE", "1")
    g = atlas_amd.Grid("O1280")
    T, nf = 1279, 137
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, nf)
    gp = run_device(tr, nf, sp).reshape(nf, -1)
    rows, classes = rows_of_every_fft_class(tr)
    nat = [c for c in classes if c[2] == 4]
    assert {c[1] // 10 for c in nat} == {3, 5, 7, 9, 11, 13, 15, 17, 19, 23, 29, 31} and {c[1] % 10 for c in nat} >= {3, 4}
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for r, ref in zip(rows, op.invtrans_rows(nf, sp, rows, use_fft=True)):
        err = compute_rms(gp[:, off[r]:off[r + 1]], ref)
        assert err < 1e-12, (r, tuple(tr.fft_row_classes()[r]), err)




def test_two_jobs_in_sequence_rows_are_bitwise_equal_to_the_product_rows(monkeypatch):
    """[r5] tools/experiments/fft_ct_rows_seq.inc (ATLAS_AMD_FFT_SEQ=1): a workgroup transforms two fields of a row one after the
    other, the second job's modes requested into registers behind the first job's last phases -- same arithmetic, same bits;
    an odd number of field groups leaves the last workgroup of a row with one job"""
    g = atlas_amd.Grid("O1280")
    T, nf = 1279, 21                                   # three field groups: pairs (0, 1) and a single (2), the last group with 5 fields
    tr = atlas_amd.Trans(g, T)
    sp = red_spectra(T, nf, seed=97)
    ref = run_device(tr, nf, sp)
    monkeypatch.setenv("ATLAS_AMD_FFT_SEQ", "1")
    got = run_device(atlas_amd.Trans(g, T), nf, sp)
    assert np.array_equal(got, ref)
