"""TEST INFRASTRUCTURE ONLY -- the TransLocal inverse transform with library kernels on the CPU.

Same algorithm and tables as the plain restatement (oracle/translocal_oracle.c): per zonal wavenumber split the spectra by
parity, two dense products against the Legendre tables, merge the hemispheres (TransLocal.cc:939-1097), then one
unnormalised Hermitian c2r FFT per (field, latitude) (TransLocal.cc:1101-1196).  The two third-party kernels are the
library ones a production build of the reference uses:
  * GEMM: BLAS dgemm through numpy (the reference: eckit::linalg backend "lapack", MatrixMultiply_EckitLinalg.cc:64-67)
  * FFT : pocketfft through scipy.fft.irfft (the reference: ATLAS_LINALG_FFT_BACKEND=pocketfft, pocketfft.cc:32-60)
Used by bench.py as a second, tuned CPU baseline and validated against the plain oracle in tests/test_oracle_kat.py."""
import numpy as np
import scipy.fft


def invtrans_blas(plan, nf, sp, workers=1):
    """plan: oracle.OraclePlan (with tables); sp: spectra in the invtrans layout; returns gp[nf * npts]"""
    T, nlats = plan.T, plan.nlats
    if nlats % 2:
        raise ValueError("invtrans_blas: grids without an equator row only (Gaussian grids)")
    nleg = (nlats + 1) // 2
    nlat0 = [int(v) for v in plan.nlat0]
    sym, asym = plan.tables()
    sp = np.asarray(sp, dtype=np.float64)
    nh = nlats // 2                         # rows per hemisphere (an equator row, if any, belongs to the north)
    nnh = nlats - nh
    # Fourier intermediate, written per wavenumber as Fm[m][lat][fld] (contiguous slabs);
    # zero where a wavenumber is not kept (jlat < nlat0[m])
    Fm = np.zeros((T + 1, nlats, nf), dtype=np.complex128)
    nlegr = nleg                            # global grids: every Legendre row is used
    for m in range(T):                      # m == T is dropped by the scalar path (TransLocal.cc:982: jm < truncation)
        L = nlegr - int(nlat0[m])
        if L <= 0:
            continue
        ks, ka = (T + 1 - m + 2) // 2, (T + 1 - m + 1) // 2
        n_imag = 2 if m else 1
        ioff = (2 * T + 3 - m) * m // 2 * nf * 2
        blk = sp[ioff:ioff + 2 * (T - m + 1) * nf].reshape(T - m + 1, 2, nf)[:, :n_imag, :]   # [n-m][imag][fld]
        # n descending from T+1 (zero row) down to m, split by parity of n - m (TransLocal.cc:970-1003)
        full = np.zeros((T + 2 - m, n_imag * nf))
        full[1:] = blk[::-1].reshape(T - m + 1, n_imag * nf)          # index 0 <-> n = T+1
        par_top = (T + 1 - m) % 2                                       # parity of the first (n = T+1) entry
        # contiguous copies: numpy's matmul leaves BLAS for row-strided operands
        a_sym = np.ascontiguousarray(full[0::2] if par_top == 0 else full[1::2])
        a_asym = np.ascontiguousarray(full[1::2] if par_top == 0 else full[0::2])
        bs, ba = plan.begin(m)
        b_sym = sym[bs:bs + ks * nleg].reshape(nleg, ks)[int(nlat0[m]):nlegr]      # [lat][k]
        b_asym = asym[ba:ba + ka * nleg].reshape(nleg, ka)[int(nlat0[m]):nlegr]
        c_sym = b_sym @ a_sym[:ks]                                      # dgemm: [L][n_imag*nf]
        c_asym = b_asym @ a_asym[:ka]
        north = (c_sym + c_asym).reshape(L, n_imag, nf)
        south = (c_sym - c_asym).reshape(L, n_imag, nf)
        j0 = int(nlat0[m])
        if n_imag == 2:
            Fm[m, j0:nnh].real = north[:, 0]
            Fm[m, j0:nnh].imag = north[:, 1]
            Fm[m, nh:nlats - j0].real = south[::-1, 0]
            Fm[m, nh:nlats - j0].imag = south[::-1, 1]
        else:
            Fm[m, j0:nnh].real = north[:, 0]
            Fm[m, nh:nlats - j0].real = south[::-1, 0]
    # Fourier stage: per latitude all fields at once, transforming along the wavenumber axis of Fm (pocketfft gathers
    # the strided lines itself; zero-padding / truncation to n/2+1 modes is irfft's `n` argument)
    Fm[0].imag = 0.                                                     # in[0] = (re, 0)   (TransLocal.cc:1165-1170)
    out = np.empty((nf, plan.npts))
    off = 0
    for j in range(nlats):
        n = int(plan.nx[j])
        k = min(n // 2 + 1, T + 1)
        out[:, off:off + n] = scipy.fft.irfft(Fm[:k, j, :], n=n, axis=0, norm="forward", workers=workers).T
        off += n
    return out.reshape(-1)
