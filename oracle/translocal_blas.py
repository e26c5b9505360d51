"""TEST INFRASTRUCTURE ONLY -- the TransLocal inverse transform with library kernels on the CPU.

Same algorithm and tables as the plain restatement (oracle/translocal_oracle.c): per zonal wavenumber split the spectra by
parity, two dense products against the Legendre tables, merge the hemispheres (TransLocal.cc:939-1097), then one
unnormalised Hermitian c2r FFT per (field, latitude) (TransLocal.cc:1101-1196).  The two third-party kernels are the
library ones a production build of the reference uses:
  * GEMM: BLAS dgemm through numpy (the reference: eckit::linalg backend "lapack", MatrixMultiply_EckitLinalg.cc:64-67)
  * FFT : pocketfft through scipy.fft.irfft (the reference: ATLAS_LINALG_FFT_BACKEND=pocketfft, pocketfft.cc:32-60)
Used by bench.py as the CPU baseline and validated against the plain oracle in tests/test_oracle_kat.py.

Round 6: the Python glue no longer serialises the box.  Both stages run on a pool of `threads` Python threads
(numpy copies, BLAS and pocketfft release the interpreter lock): the Legendre stage over wavenumbers, heaviest first, each thread
calling single-threaded dgemm on its own m (a 64-thread dgemm per m from one thread spent its time in fork / join around 20 ms
products); the Fourier stage over groups of rows of one length, every group one batched irfft over (rows x fields) lines.  The
intermediate is [lat][m][field] so that a row's modes are one contiguous block.  `timings` (a dict, optional) receives the
wall-clock split legendre_s / fourier_s and layout_s = thread-time spent in the split / merge / transpose copies divided by the
thread count (the part of the wall clock that is neither GEMM nor FFT)."""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.fft

try:
    from threadpoolctl import threadpool_limits
except Exception:   # pragma: no cover
    threadpool_limits = None


def invtrans_blas(plan, nf, sp, workers=1, timings=None, threads=None):
    """plan: oracle.OraclePlan (with tables); sp: spectra in the invtrans layout; returns gp[nf * npts].
    workers: cores to use (historical name: it used to be pocketfft's `workers`); threads: Python threads (default = workers)"""
    T, nlats = plan.T, plan.nlats
    if nlats % 2:
        raise ValueError("invtrans_blas: grids without an equator row only (Gaussian grids)")
    threads = max(1, int(threads if threads is not None else workers))
    nleg = (nlats + 1) // 2
    nlat0 = [int(v) for v in plan.nlat0]
    sym, asym = plan.tables()
    sp = np.asarray(sp, dtype=np.float64)
    nh = nlats // 2                         # rows per hemisphere (an equator row, if any, belongs to the north)
    nnh = nlats - nh
    nx = np.asarray(plan.nx, dtype=np.int64)
    # Fourier intermediate F[lat][m][fld] (a row's modes contiguous); only the modes a row's FFT reads are allocated:
    # k <= min(n/2, T); zero where a wavenumber is not kept (jlat < nlat0[m])
    kmax = int(min(int(nx.max()) // 2 + 1, T + 1))
    F = np.zeros((nlats, kmax, nf), dtype=np.complex128)
    nlegr = nleg                            # global grids: every Legendre row is used

    def legendre_m(args):
        m = args
        L = nlegr - int(nlat0[m])
        if L <= 0 or m >= kmax:
            return 0.0
        ks, ka = (T + 1 - m + 2) // 2, (T + 1 - m + 1) // 2
        n_imag = 2 if m else 1
        ioff = (2 * T + 3 - m) * m // 2 * nf * 2
        t0 = time.perf_counter()
        blk = sp[ioff:ioff + 2 * (T - m + 1) * nf].reshape(T - m + 1, 2, nf)[:, :n_imag, :]   # [n-m][imag][fld]
        # n descending from T+1 (zero row) down to m, split by parity of n - m (TransLocal.cc:970-1003)
        full = np.zeros((T + 2 - m, n_imag * nf))
        full[1:] = blk[::-1].reshape(T - m + 1, n_imag * nf)          # index 0 <-> n = T+1
        par_top = (T + 1 - m) % 2                                       # parity of the first (n = T+1) entry
        # contiguous copies: numpy's matmul leaves BLAS for row-strided operands
        a_sym = np.ascontiguousarray(full[0::2] if par_top == 0 else full[1::2])
        a_asym = np.ascontiguousarray(full[1::2] if par_top == 0 else full[0::2])
        t1 = time.perf_counter()
        bs, ba = plan.begin(m)
        b_sym = sym[bs:bs + ks * nleg].reshape(nleg, ks)[int(nlat0[m]):nlegr]      # [lat][k]
        b_asym = asym[ba:ba + ka * nleg].reshape(nleg, ka)[int(nlat0[m]):nlegr]
        c_sym = b_sym @ a_sym[:ks]                                      # dgemm: [L][n_imag*nf]
        c_asym = b_asym @ a_asym[:ka]
        t2 = time.perf_counter()
        j0 = int(nlat0[m])
        # merge the hemispheres (TransLocal.cc:1031-1080) into complex rows, then one block write per hemisphere
        tmp = np.zeros((L, nf), dtype=np.complex128)                    # in[0] = (re, 0) for m = 0   (TransLocal.cc:1165-1170)
        cs, ca = c_sym.reshape(L, n_imag, nf), c_asym.reshape(L, n_imag, nf)
        np.add(cs[:, 0], ca[:, 0], out=tmp.real)
        if n_imag == 2:
            np.add(cs[:, 1], ca[:, 1], out=tmp.imag)
        F[j0:nnh, m] = tmp
        np.subtract(cs[:, 0], ca[:, 0], out=tmp.real)
        if n_imag == 2:
            np.subtract(cs[:, 1], ca[:, 1], out=tmp.imag)
        F[nh:nlats - j0, m] = tmp[::-1]
        return (t1 - t0) + (time.perf_counter() - t2)

    out = np.empty((nf, plan.npts))
    offs = np.concatenate([[0], np.cumsum(nx)])
    # rows of one length together (the two hemispheres of a reduced grid; all rows of a regular one), in slices that keep
    # a batch near 64 MB; longest rows first
    groups = []
    for n in sorted(set(int(v) for v in nx), reverse=True):
        rows = np.nonzero(nx == n)[0]
        per = max(1, int((64 << 20) // max(1, n * nf * 8)))
        for i in range(0, len(rows), per):
            groups.append((n, rows[i:i + per]))

    def fourier_group(args):
        n, rows = args
        k = min(n // 2 + 1, T + 1)
        t0 = time.perf_counter()
        X = F[rows, :k, :]                                              # [rows][k][fld] (a copy: fancy index)
        t1 = time.perf_counter()
        # lines along the wavenumber axis, gathered by pocketfft itself; the result comes out [rows][fld][n], the layout of gp
        Y = scipy.fft.irfft(X.transpose(0, 2, 1), n=n, axis=2, norm="forward", workers=1 if threads > 1 else workers)
        t2 = time.perf_counter()
        for i, j in enumerate(rows):
            out[:, offs[j]:offs[j] + n] = Y[i]
        return (t1 - t0) + (time.perf_counter() - t2)

    def run(fn, work):   # returns the thread-time spent in layout copies
        if threads == 1:
            return sum(fn(w) for w in work)
        with ThreadPoolExecutor(threads) as ex:
            return sum(ex.map(fn, work))

    import contextlib
    limit = threadpool_limits(limits=1, user_api="blas") if (threadpool_limits is not None and threads > 1) else contextlib.nullcontext()
    with limit:
        t0 = time.perf_counter()
        copy_leg = run(legendre_m, list(range(T)))   # m == T is dropped by the scalar path (TransLocal.cc:982: jm < truncation); heaviest (m = 0) first
        t1 = time.perf_counter()
        copy_fft = run(fourier_group, groups)
        t2 = time.perf_counter()
    if timings is not None:
        timings.update({"legendre_s": t1 - t0, "fourier_s": t2 - t1, "layout_s": (copy_leg + copy_fft) / threads,
                        "layout_in_legendre_s": copy_leg / threads, "layout_in_fourier_s": copy_fft / threads, "threads": threads})
    return out.reshape(-1)
