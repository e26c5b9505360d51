"""Shared helpers of the parity tests (test infrastructure)."""
import math

import numpy as np

EARTH_RADIUS = 6371229.0  # atlas util::Earth::radius()


def red_spectra(T, nf, seed=20251114, trc=None):
    """synthetic spectra of SURVEY 8(d): N(0,1)*(1+n)^(-5/6), Im(m=0)=0; layout sp[(2*pos(m,n)+imag)*nf+fld]"""
    trc = T if trc is None else trc
    rng = np.random.default_rng(seed)
    sp = np.zeros(((trc + 1) * (trc + 2) // 2, 2, nf))
    k = 0
    for m in range(trc + 1):
        cnt = trc + 1 - m
        n = np.arange(m, trc + 1)
        sp[k:k + cnt] = rng.standard_normal((cnt, 2, nf)) * ((1.0 + n) ** (-5.0 / 6.0))[:, None, None]
        if m == 0:
            sp[k:k + cnt, 1, :] = 0.0
        k += cnt
    return sp.reshape(-1)


def rows_of_every_fft_class(tr, extra=()):
    """One northern and one southern latitude row of every Fourier kernel class the transform launches -- (method, transform
    length, kernel) as reported by the library (atlas_amd__Trans__fft_row_classes) -- plus `extra`.  Northern: the longest
    row of the class (most kept wavenumbers), southern: the shortest, so that two different lengths of a class are compared.
    The reference runs one code path for every row (TransLocal.cc:1155-1196); here every class is its own kernel instance."""
    cls = tr.fft_row_classes()
    ny = len(cls)
    north, south = {}, {}
    for j in range(ny):
        key = tuple(int(v) for v in cls[j])
        if j < ny // 2:
            north[key] = j          # the last northern row of a class is its longest
        else:
            south[key] = j          # the last southern row of a class is its shortest
    rows = sorted(set(north.values()) | set(south.values()) | set(int(r) for r in extra))
    return rows, sorted(set(north) | set(south))


def pos(T, m, n):
    return (2 * T + 3 - m) * m // 2 + (n - m)


def unit_spectrum(T, nf, n, m, imag, fld=0):
    sp = np.zeros((T + 1) * (T + 2) * nf)
    sp[(2 * pos(T, m, n) + imag) * nf + fld] = 1.0
    return sp


def compute_rms(a, b):
    """RMS(a-b) / max|b|  -- the metric of the reference tests (test_transgeneral.cc:472-489)"""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    mx = np.abs(b).max()
    if mx == 0.0:
        return 0.0
    return float(np.sqrt(np.mean((a - b) ** 2)) / mx)


def pbar(n, m, x):
    """normalised associated Legendre function of the IFS / Atlas convention: 1/2 * int_{-1}^{1} Pbar^2 dx = 1, no
    Condon-Shortley phase (so that Pbar_1^1 = +sqrt(3/2) cos(lat), test_transgeneral.cc:129-131).  Evaluated with
    mpmath-free closed recurrences in extended precision via scipy's lpmv would carry the CS phase; we use the
    explicit formula  Pbar_n^m = sqrt((2n+1) (n-m)!/(n+m)!) * (1-x^2)^{m/2} d^m/dx^m P_n(x)."""
    from numpy.polynomial import legendre as L
    x = np.asarray(x, dtype=np.longdouble)
    c = np.zeros(n + 1)
    c[n] = 1.0
    d = L.legder(c, m) if m > 0 else c
    val = L.legval(np.asarray(x, dtype=np.float64), d)
    norm = math.sqrt((2 * n + 1) * math.factorial(n - m) / math.factorial(n + m))
    return norm * (1.0 - np.asarray(x, dtype=np.float64) ** 2) ** (m / 2.0) * val


def analytic_scalar(grid_nx, grid_lat_deg, n, m, imag, row_mask=None):
    """expected grid-point field of a unit spectral coefficient (n,m,imag)
    (reference: sphericalharmonics_analytic_point + spectral_transform_grid_analytic,
     src/tests/trans/test_transgeneral.cc:80-131,381-460): Pbar_n^m(sin lat) * (m>0 ? 2 : 1) * (cos(m lon) | -sin(m lon)),
    rows with fourier truncation <= m are zero."""
    out = []
    for j, (nx, lat) in enumerate(zip(grid_nx, grid_lat_deg)):
        lon = np.arange(nx) * (360.0 / nx) * (math.pi / 180.0)
        rft = (2.0 if m > 0 else 1.0) * (np.cos(m * lon) if imag == 0 else -np.sin(m * lon))
        val = float(pbar(n, m, math.sin(lat * math.pi / 180.0))) * rft
        if row_mask is not None and not row_mask[j]:
            val = np.zeros(nx)
        out.append(val)
    return np.concatenate(out)


# closed forms quoted by the reference test for n <= 3 (test_transgeneral.cc:117-272) -- known answers
CLOSED_FORMS = {
    (0, 0): lambda s, c: 1.0 + 0 * s,
    (1, 0): lambda s, c: math.sqrt(3.0) * s,
    (2, 0): lambda s, c: math.sqrt(5.0) / 2.0 * (3.0 * s * s - 1.0),
    (3, 0): lambda s, c: math.sqrt(7.0) / 2.0 * (5.0 * s * s - 3.0) * s,
    (1, 1): lambda s, c: math.sqrt(3.0 / 2.0) * c,
    (2, 1): lambda s, c: math.sqrt(15.0 / 2.0) * s * c,
    (3, 1): lambda s, c: math.sqrt(21.0) / 4.0 * c * (5.0 * s * s - 1.0),
    (2, 2): lambda s, c: math.sqrt(15.0 / 2.0) / 2.0 * c * c,
    (3, 2): lambda s, c: math.sqrt(105.0 / 2.0) / 2.0 * c * c * s,
    (3, 3): lambda s, c: math.sqrt(35.0) / 4.0 * c * c * c,
}


def wind_kat(ivar_in, ivar_out, n, m, imag, lon, lat):
    """analytic u / v of a unit vorticity (ivar_in=0) or divergence (ivar_in=1) coefficient,
    known answers of src/tests/trans/test_transgeneral.cc:286-371 for (n,m) in {(0,0),(1,0),(1,1)}"""
    a = EARTH_RADIUS
    s, c = math.sin(lat), math.cos(lat)
    sl, cl = np.sin(m * lon), np.cos(m * lon)
    z = np.zeros_like(lon)
    if (n, m) == (0, 0):
        return z
    if ivar_in == 0:  # vorticity
        if ivar_out == 0:
            if (n, m) == (1, 0):
                return z + (math.sqrt(3.0) * a / 2.0 * c if imag == 0 else 0.0)
            if (n, m) == (1, 1):
                return -a * math.sqrt(1.5) * cl * s if imag == 0 else a * math.sqrt(1.5) * sl * s
        else:
            if (n, m) == (1, 0):
                return z
            if (n, m) == (1, 1):
                return a * math.sqrt(1.5) * sl if imag == 0 else a * math.sqrt(1.5) * cl
    else:  # divergence
        if ivar_out == 0:
            if (n, m) == (1, 0):
                return z
            if (n, m) == (1, 1):
                return a * math.sqrt(1.5) * sl if imag == 0 else a * math.sqrt(1.5) * cl
        else:
            if (n, m) == (1, 0):
                return z + (-math.sqrt(3.0) * a / 2.0 * c if imag == 0 else 0.0)
            if (n, m) == (1, 1):
                return a * math.sqrt(1.5) * cl * s if imag == 0 else -a * math.sqrt(1.5) * sl * s
    raise KeyError((ivar_in, ivar_out, n, m))
