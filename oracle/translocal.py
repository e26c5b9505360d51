"""ctypes front-end of oracle/translocal_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("translocal_oracle.c",)]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
        L.orc_fourier_truncation.restype = i
        L.orc_fourier_truncation.argtypes = [i, i, i, i, d, i]
        L.orc_num_n.restype = sz
        L.orc_num_n.argtypes = [i, i, i]
        L.orc_compute_zfn.argtypes = [i, vp]
        L.orc_legendre_lat.argtypes = [i, d, vp, vp, vp, vp]
        L.orc_plan_create.restype = vp
        L.orc_plan_create.argtypes = [i, i, vp, vp, i, i]
        L.orc_plan_destroy.argtypes = [vp]
        L.orc_plan_nlat0.restype = i
        L.orc_plan_nlat0.argtypes = [vp, i]
        L.orc_plan_nlats_leg.restype = i
        L.orc_plan_nlats_leg.argtypes = [vp]
        L.orc_plan_nlats_legr.restype = i
        L.orc_plan_nlats_legr.argtypes = [vp]
        L.orc_plan_npts.restype = C.c_int64
        L.orc_plan_npts.argtypes = [vp]
        for f in ("orc_plan_size_sym", "orc_plan_size_asym"):
            getattr(L, f).restype = sz
            getattr(L, f).argtypes = [vp]
        for f in ("orc_plan_leg_sym", "orc_plan_leg_asym"):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = [vp]
        for f in ("orc_plan_begin_sym", "orc_plan_begin_asym"):
            getattr(L, f).restype = sz
            getattr(L, f).argtypes = [vp, i]
        L.orc_invtrans.argtypes = [vp, i, vp, vp, i]
        L.orc_invtrans_uv.argtypes = [vp, i, i, i, vp, vp, i]
        L.orc_invtrans_legendre_only.argtypes = [vp, i, i, vp, vp]
        L.orc_invtrans_fourier.argtypes = [vp, i, vp, vp, i]
        L.orc_invtrans_rows.argtypes = [vp, i, i, vp, i, vp, vp, i]
        L.orc_invtrans_vordiv.argtypes = [vp, i, vp, i, vp, vp, vp, i]
        L.orc_invtrans_vordiv_rows.argtypes = [vp, i, vp, i, vp, vp, i, vp, vp, i]
        L.orc_vd2uv.argtypes = [i, i, vp, vp, vp, vp]
        L.orc_gemm.argtypes = [i, sz, i, vp, vp, vp]
        L.orc_c2r_direct.argtypes = [i, vp, vp]
        L.orc_c2r_fft.argtypes = [i, vp, vp]
        _lib = L
    return _lib


def fourier_truncation(truncation, nx, nxmax, ndgl, lat_rad, fullgrid):
    return lib().orc_fourier_truncation(truncation, nx, nxmax, ndgl, float(lat_rad), int(bool(fullgrid)))


def legendre_lat(trc, lat_rad):
    """packed triangle legpol[idx(m,n)], idx = (2*trc+3-m)*m/2 + n-m  (LegendrePolynomials.cc:47-151)"""
    L = lib()
    zfn = np.zeros((trc + 1) * (trc + 1))
    L.orc_compute_zfn(trc, zfn.ctypes.data)
    legpol = np.zeros((trc + 2) * (trc + 1) // 2)
    vs, vc = np.zeros(trc + 1), np.zeros(trc + 1)
    L.orc_legendre_lat(trc, float(lat_rad), legpol.ctypes.data, zfn.ctypes.data, vs.ctypes.data, vc.ctypes.data)
    return legpol


def gemm(A, B):
    """C = A @ B through the oracle's own GEMM (column-major operands, the contraction of the Legendre stage)"""
    A = np.asfortranarray(A, dtype=np.float64)
    B = np.asfortranarray(B, dtype=np.float64)
    rows, K = A.shape
    K2, L = B.shape
    assert K == K2
    C_ = np.zeros((rows, L), order="F")
    lib().orc_gemm(rows, K, L, A.ctypes.data, B.ctypes.data, C_.ctypes.data)
    return np.ascontiguousarray(C_)


def c2r_direct(n, half_spectrum):
    x = np.ascontiguousarray(half_spectrum, dtype=np.complex128)
    assert x.size == n // 2 + 1
    out = np.zeros(n)
    lib().orc_c2r_direct(n, x.ctypes.data, out.ctypes.data)
    return out


def c2r_fft(n, half_spectrum):
    x = np.ascontiguousarray(half_spectrum, dtype=np.complex128)
    assert x.size == n // 2 + 1
    out = np.zeros(n)
    lib().orc_c2r_fft(n, x.ctypes.data, out.ctypes.data)
    return out


class OraclePlan:
    """TransLocal constructor geometry + tables for a global structured grid (oracle side)."""

    def __init__(self, truncation, nx, lat_deg, regular=None, with_tables=True):
        self.T = int(truncation)
        self.nx = np.ascontiguousarray(nx, dtype=np.int32)
        self.lat = np.ascontiguousarray(lat_deg, dtype=np.float64)
        self.nlats = len(self.nx)
        self.regular = bool(np.all(self.nx == self.nx[0])) if regular is None else bool(regular)
        self._h = lib().orc_plan_create(self.T, self.nlats, self.nx.ctypes.data, self.lat.ctypes.data,
                                        int(self.regular), int(with_tables))
        self.npts = int(lib().orc_plan_npts(self._h))
        self.with_tables = with_tables

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_plan_destroy(self._h)
            self._h = None

    @property
    def nlat0(self):
        return np.array([lib().orc_plan_nlat0(self._h, m) for m in range(self.T + 1)], dtype=np.int32)

    def nspec(self, nf, trc=None):
        trc = self.T if trc is None else trc
        return (trc + 1) * (trc + 2) * nf

    def tables(self):
        L = lib()
        ns, na = L.orc_plan_size_sym(self._h), L.orc_plan_size_asym(self._h)
        sym = np.ctypeslib.as_array(C.cast(L.orc_plan_leg_sym(self._h), C.POINTER(C.c_double)), shape=(ns,))
        asym = np.ctypeslib.as_array(C.cast(L.orc_plan_leg_asym(self._h), C.POINTER(C.c_double)), shape=(na,))
        return sym, asym

    def begin(self, m):
        return lib().orc_plan_begin_sym(self._h, m), lib().orc_plan_begin_asym(self._h, m)

    def invtrans(self, nf, sp, use_fft=False):
        sp = np.ascontiguousarray(sp, dtype=np.float64)
        assert sp.size == self.nspec(nf)
        gp = np.zeros(nf * self.npts)
        lib().orc_invtrans(self._h, nf, sp.ctypes.data, gp.ctypes.data, int(use_fft))
        return gp

    def invtrans_uv(self, trc, nf, nb_vordiv, sp, use_fft=False):
        sp = np.ascontiguousarray(sp, dtype=np.float64)
        assert sp.size == self.nspec(nf, trc)
        gp = np.zeros(nf * self.npts)
        lib().orc_invtrans_uv(self._h, trc, nf, nb_vordiv, sp.ctypes.data, gp.ctypes.data, int(use_fft))
        return gp

    def invtrans_vordiv(self, ns, sp, nvd, vor, div, use_fft=False):
        """TransLocal::invtrans(ns, sp, nvd, vor, div, gp): gp = [u fields][v fields][scalar fields]"""
        vor = np.ascontiguousarray(vor, dtype=np.float64)
        div = np.ascontiguousarray(div, dtype=np.float64)
        sp_p = np.ascontiguousarray(sp, dtype=np.float64).ctypes.data if ns > 0 else None
        gp = np.zeros((ns + 2 * nvd) * self.npts)
        lib().orc_invtrans_vordiv(self._h, ns, sp_p, nvd, vor.ctypes.data, div.ctypes.data, gp.ctypes.data,
                                  int(use_fft))
        return gp

    def legendre(self, nf, sp, trc=None):
        """Fourier intermediate in the reference layout [fld][lat][m][re,im] (TransLocal.h:177-180)"""
        trc = self.T if trc is None else trc
        sp = np.ascontiguousarray(sp, dtype=np.float64)
        out = np.zeros(nf * 2 * self.nlats * (self.T + 1))
        lib().orc_invtrans_legendre_only(self._h, trc, nf, sp.ctypes.data, out.ctypes.data)
        return out.reshape(nf, self.nlats, self.T + 1, 2)

    def fourier(self, nf, scl_fourier, use_fft=False):
        f = np.ascontiguousarray(scl_fourier, dtype=np.float64)
        gp = np.zeros(nf * self.npts)
        lib().orc_invtrans_fourier(self._h, nf, f.ctypes.data, gp.ctypes.data, int(use_fft))
        return gp

    def invtrans_rows(self, nf, sp, rows, trc=None, use_fft=False):
        """grid-point rows `rows` (all nf fields each) without building tables: list of arrays [nf][nx(row)]"""
        trc = self.T if trc is None else trc
        sp = np.ascontiguousarray(sp, dtype=np.float64)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        total = int(sum(nf * int(self.nx[r]) for r in rows))
        out = np.zeros(total)
        lib().orc_invtrans_rows(self._h, trc, nf, sp.ctypes.data, len(rows), rows.ctypes.data, out.ctypes.data,
                                int(use_fft))
        res, off = [], 0
        for r in rows:
            n = int(self.nx[r])
            res.append(out[off:off + nf * n].reshape(nf, n))
            off += nf * n
        return res


    def invtrans_vordiv_rows(self, ns, sp, nvd, vor, div, rows, use_fft=False):
        """rows `rows` of TransLocal::invtrans(ns, sp, nvd, vor, div, gp) without tables (full-size parity of the vor/div
        path): list of arrays [2 nvd + ns][nx(row)], fields ordered [u..][v..][scalars..] (TransLocal.cc:1523-1597,1443-1469)"""
        OraclePlan_rows_check(self, rows)
        vor = np.ascontiguousarray(vor, dtype=np.float64)
        div = np.ascontiguousarray(div, dtype=np.float64)
        assert vor.size == self.nspec(nvd) and div.size == self.nspec(nvd)
        sp_c = np.ascontiguousarray(sp, dtype=np.float64) if ns > 0 else None
        assert ns == 0 or sp_c.size == self.nspec(ns)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        nall = 2 * nvd + ns
        out = np.zeros(int(sum(nall * int(self.nx[r]) for r in rows)))
        lib().orc_invtrans_vordiv_rows(self._h, ns, sp_c.ctypes.data if ns > 0 else None, nvd, vor.ctypes.data,
                                       div.ctypes.data, len(rows), rows.ctypes.data, out.ctypes.data, int(use_fft))
        res, off = [], 0
        for r in rows:
            n = int(self.nx[r])
            res.append(out[off:off + nall * n].reshape(nall, n))
            off += nall * n
        return res


def OraclePlan_rows_check(plan, rows):
    rows = np.asarray(rows)
    assert rows.size > 0 and rows.min() >= 0 and rows.max() < plan.nlats


def vd2uv(trc, nf, vor, div):
    """VorDivToUVLocal.cc:62-184: (U, V) spectra, layout and truncation of the inputs"""
    vor = np.ascontiguousarray(vor, dtype=np.float64)
    div = np.ascontiguousarray(div, dtype=np.float64)
    U = np.zeros_like(vor)
    V = np.zeros_like(vor)
    lib().orc_vd2uv(int(trc), int(nf), vor.ctypes.data, div.ctypes.data, U.ctypes.data, V.ctypes.data)
    return U, V


def invtrans_regional(truncation, lats_deg, lons_deg, nf, sp, trc_in=None, nb_vordiv=0):
    """TransLocal's branch for a regular target that is not a crop of a global grid (no_nest), restated with numpy on top of
    the oracle's Legendre recurrence: Legendre polynomials of truncation + 1 at the target's own latitudes, clamped to
    +-89.9999999 (TransLocal.cc:535-557); every wavenumber kept at every latitude (nlat0 = 0, :463-468) except m >= trc_in,
    which the Legendre stage never transforms (jm < truncation, :982); Fourier stage = matrix cos(m lon) * factor,
    -sin(m lon) * factor with factor 2 for m > 0 (:719-738) times the Fourier coefficients (:1139-1148); the first
    2 nb_vordiv fields (u, v) are divided by cos(lat) (:1443-1469).
    sp: [(n, m) position of truncation trc_in][re, im][field]; returns gp[field][lat][lon]."""
    T = int(truncation)
    trc = T if trc_in is None else int(trc_in)
    TL = T + 1
    sp = np.asarray(sp, dtype=np.float64).reshape((trc + 1) * (trc + 2) // 2, 2, nf)
    lats = np.clip(np.asarray(lats_deg, dtype=np.float64), -89.9999999, 89.9999999) * (np.pi / 180.0)
    lons = np.asarray(lons_deg, dtype=np.float64) * (np.pi / 180.0)
    nmax = min(trc, TL)
    gp = np.zeros((nf, len(lats), len(lons)))
    for j, lat in enumerate(lats):
        leg = legendre_lat(TL, lat)
        for m in range(min(trc, T + 1)):         # jm < truncation of the input, and at most T
            bl = (2 * TL + 3 - m) * m // 2       # table position of (m, m)
            bs = (2 * trc + 3 - m) * m // 2      # spectra position of (m, m)
            cnt = nmax - m + 1
            P = leg[bl:bl + cnt]                 # n = m .. nmax
            re = P @ sp[bs:bs + cnt, 0, :]
            im = P @ sp[bs:bs + cnt, 1, :]
            factor = 2.0 if m > 0 else 1.0
            c, s_ = np.cos(m * lons) * factor, -np.sin(m * lons) * factor
            gp[:, j, :] += re[:, None] * c[None, :] + (im[:, None] * s_[None, :] if m > 0 else 0.0)
        if nb_vordiv > 0:
            gp[:2 * nb_vordiv, j, :] /= np.cos(lat)
    return gp


def invtrans_regional_vordiv(truncation, lats_deg, lons_deg, ns, sp, nvd, vor, div):
    """TransLocal::invtrans(ns, sp, nvd, vor, div, gp) for such a target: extend_truncation to T + 1 (TransLocal.cc:1496-1519),
    vd2uv there (VorDivToUVLocal.cc:62-184), fields interleaved [U..][V..][scalars..] per coefficient (:1523-1597), then the
    scalar path with the 1 / cos(lat) scaling of u and v.  Returns gp[2 nvd + ns][lat][lon]."""
    T = int(truncation)

    def extend(a, nf):
        a = np.asarray(a, dtype=np.float64).reshape((T + 1) * (T + 2) // 2, 2, nf)
        out = np.zeros(((T + 2) * (T + 3) // 2, 2, nf))
        for m in range(T + 1):
            b0, b1 = (2 * T + 3 - m) * m // 2, (2 * (T + 1) + 3 - m) * m // 2
            out[b1:b1 + T - m + 1] = a[b0:b0 + T - m + 1]
        return out

    ve, de = extend(vor, nvd), extend(div, nvd)
    U, V = vd2uv(T + 1, nvd, ve.ravel(), de.ravel())
    parts = [U.reshape(-1, 2, nvd), V.reshape(-1, 2, nvd)]
    if ns > 0:
        parts.append(extend(sp, ns))
    allsp = np.concatenate(parts, axis=2)
    return invtrans_regional(T, lats_deg, lons_deg, 2 * nvd + ns, allsp, trc_in=T + 1, nb_vordiv=nvd)


def invtrans_unstructured(truncation, lons_deg, lats_deg, nf, sp, trc_in=None, nb_vordiv=0):
    """TransLocal's unstructured path (TransLocal.cc:1200-1291): per point the Legendre polynomials at its latitude and
    the Fourier sum 1, 2 cos(m lon), -2 sin(m lon) over jm < truncation; u and v divided by cos(lat) of the point.
    Returns gp[field][point].  (Restated on the regional formulation above, one point at a time.)"""
    lons, lats = np.asarray(lons_deg, dtype=np.float64), np.asarray(lats_deg, dtype=np.float64)
    out = np.zeros((nf, len(lons)))
    for i in range(len(lons)):
        out[:, i] = invtrans_regional(truncation, [lats[i]], [lons[i]], nf, sp, trc_in=trc_in, nb_vordiv=nb_vordiv)[:, 0, 0]
    return out
