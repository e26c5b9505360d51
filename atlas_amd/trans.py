"""Host-side mirror of atlas::trans::Trans for type "local"
(reference: src/atlas/trans/Trans.h:42-189, src/atlas/trans/local/TransLocal.cc:818-934,1409-1597).

All arithmetic happens in the HIP library behind the C ABI (include/atlas_amd.h); this class only checks shapes
and forwards pointers.  NumPy arrays go through the host-pointer entry points, torch CUDA tensors (or anything with
`data_ptr()` and `is_cuda`) through the device-pointer entry points, asynchronously on the Trans stream."""
import ctypes as C

import numpy as np

from . import _lib
from .grid import StructuredGrid


def _is_device(a):
    return hasattr(a, "data_ptr") and bool(getattr(a, "is_cuda", False))


def _ptr(a, n=None, name="array", writable=False):
    """pointer of a contiguous float64 array/tensor with at least n elements"""
    if a is None:
        return None
    if _is_device(a):
        import torch
        if a.dtype != torch.float64 or not a.is_contiguous():
            raise TypeError(f"{name}: need a contiguous float64 tensor")
        if n is not None and a.numel() < n:
            raise ValueError(f"{name}: {a.numel()} elements, need {n}")
        return a.data_ptr()
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
        raise TypeError(f"{name}: need a C-contiguous float64 numpy array")
    if writable and not a.flags.writeable:
        raise TypeError(f"{name}: not writable")
    if n is not None and a.size < n:
        raise ValueError(f"{name}: {a.size} elements, need {n}")
    return a.ctypes.data


class Trans:
    """trans::Trans(grid, truncation, config) with option::type("local") semantics, running on an MI355X."""

    def __init__(self, grid, truncation, profile=False, nparts=1, part=0, legendre_cache=None, shard="m", rows=None,
                 tables=None, domain=None, **options):
        if isinstance(grid, str):
            grid = StructuredGrid(name=grid)
        self.grid = grid
        # shard (nparts > 1): "m" = this device owns wavenumbers m % nparts == part (stage API + exchange),
        #                      "band" = it owns latitude band `part` of both stages (plain invtrans on device arrays),
        #                      "mirror" = it owns a northern band of rows and its mirror image (plain invtrans; output =
        #                      the northern rows, then the southern rows; see owned_rows())
        cfg = f"profile={int(bool(profile))};nparts={int(nparts)};part={int(part)};shard={shard}"
        if tables is not None:  # "host" | "device": where the Legendre table is computed (default: ATLAS_AMD_TABLES)
            cfg += f";tables={tables}"
        if rows is not None:   # (j0, j1): zonal-band crop of the grid (regional domain that keeps whole latitude rows)
            cfg += f";rows={int(rows[0])}:{int(rows[1])}"
        if domain is not None:  # (west, east, south, north): trans::Trans(grid, RectangularDomain, truncation)
            if rows is not None:
                raise ValueError("give rows= or domain=, not both")
            cfg += ";domain=" + ",".join(repr(float(v)) for v in domain)
        for k, v in options.items():   # atlas option:: keys (fft="FFTW", matrix_multiply=..., warning=0, ...): passed through
            v = int(v) if isinstance(v, bool) else v
            if ";" in str(k) or "=" in str(k) or ";" in str(v):   # the config string is ';'-separated key=value items
                raise ValueError(f"option {k!r}={v!r}: ';' cannot be passed through the configuration string")
            cfg += f";{k}={v}"
        cache_ptr, cache_size = None, 0
        if legendre_cache is not None:
            self._cache = np.ascontiguousarray(np.frombuffer(legendre_cache, dtype=np.uint8))
            cache_ptr, cache_size = self._cache.ctypes.data, self._cache.size
        self._h = _lib.check_ptr(_lib.Trans_new_config(grid._h, int(truncation), cfg.encode(), cache_ptr, cache_size))
        self.notes = _lib.last_note().decode("utf-8", "replace")   # e.g. the TransLocal option keys that were accepted and ignored
        self.nparts, self.part, self.shard = int(nparts), int(part), shard
        self.rows = rows
        self.domain = domain
        if domain is not None:
            self.rows, self.window_first, self.window_count = grid.crop_to_domain(*domain)

    def __del__(self):
        h = getattr(self, "_h", None)
        delete = getattr(_lib, "Trans_delete", None) if _lib is not None else None   # module globals go first at interpreter exit
        if h and delete is not None:
            delete(h)
            self._h = None

    # ---- atlas::trans::TransImpl accessors (TransImpl.h:38-60) ----
    def truncation(self):
        return _lib.Trans_truncation(self._h)

    def nb_spectral_coefficients(self):
        return _lib.Trans_nb_spectral_coefficients(self._h)

    def nb_spectral_coefficients_global(self):
        return _lib.Trans_nb_spectral_coefficients(self._h)

    def nb_gridpoints(self):
        return _lib.Trans_nb_gridpoints(self._h)

    def nb_gridpoints_global(self):
        return _lib.Trans_nb_gridpoints_global(self._h)

    # ---- inverse transforms ----
    def invtrans(self, nb_scalar_fields, scalar_spectra, *args):
        """invtrans(nb_scalar, sp, gp)                                    TransLocal.cc:931-934
           invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp)               TransLocal.cc:1523-1597"""
        ncoef = self.nb_spectral_coefficients()
        npts = self.nb_gridpoints() if (self.nparts > 1 or self.rows is not None) else self.nb_gridpoints_global()
        if len(args) == 1:
            (gp,) = args
            nf = int(nb_scalar_fields)
            dev = _is_device(gp)
            if dev and _is_device(scalar_spectra):
                import torch
                if gp.dtype == torch.float32 and scalar_spectra.dtype == torch.float32:
                    # fp32 variant (an extension; BASELINE config C5): float device arrays in the same layouts
                    if not (gp.is_contiguous() and scalar_spectra.is_contiguous()):
                        raise TypeError("need contiguous float32 tensors")
                    if scalar_spectra.numel() < ncoef * nf or gp.numel() < npts * nf:
                        raise ValueError("float32 arrays too small")
                    with _lib.torch_stream_order(self.stream()):
                        _lib.check(_lib.Trans_invtrans_scalar_device_f32(self._h, nf, scalar_spectra.data_ptr(),
                                                                         gp.data_ptr()))
                    return gp
            if (not dev and isinstance(gp, np.ndarray) and isinstance(scalar_spectra, np.ndarray)
                    and gp.dtype == np.float32 and scalar_spectra.dtype == np.float32):
                if not (gp.flags.c_contiguous and scalar_spectra.flags.c_contiguous and gp.flags.writeable):
                    raise TypeError("need C-contiguous float32 arrays")
                if scalar_spectra.size < ncoef * nf or gp.size < npts * nf:
                    raise ValueError("float32 arrays too small")
                _lib.check(_lib.Trans_invtrans_scalar_f32(self._h, nf, scalar_spectra.ctypes.data, gp.ctypes.data))
                return gp
            if dev != _is_device(scalar_spectra):
                raise TypeError("spectra and grid-point arrays must both be host or both be device")
            sp_p = _ptr(scalar_spectra, ncoef * nf, "scalar_spectra")
            gp_p = _ptr(gp, npts * nf, "gp_fields", writable=True)
            if dev:
                with _lib.torch_stream_order(self.stream()):   # torch tensors: stream-ordered with their producers
                    _lib.check(_lib.Trans_invtrans_scalar_device(self._h, nf, sp_p, gp_p))
            else:
                _lib.check(_lib.Trans_invtrans_scalar(self._h, nf, sp_p, gp_p))
            return gp
        if len(args) == 4:
            nvd, vor, div, gp = args
            ns, nvd = int(nb_scalar_fields), int(nvd)
            dev = _is_device(gp)
            if dev:
                import torch
                if gp.dtype == torch.float32:   # fp32 variant (an extension): all arrays float32 on the device
                    arrs = [a for a in (scalar_spectra if ns > 0 else None, vor, div) if a is not None]
                    if not all(_is_device(a) and a.dtype == torch.float32 and a.is_contiguous() for a in arrs) or not gp.is_contiguous():
                        raise TypeError("fp32 vor/div call: contiguous float32 device tensors throughout")
                    if (ns > 0 and scalar_spectra.numel() < ncoef * ns) or vor.numel() < ncoef * nvd or div.numel() < ncoef * nvd \
                            or gp.numel() < npts * (ns + 2 * nvd):
                        raise ValueError("float32 arrays too small")
                    with _lib.torch_stream_order(self.stream()):
                        _lib.check(_lib.Trans_invtrans_device_f32(self._h, ns, scalar_spectra.data_ptr() if ns > 0 else None, nvd,
                                                                   vor.data_ptr(), div.data_ptr(), gp.data_ptr()))
                    return gp
            sp_p = _ptr(scalar_spectra, ncoef * ns, "scalar_spectra") if ns > 0 else None
            vor_p = _ptr(vor, ncoef * nvd, "vorticity_spectra") if nvd > 0 else None
            div_p = _ptr(div, ncoef * nvd, "divergence_spectra") if nvd > 0 else None
            gp_p = _ptr(gp, npts * (ns + 2 * nvd), "gp_fields", writable=True)
            if dev:
                with _lib.torch_stream_order(self.stream()):
                    _lib.check(_lib.Trans_invtrans_device(self._h, ns, sp_p, nvd, vor_p, div_p, gp_p))
            else:
                _lib.check(_lib.Trans_invtrans(self._h, ns, sp_p, nvd, vor_p, div_p, gp_p))
            return gp
        raise TypeError("invtrans(nb_scalar, sp, gp) or invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp)")

    # ---- Field / FieldSet overloads (TransLocal.cc:818-897): host numpy arrays stand in for atlas::Field ----
    @staticmethod
    def _field(a, name, writable=False):
        if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
            raise TypeError(f"{name}: C-contiguous float64 numpy array expected")
        if writable and not a.flags.writeable:
            raise TypeError(f"{name}: array is read-only")
        f = _lib.Field()
        f.data, f.rank = a.ctypes.data, a.ndim
        for i in range(min(a.ndim, 2)):
            f.shape[i] = a.shape[i]
        return f

    def invtrans_field(self, spfield, gpfield):
        """invtrans(const Field& spfield, Field& gpfield)                        TransLocal.cc:818-834 (rank-1 only)"""
        _lib.check(_lib.Trans_invtrans_field(self._h, self._field(spfield, "spfield"),
                                             self._field(gpfield, "gpfield", True)))
        return gpfield

    def invtrans_fieldset(self, spfields, gpfields):
        """invtrans(const FieldSet&, FieldSet&)                                  TransLocal.cc:838-844"""
        sp = (_lib.Field * max(len(spfields), 1))(*[self._field(a, "spfields[]") for a in spfields])
        gp = (_lib.Field * max(len(gpfields), 1))(*[self._field(a, "gpfields[]", True) for a in gpfields])
        _lib.check(_lib.Trans_invtrans_fieldset(self._h, sp, len(spfields), gp, len(gpfields)))
        return gpfields

    def invtrans_vordiv2wind_field(self, spvor, spdiv, gpwind):
        """invtrans_vordiv2wind(const Field& spvor, const Field& spdiv, Field& gpwind)   TransLocal.cc:871-897;
        gpwind of shape (2, npts) or (npts, 2)"""
        _lib.check(_lib.Trans_invtrans_vordiv2wind_field(self._h, self._field(spvor, "spvor"),
                                                         self._field(spdiv, "spdiv"),
                                                         self._field(gpwind, "gpwind", True)))
        return gpwind

    def invtrans_grad_field(self, spfield, gradfield):
        _lib.check(_lib.Trans_invtrans_grad_field(self._h, None, None))

    def invtrans_adj_field(self, gpfield, spfield):
        _lib.check(_lib.Trans_invtrans_adj_field(self._h, None, None))

    def dirtrans_field(self, gpfield, spfield):
        _lib.check(_lib.Trans_dirtrans_field(self._h, None, None))

    # ---- backend registry (Trans.cc:37-48) ----
    @staticmethod
    def hasBackend(name):
        return bool(_lib.Trans_has_backend(name.encode()))

    @staticmethod
    def backend(name=None):
        """Trans.backend() -> current backend name; Trans.backend(name) selects it (asserts hasBackend)"""
        import ctypes as C
        if name is not None:
            _lib.check(_lib.Trans_set_backend(name.encode()))
            return name
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.Trans_backend(C.byref(p), C.byref(n)))
        s = C.string_at(p.value, n.value).decode()
        C.CDLL(None).free(p)
        return s

    def invtrans_vordiv2wind(self, nb_fields, vorticity_spectra, divergence_spectra, wind_fields):
        """TransLocal.cc:1486-1490 (pointer API)"""
        return self.invtrans(0, None, nb_fields, vorticity_spectra, divergence_spectra, wind_fields)

    # ---- not implemented by TransLocal (TransLocal.cc:848-857,899-927,1599-1685) ----
    def dirtrans(self, nb_fields, scalar_fields, scalar_spectra):
        _lib.check(_lib.Trans_dirtrans_scalar(self._h, nb_fields, None, None))

    def dirtrans_wind2vordiv(self, nb_fields, wind, vor, div):
        _lib.check(_lib.Trans_dirtrans_wind2vordiv(self._h, nb_fields, None, None, None))

    def invtrans_adj(self, nb_fields, gp, sp):
        _lib.check(_lib.Trans_invtrans_adj_scalar(self._h, nb_fields, None, None))

    # ---- streams / profiling ----
    def synchronize(self):
        _lib.check(_lib.Trans_synchronize(self._h))

    def stream(self):
        return _lib.Trans_stream(self._h)

    def set_stream(self, hip_stream):
        _lib.check(_lib.Trans_set_stream(self._h, hip_stream))

    def use_torch_stream(self):
        """run on torch's current CUDA(HIP) stream so that torch tensors are stream-ordered with the transform"""
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def set_profile(self, on):
        _lib.check(_lib.Trans_set_profile(self._h, int(bool(on))))

    def timings(self, reset=False):
        out = (C.c_double * 4)()
        vd = (C.c_double * 2)()
        _lib.check(_lib.Trans_timings_vordiv(self._h, vd, 0))
        _lib.check(_lib.Trans_timings(self._h, out, int(reset)))
        return {"legendre_ms": out[0], "legendre_calls": int(out[1]), "fourier_ms": out[2],
                "fourier_calls": int(out[3]), "prepare_ms": vd[0], "prepare_calls": int(vd[1])}

    # ---- Legendre cache (TransLocal.cc:608-647) ----
    def legendre_cache(self):
        n = _lib.Trans_legendre_cache_size(self._h)
        buf = np.zeros(n, dtype=np.uint8)
        _lib.check(_lib.Trans_legendre_cache_export(self._h, buf.ctypes.data, n))
        return buf

    # ---- stage API (multi-GPU driver, stage-level parity tests) ----
    def fourier_row_pitch(self, nf):
        return _lib.Trans_fourier_row_pitch(self._h, nf)

    def fourier_size(self, nf):
        return _lib.Trans_fourier_size(self._h, nf)

    def owned_wavenumbers(self):
        return _lib.Trans_owned_wavenumbers(self._h)

    def bands(self):
        out = np.zeros(self.nparts + 1, dtype=np.int32)
        _lib.Trans_bands(self._h, out.ctypes.data)
        return out

    def mirror_rows(self):
        """shard="mirror": (b0, b1) -- this object transforms rows [b0, b1) and their mirror images [ny-b1, ny-b0)"""
        out = np.zeros(2, dtype=np.int32)
        _lib.check(_lib.Trans_mirror_rows(self._h, out.ctypes.data))
        return int(out[0]), int(out[1])

    def owned_rows(self):
        """latitude rows of the grid whose points the invtrans output holds, in output order"""
        ny = len(self.grid.nx())
        if self.shard == "mirror":
            b0, b1 = self.mirror_rows()
            return np.concatenate([np.arange(b0, b1), np.arange(ny - b1, ny - b0)])
        if self.rows is not None:
            return np.arange(int(self.rows[0]), int(self.rows[1]))
        if self.nparts > 1:
            b = self.bands()
            return np.arange(int(b[self.part]), int(b[self.part + 1]))
        return np.arange(ny)

    def nlat0(self):
        out = np.zeros(self.truncation() + 1, dtype=np.int32)
        _lib.Trans_nlat0(self._h, out.ctypes.data)
        return out

    def fft_row_classes(self):
        """(nlats, 3) int array: FftMethod, transform length M and kernel kind of every latitude row (include/atlas_amd.h)"""
        out = np.zeros((len(self.grid.nx()), 3), dtype=np.int32)
        _lib.check(_lib.Trans_fft_row_classes(self._h, out.ctypes.data))
        return out

    def fourier_launch_plan(self):
        """{launches of one Fourier stage, the fused coarse-class launch among them, native-row launches with two fields per workgroup}"""
        out = (C.c_int * 3)()
        _lib.check(_lib.Trans_fourier_launch_plan(self._h, out))
        return {"launches": out[0], "coarse_fused": out[1], "native_two_fields": out[2]}

    def legendre_flops(self, nf):
        return _lib.Trans_legendre_flops(self._h, nf)

    def legendre_table_bytes(self):
        return _lib.Trans_legendre_table_bytes(self._h)

    def legendre_table(self):
        """the tile-blocked Legendre table as it sits in device memory (test hook)"""
        out = np.empty(self.legendre_table_bytes() // 8, dtype=np.float64)
        _lib.check(_lib.Trans_legendre_table_download(self._h, out.ctypes.data, out.size))
        return out

    def legendre_device(self, truncation_in, nf, spectra, fourier):
        with _lib.torch_stream_order(self.stream()):
            _lib.check(_lib.Trans_legendre_device(self._h, truncation_in, nf, _ptr(spectra), _ptr(fourier)))

    def fourier_device(self, nf, nb_vordiv, parts, part_cnt, gp):
        n = len(parts)
        bases = (C.c_void_p * n)(*[_ptr(p) for p in parts])
        cnts = (C.c_int * n)(*[int(c) for c in part_cnt])
        with _lib.torch_stream_order(self.stream()):
            _lib.check(_lib.Trans_fourier_device(self._h, nf, nb_vordiv, bases, cnts, _ptr(gp)))


class VorDivToUV:
    """atlas::trans::VorDivToUV (src/atlas/trans/VorDivToUV.h:36-133, "local": VorDivToUVLocal.cc:62-189)"""

    def __init__(self, truncation, type="local"):
        if type not in ("local", "mi355x"):
            raise ValueError(f"no VorDivToUV backend '{type}'")
        self._truncation = int(truncation)

    def truncation(self):
        return self._truncation

    def execute(self, nb_coeff, nb_fields, vorticity, divergence, U, V):
        """U, V = u cos(lat), v cos(lat) spectra; arrays of nb_coeff * nb_fields doubles in the invtrans layout"""
        n = int(nb_coeff) * int(nb_fields)
        dev = _is_device(U)
        ptrs = [_ptr(a, n, name, writable=w) for a, name, w in ((vorticity, "vorticity", False),
                                                                (divergence, "divergence", False),
                                                                (U, "U", True), (V, "V", True))]
        if dev:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.VorDivToUV_execute_device(self._truncation, int(nb_coeff), int(nb_fields), *ptrs, stream))
        else:
            _lib.check(_lib.VorDivToUV_execute(self._truncation, int(nb_coeff), int(nb_fields), *ptrs))
        return U, V


def legendre_cache_grid_hash(y_degrees):
    """hash(const Grid&) for structured grids (LegendreCacheCreatorLocal.cc:40-53): MD5 over std::lround(y * 1e8) of every
    row, each added as a 64-bit integer (csrc/legendre_cache_uid.cpp).  Pinned by the uid strings the reference's own test
    expects, src/tests/trans/test_trans.cc:600-696: tests/test_host_logic.py."""
    y = np.ascontiguousarray(y_degrees, dtype=np.float64)
    g = StructuredGrid(nx=[4] * len(y), y=y)
    uid = LegendreCacheCreator(g, 0).uid()
    if not uid.startswith("local-T0-grid-"):
        raise ValueError("these rows form a named family (L / S grid): no grid hash in their uid")
    return uid[len("local-T0-grid-"):len("local-T0-grid-") + 10]


class LegendreCacheCreator:
    """atlas::trans::LegendreCacheCreator for type "local" (src/atlas/trans/LegendreCacheCreator.h:30-111,
    local/LegendreCacheCreatorLocal.cc:66-165), through the C ABI (atlas_amd__LegendreCacheCreator__*)"""

    def __init__(self, grid, truncation, flt=False):
        if isinstance(grid, str):
            grid = StructuredGrid(name=grid)
        self.grid, self._truncation, self._flt = grid, int(truncation), bool(flt)

    def supported(self):
        return bool(_lib.LegendreCacheCreator_supported(self.grid._h))

    def uid(self):
        buf = C.create_string_buffer(256)
        _lib.check(_lib.LegendreCacheCreator_uid(self.grid._h, self._truncation, int(self._flt), buf, 256))
        return buf.value.decode()

    def estimate(self):
        return int(_lib.LegendreCacheCreator_estimate(self._truncation))          # LegendreCacheCreatorLocal.cc:162-164

    def create(self, path=None):
        """create() -> cache bytes (np.uint8), create(path) -> writes the file TransLocal's write_legendre would
        (TransLocal.cc:608-647); needs a GPU (the tables are built by a Trans object)"""
        blob = Trans(self.grid, self._truncation).legendre_cache()
        if path is None:
            return blob
        with open(path, "wb") as f:
            f.write(blob.tobytes())
        return path


_RT_new = _lib._sig("atlas_amd__RegionalTrans__new", C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int)
_RT_new_points = _lib._sig("atlas_amd__RegionalTrans__new_unstructured", C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int)
_RT_delete = _lib._sig("atlas_amd__RegionalTrans__delete", None, C.c_void_p)
_RT_npts = _lib._sig("atlas_amd__RegionalTrans__nb_gridpoints", C.c_int64, C.c_void_p)
_RT_invtrans = _lib._sig("atlas_amd__RegionalTrans__invtrans_scalar", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
_RT_invtrans_dev = _lib._sig("atlas_amd__RegionalTrans__invtrans_scalar_device", C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                             C.c_void_p)
_RT_invtrans_vd = _lib._sig("atlas_amd__RegionalTrans__invtrans_vordiv", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                            C.c_void_p, C.c_void_p, C.c_void_p)
_RT_sync = _lib._sig("atlas_amd__RegionalTrans__synchronize", C.c_int, C.c_void_p)
_RT_stream = _lib._sig("atlas_amd__RegionalTrans__stream", C.c_void_p, C.c_void_p)


class RegionalTrans:
    """trans::Trans for a regular longitude-latitude target that is not a crop of a global grid (TransLocal's no_nest branch):
    latitudes `lats` (degrees, any order), longitudes west + i * dlon for i < nlon.  Scalar fields; grid points
    [field][lat][lon]."""

    def __init__(self, nlon, west, dlon, lats, truncation):
        self._lats = np.ascontiguousarray(lats, dtype=np.float64)
        self.nlon, self.nlat, self.truncation = int(nlon), int(self._lats.size), int(truncation)
        self._h = _lib.check_ptr(_RT_new(self.nlon, float(west), float(dlon), self.nlat, self._lats.ctypes.data, self.truncation))

    @classmethod
    def unstructured(cls, lons, lats, truncation):
        """target = a list of (lon, lat) points in degrees (TransLocal's unstructured path); grid points [field][point]"""
        self = cls.__new__(cls)
        self._lons = np.ascontiguousarray(lons, dtype=np.float64)
        self._lats = np.ascontiguousarray(lats, dtype=np.float64)
        assert self._lons.size == self._lats.size
        self.nlon, self.nlat, self.truncation = 0, int(self._lats.size), int(truncation)
        self._h = _lib.check_ptr(_RT_new_points(self.nlat, self._lons.ctypes.data, self._lats.ctypes.data, self.truncation))
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _RT_delete is not None:
            _RT_delete(h)
            self._h = None

    def nb_gridpoints(self):
        return int(_RT_npts(self._h))

    def stream(self):
        return _RT_stream(self._h)

    def synchronize(self):
        _lib.check(_RT_sync(self._h))

    def invtrans(self, nb_fields, scalar_spectra, gp_fields):
        """host arrays (numpy) or device arrays (torch, asynchronous, ordered against torch's current stream)"""
        if isinstance(scalar_spectra, np.ndarray):
            sp = np.ascontiguousarray(scalar_spectra, dtype=np.float64)
            assert gp_fields.dtype == np.float64 and gp_fields.flags.c_contiguous
            assert gp_fields.size == nb_fields * self.nb_gridpoints()
            _lib.check(_RT_invtrans(self._h, int(nb_fields), sp.ctypes.data, gp_fields.ctypes.data))
            return gp_fields
        assert gp_fields.numel() == nb_fields * self.nb_gridpoints()
        with _lib.torch_stream_order(self.stream()):
            _lib.check(_RT_invtrans_dev(self._h, int(nb_fields), scalar_spectra.data_ptr(), gp_fields.data_ptr()))
        return gp_fields

    def invtrans_vordiv(self, nb_scalar, scalar_spectra, nb_vordiv, vorticity_spectra, divergence_spectra, gp_fields):
        """host arrays: gp = [u fields][v fields][scalar fields], each [lat][lon]"""
        sp = np.ascontiguousarray(scalar_spectra, dtype=np.float64) if nb_scalar > 0 else None
        vor = np.ascontiguousarray(vorticity_spectra, dtype=np.float64)
        div = np.ascontiguousarray(divergence_spectra, dtype=np.float64)
        assert gp_fields.dtype == np.float64 and gp_fields.size == (nb_scalar + 2 * nb_vordiv) * self.nb_gridpoints()
        _lib.check(_RT_invtrans_vd(self._h, int(nb_scalar), sp.ctypes.data if sp is not None else None, int(nb_vordiv),
                                   vor.ctypes.data, div.ctypes.data, gp_fields.ctypes.data))
        return gp_fields
