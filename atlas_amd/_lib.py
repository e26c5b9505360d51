"""ctypes loader for the C-ABI shared library (include/atlas_amd.h).

The library is the product: there is NO Python/NumPy fallback for any compute path.  If the shared object is
missing this module raises at import time; if no HIP device is visible, constructing a Trans raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libatlas_amd.so")
if os.environ.get("ATLAS_AMD_LIB"):   # dev builds of the same library (tools/: trace / ablation variants), never a fallback
    LIB_PATH = os.path.abspath(os.environ["ATLAS_AMD_LIB"])


class AtlasAmdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C atlas_amd/csrc`).  atlas_amd has no pure-Python fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_void_p = C.c_void_p


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


last_error = _sig("atlas_amd__last_error", C.c_char_p)
last_note = _sig("atlas_amd__last_note", C.c_char_p)
version = _sig("atlas_amd__version", C.c_char_p)
set_ignore_env = _sig("atlas_amd__set_ignore_env", C.c_int, C.c_int)
_effective_config = _sig("atlas_amd__effective_config", C.c_longlong, C.c_char_p, C.c_longlong)


def effective_config():
    """every environment switch of the library with the value in effect: {name: {"class", "value", "source", "default", "what"}}
    (source: default | env | ignored | compiled out) -- csrc/env.cpp, include/atlas_amd.h"""
    n = _effective_config(None, 0)
    buf = C.create_string_buffer(int(n))
    _effective_config(buf, n)
    out = {}
    for ln in buf.value.decode().splitlines():
        name, cls, value, source, default, what = ln.split("\t")
        out[name] = {"class": cls, "value": value, "source": source, "default": default, "what": what}
    return out
device_count = _sig("atlas_amd__device_count", C.c_int)
stream_wait_stream = _sig("atlas_amd__stream_wait_stream", C.c_int, c_void_p, c_void_p)
_diag_mfma_f64_rate = _sig("atlas_amd__diag_mfma_f64_rate", C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double))


def diag_mfma_f64_rate(target_ms=25.0, repeats=3):
    """TFLOP/s the current device sustains on v_mfma_f64_16x16x4_f64 alone (measurement aid, csrc/diag.hip)"""
    out = C.c_double(0.0)
    check(_diag_mfma_f64_rate(float(target_ms), int(repeats), C.byref(out)))
    return out.value


_diag_mfma_f32_rate = _sig("atlas_amd__diag_mfma_f32_rate", C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double))


def diag_mfma_f32_rate(target_ms=25.0, repeats=3):
    """TFLOP/s the current device sustains on v_mfma_f32_16x16x4_f32 alone (the fp32 variant's Legendre instruction)"""
    out = C.c_double(0.0)
    check(_diag_mfma_f32_rate(float(target_ms), int(repeats), C.byref(out)))
    return out.value


class torch_stream_order:
    """`with torch_stream_order(obj_stream):` -- device tensors handed to the library were produced on torch's current
    stream and will be consumed there: the object's stream first waits for torch's stream, and torch's stream then waits
    for what was submitted inside the block (events, no host synchronisation)."""

    def __init__(self, obj_stream):
        import torch
        self.obj = obj_stream
        self.torch = torch.cuda.current_stream().cuda_stream

    def __enter__(self):
        check(stream_wait_stream(self.obj, self.torch))
        return self

    def __exit__(self, *exc):
        check(stream_wait_stream(self.torch, self.obj))
        return False

Grid_new_gaussian = _sig("atlas_amd__Grid__new_gaussian", c_void_p, C.c_char_p)
Grid_new_structured = _sig("atlas_amd__Grid__new_structured", c_void_p, C.c_int, c_void_p, c_void_p)
Grid_delete = _sig("atlas_amd__Grid__delete", None, c_void_p)
Grid_ny = _sig("atlas_amd__Grid__ny", C.c_int, c_void_p)
Grid_nxmax = _sig("atlas_amd__Grid__nxmax", C.c_int, c_void_p)
Grid_size = _sig("atlas_amd__Grid__size", C.c_int64, c_void_p)
Grid_regular = _sig("atlas_amd__Grid__regular", C.c_int, c_void_p)
Grid_nx = _sig("atlas_amd__Grid__nx", C.c_int, c_void_p, c_void_p)
Grid_y = _sig("atlas_amd__Grid__y", C.c_int, c_void_p, c_void_p)
Grid_crop_to_domain = _sig("atlas_amd__Grid__crop_to_domain", C.c_int, c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                           C.POINTER(C.c_int), C.POINTER(C.c_int), c_void_p, c_void_p, C.c_int)
gaussian_latitudes_npole_spole = _sig("atlas_amd__gaussian_latitudes_npole_spole", C.c_int, C.c_int, c_void_p)

Trans_new = _sig("atlas_amd__Trans__new", c_void_p, c_void_p, C.c_int)
Trans_new_config = _sig("atlas_amd__Trans__new_config", c_void_p, c_void_p, C.c_int, C.c_char_p, c_void_p, C.c_size_t)
Trans_delete = _sig("atlas_amd__Trans__delete", None, c_void_p)
Trans_truncation = _sig("atlas_amd__Trans__truncation", C.c_int, c_void_p)
Trans_nb_gridpoints = _sig("atlas_amd__Trans__nb_gridpoints", C.c_int64, c_void_p)
Trans_nb_gridpoints_global = _sig("atlas_amd__Trans__nb_gridpoints_global", C.c_int64, c_void_p)
Trans_nb_spectral_coefficients = _sig("atlas_amd__Trans__nb_spectral_coefficients", C.c_int64, c_void_p)
Trans_invtrans_scalar = _sig("atlas_amd__Trans__invtrans_scalar", C.c_int, c_void_p, C.c_int, c_void_p, c_void_p)
Trans_invtrans = _sig("atlas_amd__Trans__invtrans", C.c_int, c_void_p, C.c_int, c_void_p, C.c_int, c_void_p,
                      c_void_p, c_void_p)
Trans_invtrans_vordiv2wind = _sig("atlas_amd__Trans__invtrans_vordiv2wind", C.c_int, c_void_p, C.c_int, c_void_p,
                                  c_void_p, c_void_p)
Trans_invtrans_scalar_device = _sig("atlas_amd__Trans__invtrans_scalar_device", C.c_int, c_void_p, C.c_int, c_void_p,
                                    c_void_p)
Trans_invtrans_device = _sig("atlas_amd__Trans__invtrans_device", C.c_int, c_void_p, C.c_int, c_void_p, C.c_int,
                             c_void_p, c_void_p, c_void_p)
Trans_invtrans_scalar_device_f32 = _sig("atlas_amd__Trans__invtrans_scalar_device_f32", C.c_int, c_void_p, C.c_int,
                                        c_void_p, c_void_p)
Trans_invtrans_device_f32 = _sig("atlas_amd__Trans__invtrans_device_f32", C.c_int, c_void_p, C.c_int, c_void_p, C.c_int,
                                 c_void_p, c_void_p, c_void_p)
Trans_invtrans_scalar_f32 = _sig("atlas_amd__Trans__invtrans_scalar_f32", C.c_int, c_void_p, C.c_int, c_void_p, c_void_p)
Trans_dirtrans_scalar = _sig("atlas_amd__Trans__dirtrans_scalar", C.c_int, c_void_p, C.c_int, c_void_p, c_void_p)
Trans_dirtrans_wind2vordiv = _sig("atlas_amd__Trans__dirtrans_wind2vordiv", C.c_int, c_void_p, C.c_int, c_void_p,
                                  c_void_p, c_void_p)
Trans_invtrans_adj_scalar = _sig("atlas_amd__Trans__invtrans_adj_scalar", C.c_int, c_void_p, C.c_int, c_void_p,
                                 c_void_p)
Trans_invtrans_adj = _sig("atlas_amd__Trans__invtrans_adj", C.c_int, c_void_p, C.c_int, c_void_p, C.c_int, c_void_p,
                          c_void_p, c_void_p)
Trans_invtrans_vordiv2wind_adj = _sig("atlas_amd__Trans__invtrans_vordiv2wind_adj", C.c_int, c_void_p, C.c_int,
                                      c_void_p, c_void_p, c_void_p)
Trans_has_backend = _sig("atlas_amd__Trans__has_backend", C.c_int, C.c_char_p)
Trans_set_backend = _sig("atlas_amd__Trans__set_backend", C.c_int, C.c_char_p)
Trans_backend = _sig("atlas_amd__Trans__backend", C.c_int, C.POINTER(c_void_p), C.POINTER(C.c_size_t))
Trans_grid = _sig("atlas_amd__Trans__grid", c_void_p, c_void_p)


class Field(C.Structure):
    """atlas_amd_Field: host data pointer + C-order shape (what array::make_view sees)"""
    _fields_ = [("data", c_void_p), ("rank", C.c_int), ("shape", C.c_long * 2)]


_FP = C.POINTER(Field)
Trans_invtrans_field = _sig("atlas_amd__Trans__invtrans_field", C.c_int, c_void_p, _FP, _FP)
Trans_invtrans_fieldset = _sig("atlas_amd__Trans__invtrans_fieldset", C.c_int, c_void_p, _FP, C.c_int, _FP, C.c_int)
Trans_invtrans_vordiv2wind_field = _sig("atlas_amd__Trans__invtrans_vordiv2wind_field", C.c_int, c_void_p, _FP, _FP, _FP)
Trans_invtrans_grad_field = _sig("atlas_amd__Trans__invtrans_grad_field", C.c_int, c_void_p, _FP, _FP)
Trans_invtrans_adj_field = _sig("atlas_amd__Trans__invtrans_adj_field", C.c_int, c_void_p, _FP, _FP)
Trans_invtrans_adj_fieldset = _sig("atlas_amd__Trans__invtrans_adj_fieldset", C.c_int, c_void_p, _FP, C.c_int, _FP,
                                   C.c_int)
Trans_invtrans_grad_adj_field = _sig("atlas_amd__Trans__invtrans_grad_adj_field", C.c_int, c_void_p, _FP, _FP)
Trans_invtrans_vordiv2wind_adj_field = _sig("atlas_amd__Trans__invtrans_vordiv2wind_adj_field", C.c_int, c_void_p, _FP,
                                            _FP, _FP)
Trans_dirtrans_field = _sig("atlas_amd__Trans__dirtrans_field", C.c_int, c_void_p, _FP, _FP)
Trans_dirtrans_fieldset = _sig("atlas_amd__Trans__dirtrans_fieldset", C.c_int, c_void_p, _FP, C.c_int, _FP, C.c_int)
Trans_dirtrans_wind2vordiv_field = _sig("atlas_amd__Trans__dirtrans_wind2vordiv_field", C.c_int, c_void_p, _FP, _FP, _FP)
VorDivToUV_execute = _sig("atlas_amd__VorDivToUV__execute", C.c_int, C.c_int, C.c_int, C.c_int, c_void_p, c_void_p,
                          c_void_p, c_void_p)
VorDivToUV_execute_device = _sig("atlas_amd__VorDivToUV__execute_device", C.c_int, C.c_int, C.c_int, C.c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p)
Trans_stream = _sig("atlas_amd__Trans__stream", c_void_p, c_void_p)
Trans_set_stream = _sig("atlas_amd__Trans__set_stream", C.c_int, c_void_p, c_void_p)
Trans_synchronize = _sig("atlas_amd__Trans__synchronize", C.c_int, c_void_p)
Trans_legendre_cache_size = _sig("atlas_amd__Trans__legendre_cache_size", C.c_size_t, c_void_p)
Trans_legendre_cache_export = _sig("atlas_amd__Trans__legendre_cache_export", C.c_int, c_void_p, c_void_p, C.c_size_t)
Trans_fourier_row_pitch = _sig("atlas_amd__Trans__fourier_row_pitch", C.c_int, c_void_p, C.c_int)
Trans_fourier_size = _sig("atlas_amd__Trans__fourier_size", C.c_int64, c_void_p, C.c_int)
Trans_owned_wavenumbers = _sig("atlas_amd__Trans__owned_wavenumbers", C.c_int, c_void_p)
Trans_bands = _sig("atlas_amd__Trans__bands", C.c_int, c_void_p, c_void_p)
Trans_legendre_device = _sig("atlas_amd__Trans__legendre_device", C.c_int, c_void_p, C.c_int, C.c_int, c_void_p,
                             c_void_p)
Trans_fourier_device = _sig("atlas_amd__Trans__fourier_device", C.c_int, c_void_p, C.c_int, C.c_int, c_void_p,
                            c_void_p, c_void_p)
Trans_nlat0 = _sig("atlas_amd__Trans__nlat0", C.c_int, c_void_p, c_void_p)
Trans_fft_row_classes = _sig("atlas_amd__Trans__fft_row_classes", C.c_int, c_void_p, c_void_p)
Trans_legendre_flops = _sig("atlas_amd__Trans__legendre_flops", C.c_double, c_void_p, C.c_int)
Trans_legendre_table_bytes = _sig("atlas_amd__Trans__legendre_table_bytes", C.c_int64, c_void_p)
Trans_mirror_rows = _sig("atlas_amd__Trans__mirror_rows", C.c_int, c_void_p, c_void_p)
mirror_bands = _sig("atlas_amd__mirror_bands", C.c_int, c_void_p, C.c_int, c_void_p)
latitude_bands = _sig("atlas_amd__latitude_bands", C.c_int, c_void_p, C.c_int, C.c_int, c_void_p)
trans_geometry_probe = _sig("atlas_amd__trans_geometry_probe", C.c_int, c_void_p, C.c_int, C.c_int, c_void_p, c_void_p)
Trans_legendre_table_download = _sig("atlas_amd__Trans__legendre_table_download", C.c_int, c_void_p, c_void_p,
                                     C.c_size_t)
legendre_gen_host_selfcheck = _sig("atlas_amd__legendre_gen_host_selfcheck", C.c_int, c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
Trans_timings = _sig("atlas_amd__Trans__timings", C.c_int, c_void_p, c_void_p, C.c_int)
Trans_fourier_launch_plan = _sig("atlas_amd__Trans__fourier_launch_plan", C.c_int, c_void_p, c_void_p)
Trans_timings_vordiv = _sig("atlas_amd__Trans__timings_vordiv", C.c_int, c_void_p, c_void_p, C.c_int)
Trans_set_profile = _sig("atlas_amd__Trans__set_profile", C.c_int, c_void_p, C.c_int)
Trans_fft_phase_profile = _sig("atlas_amd__Trans__fft_phase_profile", C.c_int, c_void_p, C.c_int, c_void_p)
Trans_fft_trace = _sig("atlas_amd__Trans__fft_trace", C.c_int, c_void_p, C.c_ulonglong, c_void_p)

fourier_truncation = _sig("atlas_amd__fourier_truncation", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                          C.c_int)
legendre_reference_sizes = _sig("atlas_amd__legendre_reference_sizes", C.c_int, c_void_p, C.c_int,
                                C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
legendre_reference_tables = _sig("atlas_amd__legendre_reference_tables", C.c_int, c_void_p, C.c_int, c_void_p,
                                 C.c_size_t, c_void_p, C.c_size_t)
LegendreCacheCreator_uid = _sig("atlas_amd__LegendreCacheCreator__uid", C.c_int, c_void_p, C.c_int, C.c_int, C.c_char_p,
                                C.c_size_t)
LegendreCacheCreator_estimate = _sig("atlas_amd__LegendreCacheCreator__estimate", C.c_int64, C.c_int)
LegendreCacheCreator_supported = _sig("atlas_amd__LegendreCacheCreator__supported", C.c_int, c_void_p)
fft_host_row = _sig("atlas_amd__fft_host_row", C.c_int, C.c_int, c_void_p, C.c_int, c_void_p)
fft_host_row_generic = _sig("atlas_amd__fft_host_row_generic", C.c_int, C.c_int, c_void_p, C.c_int, c_void_p)
fft_host_row_bluestein = _sig("atlas_amd__fft_host_row_bluestein", C.c_int, C.c_int, c_void_p, C.c_int, c_void_p)
fft_host_row_native = _sig("atlas_amd__fft_host_row_native", C.c_int, C.c_int, c_void_p, C.c_int, c_void_p)
fft_plan_info = _sig("atlas_amd__fft_plan_info", C.c_int, C.c_int, C.c_int, c_void_p)
fft_host_row_hybrid = _sig("atlas_amd__fft_host_row_hybrid", C.c_int, C.c_int, c_void_p, C.c_int, c_void_p)
fft_host_row_coarse = _sig("atlas_amd__fft_host_row_coarse", C.c_int, C.c_int, c_void_p, C.c_int, c_void_p)


def check(rc):
    if rc != 0:
        msg = last_error().decode("utf-8", "replace")
        if msg.startswith("Not implemented"):
            raise NotImplementedError(msg)
        raise AtlasAmdError(msg)


def check_ptr(p):
    if not p:
        raise AtlasAmdError(last_error().decode("utf-8", "replace"))
    return p
