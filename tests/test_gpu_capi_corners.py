"""Entry points of include/atlas_amd.h that no other test reaches, called as a C / Fortran program would (ctypes, plain pointers):
the strided HaloExchange calls of atlas_f for int / long / float (HaloExchange.cc:195-327; the double ones: test_gpu_halo.py and
tests/fortran), the shape-less execute_<T>(field, var_rank) twins, atlas__Trans__spectral / atlas__Trans__handle
(TransInterface.h:98,104) and the device helpers for callers without HIP headers."""
import ctypes as C

import numpy as np
import pytest

import atlas_amd
from atlas_amd import _lib
from atlas_amd.parallel import HaloExchange, HX_strided
from oracle.halo import HaloExchangeOracle

pytestmark = pytest.mark.gpu

_sig = _lib._sig
NAMES = {np.dtype(np.int32): "int", np.dtype(np.int64): "long", np.dtype(np.float32): "float", np.dtype(np.float64): "double"}


def _single_process_pattern(n=500, nowned=430, seed=11):
    rng = np.random.default_rng(seed)
    part = np.zeros(n, dtype=np.int32)
    ridx = np.arange(n, dtype=np.int32)
    ridx[nowned:] = rng.integers(0, nowned, n - nowned)
    hx = HaloExchange()
    hx.setup(part, ridx, 0, n)
    orc = [HaloExchangeOracle(0, 1)]
    HaloExchangeOracle.setup(orc, [part], [ridx], 0, [n])
    return hx, orc, n, rng


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
def test_strided_calls_of_every_type_as_atlas_f_passes_them(dtype):
    """atlas_HaloExchange_module.fypp:112-132: field(nvar, nnodes) -> strides (nvar, 1), extents (1, nvar), rank 2;
    field(nlev, nvar, nnodes) -> strides (nvar nlev, nlev, 1), extents (1, nvar, nlev), rank 3.  Bit-exact against the oracle,
    forward and adjoint."""
    hx, orc, n, rng = _single_process_pattern()
    for shape, strides, extents in (((n, 4), [4, 1], [1, 4]), ((n, 3, 5), [15, 5, 1], [1, 3, 5]), ((n,), [1], [1])):
        a = (rng.standard_normal(shape) * 100).astype(dtype)
        for adjoint in (False, True):
            ref = a.copy()
            (HaloExchangeOracle.execute_adjoint if adjoint else HaloExchangeOracle.execute)(orc, [ref])
            got = a.copy()
            fn = HX_strided[("_adjoint" if adjoint else "", NAMES[np.dtype(dtype)])]
            vs, ve = (C.c_int * len(strides))(*strides), (C.c_int * len(extents))(*extents)
            _lib.check(fn(hx._h, got.ctypes.data, vs, ve, len(extents)))
            assert np.array_equal(got, ref), (dtype, shape, adjoint)


@pytest.mark.parametrize("name,dtype", [("int", np.int32), ("float", np.float32), ("double", np.float64)])
def test_shapeless_execute_twins(name, dtype):
    """atlas__HaloExchange__execute[_adjoint]_{int,float,double}(This, field, var_rank) are declared without a shape
    (HaloExchange.h:441-443,453-455): defined here for var_rank 0 (one value per node), an error otherwise"""
    hx, orc, n, rng = _single_process_pattern(seed=5)
    a = (rng.standard_normal(n) * 50).astype(dtype)
    for adj in ("", "_adjoint"):
        fn = _sig(f"atlas_amd__HaloExchange__execute{adj}_{name}", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        ref = a.copy()
        (HaloExchangeOracle.execute_adjoint if adj else HaloExchangeOracle.execute)(orc, [ref])
        got = a.copy()
        _lib.check(fn(hx._h, got.ctypes.data, 0))
        assert np.array_equal(got, ref)
        untouched = a.copy()
        assert fn(hx._h, untouched.ctypes.data, 2) != 0 and b"var_rank" in _lib.last_error()
        assert np.array_equal(untouched, a)


def test_spectral_function_space_and_handle_of_a_trans():
    T = 47
    tr = atlas_amd.Trans(atlas_amd.Grid("O48"), T)
    spectral = _sig("atlas_amd__Trans__spectral", C.c_void_p, C.c_void_p)(tr._h)
    assert spectral
    assert _sig("atlas_amd__Spectral__truncation", C.c_int, C.c_void_p)(spectral) == T
    for fn in ("atlas_amd__Spectral__nb_spectral_coefficients", "atlas_amd__Spectral__nb_spectral_coefficients_global"):
        assert _sig(fn, C.c_int64, C.c_void_p)(spectral) == (T + 1) * (T + 2) == tr.nb_spectral_coefficients()
    # TransImpl::handle() is ATLAS_NOTIMPLEMENTED for TransLocal (TransImpl.cc:20-22)
    h = C.c_int(-7)
    assert _sig("atlas_amd__Trans__handle", C.c_int, C.c_void_p, C.POINTER(C.c_int))(tr._h, C.byref(h)) != 0
    assert _lib.last_error().startswith(b"Not implemented") and h.value == -7


def test_device_helpers_for_callers_without_hip_headers():
    """a C / Fortran driver of the *_device entry points: set_device, device_malloc, memcpy both ways, device_synchronize"""
    assert _sig("atlas_amd__set_device", C.c_int, C.c_int)(0) == 0
    assert _sig("atlas_amd__set_device", C.c_int, C.c_int)(_lib.device_count()) != 0 and _lib.last_error()
    malloc = _sig("atlas_amd__device_malloc", C.c_void_p, C.c_size_t)
    free = _sig("atlas_amd__device_free", C.c_int, C.c_void_p)
    h2d = _sig("atlas_amd__device_memcpy_h2d", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
    d2h = _sig("atlas_amd__device_memcpy_d2h", C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
    sync = _sig("atlas_amd__device_synchronize", C.c_int)
    T, nf = 31, 2
    g = atlas_amd.Grid("F32")
    tr = atlas_amd.Trans(g, T)
    rng = np.random.default_rng(2)
    sp = rng.standard_normal(tr.nb_spectral_coefficients() * nf)
    gp = np.full(nf * g.size(), np.nan)
    d_sp, d_gp = malloc(sp.nbytes), malloc(gp.nbytes)
    assert d_sp and d_gp
    _lib.check(h2d(d_sp, sp.ctypes.data, sp.nbytes))
    fn = _sig("atlas_amd__Trans__invtrans_scalar_device", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
    _lib.check(fn(tr._h, nf, d_sp, d_gp))
    _lib.check(_sig("atlas_amd__Trans__synchronize", C.c_int, C.c_void_p)(tr._h))
    _lib.check(sync())
    _lib.check(d2h(gp.ctypes.data, d_gp, gp.nbytes))
    ref = np.empty_like(gp)
    tr.invtrans(nf, sp, ref)                       # the host-pointer entry point
    assert np.array_equal(gp, ref)
    _lib.check(free(d_sp))
    _lib.check(free(d_gp))


@pytest.mark.parametrize("form", ["tiles", "rows", ""])
def test_both_forms_of_the_field_transposition_of_the_halo_path(form, monkeypatch):
    """csrc/vd2uv_kernel.hip: grid points of a band [nf][npts] -> StructuredColumns field [npts][nf] (in front of the halo exchange
    of atlas_amd__Trans__invtrans_distributed_many_halo), 32 x 32 tiles or whole rows: a pure permutation, bitwise against torch;
    ragged point and field counts either side of the 64-blocks"""
    import torch
    if form:
        monkeypatch.setenv("ATLAS_AMD_GP_TO_FIELD", form)
    fn = _sig("atlas_amd__diag_gp_to_field", C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_double))
    for npts, nf in ((70001, 137), (64, 64), (65, 63), (1, 1), (4099, 200), (100000, 33)):
        gp = torch.randn(nf, npts, dtype=torch.float64, device="cuda")
        out = torch.full((npts, nf), float("nan"), dtype=torch.float64, device="cuda")
        ms = C.c_double(0.0)
        _lib.check(fn(gp.data_ptr(), out.data_ptr(), npts, nf, 1, C.byref(ms)))
        torch.cuda.synchronize()
        assert torch.equal(out, gp.t().contiguous()), (form, npts, nf)


def test_the_device_array_transform_can_be_captured_into_a_hip_graph():
    """[r6] A caller that replays a fixed sequence (a time step) captures it: after the first call of a shape (buffers, row tables and
    side streams exist) the device-array entry point issues only kernels and event fork / joins of its side streams from the caller's
    stream -- no allocation, no synchronisation -- so a stream capture (torch.cuda.CUDAGraph = hipStreamBeginCapture) takes it, and a
    replay is bit-identical to the direct call (scalar, fp32 and vor/div calls).  tools/probe/graph_probe.py: a replay costs what the
    direct call costs at TL1279 (15.28 against 15.36 ms) and 4 % less at TL319."""
    import torch
    from helpers import red_spectra
    g = atlas_amd.Grid("O160")
    T, nf, nvd = 159, 7, 3
    tr = atlas_amd.Trans(g, T)
    sp = torch.from_numpy(red_spectra(T, nf, seed=3)).cuda()
    vor = torch.from_numpy(red_spectra(T, nvd, seed=4)).cuda()
    div = torch.from_numpy(red_spectra(T, nvd, seed=5)).cuda()
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    gp32 = torch.zeros(nf * g.size(), dtype=torch.float32, device="cuda")
    sp32 = sp.float()
    gpv = torch.zeros((nf + 2 * nvd) * g.size(), dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        tr.use_torch_stream()

        def calls():
            tr.invtrans(nf, sp, gp)
            tr.invtrans(nf, sp32, gp32)
            tr.invtrans(nf, sp, nvd, vor, div, gpv)
        calls()                                   # first use of every shape: allocations happen here
        side.synchronize()
        refs = [t.clone() for t in (gp, gp32, gpv)]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            calls()
        for _ in range(2):
            for t in (gp, gp32, gpv):
                t.fill_(float("nan"))
            graph.replay()
            side.synchronize()
            assert all(torch.equal(a, b) for a, b in zip((gp, gp32, gpv), refs))
