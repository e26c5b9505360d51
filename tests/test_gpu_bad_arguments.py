"""Bad arguments come back as error codes with a message, never as a crash or a silent wrong answer (the reference throws
eckit::Exception out of ATLAS_ASSERT / ATLAS_NOTIMPLEMENTED; across the C ABI that is a nonzero return and atlas_amd__last_error)."""
import ctypes as C

import numpy as np
import pytest

import atlas_amd
from atlas_amd import _lib

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
_sig = _lib._sig


def _err():
    return _lib.last_error().decode()


def test_trans_constructor_refuses_nonsense():
    g = atlas_amd.Grid("O8")
    new = _sig("atlas_amd__Trans__new", C.c_void_p, C.c_void_p, C.c_int)
    assert not new(None, 7) and _err()
    assert not new(g._h, -1) and _err()
    newc = _sig("atlas_amd__Trans__new_config", C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_size_t)
    assert not newc(g._h, 7, b"no_such_key=1", None, 0) and "no_such_key" in _err()
    assert not newc(g._h, 7, b"nparts=2;part=2", None, 0) and _err()
    assert not newc(g._h, 7, b"fft=cufft", None, 0) and _err()
    junk = (C.c_char * 64)(*b"not a legendre cache")
    assert not newc(g._h, 7, b"", junk, 64) and _err()
    # and the object is still usable afterwards
    tr = atlas_amd.Trans(g, 7)
    assert tr.truncation() == 7


def test_transform_calls_refuse_null_and_negative():
    g = atlas_amd.Grid("O8")
    tr = atlas_amd.Trans(g, 7)
    n = tr.nb_spectral_coefficients()
    sp = np.zeros(2 * n)
    gp = np.full(6 * g.size(), 3.0)
    scalar = _sig("atlas_amd__Trans__invtrans_scalar", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
    assert scalar(None, 1, sp.ctypes.data, gp.ctypes.data) != 0 and _err()
    assert scalar(tr._h, -1, sp.ctypes.data, gp.ctypes.data) != 0 and _err()
    assert scalar(tr._h, 1, None, gp.ctypes.data) != 0 and _err()
    assert scalar(tr._h, 1, sp.ctypes.data, None) != 0 and _err()
    inv = _sig("atlas_amd__Trans__invtrans", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
    assert inv(tr._h, 1, None, 0, None, None, gp.ctypes.data) != 0 and _err()            # scalars announced, none given
    assert inv(tr._h, 0, None, 1, sp.ctypes.data, None, gp.ctypes.data) != 0 and _err()  # divergence missing
    assert inv(tr._h, 0, None, -2, sp.ctypes.data, sp.ctypes.data, gp.ctypes.data) != 0 and _err()
    dev = _sig("atlas_amd__Trans__invtrans_scalar_device", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
    assert dev(tr._h, 1, None, None) != 0 and _err()
    assert (gp == 3.0).all()
    # the object still transforms
    out = np.full(g.size(), np.nan)
    tr.invtrans(1, sp[:n], out)
    assert not out.any()


def test_halo_exchange_refuses_use_before_setup_and_bad_ranks():
    hx_new = _sig("atlas_amd__HaloExchange__new", C.c_void_p)
    h = hx_new()
    f = np.zeros(10)
    ex = _sig("atlas_amd__HaloExchange__execute_strided_double", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
    one = (C.c_int * 1)(1)
    assert ex(h, f.ctypes.data, one, one, 1) != 0 and _err()                    # not set up
    setup = _sig("atlas_amd__HaloExchange__setup", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
    part = np.zeros(10, dtype=np.int32)
    ridx = np.arange(10, dtype=np.int32)
    ridx[9] = 99                                                                # owner index outside the partition
    assert setup(h, part.ctypes.data, ridx.ctypes.data, 0, 10) != 0 and _err()
    ridx[9] = 2
    assert setup(h, part.ctypes.data, ridx.ctypes.data, 0, 10) == 0
    assert ex(h, f.ctypes.data, one, one, 7) != 0 and _err()                    # rank the reference does not support either
    assert ex(h, None, one, one, 1) != 0 and _err()
    f[:] = np.arange(10)
    assert ex(h, f.ctypes.data, one, one, 1) == 0 and f[9] == 2.0
    _sig("atlas_amd__HaloExchange__delete", None, C.c_void_p)(h)
