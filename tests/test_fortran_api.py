"""A Fortran caller of the C ABI: tests/fortran/test_trans_f.f90 binds include/atlas_amd.h with bind(C) interface blocks the way
atlas_f binds atlas__Trans__* (src/atlas_f/trans/atlas_Trans_module.F90:156-177; TransInterface.h:74-79), is compiled with
amdflang against the shared library and run -- sizes, grids and error reporting on CPU; on the GPU the analytic spherical
harmonics at 1e-13 (src/tests/trans/test_transgeneral.cc:829-839) on F32 and O32 and the IFS-style call
invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp) on a solid-body rotation, and atlas_HaloExchange's setup / execute /
execute_adjoint with atlas_f's strides and extents (src/atlas_f/parallel/atlas_HaloExchange_module.fypp:88-120)."""
import os
import shutil
import subprocess

import pytest

from atlas_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FC = shutil.which("amdflang") or ("/opt/rocm/bin/amdflang" if os.path.exists("/opt/rocm/bin/amdflang") else None)


@pytest.fixture(scope="module")
def fortran_binary(tmp_path_factory):
    if FC is None:
        pytest.skip("no Fortran compiler (amdflang) in this image")
    d = tmp_path_factory.mktemp("fortran")
    out = str(d / "test_trans_f")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [FC, "-O1", "-J", str(d), os.path.join(ROOT, "tests", "fortran", "test_trans_f.f90"), "-o", out,
           "-L", libdir, "-latlas_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(d))
    assert r.returncode == 0, r.stderr
    return out


def _run(binary, *args):
    r = subprocess.run([binary, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_fortran_binding_host_cases(fortran_binary):
    out = _run(fortran_binary, "--host-only")
    assert "0 failure(s)" in out
    for case in ("grids_and_sizes", "errors_are_reported"):
        assert f"ok     {case}" in out


@pytest.mark.gpu
def test_fortran_caller_on_device(fortran_binary):
    out = _run(fortran_binary)
    assert "0 failure(s)" in out, out
    for case in ("invtrans_analytic_F32", "invtrans_analytic_O32", "invtrans_vordiv_with_scalar", "halo_exchange"):
        assert f"ok     {case}" in out
