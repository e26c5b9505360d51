"""The C++ host interface include/atlas_amd.hpp (mirror of atlas::trans::Trans, parallel::HaloExchange,
functionspace::StructuredColumns above the C ABI): tests/cpp/test_trans_cxx.cc is compiled with g++ against the shared
library and run -- host-only cases on CPU, the transforms (analytic spherical harmonics at 1e-13 like
src/tests/trans/test_transgeneral.cc:829-839, solid-body rotation on the vor/div path, halo exchange on
StructuredColumns) on the GPU."""
import os
import subprocess

import pytest

from atlas_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cxx_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cxx") / "test_trans_cxx")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_trans_cxx.cc"), "-o", out,
           "-L", libdir, "-latlas_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _run(binary, *args):
    r = subprocess.run([binary, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_cxx_interface_host_cases(cxx_binary):
    out = _run(cxx_binary, "--host-only")
    assert "0 failure(s)" in out
    for case in ("backend_registry", "grids", "halo_index_logic", "node_columns_contract"):
        assert f"ok     {case}" in out


def test_c_header_compiles_as_c(tmp_path):
    # the boundary is a C ABI: the header must be usable from plain C
    src = tmp_path / "use.c"
    src.write_text('#include "atlas_amd.h"\nint main(void) { return atlas_amd__version() == 0; }\n')
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(tmp_path / "use"), "-L", libdir, "-latlas_amd", f"-Wl,-rpath,{libdir}",
                        "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(tmp_path / "use")]).returncode == 0


@pytest.mark.gpu
def test_cxx_interface_on_device(cxx_binary):
    out = _run(cxx_binary)
    assert "0 failure(s)" in out
    for case in ("invtrans_analytic_F32", "invtrans_analytic_O32", "invtrans_domain_analytic", "vordiv2wind_and_not_implemented",
                 "halo_exchange_on_structured_columns"):
        assert f"ok     {case}" in out


@pytest.fixture(scope="module")
def dist_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cxxdist") / "test_dist_cxx")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_dist_cxx.cc"), "-o", out,
           "-L", libdir, "-latlas_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_cxx_distributed_driver_builds_and_refuses_without_a_device(dist_binary):
    """tests/cpp/test_dist_cxx.cc (C ABI only: communicators, distributed transform, halo exchange between ranks) builds
    with g++; without a GPU it stops at the device check"""
    if _lib.device_count() > 0:
        pytest.skip("a device is present: covered by the gpu test")
    r = subprocess.run([dist_binary], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,grid,T,nf", [(2, "O32", 31, 3), (3, "O64", 63, 5), (4, "F32", 31, 2)])
def test_cxx_distributed_driver_on_device(dist_binary, nranks, grid, T, nf):
    """C++ driver, P ranks as threads over the library's local communicator on one GPU + one rank over real RCCL: the
    bands of atlas_amd__Trans__invtrans_distributed[_many] equal the single-device transform bit for bit; the halo
    exchange between the ranks (setup_comm / execute_comm) delivers every owner's value"""
    out = _run(dist_binary, str(nranks), grid, str(T), str(nf))
    assert "0 failure(s)" in out and out.count("ok     distributed transform") >= 2, out


@pytest.mark.gpu
def test_cxx_distributed_driver_with_real_peers_over_rccl(dist_binary):
    """only where >= 2 devices are visible (skipped on the one-GPU boxes of this pool): the same C++ driver with one rank per
    DEVICE over a real RCCL communicator -- the multi-peer ncclGroupStart .. ncclGroupEnd of csrc/comm.hip with an actual peer
    (VERDICT r5 item 5c); bands bit-identical to the single-device transform, halo exchange between the devices"""
    if _lib.device_count() < 2:
        pytest.skip("one visible device: RCCL with a real peer needs two")
    n = min(_lib.device_count(), 4)
    out = _run(dist_binary, str(n), "O64", "63", "5")
    assert "0 failure(s)" in out and f"{n} ranks over RCCL, one device each" in out and "skipped" not in out, out
