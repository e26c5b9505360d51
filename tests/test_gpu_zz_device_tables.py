"""Last GPU test module of the session (the name sorts it after the others).

1. Device generation of the Legendre table (csrc/legendre_gen_kernel.hip; config key tables=device): the table in
device memory must equal the host-generated one bit for bit -- the kernels perform the reference's multiplications and
additions in the reference's order (LegendrePolynomials.cc:85-149) without contraction -- for the whole table and for
the wavenumber-sharded and latitude-band decompositions; the transform built on it then gives identical results.
2. The mirror-band decomposition (Trans(shard="mirror"), functionspace.MirrorBandColumns) and the adjoint halo exchange
on the device against the reference's expected arrays."""
import numpy as np
import pytest

import atlas_amd
from helpers import red_spectra

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gridname,T,kw", [("O32", 31, {}), ("F64", 63, {}), ("O160", 159, {}),
                                           ("O64", 63, {"nparts": 3, "part": 1}),
                                           ("O64", 63, {"nparts": 4, "part": 2, "shard": "band"}),
                                           ("O320", 319, {"nparts": 8, "part": 5, "shard": "band"})])
def test_device_generated_table_is_bit_identical(gridname, T, kw):
    g = atlas_amd.Grid(gridname)
    th = atlas_amd.Trans(g, T, tables="host", **kw)
    td = atlas_amd.Trans(g, T, tables="device", **kw)
    a, b = th.legendre_table(), td.legendre_table()
    assert a.size == b.size and a.size > 0
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), int((a.view(np.uint64) != b.view(np.uint64)).sum())


def test_transform_on_device_generated_table():
    g = atlas_amd.Grid("O64")
    T, nf = 63, 3
    sp = red_spectra(T, nf, seed=3)
    out = []
    for tables in ("host", "device"):
        tr = atlas_amd.Trans(g, T, tables=tables)
        gp = np.zeros(nf * g.size())
        tr.invtrans(nf, sp, gp)
        out.append(gp)
    assert np.abs(out[0]).max() > 0 and np.array_equal(out[0], out[1])


# ---------------------------------------------------------------- mirror-band decomposition (shard="mirror")
# The kernels and the crop path it uses are the tested ones, the geometry equivalence is tested on the CPU
# (tests/test_host_logic.py); run in a child process (several Trans objects per part).
@pytest.mark.parametrize("gridname,T,nf,nparts", [("O64", 63, 3, 2), ("O64", 63, 5, 3), ("F32", 31, 4, 4),
                                                  ("O160", 159, 9, 8)])
def test_mirror_band_sharding_reproduces_single_device_result(gridname, T, nf, nparts):
    """every part transforms a northern band of rows and its mirror image; the rows must equal those of the
    single-device transform bit for bit (same arithmetic per (m, latitude) and per row), scalar and vor/div paths"""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mirror_check.py")
    r = subprocess.run([sys.executable, script, gridname, str(T), str(nf), str(nparts)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "MIRROR OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("gridname,nparts,halo", [("O16", 2, 2), ("O16", 3, 1)])
def test_mirror_band_columns_halo_exchange_emulated(gridname, nparts, halo):
    """function space of the mirror-band decomposition (two row ranges per part): after the halo exchange (device pack /
    unpack, device copies for the transport) every node of every part holds its global index"""
    import torch
    from atlas_amd.functionspace import MirrorBandColumns
    from atlas_amd.parallel import HaloExchange
    from test_gpu_halo import exchange_emulated
    g = atlas_amd.Grid(gridname)
    fss = [MirrorBandColumns(g, halo=halo, nparts=nparts, part=p) for p in range(nparts)]
    hxs = [f.begin_halo_exchange() for f in fss]
    HaloExchange.finish_emulated(hxs)
    fields = []
    for f in fss:
        a = np.where(f.ghost() == 0, f.global_index(), -1).astype(np.int64)
        fields.append(torch.from_numpy(np.repeat(a[:, None], 3, axis=1).copy()).cuda())
    exchange_emulated(hxs, fields)
    for f, a in zip(fss, fields):
        assert np.array_equal(a.cpu().numpy()[:, 0], f.global_index())
        assert np.array_equal(a.cpu().numpy()[:, 2], f.global_index())


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int32, np.int64])
@pytest.mark.parametrize("case", ["rank0_arrview", "rank1", "rank1_strided_v1", "rank1_strided_v2", "rank2", "rank2_l1",
                                  "rank2_l2_v2", "rank2_v2", "rank0_wrap", "rank1_paralleldim1", "rank2_paralleldim2"])
def test_reference_adjoint_fixture_on_device(case, dtype):
    """execute_adjoint on the device (pack_adjoint / unpack_adjoint / zero_halos kernels on strided views) against every
    expected array of src/tests/parallel/test_haloexchange_adjoint.cc (tests/golden/halo_adjoint_fixture.json)"""
    import torch
    from test_gpu_halo import exchange_emulated, fixture_objs
    from test_oracle_halo import FIX, adjoint_expected, make_adjoint_fields
    objs = fixture_objs()
    full, views, pdim = make_adjoint_fields(case, dtype)
    dev_full = [torch.from_numpy(a).cuda() for a in full]
    dev_views = []
    for a, v, d in zip(full, views, dev_full):
        off = (v.__array_interface__["data"][0] - a.__array_interface__["data"][0]) // a.itemsize
        dev_views.append(torch.as_strided(d, v.shape, [s // a.itemsize for s in v.strides], off))
    exchange_emulated(objs, dev_views, pdim, adjoint=True)
    checked = 0
    for r in range(FIX["nranks"]):
        want = adjoint_expected(case, r, full[r].size)
        if want is not None:
            assert dev_full[r].cpu().numpy().ravel().tolist() == want, (case, r)
            checked += 1
    assert checked >= 2
