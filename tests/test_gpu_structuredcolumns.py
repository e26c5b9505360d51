"""GPU tests of the StructuredColumns halo exchange (index construction on the host, pack/unpack kernels on the
device): after an exchange every halo point carries its owner's value (what test_structuredcolumns.cc:278-318 checks
with microdeg(x)), for one partition and for band partitions emulated on one device; vector fields change sign in the
halo rows beyond the poles (StructuredColumns.cc:732-808)."""
import numpy as np
import pytest

import atlas_amd
from atlas_amd.functionspace import StructuredColumns
from atlas_amd.parallel import HaloExchange
from test_gpu_halo import exchange_emulated

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def microdeg(x):
    return np.round(np.asarray(x) * 1e6).astype(np.int64)


@pytest.mark.parametrize("gridname,halo,levels", [("O8", 2, 10), ("O32", 1, 137), ("F16", 3, 1)])
def test_single_partition_halo_exchange(gridname, halo, levels):
    g = atlas_amd.Grid(gridname)
    fs = StructuredColumns(g, halo=halo, periodic_points=True)
    n = fs.sizeHalo()
    glb = fs.global_index()
    # value = microdeg(longitude of the OWNING point), as in the reference test; halo entries start as garbage
    off = np.concatenate([[0], np.cumsum(g.nx())])
    rows = np.searchsorted(off, glb - 1, side="right") - 1
    x_owner = (glb - 1 - off[rows]) * (360.0 / g.nx()[rows])
    want = np.repeat(microdeg(x_owner)[:, None], levels, axis=1)
    field = want.copy()
    field[fs.ghost() == 1] = -777
    d = torch.from_numpy(field).cuda()
    fs.haloExchange(d)
    fs.halo_exchange().synchronize()
    assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.parametrize("nparts,dist", [(3, "equal_bands"), (4, "equal_bands")])
def test_band_partitions_emulated(nparts, dist):
    g = atlas_amd.Grid("O16")
    fss = [StructuredColumns(g, halo=2, periodic_points=True, nparts=nparts, part=p, distribution=dist)
           for p in range(nparts)]
    hxs = [f.begin_halo_exchange() for f in fss]
    HaloExchange.finish_emulated(hxs)
    fields = []
    for f in fss:
        a = np.where(f.ghost() == 0, f.global_index(), -1).astype(np.int64)
        fields.append(torch.from_numpy(np.repeat(a[:, None], 5, axis=1).copy()).cuda())
    exchange_emulated(hxs, fields)
    for f, a in zip(fss, fields):
        assert np.array_equal(a.cpu().numpy()[:, 0], f.global_index())
        assert np.array_equal(a.cpu().numpy()[:, 4], f.global_index())


def test_vector_fields_flip_sign_beyond_the_poles():
    g = atlas_amd.Grid("O8")
    fs = StructuredColumns(g, halo=2, periodic_points=False)
    n, lev = fs.sizeHalo(), 3
    own = fs.global_index().astype(np.float64)
    a = np.zeros((n, lev, 3))
    for v in range(3):
        a[:, :, v] = (own * (v + 1))[:, None]
    a[fs.ghost() == 1] = 0.0
    d = torch.from_numpy(a.copy()).cuda()
    fs.haloExchange(d, vector=True)
    fs.halo_exchange().synchronize()
    got = d.cpu().numpy()
    want = np.zeros_like(a)
    for v in range(3):
        want[:, :, v] = (own * (v + 1))[:, None]
    pole = fs.pole_row_nodes()
    assert len(pole) > 0
    want[pole, :, 0] *= -1
    want[pole, :, 1] *= -1            # XX and YY only; further components untouched
    assert np.array_equal(got, want)


def test_equal_regions_like_distribution_emulated():
    """explicit grid::Distribution (latitude bands cut into longitude sectors, as equal_regions produces): after the halo
    exchange every node of every part holds its global index (device pack/unpack, device copies for the transport)."""
    from test_host_structuredcolumns import sector_distribution
    g = atlas_amd.Grid("O16")
    ny = g.ny()
    dist, nparts = sector_distribution(g, [(0, 5), (5, 14), (14, 21), (21, 27), (27, ny)], [1, 3, 4, 3, 1])
    fss = [StructuredColumns(g, halo=2, periodic_points=True, nparts=nparts, part=p, distribution=dist)
           for p in range(nparts)]
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


@pytest.mark.parametrize("gridname,nparts,halo", [("O16", 12, 2), ("O32", 7, 1)])
def test_equal_regions_halo_exchange_emulated(gridname, nparts, halo):
    """distribution="equal_regions" (Atlas's default; mirror of EqualRegionsPartitioner): after the exchange every node
    of every part holds its global index"""
    g = atlas_amd.Grid(gridname)
    fss = [StructuredColumns(g, halo=halo, periodic_points=True, nparts=nparts, part=p, distribution="equal_regions")
           for p in range(nparts)]
    assert sum(f.sizeOwned() for f in fss) == g.size()
    hxs = [f.begin_halo_exchange() for f in fss]
    HaloExchange.finish_emulated(hxs)
    fields = []
    for f in fss:
        a = np.where(f.ghost() == 0, f.global_index(), -1).astype(np.int64)
        fields.append(torch.from_numpy(np.repeat(a[:, None], 2, axis=1).copy()).cuda())
    exchange_emulated(hxs, fields)
    for f, a in zip(fss, fields):
        assert np.array_equal(a.cpu().numpy()[:, 1], f.global_index())


def test_full_size_O1280_137_levels_over_eight_parts_emulated():
    """maximum size (BASELINE config C4's grid and level count): O1280 over eight equal-regions parts, halo 3, a
    137-level fp64 field per part (0.92 GB each) whose owned entries hold global index + level/256 and whose halo entries
    hold NaN -- after one exchange (device pack, device copies for the transport, device unpack) every entry of every
    part holds its owner's value, bit for bit; the index construction itself is compared with the oracle at this size
    in test_host_structuredcolumns.py."""
    g = atlas_amd.Grid("O1280")
    nparts, lev = 8, 137
    fss = [StructuredColumns(g, halo=3, periodic_points=True, nparts=nparts, part=p, distribution="equal_regions")
           for p in range(nparts)]
    assert sum(f.sizeOwned() for f in fss) == g.size()
    hxs = [f.begin_halo_exchange() for f in fss]
    HaloExchange.finish_emulated(hxs)
    levs = torch.arange(lev, dtype=torch.float64, device="cuda") / 256.0
    fields, wants = [], []
    for f in fss:
        gi = torch.from_numpy(f.global_index().astype(np.float64)).cuda()
        want = gi[:, None] + levs[None, :]
        a = want.clone()
        a[torch.from_numpy(f.ghost() == 1).cuda()] = float("nan")
        fields.append(a)
        wants.append(want)
    exchange_emulated(hxs, fields)
    for a, want in zip(fields, wants):
        assert torch.equal(a, want)
