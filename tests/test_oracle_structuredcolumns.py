"""Pins the StructuredColumns ORACLE (oracle/structured_columns.py) against the sizes the reference asserts
(src/tests/functionspace/test_structuredcolumns_haloexchange.cc:64-121) and against self-consistency properties
the reference checks (halo values equal the owner's values after an exchange, test_structuredcolumns.cc:278-318)."""
import numpy as np
import pytest

from oracle.halo import HaloExchangeOracle
from oracle.structured_columns import StructuredColumnsOracle, bands_partition


def lonlat(nxl, nyl):
    y = [90.0 - j * 180.0 / (nyl - 1) for j in range(nyl)]
    return [nxl] * nyl, y


def test_reference_sizes_single_partition():
    nx, y = lonlat(400, 200)
    fs1 = StructuredColumnsOracle(nx, y, halo=1, periodic_points=True)
    fs2 = StructuredColumnsOracle(nx, y, halo=1, periodic_points=False)
    assert (fs1.size_owned, fs2.size_owned) == (80000, 80000)
    assert (fs1.size_halo, fs2.size_halo) == (81406, 81204)           # :70-75
    nx, y = lonlat(400, 201)
    fs = StructuredColumnsOracle(nx, y, halo=1)
    assert (fs.size_owned, fs.size_halo) == (80400, 81606)            # :101-106


def test_reference_sizes_regular_bands_4_ranks():
    nx, y = lonlat(400, 201)                                           # :107-112, dist2 = regular_bands
    owned = [20400, 20000, 20000, 20000]
    halo = [21306, 20904, 20904, 20904]
    for r in range(4):
        fs = StructuredColumnsOracle(nx, y, halo=1, nparts=4, part=r, blocksize=400)
        assert (fs.size_owned, fs.size_halo) == (owned[r], halo[r]), r


def test_bands_distribution_properties():
    # src/tests/grid/test_distribution_regular_bands.cc:49-140: monotone in the global index; with blocksize = nx each
    # latitude lies on one partition
    npts, P = 400 * 201, 4
    parts = [bands_partition(g, npts, P, 400) for g in range(npts)]
    assert all(b >= a for a, b in zip(parts, parts[1:])) and parts[0] == 0 and parts[-1] == P - 1
    for j in range(201):
        assert len(set(parts[j * 400:(j + 1) * 400])) == 1


@pytest.mark.parametrize("nparts", [1, 3])
def test_halo_values_equal_owner_values(nparts):
    # O8-like octahedral rows, halo 2, periodic points: exchange a field holding the global index
    N = 8
    nxh = [20 + 4 * j for j in range(N)]
    nx = nxh + nxh[::-1]
    x, _ = np.polynomial.legendre.leggauss(2 * N)
    y = np.degrees(np.arcsin(x[::-1])).tolist()
    fss = [StructuredColumnsOracle(nx, y, halo=2, periodic_points=True, nparts=nparts, part=p) for p in range(nparts)]
    ranks = [HaloExchangeOracle(p, nparts) for p in range(nparts)]
    parts, ridx, sizes, hb = zip(*[(f.partition_f, f.remote_idx, f.size_halo, f.size_owned) for f in fss])
    HaloExchangeOracle.setup(ranks, list(parts), list(ridx), 0, list(sizes), list(hb))
    fields = []
    for f in fss:
        a = np.where(f.ghost == 0, f.glb_idx, -1).astype(np.int64)
        fields.append(a)
    HaloExchangeOracle.execute(ranks, fields)
    for f, a in zip(fss, fields):
        assert np.array_equal(a, f.glb_idx)
