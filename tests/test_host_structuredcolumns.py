"""CPU tests: the product's StructuredColumns host logic is identical to the oracle's (all index fields, bitwise xy)
and reproduces the sizes asserted by the reference tests."""
import numpy as np
import pytest

import atlas_amd
from atlas_amd.functionspace import StructuredColumns
from oracle.structured_columns import StructuredColumnsOracle


def lonlat_grid(nxl, nyl):
    y = np.array([90.0 - j * 180.0 / (nyl - 1) for j in range(nyl)])
    return atlas_amd.StructuredGrid(nx=np.full(nyl, nxl), y=y)


def compare(fs, orc):
    assert (fs.sizeOwned(), fs.sizeHalo()) == (orc.size_owned, orc.size_halo)
    assert np.array_equal(fs.partition(), orc.partition_f)
    assert np.array_equal(fs.ghost(), orc.ghost)
    assert np.array_equal(fs.global_index(), orc.glb_idx)
    assert np.array_equal(fs.index_i(), orc.index_i) and np.array_equal(fs.index_j(), orc.index_j)
    assert np.array_equal(fs.remote_index(), orc.remote_idx)
    assert np.array_equal(fs.xy(), orc.xy)
    assert np.array_equal(fs.pole_row_nodes(), orc.pole_rows_nodes())
    assert (fs.j_begin(), fs.j_end(), fs.j_begin_halo(), fs.j_end_halo()) == \
        (orc.j_begin, orc.j_end, orc.j_begin_halo, orc.j_end_halo)
    for j in range(orc.j_begin_halo, orc.j_end_halo):
        assert (fs.i_begin_halo(j), fs.i_end_halo(j)) == (orc.i_begin_halo[j], orc.i_end_halo[j])
        for i in (orc.i_begin_halo[j], orc.i_end_halo[j] - 1):
            assert fs.index(i, j) == orc.index(i, j)


def test_reference_sizes():
    # src/tests/functionspace/test_structuredcolumns_haloexchange.cc:70-75, 101-112
    g = lonlat_grid(400, 200)
    assert StructuredColumns(g, halo=1, periodic_points=True).sizeHalo() == 81406
    assert StructuredColumns(g, halo=1).sizeHalo() == 81204
    g = lonlat_grid(400, 201)
    assert StructuredColumns(g, halo=1).sizeHalo() == 81606
    owned, halo = [20400, 20000, 20000, 20000], [21306, 20904, 20904, 20904]
    for r in range(4):
        fs = StructuredColumns(g, halo=1, nparts=4, part=r, distribution="regular_bands")
        assert (fs.sizeOwned(), fs.sizeHalo()) == (owned[r], halo[r])


@pytest.mark.parametrize("gridname,halo,pp,nparts", [("O8", 2, True, 1), ("O8", 1, False, 3), ("O16", 3, True, 4),
                                                     ("F8", 2, True, 2), ("O32", 1, False, 8)])
def test_identical_to_oracle_gaussian(gridname, halo, pp, nparts):
    g = atlas_amd.Grid(gridname)
    for part in range(nparts):
        fs = StructuredColumns(g, halo=halo, periodic_points=pp, nparts=nparts, part=part)
        orc = StructuredColumnsOracle(g.nx(), g.y(), halo=halo, periodic_points=pp, nparts=nparts, part=part)
        compare(fs, orc)


def test_identical_to_oracle_lonlat_with_poles():
    g = lonlat_grid(36, 19)
    for nparts, bs in ((1, 1), (3, 36)):
        for part in range(nparts):
            fs = StructuredColumns(g, halo=2, periodic_points=True, nparts=nparts, part=part,
                                   distribution="regular_bands" if bs > 1 else "equal_bands")
            orc = StructuredColumnsOracle(g.nx(), g.y(), halo=2, periodic_points=True, nparts=nparts, part=part,
                                          blocksize=bs)
            compare(fs, orc)


def test_row_bands_distribution_matches_oracle_and_the_transform_bands():
    """distribution="row_bands": whole rows, each with the equal_bands part of its first point -- the latitude bands of
    the multi-GPU transform (atlas_amd/csrc/trans_plan.cpp: latitude_bands)."""
    g = atlas_amd.Grid("O16")
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for nparts in (2, 3, 5):
        owned = []
        for part in range(nparts):
            fs = StructuredColumns(g, halo=2, periodic_points=True, nparts=nparts, part=part, distribution="row_bands")
            orc = StructuredColumnsOracle(g.nx(), g.y(), halo=2, periodic_points=True, nparts=nparts, part=part,
                                          blocksize=0)
            compare(fs, orc)
            gi = fs.global_index()[:fs.sizeOwned()] - 1
            rows = np.searchsorted(off, gi, side="right") - 1
            # whole rows, contiguous, in global order
            assert np.array_equal(gi, np.arange(off[rows.min()], off[rows.max() + 1]))
            assert all(((off[j] * nparts) // off[-1]) == part for j in np.unique(rows))
            owned.append(gi)
        assert np.array_equal(np.concatenate(owned), np.arange(off[-1]))


def sector_distribution(g, bands, sectors):
    """equal-regions-like distribution: latitude bands (row ranges), band b cut into sectors[b] longitude sectors"""
    nx, part, p0 = g.nx(), [], 0
    for (j0, j1), ns in zip(bands, sectors):
        for j in range(j0, j1):
            part.append(p0 + (np.arange(nx[j]) * ns) // nx[j])
        p0 += ns
    return np.concatenate(part).astype(np.int32), p0


def test_explicit_distribution_matches_oracle():
    """an explicit grid::Distribution, shaped like equal_regions output (polar caps + collars of several sectors):
    every index array of every part equals the oracle's, and the parts tile the grid."""
    g = atlas_amd.Grid("O16")
    ny = g.ny()
    dist, nparts = sector_distribution(g, [(0, 5), (5, 14), (14, 21), (21, 27), (27, ny)], [1, 3, 4, 3, 1])
    seen = []
    for part in range(nparts):
        fs = StructuredColumns(g, halo=2, periodic_points=True, nparts=nparts, part=part, distribution=dist)
        orc = StructuredColumnsOracle(g.nx(), g.y(), halo=2, periodic_points=True, nparts=nparts, part=part,
                                      distribution=dist)
        compare(fs, orc)
        gi = fs.global_index()[:fs.sizeOwned()] - 1
        assert np.all(dist[gi] == part)
        seen.append(gi)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(g.size()))
    # remote index really is the owner's local index of the same global point (what the reference learns from the owner)
    fss = [StructuredColumns(g, halo=2, periodic_points=True, nparts=nparts, part=p, distribution=dist)
           for p in range(nparts)]
    for fs in fss:
        part, ridx, gi = fs.partition(), fs.remote_index(), fs.global_index()
        for n in range(fs.sizeOwned(), fs.sizeHalo()):
            assert fss[part[n]].global_index()[ridx[n]] == gi[n]
    with pytest.raises(Exception):
        StructuredColumns(g, halo=1, nparts=nparts, part=0, distribution=dist[:-1])


@pytest.mark.parametrize("gridname,halo,nparts", [("O16", 2, 1), ("O16", 1, 2), ("O16", 2, 3), ("F16", 1, 4)])
def test_mirror_band_columns_indices_are_consistent(gridname, halo, nparts):
    """MirrorBandColumns (function space of Trans(shard="mirror"): two row ranges per part, composed of two
    StructuredColumns blocks): the owned points are the part's points in transform-output order, and every local point's
    (partition, remote_index) names the same global point in its owner's numbering -- what HaloExchange::setup needs"""
    from atlas_amd.functionspace import MirrorBandColumns, mirror_band_distribution
    g = atlas_amd.Grid(gridname)
    dist, b = mirror_band_distribution(g, nparts)
    fss = [MirrorBandColumns(g, halo=halo, nparts=nparts, part=p) for p in range(nparts)]
    assert sum(fs.sizeOwned() for fs in fss) == g.size()
    off = np.concatenate([[0], np.cumsum(g.nx())])
    for p, fs in enumerate(fss):
        gi, part, ridx = fs.global_index(), fs.partition(), fs.remote_index()
        own = gi[:fs.sizeOwned()] - 1
        rows = np.concatenate([np.arange(b[p], b[p + 1]), np.arange(g.ny() - b[p + 1], g.ny() - b[p])])
        assert np.array_equal(own, np.concatenate([np.arange(off[j], off[j + 1]) for j in rows]))   # invtrans order
        assert np.all(dist[own] == p)
        assert np.all(part[:fs.sizeOwned()] == p) and np.array_equal(ridx[:fs.sizeOwned()], np.arange(fs.sizeOwned()))
        assert fs.sizeHalo() > fs.sizeOwned() and np.all(fs.ghost()[fs.sizeOwned():] == 1)
        for n in range(fs.sizeOwned(), fs.sizeHalo()):
            assert ridx[n] < fss[part[n]].sizeOwned() and fss[part[n]].global_index()[ridx[n]] == gi[n]


@pytest.mark.parametrize("gridname", ["L40x21", "L40x20", "Slat100x50"])
@pytest.mark.parametrize("nparts", [1, 2, 3, 4, 5])
def test_regular_bands_properties_of_the_reference_test(gridname, nparts):
    """src/tests/grid/test_distribution_regular_bands.cc:60-140 (CASE test_regular_bands): with the regular_bands
    distribution every latitude lies in one partition, the partitions count all points once, and a StructuredColumns
    with halo 1 and periodic points has i_begin_halo = -1, i_end_halo = nx + 2 on every row of its halo range"""
    kind, dims = gridname[:-len(gridname.lstrip("LSlat"))], gridname.lstrip("LSlat")
    nx, ny = (int(v) for v in dims.split("x"))
    if kind == "L":
        y = np.array([90.0 - j * 180.0 / (ny - 1) for j in range(ny)])          # poles included
    else:
        y = np.array([90.0 - (j + 0.5) * 180.0 / ny for j in range(ny)])        # shifted latitudes
    g = atlas_amd.StructuredGrid(nx=np.full(ny, nx), y=y)
    counted = 0
    for part in range(nparts):
        fs = StructuredColumns(g, halo=1, periodic_points=True, nparts=nparts, part=part, distribution="regular_bands")
        owned_rows = fs.index_j()[:fs.sizeOwned()]
        assert fs.sizeOwned() == nx * len(np.unique(owned_rows))               # whole latitudes only
        assert np.all(fs.partition()[:fs.sizeOwned()] == part)
        counted += fs.sizeOwned()
        for j in range(fs.j_begin_halo(), fs.j_end_halo()):
            assert (fs.i_begin_halo(j), fs.i_end_halo(j)) == (-1, nx + 2), (part, j)
    assert counted == nx * ny


def test_distributions_the_construction_cannot_describe_are_rejected():
    """the owned region is one row range with one i-range per row (StructuredColumns_setup.cc:125-226): a part with two
    separate row ranges, or with a hole inside a row, is refused (product and oracle) instead of building halos around
    empty rows"""
    g = atlas_amd.Grid("O16")
    off = np.concatenate([[0], np.cumsum(g.nx())])
    two_ranges = np.zeros(g.size(), dtype=np.int32)
    two_ranges[off[4]:off[28]] = 1                       # part 0: rows 0-3 and 28-31
    hole = np.zeros(g.size(), dtype=np.int32)
    hole[off[10] + 5:off[10] + 9] = 1                    # part 0: a hole in row 10
    for dist in (two_ranges, hole):
        with pytest.raises(Exception, match="contiguous"):
            StructuredColumns(g, halo=1, nparts=2, part=0, distribution=dist)
        with pytest.raises(ValueError):
            StructuredColumnsOracle(g.nx(), g.y(), halo=1, nparts=2, part=0, distribution=dist)
        fs = StructuredColumns(g, halo=1, nparts=2, part=1, distribution=dist)     # the other part is fine
        assert fs.sizeOwned() == int((dist == 1).sum())


def test_equal_regions_partitioner_goldens_and_structuredcolumns():
    """eq_caps / EqualRegionsPartitioner against the reference's expected values (src/tests/mesh/test_rgg.cc:103-165), and
    StructuredColumns built on its output against the oracle (explicit grid::Distribution)."""
    from atlas_amd.partitioner import EqualRegionsPartitioner, eq_caps
    assert eq_caps(6)[0] == [1, 4, 1] and eq_caps(10)[0] == [1, 4, 4, 1]
    for N, want in ((12, [1, 5, 5, 1]), (24, [1, 6, 10, 6, 1]), (48, [1, 6, 11, 12, 11, 6, 1]),
                    (96, [1, 6, 11, 14, 16, 16, 14, 11, 6, 1])):
        p = EqualRegionsPartitioner(N)
        assert p.nb_bands() == len(want) and [p.nb_regions(b) for b in range(p.nb_bands())] == want
    g = atlas_amd.Grid("O16")
    for N in (1, 2, 5, 12):
        part = EqualRegionsPartitioner(N).partition(g)
        cnt = np.bincount(part, minlength=N)
        assert cnt.sum() == g.size() and cnt.max() - cnt.min() <= 1      # chunks of npts/N (+1)
        off = np.concatenate([[0], np.cumsum(g.nx())])
        for j in range(g.ny()):                                           # one contiguous i-range per part and row
            row = part[off[j]:off[j + 1]]
            for q in np.unique(row):
                idx = np.nonzero(row == q)[0]
                assert idx.max() - idx.min() + 1 == idx.size
    part = EqualRegionsPartitioner(12).partition(g)
    fss = []
    for p in range(12):
        fs = StructuredColumns(g, halo=1, periodic_points=True, nparts=12, part=p, distribution=part)
        compare(fs, StructuredColumnsOracle(g.nx(), g.y(), halo=1, periodic_points=True, nparts=12, part=p,
                                            distribution=part))
        fss.append(fs)
    for fs in fss:
        pp, ridx, gi = fs.partition(), fs.remote_index(), fs.global_index()
        for n in range(fs.sizeOwned(), fs.sizeHalo()):
            assert fss[pp[n]].global_index()[ridx[n]] == gi[n]


def test_eq_caps_of_the_library_and_the_oracle_agree_for_every_count_incl_the_ties():
    """eq_caps rounds the zones' ideal shares of the N regions (round_to_naturals, EqualRegionsPartitioner.cc:229-245); where a share
    is x.5 up to the last bit (N = 9: 3.5 + 3.5; N = 31) the result hangs on the order of the floating-point operations in
    area_of_cap (:125).  Round 5 found the library forming (4 pi s) s instead of 4 pi (s s): N = 9 gave [1, 4, 3, 1].  Library
    (csrc/equal_regions.cpp) and oracle (oracle/partitioner.py) are two restatements of the same source lines: equal for every N,
    region counts and cap colatitudes to the bit."""
    from atlas_amd.partitioner import eq_caps
    from oracle.partitioner import eq_caps as eq_caps_oracle
    assert eq_caps(9)[0] == [1, 3, 4, 1] and eq_caps(31)[0] == [1, 6, 8, 9, 6, 1]
    for N in list(range(1, 1025)) + [1280, 2047, 4096, 10000]:
        a, b = eq_caps(N), eq_caps_oracle(N)
        assert a[0] == list(b[0]) and a[1] == list(b[1]), N
        assert sum(a[0]) == N


def test_equal_regions_partition_of_O8_over_five_parts_is_the_reference_vector():
    """the expected array of src/tests/functionspace/test_structuredcolumns.cc:87-106 (partition of every point of O8 with
    the default partitioner on 5 MPI tasks; tests/golden/equal_regions_O8_5.json), from the library (C++) and from the
    numpy restatement in oracle/; StructuredColumns built on it owns exactly those points"""
    import json
    import os
    from atlas_amd.partitioner import EqualRegionsPartitioner
    from oracle.partitioner import EqualRegionsPartitioner as OraclePartitioner
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "equal_regions_O8_5.json")))
    g = atlas_amd.Grid(fix["grid"])
    want = np.asarray(fix["partition"], dtype=np.int32)
    assert want.size == g.size() == 544
    got = EqualRegionsPartitioner(fix["nparts"]).partition(g)
    assert np.array_equal(got, want)
    assert np.array_equal(OraclePartitioner(fix["nparts"]).partition(g), want)
    for N in (2, 3, 7, 12, 40):
        assert np.array_equal(EqualRegionsPartitioner(N).partition(g), OraclePartitioner(N).partition(g))
    for p in range(fix["nparts"]):
        fs = StructuredColumns(g, halo=0, nparts=fix["nparts"], part=p, distribution="equal_regions")
        assert fs.sizeOwned() == int((want == p).sum())
        assert np.array_equal(np.sort(fs.global_index()[:fs.sizeOwned()]) - 1, np.nonzero(want == p)[0])


def test_full_size_O1280_over_eight_equal_regions_parts_matches_oracle():
    """maximum size of BASELINE config C4: O1280 (6 599 680 points) over eight equal-regions parts with halo 3 -- the
    partitioner's output for 8 and 64 parts, and the whole index construction of a polar-cap part and of a collar part,
    equal the oracle's element for element; the eight parts own the grid exactly once."""
    from atlas_amd.partitioner import EqualRegionsPartitioner
    from oracle.partitioner import EqualRegionsPartitioner as OraclePartitioner
    g = atlas_amd.Grid("O1280")
    for N in (8, 64):
        part = EqualRegionsPartitioner(N).partition(g)
        assert np.array_equal(part, OraclePartitioner(N).partition(g))
        cnt = np.bincount(part, minlength=N)
        assert cnt.sum() == g.size() and cnt.max() - cnt.min() <= 1
    part = EqualRegionsPartitioner(8).partition(g)
    owned = 0
    for p in range(8):
        fs = StructuredColumns(g, halo=3, periodic_points=True, nparts=8, part=p, distribution="equal_regions")
        owned += fs.sizeOwned()
        if p in (0, 3):
            compare(fs, StructuredColumnsOracle(g.nx(), g.y(), halo=3, periodic_points=True, nparts=8, part=p,
                                                distribution=part))
    assert owned == g.size()
