"""CPU tests of the product's HaloExchange HOST logic (setup) against the oracle and the reference fixture."""
import json
import os

import numpy as np
import pytest

from atlas_amd.parallel import HaloExchange
from oracle.halo import HaloExchangeOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "halo_fixture.json")))


def random_decomposition(rng, nproc, nown, nhalo):
    """random parts: each rank owns `nown` nodes and has `nhalo` halo nodes that copy random owned nodes"""
    parts, ridxs, sizes = [], [], []
    for r in range(nproc):
        n = nown + nhalo
        part = np.full(n, r, dtype=np.int32)
        ridx = np.arange(n, dtype=np.int32)
        owner = rng.integers(0, nproc, nhalo)
        part[nown:] = owner
        ridx[nown:] = rng.integers(0, nown, nhalo)
        parts.append(part); ridxs.append(ridx); sizes.append(n)
    return parts, ridxs, sizes


def product_plans(parts, ridxs, base, sizes):
    n = len(parts)
    objs = [HaloExchange() for _ in range(n)]
    for r, o in enumerate(objs):
        o.setup_emulated(n, r, parts[r], ridxs[r] + base, base, sizes[r])
    HaloExchange.finish_emulated(objs)
    return objs


def test_fixture_setup_matches_oracle():
    n = FIX["nranks"]
    objs = product_plans([np.array(p, dtype=np.int32) for p in FIX["part"]],
                         [np.array(p, dtype=np.int32) for p in FIX["ridx"]], 0, FIX["nb_nodes"])
    ranks = [HaloExchangeOracle(r, n) for r in range(n)]
    HaloExchangeOracle.setup(ranks, FIX["part"], FIX["ridx"], FIX["base"], FIX["nb_nodes"])
    for o, k in zip(objs, ranks):
        p = o.plan()
        for key in ("sendcounts", "recvcounts", "senddispls", "recvdispls", "sendmap", "recvmap"):
            assert p[key].tolist() == getattr(k, key).tolist(), key


@pytest.mark.parametrize("nproc,base", [(1, 0), (2, 1), (5, 0), (8, 1)])
def test_random_setup_matches_oracle(nproc, base):
    rng = np.random.default_rng(nproc * 10 + base)
    parts, ridxs, sizes = random_decomposition(rng, nproc, 200, 57)
    objs = product_plans(parts, ridxs, base, sizes)
    ranks = [HaloExchangeOracle(r, nproc) for r in range(nproc)]
    HaloExchangeOracle.setup(ranks, parts, [r + base for r in ridxs], base, sizes)
    for o, k in zip(objs, ranks):
        p = o.plan()
        for key in ("sendcounts", "recvcounts", "sendmap", "recvmap"):
            assert p[key].tolist() == getattr(k, key).tolist(), key


def test_serial_setup_and_halo_begin():
    # one process: periodic duplicates are ghosts of the same rank (HaloExchange.cc:39-46)
    n = 30
    part = np.zeros(n, dtype=np.int32)
    ridx = np.arange(n, dtype=np.int32) + 1  # base 1 (Fortran numbering)
    ridx[25:] = [1, 2, 3, 4, 5]
    hx = HaloExchange()
    hx.setup(part, ridx, 1, n)
    assert hx.recvcnt() == 5 and hx.plan()["recvmap"].tolist() == [25, 26, 27, 28, 29]
    assert hx.plan()["sendmap"].tolist() == [0, 1, 2, 3, 4]
    hx2 = HaloExchange()
    hx2.setup(part, ridx, 1, n, halo_begin=27)   # only nodes >= halo_begin are examined (HaloExchange.cc:102)
    assert hx2.plan()["recvmap"].tolist() == [27, 28, 29]


def test_execute_before_setup_raises():
    from atlas_amd import _lib
    with pytest.raises(_lib.AtlasAmdError, match="not setup"):
        HaloExchange().execute(np.zeros(4))
