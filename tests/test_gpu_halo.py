"""GPU parity tests of HaloExchange (pack / unpack / adjoint HIP kernels behind the C ABI): bit-exact against the
reference's 3-rank fixture (tests/golden/halo_fixture.json, from src/tests/parallel/test_haloexchange.cc) with the
three ranks emulated on one device, and against the oracle on random decompositions, every supported dtype / rank /
parallel dimension / stride pattern."""
import json
import os

import numpy as np
import pytest

from atlas_amd.parallel import HaloExchange
from oracle.halo import HaloExchangeOracle
from test_host_halo import random_decomposition
from test_oracle_halo import FIX, make_fields

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def exchange_emulated(objs, dev_views, pdim=0, adjoint=False):
    """pack on every 'rank', move the per-peer segments (device copies stand in for RCCL send/recv), unpack"""
    n = len(objs)
    plans = [o.plan() for o in objs]
    vs = [o.var_size(v, pdim) for o, v in zip(objs, dev_views)]
    key_out, key_in = ("recv", "send") if adjoint else ("send", "recv")
    outb = [torch.zeros(int(plans[r][key_out + "counts"].sum()) * vs[r], dtype=dev_views[r].dtype, device="cuda")
            for r in range(n)]
    inb = [torch.zeros(int(plans[r][key_in + "counts"].sum()) * vs[r], dtype=dev_views[r].dtype, device="cuda")
           for r in range(n)]
    for r in range(n):
        (objs[r].pack_adjoint if adjoint else objs[r].pack)(dev_views[r], outb[r], pdim)
        objs[r].synchronize()
    for r in range(n):
        for p in range(n):
            c = int(plans[r][key_in + "counts"][p])
            if c:
                d0 = int(plans[r][key_in + "displs"][p]) * vs[r]
                s0 = int(plans[p][key_out + "displs"][r]) * vs[r]
                inb[r][d0:d0 + c * vs[r]] = outb[p][s0:s0 + c * vs[r]]
    torch.cuda.synchronize()
    for r in range(n):
        if adjoint:
            objs[r].unpack_adjoint(inb[r], dev_views[r], pdim)
            objs[r].zero_halos(dev_views[r], pdim)
        else:
            objs[r].unpack(inb[r], dev_views[r], pdim)
        objs[r].synchronize()


def fixture_objs(on_device=False):
    n = FIX["nranks"]
    objs = [HaloExchange() for _ in range(n)]
    for r, o in enumerate(objs):
        o.setup_emulated(n, r, np.array(FIX["part"][r]), np.array(FIX["ridx"][r]), 0, FIX["nb_nodes"][r],
                         on_device=on_device)
    HaloExchange.finish_emulated(objs)
    return objs


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int32, np.int64])
@pytest.mark.parametrize("case", sorted(FIX["cases"].keys()))
def test_reference_fixture(case, dtype):
    objs = fixture_objs()
    full, views, pdim = make_fields(case, dtype)
    dev_full = [torch.from_numpy(a).cuda() for a in full]
    # re-create the same views on the device tensors
    dev_views = []
    for a, v, d in zip(full, views, dev_full):
        off = (v.__array_interface__["data"][0] - a.__array_interface__["data"][0]) // a.itemsize
        dev_views.append(torch.as_strided(d, v.shape, [s // a.itemsize for s in v.strides], off))
    exchange_emulated(objs, dev_views, pdim)
    for r in range(FIX["nranks"]):
        assert dev_full[r].cpu().numpy().ravel().tolist() == FIX["cases"][case]["expected"][r], (case, r)


def test_device_ballot_compaction_setup_equals_host_setup():
    rng = np.random.default_rng(5)
    parts, ridxs, sizes = random_decomposition(rng, 4, 3000, 777)
    host = [HaloExchange() for _ in range(4)]
    devc = [HaloExchange() for _ in range(4)]
    for r in range(4):
        host[r].setup_emulated(4, r, parts[r], ridxs[r], 0, sizes[r])
        devc[r].setup_emulated(4, r, parts[r], ridxs[r], 0, sizes[r], on_device=True)
    HaloExchange.finish_emulated(host)
    HaloExchange.finish_emulated(devc)
    for a, b in zip(host, devc):
        pa, pb = a.plan(), b.plan()
        for k in pa:
            assert pa[k].tolist() == pb[k].tolist(), k


@pytest.mark.parametrize("shape_tail,pdim", [((), 0), ((137,), 0), ((3,), 0), ((5, 7), 0), ((2, 3, 4), 0),
                                             ((137,), 1), ((4, 6), 1), ((4, 6), 2)])
def test_random_decomposition_against_oracle(shape_tail, pdim):
    rng = np.random.default_rng(len(shape_tail) * 7 + pdim)
    nproc = 3
    parts, ridxs, sizes = random_decomposition(rng, nproc, 500, 123)
    objs = [HaloExchange() for _ in range(nproc)]
    for r, o in enumerate(objs):
        o.setup_emulated(nproc, r, parts[r], ridxs[r], 0, sizes[r])
    HaloExchange.finish_emulated(objs)
    ranks = [HaloExchangeOracle(r, nproc) for r in range(nproc)]
    HaloExchangeOracle.setup(ranks, parts, ridxs, 0, sizes)
    host, dev = [], []
    for r in range(nproc):
        shape = list(shape_tail)
        shape.insert(pdim, sizes[r])
        a = rng.standard_normal(shape)
        host.append(a)
        dev.append(torch.from_numpy(a.copy()).cuda())
    HaloExchangeOracle.execute(ranks, host, pdim)
    exchange_emulated(objs, dev, pdim)
    for r in range(nproc):
        assert np.array_equal(dev[r].cpu().numpy(), host[r])
    # adjoint, same decomposition
    HaloExchangeOracle.execute_adjoint(ranks, host, pdim)
    exchange_emulated(objs, dev, pdim, adjoint=True)
    for r in range(nproc):
        assert np.array_equal(dev[r].cpu().numpy(), host[r])


def test_single_process_device_and_host_entry_points():
    n = 1000
    rng = np.random.default_rng(3)
    part = np.zeros(n, dtype=np.int32)
    ridx = np.arange(n, dtype=np.int32)
    ridx[900:] = rng.integers(0, 900, 100)
    hx = HaloExchange()
    hx.setup(part, ridx, 0, n)
    orc = [HaloExchangeOracle(0, 1)]
    HaloExchangeOracle.setup(orc, [part], [ridx], 0, [n])
    a = rng.standard_normal((n, 10, 3))
    ref = a.copy()
    HaloExchangeOracle.execute(orc, [ref])
    d = torch.from_numpy(a.copy()).cuda()
    hx.execute(d); hx.synchronize()
    assert np.array_equal(d.cpu().numpy(), ref)
    h = a.copy()
    hx.execute(h)                                    # host array: staged through the device
    assert np.array_equal(h, ref)
    # C-interface form (atlas__HaloExchange__execute_strided_double): strides / shape of the non-parallel dims
    c = a.copy()
    hx.execute_strided(c, [3, 1], [10, 3])
    assert np.array_equal(c, ref)
    # adjoint through the C interface
    ref_adj = a.copy()
    HaloExchangeOracle.execute_adjoint(orc, [ref_adj])
    c = a.copy()
    hx.execute_strided(c, [3, 1], [10, 3], adjoint=True)
    assert np.array_equal(c, ref_adj)


def test_adjoint_dot_product_identity_on_device():
    rng = np.random.default_rng(9)
    nproc = 4
    parts, ridxs, sizes = random_decomposition(rng, nproc, 300, 90)
    objs = [HaloExchange() for _ in range(nproc)]
    for r, o in enumerate(objs):
        o.setup_emulated(nproc, r, parts[r], ridxs[r], 0, sizes[r])
    HaloExchange.finish_emulated(objs)
    x = [rng.standard_normal((s, 4)) for s in sizes]
    y = [rng.standard_normal((s, 4)) for s in sizes]
    for r in range(nproc):
        x[r][objs[r].plan()["recvmap"]] = 0
    hx = [torch.from_numpy(a.copy()).cuda() for a in x]
    hty = [torch.from_numpy(a.copy()).cuda() for a in y]
    exchange_emulated(objs, hx)
    exchange_emulated(objs, hty, adjoint=True)
    lhs = sum(float((a.cpu().numpy() * b).sum()) for a, b in zip(hx, y))
    rhs = sum(float((a * b.cpu().numpy()).sum()) for a, b in zip(x, hty))
    assert abs(lhs - rhs) < 1e-11 * max(1.0, abs(lhs))


def test_nodecolumns_halo_exchange_mirror():
    """functionspace::NodeColumns::haloExchange (NodeColumns.cc:101-113,357-459): setup from (partition, remote_index,
    REMOTE_IDX_BASE, nb_nodes) without halo_begin, Field or FieldSet, rank 1..4, the four datatypes; the reference's
    3-rank fixture must come out as in test_haloexchange.cc."""
    from atlas_amd.functionspace import NodeColumns
    n = FIX["nranks"]
    ncs = [NodeColumns(FIX["part"][r], FIX["ridx"][r], nb_nodes=FIX["nb_nodes"][r], emulate=(n, r)) for r in range(n)]
    HaloExchange.finish_emulated([nc.halo_exchange() for nc in ncs])
    case = sorted(FIX["cases"].keys())[0]
    full, views, pdim = make_fields(case, np.float64)
    if pdim == 0 and all(v.flags.c_contiguous for v in views):
        dev = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in views]
        exchange_emulated([nc.halo_exchange() for nc in ncs], dev, 0)
        for r in range(n):
            assert dev[r].cpu().numpy().ravel().tolist() == FIX["cases"][case]["expected"][r]
    # single rank: a periodic ring, node i of the second half is a ghost of node i - m
    m = 50
    part = np.zeros(2 * m, dtype=np.int32)
    ridx = np.concatenate([np.arange(m), np.arange(m)]).astype(np.int32)
    nc = NodeColumns(part, ridx)
    for dtype in (torch.int32, torch.int64, torch.float32, torch.float64):
        for tail in ((), (3,), (2, 3), (2, 3, 4)):
            f = torch.zeros((2 * m,) + tail, dtype=dtype, device="cuda")
            f[:m] = (torch.arange(m, device="cuda").reshape((m,) + (1,) * len(tail)) + 1).to(dtype)
            nc.haloExchange(f)
            nc.halo_exchange().synchronize()
            assert torch.equal(f[m:], f[:m])
    fs = [torch.zeros(2 * m, dtype=torch.float64, device="cuda") for _ in range(2)]
    for i, f in enumerate(fs):
        f[:m] = i + 1.0
    nc.haloExchange(fs)                                            # FieldSet
    nc.halo_exchange().synchronize()
    assert all(torch.equal(f[m:], f[:m]) for f in fs)
    with pytest.raises(TypeError, match="datatype not supported"):
        nc.haloExchange(torch.zeros(2 * m, dtype=torch.float16, device="cuda"))
    with pytest.raises(ValueError, match="Rank not supported"):
        nc.haloExchange(torch.zeros((2 * m, 1, 1, 1, 1), dtype=torch.float64, device="cuda"))
    a = torch.ones(2 * m, dtype=torch.float64, device="cuda")
    nc.adjointHaloExchange(a)                                      # owned += ghost contribution, ghosts zeroed
    nc.halo_exchange().synchronize()
    assert torch.equal(a[:m], torch.full((m,), 2.0, device="cuda", dtype=torch.float64)) and float(a[m:].abs().sum()) == 0.0
