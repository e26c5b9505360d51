"""The distributed paths that run INSIDE the library (csrc/comm.hip, dist_trans.hip, halo_exchange.hip: execute_comm),
driven from Python threads: P emulated ranks on one GPU over the "local" communicator, and one rank over real RCCL.

Oracle: the single-device transform (itself compared with the CPU oracle in test_gpu_trans.py) -- the sharded result must
equal it bit for bit, because every (m, latitude) product and every row goes through the same arithmetic; for the halo
exchange: the owner's value at every halo node (src/tests/functionspace/test_structuredcolumns_haloexchange.cc:38-60)."""
import threading

import numpy as np
import pytest
import torch

import atlas_amd
from atlas_amd.comm import Comm, CommHub
from atlas_amd.dist import DistributedTrans
from atlas_amd.functionspace import StructuredColumns
from atlas_amd.parallel import HaloExchange
from helpers import red_spectra

pytestmark = pytest.mark.gpu


def run_ranks(nparts, fn):
    hub = CommHub(nparts)
    comms = [hub.comm(r) for r in range(nparts)]
    out, err = [None] * nparts, [None] * nparts

    def worker(r):
        try:
            torch.cuda.set_device(0)
            out[r] = fn(comms[r])
        except BaseException as e:  # noqa: BLE001  (reported below)
            err[r] = e
    th = [threading.Thread(target=worker, args=(r,)) for r in range(nparts)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("gridname,T,nf,nparts,maxmsg", [("O64", 63, 3, 2, None), ("O64", 63, 5, 3, 8192),
                                                         ("F32", 31, 4, 4, None), ("O160", 159, 9, 8, 1 << 16),
                                                         ("O320", 319, 70, 2, None)])   # [r6] 9 column tiles: 6 + 3 (mixed tiling)
def test_native_distributed_transform_equals_single_device(gridname, T, nf, nparts, maxmsg):
    g = atlas_amd.Grid(gridname)
    sps = [torch.from_numpy(red_spectra(T, nf, seed=s)).cuda() for s in (1, 2, 3)]
    tr = atlas_amd.Trans(g, T)
    refs = []
    for sp in sps:
        gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp, gp)
        tr.synchronize()
        refs.append(gp.cpu().numpy().reshape(nf, -1))
    off = np.concatenate([[0], np.cumsum(g.nx())])

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        if maxmsg:
            d.set_max_message_bytes(maxmsg)
        n = d.trans.nb_gridpoints()
        gps = [torch.zeros(nf * n, dtype=torch.float64, device="cuda") for _ in sps]
        d.invtrans(nf, sps[0], gps[0])
        d.trans.synchronize()
        one = gps[0].cpu().numpy().copy()
        d.invtrans_many(nf, sps, gps)            # pipelined: exchange of i next to Legendre of i+1 / Fourier of i-1
        d.trans.synchronize()
        return d.bands[comm.rank()], d.bands[comm.rank() + 1], one, [x.cpu().numpy() for x in gps]

    for b0, b1, one, many in run_ranks(nparts, rank):
        sl = slice(off[b0], off[b1])
        assert np.array_equal(one.reshape(nf, -1), refs[0][:, sl])
        for got, ref in zip(many, refs):
            assert np.array_equal(got.reshape(nf, -1), ref[:, sl])


@pytest.mark.parametrize("gridname,T,nf,nparts,maxmsg", [("O64", 63, 3, 2, None), ("O64", 47, 5, 3, 8192), ("O160", 159, 9, 8, 1 << 15)])
def test_transposition_reads_no_uninitialised_bytes(gridname, T, nf, nparts, maxmsg, monkeypatch):
    """[r3] the transposition ships per row only the kept wavenumbers and the live columns (dist_trans.h: packed runs).
    With ATLAS_AMD_DIST_POISON=1 the library fills the intermediate, the packed send buffer and the receive buffer with NaN
    before every transform: anything the Legendre stage did not write that reached the wire or the Fourier stage would show
    up in the grid points, which must still equal the single-device transform bit for bit."""
    monkeypatch.setenv("ATLAS_AMD_DIST_POISON", "1")
    g = atlas_amd.Grid(gridname)
    sps = [torch.from_numpy(red_spectra(T, nf, seed=s)).cuda() for s in (5, 6, 7)]
    tr = atlas_amd.Trans(g, T)
    refs = []
    for sp in sps:
        gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp, gp)
        tr.synchronize()
        refs.append(gp.cpu().numpy().reshape(nf, -1))
    off = np.concatenate([[0], np.cumsum(g.nx())])

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        if maxmsg:
            d.set_max_message_bytes(maxmsg)
        n = d.trans.nb_gridpoints()
        gps = [torch.full((nf * n,), float("nan"), dtype=torch.float64, device="cuda") for _ in sps]
        d.invtrans_many(nf, sps, gps)
        d.trans.synchronize()
        # a limit set after the first transform takes effect (the message list is rebuilt), identically on every rank
        d.set_max_message_bytes(4096)
        again = torch.full((nf * n,), float("nan"), dtype=torch.float64, device="cuda")
        d.invtrans(nf, sps[1], again)
        d.trans.synchronize()
        return d.bands[comm.rank()], d.bands[comm.rank() + 1], [x.cpu().numpy() for x in gps], again.cpu().numpy()

    for b0, b1, many, again in run_ranks(nparts, rank):
        sl = slice(off[b0], off[b1])
        for got, ref in zip(many, refs):
            assert np.isfinite(got).all()
            assert np.array_equal(got.reshape(nf, -1), ref[:, sl])
        assert np.array_equal(again.reshape(nf, -1), refs[1][:, sl])


@pytest.mark.parametrize("gridname,T,nf,nparts", [("O64", 63, 5, 2), ("O160", 159, 41, 3), ("F32", 31, 3, 8)])
def test_distributed_transform_from_spectra_scattered_by_wavenumber(gridname, T, nf, nparts):
    """[r3] SURVEY 8(e) scatters the input by m: every rank passes only the blocks of its own wavenumbers
    (atlas_amd__Trans__invtrans_distributed_sharded, layout from __spectral_shard); the bands must equal the single-device
    transform bit for bit, as with the replicated input"""
    g = atlas_amd.Grid(gridname)
    sps = [red_spectra(T, nf, seed=s) for s in (31, 32)]
    tr = atlas_amd.Trans(g, T)
    refs = []
    for sp in sps:
        gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
        tr.invtrans(nf, torch.from_numpy(sp).cuda(), gp)
        tr.synchronize()
        refs.append(gp.cpu().numpy().reshape(nf, -1))
    off = np.concatenate([[0], np.cumsum(g.nx())])
    total = (T + 1) * (T + 2) * nf

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        moff, size = d.spectral_shard()
        owned = [m for m in range(T + 1) if moff[m] >= 0]
        assert owned == list(range(comm.rank(), T + 1, nparts)) and size == sum(2 * (T + 1 - m) for m in owned)
        shards = [torch.from_numpy(d.shard_spectra(nf, sp)).cuda() for sp in sps]
        assert all(s.numel() == size * nf for s in shards)
        n = d.trans.nb_gridpoints()
        gps = [torch.full((nf * n,), float("nan"), dtype=torch.float64, device="cuda") for _ in sps]
        d.invtrans_many_sharded(nf, shards, gps)
        d.trans.synchronize()
        return d.bands[comm.rank()], d.bands[comm.rank() + 1], size * nf, [x.cpu().numpy() for x in gps]

    res = run_ranks(nparts, rank)
    assert sum(r[2] for r in res) == total            # the shards are a partition of the replicated array
    for b0, b1, _, many in res:
        for got, ref in zip(many, refs):
            assert np.array_equal(got.reshape(nf, -1), ref[:, off[b0]:off[b1]])


@pytest.mark.parametrize("gridname,nparts,halo", [("O16", 2, 1), ("O32", 3, 2), ("F16", 4, 1)])
def test_native_halo_exchange_between_ranks(gridname, nparts, halo):
    g = atlas_amd.Grid(gridname)

    def rank(comm):
        fs = StructuredColumns(g, halo=halo, nparts=nparts, part=comm.rank())
        hx = HaloExchange()
        hx.setup(fs.partition(), fs.remote_index(), 0, fs.sizeHalo(), halo_begin=fs.sizeOwned(), comm=comm)
        gidx = fs.global_index()
        a = np.where(fs.ghost() == 0, gidx, -1).astype(np.int64)
        f = torch.from_numpy(np.repeat(a[:, None], 4, axis=1).copy()).cuda()
        hx.execute(f)
        hx.synchronize()
        ok = np.array_equal(f.cpu().numpy()[:, 0], gidx) and np.array_equal(f.cpu().numpy()[:, 3], gidx)
        # adjoint: halo contributions are added to their owners, halos zeroed (HaloExchange.h:227-290)
        w = torch.ones(fs.sizeHalo(), 2, dtype=torch.float64, device="cuda")
        hx.execute_adjoint(w)
        hx.synchronize()
        wn = w.cpu().numpy()
        return ok, float(wn[:, 0].sum()), int((wn[fs.ghost() == 1] != 0).sum()), fs.sizeHalo()

    res = run_ranks(nparts, rank)
    assert all(r[0] for r in res)
    assert all(r[2] == 0 for r in res)                                  # halos zeroed
    assert sum(r[1] for r in res) == sum(r[3] for r in res)             # the adjoint conserves the sum


def test_native_paths_over_a_real_rccl_communicator_of_one_rank():
    g = atlas_amd.Grid("O32")
    T, nf = 31, 3
    comm = Comm.rccl(Comm.unique_id(), 1, 0)
    assert comm.kind() == "rccl" and comm.size() == 1 and comm.rank() == 0
    comm.barrier()
    sp = torch.from_numpy(red_spectra(T, nf)).cuda()
    ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    atlas_amd.Trans(g, T).invtrans(nf, sp, ref)
    d = DistributedTrans(g, T, comm=comm, mode="alltoall")
    gp = torch.zeros_like(ref)
    d.invtrans_many(nf, [sp, sp], [gp, gp])
    d.trans.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(gp, ref)
    fs = StructuredColumns(g, halo=2)
    hx = HaloExchange()
    hx.setup(fs.partition(), fs.remote_index(), 0, fs.sizeHalo(), halo_begin=fs.sizeOwned(), comm=comm)
    gidx = fs.global_index()
    f = torch.from_numpy(np.where(fs.ghost() == 0, gidx, -1).astype(np.int64)).cuda()
    hx.execute(f)
    hx.synchronize()
    assert np.array_equal(f.cpu().numpy(), gidx)


def test_config_C3_TL639_O640_137_levels_on_four_ranks():
    """BASELINE config C3 at its own size (TL639 -> O640, 137 levels, 4 latitude-band parts, m-sharded Legendre stage +
    transposition + halo exchange), the four ranks emulated as threads on one GPU: bands equal the single-device transform
    bit for bit (sampled rows of which are checked against the oracle), and after the halo exchange of the band fields
    every halo node of every part holds its owner's value."""
    import oracle
    from helpers import compute_rms, rows_of_every_fft_class
    g = atlas_amd.Grid("O640")
    T, nf, nparts = 639, 137, 4
    sp_h = red_spectra(T, nf, seed=5)
    sp = torch.from_numpy(sp_h).cuda()
    ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    tr = atlas_amd.Trans(g, T)
    tr.invtrans(nf, sp, ref)
    tr.synchronize()
    rows, _ = rows_of_every_fft_class(tr, extra=[0, 319, 640, 1279])   # a northern and a southern row of every kernel class
    del tr
    ref = ref.view(nf, -1)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    for r, want in zip(rows, op.invtrans_rows(nf, sp_h, rows, use_fft=True)):
        assert compute_rms(ref[:, off[r]:off[r + 1]].cpu().numpy(), want) < 1e-12, r

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        n = d.trans.nb_gridpoints()
        gp = torch.zeros(nf * n, dtype=torch.float64, device="cuda")
        d.invtrans(nf, sp, gp)
        d.trans.synchronize()
        b0, b1 = d.bands[comm.rank()], d.bands[comm.rank() + 1]
        same = torch.equal(gp.view(nf, -1), ref[:, off[b0]:off[b1]])
        # the band as a StructuredColumns field (nodes x levels), halo exchange between the ranks
        fs = StructuredColumns(g, halo=1, nparts=nparts, part=comm.rank(), distribution="row_bands")
        assert fs.sizeOwned() == n
        hx = HaloExchange()
        hx.setup(fs.partition(), fs.remote_index(), 0, fs.sizeHalo(), halo_begin=fs.sizeOwned(), comm=comm)
        f = torch.full((fs.sizeHalo(), nf), float("nan"), dtype=torch.float64, device="cuda")
        f[:n] = gp.view(nf, -1).t()
        hx.execute(f)
        hx.synchronize()
        gi = torch.from_numpy(fs.global_index() - 1).cuda()
        return same, torch.equal(f, ref.t()[gi])

    for same, halo_ok in run_ranks(nparts, rank):
        assert same and halo_ok


def test_config_C4_split_TL1279_O1280_137_levels_on_eight_ranks():
    """BASELINE config C4's own split at full size: TL1279 -> O1280, 137 levels over EIGHT latitude-band
    parts -- m-sharded Legendre stage (160 wavenumbers per rank), packed transposition (5.07 GB over 56 pair messages, the
    largest 100 MB), band Fourier stage reading the packed runs -- the eight ranks emulated as threads on one GPU: every
    band equals the single-device field bit for bit (sampled rows of which test_gpu_trans.py compares with the oracle), for
    the replicated and for the m-sharded input, and for the second transform of a pipelined call."""
    g = atlas_amd.Grid("O1280")
    T, nf, nparts = 1279, 137, 8
    sp_h = [red_spectra(T, nf, seed=s) for s in (5, 6)]
    sps = [torch.from_numpy(a).cuda() for a in sp_h]
    refs = []
    tr = atlas_amd.Trans(g, T)
    for sp in sps:
        ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp, ref)
        tr.synchronize()
        refs.append(ref.view(nf, -1))
    del tr
    off = np.concatenate([[0], np.cumsum(g.nx())])

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        n = d.trans.nb_gridpoints()
        b0, b1 = d.bands[comm.rank()], d.bands[comm.rank() + 1]
        gps = [torch.full((nf * n,), float("nan"), dtype=torch.float64, device="cuda") for _ in sps]
        d.invtrans_many(nf, sps, gps)
        d.trans.synchronize()
        ok = all(torch.equal(gp.view(nf, -1), ref[:, off[b0]:off[b1]]) for gp, ref in zip(gps, refs))
        shard = torch.from_numpy(d.shard_spectra(nf, sp_h[1])).cuda()      # 1/8 of the coefficients
        gp_s = torch.full((nf * n,), float("nan"), dtype=torch.float64, device="cuda")
        d.invtrans_many_sharded(nf, [shard], [gp_s])
        d.trans.synchronize()
        return ok, torch.equal(gp_s.view(nf, -1), refs[1][:, off[b0]:off[b1]]), shard.numel() * 7 <= sps[1].numel()

    for ok, ok_sharded, small in run_ranks(nparts, rank):
        assert ok and ok_sharded and small


def test_a_failing_rank_releases_the_others_instead_of_hanging():
    """a rank of the in-process communicator that throws between two meeting points (here: a receive whose size
    does not match the peer's send) marks the hub failed; the rank that waits at the meeting point throws too."""
    import ctypes as C
    from atlas_amd import _lib
    exchange = _lib._sig("atlas_amd__Comm__exchange", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    bufs = [torch.zeros(64, dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    errors = [None, None]

    def rank(comm):
        r = comm.rank()
        peer = (C.c_int * 1)(1 - r)
        ptr = (C.c_void_p * 1)(bufs[r].data_ptr())
        nbytes = (C.c_size_t * 1)(16 if r == 0 else 32)       # rank 0 sends 16 bytes, rank 1 expects 32
        if r == 0:
            rc = exchange(comm._h, 1, peer, ptr, nbytes, 0, None, None, None, None)
        else:
            rc = exchange(comm._h, 0, None, None, None, 1, peer, ptr, nbytes, None)
        errors[r] = _lib.last_error().decode() if rc != 0 else None
        return rc

    rcs = run_ranks(2, rank)
    assert all(rc != 0 for rc in rcs), rcs
    assert any("matching send" in (e or "") for e in errors) and any("another rank" in (e or "") or "matching" in (e or "") for e in errors)


def test_ranks_with_different_message_limits_get_an_error_not_a_hang():
    """set_max_message_bytes is collective: the ranks compare the value inside the call (the check used to sit inside the
    next transform, where only the ranks whose setting had changed entered it); a caller alternating between two field counts
    pays the comparison once per (field count, limit) pair"""
    g = atlas_amd.Grid("O32")
    T, nparts = 31, 2
    errors = [None] * nparts

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        try:
            d.set_max_message_bytes(4096 if comm.rank() == 0 else 8192)
        except Exception as e:  # noqa: BLE001
            errors[comm.rank()] = str(e)
        d.set_max_message_bytes(1 << 14)               # agreed again: usable afterwards
        n = d.trans.nb_gridpoints()
        outs = []
        for nf in (3, 1, 3, 1):                        # alternating field counts
            sp = torch.from_numpy(red_spectra(T, nf, seed=5)).cuda()
            gp = torch.zeros(nf * n, dtype=torch.float64, device="cuda")
            d.invtrans(nf, sp, gp)
            d.trans.synchronize()
            outs.append(gp.cpu().numpy())
        assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3])
        return True

    assert all(run_ranks(nparts, rank))
    assert all(e is not None and "message limit" in e for e in errors), errors


def test_one_rank_with_an_invalid_message_limit_is_an_error_on_every_rank_not_a_hang():
    """the argument check of set_max_message_bytes used to run rank-locally BEFORE the collective comparison, so a rank
    passing < 8 bytes threw and left the others waiting in the all-to-all.  Now every rank takes part and then fails: the rank with
    the bad value with the argument error, the others because the limits differ; the previous limit stays in force."""
    g = atlas_amd.Grid("O32")
    T, nparts = 31, 2
    errors = [None] * nparts

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        d.set_max_message_bytes(1 << 14)
        try:
            d.set_max_message_bytes(4 if comm.rank() == 0 else 8192)
        except Exception as e:  # noqa: BLE001
            errors[comm.rank()] = str(e)
        nf = 2
        sp = torch.from_numpy(red_spectra(T, nf, seed=5)).cuda()
        gp = torch.zeros(nf * d.trans.nb_gridpoints(), dtype=torch.float64, device="cuda")
        d.invtrans(nf, sp, gp)                         # still usable, with the limit agreed before
        d.trans.synchronize()
        return bool(torch.isfinite(gp).all())

    assert all(run_ranks(nparts, rank))
    assert errors[0] is not None and "at least one element" in errors[0], errors
    assert errors[1] is not None and "message limit" in errors[1], errors


def test_field_counts_are_compared_on_every_call_in_debug_mode(monkeypatch):
    """ATLAS_AMD_DIST_CHECK=always: ranks calling with different, individually already-compared field counts get an error
    (by default a (field count, limit) pair is compared only the first time it is used: ADVICE r4 / r5)"""
    monkeypatch.setenv("ATLAS_AMD_DIST_CHECK", "always")
    g = atlas_amd.Grid("O32")
    T, nparts = 31, 2
    errors = [None] * nparts

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall")
        n = d.trans.nb_gridpoints()
        for nf in (1, 2):                              # both counts compared once, on all ranks
            sp = torch.from_numpy(red_spectra(T, nf, seed=5)).cuda()
            gp = torch.zeros(nf * n, dtype=torch.float64, device="cuda")
            d.invtrans(nf, sp, gp)
            d.trans.synchronize()
        nf = 1 if comm.rank() == 0 else 2              # now the ranks disagree
        sp = torch.from_numpy(red_spectra(T, nf, seed=5)).cuda()
        gp = torch.zeros(nf * n, dtype=torch.float64, device="cuda")
        try:
            d.invtrans(nf, sp, gp)
        except Exception as e:  # noqa: BLE001
            errors[comm.rank()] = str(e)
        return True

    assert all(run_ranks(nparts, rank))
    assert all(e is not None and "field count" in e for e in errors), errors


def test_exchange_timings_of_the_distributed_transform():
    """[r6] VERDICT r5 item 5b: with profiling on, every rank reports the pack-kernel and send / receive-group times of its transposition
    (events on the communication stream) and what it moves per transform: bytes to / from other ranks and to the busiest peer -- the
    figures bench.py lists per rank in the N > 1 line.  What leaves the ranks must be what arrives."""
    g = atlas_amd.Grid("O160")
    T, nf, nparts = 159, 6, 4

    def rank(comm):
        d = DistributedTrans(g, T, comm=comm, mode="alltoall", profile=True)
        n = d.trans.nb_gridpoints()
        sps = [torch.from_numpy(red_spectra(T, nf, seed=s)).cuda() for s in (1, 2, 3)]
        gps = [torch.zeros(nf * n, dtype=torch.float64, device="cuda") for _ in sps]
        assert d.exchange_timings(reset=True)["transforms"] == 0
        d.invtrans_many(nf, sps, gps)
        d.trans.synchronize()
        x = d.exchange_timings(reset=True)
        assert x["transforms"] == 3 and x["pack_ms"] > 0 and x["exchange_ms"] > 0
        assert d.exchange_timings()["transforms"] == 0                      # reset
        return comm.rank(), x

    outs = run_ranks(nparts, rank)
    sent = {r: x["bytes_sent"] for r, x in outs}
    recv = {r: x["bytes_received"] for r, x in outs}
    assert sum(sent.values()) == sum(recv.values()) > 0                      # what leaves a rank arrives at another
    for r, x in outs:
        assert 0 < x["bytes_to_busiest_peer"] <= x["bytes_sent"] and x["peers"] == nparts - 1


@pytest.mark.parametrize("gridname,T,nf,nparts,mode", [("O160", 159, 9, 3, "alltoall"), ("O160", 159, 9, 4, "mirror"), ("O160", 159, 9, 3, "band"),
                                                       ("O64", 63, 4, 2, "alltoall")])
def test_rows_beyond_the_lds_in_the_distributed_transform(gridname, T, nf, nparts, mode, monkeypatch):
    """[r6] the rows whose transform does not fit a CU's LDS (O2560's four longest row lengths -- where a transform is most likely to be
    distributed) are a matrix product whose coefficients are first gathered through the Fourier stage's own reader: every layout of the
    intermediate (single piece, packed runs of the m-sharded transposition, latitude bands, mirror bands) must give the bits of the
    single-device transform.  The test hook ATLAS_AMD_FFT_LDS_ELEMS sends the longer rows of a small grid down that path on every rank."""
    hook = "300" if gridname == "O160" else "150"
    g = atlas_amd.Grid(gridname)
    sp = torch.from_numpy(red_spectra(T, nf, seed=11)).cuda()
    plain = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    t0 = atlas_amd.Trans(g, T)
    t0.invtrans(nf, sp, plain)
    t0.synchronize()
    monkeypatch.setenv("ATLAS_AMD_FFT_LDS_ELEMS", hook)          # read when an object is built: stays set for the ranks' objects
    tr = atlas_amd.Trans(g, T)
    ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp, ref)
    tr.synchronize()
    assert not torch.equal(plain, ref) and float((plain - ref).abs().max()) < 1e-11      # the hook did select another algorithm
    refv = ref.view(nf, -1)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    if mode == "alltoall":
        def rank(comm):
            d = DistributedTrans(g, T, comm=comm, mode="alltoall")
            gp = torch.full((nf * d.trans.nb_gridpoints(),), float("nan"), dtype=torch.float64, device="cuda")
            d.invtrans(nf, sp, gp)
            d.trans.synchronize()
            return d.bands, gp.view(nf, -1).clone()

        for r, (bands, gp) in enumerate(run_ranks(nparts, rank)):
            assert torch.equal(gp, refv[:, off[bands[r]]:off[bands[r + 1]]]), r
        return
    seen = []
    for part in range(nparts):          # exchange-free decompositions: every rank's object in turn
        t = atlas_amd.Trans(g, T, nparts=nparts, part=part, shard=mode)
        rows = t.owned_rows()
        cols = torch.from_numpy(np.concatenate([np.arange(off[j], off[j + 1]) for j in rows])).cuda()
        gp = torch.full((nf * cols.numel(),), float("nan"), dtype=torch.float64, device="cuda")
        t.invtrans(nf, sp, gp)
        t.synchronize()
        assert torch.equal(gp.view(nf, -1), refv[:, cols]), part
        seen.append(rows)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(g.ny()))
