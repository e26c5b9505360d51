"""CPU tests of the multi-GPU decomposition logic (atlas_amd/dist_torch.py (torch.distributed transport; the product path is native: atlas_amd/dist.py)):
  * split sizes / offsets of the m -> latitude transpose and the address formula the FFT kernel uses,
    exercised with a REAL all_to_all_single over gloo with world_size 2 and 3 (CPU tensors);
  * HaloExchange.setup over a gloo process group against the serial oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bands(nx, nparts):
    off = np.concatenate([[0], np.cumsum(nx)])
    b = [0] * (nparts + 1)
    b[nparts] = len(nx)
    prev = 0
    for j in range(len(nx)):
        part = int(off[j] * nparts // off[-1])
        for q in range(prev + 1, part + 1):
            b[q] = j
        prev = max(prev, part)
    return b


def _transpose_worker(rank, world, port, T, RP, nx, q, limits=None, async_op=False):
    sys.path.insert(0, ROOT)
    from atlas_amd.dist_torch import mode_address, owned_wavenumbers, transpose_exchange, transpose_plan
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    nlats = len(nx)
    bands = _bands(nx, world)
    plan = transpose_plan(nlats, T, RP, bands, world, rank)
    cnt = owned_wavenumbers(T, world, rank)
    # synthetic intermediate: value encodes (lat, m, r)
    F = torch.zeros(nlats, cnt, RP, dtype=torch.float64)
    for ml in range(cnt):
        m = rank + ml * world
        F[:, ml, :] = (torch.arange(nlats, dtype=torch.float64)[:, None] * 1e6 + m * 1e3
                       + torch.arange(RP, dtype=torch.float64)[None, :])
    R = torch.full((sum(plan["out_splits"]),), -1.0, dtype=torch.float64)
    if limits is None:
        # the call the 8-GPU run makes: one all_to_all_single with per-peer split sizes
        dist.all_to_all_single(R, F.reshape(-1), output_split_sizes=plan["out_splits"],
                               input_split_sizes=plan["in_splits"])
    else:
        # the product's exchange entry point; small limits force the bounded-size point-to-point path
        works = transpose_exchange(F, R, plan, bands, RP, world, rank, async_op=async_op,
                                   max_message_elems=limits)
        for w in works:
            w.wait()
    ok = True
    for lat in range(bands[rank], bands[rank + 1]):
        for m in range(T + 1):
            a = mode_address(plan, RP, lat - bands[rank], m, world)
            want = lat * 1e6 + m * 1e3 + np.arange(RP)
            ok = ok and np.array_equal(R[a:a + RP].numpy(), want)
    q.put((rank, ok, bands))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_m_to_latitude_transpose_over_gloo(world):
    nx = np.array([20 + 4 * j for j in range(8)] + [20 + 4 * j for j in range(8)][::-1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transpose_worker, args=(r, world, port, 11, 16, nx, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    bands = res[0][2]
    assert bands[0] == 0 and bands[-1] == len(nx) and all(b >= a for a, b in zip(bands, bands[1:]))


@pytest.mark.parametrize("world,limits,async_op", [(2, 1 << 26, False),   # default limit: one all_to_all_single
                                                   (2, 100, True),        # 100-double messages: K > 1
                                                   (3, 64, False),
                                                   (3, 300, True)])
def test_bounded_size_exchange_over_gloo(world, limits, async_op):
    """atlas_amd.dist_torch.transpose_exchange: the single all_to_all_single and the bounded-size point-to-point messages
    must deliver the same transposed intermediate"""
    nx = np.array([20 + 4 * j for j in range(8)] + [20 + 4 * j for j in range(8)][::-1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transpose_worker, args=(r, world, port, 11, 16, nx, q, limits, async_op))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)


def _subgroup_worker(grank, gworld, members, port, T, RP, nx, q, limits):
    """ranks `members` of a larger world form the transform's group; the others only take part in new_group()"""
    sys.path.insert(0, ROOT)
    from atlas_amd.dist_torch import mode_address, owned_wavenumbers, transpose_exchange, transpose_plan
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=grank, world_size=gworld)
    group = dist.new_group(ranks=members)
    ok = True
    if grank in members:
        world, rank = len(members), members.index(grank)
        nlats = len(nx)
        bands = _bands(nx, world)
        plan = transpose_plan(nlats, T, RP, bands, world, rank)
        cnt = owned_wavenumbers(T, world, rank)
        F = torch.zeros(nlats, cnt, RP, dtype=torch.float64)
        for ml in range(cnt):
            m = rank + ml * world
            F[:, ml, :] = (torch.arange(nlats, dtype=torch.float64)[:, None] * 1e6 + m * 1e3
                           + torch.arange(RP, dtype=torch.float64)[None, :])
        R = torch.full((sum(plan["out_splits"]),), -1.0, dtype=torch.float64)
        transpose_exchange(F, R, plan, bands, RP, world, rank, group=group, max_message_elems=limits)
        for lat in range(bands[rank], bands[rank + 1]):
            for m in range(T + 1):
                a = mode_address(plan, RP, lat - bands[rank], m, world)
                ok = ok and np.array_equal(R[a:a + RP].numpy(), lat * 1e6 + m * 1e3 + np.arange(RP))
    q.put((grank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("limits", [1 << 26, 100])
def test_exchange_inside_a_sub_group_addresses_peers_by_global_rank(limits):
    """group-local rank 1 is global rank 2: the bounded-size point-to-point path must translate (P2POp takes global ranks),
    and must deliver what the single all_to_all_single delivers"""
    nx = np.array([20 + 4 * j for j in range(8)] + [20 + 4 * j for j in range(8)][::-1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, [0, 2], port, 11, 16, nx, q, limits)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res)


def test_exchange_messages_cover_both_buffers_exactly_once():
    from atlas_amd.dist_torch import exchange_messages, transpose_plan
    nx = np.array([20 + 4 * j for j in range(40)] + [20 + 4 * j for j in range(40)][::-1])
    T, RP = 47, 32
    for world in (1, 2, 3, 8):
        bands = _bands(nx, world)
        plans = [transpose_plan(len(nx), T, RP, bands, world, r) for r in range(world)]
        for limit in (1 << 27, 3000, 700):
            msgs = [exchange_messages(plans[r], bands, RP, world, r, limit) for r in range(world)]
            for r in range(world):
                sent = np.zeros(len(nx) * plans[r]["cnt"][r] * RP, dtype=np.int32)
                recv = np.zeros(sum(plans[r]["out_splits"]), dtype=np.int32)
                for peer, sb, se, rb, re in msgs[r]:
                    sent[sb:se] += 1
                    recv[rb:re] += 1
                    one_row = RP * max(plans[r]["cnt"])
                    if limit >= one_row:       # a message is whole rows: bounded up to the rounding to one row
                        assert se - sb <= limit + one_row
                assert (sent == 1).all() and (recv == 1).all()
                # what r sends to `peer` in piece k is what `peer` expects from r in piece k
                for peer in range(world):
                    mine = [(se - sb) for p, sb, se, rb, re in msgs[r] if p == peer]
                    theirs = [(re - rb) for p, sb, se, rb, re in msgs[peer] if p == r]
                    assert mine == theirs


def _halo_worker(rank, world, port, parts, ridxs, sizes, q):
    sys.path.insert(0, ROOT)
    from atlas_amd.parallel import HaloExchange
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    hx = HaloExchange()
    hx.setup(parts[rank], ridxs[rank], 0, sizes[rank], comm=True)
    p = hx.plan()
    q.put((rank, {k: v.tolist() for k, v in p.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_setup_over_gloo_matches_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.halo import HaloExchangeOracle
    from test_host_halo import random_decomposition
    world = 2
    rng = np.random.default_rng(42)
    parts, ridxs, sizes = random_decomposition(rng, world, 150, 40)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, parts, ridxs, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    ranks = [HaloExchangeOracle(r, world) for r in range(world)]
    HaloExchangeOracle.setup(ranks, parts, ridxs, 0, sizes)
    for r in range(world):
        for k in ("sendcounts", "recvcounts", "sendmap", "recvmap"):
            assert res[r][k] == getattr(ranks[r], k).tolist(), (r, k)


def _halo_exchange_worker(rank, world, port, gridname, halo, nlev, q, mirror=False):
    sys.path.insert(0, ROOT)
    import atlas_amd
    from atlas_amd.functionspace import MirrorBandColumns, StructuredColumns
    from atlas_amd.parallel import HaloExchange
    from atlas_amd.parallel_torch import exchange_packed
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = atlas_amd.Grid(gridname)
    fs = (MirrorBandColumns if mirror else StructuredColumns)(g, halo=halo, nparts=world, part=rank)
    hx = HaloExchange()
    # StructuredColumns.cc:145-148: setup(partition, remote_index, base 0, sizeHalo, halo_begin = sizeOwned)
    hx.setup(fs.partition(), fs.remote_index(), 0, fs.sizeHalo(), halo_begin=fs.sizeOwned(), comm=True)
    p = hx.plan()
    glb = fs.global_index().astype(np.float64)
    field = np.full((fs.sizeHalo(), nlev), -1.0)
    field[:fs.sizeOwned()] = glb[:fs.sizeOwned(), None] * 10.0 + np.arange(nlev)[None, :]
    # pack / unpack of the packed-buffer layout (HaloExchange.h:318-331) in numpy -- the device kernels are tested on
    # the GPU; this test is about the communication step between real processes
    sendbuf = torch.from_numpy(np.ascontiguousarray(field[p["sendmap"]]).reshape(-1))
    recvbuf = torch.full((int(p["recvcounts"].sum()) * nlev,), -2.0, dtype=torch.float64)
    exchange_packed(sendbuf, recvbuf, p["sendcounts"], p["senddispls"], p["recvcounts"], p["recvdispls"], nlev)
    field[p["recvmap"]] = recvbuf.numpy().reshape(-1, nlev)
    ok = bool(np.array_equal(field, glb[:, None] * 10.0 + np.arange(nlev)[None, :]))
    # adjoint direction: counts swapped (HaloExchange.h:258-279) -- every owner receives one contribution per ghost copy
    back = torch.full((int(p["sendcounts"].sum()) * nlev,), -3.0, dtype=torch.float64)
    exchange_packed(recvbuf, back, p["recvcounts"], p["recvdispls"], p["sendcounts"], p["senddispls"], nlev)
    ok = ok and bool(np.array_equal(back.numpy().reshape(-1, nlev), field[p["sendmap"]]))
    q.put((rank, ok, int(p["recvcounts"].sum()), int((np.asarray(p["sendcounts"]) > 0).sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,gridname,halo,mirror", [(2, "O16", 1, False), (3, "O16", 2, False), (4, "F16", 1, False),
                                                        (2, "O16", 2, True), (3, "O16", 1, True)])
def test_halo_exchange_communication_step_over_gloo(world, gridname, halo, mirror):
    """StructuredColumns partitions (Atlas equal_bands), or the two-range parts of the mirror-band decomposition
    (MirrorBandColumns), in `world` real processes: distributed HaloExchange.setup, then
    the send/recv step of execute (atlas_amd.parallel_torch.exchange_packed) on packed buffers; afterwards every halo point
    must hold the value of the point it mirrors (test_structuredcolumns_haloexchange.cc:38-60 in spirit)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_exchange_worker, args=(r, world, port, gridname, halo, 3, q, mirror))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _, _ in res), res
    assert all(nrecv > 0 for _, _, nrecv, _ in res)


def test_latitude_bands_follow_bands_distribution_rule():
    # product bands (C++: trans_plan.cpp latitude_bands) are exposed through a Trans object -> GPU only; here the
    # pure rule: band of a row = BandsDistribution partition of the row's first point
    from oracle.structured_columns import bands_partition
    nx = np.array([20 + 4 * j for j in range(8)] * 2)
    off = np.concatenate([[0], np.cumsum(nx)])
    for P in (2, 3, 4, 8):
        b = _bands(nx, P)
        for q in range(P):
            for j in range(b[q], b[q + 1]):
                assert bands_partition(int(off[j]), int(off[-1]), P, 1) == q


@pytest.mark.parametrize("nparts,maxmsg", [(1, 1 << 26), (2, 1 << 26), (3, 4000), (8, 1 << 26), (8, 900), (5, 128)])
def test_library_transpose_messages_equal_the_python_plan(nparts, maxmsg):
    """the message list the library builds for the m -> latitude transposition (csrc/dist_trans.hip, what the RCCL path
    sends) against the Python formulation that the gloo tests above exercise between real processes"""
    import atlas_amd
    from atlas_amd.dist import transpose_messages
    from atlas_amd.dist_torch import exchange_messages, transpose_plan
    g = atlas_amd.Grid("O32")
    T, RP = 31, 16
    ny = g.ny()
    nx = np.asarray(g.nx())
    bands = np.asarray(_bands(nx, nparts), dtype=np.int32)
    assert bands[0] == 0 and bands[-1] == ny
    for part in range(nparts):
        plan = transpose_plan(ny, T, RP, bands, nparts, part)
        want = exchange_messages(plan, bands, RP, nparts, part, maxmsg)
        got = transpose_messages(T, RP, bands, nparts, part, maxmsg)
        assert got == [tuple(int(v) for v in m) for m in want]
        # every double of the receive buffer is written exactly once
        cover = np.zeros(sum(plan["out_splits"]), dtype=np.int32)
        for _, _, _, rb, re in got:
            cover[rb:re] += 1
        assert (cover == 1).all()


def _row_mmax(gridname, T):
    """Fourier truncation of every row through the library's host-only geometry probe"""
    import ctypes as C
    import atlas_amd
    from atlas_amd import _lib
    g = atlas_amd.Grid(gridname)
    nlat0 = np.zeros(T + 1, dtype=np.int32)
    mm = np.zeros(g.ny(), dtype=np.int32)
    probe = _lib._sig("atlas_amd__trans_geometry_probe", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    _lib.check(probe(g._h, T, 0, nlat0.ctypes.data, mm.ctypes.data))
    return g, np.minimum(mm, T)


def _packed_plan_py(row_mmax, cols, bands, nparts, part, maxmsg):
    """Python formulation of csrc/dist_trans.hip: make_packed_transpose_plan + packed_transpose_messages"""
    nl = len(row_mmax)
    rowoff = np.zeros((nparts, nl + 1), dtype=np.int64)
    for p in range(nparts):
        kept = np.where(row_mmax >= p, (row_mmax - p) // nparts + 1, 0)
        rowoff[p, 1:] = np.cumsum(kept * cols)
    out_off, off = [], 0
    for p in range(nparts):
        out_off.append(off)
        off += int(rowoff[p, bands[part + 1]] - rowoff[p, bands[part]])
    biggest = max(int(rowoff[p, bands[q + 1]] - rowoff[p, bands[q]]) for p in range(nparts) for q in range(nparts))
    rows = [int(bands[q + 1] - bands[q]) for q in range(nparts)]
    minrows = max(min([r for r in rows if r > 0], default=1), 1)
    K = max(1, -(-biggest // maxmsg))
    K = max(1, min(K, minrows))

    def largest_piece(kk):      # [r4] pieces are cut at equal row counts: K grows until the largest piece honours the limit
        return max(int(rowoff[p, bands[q] + rows[q] * (k + 1) // kk] - rowoff[p, bands[q] + rows[q] * k // kk])
                   for q in range(nparts) for p in range(nparts) for k in range(kk))
    while K < minrows and largest_piece(K) > maxmsg:
        K += 1
    msgs = []
    for k in range(K):
        for peer in range(nparts):
            s0, s1 = rows[peer] * k // K, rows[peer] * (k + 1) // K
            r0, r1 = rows[part] * k // K, rows[part] * (k + 1) // K
            msgs.append((peer, int(rowoff[part, bands[peer] + s0]), int(rowoff[part, bands[peer] + s1]),
                         out_off[peer] + int(rowoff[peer, bands[part] + r0] - rowoff[peer, bands[part]]),
                         out_off[peer] + int(rowoff[peer, bands[part] + r1] - rowoff[peer, bands[part]])))
    return msgs, int(rowoff[part, nl]), off, rowoff, out_off


@pytest.mark.parametrize("nparts,maxmsg", [(1, 1 << 26), (2, 1 << 26), (3, 500), (8, 1 << 26), (8, 300), (5, 64)])
def test_packed_transposition_messages(nparts, maxmsg):
    """[r3] the messages the library's distributed transform really sends: per row only the wavenumbers up to the row's
    Fourier truncation, 2 * nf columns each.  Library list == Python formulation; every double of the send buffer leaves
    exactly once, every double of the receive buffer is written exactly once; the two ends of every pair agree on the
    piece sizes in the same order; the address the Fourier kernels compute (fft_device.h: ModeReaderT, packed form) lands,
    for every kept (row, m), on the element the owner packed."""
    from atlas_amd.dist import packed_transpose_messages
    g, mm = _row_mmax("O32", 31)
    cols = 2 * 5
    bands = np.asarray(_bands(np.asarray(g.nx()), nparts), dtype=np.int32)
    per_rank = []
    for part in range(nparts):
        want, stot, rtot, rowoff, out_off = _packed_plan_py(mm, cols, bands, nparts, part, maxmsg)
        got, (gs, gr) = packed_transpose_messages(mm, cols, bands, nparts, part, maxmsg)
        assert got == want and (gs, gr) == (stot, rtot)
        send = np.zeros(stot, dtype=np.int32)
        recv = np.zeros(rtot, dtype=np.int32)
        for _, sb, se, rb, re in got:
            send[sb:se] += 1
            recv[rb:re] += 1
        assert (send == 1).all() and (recv == 1).all()
        minrows = min(int(bands[q + 1] - bands[q]) for q in range(nparts) if bands[q + 1] > bands[q])
        if len(got) // nparts < minrows:          # the limit is honoured unless a piece is already a single row of the smallest band
            assert max(se - sb for _, sb, se, _, _ in got) <= maxmsg
        per_rank.append((got, rowoff, out_off))
    for a in range(nparts):
        for b in range(nparts):
            sa = [se - sb for peer, sb, se, _, _ in per_rank[a][0] if peer == b]
            rb_ = [re - rb for peer, _, _, rb, re in per_rank[b][0] if peer == a]
            assert sa == rb_
    # reader addresses: rank q, local row r, wavenumber m -> piece p = m % P, offset inside R
    for q in range(nparts):
        _, rowoff, out_off = per_rank[q]
        seen = set()
        for r in range(int(bands[q + 1] - bands[q])):
            lat = int(bands[q]) + r
            for m in range(int(mm[lat]) + 1):
                p, ml = m % nparts, m // nparts
                o = out_off[p] + int(rowoff[p, lat] - rowoff[p, bands[q]]) + ml * cols
                assert o + cols <= out_off[p] + int(rowoff[p, bands[q + 1]] - rowoff[p, bands[q]])
                seen.add(o)
        assert len(seen) * cols == sum(int(rowoff[p, bands[q + 1]] - rowoff[p, bands[q]]) for p in range(nparts))


def test_packed_transposition_volume_at_operational_resolution():
    """TL1279 -> O1280, 137 fields: the kept part of the Fourier intermediate -- 2 312 346 (row, wavenumber) pairs x 274
    doubles -- is 5.07 GB (7.55 GB as allocated slabs of 288-column rows); a device sends (P - 1) / P of its share to the
    other devices"""
    from atlas_amd.dist import packed_transpose_messages
    g, mm = _row_mmax("O1280", 1279)
    nf = 137
    kept_total = int((mm.astype(np.int64) + 1).sum()) * 2 * nf * 8
    assert kept_total == 2312346 * 274 * 8
    nx = np.asarray(g.nx())
    for P in (2, 4, 8):
        bands = np.asarray(_bands(nx, P), dtype=np.int32)
        off_device = 0
        for part in range(P):
            msgs, (stot, rtot) = packed_transpose_messages(mm, 2 * nf, bands, P, part)
            off_device += sum(se - sb for peer, sb, se, _, _ in msgs if peer != part) * 8
        slab_total = 7.55e9
        assert abs(off_device - kept_total * (P - 1) / P) < 0.02 * kept_total      # the wavenumbers are dealt round robin
        assert off_device < 0.68 * slab_total * (P - 1) / P
