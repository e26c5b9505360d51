"""torch.distributed implementation of the multi-GPU inverse transform (backend "nccl" = RCCL, or gloo on CPU).

NOT the product path any more: atlas_amd/dist.py drives the distributed transform inside the library (C++ / RCCL,
csrc/dist_trans.hip).  This module is kept (a) for the CPU tests of the driver logic, which run real processes over gloo
with a stand-in transform (tests/test_bench_logic.py, tests/test_dist_plan.py), (b) as the cross-check / fallback
implementation bench.py can select with --dist-impl torch.

TransLocal itself is single-process (it throws if mpi::size() > 1, src/atlas/trans/local/TransLocal.cc:338-340); the
decomposition below is the one the north star asks for and the one ectrans uses:

  * Legendre stage sharded by zonal wavenumber: rank p owns the m with m % P == p (round robin balances the
    triangular cost to < 1 %) and holds only their slices of the Legendre table;
  * all-to-all m -> latitude transpose of the Fourier intermediate.  Rank p's intermediate is
    F_p[lat][m_local][RP]; the rows of latitude band q form ONE contiguous slab, so the transpose is a single
    all_to_all_single with per-peer split sizes (pairwise exchange uses all 7 xGMI links of a GPU at once);
  * Fourier stage on the local latitude band (whole rows, balanced by grid points with Atlas's BandsDistribution
    rule); the FFT kernel gathers wavenumber m from piece m % P at local index m // P, no repacking pass.

The result is distributed by latitude band in the owned-point order of functionspace::StructuredColumns, so a
halo exchange (atlas_amd.parallel.HaloExchange) can follow directly."""
import numpy as np

from .trans import Trans


def owned_wavenumbers(T, nparts, part):
    return len(range(part, T + 1, nparts))


def transpose_plan(nlats, T, RP, bands, nparts, part):
    """split sizes (in doubles) of the m -> latitude all-to-all for rank `part`:
    input  = F_part[lat][cnt_part][RP]        -> to rank q: rows bands[q]..bands[q+1]
    output = concat_p G_p[lat in my band][cnt_p][RP]"""
    cnt = [owned_wavenumbers(T, nparts, p) for p in range(nparts)]
    rows = [int(bands[q + 1] - bands[q]) for q in range(nparts)]
    in_splits = [rows[q] * cnt[part] * RP for q in range(nparts)]
    out_splits = [rows[part] * cnt[p] * RP for p in range(nparts)]
    out_offsets = [int(v) for v in np.concatenate([[0], np.cumsum(out_splits)[:-1]])]
    return {"cnt": cnt, "rows": rows, "in_splits": in_splits, "out_splits": out_splits, "out_offsets": out_offsets}


def mode_address(plan, RP, lat_local, m, nparts, f2=0):
    """offset (doubles) of wavenumber m of local row lat_local in the received buffer -- the formula the FFT kernel
    evaluates (fft_kernel.hip: ModeReader)"""
    p, ml = m % nparts, m // nparts
    return plan["out_offsets"][p] + (lat_local * plan["cnt"][p] + ml) * RP + f2


# RCCL message-size guard.  Measured on one MI355X (RCCL 2.26.6, tools/dist_selfcheck.py, profiles/
# r01_rccl_message_probe.txt): a float64 all_to_all_single delivers messages up to 1.00 GiB completely, but only the
# first half of a message of 2.00 GiB or more (the 7.03 GiB slab of a lone rank lost its southern hemisphere) -- no
# error is raised.  One all_to_all_single is therefore used only while the largest per-pair message of the whole
# exchange is <= 512 MiB (8 GPUs at TL1279/O1280/137 levels: 235 MB to a polar band, 60 MB between equatorial
# ones); otherwise the slabs go out as batched point-to-point messages of at most 512 MiB.  A lone rank copies on
# device.  The choice depends on global quantities only, so every rank takes the same path.
MAX_MESSAGE_ELEMS = 1 << 26      # doubles per message (512 MiB)
DEVICE = "cuda"                  # where the exchange buffers live (tests run the control flow on "cpu" over gloo)
FORCE_RCCL_SINGLE_RANK = False   # dev switch (tools/dist_selfcheck.py): send a lone rank's slab through RCCL as well


def exchange_messages(plan, bands, RP, nparts, part, max_message_elems=MAX_MESSAGE_ELEMS):
    """the bounded-size messages of the m -> latitude transpose for rank `part`:
    list of (peer, send_begin, send_end, recv_begin, recv_end) in doubles, offsets into F (flat) and R.  A peer's
    slab is cut by rows into K pieces; K depends only on global quantities, so both ends cut alike, and the pieces
    of one pair are listed (and therefore sent and received) in the same order on both ends."""
    cnt, rows = plan["cnt"], plan["rows"]
    biggest = max(rows) * max(cnt) * RP
    K = max(1, -(-biggest // max_message_elems))
    K = max(1, min(K, min(r for r in rows if r > 0))) if any(rows) else 1
    msgs = []
    for k in range(K):
        for peer in range(nparts):
            s0, s1 = rows[peer] * k // K, rows[peer] * (k + 1) // K          # rows of the peer's band I send
            r0, r1 = rows[part] * k // K, rows[part] * (k + 1) // K          # rows of my band I receive
            sb = (int(bands[peer]) + s0) * cnt[part] * RP
            se = (int(bands[peer]) + s1) * cnt[part] * RP
            rb = plan["out_offsets"][peer] + r0 * cnt[peer] * RP
            re = plan["out_offsets"][peer] + r1 * cnt[peer] * RP
            msgs.append((peer, sb, se, rb, re))
    return msgs


def transpose_exchange(F, R, plan, bands, RP, nparts, part, group=None, async_op=False,
                       max_message_elems=MAX_MESSAGE_ELEMS):
    """m -> latitude transpose of the Fourier intermediate, F (this rank's wavenumbers, all rows) -> R (all
    wavenumbers, this rank's rows).  Returns the list of outstanding work handles (empty when complete or when the
    remaining work is ordered on the current stream)."""
    import torch.distributed as dist
    Ff = F.reshape(-1)
    if nparts == 1 and not FORCE_RCCL_SINGLE_RANK:
        R[:plan["out_splits"][0]].copy_(Ff[:plan["in_splits"][0]])        # nothing to exchange: one device copy
        return []
    if max(plan["rows"]) * max(plan["cnt"]) * RP <= max_message_elems:      # the largest message of ANY pair
        w = dist.all_to_all_single(R, Ff, output_split_sizes=plan["out_splits"], input_split_sizes=plan["in_splits"],
                                   group=group, async_op=async_op)
        return [w] if async_op else []
    ops = []
    for peer, sb, se, rb, re in exchange_messages(plan, bands, RP, nparts, part, max_message_elems):
        if peer == part:
            R[rb:re].copy_(Ff[sb:se])
        else:
            # P2POp addresses the peer by its GLOBAL rank, also inside a sub-group (as parallel_torch.exchange_packed does)
            gpeer = dist.get_global_rank(group, peer) if group is not None else peer
            if se > sb:
                ops.append(dist.P2POp(dist.isend, Ff[sb:se], gpeer, group))
            if re > rb:
                ops.append(dist.P2POp(dist.irecv, R[rb:re], gpeer, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return list(works)
    for w in works:
        w.wait()
    return []


class DistributedTrans:
    def __init__(self, grid, truncation, group=None, profile=False, mode="auto"):
        """mode: "alltoall" = Legendre stage sharded by wavenumber, RCCL all-to-all, Fourier stage on the local band
                 "band"     = both stages on the local latitude band: no exchange, but the hemisphere symmetry cannot
                              be shared between devices, so the Legendre stage costs 2/P instead of 1/P
                 "mirror"   = both stages on a northern band of rows and its mirror image in the south: no exchange
                              AND the hemisphere symmetry is kept (work 1/P); the rank's output is two row ranges
                              (Trans.owned_rows()), not one Atlas band
                 "auto"     = "band" below 8 ranks (the transposition moves 7.55 GB * (P-1)/P^2 per device and
                              transform over P-1 point-to-point xGMI links: link-bound for P = 2, 4), else "alltoall"
        """
        import torch
        import torch.distributed as dist
        self.group = group
        self.nparts = dist.get_world_size(group)
        self.part = dist.get_rank(group)
        if mode == "auto":
            mode = "band" if self.nparts < 8 else "alltoall"
        if mode not in ("alltoall", "band", "mirror"):
            raise ValueError("mode must be 'auto', 'alltoall', 'band' or 'mirror'")
        self.mode = mode
        self.trans = Trans(grid, truncation, profile=profile, nparts=self.nparts, part=self.part,
                           shard={"band": "band", "mirror": "mirror", "alltoall": "m"}[mode])
        self.trans.use_torch_stream()
        self.T = truncation
        self.bands = self.trans.bands() if mode != "mirror" else None
        self._buf = {}
        self._torch, self._dist = torch, dist

    def _buffers(self, nf, slot):
        key = (nf, slot)
        if key not in self._buf:
            torch = self._torch
            RP = self.trans.fourier_row_pitch(nf)
            plan = transpose_plan(len(self.trans.grid.nx()), self.T, RP, self.bands, self.nparts, self.part)
            F = torch.empty(self.trans.fourier_size(nf), dtype=torch.float64, device=DEVICE)
            R = torch.empty(max(sum(plan["out_splits"]), 1), dtype=torch.float64, device=DEVICE)
            self._buf[key] = (F, R, plan, RP)
        return self._buf[key]

    def _legendre_and_exchange(self, nf, sp, slot, async_op):
        F, R, plan, RP = self._buffers(nf, slot)
        self.trans.legendre_device(self.T, nf, sp, F)
        return transpose_exchange(F, R, plan, self.bands, RP, self.nparts, self.part, group=self.group,
                                  async_op=async_op)

    def _fourier(self, nf, slot, gp):
        F, R, plan, RP = self._buffers(nf, slot)
        parts = [R[o:] for o in plan["out_offsets"]]
        self.trans.fourier_device(nf, 0, parts, plan["cnt"], gp)

    def invtrans(self, nf, sp, gp):
        """one distributed transform; sp: full spectra (replicated), gp: nf * local-band points"""
        if self.mode in ("band", "mirror"):
            return self.trans.invtrans(nf, sp, gp)
        self._legendre_and_exchange(nf, sp, 0, async_op=False)
        self._fourier(nf, 0, gp)
        return gp

    def invtrans_many(self, nf, sps, gps):
        """software pipeline over several transforms: the all-to-all of transform i runs on RCCL's stream while
        the Legendre stage of transform i+1 and the Fourier stage of transform i-1 run on the compute stream"""
        if self.mode in ("band", "mirror"):
            for sp, gp in zip(sps, gps):
                self.trans.invtrans(nf, sp, gp)
            return gps
        works = []
        for i, sp in enumerate(sps):
            works.append(self._legendre_and_exchange(nf, sp, i % 2, async_op=True))
            if i > 0:
                for w in works[i - 1]:
                    w.wait()
                self._fourier(nf, (i - 1) % 2, gps[i - 1])
        for w in works[-1]:
            w.wait()
        self._fourier(nf, (len(sps) - 1) % 2, gps[-1])
        return gps
