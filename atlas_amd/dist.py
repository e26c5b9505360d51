"""Multi-GPU inverse transform: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

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


class DistributedTrans:
    def __init__(self, grid, truncation, group=None, profile=False, mode="auto"):
        """mode: "alltoall" = Legendre stage sharded by wavenumber, RCCL all-to-all, Fourier stage on the local band
                 "band"     = both stages on the local latitude band: no exchange, but the hemisphere symmetry cannot
                              be shared between devices, so the Legendre stage costs 2/P instead of 1/P
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
        if mode not in ("alltoall", "band"):
            raise ValueError("mode must be 'auto', 'alltoall' or 'band'")
        self.mode = mode
        self.trans = Trans(grid, truncation, profile=profile, nparts=self.nparts, part=self.part,
                           shard="band" if mode == "band" else "m")
        self.trans.use_torch_stream()
        self.T = truncation
        self.bands = self.trans.bands()
        self._buf = {}
        self._torch, self._dist = torch, dist

    def _buffers(self, nf, slot):
        key = (nf, slot)
        if key not in self._buf:
            torch = self._torch
            RP = self.trans.fourier_row_pitch(nf)
            plan = transpose_plan(len(self.trans.grid.nx()), self.T, RP, self.bands, self.nparts, self.part)
            F = torch.empty(self.trans.fourier_size(nf), dtype=torch.float64, device="cuda")
            R = torch.empty(max(sum(plan["out_splits"]), 1), dtype=torch.float64, device="cuda")
            self._buf[key] = (F, R, plan, RP)
        return self._buf[key]

    def _legendre_and_exchange(self, nf, sp, slot, async_op):
        F, R, plan, RP = self._buffers(nf, slot)
        self.trans.legendre_device(self.T, nf, sp, F)
        work = self._dist.all_to_all_single(R, F, output_split_sizes=plan["out_splits"],
                                            input_split_sizes=plan["in_splits"], group=self.group, async_op=async_op)
        return work

    def _fourier(self, nf, slot, gp):
        F, R, plan, RP = self._buffers(nf, slot)
        parts = [R[o:] for o in plan["out_offsets"]]
        self.trans.fourier_device(nf, 0, parts, plan["cnt"], gp)

    def invtrans(self, nf, sp, gp):
        """one distributed transform; sp: full spectra (replicated), gp: nf * local-band points"""
        if self.mode == "band":
            return self.trans.invtrans(nf, sp, gp)
        self._legendre_and_exchange(nf, sp, 0, async_op=False)
        self._fourier(nf, 0, gp)
        return gp

    def invtrans_many(self, nf, sps, gps):
        """software pipeline over several transforms: the all-to-all of transform i runs on RCCL's stream while
        the Legendre stage of transform i+1 and the Fourier stage of transform i-1 run on the compute stream"""
        if self.mode == "band":
            for sp, gp in zip(sps, gps):
                self.trans.invtrans(nf, sp, gp)
            return gps
        works = []
        for i, sp in enumerate(sps):
            works.append(self._legendre_and_exchange(nf, sp, i % 2, async_op=True))
            if i > 0:
                works[i - 1].wait()
                self._fourier(nf, (i - 1) % 2, gps[i - 1])
        works[-1].wait()
        self._fourier(nf, (len(sps) - 1) % 2, gps[-1])
        return gps
