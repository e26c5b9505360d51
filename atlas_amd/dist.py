"""Multi-GPU inverse transform, one process (or host thread) per GPU, driven INSIDE the library.

TransLocal itself is single-process (it throws if mpi::size() > 1, src/atlas/trans/local/TransLocal.cc:338-340); the
decomposition is the one the north star asks for and the one ectrans uses (csrc/dist_trans.h):

  * Legendre stage sharded by zonal wavenumber: rank p owns the m with m % P == p (round robin balances the triangular
    cost to < 1 %) and holds only their slices of the Legendre table;
  * m -> latitude transposition of the Fourier intermediate over the library's communicator (csrc/comm.h: RCCL
    ncclSend / ncclRecv groups over xGMI, messages of at most 512 MiB), on a second HIP stream, pipelined against the
    neighbouring transforms;
  * Fourier stage on the local latitude band (whole rows, balanced by grid points with Atlas's BandsDistribution rule);
    the FFT kernels gather wavenumber m from piece m % P at local index m // P, no repacking pass.

The result is distributed by latitude band in the owned-point order of functionspace::StructuredColumns, so a halo
exchange (atlas_amd.parallel.HaloExchange over the same communicator) can follow directly.

This module only passes pointers; the torch.distributed implementation of the same driver (atlas_amd/dist_torch.py) is
kept for CPU tests over gloo and as a cross-check."""
import ctypes as C

import numpy as np

from . import _lib
from .comm import Comm
from .dist_torch import mode_address, owned_wavenumbers, transpose_plan  # noqa: F401  (index arithmetic, shared)
from .trans import Trans

c_void_p, c_int = C.c_void_p, C.c_int
_sig = _lib._sig
Trans_invtrans_distributed = _sig("atlas_amd__Trans__invtrans_distributed", c_int, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p)
Trans_invtrans_distributed_many_halo = _sig("atlas_amd__Trans__invtrans_distributed_many_halo", c_int, c_void_p, c_void_p,
                                           c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p)
Trans_invtrans_distributed_many = _sig("atlas_amd__Trans__invtrans_distributed_many", c_int, c_void_p, c_void_p, c_int,
                                       c_int, c_void_p, c_void_p)
Trans_invtrans_distributed_sharded = _sig("atlas_amd__Trans__invtrans_distributed_sharded", c_int, c_void_p, c_void_p, c_int,
                                          c_int, c_void_p, c_void_p)
Trans_spectral_shard = _sig("atlas_amd__Trans__spectral_shard", c_int, c_void_p, c_void_p, c_void_p)
Trans_pack_probe = _sig("atlas_amd__Trans__pack_probe", c_int, c_void_p, c_int, c_int, c_void_p, c_void_p)
Trans_fourier_packed_probe = _sig("atlas_amd__Trans__fourier_packed_probe", c_int, c_void_p, c_int, c_int, c_void_p)
Trans_timings_distributed = _sig("atlas_amd__Trans__timings_distributed", c_int, c_void_p, c_void_p, c_void_p, c_int)
Trans_set_max_message_bytes = _sig("atlas_amd__Trans__set_max_message_bytes", c_int, c_void_p, c_void_p, C.c_longlong)
_transpose_messages = _sig("atlas_amd__transpose_messages", c_int, c_int, c_int, c_int, c_int, c_void_p, C.c_longlong,
                           c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)


_packed_transpose_messages = _sig("atlas_amd__packed_transpose_messages", c_int, c_int, c_void_p, c_int, c_int, c_int,
                                  c_void_p, C.c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p)


def packed_transpose_messages(row_mmax, cols, bands, nparts, part, max_message_elems=1 << 26):
    """the messages of the library's distributed transform (csrc/dist_trans.hip: packed runs) for rank `part`:
    ([(peer, send_begin, send_end, recv_begin, recv_end)] in doubles, (doubles sent, doubles received))"""
    mm = np.ascontiguousarray(row_mmax, dtype=np.int32)
    b = np.ascontiguousarray(bands, dtype=np.int32)
    cap = 64 * int(nparts) + 64
    while True:
        peer = np.zeros(cap, dtype=np.int32)
        arr = [np.zeros(cap, dtype=np.int64) for _ in range(4)]
        n = c_int(0)
        tot = np.zeros(2, dtype=np.int64)
        _lib.check(_packed_transpose_messages(len(mm), mm.ctypes.data, int(cols), int(nparts), int(part), b.ctypes.data,
                                              int(max_message_elems), cap, peer.ctypes.data, *[a.ctypes.data for a in arr],
                                              C.byref(n), tot.ctypes.data))
        if n.value <= cap:
            return [(int(peer[i]),) + tuple(int(a[i]) for a in arr) for i in range(n.value)], (int(tot[0]), int(tot[1]))
        cap = n.value


def transpose_messages(T, RP, bands, nparts, part, max_message_elems=1 << 26):
    """the library's message list of the transposition for rank `part` (csrc/dist_trans.hip):
    [(peer, send_begin, send_end, recv_begin, recv_end)] in doubles"""
    b = np.ascontiguousarray(bands, dtype=np.int32)
    cap = 64 * int(nparts) + 64
    while True:
        peer = np.zeros(cap, dtype=np.int32)
        arr = [np.zeros(cap, dtype=np.int64) for _ in range(4)]
        n = c_int(0)
        _lib.check(_transpose_messages(int(T), int(RP), int(nparts), int(part), b.ctypes.data, int(max_message_elems), cap,
                                       peer.ctypes.data, *[a.ctypes.data for a in arr], C.byref(n)))
        if n.value <= cap:
            return [(int(peer[i]),) + tuple(int(a[i]) for a in arr) for i in range(n.value)]
        cap = n.value


class DistributedTrans:
    def __init__(self, grid, truncation, comm=None, group=None, profile=False, mode="auto", nparts=None, part=None):
        """comm : an atlas_amd.comm.Comm; None -> an RCCL communicator over the torch.distributed group `group`
                  (torch.distributed carries the 128-byte unique id only).
        mode: "alltoall" = Legendre stage sharded by wavenumber, transposition over the communicator, Fourier stage on
                           the local band (the decomposition BASELINE configs C3 / C4 name)
              "band"     = both stages on the local latitude band: no exchange, but the hemisphere symmetry cannot be
                           shared between devices, so the Legendre stage costs 2/P instead of 1/P
              "mirror"   = both stages on a northern band of rows and its mirror image in the south: no exchange AND the
                           hemisphere symmetry is kept (work 1/P); the rank's output is two row ranges
                           (Trans.owned_rows()), not one Atlas band
              "auto"     = "alltoall" """
        if mode == "auto":
            mode = "alltoall"
        if mode not in ("alltoall", "band", "mirror"):
            raise ValueError("mode must be 'auto', 'alltoall', 'band' or 'mirror'")
        self.mode = mode
        self.comm = comm
        if comm is None and (nparts is None or mode == "alltoall"):
            import torch.distributed as dist
            if mode == "alltoall":
                self.comm = comm = Comm.rccl_from_torch(group)
            else:
                nparts, part = dist.get_world_size(group), dist.get_rank(group)
        if comm is not None:
            nparts, part = comm.size(), comm.rank()
        self.nparts, self.part = int(nparts), int(part)
        self.trans = Trans(grid, truncation, profile=profile, nparts=self.nparts, part=self.part,
                           shard={"band": "band", "mirror": "mirror", "alltoall": "m"}[mode], tables="device")
        self.T = truncation
        self.bands = self.trans.bands() if mode != "mirror" else None

    def spectral_shard(self):
        """layout of this rank's share of the spectra for invtrans_sharded: (moff[T+1] in doubles per field, -1 where the
        wavenumber belongs to another rank; doubles per field of the whole shard)"""
        T = self.trans.truncation()
        moff = np.zeros(T + 1, dtype=np.int64)
        size = C.c_longlong(0)
        _lib.check(Trans_spectral_shard(self.trans._h, moff.ctypes.data, C.byref(size)))
        return moff, int(size.value)

    def shard_spectra(self, nf, sp_full):
        """host helper: this rank's wavenumbers out of a replicated spectral array (numpy, layout of invtrans)"""
        T = self.trans.truncation()
        moff, size = self.spectral_shard()
        out = np.empty(size * nf, dtype=np.float64)
        full = np.asarray(sp_full).reshape(-1)
        for m in range(T + 1):
            if moff[m] >= 0:
                src = (2 * T + 3 - m) * m // 2 * 2 * nf
                n = 2 * (T + 1 - m) * nf
                out[moff[m] * nf:moff[m] * nf + n] = full[src:src + n]
        return out

    def invtrans_many_sharded(self, nf, sp_shards, gps):
        """as invtrans_many with every sp_shards[i] holding only this rank's wavenumbers (device tensors)"""
        n = len(sp_shards)
        a = (C.c_void_p * n)(*[s.data_ptr() for s in sp_shards])
        b = (C.c_void_p * n)(*[g.data_ptr() for g in gps])
        with _lib.torch_stream_order(self.trans.stream()):
            _lib.check(Trans_invtrans_distributed_sharded(self.trans._h, self.comm._h, n, int(nf), a, b))
        return gps

    def exchange_timings(self, reset=False):
        """per-transform averages of this rank's transposition since the last reset (Trans made with profile=True): pack kernel and
        send / receive group on the communication stream, bytes per transform to / from other ranks and to the busiest peer"""
        if self.mode != "alltoall":
            return None
        out = (C.c_double * 8)()
        _lib.check(Trans_timings_distributed(self.trans._h, self.comm._h, out, int(reset)))
        n = max(int(out[2]), 1)
        return {"pack_ms": out[0] / n, "exchange_ms": out[1] / n, "transforms": int(out[2]), "bytes_sent": int(out[3]),
                "bytes_received": int(out[4]), "bytes_to_busiest_peer": int(out[5]), "peers": int(out[6])}

    def set_max_message_bytes(self, nbytes):
        _lib.check(Trans_set_max_message_bytes(self.trans._h, self.comm._h, int(nbytes)))

    def invtrans(self, nf, sp, gp):
        """one distributed transform; sp: full spectra (replicated) on the device, gp: nf * local-band points"""
        if self.mode in ("band", "mirror"):
            return self.trans.invtrans(nf, sp, gp)
        with _lib.torch_stream_order(self.trans.stream()):
            _lib.check(Trans_invtrans_distributed(self.trans._h, self.comm._h, int(nf), sp.data_ptr(), gp.data_ptr()))
        return gp

    def invtrans_many(self, nf, sps, gps):
        """several transforms, software-pipelined inside the library: the exchange of transform i runs on the
        communication stream while the Legendre stage of i+1 and the Fourier stage of i-1 run on the Trans stream"""
        if self.mode in ("band", "mirror"):
            for sp, gp in zip(sps, gps):
                self.trans.invtrans(nf, sp, gp)
            return gps
        n = len(sps)
        spp = (c_void_p * n)(*[s.data_ptr() for s in sps])
        gpp = (c_void_p * n)(*[g.data_ptr() for g in gps])
        with _lib.torch_stream_order(self.trans.stream()):
            _lib.check(Trans_invtrans_distributed_many(self.trans._h, self.comm._h, n, int(nf), spp, gpp))
        return gps

    def invtrans_many_halo(self, nf, sps, gps, hx, fields):
        """invtrans_many whose outputs also go, transposed, into the StructuredColumns fields `fields[i]` of shape
        (size_halo, nf) on this rank's band partition, followed by their halo exchange (`hx`: a parallel.HaloExchange set up
        on the same communicator) -- on the communication stream, beside the Legendre stage of the next transforms."""
        if self.mode != "alltoall":
            raise NotImplementedError("invtrans_many_halo needs the wavenumber-sharded decomposition")
        n = len(sps)
        spp = (c_void_p * n)(*[s.data_ptr() for s in sps])
        gpp = (c_void_p * n)(*[g.data_ptr() for g in gps])
        fpp = (c_void_p * n)(*[f.data_ptr() for f in fields])
        with _lib.torch_stream_order(self.trans.stream()):
            _lib.check(Trans_invtrans_distributed_many_halo(self.trans._h, self.comm._h, n, int(nf), spp, gpp, hx._h, fpp))
        return fields
