"""Communicators of the library (csrc/comm.h): the inter-GPU transport of the distributed transform and of the halo
exchange.  RCCL (one process per GPU; ncclSend / ncclRecv groups over xGMI) or "local" (N ranks inside one process, one
host thread per rank on one device: tests and single-process drivers).  Nothing here moves field data through Python."""
import ctypes as C

from . import _lib

c_void_p, c_int = C.c_void_p, C.c_int
_sig = _lib._sig
Comm_id_bytes = _sig("atlas_amd__Comm__unique_id_bytes", c_int)
Comm_get_unique_id = _sig("atlas_amd__Comm__get_unique_id", c_int, c_void_p)
Comm_new_rccl = _sig("atlas_amd__Comm__new_rccl", c_void_p, c_void_p, c_int, c_int)
Hub_new = _sig("atlas_amd__CommHub__new", c_void_p, c_int)
Hub_delete = _sig("atlas_amd__CommHub__delete", None, c_void_p)
Comm_new_local = _sig("atlas_amd__Comm__new_local", c_void_p, c_void_p, c_int)
Comm_delete = _sig("atlas_amd__Comm__delete", None, c_void_p)
Comm_size = _sig("atlas_amd__Comm__size", c_int, c_void_p)
Comm_rank = _sig("atlas_amd__Comm__rank", c_int, c_void_p)
Comm_kind = _sig("atlas_amd__Comm__kind", C.c_char_p, c_void_p)
Comm_barrier = _sig("atlas_amd__Comm__barrier", c_int, c_void_p)


class CommHub:
    """rendezvous point of N emulated ranks (threads of this process)"""

    def __init__(self, nranks):
        self._h = _lib.check_ptr(Hub_new(int(nranks)))
        self.nranks = int(nranks)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and Hub_delete is not None:
            Hub_delete(h)
            self._h = None

    def comm(self, rank):
        return Comm(_lib.check_ptr(Comm_new_local(self._h, int(rank))), keep=self)


class Comm:
    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and Comm_delete is not None:
            Comm_delete(h)
            self._h = None

    @staticmethod
    def unique_id():
        buf = (C.c_char * Comm_id_bytes())()
        _lib.check(Comm_get_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def rccl(unique_id, nranks, rank):
        """collective over the ranks: every rank passes the id made by Comm.unique_id() on one of them; the HIP device
        the rank will use must be current"""
        if len(unique_id) != Comm_id_bytes():
            raise ValueError("unique id has the wrong size")
        buf = (C.c_char * len(unique_id)).from_buffer_copy(unique_id)
        return Comm(_lib.check_ptr(Comm_new_rccl(buf, int(nranks), int(rank))))

    @staticmethod
    def rccl_from_torch(group=None):
        """RCCL communicator over the ranks of a torch.distributed group; torch.distributed only carries the 128-byte
        unique id (control plane)"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [Comm.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return Comm.rccl(box[0], world, rank)

    def size(self):
        return Comm_size(self._h)

    def rank(self):
        return Comm_rank(self._h)

    def kind(self):
        return Comm_kind(self._h).decode()

    def barrier(self):
        _lib.check(Comm_barrier(self._h))
