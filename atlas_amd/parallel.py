"""Host-side mirror of atlas::parallel::HaloExchange (reference: src/atlas/parallel/HaloExchange.h:40-148).

setup() is the reference's index logic (C++ host code behind the C ABI); pack / unpack / adjoint run as HIP kernels;
between processes both the setup collectives and the peer-to-peer step run inside the library over its communicator
(atlas_amd.comm.Comm: RCCL ncclSend / ncclRecv groups over xGMI) where the reference uses eckit::mpi
(HaloExchange.h:333-369).  A single process needs no transport at all.  For CPU tests of the index logic between real
processes a torch.distributed (gloo) transport is kept in atlas_amd/parallel_torch.py."""
import ctypes as C

import numpy as np

from . import _lib

c_void_p, c_int = C.c_void_p, C.c_int
_sig = _lib._sig
HX_new = _sig("atlas_amd__HaloExchange__new", c_void_p)
HX_delete = _sig("atlas_amd__HaloExchange__delete", None, c_void_p)
HX_setup = _sig("atlas_amd__HaloExchange__setup", c_int, c_void_p, c_void_p, c_void_p, c_int, c_int)
HX_setup_hb = _sig("atlas_amd__HaloExchange__setup_halo_begin", c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int)
HX_setup_begin = _sig("atlas_amd__HaloExchange__setup_begin", c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                      c_int, c_int)
HX_setup_begin_device = _sig("atlas_amd__HaloExchange__setup_begin_device", c_int, c_void_p, c_int, c_int, c_void_p,
                             c_void_p, c_int, c_int, c_int)
HX_setup_finish = _sig("atlas_amd__HaloExchange__setup_finish", c_int, c_void_p, c_void_p, c_void_p)
HX_nproc = _sig("atlas_amd__HaloExchange__nproc", c_int, c_void_p)
HX_sendcnt = _sig("atlas_amd__HaloExchange__sendcnt", c_int, c_void_p)
HX_recvcnt = _sig("atlas_amd__HaloExchange__recvcnt", c_int, c_void_p)
HX_get = _sig("atlas_amd__HaloExchange__get", c_int, c_void_p, C.c_char_p, c_void_p)
HX_field_op = _sig("atlas_amd__HaloExchange__field_op", c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                   c_void_p, c_int, c_void_p, c_int)
HX_stream = _sig("atlas_amd__HaloExchange__stream", c_void_p, c_void_p)
HX_set_stream = _sig("atlas_amd__HaloExchange__set_stream", c_int, c_void_p, c_void_p)
HX_sync = _sig("atlas_amd__HaloExchange__synchronize", c_int, c_void_p)
HX_setup_comm = _sig("atlas_amd__HaloExchange__setup_comm", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                     c_int)
HX_execute_comm = _sig("atlas_amd__HaloExchange__execute_comm", c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                       c_void_p, c_int, c_int)
HX_strided = {name: _sig(f"atlas_amd__HaloExchange__execute{adj}_strided_{name}", c_int, c_void_p, c_void_p, c_void_p,
                         c_void_p, c_int)
              for adj in ("", "_adjoint") for name in ("int", "long", "float", "double")} and {
    (adj, name): _sig(f"atlas_amd__HaloExchange__execute{adj}_strided_{name}", c_int, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_int)
    for adj in ("", "_adjoint") for name in ("int", "long", "float", "double")}

OP_EXECUTE, OP_ADJOINT, OP_PACK, OP_UNPACK, OP_PACK_ADJ, OP_UNPACK_ADJ, OP_ZERO = range(7)
_NP_DTYPES = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}


def _is_torch(a):
    return hasattr(a, "data_ptr")


def _describe(field):
    """(dtype code, pointer, rank, shape[], strides[] in elements, on_device, itemsize)"""
    if _is_torch(field):
        import torch
        codes = {torch.int32: 0, torch.int64: 1, torch.float32: 2, torch.float64: 3}
        if field.dtype not in codes:
            raise TypeError(f"unsupported dtype {field.dtype}")
        return codes[field.dtype], field.data_ptr(), field.dim(), list(field.shape), list(field.stride()), \
            bool(field.is_cuda), field.element_size()
    if not isinstance(field, np.ndarray) or field.dtype not in _NP_DTYPES:
        raise TypeError("field must be a numpy array / torch tensor of int32, int64, float32 or float64")
    return _NP_DTYPES[field.dtype], field.ctypes.data, field.ndim, list(field.shape), \
        [s // field.itemsize for s in field.strides], False, field.itemsize


class HaloExchange:
    def __init__(self, name=""):
        self.name = name
        self._h = _lib.check_ptr(HX_new())
        self._group = None
        self._is_setup = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and HX_delete is not None:
            HX_delete(h)
            self._h = None

    # ------------------------------------------------------------------ setup (HaloExchange.cc:66-172)
    def setup(self, part, remote_idx, base, size, halo_begin=0, comm=None):
        """comm=None: one process.  comm=a torch.distributed process group (or True for the default group): the
        two collective steps of the reference setup run over torch.distributed."""
        part = np.ascontiguousarray(part, dtype=np.int32)
        ridx = np.ascontiguousarray(remote_idx, dtype=np.int32)
        if len(part) < size or len(ridx) < size:
            raise ValueError("part / remote_idx shorter than size")
        if comm is None:
            _lib.check(HX_setup_hb(self._h, part.ctypes.data, ridx.ctypes.data, int(base), int(size), int(halo_begin)))
            self._is_setup = True
            return
        from .comm import Comm
        if not isinstance(comm, Comm):
            # a torch.distributed process group (True: the default group)
            import torch.distributed as dist
            group = None if comm is True else comm
            if dist.get_backend(group) != "nccl":
                from .parallel_torch import setup_over_torch   # CPU tests over gloo
                setup_over_torch(self, part, ridx, base, size, halo_begin, group)
                return
            comm = Comm.rccl_from_torch(group)
        _lib.check(HX_setup_comm(self._h, comm._h, part.ctypes.data, ridx.ctypes.data, int(base), int(size),
                                 int(halo_begin)))
        self._comm = comm
        self._is_setup = True

    def setup_emulated(self, nproc, myproc, part, remote_idx, base, size, halo_begin=0, on_device=False):
        """phase 1 of a multi-rank setup inside ONE process (tests / single-process multi-device drivers);
        call HaloExchange.finish_emulated(list_of_objects) afterwards."""
        part = np.ascontiguousarray(part, dtype=np.int32)
        ridx = np.ascontiguousarray(remote_idx, dtype=np.int32)
        if on_device:
            import torch
            self._keep = (torch.from_numpy(part).cuda(), torch.from_numpy(ridx).cuda())
            _lib.check(HX_setup_begin_device(self._h, nproc, myproc, self._keep[0].data_ptr(), self._keep[1].data_ptr(),
                                             int(base), int(size), int(halo_begin)))
        else:
            _lib.check(HX_setup_begin(self._h, nproc, myproc, part.ctypes.data, ridx.ctypes.data, int(base), int(size),
                                      int(halo_begin)))

    @staticmethod
    def finish_emulated(objs):
        n = len(objs)
        rc = [o._get("recvcounts", n) for o in objs]
        rd = [o._get("recvdispls", n) for o in objs]
        rq = [o._get("send_requests", HX_recvcnt(o._h)) for o in objs]
        for r, o in enumerate(objs):
            sendcounts = np.array([rc[p][r] for p in range(n)], dtype=np.int32)
            parts = [rq[p][rd[p][r]:rd[p][r] + rc[p][r]] for p in range(n)]
            recv_req = np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(0), dtype=np.int32)
            if recv_req.size == 0:
                recv_req = np.zeros(1, dtype=np.int32)
            _lib.check(HX_setup_finish(o._h, sendcounts.ctypes.data, recv_req.ctypes.data))
            o._is_setup = True

    def _get(self, what, n):
        out = np.zeros(max(int(n), 1), dtype=np.int32)
        _lib.check(HX_get(self._h, what.encode(), out.ctypes.data))
        return out[:int(n)]

    # plan accessors (test / driver use)
    def nproc(self):
        return HX_nproc(self._h)

    def sendcnt(self):
        return HX_sendcnt(self._h)

    def recvcnt(self):
        return HX_recvcnt(self._h)

    def plan(self):
        n = self.nproc()
        return {k: self._get(k, n) for k in ("sendcounts", "recvcounts", "senddispls", "recvdispls")} | {
            "sendmap": self._get("sendmap", self.sendcnt()), "recvmap": self._get("recvmap", self.recvcnt())}

    # ------------------------------------------------------------------ execute (HaloExchange.h:151-290)
    def _op(self, op, field, parallel_dim=0, buffer=None):
        dt, ptr, rank, shape, strides, on_dev, _ = _describe(field)
        shp = (C.c_int * rank)(*shape)
        strd = (C.c_longlong * rank)(*strides)
        bptr = None
        if buffer is not None:
            bptr = buffer.data_ptr() if _is_torch(buffer) else buffer.ctypes.data
        if on_dev and _is_torch(field):
            # the tensor was produced on torch's current stream and is consumed there: order the object's stream
            # after it, and torch's stream after the operation (events only)
            with _lib.torch_stream_order(HX_stream(self._h)):
                _lib.check(HX_field_op(self._h, op, dt, ptr, rank, shp, strd, int(parallel_dim), bptr, int(on_dev)))
        else:
            _lib.check(HX_field_op(self._h, op, dt, ptr, rank, shp, strd, int(parallel_dim), bptr, int(on_dev)))

    def var_size(self, field, parallel_dim=0):
        shape = list(field.shape)
        del shape[parallel_dim]
        return int(np.prod(shape)) if shape else 1

    def execute(self, field, parallel_dim=0):
        if not self._is_setup:
            raise _lib.AtlasAmdError("HaloExchange was not setup")
        if getattr(self, "_comm", None) is not None:
            return self._execute_comm(field, parallel_dim, adjoint=False)
        if getattr(self, "_dist", False) and self.nproc() > 1:
            from .parallel_torch import execute_over_torch
            return execute_over_torch(self, field, parallel_dim, adjoint=False)
        self._op(OP_EXECUTE, field, parallel_dim)
        return field

    def execute_adjoint(self, field, parallel_dim=0):
        if not self._is_setup:
            raise _lib.AtlasAmdError("HaloExchange was not setup")
        if getattr(self, "_comm", None) is not None:
            return self._execute_comm(field, parallel_dim, adjoint=True)
        if getattr(self, "_dist", False) and self.nproc() > 1:
            from .parallel_torch import execute_over_torch
            return execute_over_torch(self, field, parallel_dim, adjoint=True)
        self._op(OP_ADJOINT, field, parallel_dim)
        return field

    def pack(self, field, sendbuf, parallel_dim=0):
        self._op(OP_PACK, field, parallel_dim, sendbuf)

    def unpack(self, recvbuf, field, parallel_dim=0):
        self._op(OP_UNPACK, field, parallel_dim, recvbuf)

    def pack_adjoint(self, field, buf, parallel_dim=0):
        self._op(OP_PACK_ADJ, field, parallel_dim, buf)

    def unpack_adjoint(self, buf, field, parallel_dim=0):
        self._op(OP_UNPACK_ADJ, field, parallel_dim, buf)

    def zero_halos(self, field, parallel_dim=0):
        self._op(OP_ZERO, field, parallel_dim)

    def synchronize(self):
        _lib.check(HX_sync(self._h))

    def use_torch_stream(self):
        import torch
        _lib.check(HX_set_stream(self._h, torch.cuda.current_stream().cuda_stream))

    def _execute_comm(self, field, parallel_dim, adjoint):
        """HaloExchange::execute between the ranks of the communicator, inside the library: pack kernel -> grouped
        send/recv per peer -> unpack kernel on the object's stream"""
        dt, ptr, rank, shape, strides, on_dev, _ = _describe(field)
        if not on_dev:
            raise TypeError("halo exchange between ranks needs a device (HIP) tensor")
        shp = (C.c_int * rank)(*shape)
        strd = (C.c_longlong * rank)(*strides)
        with _lib.torch_stream_order(HX_stream(self._h)):
            _lib.check(HX_execute_comm(self._h, self._comm._h, dt, ptr, rank, shp, strd, int(parallel_dim), int(adjoint)))
        return field

    # ------------------------------------------------------------------ C-interface style entry points
    def execute_strided(self, field, var_strides, var_shape, adjoint=False):
        """atlas__HaloExchange__execute[_adjoint]_strided_<T>(This, field, var_strides, var_shape, var_rank)"""
        name = {np.dtype(np.int32): "int", np.dtype(np.int64): "long", np.dtype(np.float32): "float",
                np.dtype(np.float64): "double"}[field.dtype]
        vs = (C.c_int * max(len(var_strides), 1))(*var_strides)
        vsh = (C.c_int * max(len(var_shape), 1))(*var_shape)
        fn = HX_strided[("_adjoint" if adjoint else "", name)]
        _lib.check(fn(self._h, field.ctypes.data, vs, vsh, len(var_shape)))
