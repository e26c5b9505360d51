"""Host-side mirror of atlas::parallel::HaloExchange (reference: src/atlas/parallel/HaloExchange.h:40-148).

setup() is the reference's index logic (C++ host code behind the C ABI); pack / unpack / adjoint run as HIP kernels;
the peer-to-peer step uses torch.distributed (backend "nccl" = RCCL send/recv over xGMI on MI355X nodes) where the
reference uses eckit::mpi iSend/iReceive (HaloExchange.h:333-369).  A single process needs no transport at all."""
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


def exchange_packed(outbuf, inbuf, out_cnt, out_dsp, in_cnt, in_dsp, var_size, group=None):
    """the communication step of HaloExchange::execute (HaloExchange.h:191-215: iReceive / iSend per peer with
    counts and displacements scaled by var_size, :318-331) on packed buffers: one batched send/recv per peer over
    torch.distributed (RCCL on device tensors; any backend in tests), the rank's own part (periodic / pole duplicates)
    as a local copy"""
    import torch.distributed as dist
    me, ops = dist.get_rank(group), []
    for peer in range(len(out_cnt)):
        o = outbuf[int(out_dsp[peer]) * var_size:int(out_dsp[peer] + out_cnt[peer]) * var_size]
        i = inbuf[int(in_dsp[peer]) * var_size:int(in_dsp[peer] + in_cnt[peer]) * var_size]
        if peer == me:
            i.copy_(o)
            continue
        g = dist.get_global_rank(group, peer) if group is not None else peer
        if i.numel():
            ops.append(dist.P2POp(dist.irecv, i, g, group=group))
        if o.numel():
            ops.append(dist.P2POp(dist.isend, o, g, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


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
        import torch
        import torch.distributed as dist
        group = None if comm is True else comm
        nproc, me = dist.get_world_size(group), dist.get_rank(group)
        _lib.check(HX_setup_begin(self._h, nproc, me, part.ctypes.data, ridx.ctypes.data, int(base), int(size),
                                  int(halo_begin)))
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        recvcounts = self._get("recvcounts", nproc)
        sendcounts_t = torch.zeros(nproc, dtype=torch.int32, device=dev)
        dist.all_to_all_single(sendcounts_t, torch.from_numpy(recvcounts).to(dev), group=group)   # allToAll, :118
        sendcounts = sendcounts_t.cpu().numpy().astype(np.int32)
        req = self._get("send_requests", HX_recvcnt(self._h))
        recv_req_t = torch.zeros(int(sendcounts.sum()), dtype=torch.int32, device=dev)
        dist.all_to_all_single(recv_req_t, torch.from_numpy(req).to(dev), output_split_sizes=sendcounts.tolist(),
                               input_split_sizes=recvcounts.tolist(), group=group)                # allToAllv, :156
        recv_req = np.ascontiguousarray(recv_req_t.cpu().numpy().astype(np.int32))
        _lib.check(HX_setup_finish(self._h, np.ascontiguousarray(sendcounts).ctypes.data, recv_req.ctypes.data))
        self._group = group
        self._dist = True
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
        _lib.check(HX_field_op(self._h, op, dt, ptr, rank, shp, strd, int(parallel_dim), bptr, int(on_dev)))

    def var_size(self, field, parallel_dim=0):
        shape = list(field.shape)
        del shape[parallel_dim]
        return int(np.prod(shape)) if shape else 1

    def execute(self, field, parallel_dim=0):
        if not self._is_setup:
            raise _lib.AtlasAmdError("HaloExchange was not setup")
        if getattr(self, "_dist", False) and self.nproc() > 1:
            return self._execute_distributed(field, parallel_dim, adjoint=False)
        self._op(OP_EXECUTE, field, parallel_dim)
        return field

    def execute_adjoint(self, field, parallel_dim=0):
        if not self._is_setup:
            raise _lib.AtlasAmdError("HaloExchange was not setup")
        if getattr(self, "_dist", False) and self.nproc() > 1:
            return self._execute_distributed(field, parallel_dim, adjoint=True)
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

    def _execute_distributed(self, field, parallel_dim, adjoint):
        """pack -> RCCL send/recv per peer (self part: device copy) -> unpack, on torch's current stream"""
        import torch
        import torch.distributed as dist
        if not (_is_torch(field) and field.is_cuda):
            raise TypeError("distributed halo exchange needs a CUDA (HIP) tensor")
        self.use_torch_stream()
        p = self.plan()
        vs = self.var_size(field, parallel_dim)
        out_cnt, in_cnt = (p["recvcounts"], p["sendcounts"]) if adjoint else (p["sendcounts"], p["recvcounts"])
        out_dsp, in_dsp = (p["recvdispls"], p["senddispls"]) if adjoint else (p["senddispls"], p["recvdispls"])
        outbuf = torch.empty(int(out_cnt.sum()) * vs, dtype=field.dtype, device=field.device)
        inbuf = torch.empty(int(in_cnt.sum()) * vs, dtype=field.dtype, device=field.device)
        (self.pack_adjoint if adjoint else self.pack)(field, outbuf, parallel_dim)
        exchange_packed(outbuf, inbuf, out_cnt, out_dsp, in_cnt, in_dsp, vs, self._group)
        if adjoint:
            self.unpack_adjoint(inbuf, field, parallel_dim)
            self.zero_halos(field, parallel_dim)
        else:
            self.unpack(inbuf, field, parallel_dim)
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


def smoke_halo():
    """tiny single-process halo exchange on cuda:0 (periodic duplicates), checked against the oracle"""
    import torch
    from oracle.halo import HaloExchangeOracle
    n = 40
    part = np.zeros(n, dtype=np.int32)
    ridx = np.arange(n, dtype=np.int32)
    ridx[32:] = np.arange(8)          # nodes 32..39 are periodic copies of nodes 0..7
    hx = HaloExchange()
    hx.setup(part, ridx, 0, n)
    f = torch.arange(n * 5, dtype=torch.float64, device="cuda").reshape(n, 5).contiguous()
    ref = f.cpu().numpy().copy()
    orc = [HaloExchangeOracle(0, 1)]
    HaloExchangeOracle.setup(orc, [part], [ridx], 0, [n])
    HaloExchangeOracle.execute(orc, [ref])
    hx.execute(f)
    hx.synchronize()
    assert np.array_equal(f.cpu().numpy(), ref), "smoke: halo exchange mismatch"
