"""torch.distributed transport for atlas_amd.parallel.HaloExchange -- NOT the product path.

On MI355X nodes the halo exchange between ranks runs inside the library over its RCCL communicator
(atlas_amd/parallel.py -> csrc/halo_exchange.hip: execute_comm).  This module keeps the earlier implementation, where
the two collectives of the setup and the per-peer send/recv go through torch.distributed, so that the index logic can be
tested between real processes on CPUs (gloo) -- tests/test_dist_plan.py."""
import numpy as np

from . import _lib
from .parallel import HX_recvcnt, HX_setup_begin, HX_setup_finish


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




def setup_over_torch(hx, part, ridx, base, size, halo_begin, group):
    """the reference setup with its two collective steps over torch.distributed (HaloExchange.cc:118,156)"""
    import torch
    import torch.distributed as dist
    nproc, me = dist.get_world_size(group), dist.get_rank(group)
    _lib.check(HX_setup_begin(hx._h, nproc, me, part.ctypes.data, ridx.ctypes.data, int(base), int(size),
                              int(halo_begin)))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    recvcounts = hx._get("recvcounts", nproc)
    sendcounts_t = torch.zeros(nproc, dtype=torch.int32, device=dev)
    dist.all_to_all_single(sendcounts_t, torch.from_numpy(recvcounts).to(dev), group=group)   # allToAll, :118
    sendcounts = sendcounts_t.cpu().numpy().astype(np.int32)
    req = hx._get("send_requests", HX_recvcnt(hx._h))
    recv_req_t = torch.zeros(int(sendcounts.sum()), dtype=torch.int32, device=dev)
    dist.all_to_all_single(recv_req_t, torch.from_numpy(req).to(dev), output_split_sizes=sendcounts.tolist(),
                           input_split_sizes=recvcounts.tolist(), group=group)                # allToAllv, :156
    recv_req = np.ascontiguousarray(recv_req_t.cpu().numpy().astype(np.int32))
    if recv_req.size == 0:
        recv_req = np.zeros(1, dtype=np.int32)
    _lib.check(HX_setup_finish(hx._h, np.ascontiguousarray(sendcounts).ctypes.data, recv_req.ctypes.data))
    hx._group = group
    hx._dist = True
    hx._is_setup = True


def execute_over_torch(hx, field, parallel_dim, adjoint):
    """pack -> send/recv per peer over torch.distributed (self part: device copy) -> unpack, on torch's current stream"""
    import torch
    from .parallel import _is_torch
    if not (_is_torch(field) and field.is_cuda):
        raise TypeError("distributed halo exchange needs a CUDA (HIP) tensor")
    hx.use_torch_stream()
    p = hx.plan()
    vs = hx.var_size(field, parallel_dim)
    out_cnt, in_cnt = (p["recvcounts"], p["sendcounts"]) if adjoint else (p["sendcounts"], p["recvcounts"])
    out_dsp, in_dsp = (p["recvdispls"], p["senddispls"]) if adjoint else (p["senddispls"], p["recvdispls"])
    outbuf = torch.empty(int(out_cnt.sum()) * vs, dtype=field.dtype, device=field.device)
    inbuf = torch.empty(int(in_cnt.sum()) * vs, dtype=field.dtype, device=field.device)
    (hx.pack_adjoint if adjoint else hx.pack)(field, outbuf, parallel_dim)
    exchange_packed(outbuf, inbuf, out_cnt, out_dsp, in_cnt, in_dsp, vs, hx._group)
    if adjoint:
        hx.unpack_adjoint(inbuf, field, parallel_dim)
        hx.zero_halos(field, parallel_dim)
    else:
        hx.unpack(inbuf, field, parallel_dim)
    return field
