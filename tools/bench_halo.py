"""Dev tool / evidence (GPU box): halo pack / unpack kernel rates for a StructuredColumns field (size_halo, levels) fp64 on
O1280, row-band partitions emulated one rank at a time (halo = 2 x packed bytes, HBM-bound gather/scatter).
    python tools/bench_halo.py [grid] [levels]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, atlas_amd
from atlas_amd.functionspace import StructuredColumns
from atlas_amd.parallel import HaloExchange

grid = sys.argv[1] if len(sys.argv) > 1 else "O1280"
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 137
g = atlas_amd.Grid(grid)
out = []
for nparts, halo in ((4, 1), (8, 1), (8, 2), (1, 2)):
    part = nparts // 2
    fss = [StructuredColumns(g, halo=halo, periodic_points=True, nparts=nparts, part=p, distribution="row_bands")
           for p in range(nparts)]
    hxs = [f.begin_halo_exchange() for f in fss]
    HaloExchange.finish_emulated(hxs)
    fs, hx = fss[part], hxs[part]
    plan = hx.plan()
    nsend, nrecv = int(plan["sendcounts"].sum()), int(plan["recvcounts"].sum())
    field = torch.zeros((fs.sizeHalo(), lev), dtype=torch.float64, device="cuda")
    sbuf = torch.zeros(max(nsend, 1) * lev, dtype=torch.float64, device="cuda")
    rbuf = torch.zeros(max(nrecv, 1) * lev, dtype=torch.float64, device="cuda")
    hx.use_torch_stream()
    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    tp = timed(lambda: hx.pack(field, sbuf))
    tu = timed(lambda: hx.unpack(rbuf, field))
    # cold calls: the touched rows of the warm loop above (tens of MB) sit in the 256 MiB Infinity Cache; before every
    # timed call a 1 GiB buffer is overwritten (evicts L2 and the Infinity Cache), the call is timed alone with events
    flush = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    def cold(fn, reps=9):
        ts = []
        for i in range(reps):
            flush.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        return float(np.median(ts))
    cp = cold(lambda: hx.pack(field, sbuf))
    cu = cold(lambda: hx.unpack(rbuf, field))
    del flush
    rec = {"grid": grid, "levels": lev, "nparts": nparts, "part": part, "halo": halo, "size_owned": fs.sizeOwned(),
           "size_halo": fs.sizeHalo(), "send_nodes": nsend, "recv_nodes": nrecv,
           "pack_us": tp * 1e6, "unpack_us": tu * 1e6,
           "pack_GBs": 2 * nsend * lev * 8 / tp / 1e9, "unpack_GBs": 2 * nrecv * lev * 8 / tu / 1e9,
           "cold_pack_us": cp * 1e6, "cold_unpack_us": cu * 1e6,
           "cold_pack_GBs": 2 * nsend * lev * 8 / cp / 1e9, "cold_unpack_GBs": 2 * nrecv * lev * 8 / cu / 1e9,
           "note": "warm = 20 back-to-back calls on the same field (cache-resident rows); cold = median of 9 single calls, "
                   "each after a 1 GiB overwrite; bytes = 2 x packed bytes (read + write)"}
    out.append(rec)
    print(json.dumps(rec), flush=True)
    del fss, hxs
