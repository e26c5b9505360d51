"""Dev tool (GPU box, one rank):
  1. probe RCCL's all_to_all_single with float64 self-messages of growing size (how many elements arrive);
  2. compare DistributedTrans(mode="alltoall") at world_size 1 with the single-device Trans at the bench size
     (lone-rank device copy) and at a small field count with the slab forced through RCCL.
Usage: python tools/dist_selfcheck.py [grid] [T] [nf]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import numpy as np
import torch
import torch.distributed as dist

import atlas_amd
import atlas_amd.dist_torch as aadist
from atlas_amd.dist_torch import DistributedTrans
from helpers import red_spectra


def probe_rccl_self_messages():
    sizes = [1 << 20, 14_680_064, 117_964_800, 1 << 27, (1 << 28) - 1024, (1 << 28) + 1024, (1 << 29) + 1024,
             943_718_400]
    for n in sizes:
        src = torch.arange(n, dtype=torch.float64, device="cuda")
        dst = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
        dist.all_to_all_single(dst, src, output_split_sizes=[n], input_split_sizes=[n])
        torch.cuda.synchronize()
        good = dst == src
        ngood = int(good.sum())
        first_bad = int(torch.nonzero(~good)[0]) if ngood < n else -1
        print(f"rccl self all_to_all_single float64 n={n} ({n * 8 / 2**30:.2f} GiB): delivered {ngood} "
              f"({ngood / n:.4f}), first missing index {first_bad}", flush=True)
        del src, dst, good


def compare(tag, d, tr, g, T, nf, off):
    sp = torch.from_numpy(red_spectra(T, nf)).cuda()
    ref = torch.full((nf * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
    tr.invtrans(nf, sp, ref)
    F, R, plan, RP = d._buffers(nf, 0)
    F.fill_(float("nan"))      # what torch.empty may hold: NaN patterns left by earlier tensors
    R.fill_(float("nan"))
    gp = torch.full((nf * g.size(),), float("nan"), dtype=torch.float64, device="cuda")
    d.invtrans(nf, sp, gp)
    torch.cuda.synchronize()
    bad = (gp != ref) | torch.isnan(gp)
    nbad = int(bad.sum())
    msg = f"{tag}: nf={nf}, reference finite {bool(torch.isfinite(ref).all())}, equal {nbad == 0}, mismatching {nbad}"
    if nbad:
        idx = torch.nonzero(bad).flatten()[:200000].cpu().numpy()
        rows = np.searchsorted(off, idx % g.size(), side="right") - 1
        msg += f"; rows {np.unique(rows)[:8]} (of {len(np.unique(rows))})"
    print(msg, flush=True)
    d._buf.clear()


def main():
    grid = sys.argv[1] if len(sys.argv) > 1 else "O1280"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1279
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 137
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        probe_rccl_self_messages()
    except Exception as e:
        print("probe failed:", type(e).__name__, e, flush=True)
    g = atlas_amd.Grid(grid)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    tr = atlas_amd.Trans(g, T)
    tr.use_torch_stream()
    d = DistributedTrans(g, T, mode="alltoall")
    compare("lone rank, device copy", d, tr, g, T, nf, off)
    aadist.FORCE_RCCL_SINGLE_RANK = True
    compare("lone rank, slab through RCCL all_to_all_single", d, tr, g, T, 14, off)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
