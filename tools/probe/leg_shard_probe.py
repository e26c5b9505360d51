import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, atlas_amd
from helpers import red_spectra
T, nf = 1279, 137
g = atlas_amd.Grid("O1280")
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
for P, part in ((1, 0), (8, 0), (8, 3)):
    tr = atlas_amd.Trans(g, T, profile=True, nparts=P, part=part, shard="m")
    tr.use_torch_stream()
    F = torch.zeros(tr.fourier_size(nf), dtype=torch.float64, device="cuda")
    for _ in range(3):
        tr.legendre_device(T, nf, sp, F)
    torch.cuda.synchronize()
    tr.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(20):
        tr.legendre_device(T, nf, sp, F)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    tm = tr.timings()
    print(f"P={P} part={part}: events {tm['legendre_ms']/tm['legendre_calls']:.3f} ms, wall {wall:.3f} ms", flush=True)
    del tr, F
