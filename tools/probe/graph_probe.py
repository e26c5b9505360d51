"""Probe (GPU box) [r6]: is the device-array transform capturable into a HIP graph (torch.cuda.CUDAGraph on the stream the Trans object
runs on), is a replay bit-identical, and what does a replay cost against a direct call for a small configuration (C2: TL159 -> O160,
60 levels, two kernels of ~ 20 and ~ 60 us) and the headline?   python tools/probe/graph_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402


def run(gridname, T, nf, reps):
    g = atlas_amd.Grid(gridname)
    tr = atlas_amd.Trans(g, T)
    sp = torch.from_numpy(red_spectra(T, nf)).cuda()
    gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        tr.use_torch_stream()
        for _ in range(3):
            tr.invtrans(nf, sp, gp)
        side.synchronize()
        ref = gp.clone()
        t0 = time.perf_counter()
        for _ in range(reps):
            tr.invtrans(nf, sp, gp)
        side.synchronize()
        direct = (time.perf_counter() - t0) / reps * 1e3
        graph = torch.cuda.CUDAGraph()
        gp.zero_()
        try:
            with torch.cuda.graph(graph, stream=side):
                tr.invtrans(nf, sp, gp)
        except Exception as e:   # noqa: BLE001
            print(f"{gridname} T{T} nf={nf}: capture FAILED: {type(e).__name__}: {e}")
            return
        graph.replay()
        side.synchronize()
        same = bool(torch.equal(gp, ref))
        pairs = []
        for _ in range(4):   # alternating blocks
            t0 = time.perf_counter()
            for _ in range(reps):
                tr.invtrans(nf, sp, gp)
            side.synchronize()
            d = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            side.synchronize()
            pairs.append((round(d, 4), round((time.perf_counter() - t0) / reps * 1e3, 4)))
    print(f"{gridname} T{T} nf={nf}: (direct, graph replay) ms per call in alternating blocks {pairs}, bitwise equal {same}")


if __name__ == "__main__":
    run("O160", 159, 60, 300)
    run("O320", 319, 60, 100)
    run("O640", 639, 137, 30)
    run("O1280", 1279, 137, 20)
