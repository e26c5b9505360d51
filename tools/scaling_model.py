"""Per-rank cost record and a predicted 1 -> 8 curve, measured on ONE GPU.

    python tools/scaling_model.py [--out profiles/r05_scaling_model.json] [--ranks all|ends]

For P = 2, 4, 8 and EVERY rank p of P, alone on the device (the rank's own share, nothing beside it):
  * Legendre stage of the wavenumbers m % P == p              (Trans(nparts=P, part=p, shard="m").legendre_device, HIP events)
  * pack kernel of that rank's intermediate                   (atlas_amd__Trans__pack_probe: kept wavenumbers x 2 nf columns)
  * Fourier stage of latitude band p as the transposed transform runs it: modes read from the P packed runs of the receive buffer
                                                              (atlas_amd__Trans__fourier_packed_probe; beside it the same rows from
                                                               a rank-local intermediate, Trans(shard="band"): `fourier_band_ms`)
and from the library's own message plan (host code, no device): bytes every pair exchanges.  The model of the pipelined call
(csrc/dist_trans.hip: L(i+1) and F(i-1) on the Trans stream beside pack + exchange of i on the communication stream):
    ms per transform = max( max_p L_p + max_p F_p ,  max_p pack_p + X )        X = largest pair message / link rate
(xGMI is point to point: every pair has its own link, the pairs of a GPU run concurrently), transforms/s = 1000 / that,
exposed exchange time = what the communication stream exceeds the compute stream by.  Link rates 50 and 150 GB/s bracket
what RCCL send/recv reaches on one xGMI link (153 GB/s raw per link, MI355X_MICROARCH.md).  A model, not a measurement: the
driver's SCALE run is the measurement; this says where the curve should bend and what each rank launches.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRID, T, NF = "O1280", 1279, 137


def measure_rank(g, sp, P, part, reps=25):
    import torch
    import atlas_amd
    from atlas_amd import _lib
    from atlas_amd.dist import Trans_fourier_packed_probe, Trans_pack_probe
    out = {"part": part}
    # Legendre stage, m-sharded
    tr = atlas_amd.Trans(g, T, profile=True, nparts=P, part=part, shard="m")
    tr.use_torch_stream()
    F = torch.zeros(tr.fourier_size(NF), dtype=torch.float64, device="cuda")
    for _ in range(5):   # (a 1 ms kernel timed over 5 launches after 2 read 15 % slow: the clocks were still ramping)
        tr.legendre_device(T, NF, sp, F)
    torch.cuda.synchronize()
    tr.timings(reset=True)
    for _ in range(reps):
        tr.legendre_device(T, NF, sp, F)
    torch.cuda.synchronize()
    tm = tr.timings()
    out["legendre_ms"] = tm["legendre_ms"] / max(tm["legendre_calls"], 1)
    out["legendre_gflop"] = tr.legendre_flops(NF) / P / 1e9          # the rank's share of SURVEY 8(d)'s 470.74 GF (cost-balanced m)
    out["legendre_frac_of_78.6"] = out["legendre_gflop"] / out["legendre_ms"] / 78.6
    ms, nbytes = C.c_double(0.0), C.c_longlong(0)
    _lib.check(Trans_pack_probe(tr._h, NF, 10, C.byref(ms), C.byref(nbytes)))
    out["pack_ms"], out["pack_bytes"] = ms.value, int(nbytes.value)
    out["pack_GBs_read_plus_write"] = 2 * nbytes.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else None
    # the rank's Fourier stage as the distributed transform runs it: band rows, modes read from the P packed runs
    fms = C.c_double(0.0)
    _lib.check(Trans_fourier_packed_probe(tr._h, NF, reps, C.byref(fms)))
    out["fourier_packed_ms"] = fms.value
    del tr, F
    torch.cuda.empty_cache()
    # Fourier stage of the rank's latitude band
    tb = atlas_amd.Trans(g, T, profile=True, nparts=P, part=part, shard="band")
    tb.use_torch_stream()
    gp = torch.zeros(NF * tb.nb_gridpoints(), dtype=torch.float64, device="cuda")
    for _ in range(5):
        tb.invtrans(NF, sp, gp)
    torch.cuda.synchronize()
    tb.timings(reset=True)
    for _ in range(reps):
        tb.invtrans(NF, sp, gp)
    torch.cuda.synchronize()
    tm = tb.timings()
    out["fourier_band_ms"] = tm["fourier_ms"] / max(tm["fourier_calls"], 1)   # the same rows from a rank-local intermediate (mode "band")
    out["fourier_ms"] = out["fourier_packed_ms"]
    out["band_points"] = int(tb.nb_gridpoints())
    del tb, gp
    torch.cuda.empty_cache()
    return out


def message_plan(P):
    """per-pair bytes of the packed transposition, from the library's host code (no device)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(P), "--dry-run"], capture_output=True,
                       text=True, check=True)
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return {"kept_intermediate_bytes": d["kept_intermediate_bytes"],
            "bytes_leaving_a_gpu_max": d["bytes_leaving_a_gpu_per_transform"]["max"],
            "largest_pair_message_bytes": d["largest_pair_message_bytes"],
            "latitude_bands_rows": d["latitude_bands_rows"], "plan_checks": d["plan_checks"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--ranks", default="all", choices=["all", "ends"])
    a = ap.parse_args()
    import torch
    import atlas_amd
    from helpers import red_spectra
    g = atlas_amd.Grid(GRID)
    sp = torch.from_numpy(red_spectra(T, NF)).cuda()
    # the single-device transform on the same box, for scale
    tr = atlas_amd.Trans(g, T, profile=True)
    tr.use_torch_stream()
    gp = torch.zeros(NF * g.size(), dtype=torch.float64, device="cuda")
    for _ in range(3):
        tr.invtrans(NF, sp, gp)
    torch.cuda.synchronize()
    tr.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(10):
        tr.invtrans(NF, sp, gp)
    torch.cuda.synchronize()
    one_ms = (time.perf_counter() - t0) / 10 * 1e3
    tm = tr.timings()
    res = {"workload": f"TL{T} -> {GRID}, {NF} fields per transform, fp64", "measured_on": "one MI355X, each rank's share alone on the device",
           "single_gpu": {"ms_per_transform": one_ms, "legendre_ms": tm["legendre_ms"] / tm["legendre_calls"],
                          "fourier_ms": tm["fourier_ms"] / tm["fourier_calls"], "transforms_per_s": 1e3 / one_ms},
           "P": {}}
    del tr, gp
    torch.cuda.empty_cache()
    for P in (2, 4, 8):
        parts = range(P) if a.ranks == "all" else sorted({0, P // 2 - 1, P - 1})
        ranks = [measure_rank(g, sp, P, p) for p in parts]
        plan = message_plan(P)
        L = max(r["legendre_ms"] for r in ranks)
        Fm = max(r["fourier_ms"] for r in ranks)
        pk = max(r["pack_ms"] for r in ranks)
        entry = {"ranks": ranks, "message_plan": plan,
                 "slowest_rank": {"legendre_ms": L, "pack_ms": pk, "fourier_ms": Fm},
                 "compute_stream_ms": L + Fm, "predicted": {}}
        for gbs in (50.0, 150.0):
            X = plan["largest_pair_message_bytes"] / (gbs * 1e9) * 1e3
            per = max(L + Fm, pk + X)
            entry["predicted"][f"{gbs:.0f}_GBs_per_link"] = {
                "exchange_ms": X, "comm_stream_ms": pk + X, "ms_per_transform": per, "transforms_per_s": 1e3 / per,
                "exposed_exchange_ms": max(0.0, pk + X - (L + Fm)),
                "efficiency_vs_single_gpu": (1e3 / per) / (P * res["single_gpu"]["transforms_per_s"])}
        res["P"][str(P)] = entry
        print(f"P={P}: slowest rank L {L:.3f} pack {pk:.3f} F {Fm:.3f} ms; predicted "
              + ", ".join(f"{k}: {v['transforms_per_s']:.1f}/s (eff {v['efficiency_vs_single_gpu']:.2f})"
                          for k, v in entry["predicted"].items()), flush=True)
    txt = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    else:
        print(txt)


if __name__ == "__main__":
    main()
