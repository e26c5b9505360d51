"""Bench-grade lines for the BASELINE.json configurations other than the headline one (single MI355X):
    python tools/bench_configs.py [--out profiles/r03_configs.jsonl] [--only C2,C5]
One JSON line per configuration, same schema as bench.py (metric / value / unit / ms_per_step / dtype / config.workload /
roofline / roofline_kernels); `value` counts transforms of 137-level fields where the configuration has 137 levels, of its
own level count otherwise (C2: 60).  Inputs resident in HBM, HIP events per stage (Trans profile=True), synthetic red spectra.
Template: the reference's own benchmark driver, src/sandbox/benchmark_trans/atlas-benchmark-trans.cc:257-289.

  C2        TL159  -> O160,  60 levels                      (BASELINE.json configs[1])
  C3x1      TL639  -> O640,  137 levels on ONE device       (configs[2] is a 4-GPU run; this is its single-device workload)
  C4batch   TL1279 -> O1280, 1370 fields in ONE invtrans    (configs[3]: "10 fields" x 137 levels; single device)
  C5        TL1279 -> F1280 ("N1280 full"), 137 levels, fp32 (configs[4]);  C5f64: the same grid in fp64
  C5n       TL1279 -> N1280 (classic reduced Gaussian), fp32;  C5nf64: the same grid in fp64
  C4f32     TL1279 -> O1280, 137 levels, fp32 (the headline grid in the precision of C5; not a BASELINE configuration)
  C4vd      TL1279 -> O1280, nscalar 137 + nvordiv 137 in ONE invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp): the other axis of the
            reference's benchmark (atlas-benchmark-trans.cc:66-76,148-149 `--nscalar --nvordiv`; TransLocal.cc:1523-1597): 411 output
            fields (137 u, 137 v, 137 scalars); stage times incl. the spectra_prepare kernel (extend_truncation + vd2uv + interleave)
  C2vd      TL159 -> O160, nscalar 60 + nvordiv 60;  C4vdf32: C4vd in the fp32 variant
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_MFMA_PEAK, FP32_MFMA_PEAK, HBM_PEAK = 78.6, 157.3, 8000.0   # TFLOP/s, TFLOP/s, GB/s (MI355X_MICROARCH.md)

CONFIGS = {
    "C2": ("O160", 159, 60, False, 200, 20),     # (steps, warm-up): short calls are timed over enough launches for the clocks to settle
    "C3x1": ("O640", 639, 137, False, 50, 10),
    "C4batch": ("O1280", 1279, 1370, False, 3, 1),
    "C5": ("F1280", 1279, 137, True, 10, 3),
    "C5f64": ("F1280", 1279, 137, False, 10, 3),
    "C5n": ("N1280", 1279, 137, True, 10, 3),
    "C5nf64": ("N1280", 1279, 137, False, 10, 3),
    "C4f32": ("O1280", 1279, 137, True, 10, 3),
    "C4vd": ("O1280", 1279, 137, False, 8, 2, 137),
    "C2vd": ("O160", 159, 60, False, 200, 20, 60),
    "C4vdf32": ("O1280", 1279, 137, True, 8, 2, 137),
}


def run(name, grid, T, nf, f32, steps, warmup, nvd=0):
    import numpy as np
    import torch
    import atlas_amd
    from helpers import red_spectra
    g = atlas_amd.Grid(grid)
    t0 = time.time()
    tr = atlas_amd.Trans(g, T, profile=True)
    setup_s = time.time() - t0
    tr.use_torch_stream()
    if nf <= 200:
        sp = torch.from_numpy(red_spectra(T, nf)).cuda()
    else:   # many fields: tile a 137-field block on the device (generating 18 GB on the host takes minutes)
        blk = torch.from_numpy(red_spectra(T, 137)).cuda().reshape(-1, 137)
        sp = blk.repeat(1, (nf + 136) // 137)[:, :nf].contiguous().reshape(-1)
        del blk
    dt_t = torch.float32 if f32 else torch.float64
    ns, nf = nf, nf + 2 * nvd          # a vor/div call transforms 2 nvd wind fields + ns scalars (TransLocal.cc:1523-1597)
    gp = torch.zeros(nf * g.size(), dtype=dt_t, device="cuda")
    if f32:
        sp = sp.to(torch.float32)
    if nvd:
        vor = torch.from_numpy(red_spectra(T, nvd, seed=2)).cuda().to(dt_t)
        div = torch.from_numpy(red_spectra(T, nvd, seed=3)).cuda().to(dt_t)
        call = lambda: tr.invtrans(ns, sp, nvd, vor, div, gp)
    else:
        call = lambda: tr.invtrans(nf, sp, gp)
    for _ in range(warmup):
        call()
    torch.cuda.synchronize()
    tr.timings(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = tr.timings()
    leg_ms = tm["legendre_ms"] / max(tm["legendre_calls"], 1)
    fft_ms = tm["fourier_ms"] / max(tm["fourier_calls"], 1)
    ny = g.ny()
    nlat0 = tr.nlat0()
    kept = float(sum(int(nlat0[m] < ny // 2) * 2 * (ny // 2 - int(nlat0[m])) for m in range(T + 1)))
    esz = 4 if f32 else 8
    leg_flops = tr.legendre_flops(nf)   # (a vor/div call runs this stage at truncation T + 1: + 0.2 % that are not counted)
    fft_bytes = kept * nf * 2 * esz + nf * g.size() * esz
    leg_tf = leg_flops / (leg_ms * 1e-3) / 1e12
    fft_gbs = fft_bytes / (fft_ms * 1e-3) / 1e9
    peak = FP32_MFMA_PEAK if f32 else FP64_MFMA_PEAK
    kernels = [
        {"kernel": "legendre stage (one launch)", "bound": "mfma", "achieved": leg_tf, "peak": peak, "unit": "TFLOP/s",
         "frac": leg_tf / peak, "avg_ms": leg_ms, "traffic": None},
        {"kernel": "Fourier stage (all row classes)", "bound": "hbm", "achieved": fft_gbs, "peak": HBM_PEAK, "unit": "GB/s",
         "frac": fft_gbs / HBM_PEAK, "avg_ms": fft_ms, "traffic": None},
    ]
    try:   # what this box sustains on the Legendre stage's MFMA instruction alone (csrc/diag.hip), beside the datasheet peak
        from atlas_amd import _lib
        sustained = _lib.diag_mfma_f32_rate(25.0, 3) if f32 else _lib.diag_mfma_f64_rate(25.0, 3)
        kernels[0]["peak_sustained_measured"] = sustained
        kernels[0]["frac_of_sustained"] = leg_tf / sustained
    except Exception as e:
        kernels[0]["peak_sustained_measured"] = None
        kernels[0]["peak_sustained_error"] = f"{type(e).__name__}: {e}"
    cands = [k for k in kernels if k["avg_ms"] >= 0.25 * (leg_ms + fft_ms)] or kernels
    dom = min(cands, key=lambda k: k["frac"])
    per137 = nf / 137.0 if nf % 137 == 0 else 1.0
    out = {
        "metric": f"inverse SH transforms/sec (T{T}, {grid}, {nf if per137 == 1.0 else 137} lev)",
        "value": steps * per137 / dt, "unit": "transforms/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if f32 else "f64", "data": "synthetic",
        "config": {"name": name, "workload": f"TransLocal invtrans T{T} -> {grid}, nb_scalar_fields={nf} in one call"
                                             + (f" ({nf // 137} x 137 levels; value counts 137-level transforms)" if per137 != 1.0 else ""),
                   "grid": grid, "truncation": T, "fields": nf, "points": int(g.size()), "setup_s": round(setup_s, 2)},
        "roofline": {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_ms")},
        "roofline_kernels": kernels,
    }
    if nvd:
        prep_ms = tm["prepare_ms"] / max(tm["prepare_calls"], 1)
        ncoef = (T + 1) * (T + 2)
        prep_bytes = (2 * nvd + ns) * ncoef * esz + (2 * nvd + ns) * (T + 2) * (T + 3) * esz   # vor, div, sp read; merged spectra written
        out["metric"] = f"inverse SH transforms/sec (T{T}, {grid}, nscalar {ns} + nvordiv {nvd})"
        out["value"] = steps * (nf / 137.0 if nf % 137 == 0 else 1.0) / dt
        out["config"]["workload"] = (f"TransLocal invtrans(nb_scalar={ns}, sp, nb_vordiv={nvd}, vor, div, gp) T{T} -> {grid}: {nf} output "
                                     f"fields (u, v, scalars) in one call" + ("; value counts 137-level output fields" if nf % 137 == 0 else ""))
        out["config"].update({"nb_scalar": ns, "nb_vordiv": nvd, "fields": nf})
        out["roofline_kernels"].append(
            {"kernel": "spectra_prepare_kernel (extend_truncation + vd2uv + interleave)", "bound": "hbm",
             "achieved": prep_bytes / (prep_ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s",
             "frac": prep_bytes / (prep_ms * 1e-3) / 1e9 / HBM_PEAK, "avg_ms": prep_ms, "traffic": None})
    if nf > 200:   # field k and field k + 137 carry the same spectra: the results must be identical bits
        v = gp.reshape(nf, -1)
        out["config"]["tiled_fields_identical"] = bool(torch.equal(v[3], v[3 + 137]))
    del tr, sp, gp
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    names = a.only.split(",") if a.only else list(CONFIGS)
    lines = []
    for n in names:
        try:
            lines.append(json.dumps(run(n, *CONFIGS[n])))
        except Exception as e:   # one configuration must not cost the others
            lines.append(json.dumps({"config": {"name": n}, "error": f"{type(e).__name__}: {e}"}))
        print(lines[-1], flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
