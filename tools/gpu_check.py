"""Dev tool: run on the GPU box.  Stage-level and end-to-end parity of the HIP path against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atlas_amd, oracle


def spectra(T, nf, seed=20251114, trc=None):
    trc = T if trc is None else trc
    rng = np.random.default_rng(seed)
    sp = np.zeros(((trc + 1) * (trc + 2) // 2, 2, nf))
    k = 0
    for m in range(trc + 1):
        for n in range(m, trc + 1):
            sp[k] = rng.standard_normal((2, nf)) * (1 + n) ** (-5. / 6.)
            if m == 0:
                sp[k, 1] = 0.
            k += 1
    return sp.reshape(-1)


def relrms(a, b):
    d = a - b
    mx = np.abs(b).max()
    return 0. if mx == 0 else float(np.sqrt((d * d).mean()) / mx)


def check(gridname, T, nf, full_oracle=True, nrows=12):
    g = atlas_amd.Grid(gridname)
    t0 = time.time()
    tr = atlas_amd.Trans(g, T)
    t1 = time.time()
    sp = spectra(T, nf)
    sp_d = torch.from_numpy(sp).cuda()
    gp_d = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    tr.invtrans(nf, sp_d, gp_d)
    tr.synchronize()
    gp = gp_d.cpu().numpy()
    print(f"[{gridname} T{T} nf={nf}] setup {t1 - t0:.2f}s  finite={np.isfinite(gp).all()} absmax={np.abs(gp).max():.3g}")
    if full_oracle:
        op = oracle.OraclePlan(T, g.nx(), g.y())
        # stage 1: Legendre
        RP = tr.fourier_row_pitch(nf)
        F_d = torch.zeros(tr.fourier_size(nf), dtype=torch.float64, device="cuda")
        tr.legendre_device(T, nf, sp_d, F_d)
        tr.synchronize()
        F = F_d.cpu().numpy().reshape(g.ny(), T + 1, RP)
        ref = op.legendre(nf, sp)  # [f][lat][m][2]
        mine = F[:, :, :2 * nf].reshape(g.ny(), T + 1, nf, 2).transpose(2, 0, 1, 3)
        # only compare where m is kept for that row
        nlat0 = op.nlat0
        nl = g.ny()
        mask = np.zeros((nl, T + 1), dtype=bool)
        for m in range(T + 1):
            for j in range(nl):
                jl = j if j < nl // 2 else nl - 1 - j
                mask[j, m] = jl >= nlat0[m]
        a = mine[:, mask, :]
        b = ref[:, mask, :]
        print("   legendre stage rel-rms", relrms(a, b), "max abs", np.abs(a - b).max())
        gref = op.invtrans(nf, sp)
        print("   end-to-end vs oracle rel-rms", relrms(gp, gref), "max abs", np.abs(gp - gref).max())
        # FFT stage alone from the oracle's intermediate
    else:
        op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
        rows = np.unique(np.linspace(0, g.ny() - 1, nrows).astype(int))
        res = op.invtrans_rows(nf, sp, rows, use_fft=True)
        off = np.concatenate([[0], np.cumsum(g.nx())])
        worst = 0
        for r, ref in zip(rows, res):
            mine = np.stack([gp[f * g.size() + off[r]: f * g.size() + off[r + 1]] for f in range(nf)])
            e = relrms(mine, ref)
            worst = max(worst, e)
        print("   sampled rows", list(rows), "worst rel-rms", worst)
    return tr


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), atlas_amd._lib.version().decode())
    which = sys.argv[1:] or ["small"]
    if "small" in which:
        check("O64", 63, 1)
        check("O64", 63, 3)
        check("F64", 63, 2)
        check("O32", 31, 20)
    if "mid" in which:
        check("O160", 159, 60)
    if "big" in which:
        check("O640", 639, 137, full_oracle=False)
    if "huge" in which:
        check("O1280", 1279, 137, full_oracle=False, nrows=8)


def timeit(gridname, T, nf, iters=5):
    g = atlas_amd.Grid(gridname)
    t0 = time.time()
    tr = atlas_amd.Trans(g, T, profile=True)
    print(f"[time {gridname} T{T} nf={nf}] setup {time.time() - t0:.1f}s table {tr.legendre_table_bytes() / 1e9:.2f} GB")
    sp_d = torch.from_numpy(spectra(T, nf)).cuda()
    gp_d = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    for _ in range(2):
        tr.invtrans(nf, sp_d, gp_d)
    tr.synchronize()
    tr.timings(reset=True)
    t0 = time.time()
    for _ in range(iters):
        tr.invtrans(nf, sp_d, gp_d)
    tr.synchronize()
    wall = (time.time() - t0) / iters
    tm = tr.timings()
    lm, fm = tm["legendre_ms"] / tm["legendre_calls"], tm["fourier_ms"] / tm["fourier_calls"]
    fl = tr.legendre_flops(nf)
    print(f"   wall {wall * 1e3:.2f} ms/call  legendre {lm:.2f} ms ({fl / lm / 1e9:.1f} TF/s)  fourier {fm:.2f} ms "
          f"({(tr.fourier_size(nf) * 8 * 0.82 + gp_d.numel() * 8) / fm / 1e9:.2f} TB/s alg.)  -> {1 / wall:.1f} transforms/s")


if __name__ == "__main__":
    if "time" in sys.argv[1:]:
        timeit("O160", 159, 60)
        timeit("O640", 639, 137)
        timeit("O1280", 1279, 137)
