"""Offline analysis of tools/fft_trace.py raw dumps (FFT_TRACE_SAVE=...npz): per-CU timelines (the s_memtime counters of
different CUs are not synchronised: all times are relative to the first wave start on the same CU)."""
import sys, collections
import numpy as np
for path in sys.argv[1:]:
    d = np.load(path); rec = d['rec']; fms = float(d['fourier_ms'])
    hw = rec[:, 0] & np.uint64(0xFFFFFFFF); xcc = ((rec[:, 0] >> np.uint64(32)) & np.uint64(0xF)).astype(int)
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int); cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int)
    sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
    t = rec[:, 1:].astype(np.int64)
    key = (xcc << 12) | (se << 8) | (sh << 4) | cu
    cus = np.unique(key)
    spans = []
    for c in cus:
        s = key == c
        t[s] -= t[s, 0].min()
        spans.append(t[s].max())
    span = float(np.median(spans))
    tick = fms * 1e3 / span   # us per tick
    print(f"{path}: fourier {fms:.3f} ms; {len(rec)} waves on {len(cus)} CUs; median CU span {span:.0f} ticks => {1/tick/1e3:.3f} GHz counter")
    nst = t.shape[1]
    end = t[:, nst - 1]; start = t[:, 0]
    life = (end - start) * tick
    print("  wave life us: mean %.2f p10 %.2f p50 %.2f p90 %.2f" % (life.mean(), *np.percentile(life, [10, 50, 90])))
    names = ["gather", "ph0", "ph1", "ph2", "ph3", "ph4"] if t.shape[1] == 7 else ["gather", "pf issue", "rd issue", "rd return", "c2r+chirp", "bfly", "twiddle", "barrier", "write", "flt req", "barrier", "ph1", "ph2", "ph3+4"]
    for k in range(nst - 1):
        dd = (t[:, k + 1] - t[:, k]) * tick
        print("   %-7s mean %6.2f p10 %6.2f p50 %6.2f p90 %6.2f us" % (names[k] if k < len(names) else str(k), dd.mean(), *np.percentile(dd, [10, 50, 90])))
    print("  avg live waves per CU: %.2f" % ((end - start).sum() / sum(spans)))
    cnt = collections.Counter(); simdh = collections.Counter()
    for c in cus[::8]:
        s = np.where(key == c)[0]
        for x in np.linspace(span * 0.1, span * 0.9, 50):
            alive = s[(start[s] <= x) & (end[s] > x)]
            cnt[len(alive)] += 1
            simdh[tuple(sorted(np.bincount(simd[alive], minlength=4).tolist(), reverse=True))] += 1
    tot = sum(cnt.values())
    print("  live waves per CU (share of samples):", {k: round(v / tot, 3) for k, v in sorted(cnt.items())})
    print("  waves per SIMD patterns:", [(k, round(v / tot, 3)) for k, v in simdh.most_common(6)])
    c0 = cus[len(cus) // 3]; s = np.where(key == c0)[0]
    wg = collections.defaultdict(list)
    for i in s:
        wg[int(start[i]) // 64].append(i)   # waves of one workgroup start within a few ticks
    starts = sorted(wg)
    print("  one CU, consecutive workgroups: (start us, life us, phases of wave 0)")
    for k in starts[20:32]:
        i = wg[k][0]
        print("    %9.2f %6.2f  " % (start[i] * tick, life[i]), " ".join("%5.2f" % ((t[i, j + 1] - t[i, j]) * tick) for j in range(nst - 1)), " nwaves", len(wg[k]))
