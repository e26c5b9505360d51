"""Dev tool (GPU box): per-wavefront timeline of one Fourier row class (library built with HIPFLAGS_EXTRA=-DAA_FFT_TRACE).
    ATLAS_AMD_FFT_ONLY_M=5120 ATLAS_AMD_FFT_STREAMS=1 python tools/fft_trace.py [grid T nf]
Every wavefront of the specialised Bluestein kernel records HW_ID / XCC_ID and s_memtime at its phase boundaries
(fft_kernel.hip: AA_TRACE_STAMP).  Output: residency per CU over time, phase durations, gaps between workgroups on one CU,
waves per SIMD."""
import sys, os, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, atlas_amd
from atlas_amd import _lib
from helpers import red_spectra

grid, T, nf = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("O1280", 1279, 137)
g = atlas_amd.Grid(grid)
tr = atlas_amd.Trans(g, T, profile=True)
sp = torch.from_numpy(red_spectra(T, nf)).cuda()
gp = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
tr.invtrans(nf, sp, gp); tr.synchronize()
WORDS = 8 * 1024 * 1024
_lib.check(_lib.Trans_fft_trace(tr._h, WORDS, None))
tr.timings(reset=True)
tr.invtrans(nf, sp, gp); tr.synchronize()
tm = tr.timings()
buf = np.zeros(WORDS, dtype=np.uint64)
_lib.check(_lib.Trans_fft_trace(tr._h, WORDS, buf.ctypes.data_as(C.c_void_p)))
print("fourier ms", tm["fourier_ms"], "ONLY_M", os.environ.get("ATLAS_AMD_FFT_ONLY_M"))
W = int(os.environ.get('FFT_TRACE_WORDS', '8'))   # 16 for a -DAA_FFT_TRACE_PH0 build
rec = buf.reshape(-1, W)
rec = rec[rec[:, 1] != 0]
if os.environ.get("FFT_TRACE_SAVE"):
    np.savez_compressed(os.environ["FFT_TRACE_SAVE"], rec=rec, fourier_ms=tm["fourier_ms"])
print("wavefront records:", len(rec))
if len(rec) == 0:
    sys.exit(0)
hw = rec[:, 0] & np.uint64(0xFFFFFFFF)
xcc = (rec[:, 0] >> np.uint64(32)) & np.uint64(0xF)
wave_id = hw & np.uint64(0xF); simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF)
sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
t = rec[:, 1:].astype(np.int64)
for x in np.unique(xcc):   # every XCD has its own counter: common origin per XCD (the kernel starts everywhere at once)
    sel = xcc == x
    t[sel] -= t[sel, 0].min()
# stamps not written stay at -t0 (zero before the shift): find the number of valid stamps
valid = (rec[:, 1:] != 0)
nvalid = valid.sum(axis=1)
print("stamps per wave (histogram):", collections.Counter(nvalid.tolist()))
end = np.array([t[i, nvalid[i] - 1] for i in range(len(rec))])
span = end.max()
print("kernel span in s_memtime ticks:", span, " (ticks per ms: %.0f)" % (span / tm["fourier_ms"]))
tick_ns = tm["fourier_ms"] * 1e6 / span
life = end - t[:, 0]
print("wave lifetime: mean %.1f us  p10 %.1f  p50 %.1f  p90 %.1f" % tuple(x * tick_ns / 1e3 for x in (life.mean(), np.percentile(life, 10), np.percentile(life, 50), np.percentile(life, 90))))
names = ["gather+barrier", "phase0", "phase1", "phase2", "phase3", "phase4"] if W == 8 else ["gather+barrier", "prefetch issue", "staging reads issue", "reads return", "c2r+chirp", "butterfly", "twiddles", "barrier", "write", "filter request", "barrier", "phase1", "phase2", "phase3+4"]
nph = int(nvalid.max()) - 1
for k in range(nph):
    ok = nvalid > k + 1
    d = (t[ok, k + 1] - t[ok, k]) * tick_ns / 1e3
    print("  %-15s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f   (%4.1f %% of mean lifetime)" %
          (names[k] if k < len(names) else "phase%d" % (k - 1), d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90),
           100 * d.mean() / (life.mean() * tick_ns / 1e3)))
# residency: per CU (xcc, se, sh, cu) integrate the number of live waves over time
key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
cus = np.unique(key)
print("distinct CUs seen:", len(cus), " distinct (se,sh,cu) per xcc:", len(np.unique(key & 0xFFF)), " xccs:", len(np.unique(xcc)))
tot_live = life.sum()
print("average live waves per CU over the kernel span: %.2f" % (tot_live / span / len(cus)))
# one CU in detail
c0 = cus[len(cus) // 2]
idx = np.where(key == c0)[0]
order = idx[np.argsort(t[idx, 0])]
print("CU %x: %d waves; first 40 (start us, end us, simd, wave slot, stamps):" % (c0, len(idx)))
for i in order[:40]:
    print("   %8.2f %8.2f  simd %d slot %d  %s" % (t[i, 0] * tick_ns / 1e3, end[i] * tick_ns / 1e3, simd[i], wave_id[i],
                                              " ".join("%6.2f" % ((t[i, k + 1] - t[i, k]) * tick_ns / 1e3) for k in range(nvalid[i] - 1))))
# waves per SIMD alive at sample times on that CU
samples = np.linspace(span * 0.2, span * 0.8, 200)
hist = collections.Counter()
for s in samples:
    live = idx[(t[idx, 0] <= s) & (end[idx] > s)]
    cnt = np.bincount(simd[live].astype(np.int64), minlength=4)
    hist[tuple(sorted(cnt.tolist(), reverse=True))] += 1
print("waves per SIMD (sorted) on that CU at 200 sample times:", hist.most_common(8))
live_all = collections.Counter()
for s in samples[::10]:
    alive = (t[:, 0] <= s) & (end > s)
    per_cu = np.bincount(np.searchsorted(cus, key[alive]), minlength=len(cus))
    for v in per_cu:
        live_all[int(v)] += 1
print("live waves per CU, all CUs, 20 sample times:", sorted(live_all.items()))
