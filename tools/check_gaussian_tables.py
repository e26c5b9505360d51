"""Dev-container-only tool: parse the reference's tabulated Gaussian latitudes (plain data,
src/atlas/grid/detail/spacing/gaussian/N*.cc) and compare them element-wise with exactly rounded
Gauss-Legendre nodes computed here in extended precision.  Reads /root/reference; never shipped."""
import re, sys, glob, os
import numpy as np

REF = "/root/reference/src/atlas/grid/detail/spacing/gaussian"

def ref_table(N):
    txt = open(f"{REF}/N{N}.cc").read()
    body = txt[txt.index("DEFINE_GAUSSIAN_LATITUDES"):]
    nums = re.findall(r"(?<![\w.])(\d+\.\d+)", body)
    return nums  # strings

def exact_nodes(N):
    """colatitude Newton in long double on P_{2N}(cos(theta))"""
    ld = np.longdouble
    n = 2 * N
    pi = ld(np.pi) + ld(1.2246467991473532e-16)  # pi to long-double precision
    out = []
    k = np.arange(1, N + 1, dtype=ld)
    z = (4 * k - 1) * pi / (4 * n + 2)
    theta = z + 1 / (np.tan(z) * 8 * ld(n) * ld(n))
    for it in range(8):
        x = np.cos(theta)
        p0 = np.ones_like(x); p1 = x.copy()
        for j in range(2, n + 1):
            p0, p1 = p1, ((2 * j - 1) * x * p1 - (j - 1) * p0) / j
        # p1 = P_n, p0 = P_{n-1};  dP/dx = n (x P_n - P_{n-1})/(x^2-1);  dP/dtheta = -sin * dP/dx
        dpdx = n * (x * p1 - p0) / (x * x - 1)
        dth = p1 / (-np.sin(theta) * dpdx)
        theta = theta - dth
    lat = ld(90) - theta * ld(180) / pi
    return lat

if __name__ == "__main__":
    Ns = sorted(int(os.path.basename(f)[1:-3]) for f in glob.glob(f"{REF}/N[0-9]*.cc"))
    for N in Ns:
        if N > 2000 and "--all" not in sys.argv: continue
        tab = ref_table(N)
        assert len(tab) == N, (N, len(tab))
        lat = exact_nodes(N)
        mism = 0; maxd = 0
        for s, v in zip(tab, lat):
            r = np.round(v * np.longdouble(1e12))
            si = int(s.replace(".", "")) if len(s.split(".")[1]) == 12 else None
            if si is None: 
                mism += 1; continue
            d = abs(int(r) - si)
            maxd = max(maxd, d)
            if d: mism += 1
        print(f"N{N}: entries={N} last-digit mismatches={mism} max|diff|={maxd}e-12 deg")
