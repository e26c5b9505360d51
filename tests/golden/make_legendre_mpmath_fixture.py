"""Writes tests/golden/legendre_mpmath.json: normalised associated Legendre functions (1/2 * int P^2 dmu = 1, no
Condon-Shortley phase: LegendrePolynomials.cc:29, test_transgeneral.cc:120-131) at sampled (n, m, latitude) with n up to
1280, evaluated with mpmath in 60-digit arithmetic by the standard three-term recurrence in n at fixed m -- a different
recurrence, in a different precision, from the one the oracle and the product share (Belousov's, LegendrePolynomials.cc:
136-149).  Independent pin of the tables at high degree.  Needs mpmath (build container)."""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def pbar(n, m, lat_rad):
    x, c = mp.sin(lat_rad), mp.cos(lat_rad)
    p = mp.mpf(1)
    for k in range(1, m + 1):
        p *= mp.sqrt(mp.mpf(2 * k + 1) / (2 * k)) * c
    if n == m:
        return p
    pm2, pm1 = p, mp.sqrt(2 * m + 3) * x * p
    for l in range(m + 2, n + 1):
        a = mp.sqrt(mp.mpf(4 * l * l - 1) / (l * l - m * m))
        b = mp.sqrt(mp.mpf((l - 1) ** 2 - m * m) / (4 * (l - 1) ** 2 - 1))
        pm2, pm1 = pm1, a * (x * pm1 - b * pm2)
    return pm1


rng = np.random.default_rng(20260927)
samples = [(1, 0), (1, 1), (2, 1), (3, 3), (45, 45), (63, 0), (64, 63)]
for n in (159, 319, 639, 1000, 1279, 1280):
    samples += [(n, 0), (n, 1), (n, n // 2), (n, n - 1), (n, n)]
while len(samples) < 120:
    n = int(rng.integers(2, 1281))
    samples.append((n, int(rng.integers(0, n + 1))))
lats_deg = [89.9, 85.2345678, 63.5, 45.0, 20.123, 3.0, 0.0351293, -37.7]
out = []
for i, (n, m) in enumerate(samples):
    for lat in (lats_deg[i % len(lats_deg)], lats_deg[(3 * i + 1) % len(lats_deg)]):
        lat_rad = float(np.deg2rad(lat))          # the double the oracle gets; evaluated exactly at that double
        out.append({"n": n, "m": m, "lat_rad": lat_rad, "value": float(pbar(n, m, mp.mpf(lat_rad)))})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "legendre_mpmath.json")
json.dump({"normalisation": "1/2 * integral of P^2 over mu = 1, no Condon-Shortley phase", "dps": 60, "samples": out},
          open(path, "w"))
print(len(out), "samples ->", path)
