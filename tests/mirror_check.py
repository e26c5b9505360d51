"""child process of tests/test_gpu_zz_device_tables.py::test_mirror_band_sharding_reproduces_single_device_result:
python tests/mirror_check.py <grid> <T> <nf> <nparts>   (GPU needed)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import atlas_amd  # noqa: E402
from helpers import red_spectra  # noqa: E402


def main():
    gridname, T, nf, nparts = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    g = atlas_amd.Grid(gridname)
    npts = g.size()
    off = np.concatenate([[0], np.cumsum(g.nx())])
    sp = red_spectra(T, nf, seed=21)
    vor, div = red_spectra(T, 2, seed=22), red_spectra(T, 2, seed=23)
    tr1 = atlas_amd.Trans(g, T)
    ref = np.zeros(nf * npts)
    tr1.invtrans(nf, sp, ref)
    ref_uv = np.zeros((nf + 4) * npts)
    tr1.invtrans(nf, sp, 2, vor, div, ref_uv)
    ref, ref_uv = ref.reshape(nf, -1), ref_uv.reshape(nf + 4, -1)
    assert np.abs(ref).max() > 0
    seen = []
    for part in range(nparts):
        tr = atlas_amd.Trans(g, T, nparts=nparts, part=part, shard="mirror")
        rows = tr.owned_rows()
        cols = np.concatenate([np.arange(off[j], off[j + 1]) for j in rows])
        assert tr.nb_gridpoints() == cols.size, (part, tr.nb_gridpoints(), cols.size)
        gp = np.full(nf * cols.size, np.nan)
        tr.invtrans(nf, sp, gp)
        assert np.array_equal(gp.reshape(nf, -1), ref[:, cols]), ("scalar", part)
        gp = np.full((nf + 4) * cols.size, np.nan)
        tr.invtrans(nf, sp, 2, vor, div, gp)
        assert np.array_equal(gp.reshape(nf + 4, -1), ref_uv[:, cols]), ("vordiv", part)
        seen.append(rows)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(g.ny()))
    print("MIRROR OK", gridname, T, nf, nparts)


if __name__ == "__main__":
    main()
