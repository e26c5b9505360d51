"""Stand-in for atlas_amd.Trans in the CPU tests of the DRIVERS (bench.py, atlas_amd/dist.py): same constructor
arguments, sharding semantics and buffer layouts, but the "transform" is a trivial linear map on CPU tensors

    gp[f, every point of row j] = scale(sp) * sum_m v(j, m, f),      v(j, m, f) = (j + 1) / 64 + m / 4096 + f

so that the decompositions can be compared with each other exactly like the real ones (every path builds the same
[row][m][field] array before summing).  It never stands in for the product: only tests/test_bench_logic.py uses it."""
import numpy as np
import torch

from atlas_amd import _lib


def latitude_bands(nx, nparts):
    """whole-row bands, each row with the equal_bands part of its first point (trans_plan.cpp: latitude_bands)"""
    off = np.concatenate([[0], np.cumsum(nx)])
    b = [len(nx)] * (nparts + 1)
    b[0] = 0
    prev = 0
    for j in range(len(nx)):
        part = int(off[j] * nparts // off[-1])
        for q in range(prev + 1, part + 1):
            b[q] = j
        prev = max(prev, part)
    return np.array(b, dtype=np.int32)


class FakeTrans:
    corrupt_mirror = False      # set by a test: the mirror-band output is wrong -> the bench must fall back

    def __init__(self, grid, truncation, profile=False, nparts=1, part=0, legendre_cache=None, shard="m", rows=None,
                 tables=None):
        import atlas_amd
        self.grid = atlas_amd.Grid(grid) if isinstance(grid, str) else grid
        self.T, self.nparts, self.part, self.shard, self.rows = int(truncation), int(nparts), int(part), shard, rows
        self.nx = np.asarray(self.grid.nx())
        self.ny = len(self.nx)
        self.off = np.concatenate([[0], np.cumsum(self.nx)])
        self._calls = {"legendre": 0, "fourier": 0}
        if shard == "mirror":
            b = np.zeros(self.nparts + 1, dtype=np.int32)
            _lib.check(_lib.mirror_bands(self.grid._h, self.nparts, b.ctypes.data))
            self._mirror = (int(b[self.part]), int(b[self.part + 1]))
        self._bands = latitude_bands(self.nx, self.nparts)

    # ---- what the drivers call
    def use_torch_stream(self):
        pass

    def synchronize(self):
        pass

    def stream(self):
        return None

    def truncation(self):
        return self.T

    def bands(self):
        return self._bands

    def mirror_rows(self):
        return self._mirror

    def owned_rows(self):
        if self.shard == "mirror":
            b0, b1 = self._mirror
            return np.concatenate([np.arange(b0, b1), np.arange(self.ny - b1, self.ny - b0)])
        if self.rows is not None:
            return np.arange(int(self.rows[0]), int(self.rows[1]))
        return np.arange(int(self._bands[self.part]), int(self._bands[self.part + 1]))

    def nb_gridpoints(self):
        return int(sum(self.nx[j] for j in self.owned_rows()))

    def nb_gridpoints_global(self):
        return int(self.off[-1])

    def nlat0(self):
        return np.zeros(self.T + 1, dtype=np.int32)

    def legendre_flops(self, nf):
        return 1.0e9 * nf

    def timings(self, reset=False):
        out = {"legendre_ms": 2.0 * self._calls["legendre"], "legendre_calls": self._calls["legendre"],
               "fourier_ms": 3.0 * self._calls["fourier"], "fourier_calls": self._calls["fourier"]}
        if reset:
            self._calls = {"legendre": 0, "fourier": 0}
        return out

    # ---- the stand-in arithmetic
    @staticmethod
    def _scale(sp):
        return 1.0 + float(sp.reshape(-1)[0])

    def _v(self, rows, ms, nf):
        j = np.asarray(rows, dtype=np.float64)[:, None, None]
        m = np.asarray(ms, dtype=np.float64)[None, :, None]
        f = np.arange(nf, dtype=np.float64)[None, None, :]
        return (j + 1.0) / 64.0 + m / 4096.0 + f                    # [row][m][field]

    def _store_rows(self, nf, values, gp):
        """values[row_local][field] -> gp[field][points of the owned rows]"""
        rows = self.owned_rows()
        out = gp.view(nf, -1)
        first = 0
        for r, j in enumerate(rows):
            n = int(self.nx[j])
            out[:, first:first + n] = torch.from_numpy(values[r])[:, None]
            first += n
        assert first == out.shape[1]

    def invtrans(self, nf, sp, gp, *args):
        assert not args, "the stand-in only has the scalar path"
        self._calls["legendre"] += 1
        self._calls["fourier"] += 1
        v = np.ascontiguousarray(self._v(self.owned_rows(), np.arange(self.T + 1), nf) * self._scale(sp))
        self._store_rows(nf, v.sum(axis=1), gp)
        if self.shard == "mirror" and FakeTrans.corrupt_mirror:
            gp.view(-1)[0] += 1.0
        return gp

    # ---- stage API of the wavenumber-sharded decomposition (layouts of include/atlas_amd.h)
    def owned_wavenumbers(self):
        return len(range(self.part, self.T + 1, self.nparts))

    def fourier_row_pitch(self, nf):
        return (2 * nf + 15) // 16 * 16

    def fourier_size(self, nf):
        return self.ny * self.owned_wavenumbers() * self.fourier_row_pitch(nf)

    def legendre_device(self, truncation_in, nf, sp, F):
        self._calls["legendre"] += 1
        ms = np.arange(self.part, self.T + 1, self.nparts)
        RP = self.fourier_row_pitch(nf)
        Fv = F.view(self.ny, len(ms), RP)
        Fv.zero_()
        Fv[:, :, 0:2 * nf:2] = torch.from_numpy(self._v(np.arange(self.ny), ms, nf) * self._scale(sp))

    def fourier_device(self, nf, nb_vordiv, parts, part_cnt, gp):
        self._calls["fourier"] += 1
        rows = self.owned_rows()
        RP = self.fourier_row_pitch(nf)
        v = np.zeros((len(rows), self.T + 1, nf))
        for p, (buf, cnt) in enumerate(zip(parts, part_cnt)):
            piece = buf[:len(rows) * cnt * RP].view(len(rows), cnt, RP).numpy()
            v[:, p::len(parts), :] = piece[:, :, 0:2 * nf:2]
        self._store_rows(nf, np.ascontiguousarray(v).sum(axis=1), gp)
