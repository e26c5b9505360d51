"""TEST INFRASTRUCTURE ONLY -- Python restatement of functionspace::StructuredColumns halo index construction for
GLOBAL structured grids (xmin = 0, no projection), band distributions.

Reference: src/atlas/functionspace/detail/StructuredColumns_setup.cc:88-663 (owned bounds :125-226, halo bounds
:369-455, point ordering :469-571, fields :583-662, pole reflection compute_j :263-287, 180-degree shift compute_g
:331-355), StructuredColumns_create_remote_index.cc:37-255 (remote index = local index of the point on its owner),
src/atlas/grid/detail/distribution/BandsDistribution.h:32-34 (partition function)."""
import math

import numpy as np


def bands_partition(g, npts, nparts, blocksize=1):
    nb_blocks = (npts + blocksize - 1) // blocksize
    return ((g // blocksize) * nparts) // nb_blocks


class StructuredColumnsOracle:
    def __init__(self, nx, y, halo=0, periodic_points=False, nparts=1, part=0, blocksize=1, distribution=None):
        self.distribution = None if distribution is None else np.asarray(distribution, dtype=np.int64)
        self.nx = [int(v) for v in nx]
        self.y = [float(v) for v in y]
        self.ny = len(self.nx)
        self.halo, self.periodic_points, self.nparts, self.part, self.bs = halo, periodic_points, nparts, part, blocksize
        self.npts = sum(self.nx)
        self.offsets = np.concatenate([[0], np.cumsum(self.nx)]).astype(np.int64)
        self._setup()

    # grid accessors (Structured.h:300-314 with xmin = 0)
    def gx(self, i, j):
        return 0.0 + float(i) * (360.0 / float(self.nx[j]))

    def partition(self, g):
        if self.distribution is not None:   # distribution.partition(c), StructuredColumns_setup.cc:141
            return int(self.distribution[int(g)])
        if self.bs == 0:   # "row_bands": a whole row goes to the equal_bands part of its first point
            j = int(np.searchsorted(self.offsets, int(g), side="right")) - 1
            return bands_partition(int(self.offsets[j]), self.npts, self.nparts, 1)
        return bands_partition(int(g), self.npts, self.nparts, self.bs)

    def compute_j(self, j):   # :263-287 (not periodic in y)
        ny = self.ny
        if j < 0:
            j = -j if self.y[0] == 90.0 else -j - 1
        elif j >= ny:
            jlast = ny - 1
            j = jlast - 1 - (j - ny) if self.y[jlast] == -90.0 else jlast - (j - ny)
        if j < 0 or j >= ny:
            j = self.compute_j(j)
        return j

    def compute_i(self, i, j):
        nx = self.nx[j]
        while i >= nx:
            i -= nx
        while i < 0:
            i += nx
        return i

    def compute_x(self, i, j):   # :290-297
        jj = self.compute_j(j)
        ii = self.compute_i(i, jj)
        nx = self.nx[jj]
        a = math.trunc((ii - i) / nx) if False else int((ii - i) / nx)   # integer division truncating toward zero
        return self.gx(ii, jj) - a * self.gx(nx, jj)

    def compute_x_fast(self, i, jj, nx):   # :299-304
        ii = i
        while ii >= nx:
            ii -= nx
        while ii < 0:
            ii += nx
        a = int((ii - i) / nx)
        return self.gx(ii, jj) - a * (self.gx(nx, jj) - self.gx(0, jj))

    def compute_y(self, j):   # :312-324
        jj = self.compute_j(j)
        if j < 0:
            return 90.0 + (90.0 - self.y[jj])
        if j >= self.ny:
            return -90.0 + (-90.0 - self.y[jj])
        return self.y[jj]

    def compute_g(self, i, j):   # :331-355, 1-based global index
        jj = self.compute_j(j)
        ii = self.compute_i(i, jj)
        if jj != j:
            nx = self.nx[jj]
            if nx % 2 == 0:
                ii = ii + nx // 2 if ii < nx // 2 else ii - nx // 2
            else:
                if ii < nx // 2 + 1:
                    ii += nx // 2 + 1
                else:
                    ii -= nx // 2 + 1
        return int(self.offsets[jj]) + ii + 1

    def _setup(self):
        eps = 1e-12
        ny, halo = self.ny, self.halo
        BIG = 2 ** 31 - 1
        # ---- owned bounds (:125-226)
        self.i_begin = [BIG] * ny
        self.i_end = [-BIG - 1] * ny
        if self.nparts == 1:
            self.j_begin, self.j_end = 0, ny
            for j in range(ny):
                self.i_begin[j], self.i_end[j] = 0, self.nx[j]
            owned = self.npts
        else:
            self.j_begin, self.j_end, owned, c = BIG // 2, -(BIG // 2), 0, 0
            for j in range(ny):
                for i in range(self.nx[j]):
                    if self.partition(c) == self.part:
                        self.j_begin = min(self.j_begin, j)
                        self.j_end = max(self.j_end, j + 1)
                        self.i_begin[j] = min(self.i_begin[j], i)
                        self.i_end[j] = max(self.i_end[j], i + 1)
                        owned += 1
                    c += 1
            # the reference describes the owned region by one row range and one i-range per row (:125-226); anything
            # else makes it build halos around empty rows -- rejected here as in the product
            if owned == 0 or any(self.i_end[j] <= self.i_begin[j] for j in range(self.j_begin, self.j_end)) or \
                    sum(self.i_end[j] - self.i_begin[j] for j in range(self.j_begin, self.j_end)) != owned:
                raise ValueError("the points of a partition must form one row range with one i-range per row")
        self.size_owned = owned
        self.j_begin_halo, self.j_end_halo = self.j_begin - halo, self.j_end + halo
        ibh, ieh = {}, {}
        for j in range(self.j_begin - halo, self.j_end + halo):
            ibh[j], ieh[j] = BIG, -BIG
        # ---- halo bounds (:369-455)
        for j in range(self.j_begin, self.j_end):
            for i in (self.i_begin[j], self.i_end[j] - 1):
                if self.periodic_points and i == self.nx[j] - 1:
                    i += 1
                x, x_next, x_prev = self.gx(i, j), self.gx(i + 1, j), self.gx(i - 1, j)
                for jj in range(j - halo, j + halo + 1):
                    jjj = self.compute_j(jj)
                    nxj = self.nx[jjj]
                    last = nxj - 1
                    if i == self.nx[j]:
                        last += 1
                    dx = 360.0 / nxj
                    ii = int(math.floor((x + eps - 0.0) / dx))
                    while self.compute_x_fast(ii - 1, jjj, nxj) > x_prev + eps:
                        ii -= 1
                    i_minus = ii - halo
                    iii = ii
                    while self.compute_x_fast(iii + 1, jjj, nxj) < x_next - eps:
                        iii += 1
                    iii = min(iii, last)
                    i_plus = iii + halo
                    ibh[jj] = min(ibh[jj], i_minus)
                    ieh[jj] = max(ieh[jj], i_plus + 1)
        self.i_begin_halo, self.i_end_halo = ibh, ieh
        # ---- point ordering (:469-571)
        pts = []
        for j in range(self.j_begin, self.j_end):
            for i in range(self.i_begin[j], self.i_end[j]):
                pts.append((i, j))
        assert len(pts) == owned
        for j in range(self.j_begin_halo, self.j_begin):
            pts += [(i, j) for i in range(ibh[j], ieh[j])]
        for j in range(self.j_begin, self.j_end):
            pts += [(i, j) for i in range(ibh[j], self.i_begin[j])]
            pts += [(i, j) for i in range(self.i_end[j], ieh[j])]
        for j in range(self.j_end, self.j_end_halo):
            pts += [(i, j) for i in range(ibh[j], ieh[j])]
        self.size_halo = len(pts)
        n = self.size_halo
        self.index_i = np.array([p[0] for p in pts], dtype=np.int32)
        self.index_j = np.array([p[1] for p in pts], dtype=np.int32)
        self.ij2gp = {p: r for r, p in enumerate(pts)}
        # ---- fields (:583-662)
        self.xy = np.zeros((n, 2))
        self.partition_f = np.zeros(n, dtype=np.int32)
        self.glb_idx = np.zeros(n, dtype=np.int64)
        self.ghost = np.zeros(n, dtype=np.int32)
        self.ghost[owned:] = 1
        for r, (i, j) in enumerate(pts):
            if 0 <= j < ny:
                self.xy[r] = (self.gx(i, j), self.y[j])
            else:
                self.xy[r] = (self.compute_x(i, j), self.compute_y(j))
            if 0 <= j < ny and 0 <= i < self.nx[j]:
                k = int(self.offsets[j]) + i
                self.partition_f[r] = self.partition(k)
                self.glb_idx[r] = k + 1
            else:
                g = self.compute_g(i, j)
                self.glb_idx[r] = g
                self.partition_f[r] = self.partition(g - 1)
        # ---- remote index (create_remote_index.cc): owned -> own index; halo -> index of the global point in its
        #      owner's owned ordering.  Bands own contiguous global-index ranges, so that index is g - first(owner).
        first = {}
        for p in range(self.nparts):
            first[p] = next(g for g in range(self.npts) if self.partition(g) == p) if self.nparts > 1 else 0
        self.remote_idx = np.zeros(n, dtype=np.int32)
        if self.distribution is not None and self.nparts > 1:
            # general distribution: replay every owner's own numbering (owned points row by row over its
            # [i_begin, i_end) range, setup.cc:591-616) -- what the reference learns by asking the owner
            local = {}
            for p in range(self.nparts):
                cnt = 0
                for j in range(self.ny):
                    row = self.distribution[self.offsets[j]:self.offsets[j + 1]]
                    idx = np.nonzero(row == p)[0]
                    if idx.size:
                        for i in range(int(idx.min()), int(idx.max()) + 1):
                            local[(p, int(self.offsets[j]) + i)] = cnt
                            cnt += 1
            for r in range(n):
                self.remote_idx[r] = r if r < owned else local[(int(self.partition_f[r]), int(self.glb_idx[r]) - 1)]
            return
        for r in range(n):
            self.remote_idx[r] = r if r < owned else int(self.glb_idx[r]) - 1 - first[int(self.partition_f[r])]

    def index(self, i, j):
        return self.ij2gp[(i, j)]

    def pole_rows_nodes(self):
        """nodes of halo rows beyond the poles (j < 0 or j >= ny): FixupHaloForVectors, StructuredColumns.cc:745-760"""
        return np.array([r for r in range(self.size_halo) if self.index_j[r] < 0 or self.index_j[r] >= self.ny],
                        dtype=np.int32)
