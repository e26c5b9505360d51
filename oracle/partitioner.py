"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of Atlas's default partitioner for structured grids, `equal_regions`
(src/atlas/grid/detail/partitioner/EqualRegionsPartitioner.cc:70-347,443-605), and of the bands rule
(BandsDistribution.h:32-34).  The output -- one partition number per grid point in global order -- is the explicit
grid::Distribution that `functionspace.StructuredColumns(grid, distribution=...)` accepts.

Leopardi's zonal equal-area partition of the sphere: N regions in polar caps plus collars (`eq_caps`); the grid points, in
their north->south / west->east order, are then cut into bands holding `regions(band)` chunks of npts/N (+1 for the first
npts % N chunks) points, and every band is cut into sectors after sorting its points west->east / north->south."""
import math

import numpy as np


def _gamma(x):
    """EqualRegionsPartitioner.cc:70-115 (polynomial approximation used for the area of the sphere)"""
    p = [0.999999999999999990e+00, -0.422784335098466784e+00, -0.233093736421782878e+00, 0.191091101387638410e+00,
         -0.024552490005641278e+00, -0.017645244547851414e+00, 0.008023273027855346e+00, -0.000804329819255744e+00,
         -0.000360837876648255e+00, 0.000145596568617526e+00, -0.000017545539395205e+00, -0.000002591225267689e+00,
         0.000001337767384067e+00, -0.000000199542863674e+00]
    n = int(math.floor(x - 2 + 0.5)) if x - 2 >= 0 else -int(math.floor(-(x - 2) + 0.5))   # C round(): half away from 0
    w = x - (n + 2)
    y = p[13]
    for c in p[12::-1]:
        y = y * w + c
    if n > 0:
        w = x - 1
        for k in range(2, n + 1):
            w = w * (x - k)
    else:
        w = 1.0
        for k in range(0, -n):
            y = y * (x + k)
    return w / y


def _c_round(v):
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def _area_of_cap(s_cap):
    # EqualRegionsPartitioner.cc:125: 4.0 * M_PI * std::pow(std::sin(0.5 * s_cap), 2) -- the square first (g++ compiles the pow to a
    # multiplication), then times 4 pi; the order matters: zone shares of x.5 are rounded from differences of these areas
    s = math.sin(0.5 * s_cap)
    return 4.0 * math.pi * (s * s)


def _sradius_of_cap(area):
    return 2.0 * math.asin(0.5 * math.sqrt(area / math.pi))


def _area_of_ideal_region(N):
    return 2.0 * math.pi ** 1.5 / _gamma(1.5) / float(N)


def eq_caps(N):
    """eq_caps (EqualRegionsPartitioner.cc:276-343): (n_regions per zone north->south, cap colatitudes)"""
    if N == 1:
        return [1], [math.pi]
    c_polar = 0.5 * math.pi if N == 2 else _sradius_of_cap(_area_of_ideal_region(N))   # polar_colat, :154-169
    a_ideal = math.sqrt(_area_of_ideal_region(N))
    n_collars = max(1, int(_c_round((math.pi - 2.0 * c_polar) / a_ideal))) if (N > 2 and a_ideal > 0) else 0
    r_regions = [0.0] * (n_collars + 2)
    r_regions[0] = 1.0
    if n_collars > 0:
        a_fitting = (math.pi - 2.0 * c_polar) / float(n_collars)
        ideal = _area_of_ideal_region(N)
        for c in range(n_collars):
            collar = _area_of_cap(c_polar + (c + 1) * a_fitting) - _area_of_cap(c_polar + c * a_fitting)
            r_regions[1 + c] = collar / ideal
    r_regions[n_collars + 1] = 1.0
    n_regions, discrepancy = [], 0.0
    for r in r_regions:                      # round_to_naturals
        n = int(_c_round(r + discrepancy))
        n_regions.append(n)
        discrepancy += r - n
    s_cap = [c_polar]
    ideal, subtotal = _area_of_ideal_region(N), 1
    for c in range(n_collars):               # cap_colats
        subtotal += n_regions[1 + c]
        s_cap.append(_sradius_of_cap(subtotal * ideal))
    s_cap.append(math.pi)
    return n_regions, s_cap


def microdeg(deg):
    """util/MicroDeg.h:18-22"""
    return int(deg * 1.e6 - 0.5) if deg < 0 else int(deg * 1.e6 + 0.5)


class EqualRegionsPartitioner:
    """grid::Partitioner("equal_regions", N)"""

    def __init__(self, N):
        self.N = int(N)
        self.sectors, s_cap = eq_caps(self.N)
        self.bands = [0.5 * math.pi - s for s in s_cap]

    def nb_bands(self):
        return len(self.bands)

    def nb_regions(self, band):
        return self.sectors[band]

    def partition(self, grid):
        """partition(const Grid&, int part[]) for a structured grid (EqualRegionsPartitioner.cc:544-605,614-700):
        returns part[npts] in the grid's global point order"""
        nx, y = grid.nx(), grid.y()
        npts = int(nx.sum())
        if self.N == 1:
            return np.zeros(npts, dtype=np.int32)
        # integer micro-degree coordinates of every point, grid order = north->south, west->east
        xs = np.concatenate([np.array([microdeg(i * (360.0 / int(n))) for i in range(int(n))], dtype=np.int64)
                             for n in nx])
        ys = np.concatenate([np.full(int(n), microdeg(float(v)), dtype=np.int64) for n, v in zip(nx, y)])
        order = np.arange(npts)
        chunk, rem = divmod(npts, self.N)
        counts, end = [], 0
        for band in range(self.nb_bands()):
            begin = end
            for _ in range(self.nb_regions(band)):
                counts.append(chunk + (1 if rem > 0 else 0))
                rem -= 1
                end += counts[-1]
            seg = order[begin:end]
            # compare_WE_NS: x ascending, then y descending
            order[begin:end] = seg[np.lexsort((-ys[seg], xs[seg]))]
        part = np.zeros(npts, dtype=np.int32)
        end = 0
        for p, c in enumerate(counts):
            part[order[end:end + c]] = p
            end += c
        return part


def bands_partition(grid, nparts, blocksize=1):
    """BandsDistribution (grid/detail/distribution/BandsDistribution.h:32-34): equal_bands (blocksize 1) ..."""
    npts = int(grid.nx().sum())
    nb_blocks = (npts + blocksize - 1) // blocksize
    g = np.arange(npts, dtype=np.int64)
    return (((g // blocksize) * nparts) // nb_blocks).astype(np.int32)
