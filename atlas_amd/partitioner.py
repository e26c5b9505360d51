"""Atlas's default partitioner for structured grids, `equal_regions`
(src/atlas/grid/detail/partitioner/EqualRegionsPartitioner.cc:70-347,443-605), and the bands rule
(BandsDistribution.h:32-34), through the library (csrc/equal_regions.cpp).  The output -- one partition number per grid
point in global order -- is the explicit grid::Distribution that `functionspace.StructuredColumns(grid, distribution=...)`
accepts."""
import ctypes as C

import numpy as np

from . import _lib

_eq_caps = _lib._sig("atlas_amd__eq_caps", C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_equal_regions_partition = _lib._sig("atlas_amd__equal_regions_partition", C.c_int, C.c_void_p, C.c_int, C.c_void_p)


def eq_caps(N):
    """eq_caps (EqualRegionsPartitioner.cc:276-343): (regions per zone north->south, cap colatitudes)"""
    n = C.c_int(0)
    _lib.check(_eq_caps(int(N), 0, None, None, C.byref(n)))
    r = np.zeros(n.value, dtype=np.int32)
    c = np.zeros(n.value, dtype=np.float64)
    _lib.check(_eq_caps(int(N), n.value, r.ctypes.data, c.ctypes.data, C.byref(n)))
    return [int(v) for v in r], [float(v) for v in c]


class EqualRegionsPartitioner:
    """grid::Partitioner("equal_regions", N)"""

    def __init__(self, N):
        self.N = int(N)
        self.sectors, s_cap = eq_caps(self.N)
        self.bands = [0.5 * np.pi - s for s in s_cap]

    def nb_bands(self):
        return len(self.bands)

    def nb_regions(self, band):
        return self.sectors[band]

    def partition(self, grid):
        """partition(const Grid&, int part[]): part[npts] in the grid's global point order"""
        out = np.zeros(int(grid.size()), dtype=np.int32)
        _lib.check(_equal_regions_partition(grid._h, self.N, out.ctypes.data))
        return out


def bands_partition(grid, nparts, blocksize=1):
    """BandsDistribution (grid/detail/distribution/BandsDistribution.h:32-34): equal_bands (blocksize 1) ..."""
    npts = int(grid.nx().sum())
    nb_blocks = (npts + blocksize - 1) // blocksize
    g = np.arange(npts, dtype=np.int64)
    return (((g // blocksize) * nparts) // nb_blocks).astype(np.int32)
