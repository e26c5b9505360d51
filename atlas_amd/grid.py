"""Global structured grids, mirroring the part of atlas::StructuredGrid that TransLocal reads
(reference: src/atlas/grid/detail/grid/Structured.h:300-330, Gaussian.cc:86-177)."""
import numpy as np

from . import _lib


class StructuredGrid:
    """atlas::StructuredGrid look-alike: ny(), nx(j), y(j), x(i,j), size(), nxmax()."""

    def __init__(self, name=None, nx=None, y=None):
        if name is not None:
            self._h = _lib.check_ptr(_lib.Grid_new_gaussian(name.encode()))
            self.name = name
        else:
            nx = np.ascontiguousarray(nx, dtype=np.int32)
            y = np.ascontiguousarray(y, dtype=np.float64)
            if nx.shape != y.shape or nx.ndim != 1:
                raise ValueError("nx and y must be 1-d arrays of equal length")
            self._h = _lib.check_ptr(_lib.Grid_new_structured(len(nx), nx.ctypes.data, y.ctypes.data))
            self.name = "structured"
        n = _lib.Grid_ny(self._h)
        self._nx = np.zeros(n, dtype=np.int32)
        self._y = np.zeros(n, dtype=np.float64)
        _lib.Grid_nx(self._h, self._nx.ctypes.data)
        _lib.Grid_y(self._h, self._y.ctypes.data)

    def __del__(self):
        h = getattr(self, "_h", None)
        delete = getattr(_lib, "Grid_delete", None) if _lib is not None else None   # module globals go first at interpreter exit
        if h and delete is not None:
            delete(h)
            self._h = None

    def crop_to_domain(self, west, east, south, north):
        """atlas::Grid(grid, RectangularDomain({west, east}, {south, north})): ((row_begin, row_end), first global index
        of every kept row, number of points of every kept row (taken with wrap-around))"""
        import ctypes as C
        j0, j1 = C.c_int(0), C.c_int(0)
        i0 = np.zeros(self.ny(), dtype=np.int32)
        n = np.zeros(self.ny(), dtype=np.int32)
        _lib.check(_lib.Grid_crop_to_domain(self._h, float(west), float(east), float(south), float(north), C.byref(j0),
                                            C.byref(j1), i0.ctypes.data, n.ctypes.data, int(self.ny())))
        k = j1.value - j0.value
        return (j0.value, j1.value), i0[:k].copy(), n[:k].copy()

    def ny(self):
        return len(self._nx)

    def nx(self, j=None):
        return self._nx.copy() if j is None else int(self._nx[j])

    def y(self, j=None):
        return self._y.copy() if j is None else float(self._y[j])

    def x(self, i, j):
        return 0.0 + float(i) * (360.0 / float(self._nx[j]))  # Structured.h:308

    def nxmax(self):
        return int(self._nx.max())

    def size(self):
        return int(self._nx.sum())

    def regular(self):
        return bool(_lib.Grid_regular(self._h))


def Grid(name):
    """atlas::Grid("F64") / Grid("O1280")"""
    return StructuredGrid(name=name)


def gaussian_latitudes(N):
    """2N Gaussian latitudes in degrees, north pole to south pole (Latitudes.cc:59-67)."""
    out = np.zeros(2 * N)
    _lib.check(_lib.gaussian_latitudes_npole_spole(N, out.ctypes.data))
    return out
