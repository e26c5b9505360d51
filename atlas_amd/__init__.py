"""atlas_amd: MI355X-native TransLocal inverse spherical-harmonics transform + HaloExchange (ecmwf/atlas 0.44.1
hot path), behind the C ABI declared in include/atlas_amd.h.  This package is the thin host-side mirror of the
reference's operator interface (atlas::trans::Trans, atlas::parallel::HaloExchange) used by tests and bench."""
from . import _lib
from .grid import Grid, StructuredGrid, gaussian_latitudes
from .trans import LegendreCacheCreator, RegionalTrans, Trans, VorDivToUV

__all__ = ["Grid", "StructuredGrid", "gaussian_latitudes", "Trans", "VorDivToUV", "LegendreCacheCreator"]
