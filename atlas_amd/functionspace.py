"""Host-side mirror of atlas::functionspace::StructuredColumns for global structured grids with band distributions
(reference: src/atlas/functionspace/StructuredColumns.h, detail/StructuredColumns.cc:811-911 haloExchange dispatch)."""
import ctypes as C

import numpy as np

from . import _lib
from .grid import StructuredGrid
from .parallel import HaloExchange, _describe

c_void_p, c_int = C.c_void_p, C.c_int
_sig = _lib._sig
SC_new = _sig("atlas_amd__StructuredColumns__new", c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int)
SC_new_distribution = _sig("atlas_amd__StructuredColumns__new_distribution", c_void_p, c_void_p, c_int, c_int, c_int,
                           c_int, c_void_p, C.c_longlong)
SC_delete = _sig("atlas_amd__StructuredColumns__delete", None, c_void_p)
SC_size_owned = _sig("atlas_amd__StructuredColumns__size_owned", c_int, c_void_p)
SC_size_halo = _sig("atlas_amd__StructuredColumns__size_halo", c_int, c_void_p)
SC_bounds = _sig("atlas_amd__StructuredColumns__bounds", c_int, c_void_p, c_void_p)
SC_row_bounds = _sig("atlas_amd__StructuredColumns__row_bounds", c_int, c_void_p, c_int, c_void_p)
SC_index = _sig("atlas_amd__StructuredColumns__index", c_int, c_void_p, c_int, c_int, c_void_p)
SC_get_int = _sig("atlas_amd__StructuredColumns__get_int", c_int, c_void_p, C.c_char_p, c_void_p)
SC_npole = _sig("atlas_amd__StructuredColumns__nb_pole_row_nodes", c_int, c_void_p)
SC_glb = _sig("atlas_amd__StructuredColumns__global_index", c_int, c_void_p, c_void_p)
SC_xy = _sig("atlas_amd__StructuredColumns__xy", c_int, c_void_p, c_void_p)
SC_setup_hx = _sig("atlas_amd__StructuredColumns__setup_halo_exchange", c_int, c_void_p, c_void_p, c_int, c_int)
SC_fixup = _sig("atlas_amd__StructuredColumns__fixup_halo_for_vectors", c_int, c_void_p, c_int, c_void_p, c_int,
                C.c_longlong, C.c_longlong, C.c_longlong, c_void_p)


class StructuredColumns:
    """functionspace::StructuredColumns(grid, distribution, halo=..., periodic_points=...)"""

    def __init__(self, grid, halo=0, periodic_points=False, nparts=1, part=0, distribution="equal_bands"):
        if isinstance(grid, str):
            grid = StructuredGrid(name=grid)
        self.grid = grid
        if isinstance(distribution, str) and distribution == "equal_regions":
            # Atlas's default partitioner for structured grids (EqualRegionsPartitioner.cc), mirrored in partitioner.py
            from .partitioner import EqualRegionsPartitioner
            distribution = EqualRegionsPartitioner(int(nparts)).partition(grid)
        if not isinstance(distribution, str):
            # explicit grid::Distribution: partition of every grid point in global order (equal_regions, checkerboard,
            # ... computed by the caller, as Atlas does before it constructs the function space)
            dist = np.ascontiguousarray(distribution, dtype=np.int32)
            self.nparts, self.part = int(nparts), int(part)
            self._h = _lib.check_ptr(SC_new_distribution(grid._h, int(halo), int(bool(periodic_points)), self.nparts,
                                                         self.part, dist.ctypes.data, dist.size))
            self._halo_exchange = None
            return
        if distribution == "equal_bands":
            bs = 1
        elif distribution == "regular_bands":
            if not grid.regular():
                raise ValueError("regular_bands needs a regular grid")   # RegularBandsPartitioner
            bs = grid.nxmax()
        elif distribution == "row_bands":
            # whole rows, each with the equal_bands part of its first point: Trans.bands(), the output decomposition of
            # the multi-GPU transform (for Atlas: a user-supplied grid::Distribution)
            bs = 0
        else:
            raise NotImplementedError(f"distribution '{distribution}' (supported: equal_regions, equal_bands, regular_bands, row_bands, or an explicit partition array)")
        self.nparts, self.part = int(nparts), int(part)
        self._h = _lib.check_ptr(SC_new(grid._h, int(halo), int(bool(periodic_points)), self.nparts, self.part, bs))
        self._halo_exchange = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and SC_delete is not None:
            SC_delete(h)
            self._h = None

    def sizeOwned(self):
        return SC_size_owned(self._h)

    def sizeHalo(self):
        return SC_size_halo(self._h)

    size = sizeHalo

    def _bounds(self):
        out = (c_int * 4)()
        SC_bounds(self._h, out)
        return list(out)

    def j_begin(self):
        return self._bounds()[0]

    def j_end(self):
        return self._bounds()[1]

    def j_begin_halo(self):
        return self._bounds()[2]

    def j_end_halo(self):
        return self._bounds()[3]

    def _row(self, j):
        out = (c_int * 4)()
        _lib.check(SC_row_bounds(self._h, int(j), out))
        return list(out)

    def i_begin(self, j):
        return self._row(j)[0]

    def i_end(self, j):
        return self._row(j)[1]

    def i_begin_halo(self, j):
        return self._row(j)[2]

    def i_end_halo(self, j):
        return self._row(j)[3]

    def index(self, i, j):
        out = c_int()
        _lib.check(SC_index(self._h, int(i), int(j), C.byref(out)))
        return out.value

    def _ints(self, what, n=None):
        out = np.zeros(max(self.sizeHalo() if n is None else n, 1), dtype=np.int32)
        _lib.check(SC_get_int(self._h, what.encode(), out.ctypes.data))
        return out[:self.sizeHalo() if n is None else n]

    def partition(self):
        return self._ints("partition")

    def ghost(self):
        return self._ints("ghost")

    def index_i(self):
        return self._ints("index_i")

    def index_j(self):
        return self._ints("index_j")

    def remote_index(self):
        return self._ints("remote_idx")

    def pole_row_nodes(self):
        return self._ints("pole_row_nodes", SC_npole(self._h))

    def global_index(self):
        out = np.zeros(self.sizeHalo(), dtype=np.int64)
        SC_glb(self._h, out.ctypes.data)
        return out

    def xy(self):
        out = np.zeros((self.sizeHalo(), 2))
        SC_xy(self._h, out.ctypes.data)
        return out

    # ---- halo exchange (StructuredColumns.cc:104-152, 811-911)
    def halo_exchange(self):
        if self._halo_exchange is None:
            if self.nparts != 1:
                raise RuntimeError("multi-partition function space: use begin_halo_exchange() on every partition, then "
                                   "HaloExchange.finish_emulated([...]) or a distributed setup")
            hx = HaloExchange()
            _lib.check(SC_setup_hx(self._h, hx._h, 1, 0))
            hx._is_setup = True
            self._halo_exchange = hx
        return self._halo_exchange

    def begin_halo_exchange(self):
        hx = HaloExchange()
        _lib.check(SC_setup_hx(self._h, hx._h, self.nparts, self.part))
        self._halo_exchange = hx
        return hx

    def fixup_halo_for_vectors(self, field, stream=None):
        """field(n, var) or field(n, lev, var) on the device; negates var 0,1 beyond the poles"""
        dt, ptr, rank, shape, strides, on_dev, _ = _describe(field)
        if not on_dev:
            raise TypeError("fixup_halo_for_vectors needs a device tensor")
        if rank == 2:
            lev, sn, sk, sv = 1, strides[0], 0, strides[1]
        elif rank == 3:
            lev, sn, sk, sv = shape[1], strides[0], strides[1], strides[2]
        else:
            raise NotImplementedError("vector fields must have rank 2 or 3")   # StructuredColumns.cc:735-741
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream   # stream-ordered with the tensor's producer
            _lib.check(SC_fixup(self._h, dt, ptr, lev, sn, sk, sv, stream))
        else:
            with _lib.torch_stream_order(stream):
                _lib.check(SC_fixup(self._h, dt, ptr, lev, sn, sk, sv, stream))

    def haloExchange(self, field, vector=False):
        """fs.haloExchange(field): exchange + pole fix-up for fields whose metadata type is 'vector'"""
        hx = self.halo_exchange()
        hx.execute(field)
        if vector:
            if hasattr(field, "is_cuda") and field.is_cuda:
                from .parallel import HX_stream
                self.fixup_halo_for_vectors(field, HX_stream(hx._h))
            else:
                raise TypeError("vector fix-up is implemented for device tensors")
        return field


def mirror_band_distribution(grid, nparts):
    """grid::Distribution of the mirror-band decomposition of the transform (Trans(shard="mirror")): part q owns the
    rows [b[q], b[q+1]) and their mirror images [ny-b[q+1], ny-b[q]).  Returns (partition of every grid point, b)."""
    b = np.zeros(int(nparts) + 1, dtype=np.int32)
    _lib.check(_lib.mirror_bands(grid._h, int(nparts), b.ctypes.data))
    nx = np.asarray(grid.nx())
    ny = len(nx)
    row_part = np.zeros(ny, dtype=np.int32)
    for q in range(int(nparts)):
        row_part[b[q]:b[q + 1]] = q
        row_part[ny - b[q + 1]:ny - b[q]] = q
    return np.repeat(row_part, nx).astype(np.int32), b


class MirrorBandColumns:
    """Function space of the mirror-band decomposition: every part owns TWO row ranges (a northern band and its mirror
    image).  StructuredColumns -- the reference's and ours -- describes a part by one row range, so this is a
    composition of two StructuredColumns blocks built on the 2*nparts single-range parts
        virtual part 2q = northern band of q,   2q+1 = its mirror image
    with the owners and remote indices mapped back to the real parts.  Local numbering:
        [owned north][owned south][halo of the north block][halo of the south block]
    i.e. the owned part is exactly the output order of Trans(shard="mirror").invtrans.  partition() / remote_index() are
    what HaloExchange::setup needs (StructuredColumns.cc:145-148); halo_begin = sizeOwned()."""

    def __init__(self, grid, halo=1, periodic_points=False, nparts=1, part=0):
        if isinstance(grid, str):
            grid = StructuredGrid(name=grid)
        self.grid, self.nparts, self.part = grid, int(nparts), int(part)
        dist, self.bands = mirror_band_distribution(grid, nparts)
        nx = np.asarray(grid.nx())
        ny = len(nx)
        off = np.concatenate([[0], np.cumsum(nx)])
        # the same distribution with the two ranges of a part told apart: virtual part = 2*part + (row in the south)
        south = np.repeat((np.arange(ny) >= ny // 2).astype(np.int32), nx)
        vdist = (2 * dist + south).astype(np.int32)
        self.blocks = [StructuredColumns(grid, halo=halo, periodic_points=periodic_points, nparts=2 * self.nparts,
                                         part=2 * self.part + s, distribution=vdist) for s in (0, 1)]
        # owned size of the northern block of every part: the offset of its southern block in the owner's numbering
        north_owned = np.array([int(off[self.bands[q + 1]] - off[self.bands[q]]) for q in range(self.nparts)])
        nA, nB = self.blocks[0].sizeOwned(), self.blocks[1].sizeOwned()
        hA = self.blocks[0].sizeHalo() - nA
        hB = self.blocks[1].sizeHalo() - nB
        self._owned, self._size = nA + nB, nA + nB + hA + hB
        # local index of block-local point i
        self._local = [np.concatenate([np.arange(nA), nA + nB + np.arange(hA)]),
                       np.concatenate([nA + np.arange(nB), nA + nB + hA + np.arange(hB)])]

        def gather(name, dtype):
            out = np.zeros(self._size, dtype=dtype)
            for blk, loc in zip(self.blocks, self._local):
                out[loc] = getattr(blk, name)()
            return out

        vpart = gather("partition", np.int32)
        vridx = gather("remote_index", np.int32)
        self._partition = (vpart // 2).astype(np.int32)
        self._remote = (vridx + np.where(vpart % 2 == 1, north_owned[vpart // 2], 0)).astype(np.int32)
        self._glb = gather("global_index", np.int64)
        self._index_i, self._index_j = gather("index_i", np.int32), gather("index_j", np.int32)
        self._ghost = gather("ghost", np.int32)
        self._halo_exchange = None

    def sizeOwned(self):
        return self._owned

    def sizeHalo(self):
        return self._size

    def partition(self):
        return self._partition

    def remote_index(self):
        return self._remote

    def global_index(self):
        return self._glb

    def index_i(self):
        return self._index_i

    def index_j(self):
        return self._index_j

    def ghost(self):
        return self._ghost

    def begin_halo_exchange(self, comm=None):
        """HaloExchange::setup(partition, remote_index, 0, sizeHalo, sizeOwned): comm = a torch.distributed group (or
        True) for one part per process; None (with nparts == 1) for a single part"""
        hx = HaloExchange()
        if comm is None and self.nparts != 1:
            hx.setup_emulated(self.nparts, self.part, self._partition, self._remote, 0, self._size, self._owned)
        else:
            hx.setup(self._partition, self._remote, 0, self._size, halo_begin=self._owned, comm=comm)
        self._halo_exchange = hx
        return hx


class NodeColumns:
    """Halo exchange of functionspace::NodeColumns (src/atlas/functionspace/NodeColumns.cc:101-113,357-459).

    Atlas builds it from a Mesh; here the three node arrays the reference hands to HaloExchange::setup are given
    directly: `partition` = mesh.nodes().partition(), `remote_index` = mesh.nodes().remote_index() (base
    REMOTE_IDX_BASE), and `nb_nodes` = metadata "nb_nodes_including_halo[halo]" (default: all nodes).  There is no
    halo_begin: every node is tested for ghost-ness (NodeColumns.cc:110-111)."""

    REMOTE_IDX_BASE = 0   # C++ value; the Fortran interface uses 1

    def __init__(self, partition, remote_index, nb_nodes=None, remote_idx_base=None, comm=None, emulate=None):
        from .parallel import HaloExchange
        self.partition = np.ascontiguousarray(partition, dtype=np.int32)
        self.remote_index = np.ascontiguousarray(remote_index, dtype=np.int32)
        self.nb_nodes = int(len(self.partition) if nb_nodes is None else nb_nodes)
        self.base = self.REMOTE_IDX_BASE if remote_idx_base is None else int(remote_idx_base)
        self._hx = HaloExchange()
        if emulate is not None:       # (nproc, myproc): single-process multi-rank setups (tests)
            self._hx.setup_emulated(emulate[0], emulate[1], self.partition, self.remote_index, self.base, self.nb_nodes)
        else:
            self._hx.setup(self.partition, self.remote_index, self.base, self.nb_nodes, comm=comm)

    def halo_exchange(self):
        return self._hx

    @staticmethod
    def _fieldset(fields):
        return list(fields) if isinstance(fields, (list, tuple)) else [fields]

    @staticmethod
    def _check(field):
        from .parallel import _describe
        try:
            _describe(field)
        except TypeError:
            raise TypeError("datatype not supported")            # NodeColumns.cc:374,394
        if not 1 <= len(field.shape) <= 4:
            raise ValueError("Rank not supported")               # NodeColumns.cc:417,439

    def haloExchange(self, fields, on_device=None):
        """haloExchange(Field | FieldSet): rank 1..4, int / long / float / double (NodeColumns.cc:357-421,446-450)"""
        for field in self._fieldset(fields):
            self._check(field)
            self._hx.execute(field)
        return fields

    def adjointHaloExchange(self, fields, on_device=None):
        for field in self._fieldset(fields):
            self._check(field)
            self._hx.execute_adjoint(field)
        return fields
