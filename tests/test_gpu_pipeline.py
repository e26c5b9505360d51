"""End-to-end on one device: multi-GPU inverse transform (latitude-band decomposition, one Trans object per emulated rank)
-> StructuredColumns field in the same decomposition (distribution="row_bands") -> halo exchange (pack / per-peer
segments / unpack, device copies standing in for RCCL send/recv).  Every owned and every halo value of every rank must
equal the single-device transform at that node's global index, bit for bit.  BASELINE config C3 (TL639 -> O640, 137
levels, 4 partitions) at full size; tolerance against the oracle (sampled rows): rel-RMS <= 1e-12."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import atlas_amd
from atlas_amd.functionspace import StructuredColumns
from atlas_amd.parallel import HaloExchange
import oracle
from helpers import red_spectra, compute_rms
from test_gpu_halo import exchange_emulated

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gridname,T,nf,nparts,halo", [("O32", 31, 5, 3, 2), ("O640", 639, 137, 4, 1)])
def test_transform_then_halo_exchange(gridname, T, nf, nparts, halo):
    g = atlas_amd.Grid(gridname)
    sp = red_spectra(T, nf, seed=21)
    sp_d = torch.from_numpy(sp).cuda()
    tr1 = atlas_amd.Trans(g, T)
    ref = torch.zeros(nf * g.size(), dtype=torch.float64, device="cuda")
    tr1.invtrans(nf, sp_d, ref)
    tr1.synchronize()
    ref = ref.cpu().numpy().reshape(nf, -1)
    off = np.concatenate([[0], np.cumsum(g.nx())])
    # absolute parity of the reference itself on a few rows
    rows = [0, g.ny() // 4, g.ny() // 2, g.ny() - 1]
    op = oracle.OraclePlan(T, g.nx(), g.y(), with_tables=False)
    for r, want in zip(rows, op.invtrans_rows(nf, sp, rows, use_fft=(T > 100))):
        assert compute_rms(ref[:, off[r]:off[r + 1]], want) < 1e-12, r
    del tr1

    fss, fields = [], []
    for p in range(nparts):
        tr = atlas_amd.Trans(g, T, nparts=nparts, part=p, shard="band")
        fs = StructuredColumns(g, halo=halo, periodic_points=True, nparts=nparts, part=p, distribution="row_bands")
        bands = tr.bands()
        assert fs.sizeOwned() == tr.nb_gridpoints() == off[bands[p + 1]] - off[bands[p]]
        gp = torch.zeros(nf * tr.nb_gridpoints(), dtype=torch.float64, device="cuda")
        tr.invtrans(nf, sp_d, gp)
        tr.synchronize()
        field = torch.full((fs.sizeHalo(), nf), float("nan"), dtype=torch.float64, device="cuda")
        field[:fs.sizeOwned()] = gp.reshape(nf, -1).T     # (nodes, levels), owned nodes in global order
        fss.append(fs)
        fields.append(field)
        del tr, gp
    hxs = [f.begin_halo_exchange() for f in fss]
    HaloExchange.finish_emulated(hxs)
    exchange_emulated(hxs, fields)
    for p, (fs, field) in enumerate(zip(fss, fields)):
        gi = fs.global_index() - 1
        got = field.cpu().numpy()
        assert np.array_equal(got, ref[:, gi].T), p
