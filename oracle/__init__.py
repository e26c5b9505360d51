"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the TransLocal inverse transform / HaloExchange hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing under
atlas_amd/ does.  See oracle/README.md for how the oracle itself is pinned (parity status: PINNED against the
reference's own analytic known-answer tests and halo-exchange fixtures; the reference binary is unbuildable here)."""
from .translocal import (OraclePlan, c2r_direct, c2r_fft, fourier_truncation, gemm, invtrans_regional,  # noqa: F401
                        invtrans_regional_vordiv, invtrans_unstructured,
                        legendre_lat, vd2uv)
