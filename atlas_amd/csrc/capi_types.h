// Opaque handle types of include/atlas_amd.h (internal definitions shared by the capi_*.hip translation units).
#pragma once
#include <memory>

#include "gaussian.h"
#include "halo_exchange.h"

namespace atlas_amd {
namespace trans {
class Trans;
class DistributedTrans;
}
}  // namespace atlas_amd

struct atlas_amd_Grid {
    atlas_amd::grid::StructuredGrid g;
};
struct atlas_amd_Spectral {
    int truncation = -1;
};
struct atlas_amd_Trans {
    atlas_amd::trans::Trans* impl;
    const atlas_amd_Grid* grid = nullptr;  // borrowed: what atlas__Trans__grid returns
    int mirror_b0 = -1, mirror_b1 = -1;    // shard=mirror: the Legendre rows [b0, b1) this object transforms in both hemispheres
    std::shared_ptr<atlas_amd::trans::DistributedTrans> dist;   // buffers / streams of invtrans_distributed (made on first use)
    struct atlas_amd_Comm* dist_comm = nullptr;                // the communicator `dist` was made for (borrowed)
    unsigned long long dist_comm_id = 0;                       // ... and its serial number (a new communicator at the same address is a different one)
    atlas_amd_Spectral spectral;           // functionspace::Spectral(truncation), what atlas__Trans__spectral returns
};
struct atlas_amd_HaloExchange {
    atlas_amd::parallel::HaloExchange impl;
};
