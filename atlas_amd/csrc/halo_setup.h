// Host logic of atlas::parallel::HaloExchange::setup, split into a local phase and a finish phase so that the
// two collective steps (allToAll of counts, allToAllv of requested remote indices; reference:
// src/atlas/parallel/HaloExchange.cc:118,156-159) can be carried by whatever transport the caller has
// (MPI in an Atlas build, torch.distributed / RCCL here, plain memcpy for one process).
#pragma once
#include <vector>

namespace atlas_amd {
namespace parallel {

struct HaloPlan {
    int nproc = 1, myproc = 0, parsize = 0;
    std::vector<int> sendcounts, recvcounts, senddispls, recvdispls;  // per peer, in nodes
    std::vector<int> sendmap, recvmap;                                // local node indices
    std::vector<int> send_requests;  // remote indices (0-based) requested from each owner, grouped by owner
    int sendcnt = 0, recvcnt = 0;
    bool finished = false;
    // adjoint unpack ("field[sendmap[i]] += buf[i]", detail/adjoint_unpack_index.h): a node may occur several times in
    // sendmap; contributions are grouped per destination node, in ascending buffer order, so that the sum is
    // evaluated in the reference's (sequential) order without atomics
    std::vector<int> adj_nodes;  // unique destination nodes
    std::vector<int> adj_start;  // CSR offsets into adj_items [adj_nodes.size()+1]
    std::vector<int> adj_items;  // buffer positions
};

// HaloExchange.cc:78-116 + 133-150: ghost detection, recvcounts, recvmap, requests
void halo_setup_local(HaloPlan& plan, int nproc, int myproc, const int part[], const int remote_idx[], int base,
                      int parsize, int halo_begin);
// HaloExchange.cc:118-131 + 152-172: after the exchange of counts and requests
void halo_setup_finish(HaloPlan& plan, const int sendcounts[], const int recv_requests[]);
// single process: both phases (requests to self)
void halo_setup_serial(HaloPlan& plan, const int part[], const int remote_idx[], int base, int parsize,
                       int halo_begin);

}  // namespace parallel
}  // namespace atlas_amd
