// See halo_setup.h.  Host-only.
#include "halo_setup.h"

#include <algorithm>
#include <map>
#include <numeric>
#include <stdexcept>

namespace atlas_amd {
namespace parallel {

void halo_setup_local(HaloPlan& plan, int nproc, int myproc, const int part[], const int remote_idx[], int base,
                      int parsize, int halo_begin) {
    if (nproc < 1 || myproc < 0 || myproc >= nproc || parsize < 0 || halo_begin < 0) {
        throw std::invalid_argument("HaloExchange::setup: bad arguments");
    }
    plan          = HaloPlan();
    plan.nproc    = nproc;
    plan.myproc   = myproc;
    plan.parsize  = parsize;
    plan.recvcounts.assign(nproc, 0);
    plan.sendcounts.assign(nproc, 0);
    plan.recvdispls.assign(nproc, 0);
    plan.senddispls.assign(nproc, 0);
    // ghost <=> owned by another part, or a local duplicate of another local node (periodic / pole points):
    // part != me || remote_idx != base + idx   (IsGhostPoint, HaloExchange.cc:31-52)
    std::vector<int> ghosts;
    for (int jj = halo_begin; jj < parsize; ++jj) {
        if (part[jj] != myproc || remote_idx[jj] != base + jj) {
            const int p = part[jj];
            if (p < 0 || p >= nproc) {
                throw std::invalid_argument("HaloExchange::setup: partition index out of range");
            }
            ++plan.recvcounts[p];
            ghosts.push_back(jj);  // ascending index (the reference's order with one OpenMP thread)
        }
    }
    plan.recvcnt = (int)ghosts.size();
    for (int p = 1; p < nproc; ++p) {
        plan.recvdispls[p] = plan.recvcounts[p - 1] + plan.recvdispls[p - 1];
    }
    plan.send_requests.assign(plan.recvcnt, 0);
    plan.recvmap.assign(plan.recvcnt, 0);
    std::vector<int> cnt(nproc, 0);
    for (int jj : ghosts) {
        const int p                 = part[jj];
        const int req               = plan.recvdispls[p] + cnt[p]++;
        plan.send_requests[req]     = remote_idx[jj] - base;
        plan.recvmap[req]           = jj;
    }
}

void halo_setup_finish(HaloPlan& plan, const int sendcounts[], const int recv_requests[]) {
    if (plan.nproc < 1 || (int)plan.senddispls.size() != plan.nproc || (int)plan.recvcounts.size() != plan.nproc) {
        throw std::logic_error("HaloExchange::setup_finish called before setup_begin");
    }
    if (!sendcounts || (!recv_requests && std::accumulate(sendcounts, sendcounts + plan.nproc, 0) > 0)) {
        throw std::invalid_argument("HaloExchange::setup_finish: null argument");
    }
    for (int p = 0; p < plan.nproc; ++p) {
        if (sendcounts[p] < 0) {
            throw std::invalid_argument("HaloExchange::setup_finish: negative send count");
        }
    }
    plan.sendcounts.assign(sendcounts, sendcounts + plan.nproc);
    plan.sendcnt = std::accumulate(plan.sendcounts.begin(), plan.sendcounts.end(), 0);
    plan.senddispls[0] = 0;
    for (int p = 1; p < plan.nproc; ++p) {
        plan.senddispls[p] = plan.sendcounts[p - 1] + plan.senddispls[p - 1];
    }
    plan.sendmap.assign(recv_requests, recv_requests + plan.sendcnt);
    for (int v : plan.sendmap) {
        if (v < 0 || v >= plan.parsize) {
            throw std::invalid_argument("HaloExchange::setup: requested remote index out of range");
        }
    }
    // adjoint CSR: destination node -> buffer positions, ascending
    std::map<int, std::vector<int>> by_node;
    for (int i = 0; i < plan.sendcnt; ++i) {
        by_node[plan.sendmap[i]].push_back(i);
    }
    plan.adj_nodes.clear();
    plan.adj_start.assign(1, 0);
    plan.adj_items.clear();
    for (auto& kv : by_node) {
        plan.adj_nodes.push_back(kv.first);
        plan.adj_items.insert(plan.adj_items.end(), kv.second.begin(), kv.second.end());
        plan.adj_start.push_back((int)plan.adj_items.size());
    }
    plan.finished = true;
}

void halo_setup_serial(HaloPlan& plan, const int part[], const int remote_idx[], int base, int parsize,
                       int halo_begin) {
    halo_setup_local(plan, 1, 0, part, remote_idx, base, parsize, halo_begin);
    std::vector<int> counts = plan.recvcounts;
    std::vector<int> req    = plan.send_requests;
    halo_setup_finish(plan, counts.data(), req.data());
}

}  // namespace parallel
}  // namespace atlas_amd
