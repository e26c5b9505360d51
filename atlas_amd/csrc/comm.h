// atlas_amd::parallel::Comm -- the inter-GPU transport of the library: grouped point-to-point exchanges of device
// buffers, asynchronous on a HIP stream.
//
// What it replaces in the reference: the eckit::mpi communicator used by parallel::HaloExchange (iReceive / iSend per
// peer, src/atlas/parallel/HaloExchange.h:191-219,333-369; allToAll / allToAllv in setup, HaloExchange.cc:118,156).
// TransLocal itself is single-process; the m -> latitude transposition of the distributed transform (dist_trans.h)
// uses the same primitive.
//
//   RcclComm  : one process per GPU; ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI.  The 128-byte
//               unique id is created on one rank (atlas_amd__Comm__get_unique_id) and handed to the others by the
//               caller's control plane (MPI_Bcast in Atlas, a torch.distributed / TCP store broadcast in the tests).
//   LocalComm : N ranks inside ONE process (one host thread per rank, all on the current device): the exchange is a
//               rendezvous + device copies ordered by events.  Used by the single-GPU tests of the distributed code
//               paths and by single-process multi-rank drivers; no RCCL involved.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace atlas_amd {
namespace parallel {

struct Msg {
    int peer;
    void* ptr;      // device pointer
    size_t bytes;
};

class Comm {
public:
    virtual ~Comm() {}
    virtual int size() const = 0;
    virtual int rank() const = 0;
    virtual const char* kind() const = 0;
    // One grouped exchange.  Between a pair of ranks the k-th send of one side is matched with the k-th receive of the
    // other and must have the same size.  Asynchronous: ordered after the work already in `stream`, complete for
    // later work in `stream`.  Send buffers may be reused by work submitted to `stream` afterwards.
    virtual void exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t stream) = 0;

    // blocking helpers on small host arrays (set-up phases), built on exchange()
    void all_to_all(const int* send, int* recv, int n_per_peer);
    void all_to_allv(const int* send, const int* sendcounts, int* recv, const int* recvcounts);
    void barrier();

protected:
    hipStream_t helper_stream();
    hipStream_t helper_stream_ = nullptr;
    void release_helper_stream();
};

// ---- RCCL ----------------------------------------------------------------------------------------------------------
constexpr int UNIQUE_ID_BYTES = 128;   // sizeof(ncclUniqueId)
void rccl_get_unique_id(void* out128);
std::unique_ptr<Comm> make_rccl_comm(const void* unique_id128, int nranks, int rank);

// ---- emulated ranks in one process ---------------------------------------------------------------------------------
class LocalHub {
public:
    explicit LocalHub(int nranks);
    ~LocalHub();
    int size() const { return n_; }

private:
    friend class LocalComm;
    void rendezvous();   // all ranks arrive (throws on every rank once one of them has failed)
    void fail(const std::string& what);
    bool failed_ = false;
    std::string failure_;
    int n_;
    std::mutex m_;
    std::condition_variable cv_;
    int arrived_   = 0;
    long long gen_ = 0;
    std::vector<const std::vector<Msg>*> sends_;   // published by the ranks for the current exchange
    std::vector<hipEvent_t> ready_, done_;
};
std::unique_ptr<Comm> make_local_comm(std::shared_ptr<LocalHub> hub, int rank);

}  // namespace parallel
}  // namespace atlas_amd
