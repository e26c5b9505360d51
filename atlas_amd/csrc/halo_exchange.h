// atlas_amd::parallel::HaloExchange -- MI355X counterpart of atlas::parallel::HaloExchange
// (reference: src/atlas/parallel/HaloExchange.h:40-148 interface, :151-290 execute / execute_adjoint).
// setup() = reference setup (host index logic, halo_setup.h); pack / unpack / adjoint / zero run as HIP kernels on
// the object's stream.  The peer-to-peer step between pack and unpack is the caller's (MPI in Atlas, RCCL
// send/recv through torch.distributed in atlas_amd/parallel.py); for one process the self-exchange is resolved on
// the device without any copy (the send buffer IS the receive buffer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "comm.h"
#include "halo_device.h"
#include "halo_setup.h"

namespace atlas_amd {
namespace parallel {

class HaloExchange {
public:
    HaloExchange();
    ~HaloExchange();
    HaloExchange(const HaloExchange&)            = delete;
    HaloExchange& operator=(const HaloExchange&) = delete;

    // HaloExchange::setup(part, remote_idx, base, parsize[, halo_begin])   HaloExchange.cc:66-76 (one process)
    void setup(const int part[], const int remote_idx[], int base, int parsize, int halo_begin = 0);
    // multi-process: local phase, then finish with the exchanged counts / requests
    void setup_begin(int nproc, int myproc, const int part[], const int remote_idx[], int base, int parsize,
                     int halo_begin = 0);
    // same, with part / remote_idx resident on the device: ghost list by wavefront-ballot compaction
    void setup_begin_device(int nproc, int myproc, const int* part_dev, const int* remote_idx_dev, int base,
                            int parsize, int halo_begin = 0);
    void setup_finish(const int sendcounts[], const int recv_requests[]);
    // the complete reference setup between the ranks of `comm` (allToAll of the receive counts, allToAllv of the
    // requested indices: HaloExchange.cc:118,156)
    void setup_comm(Comm& comm, const int part[], const int remote_idx[], int base, int parsize, int halo_begin = 0);
    bool is_setup() const { return plan_.finished; }
    const HaloPlan& plan() const { return plan_; }

    // field = (rank, shape[], strides[] in elements, parallel dimension); shape[parallel_dim] must equal parsize
    HaloFieldDesc describe(int rank, const int shape[], const long long strides[], int parallel_dim) const;

    // device-pointer stages (asynchronous on stream())
    void pack_device(int dtype, const void* field, const HaloFieldDesc& d, void* sendbuf);          // Packer::pack
    void unpack_device(int dtype, void* field, const HaloFieldDesc& d, const void* recvbuf);        // Packer::unpack
    void pack_adjoint_device(int dtype, const void* field, const HaloFieldDesc& d, void* buf);      // AdjointPacker::pack
    void unpack_adjoint_device(int dtype, void* field, const HaloFieldDesc& d, const void* buf);    // ::unpack (+=)
    void zero_halos_device(int dtype, void* field, const HaloFieldDesc& d);                         // Zeroer::zero
    // complete exchange for one process (periodic / pole duplicates are "ghosts" of the same rank)
    void execute_device(int dtype, void* field, const HaloFieldDesc& d);
    void execute_adjoint_device(int dtype, void* field, const HaloFieldDesc& d);
    // complete exchange between the ranks of `comm` (HaloExchange.h:191-219: irecv / pack / isend / wait / unpack;
    // adjoint :227-290): pack kernel -> one grouped send/recv per peer with counts and displacements scaled by var_size
    // (:318-331) -> unpack kernel, all asynchronous on stream().  A transform running on another stream overlaps it.
    void execute_comm(Comm& comm, int dtype, void* field, const HaloFieldDesc& d, bool adjoint);
    // the same on a stream of the caller's (a driver that orders the exchange itself, e.g. behind a transform's Fourier
    // stage); the object's scratch buffers are in use until that stream has passed the call
    void execute_comm_on(Comm& comm, int dtype, void* field, const HaloFieldDesc& d, bool adjoint, hipStream_t s);
    // host-pointer variants: stage the field through device memory (synchronous)
    void execute_host(int dtype, void* field, int rank, const int shape[], const long long strides[],
                      int parallel_dim, bool adjoint);

    hipStream_t stream() const { return stream_; }
    void set_stream(hipStream_t s);
    void synchronize() const;

private:
    void upload_maps();
    void* scratch(size_t bytes, int which);

    HaloPlan plan_;
    bool has_device_    = false;    // a HIP device was found when the object was made (stream 0 is a legal stream)
    hipStream_t stream_ = nullptr;  // nullptr == the default stream when has_device_
    bool own_stream_    = false;
    int* d_sendmap_     = nullptr;
    int* d_recvmap_     = nullptr;
    int* d_adj_nodes_   = nullptr;
    int* d_adj_start_   = nullptr;
    int* d_adj_items_   = nullptr;
    void* d_scratch_[3] = {nullptr, nullptr, nullptr};
    size_t scratch_cap_[3] = {0, 0, 0};
};

}  // namespace parallel
}  // namespace atlas_amd
