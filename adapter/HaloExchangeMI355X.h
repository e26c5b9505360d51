// Atlas-side class with the public surface of atlas::parallel::HaloExchange (src/atlas/parallel/HaloExchange.h:37-58: name,
// the four setup overloads, execute<DATA_TYPE, RANK, ParallelDim>, execute_adjoint<...>) over libatlas_amd.so.
//
// parallel::HaloExchange is a concrete class without a factory, so a backend cannot be registered for it; this class is
// used where Atlas constructs one -- functionspace::StructuredColumns::setup (detail/StructuredColumns_setup.cc:652-661,
// halo_exchange_ = new parallel::HaloExchange()) and NodeColumns (HaloExchangeCache) -- by the maintainer who builds the
// plugin: `using HaloExchange = parallel::HaloExchangeMI355X;` behind ATLAS_HAVE_MI355X.  Compiled on the Atlas side (header
// only; needs eckit::mpi and Atlas's array views); checked at source level by tests/test_adapter_source.py.
//
// Division of labour (reference: HaloExchange.cc:78-172 setup, HaloExchange.h:151-290 execute / execute_adjoint):
//   * index work of setup (ghost list, recvmap, request lists, sendmap) ............ library, atlas_amd__HaloExchange__setup_begin /
//     __setup_finish; the two collectives in between stay Atlas's: comm().allToAll (:118), comm().allToAllv (:156-159)
//   * pack / unpack / adjoint accumulate / zero_halos (DevicePacker.hic:51-218) ..... HIP kernels, atlas_amd__HaloExchange__field_op
//   * point-to-point exchange of the packed buffers (HaloExchange.h:333-369) ........ Atlas's eckit::mpi, on device buffers when
//     MPI is GPU-aware (ATLAS_HAVE_GPU_AWARE_MPI), else on host copies -- exactly the reference's rule (:160-172)
// (The library can also run the whole exchange itself over RCCL: atlas_amd__HaloExchange__setup_comm / __execute_comm,
// INTEGRATION.md section 4; that path needs no MPI at all and is what the distributed transform uses.)
#pragma once
#include <numeric>
#include <string>
#include <vector>

#include "atlas/array.h"
#include "atlas/array/ArrayView.h"
#include "atlas/array/ArrayViewDefs.h"
#include "atlas/array_fwd.h"
#include "atlas/library/config.h"
#include "atlas/parallel/mpi/mpi.h"
#include "atlas/runtime/Exception.h"
#include "atlas/util/Allocate.h"

extern "C" {
#include "atlas_amd.h"
}

namespace atlas {
namespace parallel {

class HaloExchangeMI355X {
public:
    HaloExchangeMI355X() : HaloExchangeMI355X(std::string()) {}
    HaloExchangeMI355X(const std::string& name) : name_(name), h_(atlas_amd__HaloExchange__new()) {
        if (!h_) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
    }
    virtual ~HaloExchangeMI355X() { atlas_amd__HaloExchange__delete(h_); }
    HaloExchangeMI355X(const HaloExchangeMI355X&)            = delete;
    HaloExchangeMI355X& operator=(const HaloExchangeMI355X&) = delete;

public:  // methods (HaloExchange.h:47-58)
    const std::string& name() const { return name_; }

    void setup(const int part[], const idx_t remote_idx[], const int base, idx_t size) {
        setup(mpi::comm().name(), part, remote_idx, base, size, 0);
    }
    void setup(const std::string& mpi_comm, const int part[], const idx_t remote_idx[], const int base, idx_t size) {
        setup(mpi_comm, part, remote_idx, base, size, 0);
    }
    void setup(const int part[], const idx_t remote_idx[], const int base, idx_t size, idx_t halo_begin) {
        setup(mpi::comm().name(), part, remote_idx, base, size, halo_begin);
    }
    // HaloExchange.cc:78-172
    void setup(const std::string& mpi_comm, const int part[], const idx_t remote_idx[], const int base, idx_t size,
               idx_t halo_begin) {
        comm_  = &mpi::comm(mpi_comm);
        nproc  = int(comm().size());
        myproc = int(comm().rank());
        std::vector<int> ridx(remote_idx, remote_idx + size);   // idx_t may be 64 bit; the C ABI takes int
        check(atlas_amd__HaloExchange__setup_begin(h_, nproc, myproc, part, ridx.data(), base, int(size), int(halo_begin)));
        std::vector<int> recvcounts(nproc), sendcounts(nproc), recvdispls(nproc), senddispls(nproc);
        check(atlas_amd__HaloExchange__get(h_, "recvcounts", recvcounts.data()));
        comm().allToAll(recvcounts, sendcounts);                                            // HaloExchange.cc:118
        std::exclusive_scan(recvcounts.begin(), recvcounts.end(), recvdispls.begin(), 0);
        std::exclusive_scan(sendcounts.begin(), sendcounts.end(), senddispls.begin(), 0);
        const int recvcnt = std::accumulate(recvcounts.begin(), recvcounts.end(), 0);
        const int sendcnt = std::accumulate(sendcounts.begin(), sendcounts.end(), 0);
        std::vector<int> send_requests(recvcnt > 0 ? recvcnt : 1), recv_requests(sendcnt > 0 ? sendcnt : 1);
        check(atlas_amd__HaloExchange__get(h_, "send_requests", send_requests.data()));
        comm().allToAllv(send_requests.data(), recvcounts.data(), recvdispls.data(), recv_requests.data(), sendcounts.data(),
                         senddispls.data());                                               // HaloExchange.cc:156-159
        check(atlas_amd__HaloExchange__setup_finish(h_, sendcounts.data(), recv_requests.data()));
        sendcounts_ = sendcounts, recvcounts_ = recvcounts, senddispls_ = senddispls, recvdispls_ = recvdispls;
        sendcnt_ = sendcnt, recvcnt_ = recvcnt;
        is_setup_ = true;
    }

    template <typename DATA_TYPE, int RANK, typename ParallelDim = array::FirstDim>
    void execute(array::Array& field, bool on_device = false) const {
        exchange<DATA_TYPE, RANK, ParallelDim>(field, on_device, /*adjoint*/ false);
    }

    template <typename DATA_TYPE, int RANK, typename ParallelDim = array::FirstDim>
    void execute_adjoint(array::Array& field, bool on_device = false) const {
        exchange<DATA_TYPE, RANK, ParallelDim>(field, on_device, /*adjoint*/ true);
    }

private:
    static void check(int rc) {
        if (rc != 0) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
    }
    const mpi::Comm& comm() const { return *comm_; }
    template <typename T>
    static constexpr int dtype() {   // include/atlas_amd.h: 0 int, 1 long, 2 float, 3 double
        return std::is_same<T, int>::value ? 0 : std::is_same<T, long>::value ? 1 : std::is_same<T, float>::value ? 2 : 3;
    }

    // forward: pack(sendmap) -> send | recv -> unpack(recvmap)                  (HaloExchange.h:151-225)
    // adjoint: pack_adjoint(recvmap) -> send | recv -> unpack_adjoint(+= sendmap), zero_halos   (HaloExchange.h:227-290)
    template <typename DATA_TYPE, int RANK, typename ParallelDim>
    void exchange(array::Array& field, bool on_device, bool adjoint) const {
        if (!is_setup_) {
            throw_Exception("HaloExchange was not setup", Here());
        }
        // the packing kernels always run on the device; what the flag decides -- as in the reference -- is where the field
        // and the message buffers live
        const bool device_msgs = on_device && ATLAS_HAVE_GPU_AWARE_MPI;
        if (on_device) {
            ATLAS_ASSERT(field.deviceNeedsUpdate() == false);
        }
        else {
            field.updateDevice();
        }
        auto view = array::make_device_view<DATA_TYPE, RANK>(field);
        constexpr int parallelDim = array::get_parallel_dim<ParallelDim>(view);
        int shape[RANK];
        long long strides[RANK];
        idx_t var_size = 1;
        for (int d = 0; d < RANK; ++d) {
            shape[d]   = int(view.shape(d));
            strides[d] = (long long)view.stride(d);
            if (d != parallelDim) {
                var_size *= view.shape(d);
            }
        }
        // sizes of the buffer that leaves / arrives: forward sends at sendmap, adjoint sends what sits at recvmap
        const int out_size = (adjoint ? recvcnt_ : sendcnt_) * int(var_size);
        const int in_size  = (adjoint ? sendcnt_ : recvcnt_) * int(var_size);
        const std::vector<int>& out_counts = adjoint ? recvcounts_ : sendcounts_;
        const std::vector<int>& out_displs = adjoint ? recvdispls_ : senddispls_;
        const std::vector<int>& in_counts  = adjoint ? sendcounts_ : recvcounts_;
        const std::vector<int>& in_displs  = adjoint ? senddispls_ : recvdispls_;
        DATA_TYPE *out_dev = nullptr, *in_dev = nullptr;
        util::allocate_devicemem(out_dev, out_size);
        util::allocate_devicemem(in_dev, in_size);
        std::vector<DATA_TYPE> out_host, in_host;
        check(atlas_amd__HaloExchange__field_op(h_, adjoint ? 4 : 2, dtype<DATA_TYPE>(), view.data(), RANK, shape, strides,
                                                parallelDim, out_dev, 1));
        check(atlas_amd__HaloExchange__synchronize(h_));
        DATA_TYPE* out_msg = out_dev;
        DATA_TYPE* in_msg  = in_dev;
        if (!device_msgs) {   // MPI on host copies of the packed buffers (ATLAS_HAVE_GPU_AWARE_MPI == 0)
            out_host.resize(out_size);
            in_host.resize(in_size);
            check(atlas_amd__device_memcpy_d2h(out_host.data(), out_dev, size_t(out_size) * sizeof(DATA_TYPE)));
            out_msg = out_host.data();
            in_msg  = in_host.data();
        }
        const int tag = 1;
        std::vector<eckit::mpi::Request> rreq(nproc), sreq(nproc);
        for (int p = 0; p < nproc; ++p) {                                                  // HaloExchange.h:333-345
            if (in_counts[p] > 0) {
                rreq[p] = comm().iReceive(in_msg + in_displs[p] * var_size, size_t(in_counts[p] * var_size), p, tag);
            }
        }
        for (int p = 0; p < nproc; ++p) {                                                  // :347-369
            if (out_counts[p] > 0) {
                sreq[p] = comm().iSend(out_msg + out_displs[p] * var_size, size_t(out_counts[p] * var_size), p, tag);
            }
        }
        for (int p = 0; p < nproc; ++p) {
            if (in_counts[p] > 0) {
                comm().wait(rreq[p]);
            }
        }
        if (!device_msgs) {
            check(atlas_amd__device_memcpy_h2d(in_dev, in_host.data(), size_t(in_size) * sizeof(DATA_TYPE)));
        }
        check(atlas_amd__HaloExchange__field_op(h_, adjoint ? 5 : 3, dtype<DATA_TYPE>(), view.data(), RANK, shape, strides,
                                                parallelDim, in_dev, 1));
        if (adjoint) {
            check(atlas_amd__HaloExchange__field_op(h_, 6, dtype<DATA_TYPE>(), view.data(), RANK, shape, strides, parallelDim,
                                                    nullptr, 1));
        }
        check(atlas_amd__HaloExchange__synchronize(h_));
        for (int p = 0; p < nproc; ++p) {
            if (out_counts[p] > 0) {
                comm().wait(sreq[p]);
            }
        }
        util::delete_devicemem(out_dev, out_size);
        util::delete_devicemem(in_dev, in_size);
        if (on_device) {
            field.setHostNeedsUpdate(true);
        }
        else {
            field.setHostNeedsUpdate(true);
            field.updateHost();
        }
    }

private:  // data
    std::string name_;
    atlas_amd_HaloExchange* h_;
    bool is_setup_ = false;
    int sendcnt_ = 0, recvcnt_ = 0;
    std::vector<int> sendcounts_, senddispls_, recvcounts_, recvdispls_;
    int nproc = 1, myproc = 0;
    const mpi::Comm* comm_ = nullptr;
};

}  // namespace parallel
}  // namespace atlas
