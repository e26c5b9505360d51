// See halo_exchange.h.
#include "halo_exchange.h"

#include "trace.h"

#include <cstring>
#include <sstream>
#include <stdexcept>

namespace atlas_amd {
namespace parallel {

namespace {
void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        std::ostringstream ss;
        ss << "HIP error '" << hipGetErrorString(e) << "' in " << what << " (" << file << ":" << line << ")";
        (void)hipGetLastError();   // the failure is reported here, once: not left sticky for the launch checks of the next call
        throw std::runtime_error(ss.str());
    }
}
#define HIP_CHECK(x) hip_check((x), #x, __FILE__, __LINE__)

int* upload_ints(const std::vector<int>& v) {
    int* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, std::max<size_t>(v.size(), 1) * sizeof(int)));
    if (!v.empty()) {
        HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return d;
}
}  // namespace

HaloExchange::HaloExchange() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return;  // setup (host logic) still works; execute will fail loudly
    }
    has_device_ = true;
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    own_stream_ = true;
}

HaloExchange::~HaloExchange() {
    if (has_device_) {
        (void)hipStreamSynchronize(stream_);
    }
    for (void* p : {(void*)d_sendmap_, (void*)d_recvmap_, (void*)d_adj_nodes_, (void*)d_adj_start_,
                    (void*)d_adj_items_, d_scratch_[0], d_scratch_[1], d_scratch_[2]}) {
        if (p) {
            (void)hipFree(p);
        }
    }
    if (own_stream_ && stream_) {
        (void)hipStreamDestroy(stream_);
    }
}

void HaloExchange::set_stream(hipStream_t s) {
    if (!has_device_) {
        throw std::runtime_error("HaloExchange: no HIP device");
    }
    synchronize();
    if (own_stream_ && stream_) {
        (void)hipStreamDestroy(stream_);
    }
    stream_     = s;
    own_stream_ = false;
}

void HaloExchange::synchronize() const {
    if (has_device_) {
        HIP_CHECK(hipStreamSynchronize(stream_));
    }
}

void HaloExchange::setup(const int part[], const int remote_idx[], int base, int parsize, int halo_begin) {
    halo_setup_serial(plan_, part, remote_idx, base, parsize, halo_begin);
    upload_maps();
}

void HaloExchange::setup_begin(int nproc, int myproc, const int part[], const int remote_idx[], int base,
                               int parsize, int halo_begin) {
    halo_setup_local(plan_, nproc, myproc, part, remote_idx, base, parsize, halo_begin);
}

void HaloExchange::setup_begin_device(int nproc, int myproc, const int* part_dev, const int* ridx_dev, int base,
                                      int parsize, int halo_begin) {
    if (!has_device_) {
        throw std::runtime_error("HaloExchange: no HIP device");
    }
    const int n       = std::max(parsize - halo_begin, 0);
    const int nblocks = (n + 255) / 256;
    std::vector<int> counts(nblocks, 0), offsets(nblocks + 1, 0);
    int* d_counts = (int*)scratch((size_t)(nblocks + 1) * sizeof(int), 0);
    HIP_CHECK(launch_ghost_count(part_dev, ridx_dev, myproc, base, halo_begin, parsize, d_counts, nblocks, stream_));
    if (nblocks) {
        HIP_CHECK(hipMemcpyAsync(counts.data(), d_counts, nblocks * sizeof(int), hipMemcpyDeviceToHost, stream_));
    }
    synchronize();
    for (int b = 0; b < nblocks; ++b) {
        offsets[b + 1] = offsets[b] + counts[b];
    }
    const int nghost = offsets[nblocks];
    if (nblocks) {
        HIP_CHECK(hipMemcpyAsync(d_counts, offsets.data(), nblocks * sizeof(int), hipMemcpyHostToDevice, stream_));
    }
    int* d_ghosts = (int*)scratch((size_t)std::max(nghost, 1) * 3 * sizeof(int), 1);
    HIP_CHECK(launch_ghost_compact(part_dev, ridx_dev, myproc, base, halo_begin, parsize, d_counts, d_ghosts,
                                   nblocks, stream_));
    // gather part / remote_idx of the ghosts (payload 1 per node)
    HaloFieldDesc d{};
    d.node_stride = 1;
    d.next        = 0;
    d.var_size    = 1;
    HIP_CHECK(launch_halo_copy(0, HALO_INT, (void*)part_dev, d_ghosts + nghost, d_ghosts, nghost, d, stream_));
    HIP_CHECK(launch_halo_copy(0, HALO_INT, (void*)ridx_dev, d_ghosts + 2 * (size_t)nghost, d_ghosts, nghost, d,
                               stream_));
    std::vector<int> g(3 * (size_t)std::max(nghost, 1));
    if (nghost) {
        HIP_CHECK(hipMemcpyAsync(g.data(), d_ghosts, 3 * (size_t)nghost * sizeof(int), hipMemcpyDeviceToHost, stream_));
    }
    synchronize();
    // same grouping as halo_setup_local, from the compacted (ascending) ghost list
    std::vector<int> part(parsize, myproc), ridx(parsize);
    for (int i = 0; i < parsize; ++i) {
        ridx[i] = base + i;
    }
    for (int i = 0; i < nghost; ++i) {
        part[g[i]] = g[nghost + i];
        ridx[g[i]] = g[2 * (size_t)nghost + i];
    }
    halo_setup_local(plan_, nproc, myproc, part.data(), ridx.data(), base, parsize, halo_begin);
}

void HaloExchange::setup_finish(const int sendcounts[], const int recv_requests[]) {
    halo_setup_finish(plan_, sendcounts, recv_requests);
    upload_maps();
}

void HaloExchange::upload_maps() {
    if (!has_device_) {
        return;  // no device: host logic only
    }
    synchronize();
    for (int** p : {&d_sendmap_, &d_recvmap_, &d_adj_nodes_, &d_adj_start_, &d_adj_items_}) {
        if (*p) {
            HIP_CHECK(hipFree(*p));
            *p = nullptr;
        }
    }
    d_sendmap_   = upload_ints(plan_.sendmap);
    d_recvmap_   = upload_ints(plan_.recvmap);
    d_adj_nodes_ = upload_ints(plan_.adj_nodes);
    d_adj_start_ = upload_ints(plan_.adj_start);
    d_adj_items_ = upload_ints(plan_.adj_items);
}

void* HaloExchange::scratch(size_t bytes, int which) {
    if (bytes > scratch_cap_[which]) {
        synchronize();
        if (d_scratch_[which]) {
            HIP_CHECK(hipFree(d_scratch_[which]));
        }
        HIP_CHECK(hipMalloc(&d_scratch_[which], bytes));
        scratch_cap_[which] = bytes;
    }
    return d_scratch_[which];
}

HaloFieldDesc HaloExchange::describe(int rank, const int shape[], const long long strides[], int parallel_dim) const {
    if (rank < 1 || rank > 4) {
        throw std::invalid_argument("Rank not supported in halo exchange");  // HaloExchange.cc:229
    }
    if (parallel_dim < 0 || parallel_dim >= rank) {
        throw std::invalid_argument("HaloExchange: bad parallel dimension");
    }
    if (shape[parallel_dim] != plan_.parsize) {
        throw std::invalid_argument("HaloExchange: extent of the parallel dimension differs from the setup size");
    }
    HaloFieldDesc d{};
    d.node_stride = strides[parallel_dim];
    d.var_size    = 1;
    int ext[4];
    long long str[4];
    int n = 0;
    for (int i = 0; i < rank; ++i) {
        if (i == parallel_dim) {
            continue;
        }
        d.var_size *= shape[i];
        if (shape[i] == 1) {
            continue;
        }
        if (n > 0 && str[n - 1] == (long long)shape[i] * strides[i]) {
            ext[n - 1] *= shape[i];  // merge contiguous dimensions
            str[n - 1] = strides[i];
        }
        else {
            ext[n] = shape[i];
            str[n] = strides[i];
            ++n;
        }
    }
    d.next = n;
    for (int i = 0; i < 3; ++i) {
        d.ext[i] = i < n ? ext[i] : 1;
        d.str[i] = i < n ? str[i] : 0;
    }
    return d;
}

#define NEED_SETUP()                                                                        \
    if (!plan_.finished) throw std::runtime_error("HaloExchange was not setup"); /* HaloExchange.h:155 */ \
    if (!has_device_) throw std::runtime_error("HaloExchange: no HIP device")

void HaloExchange::pack_device(int dtype, const void* field, const HaloFieldDesc& d, void* sendbuf) {
    NEED_SETUP();
    HIP_CHECK(launch_halo_copy(0, dtype, (void*)field, sendbuf, d_sendmap_, plan_.sendcnt, d, stream_));
}
void HaloExchange::unpack_device(int dtype, void* field, const HaloFieldDesc& d, const void* recvbuf) {
    NEED_SETUP();
    HIP_CHECK(launch_halo_copy(1, dtype, field, (void*)recvbuf, d_recvmap_, plan_.recvcnt, d, stream_));
}
void HaloExchange::pack_adjoint_device(int dtype, const void* field, const HaloFieldDesc& d, void* buf) {
    NEED_SETUP();
    HIP_CHECK(launch_halo_copy(0, dtype, (void*)field, buf, d_recvmap_, plan_.recvcnt, d, stream_));
}
void HaloExchange::unpack_adjoint_device(int dtype, void* field, const HaloFieldDesc& d, const void* buf) {
    NEED_SETUP();
    HIP_CHECK(launch_halo_adjoint_add(dtype, field, buf, d_adj_nodes_, d_adj_start_, d_adj_items_,
                                      (int)plan_.adj_nodes.size(), d, stream_));
}
void HaloExchange::zero_halos_device(int dtype, void* field, const HaloFieldDesc& d) {
    NEED_SETUP();
    HIP_CHECK(launch_halo_copy(2, dtype, field, nullptr, d_recvmap_, plan_.recvcnt, d, stream_));
}

void HaloExchange::execute_device(int dtype, void* field, const HaloFieldDesc& d) {
    TraceRange trace("HaloExchange::execute[device]");   // HaloExchange.h:153
    NEED_SETUP();
    if (plan_.nproc != 1) {
        throw std::logic_error("execute_device: multi-process exchange is driven by the caller (pack / send / unpack)");
    }
    // HaloExchange.h:191-215 for one rank: the self-send delivers the send buffer as the receive buffer
    void* buf = scratch((size_t)plan_.sendcnt * d.var_size * halo_dtype_size(dtype), 0);
    pack_device(dtype, field, d, buf);
    unpack_device(dtype, field, d, buf);
}

void HaloExchange::execute_adjoint_device(int dtype, void* field, const HaloFieldDesc& d) {
    TraceRange trace("HaloExchange::execute_adjoint[device]");   // HaloExchange.h:232
    NEED_SETUP();
    if (plan_.nproc != 1) {
        throw std::logic_error("execute_adjoint_device: multi-process exchange is driven by the caller");
    }
    // HaloExchange.h:258-279
    void* buf = scratch((size_t)plan_.recvcnt * d.var_size * halo_dtype_size(dtype), 0);
    pack_adjoint_device(dtype, field, d, buf);
    unpack_adjoint_device(dtype, field, d, buf);
    zero_halos_device(dtype, field, d);
}

void HaloExchange::setup_comm(Comm& comm, const int part[], const int remote_idx[], int base, int parsize,
                              int halo_begin) {
    const int n = comm.size();
    halo_setup_local(plan_, n, comm.rank(), part, remote_idx, base, parsize, halo_begin);
    std::vector<int> sendcounts(n, 0);
    comm.all_to_all(plan_.recvcounts.data(), sendcounts.data(), 1);                       // HaloExchange.cc:118
    size_t nreq = 0;
    for (int c : sendcounts) {
        nreq += (size_t)c;
    }
    std::vector<int> requests(std::max<size_t>(nreq, 1), 0);
    comm.all_to_allv(plan_.send_requests.data(), plan_.recvcounts.data(), requests.data(), sendcounts.data());   // :156
    halo_setup_finish(plan_, sendcounts.data(), requests.data());
    upload_maps();
}

void HaloExchange::execute_comm(Comm& comm, int dtype, void* field, const HaloFieldDesc& d, bool adjoint) {
    TraceRange trace(adjoint ? "HaloExchange::execute_adjoint[device]" : "HaloExchange::execute[device]");
    NEED_SETUP();
    if (comm.size() != plan_.nproc || comm.rank() != plan_.myproc) {
        throw std::invalid_argument("HaloExchange::execute: communicator differs from the one of the setup");
    }
    const size_t esz = (size_t)d.var_size * halo_dtype_size(dtype);
    const std::vector<int>& out_cnt = adjoint ? plan_.recvcounts : plan_.sendcounts;
    const std::vector<int>& out_dsp = adjoint ? plan_.recvdispls : plan_.senddispls;
    const std::vector<int>& in_cnt  = adjoint ? plan_.sendcounts : plan_.recvcounts;
    const std::vector<int>& in_dsp  = adjoint ? plan_.senddispls : plan_.recvdispls;
    const int out_total             = adjoint ? plan_.recvcnt : plan_.sendcnt;
    const int in_total              = adjoint ? plan_.sendcnt : plan_.recvcnt;
    char* outbuf = (char*)scratch(std::max<size_t>((size_t)out_total * esz, 16), 0);
    char* inbuf  = (char*)scratch(std::max<size_t>((size_t)in_total * esz, 16), 1);
    if (adjoint) {
        pack_adjoint_device(dtype, field, d, outbuf);
    }
    else {
        pack_device(dtype, field, d, outbuf);
    }
    std::vector<Msg> sends, recvs;
    for (int p = 0; p < plan_.nproc; ++p) {
        if (out_cnt[p]) {
            sends.push_back(Msg{p, outbuf + (size_t)out_dsp[p] * esz, (size_t)out_cnt[p] * esz});
        }
        if (in_cnt[p]) {
            recvs.push_back(Msg{p, inbuf + (size_t)in_dsp[p] * esz, (size_t)in_cnt[p] * esz});
        }
    }
    comm.exchange(sends, recvs, stream_);
    if (adjoint) {
        unpack_adjoint_device(dtype, field, d, inbuf);
        zero_halos_device(dtype, field, d);
    }
    else {
        unpack_device(dtype, field, d, inbuf);
    }
}

void HaloExchange::execute_comm_on(Comm& comm, int dtype, void* field, const HaloFieldDesc& d, bool adjoint, hipStream_t s) {
    struct Swap {   // the kernels and the exchange below launch on stream_
        hipStream_t& ref;
        hipStream_t keep;
        Swap(hipStream_t& r, hipStream_t s) : ref(r), keep(r) { ref = s; }
        ~Swap() { ref = keep; }
    } swap(stream_, s);
    execute_comm(comm, dtype, field, d, adjoint);
}

void HaloExchange::execute_host(int dtype, void* field, int rank, const int shape[], const long long strides[],
                                int parallel_dim, bool adjoint) {
    if (!plan_.finished) {
        throw std::runtime_error("HaloExchange was not setup");  // HaloExchange.h:155
    }
    if (!has_device_) {
        throw std::runtime_error("HaloExchange::execute needs a HIP device; there is no CPU fallback");
    }
    const HaloFieldDesc d = describe(rank, shape, strides, parallel_dim);
    long long span        = 1;
    for (int i = 0; i < rank; ++i) {
        if (strides[i] < 0) {
            throw std::invalid_argument("HaloExchange: negative strides are not supported");
        }
        if (shape[i] <= 0) {
            return;  // empty field (e.g. a part without points): nothing to exchange
        }
        span += (long long)(shape[i] - 1) * strides[i];
    }
    const size_t bytes = (size_t)span * halo_dtype_size(dtype);
    void* dev          = scratch(bytes, 2);
    HIP_CHECK(hipMemcpyAsync(dev, field, bytes, hipMemcpyHostToDevice, stream_));
    if (adjoint) {
        execute_adjoint_device(dtype, dev, d);
    }
    else {
        execute_device(dtype, dev, d);
    }
    HIP_CHECK(hipMemcpyAsync(field, dev, bytes, hipMemcpyDeviceToHost, stream_));
    synchronize();
}

}  // namespace parallel
}  // namespace atlas_amd
