// See comm.h.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>

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
}  // namespace

// ---------------------------------------------------------------- helpers on host arrays
hipStream_t Comm::helper_stream() {
    if (!helper_stream_) {
        HIP_CHECK(hipStreamCreateWithFlags(&helper_stream_, hipStreamNonBlocking));
    }
    return helper_stream_;
}
void Comm::release_helper_stream() {
    if (helper_stream_) {
        (void)hipStreamDestroy(helper_stream_);
        helper_stream_ = nullptr;
    }
}

void Comm::all_to_allv(const int* send, const int* sendcounts, int* recv, const int* recvcounts) {
    const int n        = size();
    for (int p = 0; p < n; ++p) {   // before anything is allocated or summed
        if (sendcounts[p] < 0 || recvcounts[p] < 0) {
            throw std::invalid_argument("Comm::all_to_allv: negative count");
        }
    }
    const size_t nsend = std::accumulate(sendcounts, sendcounts + n, (size_t)0);
    const size_t nrecv = std::accumulate(recvcounts, recvcounts + n, (size_t)0);
    int *ds = nullptr, *dr = nullptr;
    HIP_CHECK(hipMalloc((void**)&ds, std::max<size_t>(nsend, 1) * sizeof(int)));
    HIP_CHECK(hipMalloc((void**)&dr, std::max<size_t>(nrecv, 1) * sizeof(int)));
    hipStream_t st = helper_stream();
    try {
        if (nsend) {
            HIP_CHECK(hipMemcpyAsync(ds, send, nsend * sizeof(int), hipMemcpyHostToDevice, st));
        }
        std::vector<Msg> s, r;
        size_t so = 0, ro = 0;
        for (int p = 0; p < n; ++p) {
            if (sendcounts[p] < 0 || recvcounts[p] < 0) {
                throw std::invalid_argument("Comm::all_to_allv: negative count");
            }
            if (sendcounts[p]) {
                s.push_back(Msg{p, ds + so, (size_t)sendcounts[p] * sizeof(int)});
            }
            if (recvcounts[p]) {
                r.push_back(Msg{p, dr + ro, (size_t)recvcounts[p] * sizeof(int)});
            }
            so += sendcounts[p];
            ro += recvcounts[p];
        }
        exchange(s, r, st);
        if (nrecv) {
            HIP_CHECK(hipMemcpyAsync(recv, dr, nrecv * sizeof(int), hipMemcpyDeviceToHost, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
    }
    catch (...) {
        (void)hipFree(ds);
        (void)hipFree(dr);
        throw;
    }
    (void)hipFree(ds);
    (void)hipFree(dr);
}

void Comm::all_to_all(const int* send, int* recv, int n_per_peer) {
    std::vector<int> cnt(size(), n_per_peer);
    all_to_allv(send, cnt.data(), recv, cnt.data());
}

void Comm::barrier() {
    std::vector<int> a(size(), 1), b(size(), 0);
    all_to_all(a.data(), b.data(), 1);
}

// ---------------------------------------------------------------- RCCL (bound at first use)
namespace {
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                     = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                               = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                         = nullptr;
    ncclResult_t (*GroupStart)()                                                                    = nullptr;
    ncclResult_t (*GroupEnd)()                                                                      = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)         = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)               = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                     = nullptr;
    std::string where;
};

const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        // a process that already carries an RCCL (PyTorch brings its own copy) keeps using that one
        void* h = nullptr;
        if (dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
            h       = RTLD_DEFAULT;
            a.where = "process image";
        }
        else {
            // a copy that is mapped but not in the global namespace (torch loads its bundled RCCL RTLD_LOCAL): take THAT one
            // -- RTLD_NOLOAD finds it by soname without mapping anything -- before a second RCCL enters the process
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
                if (h) {
                    a.where = std::string(name) + " (already mapped)";
                    break;
                }
            }
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                if (h) {
                    break;
                }
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    a.where = name;
                }
            }
        }
        if (!h && a.where.empty()) {
            throw std::runtime_error("RCCL (librccl.so) not found: the distributed paths of atlas_amd need it");
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(h, n);
            if (!p) {
                throw std::runtime_error(std::string("RCCL symbol missing: ") + n);
            }
            return p;
        };
        a.GetUniqueId    = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank   = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy    = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.GroupStart     = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd       = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.Send           = (decltype(a.Send))sym("ncclSend");
        a.Recv           = (decltype(a.Recv))sym("ncclRecv");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}

void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) {
        throw std::runtime_error(std::string("RCCL error in ") + what + ": " + rccl().GetErrorString(r));
    }
}
#define NCCL_CHECK(x) nccl_check((x), #x)

// one message never exceeds this (RCCL 2.26 delivered only the first half of >= 2 GiB messages of an
// all_to_all_single: profiles/r01_rccl_message_probe.txt); longer ones are cut, identically on both ends
constexpr size_t RCCL_MAX_MESSAGE_BYTES = size_t(512) << 20;

class RcclComm : public Comm {
public:
    RcclComm(const void* id128, int nranks, int rank) : n_(nranks), r_(rank) {
        static_assert(sizeof(ncclUniqueId) == UNIQUE_ID_BYTES, "ncclUniqueId size");
        if (nranks < 1 || rank < 0 || rank >= nranks || !id128) {
            throw std::invalid_argument("RcclComm: bad (unique id, nranks, rank)");
        }
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        NCCL_CHECK(rccl().CommInitRank(&comm_, nranks, id, rank));
    }
    ~RcclComm() override {
        release_helper_stream();
        if (comm_) {
            (void)rccl().CommDestroy(comm_);
        }
    }
    int size() const override { return n_; }
    int rank() const override { return r_; }
    const char* kind() const override { return "rccl"; }

    void exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t stream) override {
        // the rank's own part (periodic / pole duplicates of a halo, the diagonal block of a transposition): device copy
        std::vector<const Msg*> self_s, self_r;
        for (const Msg& m : sends) {
            if (m.peer == r_) {
                self_s.push_back(&m);
            }
        }
        for (const Msg& m : recvs) {
            if (m.peer == r_) {
                self_r.push_back(&m);
            }
        }
        if (self_s.size() != self_r.size()) {
            throw std::logic_error("Comm::exchange: unmatched messages to self");
        }
        for (size_t i = 0; i < self_s.size(); ++i) {
            if (self_s[i]->bytes != self_r[i]->bytes) {
                throw std::logic_error("Comm::exchange: self message sizes differ");
            }
            if (self_s[i]->bytes) {
                HIP_CHECK(hipMemcpyAsync(self_r[i]->ptr, self_s[i]->ptr, self_s[i]->bytes, hipMemcpyDeviceToDevice, stream));
            }
        }
        bool any = false;
        for (const Msg& m : sends) {
            any = any || (m.peer != r_ && m.bytes);
        }
        for (const Msg& m : recvs) {
            any = any || (m.peer != r_ && m.bytes);
        }
        if (!any) {
            return;
        }
        NCCL_CHECK(rccl().GroupStart());
        for (const Msg& m : recvs) {
            if (m.peer == r_) {
                continue;
            }
            for (size_t o = 0; o < m.bytes; o += RCCL_MAX_MESSAGE_BYTES) {
                const size_t nb = std::min(RCCL_MAX_MESSAGE_BYTES, m.bytes - o);
                NCCL_CHECK(rccl().Recv((char*)m.ptr + o, nb, ncclInt8, m.peer, comm_, stream));
            }
        }
        for (const Msg& m : sends) {
            if (m.peer == r_) {
                continue;
            }
            for (size_t o = 0; o < m.bytes; o += RCCL_MAX_MESSAGE_BYTES) {
                const size_t nb = std::min(RCCL_MAX_MESSAGE_BYTES, m.bytes - o);
                NCCL_CHECK(rccl().Send((const char*)m.ptr + o, nb, ncclInt8, m.peer, comm_, stream));
            }
        }
        NCCL_CHECK(rccl().GroupEnd());
    }

private:
    int n_, r_;
    ncclComm_t comm_ = nullptr;
};
}  // namespace

void rccl_get_unique_id(void* out128) {
    ncclUniqueId id;
    NCCL_CHECK(rccl().GetUniqueId(&id));
    std::memcpy(out128, &id, sizeof(id));
}

std::unique_ptr<Comm> make_rccl_comm(const void* unique_id128, int nranks, int rank) {
    return std::unique_ptr<Comm>(new RcclComm(unique_id128, nranks, rank));
}

// ---------------------------------------------------------------- emulated ranks (threads of one process)
LocalHub::LocalHub(int nranks) : n_(nranks) {
    if (nranks < 1) {
        throw std::invalid_argument("LocalHub: nranks >= 1");
    }
    sends_.assign(nranks, nullptr);
    ready_.resize(nranks);
    done_.resize(nranks);
    for (int i = 0; i < nranks; ++i) {
        HIP_CHECK(hipEventCreateWithFlags(&ready_[i], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&done_[i], hipEventDisableTiming));
    }
}
LocalHub::~LocalHub() {
    for (auto e : ready_) {
        (void)hipEventDestroy(e);
    }
    for (auto e : done_) {
        (void)hipEventDestroy(e);
    }
}
void LocalHub::rendezvous() {
    std::unique_lock<std::mutex> lk(m_);
    if (failed_) {
        throw std::runtime_error("Comm::exchange: another rank of this hub failed: " + failure_);
    }
    const long long g = gen_;
    if (++arrived_ == n_) {
        arrived_ = 0;
        ++gen_;
        cv_.notify_all();
    }
    else {
        cv_.wait(lk, [&] { return gen_ != g || failed_; });
        if (failed_) {
            throw std::runtime_error("Comm::exchange: another rank of this hub failed: " + failure_);
        }
    }
}
// a rank that throws between two meeting points would leave the others waiting for ever: it marks the hub failed first,
// every waiter (and every later arrival) then throws as well
void LocalHub::fail(const std::string& what) {
    std::lock_guard<std::mutex> lk(m_);
    if (!failed_) {
        failed_  = true;
        failure_ = what;
    }
    cv_.notify_all();
}

class LocalComm : public Comm {
public:
    LocalComm(std::shared_ptr<LocalHub> hub, int rank) : hub_(std::move(hub)), r_(rank) {
        if (!hub_ || rank < 0 || rank >= hub_->size()) {
            throw std::invalid_argument("LocalComm: bad (hub, rank)");
        }
    }
    ~LocalComm() override { release_helper_stream(); }
    int size() const override { return hub_->size(); }
    int rank() const override { return r_; }
    const char* kind() const override { return "local"; }

    void exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t stream) override {
        try {
            exchange_impl(sends, recvs, stream);
        }
        catch (const std::exception& e) {
            hub_->fail(e.what());   // releases the ranks that wait at a meeting point (a failing test fails, it does not hang)
            throw;
        }
    }

private:
    void exchange_impl(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t stream) {
        LocalHub& h = *hub_;
        const int n = h.size();
        // 1. my send buffers are complete once the work already in my stream has run
        HIP_CHECK(hipEventRecord(h.ready_[r_], stream));
        h.sends_[r_] = &sends;
        h.rendezvous();
        // 2. pull: the k-th receive from peer p is the k-th message p addressed to me
        std::vector<size_t> next(n, 0);
        std::vector<char> waited(n, 0);
        for (const Msg& m : recvs) {
            if (m.peer < 0 || m.peer >= n) {
                throw std::invalid_argument("Comm::exchange: peer out of range");
            }
            const std::vector<Msg>& ps = *h.sends_[m.peer];
            size_t& k                  = next[m.peer];
            while (k < ps.size() && ps[k].peer != r_) {
                ++k;
            }
            if (k >= ps.size() || ps[k].bytes != m.bytes) {
                throw std::logic_error("Comm::exchange: a receive has no matching send of the same size");
            }
            if (!waited[m.peer]) {
                HIP_CHECK(hipStreamWaitEvent(stream, h.ready_[m.peer], 0));
                waited[m.peer] = 1;
            }
            if (m.bytes) {
                HIP_CHECK(hipMemcpyAsync(m.ptr, ps[k].ptr, m.bytes, hipMemcpyDeviceToDevice, stream));
            }
            ++k;
        }
        for (int p = 0; p < n; ++p) {   // every message addressed to me has been received
            const std::vector<Msg>& ps = *h.sends_[p];
            size_t left                = 0;
            for (size_t k = next[p]; k < ps.size(); ++k) {
                left += ps[k].peer == r_;
            }
            if (left) {
                throw std::logic_error("Comm::exchange: a send has no matching receive");
            }
        }
        // 3. my copies are in my stream: the senders' later work must follow them
        HIP_CHECK(hipEventRecord(h.done_[r_], stream));
        h.rendezvous();
        for (const Msg& m : sends) {
            if (m.peer != r_ && !waited_done(m.peer)) {
                HIP_CHECK(hipStreamWaitEvent(stream, h.done_[m.peer], 0));
                mark_done(m.peer);
            }
        }
        done_marks_.clear();
        // 4. nobody re-records the events of the next exchange before every wait above has been enqueued
        h.rendezvous();
    }

    bool waited_done(int p) const { return std::find(done_marks_.begin(), done_marks_.end(), p) != done_marks_.end(); }
    void mark_done(int p) { done_marks_.push_back(p); }
    std::shared_ptr<LocalHub> hub_;
    int r_;
    std::vector<int> done_marks_;
};

std::unique_ptr<Comm> make_local_comm(std::shared_ptr<LocalHub> hub, int rank) {
    return std::unique_ptr<Comm>(new LocalComm(std::move(hub), rank));
}

}  // namespace parallel
}  // namespace atlas_amd
