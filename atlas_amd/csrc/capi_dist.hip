// extern "C" layer of include/atlas_amd.h: communicators, distributed inverse transform, halo exchange between ranks.
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../include/atlas_amd.h"
#include "capi_types.h"
#include "comm.h"
#include "dist_trans.h"
#include "halo_exchange.h"
#include "trans.h"

namespace atlas_amd {
void set_last_error(const std::string& s);
}

struct atlas_amd_CommHub {
    std::shared_ptr<atlas_amd::parallel::LocalHub> hub;
};
struct atlas_amd_Comm {
    // shared: a DistributedTrans made for this communicator co-owns it, so deleting the handle while a Trans still uses it
    // cannot leave a dangling reference (the communicator then lives until that Trans is deleted or switches)
    std::shared_ptr<atlas_amd::parallel::Comm> impl;
    unsigned long long id = next_id();   // serial number: handles are compared by it, not by address
    static unsigned long long next_id() {
        static std::atomic<unsigned long long> counter{0};
        return ++counter;
    }
};

#define DX_TRY try {
#define DX_CATCH                                    \
    }                                               \
    catch (const std::exception& e) {               \
        atlas_amd::set_last_error(e.what());        \
        return 1;                                   \
    }                                               \
    catch (...) {                                   \
        atlas_amd::set_last_error("unknown error"); \
        return 1;                                   \
    }                                               \
    return 0;
#define DX_CATCH_PTR                                \
    }                                               \
    catch (const std::exception& e) {               \
        atlas_amd::set_last_error(e.what());        \
        return nullptr;                             \
    }                                               \
    catch (...) {                                   \
        atlas_amd::set_last_error("unknown error"); \
        return nullptr;                             \
    }

extern "C" {

int atlas_amd__Comm__unique_id_bytes(void) {
    return atlas_amd::parallel::UNIQUE_ID_BYTES;
}
int atlas_amd__Comm__get_unique_id(void* out) {
    DX_TRY
    if (!out) {
        throw std::invalid_argument("Comm::get_unique_id: null argument");
    }
    atlas_amd::parallel::rccl_get_unique_id(out);
    DX_CATCH
}
atlas_amd_Comm* atlas_amd__Comm__new_rccl(const void* unique_id, int nranks, int rank) {
    DX_TRY
    auto* c = new atlas_amd_Comm();
    try {
        c->impl = atlas_amd::parallel::make_rccl_comm(unique_id, nranks, rank);
    }
    catch (...) {
        delete c;
        throw;
    }
    return c;
    DX_CATCH_PTR
}
atlas_amd_CommHub* atlas_amd__CommHub__new(int nranks) {
    DX_TRY
    auto* h = new atlas_amd_CommHub();
    try {
        h->hub = std::make_shared<atlas_amd::parallel::LocalHub>(nranks);
    }
    catch (...) {
        delete h;
        throw;
    }
    return h;
    DX_CATCH_PTR
}
void atlas_amd__CommHub__delete(atlas_amd_CommHub* h) {
    delete h;
}
atlas_amd_Comm* atlas_amd__Comm__new_local(atlas_amd_CommHub* hub, int rank) {
    DX_TRY
    if (!hub) {
        throw std::invalid_argument("Comm::new_local: null hub");
    }
    auto* c = new atlas_amd_Comm();
    try {
        c->impl = atlas_amd::parallel::make_local_comm(hub->hub, rank);
    }
    catch (...) {
        delete c;
        throw;
    }
    return c;
    DX_CATCH_PTR
}
void atlas_amd__Comm__delete(atlas_amd_Comm* c) {
    delete c;
}
int atlas_amd__Comm__size(const atlas_amd_Comm* c) {
    if (!c) {
        atlas_amd::set_last_error("Comm::size: null handle");
        return -1;
    }
    return c && c->impl ? c->impl->size() : 0;
}
int atlas_amd__Comm__rank(const atlas_amd_Comm* c) {
    if (!c) {
        atlas_amd::set_last_error("Comm::rank: null handle");
        return -1;
    }
    return c && c->impl ? c->impl->rank() : -1;
}
const char* atlas_amd__Comm__kind(const atlas_amd_Comm* c) {
    if (!c) {
        atlas_amd::set_last_error("Comm::kind: null handle");
        return nullptr;
    }
    return c && c->impl ? c->impl->kind() : "";
}
int atlas_amd__Comm__barrier(atlas_amd_Comm* c) {
    DX_TRY
    if (!c) {
        throw std::invalid_argument("Comm::barrier: null communicator");
    }
    c->impl->barrier();
    DX_CATCH
}
int atlas_amd__Comm__exchange(atlas_amd_Comm* c, int nsend, const int send_peer[], void* const send_ptr[],
                              const size_t send_bytes[], int nrecv, const int recv_peer[], void* const recv_ptr[],
                              const size_t recv_bytes[], void* stream) {
    DX_TRY
    if (!c || nsend < 0 || nrecv < 0) {
        throw std::invalid_argument("Comm::exchange: bad arguments");
    }
    std::vector<atlas_amd::parallel::Msg> s(nsend), r(nrecv);
    for (int i = 0; i < nsend; ++i) {
        s[i] = atlas_amd::parallel::Msg{send_peer[i], send_ptr[i], send_bytes[i]};
    }
    for (int i = 0; i < nrecv; ++i) {
        r[i] = atlas_amd::parallel::Msg{recv_peer[i], recv_ptr[i], recv_bytes[i]};
    }
    c->impl->exchange(s, r, (hipStream_t)stream);
    DX_CATCH
}

// ---------------------------------------------------------------- distributed transform
static atlas_amd::trans::DistributedTrans& dist_of(atlas_amd_Trans* t, atlas_amd_Comm* c) {
    if (!t || !c || !c->impl) {
        throw std::invalid_argument("invtrans_distributed: null Trans / Comm");
    }
    if (!t->dist || t->dist_comm != c || t->dist_comm_id != c->id) {
        t->dist.reset();
        // the deleter keeps the communicator alive as long as the DistributedTrans that refers to it
        std::shared_ptr<atlas_amd::parallel::Comm> keep = c->impl;
        t->dist = std::shared_ptr<atlas_amd::trans::DistributedTrans>(
            new atlas_amd::trans::DistributedTrans(*t->impl, *keep),
            [keep](atlas_amd::trans::DistributedTrans* d) { delete d; });
        t->dist_comm    = c;
        t->dist_comm_id = c->id;
    }
    return *t->dist;
}

int atlas_amd__Trans__invtrans_distributed(atlas_amd_Trans* t, atlas_amd_Comm* c, int nb_fields, const double* sp_dev,
                                           double* gp_dev) {
    DX_TRY
    dist_of(t, c).invtrans(nb_fields, sp_dev, gp_dev);
    DX_CATCH
}
int atlas_amd__Trans__invtrans_distributed_many(atlas_amd_Trans* t, atlas_amd_Comm* c, int ntransforms, int nb_fields,
                                                const double* const* sp_dev, double* const* gp_dev) {
    DX_TRY
    if (ntransforms > 0 && (!sp_dev || !gp_dev)) {
        throw std::invalid_argument("invtrans_distributed_many: null arrays");
    }
    dist_of(t, c).invtrans_many(ntransforms, nb_fields, sp_dev, gp_dev);
    DX_CATCH
}
int atlas_amd__Trans__invtrans_distributed_sharded(atlas_amd_Trans* t, atlas_amd_Comm* c, int ntransforms, int nb_fields,
                                                   const double* const* sp_shard_dev, double* const* gp_dev) {
    DX_TRY
    if (ntransforms > 0 && (!sp_shard_dev || !gp_dev)) {
        throw std::invalid_argument("invtrans_distributed_sharded: null arrays");
    }
    dist_of(t, c).invtrans_many_sharded(ntransforms, nb_fields, sp_shard_dev, gp_dev);
    DX_CATCH
}
int atlas_amd__Trans__spectral_shard(const atlas_amd_Trans* t, long long moff_out[], long long* size_per_field) {
    DX_TRY
    if (!t || !t->impl || !size_per_field) {
        throw std::invalid_argument("spectral_shard: null argument");
    }
    std::vector<long long> moff;
    *size_per_field = t->impl->spectral_shard_offsets(moff);
    if (moff_out) {
        for (int m = 0; m <= t->impl->truncation(); ++m) {
            moff_out[m] = moff[m];
        }
    }
    DX_CATCH
}
int atlas_amd__Trans__invtrans_distributed_many_halo(atlas_amd_Trans* t, atlas_amd_Comm* c, int ntransforms, int nb_fields,
                                                     const double* const* sp_dev, double* const* gp_dev,
                                                     atlas_amd_HaloExchange* hx, double* const* field_dev) {
    DX_TRY
    if (ntransforms > 0 && (!sp_dev || !gp_dev || !field_dev)) {
        throw std::invalid_argument("invtrans_distributed_many_halo: null arrays");
    }
    if (!hx) {
        throw std::invalid_argument("invtrans_distributed_many_halo: null halo exchange");
    }
    for (int i = 1; i < ntransforms; ++i) {
        if (gp_dev[i] == gp_dev[i - 1] || field_dev[i] == field_dev[i - 1]) {
            throw std::invalid_argument("invtrans_distributed_many_halo: consecutive transforms need distinct output buffers");
        }
    }
    dist_of(t, c).invtrans_many_halo(ntransforms, nb_fields, sp_dev, gp_dev, hx->impl, field_dev);
    DX_CATCH
}
int atlas_amd__Trans__pack_probe(atlas_amd_Trans* t, int nb_fields, int reps, double* ms, long long* bytes) {
    DX_TRY
    if (!t || !t->impl || !ms || nb_fields < 1 || reps < 1) {
        throw std::invalid_argument("pack_probe: bad arguments");
    }
    int64_t b = 0;
    *ms       = atlas_amd::trans::pack_probe(*t->impl, nb_fields, reps, &b);
    if (bytes) {
        *bytes = b;
    }
    DX_CATCH
}
int atlas_amd__Trans__fourier_packed_probe(atlas_amd_Trans* t, int nb_fields, int reps, double* ms) {
    DX_TRY
    if (!t || !t->impl || !ms || nb_fields < 1 || reps < 1) {
        throw std::invalid_argument("fourier_packed_probe: bad arguments");
    }
    *ms = atlas_amd::trans::fourier_packed_probe(*t->impl, nb_fields, reps);
    DX_CATCH
}
int atlas_amd__Trans__set_max_message_bytes(atlas_amd_Trans* t, atlas_amd_Comm* c, long long bytes) {
    DX_TRY
    // collective: an invalid value (< 8 bytes = 0 elements) is refused INSIDE, after every rank has taken part in the comparison
    dist_of(t, c).set_max_message_elems(bytes < 8 ? 0 : bytes / 8);
    DX_CATCH
}
int atlas_amd__Trans__timings_distributed(atlas_amd_Trans* t, atlas_amd_Comm* c, double out[8], int reset) {
    DX_TRY
    if (!out) {
        throw std::invalid_argument("timings_distributed: null array");
    }
    const auto x = dist_of(t, c).exchange_timings(reset != 0);
    out[0] = x.pack_ms;
    out[1] = x.exchange_ms;
    out[2] = x.calls;
    out[3] = (double)x.bytes_sent_off_device;
    out[4] = (double)x.bytes_received_off_device;
    out[5] = (double)x.bytes_largest_peer;
    out[6] = x.peers;
    out[7] = 0;
    DX_CATCH
}
int atlas_amd__transpose_messages(int truncation, int RP, int nparts, int part, const int bands[], long long max_message_elems,
                                  int capacity, int* peer, long long* send_begin, long long* send_end, long long* recv_begin,
                                  long long* recv_end, int* count) {
    DX_TRY
    if (!bands || !count || nparts < 1) {
        throw std::invalid_argument("transpose_messages: bad arguments");
    }
    std::vector<int> b(bands, bands + nparts + 1);
    const auto plan = atlas_amd::trans::make_transpose_plan(truncation, RP, b, nparts, part);
    const auto msgs = atlas_amd::trans::transpose_messages(plan, b, RP, nparts, part, max_message_elems);
    *count          = (int)msgs.size();
    for (int i = 0; i < (int)msgs.size() && i < capacity; ++i) {
        peer[i]       = msgs[i].peer;
        send_begin[i] = msgs[i].send_begin;
        send_end[i]   = msgs[i].send_end;
        recv_begin[i] = msgs[i].recv_begin;
        recv_end[i]   = msgs[i].recv_end;
    }
    DX_CATCH
}

int atlas_amd__packed_transpose_messages(int nlats, const int row_mmax[], int cols, int nparts, int part, const int bands[],
                                         long long max_message_elems, int capacity, int* peer, long long* send_begin,
                                         long long* send_end, long long* recv_begin, long long* recv_end, int* count,
                                         long long totals[2]) {
    DX_TRY
    if (!row_mmax || !bands || !count || nparts < 1 || nlats < 0 || capacity < 0 ||
        (capacity > 0 && (!peer || !send_begin || !send_end || !recv_begin || !recv_end))) {
        throw std::invalid_argument("packed_transpose_messages: bad arguments (null array with capacity > 0?)");
    }
    std::vector<int> b(bands, bands + nparts + 1), mm(row_mmax, row_mmax + nlats);
    const auto plan = atlas_amd::trans::make_packed_transpose_plan(mm, cols, b, nparts, part);
    const auto msgs = atlas_amd::trans::packed_transpose_messages(plan, b, nparts, part, max_message_elems);
    *count          = (int)msgs.size();
    for (int i = 0; i < (int)msgs.size() && i < capacity; ++i) {
        peer[i]       = msgs[i].peer;
        send_begin[i] = msgs[i].send_begin;
        send_end[i]   = msgs[i].send_end;
        recv_begin[i] = msgs[i].recv_begin;
        recv_end[i]   = msgs[i].recv_end;
    }
    if (totals) {
        totals[0] = plan.send_total;
        totals[1] = plan.out_total;
    }
    DX_CATCH
}

// ---------------------------------------------------------------- halo exchange between ranks
int atlas_amd__HaloExchange__setup_comm(atlas_amd_HaloExchange* h, atlas_amd_Comm* c, const int part[],
                                        const int remote_idx[], int base, int size, int halo_begin) {
    DX_TRY
    if (!h || !c || !part || !remote_idx) {
        throw std::invalid_argument("HaloExchange::setup_comm: null argument");
    }
    h->impl.setup_comm(*c->impl, part, remote_idx, base, size, halo_begin);
    DX_CATCH
}
int atlas_amd__HaloExchange__execute_comm(atlas_amd_HaloExchange* h, atlas_amd_Comm* c, int dtype, void* field_dev, int rank,
                                          const int shape[], const long long strides[], int parallel_dim, int adjoint) {
    DX_TRY
    if (!h || !c || !field_dev || !shape || !strides) {
        throw std::invalid_argument("HaloExchange::execute_comm: null argument");
    }
    const atlas_amd::parallel::HaloFieldDesc d = h->impl.describe(rank, shape, strides, parallel_dim);
    h->impl.execute_comm(*c->impl, dtype, field_dev, d, adjoint != 0);
    DX_CATCH
}

}  // extern "C"
