// extern "C" layer of include/atlas_amd.h (HaloExchange part).  Replaces atlas__HaloExchange__*
// (src/atlas/parallel/HaloExchange.h:429-456, HaloExchange.cc:273-327).
#include <hip/hip_runtime.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/atlas_amd.h"
#include "capi_types.h"
#include "halo_exchange.h"

namespace atlas_amd {
void set_last_error(const std::string& s);
}
using atlas_amd::parallel::HaloExchange;
using atlas_amd::parallel::HaloFieldDesc;


#define HX_TRY try {
#define HX_CATCH                                    \
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

namespace {
// execute_halo_exchange (HaloExchange.cc:195-231): the parallel dimension is the slowest one,
// shape = {parsize, var_shape...}, strides = {var_shape[0]*var_strides[0], var_strides...}
int strided(atlas_amd_HaloExchange* h, int dtype, void* field, const int var_strides[], const int var_shape[],
            int var_rank, bool adjoint) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::execute: null handle");
    }
    if (var_rank < 0 || var_rank > 3) {
        throw std::invalid_argument("Rank not supported in halo exchange");
    }
    if (!field || (var_rank > 0 && (!var_strides || !var_shape))) {
        throw std::invalid_argument("HaloExchange::execute: null array");
    }
    int shape[4];
    long long strides[4];
    shape[0]   = h->impl.plan().parsize;
    strides[0] = var_rank > 0 ? (long long)var_shape[0] * var_strides[0] : 1;
    for (int j = 0; j < var_rank; ++j) {
        shape[j + 1]   = var_shape[j];
        strides[j + 1] = var_strides[j];
    }
    h->impl.execute_host(dtype, field, var_rank + 1, shape, strides, 0, adjoint);
    HX_CATCH
}
}  // namespace

extern "C" {

atlas_amd_HaloExchange* atlas_amd__HaloExchange__new(void) {
    try {
        return new atlas_amd_HaloExchange();
    }
    catch (const std::exception& e) {
        atlas_amd::set_last_error(e.what());
        return nullptr;
    }
}
void atlas_amd__HaloExchange__delete(atlas_amd_HaloExchange* h) {
    delete h;
}
int atlas_amd__HaloExchange__setup(atlas_amd_HaloExchange* h, const int part[], const int remote_idx[], int base,
                                   int size) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::setup: null handle");
    }
    h->impl.setup(part, remote_idx, base, size, 0);
    HX_CATCH
}
int atlas_amd__HaloExchange__setup_halo_begin(atlas_amd_HaloExchange* h, const int part[], const int remote_idx[],
                                              int base, int size, int halo_begin) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::setup_halo_begin: null handle");
    }
    h->impl.setup(part, remote_idx, base, size, halo_begin);
    HX_CATCH
}
int atlas_amd__HaloExchange__setup_begin(atlas_amd_HaloExchange* h, int nproc, int myproc, const int part[],
                                         const int remote_idx[], int base, int size, int halo_begin) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::setup_begin: null handle");
    }
    h->impl.setup_begin(nproc, myproc, part, remote_idx, base, size, halo_begin);
    HX_CATCH
}
int atlas_amd__HaloExchange__setup_begin_device(atlas_amd_HaloExchange* h, int nproc, int myproc,
                                                const int* part_dev, const int* remote_idx_dev, int base, int size,
                                                int halo_begin) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::setup_begin_device: null handle");
    }
    h->impl.setup_begin_device(nproc, myproc, part_dev, remote_idx_dev, base, size, halo_begin);
    HX_CATCH
}
int atlas_amd__HaloExchange__setup_finish(atlas_amd_HaloExchange* h, const int sendcounts[],
                                          const int recv_requests[]) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::setup_finish: null handle");
    }
    h->impl.setup_finish(sendcounts, recv_requests);
    HX_CATCH
}
int atlas_amd__HaloExchange__nproc(const atlas_amd_HaloExchange* h) {
    if (!h) {
        atlas_amd::set_last_error("HaloExchange::nproc: null handle");
        return -1;
    }
    return h->impl.plan().nproc;
}
int atlas_amd__HaloExchange__sendcnt(const atlas_amd_HaloExchange* h) {
    if (!h) {
        atlas_amd::set_last_error("HaloExchange::sendcnt: null handle");
        return -1;
    }
    return h->impl.plan().sendcnt;
}
int atlas_amd__HaloExchange__recvcnt(const atlas_amd_HaloExchange* h) {
    if (!h) {
        atlas_amd::set_last_error("HaloExchange::recvcnt: null handle");
        return -1;
    }
    return h->impl.plan().recvcnt;
}
int atlas_amd__HaloExchange__get(const atlas_amd_HaloExchange* h, const char* what, int out[]) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::get: null handle");
    }
    const auto& p           = h->impl.plan();
    const std::string w     = what ? what : "";
    const std::vector<int>* v = nullptr;
    if (w == "sendcounts") v = &p.sendcounts;
    else if (w == "recvcounts") v = &p.recvcounts;
    else if (w == "senddispls") v = &p.senddispls;
    else if (w == "recvdispls") v = &p.recvdispls;
    else if (w == "sendmap") v = &p.sendmap;
    else if (w == "recvmap") v = &p.recvmap;
    else if (w == "send_requests") v = &p.send_requests;
    else throw std::invalid_argument("HaloExchange__get: unknown array '" + w + "'");
    if (!v->empty()) {
        std::memcpy(out, v->data(), v->size() * sizeof(int));
    }
    HX_CATCH
}

int atlas_amd__HaloExchange__execute_strided_int(atlas_amd_HaloExchange* h, int field[], const int var_strides[],
                                                 const int var_shape[], int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_INT, field, var_strides, var_shape, var_rank, false);
}
int atlas_amd__HaloExchange__execute_strided_long(atlas_amd_HaloExchange* h, long field[], const int var_strides[],
                                                  const int var_shape[], int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_LONG, field, var_strides, var_shape, var_rank, false);
}
int atlas_amd__HaloExchange__execute_strided_float(atlas_amd_HaloExchange* h, float field[], const int var_strides[],
                                                   const int var_shape[], int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_FLOAT, field, var_strides, var_shape, var_rank, false);
}
int atlas_amd__HaloExchange__execute_strided_double(atlas_amd_HaloExchange* h, double field[],
                                                    const int var_strides[], const int var_shape[], int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_DOUBLE, field, var_strides, var_shape, var_rank, false);
}
int atlas_amd__HaloExchange__execute_adjoint_strided_int(atlas_amd_HaloExchange* h, int field[],
                                                         const int var_strides[], const int var_shape[],
                                                         int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_INT, field, var_strides, var_shape, var_rank, true);
}
int atlas_amd__HaloExchange__execute_adjoint_strided_long(atlas_amd_HaloExchange* h, long field[],
                                                          const int var_strides[], const int var_shape[],
                                                          int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_LONG, field, var_strides, var_shape, var_rank, true);
}
int atlas_amd__HaloExchange__execute_adjoint_strided_float(atlas_amd_HaloExchange* h, float field[],
                                                           const int var_strides[], const int var_shape[],
                                                           int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_FLOAT, field, var_strides, var_shape, var_rank, true);
}
int atlas_amd__HaloExchange__execute_adjoint_strided_double(atlas_amd_HaloExchange* h, double field[],
                                                            const int var_strides[], const int var_shape[],
                                                            int var_rank) {
    return strided(h, atlas_amd::parallel::HALO_DOUBLE, field, var_strides, var_shape, var_rank, true);
}

// atlas__HaloExchange__execute[_adjoint]_<T>(This, field, var_rank): declared, never defined in the reference
// (HaloExchange.h:441-443,453-455); one value per node is the only layout that needs no shape
static int unstrided(atlas_amd_HaloExchange* h, int dtype, void* field, int var_rank, bool adjoint) {
    if (var_rank != 0) {
        atlas_amd::set_last_error("HaloExchange::execute_<T>(field, var_rank): only var_rank == 0 is defined without a shape; "
                                  "use execute_strided_<T>");
        return 1;
    }
    return strided(h, dtype, field, nullptr, nullptr, 0, adjoint);
}
int atlas_amd__HaloExchange__execute_int(atlas_amd_HaloExchange* h, int field[], int var_rank) {
    return unstrided(h, atlas_amd::parallel::HALO_INT, field, var_rank, false);
}
int atlas_amd__HaloExchange__execute_float(atlas_amd_HaloExchange* h, float field[], int var_rank) {
    return unstrided(h, atlas_amd::parallel::HALO_FLOAT, field, var_rank, false);
}
int atlas_amd__HaloExchange__execute_double(atlas_amd_HaloExchange* h, double field[], int var_rank) {
    return unstrided(h, atlas_amd::parallel::HALO_DOUBLE, field, var_rank, false);
}
int atlas_amd__HaloExchange__execute_adjoint_int(atlas_amd_HaloExchange* h, int field[], int var_rank) {
    return unstrided(h, atlas_amd::parallel::HALO_INT, field, var_rank, true);
}
int atlas_amd__HaloExchange__execute_adjoint_float(atlas_amd_HaloExchange* h, float field[], int var_rank) {
    return unstrided(h, atlas_amd::parallel::HALO_FLOAT, field, var_rank, true);
}
int atlas_amd__HaloExchange__execute_adjoint_double(atlas_amd_HaloExchange* h, double field[], int var_rank) {
    return unstrided(h, atlas_amd::parallel::HALO_DOUBLE, field, var_rank, true);
}

// general field description: rank, shape[], strides[] (elements), parallel dimension -- HaloExchange::execute<T,RANK,
// ParallelDim> (HaloExchange.h:151).  op: 0 execute (1 process), 1 execute_adjoint (1 process), 2 pack, 3 unpack,
// 4 pack_adjoint, 5 unpack_adjoint (+=), 6 zero_halos.  `on_device` != 0: field / buffer are device pointers and the
// call is asynchronous on the object's stream; on_device == 0 is supported for op 0 / 1 only (host staging).
int atlas_amd__HaloExchange__field_op(atlas_amd_HaloExchange* h, int op, int dtype, void* field, int rank,
                                      const int shape[], const long long strides[], int parallel_dim, void* buffer,
                                      int on_device) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::field_op: null handle");
    }
    if (!on_device) {
        if (op != 0 && op != 1) {
            throw std::invalid_argument("host pointers are only supported for execute / execute_adjoint");
        }
        h->impl.execute_host(dtype, field, rank, shape, strides, parallel_dim, op == 1);
        return 0;
    }
    const HaloFieldDesc d = h->impl.describe(rank, shape, strides, parallel_dim);
    switch (op) {
        case 0: h->impl.execute_device(dtype, field, d); break;
        case 1: h->impl.execute_adjoint_device(dtype, field, d); break;
        case 2: h->impl.pack_device(dtype, field, d, buffer); break;
        case 3: h->impl.unpack_device(dtype, field, d, buffer); break;
        case 4: h->impl.pack_adjoint_device(dtype, field, d, buffer); break;
        case 5: h->impl.unpack_adjoint_device(dtype, field, d, buffer); break;
        case 6: h->impl.zero_halos_device(dtype, field, d); break;
        default: throw std::invalid_argument("HaloExchange__field_op: unknown op");
    }
    HX_CATCH
}
void* atlas_amd__HaloExchange__stream(atlas_amd_HaloExchange* h) {
    if (!h) {
        atlas_amd::set_last_error("HaloExchange::stream: null handle");
        return nullptr;
    }
    return (void*)h->impl.stream();
}
int atlas_amd__HaloExchange__set_stream(atlas_amd_HaloExchange* h, void* s) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::set_stream: null handle");
    }
    h->impl.set_stream((hipStream_t)s);
    HX_CATCH
}
int atlas_amd__HaloExchange__synchronize(atlas_amd_HaloExchange* h) {
    HX_TRY
    if (!h) {
        throw std::invalid_argument("HaloExchange::synchronize: null handle");
    }
    h->impl.synchronize();
    HX_CATCH
}

}  // extern "C"
