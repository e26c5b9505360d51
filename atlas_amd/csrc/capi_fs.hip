// extern "C" layer for functionspace::StructuredColumns (halo index construction + halo-exchange dispatch with the
// pole sign fix-up for vector fields).  Replaces what atlas__FunctionSpace__halo_exchange_field does for
// StructuredColumns (src/atlas/functionspace/detail/FunctionSpaceInterface.h:40-43 ->
// StructuredColumns.cc:811-911) and the accessors of src/atlas/functionspace/detail/StructuredColumnsInterface.h.
#include <hip/hip_runtime.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/atlas_amd.h"
#include "capi_types.h"
#include "structured_columns.h"

namespace atlas_amd {
void set_last_error(const std::string& s);
}
using atlas_amd::functionspace::StructuredColumns;
using atlas_amd::functionspace::StructuredColumnsConfig;

struct atlas_amd_StructuredColumns {
    StructuredColumns impl;
    int* d_pole_nodes = nullptr;
    int npole         = 0;
    atlas_amd_StructuredColumns(const atlas_amd::grid::StructuredGrid& g, const StructuredColumnsConfig& c): impl(g, c) {}
    ~atlas_amd_StructuredColumns() {
        if (d_pole_nodes) {
            (void)hipFree(d_pole_nodes);
        }
    }
};

#define FS_TRY try {
#define FS_CATCH                                    \
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

// FixupHaloForVectors (StructuredColumns.cc:732-808): negate components XX, YY in the halo rows beyond the poles.
// field(n, [k,] var): stride_n, stride_k, stride_v in elements.
template <typename T>
__global__ void __launch_bounds__(256) fixup_vector_kernel(T* field, const int* nodes, int nnodes, int levels,
                                                           long long stride_n, long long stride_k,
                                                           long long stride_v) {
    const long long total = (long long)nnodes * levels * 2;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int v        = (int)(e % 2);
        const long long r  = e / 2;
        const int k        = (int)(r % levels);
        const int n        = nodes[r / levels];
        const long long o  = (long long)n * stride_n + (long long)k * stride_k + (long long)v * stride_v;
        field[o]           = -field[o];
    }
}

extern "C" {

atlas_amd_StructuredColumns* atlas_amd__StructuredColumns__new(const atlas_amd_Grid* grid, int halo,
                                                               int periodic_points, int nparts, int part,
                                                               int blocksize) {
    try {
        if (!grid) {
            throw std::invalid_argument("grid is NULL");
        }
        StructuredColumnsConfig c;
        c.halo            = halo;
        c.periodic_points = periodic_points != 0;
        c.nparts          = nparts;
        c.part            = part;
        c.blocksize       = blocksize;
        return new atlas_amd_StructuredColumns(grid->g, c);
    }
    catch (const std::exception& e) {
        atlas_amd::set_last_error(e.what());
        return nullptr;
    }
}
atlas_amd_StructuredColumns* atlas_amd__StructuredColumns__new_distribution(const atlas_amd_Grid* grid, int halo,
                                                                            int periodic_points, int nparts, int part,
                                                                            const int* partition, long long npts) {
    try {
        if (!grid || !partition) {
            throw std::invalid_argument("grid / partition is NULL");
        }
        StructuredColumnsConfig c;
        c.halo            = halo;
        c.periodic_points = periodic_points != 0;
        c.nparts          = nparts;
        c.part            = part;
        c.blocksize       = 1;
        c.distribution.assign(partition, partition + npts);
        return new atlas_amd_StructuredColumns(grid->g, c);
    }
    catch (const std::exception& e) {
        atlas_amd::set_last_error(e.what());
        return nullptr;
    }
}
void atlas_amd__StructuredColumns__delete(atlas_amd_StructuredColumns* fs) {
    delete fs;
}
int atlas_amd__StructuredColumns__size_owned(const atlas_amd_StructuredColumns* fs) {
    if (!fs) {
        atlas_amd::set_last_error("StructuredColumns::size_owned: null handle");
        return -1;
    }
    return fs->impl.size_owned();
}
int atlas_amd__StructuredColumns__size_halo(const atlas_amd_StructuredColumns* fs) {
    if (!fs) {
        atlas_amd::set_last_error("StructuredColumns::size_halo: null handle");
        return -1;
    }
    return fs->impl.size_halo();
}
int atlas_amd__StructuredColumns__bounds(const atlas_amd_StructuredColumns* fs, int out[4]) {
    if (!fs) {
        atlas_amd::set_last_error("StructuredColumns::bounds: null handle");
        return -1;
    }
    out[0] = fs->impl.j_begin();
    out[1] = fs->impl.j_end();
    out[2] = fs->impl.j_begin_halo();
    out[3] = fs->impl.j_end_halo();
    return 0;
}
int atlas_amd__StructuredColumns__row_bounds(const atlas_amd_StructuredColumns* fs, int j, int out[4]) {
    FS_TRY
    if (!fs) {
        throw std::invalid_argument("StructuredColumns::row_bounds: null handle");
    }
    const auto& f = fs->impl;
    if (j < f.j_begin_halo() || j >= f.j_end_halo()) {
        throw std::out_of_range("row_bounds: j outside the halo");
    }
    const bool owned_row = j >= f.j_begin() && j < f.j_end();
    out[0]               = owned_row ? f.i_begin(j) : 0;
    out[1]               = owned_row ? f.i_end(j) : 0;
    out[2]               = f.i_begin_halo(j);
    out[3]               = f.i_end_halo(j);
    FS_CATCH
}
int atlas_amd__StructuredColumns__index(const atlas_amd_StructuredColumns* fs, int i, int j, int* out) {
    FS_TRY
    if (!fs) {
        throw std::invalid_argument("StructuredColumns::index: null handle");
    }
    *out = fs->impl.index(i, j);
    FS_CATCH
}
int atlas_amd__StructuredColumns__get_int(const atlas_amd_StructuredColumns* fs, const char* what, int out[]) {
    FS_TRY
    if (!fs) {
        throw std::invalid_argument("StructuredColumns::get_int: null handle");
    }
    const std::string w       = what ? what : "";
    const std::vector<int>* v = nullptr;
    std::vector<int> tmp;
    if (w == "partition") v = &fs->impl.partition();
    else if (w == "ghost") v = &fs->impl.ghost();
    else if (w == "index_i") v = &fs->impl.index_i();
    else if (w == "index_j") v = &fs->impl.index_j();
    else if (w == "remote_idx") v = &fs->impl.remote_index();
    else if (w == "pole_row_nodes") {
        tmp = fs->impl.pole_row_nodes();
        v   = &tmp;
    }
    else throw std::invalid_argument("StructuredColumns__get_int: unknown field '" + w + "'");
    if (!v->empty()) {
        std::memcpy(out, v->data(), v->size() * sizeof(int));
    }
    FS_CATCH
}
int atlas_amd__StructuredColumns__nb_pole_row_nodes(const atlas_amd_StructuredColumns* fs) {
    if (!fs) {
        atlas_amd::set_last_error("StructuredColumns::nb_pole_row_nodes: null handle");
        return -1;
    }
    return (int)fs->impl.pole_row_nodes().size();
}
int atlas_amd__StructuredColumns__global_index(const atlas_amd_StructuredColumns* fs, int64_t out[]) {
    if (!fs) {
        atlas_amd::set_last_error("StructuredColumns::global_index: null handle");
        return -1;
    }
    std::memcpy(out, fs->impl.global_index().data(), fs->impl.global_index().size() * sizeof(int64_t));
    return 0;
}
int atlas_amd__StructuredColumns__xy(const atlas_amd_StructuredColumns* fs, double out[]) {
    if (!fs) {
        atlas_amd::set_last_error("StructuredColumns::xy: null handle");
        return -1;
    }
    std::memcpy(out, fs->impl.xy().data(), fs->impl.xy().size() * sizeof(double));
    return 0;
}
// HaloExchange::setup(partition, remote_index, REMOTE_IDX_BASE, sizeHalo, sizeOwned)  (StructuredColumns.cc:145-148):
// local phase for a multi-partition function space, complete setup for one partition
int atlas_amd__StructuredColumns__setup_halo_exchange(const atlas_amd_StructuredColumns* fs,
                                                      atlas_amd_HaloExchange* hx, int nparts, int part) {
    FS_TRY
    if (!fs || !hx) {
        throw std::invalid_argument("StructuredColumns::setup_halo_exchange: null handle");
    }
    const auto& f = fs->impl;
    if (nparts == 1) {
        hx->impl.setup(f.partition().data(), f.remote_index().data(), 0, f.size_halo(), f.size_owned());
    }
    else {
        hx->impl.setup_begin(nparts, part, f.partition().data(), f.remote_index().data(), 0, f.size_halo(),
                             f.size_owned());
    }
    FS_CATCH
}
int atlas_amd__StructuredColumns__fixup_halo_for_vectors(atlas_amd_StructuredColumns* fs, int dtype, void* field_dev,
                                                         int levels, long long stride_n, long long stride_k,
                                                         long long stride_v, void* hip_stream) {
    FS_TRY
    if (!fs) {
        throw std::invalid_argument("StructuredColumns::fixup_halo_for_vectors: null handle");
    }
    if (!fs->d_pole_nodes) {
        std::vector<int> nodes = fs->impl.pole_row_nodes();
        fs->npole              = (int)nodes.size();
        if (hipMalloc((void**)&fs->d_pole_nodes, std::max<size_t>(nodes.size(), 1) * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            throw std::runtime_error("hipMalloc failed (no HIP device?)");
        }
        if (!nodes.empty() &&
            hipMemcpy(fs->d_pole_nodes, nodes.data(), nodes.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            throw std::runtime_error("hipMemcpy failed");
        }
    }
    if (fs->npole == 0) {
        return 0;
    }
    if (levels < 1) {
        levels = 1;
    }
    hipStream_t s      = (hipStream_t)hip_stream;
    const long long n  = (long long)fs->npole * levels * 2;
    const int blocks   = (int)std::min<long long>((n + 255) / 256, 2048);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(fixup_vector_kernel<int>, dim3(blocks), dim3(256), 0, s, (int*)field_dev, fs->d_pole_nodes, fs->npole, levels, stride_n, stride_k, stride_v); break;
        case 1: hipLaunchKernelGGL(fixup_vector_kernel<long long>, dim3(blocks), dim3(256), 0, s, (long long*)field_dev, fs->d_pole_nodes, fs->npole, levels, stride_n, stride_k, stride_v); break;
        case 2: hipLaunchKernelGGL(fixup_vector_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)field_dev, fs->d_pole_nodes, fs->npole, levels, stride_n, stride_k, stride_v); break;
        case 3: hipLaunchKernelGGL(fixup_vector_kernel<double>, dim3(blocks), dim3(256), 0, s, (double*)field_dev, fs->d_pole_nodes, fs->npole, levels, stride_n, stride_k, stride_v); break;
        default: throw std::invalid_argument("datatype not supported");  // StructuredColumns.cc:830
    }
    if (hipGetLastError() != hipSuccess) {
        throw std::runtime_error("fixup_vector_kernel launch failed");
    }
    FS_CATCH
}

}  // extern "C"
