// Halo pack / unpack on gfx950: gather / scatter between a strided field and contiguous exchange buffers.
//
// Reference being replaced: host pack_index/unpack_index (src/atlas/parallel/detail/pack_index.h:18-100,
// Packer.cc:19-37), the device pack_kernel/unpack_kernel (detail/DevicePacker.hic:51-112, block (32,4), two
// device-wide synchronisations per call), and the host-only adjoint / zero variants
// (detail/adjoint_unpack_index.h, zero_index.h; Packer.h:96-120 "device = NOTIMPLEMENTED").
// Buffer order: for node in map: all non-parallel indices, row-major  (SURVEY 8 "Halo buffers").
//
// Mapping: one wavefront per node when the per-node payload is >= 32 elements (lanes run over the payload, so both
// the buffer side and -- for the usual unit-stride last dimension -- the field side are coalesced), otherwise a flat
// element mapping.  Stream-ordered; no device-wide synchronisation.
//
// halo_compact_kernel builds the ascending list of ghost nodes (part != me || remote_idx != base+idx) on the device
// with wavefront-ballot compaction, for callers whose partition / remote_index fields already live in HBM.
#include <hip/hip_runtime.h>

#include "halo_device.h"

namespace atlas_amd {
namespace parallel {

__device__ __forceinline__ long long field_offset(const HaloFieldDesc& d, int node, int v) {
    long long off = (long long)node * d.node_stride;
    // decode v (row-major over the non-parallel extents)
    if (d.next == 1) {
        off += (long long)v * d.str[0];
    }
    else if (d.next == 2) {
        const int i0 = v / d.ext[1];
        off += (long long)i0 * d.str[0] + (long long)(v - i0 * d.ext[1]) * d.str[1];
    }
    else if (d.next == 3) {
        const int e12 = d.ext[1] * d.ext[2];
        const int i0  = v / e12;
        const int r   = v - i0 * e12;
        const int i1  = r / d.ext[2];
        off += (long long)i0 * d.str[0] + (long long)i1 * d.str[1] + (long long)(r - i1 * d.ext[2]) * d.str[2];
    }
    return off;
}

// MODE 0: buf = field (pack)   1: field = buf (unpack)   2: field = 0 (zero halos)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) halo_copy_kernel(T* __restrict__ field, T* __restrict__ buf,
                                                        const int* __restrict__ map, int cnt, HaloFieldDesc d) {
    const int vs = d.var_size;
    if (vs >= 32) {
        // one wavefront per node
        const int lane  = threadIdx.x & 63;
        const int wave  = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const int nwave = (gridDim.x * blockDim.x) >> 6;
        for (int i = wave; i < cnt; i += nwave) {
            const int node = map[i];
            for (int v = lane; v < vs; v += 64) {
                const long long fo = field_offset(d, node, v);
                const long long bo = (long long)i * vs + v;
                if (MODE == 0) buf[bo] = field[fo];
                else if (MODE == 1) field[fo] = buf[bo];
                else field[fo] = T(0);
            }
        }
    }
    else {
        const long long total = (long long)cnt * vs;
        for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
             e += (long long)gridDim.x * blockDim.x) {
            const int i        = (int)(e / vs);
            const int v        = (int)(e - (long long)i * vs);
            const long long fo = field_offset(d, map[i], v);
            if (MODE == 0) buf[e] = field[fo];
            else if (MODE == 1) field[fo] = buf[e];
            else field[fo] = T(0);
        }
    }
}

// adjoint unpack: field[node] += sum over the node's buffer positions, in ascending buffer order (deterministic,
// equals the reference's sequential loop).  One thread per (unique node, payload element).
template <typename T>
__global__ void __launch_bounds__(256) halo_adjoint_add_kernel(T* __restrict__ field, const T* __restrict__ buf,
                                                               const int* __restrict__ nodes,
                                                               const int* __restrict__ start,
                                                               const int* __restrict__ items, int nnodes,
                                                               HaloFieldDesc d) {
    const int vs          = d.var_size;
    const long long total = (long long)nnodes * vs;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int u        = (int)(e / vs);
        const int v        = (int)(e - (long long)u * vs);
        const long long fo = field_offset(d, nodes[u], v);
        T acc              = field[fo];
        for (int k = start[u]; k < start[u + 1]; ++k) {
            acc += buf[(long long)items[k] * vs + v];
        }
        field[fo] = acc;
    }
}

// ---- ghost list by wavefront-ballot compaction --------------------------------------------------------------
// pass 1: per-block ghost counts; pass 2 (after an exclusive scan of the counts on the host or by
// halo_scan_kernel): each wavefront ballots its ghost flags, lane l writes its index at
// block_offset + wave_offset + popcount(ballot & lanes_below(l)).
__global__ void __launch_bounds__(256) halo_ghost_count_kernel(const int* __restrict__ part,
                                                               const int* __restrict__ ridx, int me, int base,
                                                               int halo_begin, int parsize,
                                                               int* __restrict__ block_counts) {
    __shared__ int wave_cnt[4];
    const int idx    = halo_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const bool ghost = idx < parsize && (part[idx] != me || ridx[idx] != base + idx);
    const unsigned long long b = __ballot(ghost);
    if ((threadIdx.x & 63) == 0) {
        wave_cnt[threadIdx.x >> 6] = __popcll(b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

__global__ void __launch_bounds__(256) halo_ghost_compact_kernel(const int* __restrict__ part,
                                                                 const int* __restrict__ ridx, int me, int base,
                                                                 int halo_begin, int parsize,
                                                                 const int* __restrict__ block_offsets,
                                                                 int* __restrict__ ghosts) {
    __shared__ int wave_cnt[4];
    const int lane   = threadIdx.x & 63;
    const int wave   = threadIdx.x >> 6;
    const int idx    = halo_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const bool ghost = idx < parsize && (part[idx] != me || ridx[idx] != base + idx);
    const unsigned long long b = __ballot(ghost);
    if (lane == 0) {
        wave_cnt[wave] = __popcll(b);
    }
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) {
        woff += wave_cnt[w];
    }
    if (ghost) {
        const unsigned long long below = lane == 0 ? 0ull : (b & (~0ull >> (64 - lane)));
        ghosts[block_offsets[blockIdx.x] + woff + __popcll(below)] = idx;
    }
}

template <typename T>
static hipError_t launch_copy_t(int mode, void* field, void* buf, const int* map, int cnt, const HaloFieldDesc& d,
                                hipStream_t s) {
    if (cnt <= 0) {
        return hipSuccess;
    }
    long long work = d.var_size >= 32 ? (long long)cnt * 64 : (long long)cnt * d.var_size;
    int blocks     = (int)std::min<long long>((work + 255) / 256, 256 * 8);
    T* f           = (T*)field;
    T* b           = (T*)buf;
    switch (mode) {
        case 0: hipLaunchKernelGGL((halo_copy_kernel<T, 0>), dim3(blocks), dim3(256), 0, s, f, b, map, cnt, d); break;
        case 1: hipLaunchKernelGGL((halo_copy_kernel<T, 1>), dim3(blocks), dim3(256), 0, s, f, b, map, cnt, d); break;
        case 2: hipLaunchKernelGGL((halo_copy_kernel<T, 2>), dim3(blocks), dim3(256), 0, s, f, b, map, cnt, d); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_halo_copy(int mode, int dtype, void* field, void* buf, const int* map, int cnt,
                            const HaloFieldDesc& d, hipStream_t s) {
    // pack / unpack / zero move bits: any 4-byte type as uint32, any 8-byte type as uint64
    switch (dtype) {
        case HALO_INT:
        case HALO_FLOAT: return launch_copy_t<unsigned int>(mode, field, buf, map, cnt, d, s);
        case HALO_LONG:
        case HALO_DOUBLE: return launch_copy_t<unsigned long long>(mode, field, buf, map, cnt, d, s);
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_adj_t(void* field, const void* buf, const int* nodes, const int* start, const int* items,
                               int nnodes, const HaloFieldDesc& d, hipStream_t s) {
    if (nnodes <= 0) {
        return hipSuccess;
    }
    int blocks = (int)std::min<long long>(((long long)nnodes * d.var_size + 255) / 256, 256 * 8);
    hipLaunchKernelGGL((halo_adjoint_add_kernel<T>), dim3(blocks), dim3(256), 0, s, (T*)field, (const T*)buf, nodes,
                       start, items, nnodes, d);
    return hipGetLastError();
}

hipError_t launch_halo_adjoint_add(int dtype, void* field, const void* buf, const int* nodes, const int* start,
                                   const int* items, int nnodes, const HaloFieldDesc& d, hipStream_t s) {
    switch (dtype) {
        case HALO_INT: return launch_adj_t<int>(field, buf, nodes, start, items, nnodes, d, s);
        case HALO_LONG: return launch_adj_t<long long>(field, buf, nodes, start, items, nnodes, d, s);
        case HALO_FLOAT: return launch_adj_t<float>(field, buf, nodes, start, items, nnodes, d, s);
        case HALO_DOUBLE: return launch_adj_t<double>(field, buf, nodes, start, items, nnodes, d, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_ghost_count(const int* part, const int* ridx, int me, int base, int halo_begin, int parsize,
                              int* block_counts, int nblocks, hipStream_t s) {
    if (nblocks > 0) {
        hipLaunchKernelGGL(halo_ghost_count_kernel, dim3(nblocks), dim3(256), 0, s, part, ridx, me, base, halo_begin,
                           parsize, block_counts);
    }
    return hipGetLastError();
}
hipError_t launch_ghost_compact(const int* part, const int* ridx, int me, int base, int halo_begin, int parsize,
                                const int* block_offsets, int* ghosts, int nblocks, hipStream_t s) {
    if (nblocks > 0) {
        hipLaunchKernelGGL(halo_ghost_compact_kernel, dim3(nblocks), dim3(256), 0, s, part, ridx, me, base,
                           halo_begin, parsize, block_offsets, ghosts);
    }
    return hipGetLastError();
}

}  // namespace parallel
}  // namespace atlas_amd
