// Device-side helpers shared by the Fourier kernels (fft_kernel.hip, fft_kernel_pairs.hip, fft_kernel_p.hip): block -> job maps, the reader of the
// Fourier intermediate, LDS-only barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "device_structs.h"

namespace atlas_amd {
namespace trans {

using fft::cplx;

constexpr int FFT_MAX_NTHR = 512;  // 2 waves per SIMD -> 256 VGPRs: radix-16 butterflies stay in registers
constexpr int FGROUP   = 8;  // fields whose modes share one 128-byte line of F

// Block -> (row, field).  Eight consecutive fields of one row share every 128-byte line of F, and hardware places
// block b on XCD b % 8, so blocks {b, b+8, ..., b+56} (same XCD, dispatched back to back) are given the same
// (row, field group): the line is then fetched into that XCD's L2 once.  Speed heuristic only.
// the same map for the kernels that read FftRowDesc, in two steps: workgroup -> (XCD x, sequence number idx on that XCD, field
// j of the group), then (x, idx) -> (row index ri of the launch's list, field group fg).
// p.row_affinity: the rows in full sets of 8 stay on ONE XCD each (row 8 k + x on XCD x), so that a row's tables (chirp, c2r
// factors, filter spectrum: 130 - 200 KB) are fetched into one L2 instead of all eight; the last nrows % 8 rows are dealt
// out by field group as before (balance).
__device__ __forceinline__ bool fft_unit_to_job_n(int row_affinity, int nrows, int ngr, int x, int idx, int& ri, int& fg);
__device__ __forceinline__ bool fft_unit_to_job(const FourierParams& p, int ngr, int x, int idx, int& ri, int& fg) {
    return fft_unit_to_job_n(p.row_affinity, p.nrows, ngr, x, idx, ri, fg);
}
// (the same for a sub-list of `nrows` rows: fft_rows_coarse_multi_kernel)
__device__ __forceinline__ bool fft_unit_to_job_n(int row_affinity, int nrows, int ngr, int x, int idx, int& ri, int& fg) {
    if (row_affinity) {
        const int nfull = nrows & ~7;
        const int main_ = (nfull >> 3) * ngr;
        if (idx < main_) {
            const int k = idx / ngr;
            ri          = 8 * k + x;
            fg          = idx - k * ngr;
            return true;
        }
        const int u = (idx - main_) * 8 + x;
        const int k = u / ngr;
        ri          = nfull + k;
        fg          = u - k * ngr;
        return ri < nrows;
    }
    const int u = idx * 8 + x;
    ri          = u / ngr;
    fg          = u - ri * ngr;
    return ri < nrows;
}
__device__ __forceinline__ bool fft_block_to_job_index(const FourierParams& p, int b, int& ri, int& f) {
    const int x   = b & 7;
    const int q   = b >> 3;
    const int j   = q & 7;
    const int ngr = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    int fg;
    if (!fft_unit_to_job(p, ngr, x, q >> 3, ri, fg)) {
        return false;
    }
    f = p.f_begin + fg * FGROUP + j;
    return f < p.f_end;
}
// (the kernels without a row record: direct rows, run-time shaped rows).  In the fp32 variant a 128-byte line of the intermediate
// holds SIXTEEN fields (8 bytes per mode): the group that shares an XCD is then 16 fields wide -- with 8, the two halves of every
// line went to two XCDs and were fetched from HBM twice (regular grids, BASELINE config C5) [r4]
__device__ __forceinline__ int fft_job_group_log2(const FourierParams& p) {
    return p.job_group_log2;
}
__device__ __forceinline__ bool fft_block_to_job(const FourierParams& p, int b, int& row, int& f) {
    const int gl  = fft_job_group_log2(p);
    const int gw  = 1 << gl;
    const int x   = b & 7;
    const int q   = b >> 3;
    const int j   = q & (gw - 1);
    const int u   = (q >> gl) * 8 + x;
    const int ngr = (p.f_end - p.f_begin + gw - 1) >> gl;
    const int ri  = u / ngr;
    const int fg  = u - ri * ngr;
    if (ri >= p.nrows) {
        return false;
    }
    f = p.f_begin + fg * gw + j;
    if (f >= p.f_end) {
        return false;
    }
    row = p.rows[ri];
    return true;
}
// workgroups of a launch of those kernels (host)
inline unsigned fft_job_blocks(int nrows, int nfields, int group_log2) {
    const int gw          = 1 << group_log2;
    const long long ngr   = (nfields + gw - 1) / gw;
    const long long units = (long long)nrows * ngr;
    return (unsigned)((units + 7) / 8 * 8 * gw);
}

// Input modes of one (row, field).  The Fourier intermediate may be split by zonal wavenumber over `nparts`
// producers (multi-GPU m-sharding: wavenumber m belongs to part m % nparts, local index m / nparts):
//   X[m] = *(cplx*)(base[m % nparts] + (lat_local * cnt[m % nparts] + m / nparts) * RP + 2*field)
// nparts == 1 is the single-device layout F[(lat*(T+1) + m)*RP + r].
// Packed form (p.packed_cols != 0, distributed transform [r3]): X[m] = *(cplx*)(base[m % nparts] + rowoff[m % nparts][lat_local]
//   + (m / nparts) * packed_cols + 2*field) -- only the kept wavenumbers of a row and the live columns are stored.
// STORAGE: 0 = double intermediate, 1 = float (fp32 variant), 2 = decided at run time by p.f32 (generic kernel)
template <int STORAGE>
struct ModeReaderT {
    const FourierParams& p;
    long long lat_local;
    int f2;
    // pointers into the Fourier intermediate are GLOBAL pointers, and typed so: one that comes out of the piece table would
    // otherwise be a generic pointer and every mode a flat load (LDS and vector-memory counters both)
    typedef const __attribute__((address_space(1))) double* gdouble_ptr;
    typedef const __attribute__((address_space(1))) long long* glong_ptr;
    // address of mode m (fp64 storage: a double*; fp32 storage: element index in floats, see operator())
    __device__ __forceinline__ gdouble_ptr locate(int m, long long& o) const {
        gdouble_ptr base = (gdouble_ptr)(uintptr_t)p.part_base0;
        glong_ptr ro     = (glong_ptr)(uintptr_t)p.part_rowoff0;
        int cnt          = p.part_cnt0;
        int ml           = m;
        if (p.packed_rowbase) {   // packed runs with the combined row table (the distributed transform): one table read per mode
            int part = 0;
            if (p.nparts > 1) {
                if (p.parts_shift >= 0) {
                    ml   = m >> p.parts_shift;
                    part = m & (p.nparts - 1);
                }
                else {
                    ml   = m / p.nparts;
                    part = m - ml * p.nparts;
                }
            }
            glong_ptr rb = (glong_ptr)(uintptr_t)p.packed_rowbase;
            o = rb[lat_local * p.nparts + part] + (long long)__umul24((unsigned)ml, (unsigned)p.packed_cols) + f2;
            return base;
        }
        if (p.nparts > 1) {   // the piece of wavenumber m: a (cached) table lookup per lane
            if (p.parts_shift >= 0) {
                ml = m >> p.parts_shift;
            }
            else {
                ml = m / p.nparts;
            }
            const int part = m - ml * p.nparts;
            const FourierParts* tb = p.parts;
            uintptr_t b    = (uintptr_t)tb->base[part];
            uintptr_t r    = (uintptr_t)tb->rowoff[part];
            int c          = tb->cnt[part];
            // values, not "a load from one of two addresses": left alone, the compiler turns the choice between a kernel
            // argument and a table entry into a load through a generic pointer (the argument copied to scratch), per mode
            asm volatile("" : "+v"(b), "+v"(r), "+v"(c));
            base = (gdouble_ptr)b;
            ro   = (glong_ptr)r;
            cnt  = c;
        }
#if defined(AA_FFT_ABLATE)
        if (p.abl & 1) ml = 0;
        if (p.abl & 64) ml >>= 1;    // dev probes (results wrong): 2 / 4 / 8 lanes per 128-byte line -- how the gather's cost scales
        if (p.abl & 128) ml >>= 2;   // with the number of distinct lines a request touches
        if (p.abl & 256) ml >>= 3;
#endif
        // wavenumber x record length as a 24-bit product (one full-rate instruction; the host checks (T + 1) * RP < 2^31,
        // trans.hip: fourier_fields): the 64-bit form cost the direct rows more integer multiplies than butterfly arithmetic
        if (p.packed_cols) {   // packed runs of the distributed transform (dist_trans.h): per-row offsets, no pitch padding
            o = ro[lat_local] + (long long)__umul24((unsigned)ml, (unsigned)p.packed_cols) + f2;
            return base;
        }
        o = (lat_local * cnt) * p.RP + f2 + (long long)__umul24((unsigned)ml, (unsigned)p.RP);
        return base;
    }
    __device__ __forceinline__ const char* byte_address(int m) const {   // either storage (STORAGE 0 / 1)
        long long o;
        gdouble_ptr base = locate(m, o);
        return STORAGE == 1 ? (const char*)((const float*)base + o) : (const char*)((const double*)base + o);
    }
    __device__ __forceinline__ const double* address(int m) const {   // fp64 storage only
        long long o;
        gdouble_ptr base = locate(m, o);
        return (const double*)(base + o);
    }
    __device__ __forceinline__ cplx operator()(int m) const {
        long long o;
        gdouble_ptr base = locate(m, o);
        if (STORAGE == 1 || (STORAGE == 2 && p.f32)) {  // fp32 intermediate: same element indexing, float storage
            typedef float f2_t __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(1))) float* gfloat_ptr;
            const f2_t v = *(const __attribute__((address_space(1))) f2_t*)((gfloat_ptr)base + o);
            return cplx{(double)v.x, (double)v.y};
        }
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t v = *(const __attribute__((address_space(1))) d2_t*)(base + o);
        return cplx{v.x, v.y};
    }
};
using ModeReader = ModeReaderT<2>;


// workgroup barrier that orders LDS accesses only: global loads requested before it stay in flight (__syncthreads() would
// drain them: its workgroup-scope fence waits for vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void wave_lds_fence() {
    // producer and consumer lanes are in this wavefront: LDS executes a wavefront's instructions in order, only the
    // compiler must not move accesses across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gather of the kept modes X[0..mmax] of one (row, field) into LDS (raw[m], contiguous).  fp64: LDS-DMA, 16 bytes per
// lane straight from the Fourier intermediate into LDS (no staging registers; destination = wave-uniform base + 16 * lane,
// so lane l of wave w handles m = sweep * nt + 64 w + l; lanes beyond mmax request NOTHING: their LDS slots keep what the caller
// put there -- a caller that reads them (row_ct3's unmasked phase 0) zero-fills them itself).
// fp32 intermediate: through registers (the 8-byte element has no DMA width).
template <bool F32, class C>
__device__ __forceinline__ void gather_modes_to_lds(const FourierParams& p, long long lat_local, int f, int mmax,
                                                    C* raw, int tid, int nt) {
    static_assert(F32 || sizeof(C) == 16, "the LDS-DMA gather moves 16-byte elements");
    const ModeReaderT<(F32 ? 1 : 0)> rd{p, lat_local, 2 * f};
    if (mmax < 0) {
        return;
    }
    if constexpr (F32) {
        // through registers, 8 sweeps of requests in flight before the first LDS store (left as one sweep per iteration the
        // loop is a chain of dependent round trips: request, wait, store, next request)
        constexpr int U = 8;
        for (int m0 = 0; m0 <= mmax; m0 += U * nt) {
            cplx v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + u * nt + tid;
                v[u]        = rd(m <= mmax ? m : mmax);   // (float -> double; back to float for C = cplxf: folded away)
            }
            AA_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + u * nt + tid;
                if (m <= mmax) {
                    raw[m] = C{(typename C::real)v[u].re, (typename C::real)v[u].im};
                }
            }
        }
        return;
    }
    for (int m0 = 0; m0 <= mmax; m0 += nt) {
        const int m  = m0 + tid;
        const int mc = m <= mmax ? m : mmax;
        if constexpr (F32) {
        }
        else {
            if (m <= mmax) {   // lanes past the last mode request nothing (their LDS slots belong to the caller: row_ct3 zeroes them)
                const double* src = rd.address(mc);
                cplx* dst         = raw + m0 + (tid & ~63);   // wave-uniform
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src)),
                    reinterpret_cast<__attribute__((address_space(3))) void*>(
                        static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst))),
                    16, 0, 0);
            }
        }
    }
}

// ---- helpers of the direct rows (fft_kernel.hip: fft_rows_dct_kernel; fft_kernel_pairs.hip: the two-field fp32 form)
// workers of a direct row: one butterfly per worker in every stage (max over the stages of M / radix, whole wavefronts)
template <class S>
constexpr int dct_workers() {
    int w = 64;
    for (int i = 0; i < S::NS; ++i) {
        const int nb = (S::M / S::radix(i) + 63) / 64 * 64;
        w            = nb > w ? nb : w;
    }
    return w;
}

// compile-time loops over the middle DIT stages I = 1 .. NS-2 of a reversed shape (up / down)
template <class SR, int I, class Fn>
__device__ __forceinline__ void dct_for_each_mid(Fn&& fn) {
    if constexpr (I <= SR::NS - 2) {
        fn(std::integral_constant<int, I>{});
        dct_for_each_mid<SR, I + 1>(fn);
    }
}
template <class SR, int I, class Fn>
__device__ __forceinline__ void dct_for_each_mid_down(Fn&& fn) {
    if constexpr (I >= 1) {
        fn(std::integral_constant<int, I>{});
        dct_for_each_mid_down<SR, I - 1>(fn);
    }
}

// wavefronts per SIMD a direct-row kernel is compiled for: 3 (168 registers), 2 where the first butterfly is too wide for that
// (radix >= 15 with 16-byte elements -- such rows are >= 60 KB of LDS: two workgroups per CU anyway -- or >= 20 with 8-byte elements:
// 30 - 310 spilled registers)
#ifndef AA_DCT_F32_WPS
#define AA_DCT_F32_WPS 3   // dev builds: wavefronts per SIMD the fp32-arithmetic direct rows are compiled for (A/B)
#endif
template <class S, bool F32A>
constexpr int dct_waves_per_simd() {
    if (F32A && AA_DCT_F32_WPS != 3) {
        return AA_DCT_F32_WPS;
    }
    return S::radix(0) >= (F32A ? 20 : 15) ? 2 : 3;
}


// the [R0,16,16] rows of the LDS-heavy classes (see fft_kernel.hip: row_ct3)
template <class S>
constexpr bool ct3_fast_path() {
    return S::NS == 3 && S::wave_local_middle() && S::radix(1) == 16 && S::radix(2) == 16 && S::M / S::radix(0) == 256 &&
           S::NT == 256 && S::WPS == 2;
}

}  // namespace trans
}  // namespace atlas_amd
