// Fourier stage of the fp32 variant, two fields per job [r4] (fft_pair.h): the direct rows (regular grids -- BASELINE config C5,
// TL1279 -> F1280 -- and the smooth rows of reduced grids).  The kernel is fft_rows_dct_kernel's main branch (fft_kernel.hip) on
// the pair type: phase 0 from an LDS staging area filled by 16-byte LDS-DMA requests, one butterfly per worker and stage, twiddles
// requested before the gather is waited for, the last stage fused with the store.
//
// Reference being replaced: TransLocal::invtrans_fourier_regular / _reduced (src/atlas/trans/local/TransLocal.cc:1101-1196), one
// c2r FFT per (latitude, field).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

#include "device_structs.h"
#include "dyn_lds.h"
#include "fft_device.h"
#include "fft_pair.h"

namespace atlas_amd {
namespace trans {

// Workgroup -> (row, first field of the pair): fft_block_to_job with the pair as the unit -- eight consecutive pairs of a row (the
// sixteen fields of one 128-byte line of the fp32 intermediate) go to one XCD, back to back.
__device__ __forceinline__ bool fft_block_to_pair_job(const FourierParams& p, int b, int& row, int& f) {
    const int x     = b & 7;
    const int q     = b >> 3;
    const int j     = q & 7;
    const int u     = (q >> 3) * 8 + x;
    const int npair = (p.f_end - p.f_begin + 1) >> 1;
    const int ngr   = (npair + 7) >> 3;
    const int ri    = u / ngr;
    const int fg    = u - ri * ngr;
    if (ri >= p.nrows) {
        return false;
    }
    const int pi = fg * 8 + j;
    if (pi >= npair) {
        return false;
    }
    f   = p.f_begin + 2 * pi;
    row = p.rows[ri];
    return true;
}

// Gather of the kept modes of a field pair: the LDS-DMA form of gather_modes_to_lds (fft_device.h) on the float intermediate --
// 16 bytes per lane = (re, im) of field f and of field f + 1 (f even: the request is 16-byte aligned, the record pitch is a multiple
// of 16 floats).
__device__ __forceinline__ void gather_pair_modes_to_lds(const FourierParams& p, long long lat_local, int f, int mmax, fft::cplxp* raw,
                                                         int tid, int nt) {
    const ModeReaderT<1> rd{p, lat_local, 2 * f};
    typedef const __attribute__((address_space(1))) float* gfloat_ptr;
    for (int m0 = 0; m0 <= mmax; m0 += nt) {
        const int m = m0 + tid;
        if (m <= mmax) {
            long long o;
            gfloat_ptr src  = (gfloat_ptr)rd.locate(m, o) + o;
            fft::cplxp* dst = raw + m0 + (tid & ~63);   // wave-uniform
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(src),
                                             reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                 static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst))),
                                             16, 0, 0);
        }
    }
}

// y[2k], y[2k+1] of ONE field of the pair (a row starts on an 8-byte boundary or it does not: uniform per job)
__device__ __forceinline__ void store_field_pair(float* y, bool aligned, int k, float re, float im) {
    if (aligned) {
        typedef float f2_t __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f2_t{re, im}, reinterpret_cast<f2_t*>(y + 2 * k));
    }
    else {
        __builtin_nontemporal_store(re, y + 2 * k);
        __builtin_nontemporal_store(im, y + 2 * k + 1);
    }
}

template <class S>
__global__ void __launch_bounds__(FFT_MAX_NTHR, (dct_waves_per_simd<S, false>())) fft_rows_dct_pair_kernel(FourierParams p) {
    using C  = fft::cplxp;
    using TC = fft::cplxf;   // tables: one float complex for both lanes
    using fft::both;
    extern __shared__ double lds_raw[];
    C* work = reinterpret_cast<C*>(lds_raw);
    int row, f;
    if (!fft_block_to_pair_job(p, blockIdx.x, row, f)) {
        return;
    }
    const fft::FftRowPlan* pl = p.plans + p.row_plan[row];
    const long long goff      = (long long)f * p.npts + (p.rowoff[row] - p.rowoff[p.lat0]);
    const int tid             = threadIdx.x;
    const int nt              = blockDim.x;
    const bool has_b          = f + 1 < p.f_end;
    const float cli           = (float)p.coslatinv[row];
    const fft::f32x2 scale(f < p.scale_uv_fields ? cli : 1.0f, f + 1 < p.scale_uv_fields ? cli : 1.0f);
    const int h               = pl->h;
    const int mmax0           = p.row_mmax[row];
    const int mmax            = mmax0 < h ? mmax0 : h;
    const TC* tw              = p.table_f32 + pl->off_tw;
    const TC* pre             = p.table_f32 + pl->off_pre;
    float* ya                 = reinterpret_cast<float*>(p.gp) + goff;
    float* yb                 = ya + p.npts;
    const bool al_a           = (goff & 1) == 0;
    const bool al_b           = ((goff + p.npts) & 1) == 0;

    using SR          = fft::CtShapeRev<S>;
    constexpr int RL  = SR::radix(SR::NS - 1);
    constexpr int nbl = S::M / RL;
    static_assert(dct_workers<S>() <= FFT_MAX_NTHR, "one butterfly per worker and stage");   // (launch_dct_pair: dct_pair_shape)
    gather_pair_modes_to_lds(p, (long long)(row - p.lat0), f, mmax, work, tid, nt);
    const bool act = tid < nbl;
    const int bp   = act ? tid : 0;
    // c2r factors and the stage twiddles of this worker's butterflies, as table values (8 bytes each), before the gather is waited for
    // (all RL factors where 8 bytes each fit beside the RL inputs, else in batches inside phase 0)
    constexpr bool PRELOAD = RL <= 16;
    constexpr int NB = PRELOAD ? RL : (RL % 4 == 0 ? 4 : (RL % 5 == 0 ? 5 : (RL % 3 == 0 ? 3 : (RL % 2 == 0 ? 2 : 1))));
    TC P[PRELOAD ? RL : NB];
    if constexpr (PRELOAD) {
#pragma unroll
        for (int q = 0; q < RL; ++q) P[q] = pre[bp + q * nbl];
    }
    constexpr int NMIDS = SR::NS > 2 ? SR::NS - 2 : 0;
    TC wmid[NMIDS > 0 ? NMIDS : 1];
    int mbase[NMIDS > 0 ? NMIDS : 1];
    dct_for_each_mid<SR, 1>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int R = SR::radix(I);
        constexpr int L = SR::L(I);
        const int b     = tid < SR::M / R ? tid : 0;
        int blk, j;
        fft::split_index(b, L / R, SR::lsh(I), blk, j);
        mbase[I - 1] = blk * L + j;
        wmid[I - 1]  = tw[j * (SR::M / L)];
    });
    constexpr int R0  = SR::radix(0);
    constexpr int Ls0 = SR::M / R0;
    const TC wlast    = tw[tid < Ls0 ? tid : 0];
    AA_SCHED_FENCE();
    __syncthreads();
    C x[RL];
#pragma unroll
    for (int q0 = 0; q0 < RL; q0 += NB) {
        if constexpr (!PRELOAD) {
#pragma unroll
            for (int i = 0; i < NB; ++i) P[i] = pre[bp + (q0 + i) * nbl];
            AA_SCHED_FENCE();
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = bp + (q0 + i) * nbl;
            const C a   = fft::pair_raw_mode(work, mmax, k, h);
            const C c   = fft::cconj(fft::pair_raw_mode(work, mmax, h - k, h));
            x[q0 + i]   = fft::c2r_pre(a, c, both(P[PRELOAD ? q0 + i : i]));
        }
    }
    fft::bfly<RL>(x, +1);
    lds_barrier();   // everybody has read the staging area
    if (act) {
        const int b = fft::dct_first_butterfly<S>(bp);
#pragma unroll
        for (int q = 0; q < RL; ++q) work[fft::PAD(b * RL + q)] = x[q];
    }
    __syncthreads();
    // ---- DIT stages NS-2 .. 1
    dct_for_each_mid_down<SR, SR::NS - 2>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int R = SR::radix(I);
        if (tid < SR::M / R) {
            fft::dit_butterfly_w<R>(work, mbase[I - 1], SR::L(I) / R, both(wmid[I - 1]), +1);
        }
        __syncthreads();
    });
    // ---- DIT stage 0 + store: lane x to field f, lane y to field f + 1
    if (tid < Ls0) {
        C y[R0];
#pragma unroll
        for (int q = 0; q < R0; ++q) y[q] = work[fft::PAD(tid + q * Ls0)];
        fft::twiddle_apply<R0>(y, both(wlast));
        fft::bfly<R0>(y, +1);
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            const fft::f32x2 re = y[q].re * scale, im = y[q].im * scale;
            store_field_pair(ya, al_a, tid + q * Ls0, re.v.x, im.v.x);
            if (has_b) {
                store_field_pair(yb, al_b, tid + q * Ls0, re.v.y, im.v.y);
            }
        }
    }
}

// shapes that take the pair form: one butterfly per worker and stage, and a first butterfly narrower than 20 points (24 / 20 pairs next
// to their c2r factors spill 10 - 50 registers: M = 320, 384, 6144 keep the one-field form)
template <class S>
constexpr bool dct_pair_shape() {
    using SR = fft::CtShapeRev<S>;
    return dct_workers<S>() <= FFT_MAX_NTHR && SR::radix(SR::NS - 1) < 20;
}

template <class S>
static hipError_t launch_dct_pair(const FourierParams& p, int lds_bytes, hipStream_t stream) {
    if constexpr (dct_pair_shape<S>()) {
        lds_bytes += 256;   // staging of phase 0: modes 0..mmax, mmax <= M (one element more than the work array)
        if (hipError_t e = ensure_dynamic_lds<&fft_rows_dct_pair_kernel<S>>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
            return e;
        }
        const int npair     = (p.f_end - p.f_begin + 1) / 2;
        const unsigned nblk = fft_job_blocks(p.nrows, npair, 3);
        hipLaunchKernelGGL((fft_rows_dct_pair_kernel<S>), dim3(nblk), dim3(dct_workers<S>()), lds_bytes, stream, p);
        return hipGetLastError();
    }
    else {
        return hipErrorNotSupported;
    }
}

// Can this launch take the two-field form?  (fft_kernel.hip: launch_fourier_dct asks; ATLAS_AMD_FFT_F32_PAIRS=0 switches it off.)
// The pair's 16 bytes must be one aligned element of a single-piece intermediate: first field even, no m-sharded pieces, no packed runs.
bool fourier_pairs_usable(const FourierParams& p, int ctf, int ctk) {
    const char* e = std::getenv("ATLAS_AMD_FFT_F32_PAIRS");   // read per launch: the tests switch it between calls
    const bool on = !(e && atoi(e) == 0);
    if (!on || !p.f32 || !p.table_f32 || (p.f_begin & 1) || p.nparts > 1 || p.packed_cols || (p.RP & 3)) {
        return false;
    }
    bool fits = false;
    AA_CT_DISPATCH(ctf, ctk, fits = dct_pair_shape<S>())
    return fits;
}

hipError_t launch_fourier_dct_pairs(const FourierParams& p, int ctf, int ctk, int lds_bytes, hipStream_t stream) {
    AA_CT_DISPATCH(ctf, ctk, return launch_dct_pair<S>(p, lds_bytes, stream))
    return hipErrorInvalidValue;
}

}  // namespace trans
}  // namespace atlas_amd
