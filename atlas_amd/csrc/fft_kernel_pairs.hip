// Fourier stage of the fp32 variant, two fields per job [r4] (fft_pair.h).
//   * the direct rows (regular grids -- BASELINE config C5, TL1279 -> F1280 -- and the smooth rows of reduced grids):
//     fft_rows_dct_kernel's main branch (fft_kernel.hip) on the pair type: phase 0 from an LDS staging area filled by 16-byte LDS-DMA
//     requests, one butterfly per worker and stage, twiddles requested before the gather is waited for, last stage fused with the store;
//   * the specialised Bluestein rows (reduced grids): fft_rows_ct_kernel's body on the pair type -- row_ct3 (fft_ct_rows.h) for the
//     [R0,16,16] classes, the phase loop of row_phase_ct (fft_core.h) for the others; same row records, same (float) tables.
//
// Reference being replaced: TransLocal::invtrans_fourier_regular / _reduced (src/atlas/trans/local/TransLocal.cc:1101-1196), one
// c2r FFT per (latitude, field).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

#include "env.h"
#include "device_structs.h"
#include "dyn_lds.h"
#include "fft_device.h"
#include "fft_pair.h"      // before fft_ct_rows.h: the pair overloads of the store and of the staging reads
#include "fft_ct_rows.h"

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

// Gather of the kept modes of a field pair (overload of fft_device.h: gather_modes_to_lds, chosen by the element type): its LDS-DMA form on the float intermediate --
// 16 bytes per lane = (re, im) of field f and of field f + 1 (f even: the request is 16-byte aligned, the record pitch is a multiple
// of 16 floats).
template <bool F32>
__device__ __forceinline__ void gather_modes_to_lds(const FourierParams& p, long long lat_local, int f, int mmax, fft::cplxp* raw,
                                                    int tid, int nt) {
    static_assert(F32, "a field pair is the fp32 variant");
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

// wavefronts per SIMD the pair form of a direct row is compiled for: as the fp64 rows (same element size); dev builds
// -DAA_DCT_PAIR_WPS4: four for the shapes of which four rows fit the LDS of a CU (M <= 2560, no staging slot beyond the row)
template <class S>
constexpr int dct_pair_waves_per_simd() {
#if defined(AA_DCT_PAIR_WPS4)
    if (S::M <= 2560 && dct_workers<S>() <= 256) {
        return 4;
    }
#endif
    return dct_waves_per_simd<S, false>();
}
template <class S>
__global__ void __launch_bounds__((dct_pair_waves_per_simd<S>() == 4 ? 256 : FFT_MAX_NTHR), (dct_pair_waves_per_simd<S>())) fft_rows_dct_pair_kernel(FourierParams p) {
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
    gather_modes_to_lds<true>(p, (long long)(row - p.lat0), f, mmax, work, tid, nt);
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
        if (p.T >= S::M) {
            lds_bytes += 256;   // staging of phase 0: modes 0..mmax, mmax <= M: one element more than the work array if the truncation reaches M
        }
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

// ---- run-time shaped direct rows, two fields per job [r6] --------------------------------------------------------------------------
// The {2,3,5}-smooth half lengths outside the specialised family F 2^K (2^a 25, 2^a 27, 2^a 45, ... : a quarter of the points of the
// classic N grids) run fft_rows_kernel's direct branch -- fp64 arithmetic on float storage, one field per job -- in the fp32 variant
// (fft_kernel.hip); this is the same phase list (load + c2r pre-processing into digit-reversed order | DIT stages from the shape record
// | store) on the pair type: half the jobs, packed fp32 instructions, float tables -- and likewise the ODD branch (odd {3,5}-smooth
// lengths, a complex transform of the row's own length: the many odd row lengths of the classic N grids, 1.35 of the 5.3 ms of kernel
// time of TL1279 -> N1280's Fourier stage in round 5).  The launch's row list holds FFT_DIRECT and FFT_ODD rows only (trans.hip splits
// the run-time shaped classes).
template <int R>
__device__ __forceinline__ void dit_stage_pair(fft::cplxp* d, int M, int L, int lsh, fft::PairTable tw, int t, int nt) {
    fft::dit_stage<R, fft::cplxp, fft::PairTable>(d, M, L, lsh, tw, +1, t, nt);
}
__global__ void __launch_bounds__(FFT_MAX_NTHR) fft_rows_pair_kernel(FourierParams p) {
    using C  = fft::cplxp;
    using TC = fft::cplxf;
    using fft::both;
    extern __shared__ double lds_raw[];
    C* work = reinterpret_cast<C*>(lds_raw);
    int row, f;
    if (!fft_block_to_pair_job(p, blockIdx.x, row, f)) {
        return;
    }
    const fft::FftRowPlan* pl = p.plans + p.row_plan[row];
    const fft::FftShape& sh   = pl->shape;
    const long long goff      = (long long)f * p.npts + (p.rowoff[row] - p.rowoff[p.lat0]);
    const int nx              = (int)(p.rowoff[row + 1] - p.rowoff[row]);
    const int tid             = threadIdx.x;
    const int nt              = blockDim.x;
    const bool has_b          = f + 1 < p.f_end;
    const float cli           = (float)p.coslatinv[row];
    const fft::f32x2 scale(f < p.scale_uv_fields ? cli : 1.0f, f + 1 < p.scale_uv_fields ? cli : 1.0f);
    const int h               = pl->h;
    const int M               = sh.M;
    const int mmax0           = p.row_mmax[row];
    const int mmax            = mmax0 < h ? mmax0 : h;
    const fft::PairTable tw{p.table_f32 + pl->off_tw};
    const TC* pre             = p.table_f32 + pl->off_pre;
    float* ya                 = reinterpret_cast<float*>(p.gp) + goff;
    float* yb                 = ya + p.npts;
    const bool al_a           = (goff & 1) == 0;
    const bool al_b           = ((goff + p.npts) & 1) == 0;
    const ModeReaderT<1> rd{p, (long long)(row - p.lat0), 2 * f};
    // mode m of both fields: the 16 bytes (re f, im f, re f + 1, im f + 1) of the float intermediate
    auto mode = [&](int m) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) float* gfloat_ptr;
        long long o;
        gfloat_ptr src = (gfloat_ptr)rd.locate(m, o) + o;
        const v4 q     = *reinterpret_cast<const __attribute__((address_space(1))) v4*>(src);
        return C{fft::f32x2(q.x, q.z), fft::f32x2(q.y, q.w)};
    };
    const bool odd = pl->method == fft::FFT_ODD;   // uniform per workgroup
    if (odd) {
        // ---- odd {3,5}-smooth length: complex DIT of length n = h on the Hermitian extension (fft_core.h: row_phase_odd), real parts stored
        const int n  = pl->n;
        const int mm = mmax0 < (n - 1) / 2 ? mmax0 : (n - 1) / 2;
        for (int k = tid; k < n; k += nt) {
            C z{fft::f32x2(0.f), fft::f32x2(0.f)};
            if (k <= mm) {
                z = mode(k);
                if (k == 0) {
                    z.im = fft::f32x2(0.f);   // the imaginary part of the mean is dropped
                }
            }
            else if (n - k <= mm) {
                z = fft::cconj(mode(n - k));
            }
            work[fft::PAD(fft::pos_of_freq(sh, k))] = z;
        }
    }
    else {
        // ---- load + c2r pre-processing: four elements per sweep, their loads issued as one batch (fft_core.h: row_phase)
        constexpr int NB = 4;
        for (int k0 = tid; k0 < h; k0 += NB * nt) {
            C A[NB], B[NB];
            TC P[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k  = k0 + i * nt;
                const int kc = k < h ? k : h - 1;
                A[i]         = mode(fft::row_mode_index(mmax, kc));
                B[i]         = mode(fft::row_mode_index(mmax, h - kc));
                P[i]         = pre[kc];
            }
            AA_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k = k0 + i * nt;
                if (k < h) {
                    const C a = fft::row_mode_mask(A[i], mmax, k, h);
                    const C b = fft::cconj(fft::row_mode_mask(B[i], mmax, h - k, h));
                    work[fft::PAD(fft::pos_of_freq(sh, k))] = fft::c2r_pre(a, b, both(P[i]));
                }
            }
        }
    }
    __syncthreads();
    // ---- inverse DIT, stages in reverse order
    for (int i = sh.nstages - 1; i >= 0; --i) {
        const int L = fft::stage_L(sh, i), lsh = sh.lsh[i];
        switch (sh.radix[i]) {
            case 2: dit_stage_pair<2>(work, M, L, lsh, tw, tid, nt); break;
            case 3: dit_stage_pair<3>(work, M, L, lsh, tw, tid, nt); break;
            case 4: dit_stage_pair<4>(work, M, L, lsh, tw, tid, nt); break;
            case 5: dit_stage_pair<5>(work, M, L, lsh, tw, tid, nt); break;
            case 8: dit_stage_pair<8>(work, M, L, lsh, tw, tid, nt); break;
            case 9: dit_stage_pair<9>(work, M, L, lsh, tw, tid, nt); break;
            case 16: dit_stage_pair<16>(work, M, L, lsh, tw, tid, nt); break;
        }
        __syncthreads();
    }
    if (odd) {
        // ---- store: y[j] = Re z[j]
        for (int j = tid; j < nx; j += nt) {
            const fft::f32x2 re = work[fft::PAD(j)].re * scale;
            ya[j]               = re.v.x;
            if (has_b) {
                yb[j] = re.v.y;
            }
        }
        return;
    }
    // ---- store: y[2 j] = Re z[j], y[2 j + 1] = Im z[j]; lane x to field f, lane y to field f + 1
    const bool whole = nx == 2 * h;
    for (int j = tid; j < h; j += nt) {
        const C z           = work[fft::PAD(j)];
        const fft::f32x2 re = z.re * scale, im = z.im * scale;
        if (whole) {
            store_field_pair(ya, al_a, j, re.v.x, im.v.x);
            if (has_b) {
                store_field_pair(yb, al_b, j, re.v.y, im.v.y);
            }
        }
        else {   // a row of which the grid keeps fewer points than the transform has
            if (2 * j < nx) {
                ya[2 * j] = re.v.x;
                if (has_b) yb[2 * j] = re.v.y;
            }
            if (2 * j + 1 < nx) {
                ya[2 * j + 1] = im.v.x;
                if (has_b) yb[2 * j + 1] = im.v.y;
            }
        }
    }
}

// ---- specialised Bluestein rows --------------------------------------------------------------------------------------------------
// workgroup -> (row index of the launch's list, first field of the pair): fft_block_to_job_index (fft_device.h) with the pair as the unit
__device__ __forceinline__ bool fft_block_to_pair_job_index(const FourierParams& p, int b, int& ri, int& f) {
    const int x     = b & 7;
    const int q     = b >> 3;
    const int j     = q & 7;
    const int npair = (p.f_end - p.f_begin + 1) >> 1;
    const int ngr   = (npair + 7) >> 3;
    int fg;
    if (!fft_unit_to_job(p, ngr, x, q >> 3, ri, fg)) {
        return false;
    }
    const int pi = fg * 8 + j;
    if (pi >= npair) {
        return false;
    }
    f = p.f_begin + 2 * pi;
    return true;
}

#ifndef AA_FFT_PLAIN_WPS
#define AA_FFT_PLAIN_WPS 3   // as fft_kernel.hip: wavefronts per SIMD of the rows that are not row_ct3
#endif
template <class S, bool FAST>
__global__ void __launch_bounds__(S::NT, (FAST ? S::WPS : AA_FFT_PLAIN_WPS)) fft_rows_ct_pair_kernel(FourierParams p) {
    using C = fft::cplxp;
    extern __shared__ double lds_raw[];
    C* work = reinterpret_cast<C*>(lds_raw);
    int ri, f;
    if (!fft_block_to_pair_job_index(p, blockIdx.x, ri, f)) {
        return;
    }
    const int tid        = threadIdx.x;
    constexpr int nt     = S::NT;
    constexpr int NPH    = fft::row_num_phases_ct<S>();
    const FftRowDesc d   = p.desc[ri];   // one 64-byte scalar load: everything about the row
    const long long goff = (long long)f * p.npts + d.goff_rel;
    fft::RowTablesCtT<C> r;
#if defined(AA_FFT_ABLATE)
    r.abl = p.abl;
#endif
    r.n = d.n;
    r.h = d.h;
    const fft::PairTable table{p.table_f32};
    r.tw     = table + d.off_tw;
    r.pre    = table + d.off_pre;
    r.chirp  = table + d.off_chirp;
    r.bhat_t = table + d.off_bhat_t;
    fft::RowOut io;
    io.mmax           = d.mmax;
    io.y              = reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff);
    io.aligned16      = ((goff & 1) == 0);
    io.f32            = 1;
    io.scale          = (f < p.scale_uv_fields) ? d.coslatinv : 1.0;   // (an even number of wind fields: fourier_pairs_usable)
    io.pair_stride    = p.npts;
    io.pair_b         = f + 1 < p.f_end;
    io.pair_b_aligned = (((goff + p.npts) & 1) == 0);
    if constexpr (FAST && ct3_fast_path<S>()) {
        // the lines (sixteen fields x one wavenumber) a later job of this XCD will gather: requested into L2 by the eight pair jobs of
        // this group together (fft_kernel.hip does the same for the eight fields of an fp64 group)
        PrefetchJob pfj{-1, 0, 0, 0, 1};
        if (p.pf_dist > 0) {
            const int npair = (p.f_end - p.f_begin + 1) >> 1;
            const int ngr   = (npair + 7) >> 3;
            const int pi    = (f - p.f_begin) >> 1;
            const int fg    = pi >> 3;
            int ri2, fg2;
            if (fft_unit_to_job(p, ngr, blockIdx.x & 7, (blockIdx.x >> 6) + p.pf_dist, ri2, fg2)) {
                const int left = npair - fg * 8;
                pfj.lat_local  = p.desc[ri2].row - p.lat0;
                pfj.mmax       = p.desc[ri2].mmax;
                pfj.f0         = p.f_begin + fg2 * 16;
                pfj.j          = pi - fg * 8;
                pfj.nj         = left < 8 ? left : 8;
            }
        }
        row_ct3<S, true>(p, r, io, (long long)(d.row - p.lat0), f, work, tid, pfj, [](int) {});
        return;
    }
    else {
        gather_modes_to_lds<true>(p, (long long)(d.row - p.lat0), f, io.mmax, work, tid, nt);
        __syncthreads();
        for_each_phase<S, 0>([&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            fft::row_phase_ct<S, true>(ph, tid, nt, r, work, io, work);
            if constexpr (ph < NPH - 1) {
                if constexpr (S::wave_local_middle() && ph >= 1 && ph <= NPH - 3) {
                    wave_lds_fence();   // producer and consumer lanes of the next phase are in this wavefront (fft_kernel.hip)
                }
                else {
                    __syncthreads();
                }
            }
        });
    }
}

template <class S, bool FAST>
static hipError_t launch_ct_pair_t(FourierParams p, int lds_bytes, unsigned nblk, hipStream_t stream) {
    if (hipError_t e = ensure_dynamic_lds<&fft_rows_ct_pair_kernel<S, FAST>>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    p.nvirt = nblk;
    if (FAST && !ct3_prefetch_safe<&fft_rows_ct_pair_kernel<S, FAST>>("two fields per job")) {   // fft_ct_rows.h
        p.pf_dist = 0;
    }
    hipLaunchKernelGGL((fft_rows_ct_pair_kernel<S, FAST>), dim3(nblk), dim3(S::NT), lds_bytes, stream, p);
    return hipGetLastError();
}

template <class S>
static hipError_t launch_ct_pair(const FourierParams& p, int lds_bytes, unsigned nblk, hipStream_t stream) {
    if constexpr (ct3_fast_path<S>()) {
        return launch_ct_pair_t<S, true>(p, lds_bytes, nblk, stream);
    }
    else {
        return launch_ct_pair_t<S, false>(p, lds_bytes, nblk, stream);
    }
}

hipError_t launch_fourier_ct_pairs(const FourierParams& p, int ctf, int ctk, int lds_bytes, hipStream_t stream) {
    if (!p.desc) {
        return hipErrorInvalidValue;
    }
    const int npair       = (p.f_end - p.f_begin + 1) / 2;
    const long long units = (long long)p.nrows * ((npair + 7) / 8);
    const unsigned nblk   = (unsigned)((units + 7) / 8 * 64);
    AA_CT_DISPATCH(ctf, ctk, return launch_ct_pair<S>(p, lds_bytes, nblk, stream))
    return hipErrorInvalidValue;
}

// Can this launch take the two-field form?  (fft_kernel.hip: launch_fourier_dct / launch_fourier_ct ask; ATLAS_AMD_FFT_F32_PAIRS=0
// switches it off.)  The pair's 16 bytes must be one aligned element of a single-piece intermediate: first field even, no m-sharded
// pieces, no packed runs; the wind fields (scaled by 1 / cos(lat)) must not end inside a pair.
static bool pairs_usable(const FourierParams& p) {
    const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_F32_PAIRS");   // read per launch: the tests switch it between calls
    const bool on = !(e && atoi(e) == 0);
    return on && p.f32 && p.table_f32 && !(p.f_begin & 1) && !(p.scale_uv_fields & 1) && p.nparts <= 1 && !p.packed_cols && !(p.RP & 3);
}
bool fourier_pairs_usable(const FourierParams& p, int ctf, int ctk) {   // direct rows
    if (!pairs_usable(p)) {
        return false;
    }
    bool fits = false;
    AA_CT_DISPATCH(ctf, ctk, fits = dct_pair_shape<S>())
    return fits;
}
bool fourier_ct_pairs_usable(const FourierParams& p) {   // specialised Bluestein rows: every shape
    return pairs_usable(p) && p.desc != nullptr;
}

// run-time shaped direct rows (fft_kernel.hip: launch_fourier): can the launch take the two-field form, and its launch
bool fourier_generic_pairs_usable(const FourierParams& p) {
    return pairs_usable(p);
}
hipError_t launch_fourier_generic_pairs(const FourierParams& p, int lds_bytes, int nthreads, hipStream_t stream) {
    if (hipError_t e = ensure_dynamic_lds<&fft_rows_pair_kernel>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    const int npair     = (p.f_end - p.f_begin + 1) / 2;
    const unsigned nblk = fft_job_blocks(p.nrows, npair, 3);
    hipLaunchKernelGGL(fft_rows_pair_kernel, dim3(nblk), dim3(nthreads), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_fourier_dct_pairs(const FourierParams& p, int ctf, int ctk, int lds_bytes, hipStream_t stream) {
    AA_CT_DISPATCH(ctf, ctk, return launch_dct_pair<S>(p, lds_bytes, stream))
    return hipErrorInvalidValue;
}

}  // namespace trans
}  // namespace atlas_amd
