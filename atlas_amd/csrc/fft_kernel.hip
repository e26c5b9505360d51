// Longitudinal inverse real FFT on gfx950: one workgroup per (latitude row, field); the row lives in LDS and is
// transformed in place by the barrier-separated phases of fft_core.h (c2r pre-processing, optional Bluestein
// chirp-z for row lengths with large prime factors, mixed-radix {2,3,4,5} DIF/DIT passes).
//
// Reference being replaced: TransLocal::invtrans_fourier_reduced / _regular
// (src/atlas/trans/local/TransLocal.cc:1101-1136, 1155-1196) and the FFTW / pocketfft c2r they call
// (src/atlas/linalg/fft/FFTW.cc:38-61).  Output layout gp[f*npts + rowoff(lat) + lon] as TransLocal.cc:1132,1187.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "env.h"
#include "device_structs.h"
#include "dyn_lds.h"
#include "fft_device.h"
#include "fft_ct_rows.h"

// the fp32 variant's specialised rows in fp32 ARITHMETIC (float tables, 8-byte LDS elements, packed v_pk_*_f32); a dev build with
// -DAA_FFT_F32_FP64_ARITH keeps float storage around fp64 arithmetic (rounds 1 - 2)
#ifndef AA_FFT_F32_FAST_WPS
#define AA_FFT_F32_FAST_WPS 2   // wavefronts per SIMD the fp32 row_ct3 kernels are compiled for (3: 13 - 37 spilled registers, C4f32 5.62 vs 5.44 ms)
#endif
#if defined(AA_FFT_F32_FP64_ARITH)
#define AA_FFT_F32_ARITH 0
#else
#define AA_FFT_F32_ARITH 1
#endif

namespace atlas_amd {
namespace trans {

__global__ void __launch_bounds__(FFT_MAX_NTHR) fft_rows_kernel(FourierParams p) {
    extern __shared__ double lds_raw[];
    cplx* work = reinterpret_cast<cplx*>(lds_raw);
    int row, f;
    if (!fft_block_to_job(p, blockIdx.x, row, f)) {
        return;
    }
    const fft::FftRowPlan* pl = p.plans + p.row_plan[row];
    const long long goff      = (long long)f * p.npts + (p.rowoff[row] - p.rowoff[p.lat0]);
    const int nx              = (int)(p.rowoff[row + 1] - p.rowoff[row]);
    double* y                 = p.f32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff) : p.gp + goff;
    float* yf                 = reinterpret_cast<float*>(y);
    const int tid             = threadIdx.x;
    const int FFT_NTHR        = blockDim.x;
    const double scale        = (f < p.scale_uv_fields) ? p.coslatinv[row] : 1.0;
    const int mmax            = p.row_mmax[row];
    const ModeReader rd{p, (long long)(row - p.lat0), 2 * f};
    const int method          = pl->method;
    const int n               = pl->n;
    const int h               = pl->h;

    if (method == fft::FFT_DFT) {
        // odd row length: direct sum (never used by Gaussian grids)
        const cplx* w = p.table + pl->off_pre;
        const int mm  = mmax < n / 2 ? mmax : n / 2;
        for (int k = tid; k < nx; k += FFT_NTHR) {
            double s = mmax >= 0 ? rd(0).re : 0.;
            for (int m = 1; m <= mm; ++m) {
                const cplx t = w[(long long)m * k % n];
                const cplx v = rd(m);
                s += 2.0 * (v.re * t.re - v.im * t.im);
            }
            if (p.f32) {
                yf[k] = (float)(s * scale);
            }
            else {
                y[k] = s * scale;
            }
        }
        return;
    }

    if (method == fft::FFT_ODD) {
        // odd {3,5}-smooth length (classic reduced Gaussian grids): complex DIT of length n, real part stored
        fft::RowTables ro;
        ro.n      = n;
        ro.h      = n;
        ro.method = method;
        ro.shape  = &pl->shape;
        ro.tw     = p.table + pl->off_tw;
        ro.pre = ro.chirp = ro.bhat = nullptr;
        fft::RowOut io;
        io.mmax      = mmax < (n - 1) / 2 ? mmax : (n - 1) / 2;
        io.y         = y;
        io.aligned16 = 0;
        io.scale     = scale;
        io.f32       = p.f32;
        const int nph = fft::row_num_phases_odd(ro);
        for (int ph = 0; ph < nph; ++ph) {
            fft::row_phase_odd(ph, tid, FFT_NTHR, ro, rd, io, work);
            __syncthreads();
        }
        return;
    }
    fft::RowTables r;
    r.n      = n;
    r.h      = h;
    r.method = method;
    r.shape  = &pl->shape;
    r.tw     = p.table + pl->off_tw;
    r.pre    = p.table + pl->off_pre;
    r.chirp  = p.table + pl->off_chirp;
    r.bhat   = p.table + pl->off_bhat;
    fft::RowOut io;
    io.mmax      = mmax < h ? mmax : h;
    io.y         = y;
    io.aligned16 = ((goff & 1) == 0) && (nx == n) && scale == 1.0;
    io.scale     = scale;
    io.f32       = p.f32;

    const int nph = fft::row_num_phases(r);
    unsigned long long tprev = 0;
    const bool prof = p.prof != nullptr && tid == 0;
    if (prof) {
        tprev = clock64();
    }
    for (int ph = 0; ph < nph - 1; ++ph) {
        fft::row_phase(ph, tid, FFT_NTHR, r, rd, io, work);
        __syncthreads();
        if (prof) {
            const unsigned long long tn = clock64();
            // slot: bluestein rows 0..31, direct rows 32..63
            atomicAdd(&p.prof[(method == 1 ? 0 : 32) + (ph < 30 ? ph : 30)], tn - tprev);
            tprev = tn;
        }
    }
    if (io.aligned16) {
        fft::row_phase(nph - 1, tid, FFT_NTHR, r, rd, io, work);
    }
    else {
        // generic store: scaling by 1/cos(lat) and/or unaligned rows
        for (int j = tid; j < h; j += FFT_NTHR) {
            cplx z = work[fft::PAD(j)];
            if (method == 1) {
                z = fft::cmul(z, r.chirp[j]);
            }
            if (p.f32) {
                if (2 * j < nx) yf[2 * j] = (float)(z.re * scale);
                if (2 * j + 1 < nx) yf[2 * j + 1] = (float)(z.im * scale);
            }
            else {
                if (2 * j < nx) y[2 * j] = z.re * scale;
                if (2 * j + 1 < nx) y[2 * j + 1] = z.im * scale;
            }
        }
    }
    if (prof) {
        atomicAdd(&p.prof[(method == 1 ? 0 : 32) + 31], clock64() - tprev);
    }
}

// (phase loops, the [R0,16,16] row function row_ct3 and its trace macros: fft_ct_rows.h)

// One workgroup of S::NT workers per (row, field).  Every mode of the row is fetched from the Fourier intermediate once,
// into an LDS staging area that aliases the work array (phase 0 reads it completely before writing its results).
// FAST: the row_ct3 form (256 registers, two wavefronts per SIMD) where the shape has it
#if defined(AA_COEX)
#define AA_FFT_VGPR_CAP __attribute__((amdgpu_num_vgpr(120)))   // dev build: room for a Legendre workgroup (2 x 136 registers per SIMD)
#else
#define AA_FFT_VGPR_CAP
#endif
template <class S, bool F32, bool FAST>
#ifndef AA_FFT_PLAIN_WPS
#define AA_FFT_PLAIN_WPS 3   // wavefronts per SIMD the plain (not row_ct3) Bluestein rows are compiled for (dev builds: 4)
#endif
// (plain rows whose first butterfly has 20 or 24 points -- M = 320, 384, 5120, 6144 -- get 256 registers as well: at 168 the butterfly's
// 80 - 96 data registers beside its table values spilled 30 - 73 registers; M = 320 / 384 are rows of the headline grid)
__global__ void AA_FFT_VGPR_CAP __launch_bounds__(S::NT, (FAST ? ((F32 && AA_FFT_F32_ARITH) ? AA_FFT_F32_FAST_WPS : S::WPS) : (S::radix(0) >= 20 && !(F32 && AA_FFT_F32_ARITH) ? 2 : AA_FFT_PLAIN_WPS))) fft_rows_ct_kernel(FourierParams p) {
    // the fp32 variant runs in fp32 arithmetic: float tables, 8-byte LDS elements, packed v_pk_*_f32 (-DAA_FFT_F32_FP64_ARITH:
    // float storage around fp64 arithmetic, the form of rounds 1 - 2)
    using C = std::conditional_t<(F32 && AA_FFT_F32_ARITH), fft::cplxf, cplx>;
    extern __shared__ double lds_raw[];
    C* work = reinterpret_cast<C*>(lds_raw);
    int ri, f;
    if (!fft_block_to_job_index(p, blockIdx.x, ri, f)) {
        return;
    }
    const int tid      = threadIdx.x;
    constexpr int nt   = S::NT;
    constexpr int NPH  = fft::row_num_phases_ct<S>();
    const bool prof    = p.prof != nullptr && tid == 0;
    const FftRowDesc d = p.desc[ri];   // one 64-byte scalar load: everything about the row
    const int row             = d.row;
    const long long goff      = (long long)f * p.npts + d.goff_rel;
    const double scale        = (f < p.scale_uv_fields) ? d.coslatinv : 1.0;
    fft::RowTablesCtT<C> r;
#if defined(AA_FFT_ABLATE)
    r.abl    = p.abl;
#endif
    r.n      = d.n;
    r.h      = d.h;
    const C* table;
    if constexpr (std::is_same<C, cplx>::value) {
        table = p.table;
    }
    else {
        table = p.table_f32;
    }
    r.tw     = table + d.off_tw;
    r.pre    = table + d.off_pre;
    r.chirp  = table + d.off_chirp;
    r.bhat_t = table + d.off_bhat_t;
    fft::RowOut io;
    io.mmax      = d.mmax;
    io.y         = F32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff) : p.gp + goff;
    io.aligned16 = ((goff & 1) == 0);
    io.f32       = F32 ? 1 : 0;
    io.scale     = scale;
    unsigned long long tprev = 0;
    if (prof) {
        tprev = clock64();
    }
#if defined(AA_FFT_TRACE)
    // poor man's thread trace: every wavefront records where it runs and the shader clock at its phase boundaries
    unsigned long long* trc = nullptr;
    if (p.trace) {
        const unsigned long long slot = ((unsigned long long)blockIdx.x * (nt / 64) + (tid >> 6)) * AA_FFT_TRACE_WORDS;
        if (slot + AA_FFT_TRACE_WORDS <= p.trace_cap && (tid & 63) == 0) {
            trc = p.trace + slot;
            const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
            const unsigned xcc  = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
            trc[0] = ((unsigned long long)xcc << 32) | hwid;
            trc[1] = __builtin_amdgcn_s_memtime();
        }
    }
#define AA_TRACE_STAMP(k) do { if (trc) trc[(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AA_TRACE_STAMP(k) ((void)0)
#endif
#if !defined(AA_FFT_NO_CT3)
    if constexpr (FAST && ct3_fast_path<S>()) {
        PrefetchJob pfj{-1, 0, 0, 0, 1};
        if (p.pf_dist > 0) {
            const int ngr = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
            const int fg  = (f - p.f_begin) / FGROUP;
            int ri2, fg2;
            if (fft_unit_to_job(p, ngr, blockIdx.x & 7, (blockIdx.x >> 6) + p.pf_dist, ri2, fg2)) {
                const int left = p.f_end - p.f_begin - fg * FGROUP;
                pfj.lat_local  = p.desc[ri2].row - p.lat0;
                pfj.mmax       = p.desc[ri2].mmax;
                pfj.f0         = p.f_begin + fg2 * FGROUP;
                pfj.j          = (f - p.f_begin) - fg * FGROUP;
                pfj.nj         = left < FGROUP ? left : FGROUP;
            }
        }
        row_ct3<S, F32>(p, r, io, (long long)(row - p.lat0), f, work, tid, pfj, [&](int k) { (void)k; AA_TRACE_STAMP(k); });
        return;
    }
#endif
    gather_modes_to_lds<F32>(p, (long long)(row - p.lat0), f, io.mmax, work, tid, nt);
    __syncthreads();
    AA_TRACE_STAMP(2);
    if (prof) {
        const unsigned long long tn = clock64();
        atomicAdd(&p.prof[30], tn - tprev);
        tprev = tn;
    }
    // compile-time recursion over the phases: `#pragma unroll` gives up on the largest shapes ("unrolled size is too
    // large") and would leave a run-time loop around a switch
    for_each_phase<S, 0>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        fft::row_phase_ct<S, true>(ph, tid, nt, r, work, io, work);
        if constexpr (ph < NPH - 1) {
            if constexpr (S::wave_local_middle() && ph >= 1 && ph <= NPH - 3) {
                // producer and consumer lanes of the next phase are in this wavefront: LDS executes a wavefront's
                // instructions in order, only the compiler must not move accesses across this point
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            else {
                __syncthreads();
            }
        }
        if constexpr (ph < 5) {
            AA_TRACE_STAMP(3 + ph);
        }
        if (prof) {
            const unsigned long long tn = clock64();
            atomicAdd(&p.prof[ph], tn - tprev);
            tprev = tn;
        }
    });
}

#if defined(ATLAS_AMD_EXPERIMENTS)
#include "../../tools/experiments/fft_halfwin_rows.inc"   // [R0,16,16] rows with LDS as a half-row window: built, measured, lost [r4]
#endif

// ---- small reduced grids: the coarse Bluestein classes M = 256 / 512 / 1024 (fft_plan.h: PlanOptions::coarse_classes; all three
// run with 64 workers) in ONE launch -- the workgroup derives its row's class from the row's half length and switches into the
// instantiated body.  Three launches of a few microseconds of work each were most of the Fourier stage of TL159 -> O160
// (BASELINE config C2).  Same row code as fft_rows_ct_kernel<S, F32, false>: bit-identical results.
template <class S, bool F32, class C>
__device__ __forceinline__ void coarse_row(const FourierParams& p, const FftRowDesc& d, int f, C* work, int tid) {
    static_assert(S::NT == 64, "the coarse classes share a worker count");
    constexpr int NPH    = fft::row_num_phases_ct<S>();
    const long long goff = (long long)f * p.npts + d.goff_rel;
    fft::RowTablesCtT<C> r;
    r.n = d.n;
    r.h = d.h;
    const C* table;
    if constexpr (std::is_same<C, cplx>::value) {
        table = p.table;
    }
    else {
        table = p.table_f32;
    }
    r.tw     = table + d.off_tw;
    r.pre    = table + d.off_pre;
    r.chirp  = table + d.off_chirp;
    r.bhat_t = table + d.off_bhat_t;
    fft::RowOut io;
    io.mmax      = d.mmax;
    io.y         = F32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff) : p.gp + goff;
    io.aligned16 = ((goff & 1) == 0);
    io.f32       = F32 ? 1 : 0;
    io.scale     = (f < p.scale_uv_fields) ? d.coslatinv : 1.0;
    gather_modes_to_lds<F32>(p, (long long)(d.row - p.lat0), f, io.mmax, work, tid, S::NT);
    __syncthreads();
    for_each_phase<S, 0>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        fft::row_phase_ct<S, true>(ph, tid, S::NT, r, work, io, work);
        if constexpr (ph < NPH - 1) {
            if constexpr (S::wave_local_middle() && ph >= 1 && ph <= NPH - 3) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            else {
                __syncthreads();
            }
        }
    });
}

template <bool F32>
__global__ void __launch_bounds__(64, 3) fft_rows_coarse_kernel(FourierParams p) {
    using C = std::conditional_t<(F32 && AA_FFT_F32_ARITH), fft::cplxf, cplx>;
    extern __shared__ double lds_raw[];
    C* work = reinterpret_cast<C*>(lds_raw);
    int ri, f;
    if (!fft_block_to_job_index(p, blockIdx.x, ri, f)) {
        return;
    }
    const FftRowDesc d = p.desc[ri];
    const int need     = 2 * d.h - 1;   // fft_plan.cpp: coarse_bluestein_length(2 h - 1)
    if (need <= 256) {
        coarse_row<fft::CtShape<1, 8>, F32>(p, d, f, work, (int)threadIdx.x);
    }
    else if (need <= 512) {
        coarse_row<fft::CtShape<1, 9>, F32>(p, d, f, work, (int)threadIdx.x);
    }
    else {
        coarse_row<fft::CtShape<1, 10>, F32>(p, d, f, work, (int)threadIdx.x);
    }
}

// ---- [r4] the same classes with SEVERAL FIELDS OF A ROW PER WAVEFRONT (fp64): a row of Bluestein length 256 has 16 butterflies per
// stage, one of length 512 has 32 in its middle stages -- a 64-lane workgroup per (row, field) left three quarters / half of its lanes
// idle.  Four (M = 256) or two (M = 512) consecutive fields of the row now share the wavefront: 16 / 32 workers each, every field with
// its own work array inside the workgroup's 16 KB, the same row code (row_phase_ct with NTW workers per row) -- a quarter / half of the
// jobs, each as long as before.  The fields' modes are gathered together: lane l requests mode m0 + l / NF of field f0 + l % NF (16 NF
// contiguous bytes of the intermediate per mode, LDS slot m NF + field: linear in the lane, as LDS-DMA needs it) and phase 0 reads its
// field through a strided view.  Per (row, field) the arithmetic is that of coarse_row: bit-identical results.
template <class S, int NF>
__device__ __forceinline__ void coarse_row_multi(const FourierParams& p, const FftRowDesc& d, int f0, cplx* work, int tid) {
    using C = cplx;
    static_assert(S::NT == 64 && 64 % NF == 0, "the fields of a job share one wavefront");
    // every field has a work array of its own inside the launch's LDS, which is sized for ONE row of 1024 points (with
    // -DAA_FFT_LDS_SWIZZLE=0 the padded arrays of four 256-point rows would overrun it)
    static_assert(NF * fft::padded_size(S::M) <= fft::padded_size(1024), "the fields' work arrays fit the launch's LDS");
    constexpr int SG  = 64 / NF;   // workers per field
    constexpr int NPH = fft::row_num_phases_ct<S>();
    const int sub = tid / SG, t = tid - sub * SG;
    const int f   = f0 + sub;
    const long long goff = (long long)f * p.npts + d.goff_rel;
    fft::RowTablesCtT<C> r;
    r.n      = d.n;
    r.h      = d.h;
    r.tw     = p.table + d.off_tw;
    r.pre    = p.table + d.off_pre;
    r.chirp  = p.table + d.off_chirp;
    r.bhat_t = p.table + d.off_bhat_t;
    fft::RowOut io;
    io.mmax      = d.mmax;
    io.y         = p.gp + goff;
    io.aligned16 = ((goff & 1) == 0);
    io.f32       = 0;
    io.scale     = (f < p.scale_uv_fields) ? d.coslatinv : 1.0;
    {
        const ModeReaderT<0> rd{p, (long long)(d.row - p.lat0), 2 * f0};
        const int jm = tid / NF, fs = tid - jm * NF;
        for (int m0 = 0; m0 <= io.mmax; m0 += SG) {
            const int m = m0 + jm;
            if (m <= io.mmax) {
                const double* src = rd.address(m) + 2 * fs;
                cplx* dst         = work + m0 * NF;   // wave-uniform; lane l lands in slot m0 NF + l = (m0 + l / NF) NF + l % NF
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src)),
                    reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst))),
                    16, 0, 0);
            }
        }
    }
    __syncthreads();
    const fft::StridedRaw<C> raw{work, NF, sub};
    C* mine = work + sub * fft::padded_size(S::M);
    for_each_phase<S, 0>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        fft::row_phase_ct<S, true, SG>(ph, t, SG, r, raw, io, mine);
        if constexpr (ph < NPH - 1) {
            if constexpr (S::wave_local_middle() && ph >= 1 && ph <= NPH - 3) {
                wave_lds_fence();
            }
            else {
                __syncthreads();
            }
        }
    });
}

// rows of the launch's list: [0, n0) Bluestein length 1024 (eight jobs per field group), [n0, n0 + n1) 512 (four jobs of two fields),
// the rest 256 (two jobs of four fields); the block ranges of the three follow one another
__host__ __device__ inline unsigned coarse_class_blocks(int nrows, int ngr, int jobs_per_group) {
    const long long units = (long long)nrows * ngr;
    return (unsigned)((units + 7) / 8 * 8 * jobs_per_group);
}
// (two wavefronts per SIMD: 223 registers -- the fields of a job differ per lane, so what coarse_row keeps in scalar registers (output
// pointer, scale, work array base) are vector registers here; at 168 the two shared bodies spill 40 - 55.  Eight instead of ten 16-KB
// workgroups per CU.)
__global__ void __launch_bounds__(64, 2) fft_rows_coarse_multi_kernel(FourierParams p) {
    extern __shared__ double lds_raw[];
    cplx* work    = reinterpret_cast<cplx*>(lds_raw);
    const int ngr = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    unsigned b    = blockIdx.x;
    int cls = 0, row0 = 0;
    for (; cls < 2; ++cls) {
        const unsigned nb = coarse_class_blocks(p.coarse_n[cls], ngr, 8 >> cls);
        if (b < nb) {
            break;
        }
        b -= nb;
        row0 += p.coarse_n[cls];
    }
    const int jpg = 8 >> cls;   // jobs per field group: 8, 4, 2
    const int nfj = 1 << cls;   // fields per job:       1, 2, 4
    const int x   = b & 7;
    const int q   = b >> 3;
    const int j   = q % jpg;
    int ril, fg;
    if (!fft_unit_to_job_n(p.row_affinity, p.coarse_n[cls], ngr, x, q / jpg, ril, fg)) {
        return;
    }
    const int f0 = p.f_begin + fg * FGROUP + j * nfj;
    if (f0 >= p.f_end) {
        return;
    }
    const FftRowDesc d = p.desc[row0 + ril];
    const int tid      = (int)threadIdx.x;
    if (cls == 0) {
        coarse_row<fft::CtShape<1, 10>, false>(p, d, f0, work, tid);
    }
    else if (f0 + nfj > p.f_end) {
        // a job at the end of the fields that is not full: its fields one after the other (written out: as the body of a loop the row
        // code spills -- everything in it becomes loop-invariant and is hoisted)
        if (cls == 1) {
            coarse_row<fft::CtShape<1, 9>, false>(p, d, f0, work, tid);
        }
        else {
            coarse_row<fft::CtShape<1, 8>, false>(p, d, f0, work, tid);
            if (f0 + 1 < p.f_end) {
                __syncthreads();
                coarse_row<fft::CtShape<1, 8>, false>(p, d, f0 + 1, work, tid);
            }
            if (f0 + 2 < p.f_end) {
                __syncthreads();
                coarse_row<fft::CtShape<1, 8>, false>(p, d, f0 + 2, work, tid);
            }
        }
    }
    else if (cls == 1) {
        coarse_row_multi<fft::CtShape<1, 9>, 2>(p, d, f0, work, tid);
    }
    else {
        coarse_row_multi<fft::CtShape<1, 8>, 4>(p, d, f0, work, tid);
    }
}

hipError_t launch_fourier_coarse(const FourierParams& p, int lds_bytes, hipStream_t stream) {
    if (!p.desc) {
        return hipErrorInvalidValue;
    }
    const int ngr         = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    const long long units = (long long)p.nrows * ngr;
    const unsigned nblk   = (unsigned)((units + 7) / 8 * 64);
    FourierParams q       = p;
    q.nvirt               = nblk;
    // fp64: several fields of a short row per wavefront (fft_rows_coarse_multi_kernel); ATLAS_AMD_FFT_COARSE_MULTI=0: one field per workgroup
    const char* em = atlas_amd::env_get("ATLAS_AMD_FFT_COARSE_MULTI");
    if (!p.f32 && !(em && atoi(em) == 0) && p.coarse_n[0] + p.coarse_n[1] + p.coarse_n[2] == p.nrows && p.nparts <= 1 && !p.packed_cols) {
        if (hipError_t e = ensure_dynamic_lds<&fft_rows_coarse_multi_kernel>(lds_bytes); e != hipSuccess) {
            return e;
        }
        const unsigned nb = coarse_class_blocks(p.coarse_n[0], ngr, 8) + coarse_class_blocks(p.coarse_n[1], ngr, 4) +
                            coarse_class_blocks(p.coarse_n[2], ngr, 2);
        q.nvirt = nb;
        hipLaunchKernelGGL(fft_rows_coarse_multi_kernel, dim3(nb), dim3(64), lds_bytes, stream, q);
        return hipGetLastError();
    }
    if (p.f32) {
        if (AA_FFT_F32_ARITH) {
            lds_bytes /= 2;
            if (!p.table_f32) {
                return hipErrorInvalidValue;
            }
        }
        if (hipError_t e = ensure_dynamic_lds<&fft_rows_coarse_kernel<true>>(lds_bytes); e != hipSuccess) {
            return e;
        }
        hipLaunchKernelGGL((fft_rows_coarse_kernel<true>), dim3(nblk), dim3(64), lds_bytes, stream, q);
    }
    else {
        if (hipError_t e = ensure_dynamic_lds<&fft_rows_coarse_kernel<false>>(lds_bytes); e != hipSuccess) {
            return e;
        }
        hipLaunchKernelGGL((fft_rows_coarse_kernel<false>), dim3(nblk), dim3(64), lds_bytes, stream, q);
    }
    return hipGetLastError();
}

// ---- compile-time specialised direct rows (fft_core.h: row_phase_dct): regular grids, smooth rows of reduced grids.
// F32A: the fp32 variant in fp32 ARITHMETIC (float tables, 8-byte LDS elements; the (re, im) pairs compile to packed
// v_pk_{add,mul,fma}_f32); F32 && !F32A: float storage around fp64 arithmetic
// (workers, stage loops and the wavefronts per SIMD the kernel is compiled for: fft_device.h)
template <class S, bool F32, bool F32A>
__global__ void __launch_bounds__(FFT_MAX_NTHR, (dct_waves_per_simd<S, F32A>())) fft_rows_dct_kernel(FourierParams p) {
    using C = std::conditional_t<F32A, fft::cplxf, cplx>;
    extern __shared__ double lds_raw[];
    C* work = reinterpret_cast<C*>(lds_raw);
    int row, f;
    if (!fft_block_to_job(p, blockIdx.x, row, f)) {
        return;
    }
    const fft::FftRowPlan* pl = p.plans + p.row_plan[row];
    const long long goff      = (long long)f * p.npts + (p.rowoff[row] - p.rowoff[p.lat0]);
    const int tid             = threadIdx.x;
    const int nt              = blockDim.x;
    const double scale        = (f < p.scale_uv_fields) ? p.coslatinv[row] : 1.0;
    const int mmax            = p.row_mmax[row];
    fft::RowTablesCtT<C> r;
#if defined(AA_FFT_ABLATE)
    r.abl    = p.abl;
#endif
    r.n = pl->n;
    r.h = pl->h;
    if constexpr (F32A) {
        r.tw  = p.table_f32 + pl->off_tw;
        r.pre = p.table_f32 + pl->off_pre;
    }
    else {
        r.tw  = p.table + pl->off_tw;
        r.pre = p.table + pl->off_pre;
    }
    r.chirp  = nullptr;
    r.bhat_t = nullptr;
    fft::RowOut io;
    io.mmax      = mmax < r.h ? mmax : r.h;
    io.y         = F32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff) : p.gp + goff;
    io.aligned16 = ((goff & 1) == 0);
    io.f32       = F32 ? 1 : 0;
    io.scale     = scale;
    unsigned long long tprev = 0;
    const bool prof = p.prof != nullptr && tid == 0;
    if (prof) {
        tprev = clock64();
    }
    // Phase 0 from an LDS staging area (as the Bluestein rows): every kept mode is fetched once (the row_phase_dct form reads
    // X[k] and X[h-k] from the intermediate, every mode twice, in RL / NB dependent batches: 8 - 10 round trips to L2 / HBM per
    // row); the c2r factors are requested before the gather is waited for.  The staging area aliases the work array: all
    // butterflies of phase 0 are in registers before the first result is written (M / RL <= workers).
    using SR           = fft::CtShapeRev<S>;
    constexpr int RL   = SR::radix(SR::NS - 1);
    constexpr int nbl  = S::M / RL;
    // (the launcher starts dct_workers<S>() workers: launch_dct_t; a shape with a stage of more butterflies than a workgroup
    // has workers -- M = 8192 = [16,16,16,2] -- takes the phase loop of row_phase_dct below)
    if constexpr (dct_workers<S>() <= FFT_MAX_NTHR) {
        gather_modes_to_lds<F32>(p, (long long)(row - p.lat0), f, io.mmax, work, tid, nt);
        const bool act = tid < nbl;
        const int bp   = act ? tid : 0;
        // c2r factors of this worker's butterfly: all of them before the gather is waited for where the registers allow it
        // (RL complex values next to the RL inputs: 168 registers hold 10 fp64 / 20 fp32 pairs of them), else in batches in phase 0
        constexpr bool PRELOAD = RL * sizeof(C) <= 160;
        constexpr int NB = PRELOAD ? RL : (RL % 4 == 0 ? 4 : (RL % 5 == 0 ? 5 : (RL % 3 == 0 ? 3 : (RL % 2 == 0 ? 2 : 1))));
        C P[PRELOAD ? RL : NB], x[RL];
        if constexpr (PRELOAD) {
#pragma unroll
            for (int q = 0; q < RL; ++q) P[q] = r.pre[bp + q * nbl];
        }
        // the twiddles of the later stages as well (one butterfly per worker and stage): the stages then start from LDS alone
        constexpr int NMIDS = SR::NS > 2 ? SR::NS - 2 : 0;
        C wmid[NMIDS > 0 ? NMIDS : 1];
        int mbase[NMIDS > 0 ? NMIDS : 1];
        constexpr int R0   = SR::radix(0);
        constexpr int Ls0  = SR::M / R0;
        // (first butterflies of 20 / 24 points in 16-byte elements -- h = 320, 384, 5120, 6144: F160, F192, F2560, F3072 -- fill the
        // register file by themselves: there the later stages' twiddles are requested BEHIND the first butterfly; requested ahead of it
        // they went to scratch, 16 - 62 spilled registers in the round-4 binary)
        constexpr bool LATE_TW = RL * sizeof(C) >= 320;
        C wlast{};
        auto request_stage_twiddles = [&]() {
            dct_for_each_mid<SR, 1>([&](auto ic) {
                constexpr int I  = decltype(ic)::value;
                constexpr int R  = SR::radix(I);
                constexpr int L  = SR::L(I);
                const int b      = tid < SR::M / R ? tid : 0;
                int blk, j;
                fft::split_index(b, L / R, SR::lsh(I), blk, j);
                mbase[I - 1] = blk * L + j;
                wmid[I - 1]  = r.tw[j * (SR::M / L)];
            });
            wlast = r.tw[tid < Ls0 ? tid : 0];
        };
        if constexpr (!LATE_TW) {
            request_stage_twiddles();
        }
        AA_SCHED_FENCE();
        __syncthreads();
        const int h = r.h;
#pragma unroll
        for (int q0 = 0; q0 < RL; q0 += NB) {
            if constexpr (!PRELOAD) {
#pragma unroll
                for (int i = 0; i < NB; ++i) P[i] = r.pre[bp + (q0 + i) * nbl];
                AA_SCHED_FENCE();
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k = bp + (q0 + i) * nbl;
                const C a   = fft::ct_raw_mode(work, io.mmax, k, h);
                const C c   = fft::cconj(fft::ct_raw_mode(work, io.mmax, h - k, h));
                x[q0 + i]   = fft::c2r_pre(a, c, P[PRELOAD ? q0 + i : i]);
            }
        }
        fft::bfly<RL>(x, +1);
        lds_barrier();   // everybody has read the staging area
        if (act) {
            const int b = fft::dct_first_butterfly<S>(bp);
#pragma unroll
            for (int q = 0; q < RL; ++q) work[fft::PAD(b * RL + q)] = x[q];
        }
        if constexpr (LATE_TW) {
            AA_SCHED_FENCE();
            request_stage_twiddles();
        }
        __syncthreads();
        if (prof) {
            const unsigned long long tn = clock64();
            atomicAdd(&p.prof[32], tn - tprev);
            tprev = tn;
        }
        // ---- DIT stages NS-2 .. 1 (row_phase_dct: dit_stage; here one butterfly per worker, twiddle already in a register)
        dct_for_each_mid_down<SR, SR::NS - 2>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int R = SR::radix(I);
            if (tid < SR::M / R) {
                fft::dit_butterfly_w<R>(work, mbase[I - 1], SR::L(I) / R, wmid[I - 1], +1);
            }
            __syncthreads();
        });
        // ---- DIT stage 0 + store
        if (tid < Ls0) {
            C y[R0];
#pragma unroll
            for (int q = 0; q < R0; ++q) y[q] = work[fft::PAD(tid + q * Ls0)];
            fft::twiddle_apply<R0>(y, wlast);
            fft::bfly<R0>(y, +1);
            using Real = typename C::real;
            fft::with_store_flavour(io, [&](auto f32c, auto alc) {
#pragma unroll
                for (int q = 0; q < R0; ++q) {
                    fft::store_pair_t<decltype(f32c)::value, decltype(alc)::value>(
                        io, tid + q * Ls0, C{y[q].re * (Real)io.scale, y[q].im * (Real)io.scale});
                }
            });
        }
    }
    else {
        const ModeReaderT<(F32 ? 1 : 0)> rd{p, (long long)(row - p.lat0), 2 * f};
        constexpr int NPH = fft::row_num_phases_dct<S>();
        for_each_phase_n<NPH, 0>([&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            fft::row_phase_dct<S>(ph, tid, nt, r, rd, io, work);
            if constexpr (ph < NPH - 1) {
                __syncthreads();
            }
        });
    }
}

#if defined(ATLAS_AMD_EXPERIMENTS)
#include "../../tools/experiments/fft_hybrid_rows.inc"
#else
hipError_t launch_fourier_hyb(const FourierParams&, int, int, hipStream_t) {
    return hipErrorNotSupported;   // hybrid rows are planned only in experiment builds (fft_plan.cpp)
}
hipError_t launch_fourier_nat(const FourierParams&, int, int, int, hipStream_t) {
    return hipErrorNotSupported;   // native mixed-radix rows: likewise (tools/experiments/fft_native.hip)
}
#endif

template <class S, bool F32, bool FAST>
static hipError_t launch_ct_t(FourierParams p, int lds_bytes, unsigned nblk, hipStream_t stream) {
    if (hipError_t e = ensure_dynamic_lds<&fft_rows_ct_kernel<S, F32, FAST>>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    if (F32 && AA_FFT_F32_ARITH) {
        lds_bytes /= 2;   // 8-byte elements
        if (!p.table_f32) {
            return hipErrorInvalidValue;
        }
    }
    if (!p.desc) {
        return hipErrorInvalidValue;
    }
#if defined(AA_FFT_LDS_WRAP)
    lds_bytes = std::min(lds_bytes, (AA_FFT_LDS_WRAP + 1) * 16);   // dev probe: see fft_core.h PAD()
#endif
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_LDS_PAD")) {  // dev tool: occupancy sensitivity (more LDS per workgroup)
        lds_bytes += atoi(e);
        (void)ensure_dynamic_lds<&fft_rows_ct_kernel<S, F32, FAST>>(lds_bytes);
    }
    const unsigned grid = nblk;
    p.nvirt             = nblk;
    if (FAST && !ct3_prefetch_safe<&fft_rows_ct_kernel<S, F32, FAST>>("one field per job")) {   // fft_ct_rows.h
        p.pf_dist = 0;
    }
    const bool debug = atlas_amd::env_get("ATLAS_AMD_FFT_DEBUG") != nullptr;
    if (debug) {
        int per_cu = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fft_rows_ct_kernel<S, F32, FAST>, S::NT, lds_bytes);
        hipFuncAttributes fa{};
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&fft_rows_ct_kernel<S, F32, FAST>));
        std::fprintf(stderr, "[atlas_amd] fft ct M=%d threads=%d lds=%d rows*fields=%u regs=%d scratch=%zu -> %d workgroups/CU\n",
                     S::M, S::NT, lds_bytes, nblk, fa.numRegs, (size_t)fa.localSizeBytes, per_cu);
    }
    hipLaunchKernelGGL((fft_rows_ct_kernel<S, F32, FAST>), dim3(grid), dim3(S::NT), lds_bytes, stream, p);
    return hipGetLastError();
}
#if defined(ATLAS_AMD_EXPERIMENTS)
#include "../../tools/experiments/fft_ct_rows_seq.inc"
#endif
// dev switch ATLAS_AMD_FFT_FAST_M=<M>,<M>,...: only these lengths take the row_ct3 form (A/B runs); unset: every shape that has it
static bool ct3_enabled_for(int M) {
    const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_FAST_M");
    if (!e) {
        return true;
    }
    for (const char* c = e; *c;) {
        if (atoi(c) == M) {
            return true;
        }
        while (*c && *c != ',') ++c;
        if (*c == ',') ++c;
    }
    return false;
}
template <class S>
static hipError_t launch_ct(const FourierParams& p, int lds_bytes, unsigned nblk, hipStream_t stream) {
    if constexpr (ct3_fast_path<S>()) {
#if defined(ATLAS_AMD_EXPERIMENTS)
        const bool halfwin = atlas_amd::env_get("ATLAS_AMD_FFT_HALFWIN") && atoi(atlas_amd::env_get("ATLAS_AMD_FFT_HALFWIN")) != 0;
        if (halfwin && !p.f32) {   // LDS as a half-row window: three workgroups per CU (tools/experiments/fft_halfwin_rows.inc)
            return launch_cth<S>(p, nblk, stream);
        }
#endif
#if defined(ATLAS_AMD_EXPERIMENTS)
        // two jobs per workgroup in sequence, the second gather behind the first job's tail (tools/experiments/fft_ct_rows_seq.inc)
        const char* sq = atlas_amd::env_get("ATLAS_AMD_FFT_SEQ");
        if (sq && atoi(sq) != 0 && !p.f32 && p.nparts <= 1 && !p.packed_cols && p.seq_ok) {
            return launch_ct_seq<S>(p, lds_bytes, stream);
        }
#endif
        if (ct3_enabled_for(S::M)) {
            return p.f32 ? launch_ct_t<S, true, true>(p, lds_bytes, nblk, stream)
                         : launch_ct_t<S, false, true>(p, lds_bytes, nblk, stream);
        }
    }
    return p.f32 ? launch_ct_t<S, true, false>(p, lds_bytes, nblk, stream) : launch_ct_t<S, false, false>(p, lds_bytes, nblk, stream);
}

// fft_kernel_pairs.hip: the fp32 variant's specialised Bluestein rows, two fields per job
bool fourier_ct_pairs_usable(const FourierParams& p);
hipError_t launch_fourier_ct_pairs(const FourierParams& p, int ctf, int ctk, int lds_bytes, hipStream_t stream);

hipError_t launch_fourier_ct(const FourierParams& p, int ctf, int ctk, int lds_bytes, int nthreads,
                             hipStream_t stream) {
    if (AA_FFT_F32_ARITH && fourier_ct_pairs_usable(p)) {
        return launch_fourier_ct_pairs(p, ctf, ctk, lds_bytes, stream);
    }
    const int ngr         = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    const long long units = (long long)p.nrows * ngr;
    const unsigned nblk   = (unsigned)((units + 7) / 8 * 64);
    (void)nthreads;   // the specialised Bluestein kernel has a compile-time worker count (CtShape::NT)
    AA_CT_DISPATCH(ctf, ctk, return launch_ct<S>(p, lds_bytes, nblk, stream))
    return hipErrorInvalidValue;
}

template <class S, bool F32, bool F32A>
static hipError_t launch_dct_t(const FourierParams& p, int lds_bytes, int nthreads, unsigned nblk, hipStream_t stream) {
    if (F32A) {
        lds_bytes /= 2;   // 8-byte elements
    }
    lds_bytes += 256;     // staging of phase 0: modes 0..mmax, mmax <= M (one element more than the work array)
    if (hipError_t e = ensure_dynamic_lds<&fft_rows_dct_kernel<S, F32, F32A>>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    const int workers = dct_workers<S>() <= FFT_MAX_NTHR ? dct_workers<S>() : nthreads;
    hipLaunchKernelGGL((fft_rows_dct_kernel<S, F32, F32A>), dim3(nblk), dim3(workers), lds_bytes, stream, p);
    return hipGetLastError();
}
template <class S>
static hipError_t launch_dct(const FourierParams& p, int lds_bytes, int nthreads, unsigned nblk, hipStream_t stream) {
    if (p.f32) {
        if (!AA_FFT_F32_ARITH) {
            return launch_dct_t<S, true, false>(p, lds_bytes, nthreads, nblk, stream);
        }
        return p.table_f32 ? launch_dct_t<S, true, (AA_FFT_F32_ARITH != 0)>(p, lds_bytes, nthreads, nblk, stream) : hipErrorInvalidValue;
    }
    return launch_dct_t<S, false, false>(p, lds_bytes, nthreads, nblk, stream);
}

// fft_kernel_pairs.hip: the fp32 variant's direct rows, two fields per job
bool fourier_pairs_usable(const FourierParams& p, int ctf, int ctk);
hipError_t launch_fourier_dct_pairs(const FourierParams& p, int ctf, int ctk, int lds_bytes, hipStream_t stream);

hipError_t launch_fourier_dct(const FourierParams& p, int ctf, int ctk, int lds_bytes, int nthreads,
                              hipStream_t stream) {
    if (AA_FFT_F32_ARITH && fourier_pairs_usable(p, ctf, ctk)) {
        return launch_fourier_dct_pairs(p, ctf, ctk, lds_bytes, stream);
    }
    const unsigned nblk = fft_job_blocks(p.nrows, p.f_end - p.f_begin, p.job_group_log2);   // fft_device.h: fft_block_to_job
    AA_CT_DISPATCH(ctf, ctk, return launch_dct<S>(p, lds_bytes, nthreads, nblk, stream))
    return hipErrorInvalidValue;
}

hipError_t launch_fourier(const FourierParams& p, int lds_bytes, int nthreads, hipStream_t stream) {
    if (hipError_t e = ensure_dynamic_lds<&fft_rows_kernel>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    const unsigned nblk = fft_job_blocks(p.nrows, p.f_end - p.f_begin, p.job_group_log2);   // fft_device.h: fft_block_to_job
    hipLaunchKernelGGL(fft_rows_kernel, dim3(nblk), dim3(nthreads), lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace trans
}  // namespace atlas_amd
