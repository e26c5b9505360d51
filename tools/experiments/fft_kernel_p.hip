// Persistent form of the specialised Bluestein rows [R0,16,16] (LDS-heavy classes, M >= 3840): one workgroup transforms
// one latitude row for several fields in turn, and the kept modes of the NEXT field travel from the Fourier intermediate
// into registers while the current field is in its last two phases.
//
// Why (profiles/r03_fft_trace.txt, r03_fft_ablate.txt): with one (row, field) per workgroup the gather of the row's modes
// (1280 pieces of 16 bytes, one 128-byte line each) costs 4 - 5 us of a 15 us workgroup during which its 64 - 80 KiB of LDS
// and its registers sit idle -- two such workgroups fill a CU, so nothing else can run there.  Leaving the gather out
// altogether (results wrong) takes 22 % off these classes.  Here worker t requests the modes k = t + 256 q itself
// (16 bytes per lane, each mode once), before the level-1 DIT stage of the previous field; in phase 0 the workers
// publish their modes in the work array (free at that point), and read the partner X[h - k] of the c2r pre-processing
// from there.  Same arithmetic in the same order as row_phase_ct / row_ct3.
//
// Reference being replaced: the per-row c2r of TransLocal::invtrans_fourier_reduced (TransLocal.cc:1155-1196).
//
// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3).  Parity-correct (65 / 65 GPU tests of tests/test_gpu_trans.py), but slower
// than one (row, field) per workgroup (fft_kernel.hip: row_ct3), classes serialised on one stream, TL1279 / O1280 / 137 fields:
//     M = 5120: 1.54 -> 1.79 ms    4608: 1.50 -> 1.71    3840: 1.54 -> 1.64    4096: 0.50 -> 0.52   (loads at the loop top)
//     with P / C requested ahead of the stores of phase 4:  2.33 / 2.19 / 1.76 / 0.61 ms
// The field loop makes everything derived from the worker index and the stage twiddles loop-invariant; hipcc hoists it
// (740 - 1200 bytes of scratch per lane) unless the worker index and the twiddles are laundered through an empty asm per
// field, and the values that must cross the loop boundary (next modes, c2r factors, chirp) still push the widest phases
// over 256 registers (84 - 350 bytes of scratch); scratch reloads count in vmcnt and turn the carefully ordered requests
// into vmcnt(0) waits.  Upper bound of the idea, from the ablation that leaves the gather out: -22 % on these classes.
// To build: copy next to fft_kernel.hip, declare the two launch functions in trans.hip (see git history of round 3).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "device_structs.h"
#include "fft_device.h"

namespace atlas_amd {
namespace trans {

// Block -> (row, first field, number of fields).  As fft_block_to_job: the eight fields that share the 128-byte lines of F
// go to eight blocks of one XCD dispatched back to back; a block then walks `jobs` field groups: f0, f0 + 8, ...
__device__ __forceinline__ bool fft_block_to_jobs(const FourierParams& p, int b, int& row, int& f0, int& nj) {
    const int x   = b & 7;
    const int q   = b >> 3;
    const int j   = q & 7;
    const int u   = (q >> 3) * 8 + x;
    const int ngr = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    const int ngc = (ngr + p.jobs - 1) / p.jobs;   // chunks of field groups
    const int ri  = u / ngc;
    const int c   = u - ri * ngc;
    if (ri >= p.nrows) {
        return false;
    }
    f0 = p.f_begin + c * p.jobs * FGROUP + j;
    if (f0 >= p.f_end) {
        return false;
    }
    const int left = (p.f_end - f0 + FGROUP - 1) / FGROUP;
    nj             = left < p.jobs ? left : p.jobs;
    row            = p.rows[ri];
    return true;
}

// NQ: stage-0 inputs per worker that can lie at or below the highest kept mode (the host guarantees mmax < 256 NQ and
// mmax < h for every row of the launch)
template <class S, bool F32, int NQ>
__global__ void __launch_bounds__(256, 2) fft_rows_ct3p_kernel(FourierParams p) {
    static_assert(ct3_fast_path<S>(), "[R0,16,16] family with 256 workers");
    extern __shared__ double lds_raw[];
    cplx* work = reinterpret_cast<cplx*>(lds_raw);
    int row, f0, nj;
    if (!fft_block_to_jobs(p, blockIdx.x, row, f0, nj)) {
        return;
    }
    constexpr int M    = S::M;
    constexpr int R0   = S::radix(0);
    constexpr int NZ   = (R0 + 1) / 2;
    constexpr int NMID = M / 16;
    constexpr int NBM  = (NMID + 255) / 256;
    static_assert(NQ <= NZ, "");
    const int tid             = threadIdx.x;
    const fft::FftRowPlan* pl = p.plans + p.row_plan[row];
    const int h               = pl->h;
    const int mmax            = p.row_mmax[row];   // < h, < 256 NQ
    const cplx* tw            = p.table + pl->off_tw;
    const cplx* pre           = p.table + pl->off_pre;
    const cplx* chirp         = p.table + pl->off_chirp;
    const cplx* bhat_t        = p.table + pl->off_bhat_t;
    const long long rowrel    = p.rowoff[row] - p.rowoff[p.lat0];
    const double cosinv       = p.coslatinv[row];
    ModeReaderT<(F32 ? 1 : 0)> rd{p, (long long)(row - p.lat0), 0};
#if defined(AA_FFT_TRACE)
    unsigned long long* trc = nullptr;
    if (p.trace) {
        const unsigned long long slot = ((unsigned long long)blockIdx.x * 4 + (tid >> 6)) * 8;
        if (slot + 8 <= p.trace_cap && (tid & 63) == 0) {
            trc = p.trace + slot;
            const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            const unsigned xcc  = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            trc[0] = ((unsigned long long)xcc << 32) | hwid | ((unsigned long long)nj << 48);
            trc[1] = __builtin_amdgcn_s_memtime();
        }
    }
#define AA_TRACE_STAMP_P(k) do { if (trc && j == 0) trc[(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AA_TRACE_STAMP_P(k) ((void)0)
#endif
    // modes k = t + 256 q of field f: clamped address, masked after arrival (branch-free: all requests in flight together)
    cplx X[NQ];
    auto request_modes = [&](int f, int t) {
        rd.f2 = 2 * f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = t + q * 256;
            X[q]        = rd(m <= mmax ? m : (mmax < 0 ? 0 : mmax));
        }
    };
    cplx w0 = tw[tid];
    cplx wm = tw[(tid & 15) * (M / 256)];
    int t   = tid;
    request_modes(f0, t);
    // c2r factors and chirp of the worker's stage-0 inputs / outputs (tables padded to NZ * 256 entries, fft_plan.cpp).  The
    // same for every field of the row, but held only where needed: P from the end of one field's last phase through phase 0
    // of the next, C from before phase 3 through the next phase 0.  Requests never follow the stores of phase 4 in program
    // order: vmcnt counts loads and stores in order, a wait for a load behind them would wait for the stores' acknowledgement.
    cplx P[NZ], C[NZ];
#pragma unroll
    for (int q = 0; q < NZ; ++q) {
        P[q] = pre[t + q * 256];
        C[q] = chirp[t + q * 256];
    }
    for (int j = 0; j < nj; ++j) {
        // made opaque once per field: what the phases derive from the worker index and from the two stage twiddles (addresses,
        // the 15 + R0 - 1 twiddle powers) is invariant over the fields; hoisted out of this loop it is held across all
        // phases and spills (700 - 1200 bytes of scratch per lane, measured)
        asm volatile("" : "+v"(t), "+v"(w0.re), "+v"(w0.im), "+v"(wm.re), "+v"(wm.im));
        const int f  = f0 + j * FGROUP;
        const int pt = fft::PAD(t);
        // ---- phase 0: publish the modes, read the partners, c2r pre-processing + chirp + DIF stage 0
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = t + q * 256;
            if (m > mmax) {
                X[q] = cplx{0., 0.};
            }
            if (m == 0) {
                X[q].im = 0.;   // conventions of row_mode(): the imaginary part of the mean is dropped
            }
        }
        if (j > 0) {
            lds_barrier();   // the previous field's last phase has read the work array
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            work[t + q * 256] = X[q];   // staging in natural order; entries above mmax are never read
        }
        lds_barrier();
        {
            cplx x[R0];
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int k  = t + q * 256;
                const int m2 = h - k;   // partner mode; m2 == h (k == 0) is above mmax
                cplx a       = cplx{0., 0.};
                if (q < NQ) {
                    a = X[q];
                }
                cplx c = cplx{0., 0.};
                if (m2 >= 0 && m2 <= mmax) {
                    c = work[m2];
                }
                const cplx z = fft::cmul(fft::c2r_pre(a, fft::cconj(c), P[q]), C[q]);
                x[q]         = k < h ? z : cplx{0., 0.};
            }
#pragma unroll
            for (int q = NZ; q < R0; ++q) x[q] = cplx{0., 0.};
            fft::bfly<R0>(x, -1);
            cplx w1 = w0;
            w1.im   = -w1.im;
            fft::twiddle_apply<R0>(x, w1);
            lds_barrier();   // the staging area aliases the work array: everybody has read it
#pragma unroll
            for (int q = 0; q < R0; ++q) work[pt + q * 256] = x[q];
        }
        cplx flt[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) flt[q] = bhat_t[q * NMID + t];
        AA_SCHED_FENCE();
        lds_barrier();
        AA_TRACE_STAMP_P(3);
        // ---- phase 1: DIF level 1
#pragma unroll
        for (int ib = 0; ib < NBM; ++ib) {
            const int b = t + ib * 256;
            if (b < NMID) {
                fft::dif_butterfly_w<16>(work, (b >> 4) * 256 + (b & 15), 16, wm, -1);
            }
        }
        wave_lds_fence();
        AA_TRACE_STAMP_P(4);
        // ---- phase 2: last DIF stage * filter spectrum * first DIT stage
#pragma unroll
        for (int ib = 0; ib < NBM; ++ib) {
            const int b = t + ib * 256;
            if (b < NMID) {
                if (ib > 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) flt[q] = bhat_t[q * NMID + b];
                    AA_SCHED_FENCE();
                }
                cplx x[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = work[fft::PAD(b * 16 + q)];
                fft::bfly<16>(x, -1);
#pragma unroll
                for (int q = 0; q < 16; ++q) x[q] = fft::cmul(x[q], flt[q]);
                fft::bfly<16>(x, +1);
#pragma unroll
                for (int q = 0; q < 16; ++q) work[fft::PAD(b * 16 + q)] = x[q];
            }
        }
        wave_lds_fence();
        AA_TRACE_STAMP_P(5);
        // requests: the chirp of the outputs again (phase 4) and the next field's modes (phase 0 of the next round)
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            C[q] = chirp[t + q * 256];
        }
        if (j + 1 < nj) {
            request_modes(f + FGROUP, t);
        }
        AA_SCHED_FENCE();
        // ---- phase 3: DIT level 1
#pragma unroll
        for (int ib = 0; ib < NBM; ++ib) {
            const int b = t + ib * 256;
            if (b < NMID) {
                fft::dit_butterfly_w<16>(work, (b >> 4) * 256 + (b & 15), 16, wm, +1);
            }
        }
        lds_barrier();
        AA_TRACE_STAMP_P(6);
        // ---- phase 4: DIT stage 0 + chirp + store
        {
            const long long goff = (long long)f * p.npts + rowrel;
            fft::RowOut io;
            io.mmax      = mmax;
            io.y         = F32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff) : p.gp + goff;
            io.aligned16 = ((goff & 1) == 0);
            io.f32       = F32 ? 1 : 0;
            io.scale     = (f < p.scale_uv_fields) ? cosinv : 1.0;
            cplx x[R0];
#pragma unroll
            for (int q = 0; q < R0; ++q) x[q] = work[pt + q * 256];
            fft::twiddle_apply<R0>(x, w0);
            fft::bfly<R0>(x, +1);
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                x[q]    = fft::cmul(x[q], C[q]);
                x[q].re = x[q].re * io.scale;
                x[q].im = x[q].im * io.scale;
            }
            if (j + 1 < nj) {
#pragma unroll
                for (int q = 0; q < NZ; ++q) {
                    P[q] = pre[t + q * 256];
                }
                AA_SCHED_FENCE();
            }
            fft::with_store_flavour(io, [&](auto f32c, auto alc) {
#pragma unroll
                for (int q = 0; q < NZ; ++q) {
                    const int k = t + q * 256;
                    if (k < h) {
                        fft::store_pair_t<decltype(f32c)::value, decltype(alc)::value>(io, (int64_t)k, x[q]);
                    }
                }
            });
        }
    }
#if defined(AA_FFT_TRACE)
    if (trc) {
        trc[7] = __builtin_amdgcn_s_memtime();
        if (!trc[2]) {
            trc[2] = trc[3];
        }
    }
#endif
}

template <class S, bool F32, int NQ>
static hipError_t launch_ct3p_t(const FourierParams& p, int lds_bytes, unsigned nblk, hipStream_t stream) {
    // unconditional (cheap): a per-process flag would be wrong for a second device and racy between host threads
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_ct3p_kernel<S, F32, NQ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL((fft_rows_ct3p_kernel<S, F32, NQ>), dim3(nblk), dim3(256), lds_bytes, stream, p);
    return hipGetLastError();
}

template <class S>
static hipError_t launch_ct3p(const FourierParams& p, int max_mmax, int lds_bytes, unsigned nblk, hipStream_t stream) {
    if constexpr (ct3_fast_path<S>()) {
        constexpr int NZ  = (S::radix(0) + 1) / 2;
        constexpr int NQS = NZ < 5 ? NZ : 5;   // T1279-like truncations: at most 1280 kept modes per row
        if (max_mmax < NQS * 256) {
            return p.f32 ? launch_ct3p_t<S, true, NQS>(p, lds_bytes, nblk, stream)
                         : launch_ct3p_t<S, false, NQS>(p, lds_bytes, nblk, stream);
        }
        return p.f32 ? launch_ct3p_t<S, true, NZ>(p, lds_bytes, nblk, stream) : launch_ct3p_t<S, false, NZ>(p, lds_bytes, nblk, stream);
    }
    else {
        return hipErrorInvalidValue;
    }
}

// can the class (ctf, ctk) run in the persistent form?
bool fourier_ct_persistent_supported(int ctf, int ctk) {
    bool ok = false;
    AA_CT_DISPATCH(ctf, ctk, ok = ct3_fast_path<S>())
    return ok;
}

// p.jobs (fields per workgroup, in groups of 8) must be set; max_mmax = highest kept mode over the rows of the launch
hipError_t launch_fourier_ct_persistent(const FourierParams& p, int ctf, int ctk, int max_mmax, int lds_bytes,
                                        hipStream_t stream) {
    const int ngr         = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    const int ngc         = (ngr + p.jobs - 1) / p.jobs;
    const long long units = (long long)p.nrows * ngc;
    const unsigned nblk   = (unsigned)((units + 7) / 8 * 64);
    AA_CT_DISPATCH(ctf, ctk, return launch_ct3p<S>(p, max_mmax, lds_bytes, nblk, stream))
    return hipErrorInvalidValue;
}

}  // namespace trans
}  // namespace atlas_amd
