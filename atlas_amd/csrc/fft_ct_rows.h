// Row functions of the specialised Bluestein rows shared by the Fourier kernels (fft_kernel.hip: one field per job;
// fft_kernel_pairs.hip: the fp32 variant's two fields per job): the compile-time phase loops and row_ct3, the [R0,16,16] rows of
// the LDS-heavy classes as one function.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

#include "device_structs.h"
#include "fft_device.h"

namespace atlas_amd {
namespace trans {

// (trace builds) keeps a value in its register across a stamp; fft_pair.h adds the pair type
__device__ __forceinline__ void pin_register(double& x) {
    asm volatile("" : "+v"(x));
}
__device__ __forceinline__ void pin_register(float& x) {
    asm volatile("" : "+v"(x));
}

// The stage-0 butterfly of a [R0,16,16] row addresses its R0 elements as PAD(t) + 256 q: one address register and immediate offsets
// q * 256 * sizeof(C).  The offset field of the LDS instructions is 16 bits: with 16-byte elements the slots q >= 16 (R0 = 18, 20,
// 24: M = 4608, 5120, 6144) are out of its reach, hipcc then forms one address per such slot, finds the same addresses in phase 0
// (writes) and phase 4 (reads) and keeps them alive in between -- through the two widest phases, i.e. in scratch (4 / 4 / 11
// spilled registers in the round-4 binary).  A second base, PAD(t) + 16 * 256, made opaque at each of the two sites: the far slots
// are immediate offsets again, nothing crosses the phases.
template <class C, int R0>
__device__ __forceinline__ int lds_far_slot(int pt) {
    int v = pt;
    if constexpr (sizeof(C) * 256 * (R0 - 1) > 65535) {
        v = pt + 16 * 256;
        asm volatile("" : "+v"(v));
    }
    return v;
}
template <class C>
__device__ __forceinline__ int lds_far_pick(int q, int pt, int pth) {
    return (sizeof(C) * 256 * q > 65535) ? pth + (q - 16) * 256 : pt + q * 256;
}

// ---- compile-time specialised Bluestein rows (fft_core.h: row_phase_ct) -------------------------------------------
template <int NPH, int PH, class Fn>
__device__ __forceinline__ void for_each_phase_n(Fn&& fn) {
    if constexpr (PH < NPH) {
        fn(std::integral_constant<int, PH>{});
        for_each_phase_n<NPH, PH + 1>(fn);
    }
}
template <class S, int PH, class Fn>
__device__ __forceinline__ void for_each_phase(Fn&& fn) {
    if constexpr (PH < fft::row_num_phases_ct<S>()) {
        fn(std::integral_constant<int, PH>{});
        for_each_phase<S, PH + 1>(fn);
    }
}

// ---- [R0,16,16] rows of the LDS-heavy classes (M >= 3840: two workgroups of four wavefronts per CU, 256 registers):
// the whole row in one function, so that every table value is requested a phase (or more) before its use and the ones
// used twice are kept.  Same arithmetic, in the same order, as row_phase_ct (the host emulation and the planner run that).
//   before the gather completes : stage-0 twiddle w0, c2r factors P and chirp C of this worker's stage-0 butterfly, the
//                                 level-1 twiddle wm (one entry serves the DIF and the DIT stage of the level)
//   after phase 0               : the filter spectrum of the worker's first middle butterfly (used in phase 2)
//   before phase 3              : the chirp of the outputs (phase 4); wm and w0 are kept
// Per-wavefront trace of the previous form (profiles/r03_fft_trace.txt): gather 4.1 us, phase 0 4.0 us, phase 2 1.7 - 4.5 us
// of a 16 us workgroup whose vector-ALU work is 3.8 us per wavefront.
// What a workgroup requests for a LATER workgroup of its XCD: the 128-byte lines (8 fields x one wavenumber) that job will
// gather, so that they come from L2 instead of HBM.  The 8 workgroups of a field group share the lines of the target group.
struct PrefetchJob {
    long long lat_local;   // row of the target job, -1: none
    int mmax;
    int f0;                // first field of the target group
    int j, nj;             // this workgroup's share: j of nj
};

// -DAA_FFT_TRACE_PH0 (with -DAA_FFT_TRACE): 16 words per wavefront and the stamps 3..12 INSIDE phase 0:
//   3 prefetch requested | 4 staging reads issued | 5 returned | 6 c2r + chirp | 7 butterfly | 8 twiddles | 9 past the barrier |
//   10 results written | 11 filter requested | 12 past the barrier | 13 phase 1 | 14 phase 2 | 15 end
#if defined(AA_FFT_TRACE_PH0)
#define AA_FFT_TRACE_WORDS 16
#define AA_STAMP_N(k) ((void)0)
#define AA_STAMP_0(k) do { asm volatile("" ::: "memory"); stamp(k); asm volatile("" ::: "memory"); } while (0)
#define AA_PIN(x, n) do { _Pragma("unroll") for (int q_ = 0; q_ < (n); ++q_) { pin_register((x)[q_].re); pin_register((x)[q_].im); } } while (0)
#else
#define AA_FFT_TRACE_WORDS 8
#define AA_STAMP_N(k) stamp(k)
#define AA_STAMP_0(k) ((void)0)
#define AA_PIN(x, n) ((void)0)
#endif
#ifndef AA_CT3_FLT_LATE
#define AA_CT3_FLT_LATE 1
#endif
// Rows whose middle stages have more than 256 butterflies (M = 4608, 5120, 6144) give the first wavefront(s) a second round.  Both
// rounds use the powers wm^2 .. wm^15 of the level-1 twiddle: left alone, hipcc computes the 14 powers once, ahead of the first
// round, and keeps them (56 registers) beside the round's 16 elements and the 16 filter values in flight -- more than the register
// file holds (a filter value went to scratch right after its request, with a wait for it in front of the phase-1 barrier).
// Each round gets its own opaque copy of wm: the powers are formed where they are used.
template <int NBM, class C>
__device__ __forceinline__ C round_twiddle(C w) {
    if constexpr (NBM > 1) {
        pin_register(w.re);
        pin_register(w.im);
    }
    return w;
}
template <class S, bool F32, class C, class Stamp>
__device__ __forceinline__ void row_ct3(const FourierParams& p, const fft::RowTablesCtT<C>& r, const fft::RowOut& io,
                                        long long lat_local, int f, C* work, int t, const PrefetchJob& pfj, Stamp&& stamp) {
    using Real         = typename C::real;
    constexpr int M    = S::M;
    constexpr int R0   = S::radix(0);
    constexpr int NT   = S::NT;
    constexpr int NZ   = (R0 + 1) / 2;
    constexpr int NMID = M / 16;                      // butterflies of a middle stage
    constexpr int NBM  = (NMID + NT - 1) / NT;        // rounds of the middle stages (2 for M > 4096: first wavefront only)
    const int h        = r.h;
    // rounds ib >= 1 of the middle phases (NBM > 1): worker (t + 64 * rot) % 256 takes butterfly 256 ib + that index.  Whole wavefronts
    // rotate, so the 16 workers of a block of 256 points stay in one wavefront (the middle phases are wavefront-local), and the same
    // worker takes the same butterfly in phases 1, 2 and 3.
    const int trot     = (NBM > 1 && p.mid_rot) ? ((t + 64 * (((int)blockIdx.x >> 3) + ((int)blockIdx.x >> 8))) & (NT - 1)) : t;
    const int pt       = fft::PAD(t);   // t < 256: PAD(t + 256 q) = PAD(t) + 256 q (the swizzle stays inside blocks of 256)
    static_assert(NT == 256, "stage-0 butterfly b == worker t");
#if defined(AA_FFT_ABLATE)
    if (!(p.abl & 32))   // dev: bit 5 leaves the gather out altogether (results wrong): what the phase costs
#endif
    gather_modes_to_lds<F32>(p, lat_local, f, io.mmax, work, t, NT);
    // the staging slots above the last kept mode, up to the last one phase 0 reads: zeros, so that phase 0 reads X[k] and X[h-k]
    // without masks (the exec-masked reads were 1.1 us of a 13 us workgroup, profiles/r03_fft_trace.txt)
    for (int m = io.mmax + 1 + t; m < NZ * 256; m += NT) {
        work[m] = C{0, 0};
    }
    // ---- table values of phases 0, 1, 3 and 4
    const C w0 = r.tw[t];
    const C wm = r.tw[(t & 15) * (M / 256)];
    C P[NZ], Ch[NZ];
#pragma unroll
    for (int q = 0; q < NZ; ++q) {   // the tables are padded to NZ * 256 entries (fft_plan.cpp): no clamp
        P[q] = r.pre[(t + q * 256) * AA_ABL(r, 1)];
        Ch[q] = r.chirp[(t + q * 256) * AA_ABL(r, 1)];
    }
    AA_SCHED_FENCE();
    __syncthreads();
    stamp(2);
    // ---- L2 prefetch for a later job: plain loads whose result nobody reads.  Inline assembly, so that no wait is generated for
    // them; they are older than the filter-spectrum requests below, whose wait (phase 2) therefore covers them (vector memory
    // loads return in order), and `pf` is pinned until then.
    unsigned pf = 0;
    constexpr bool PAIR = F32 && sizeof(C) == 16;   // the fp32 variant's two-field form (fft_pair.h): a line is 16 fields there
    if ((!F32 || PAIR) && pfj.lat_local >= 0) {
        const ModeReaderT<(PAIR ? 1 : 0)> rd{p, pfj.lat_local, 2 * pfj.f0};
        const int L   = pfj.mmax + 1;
        const int cnt = (L + pfj.nj - 1) / pfj.nj;
        const int m0  = pfj.j * cnt;
        const int m1  = (m0 + cnt < L) ? m0 + cnt : L;
        const int S_  = p.pf_sectors;
        const int tot = (m1 - m0) * S_;
        for (int i = t; i < tot; i += NT) {
            const int m     = m0 + i / S_;
            const int sub   = i - (i / S_) * S_;
            const char* a   = rd.byte_address(m) + sub * (128 / S_);
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(a) : "memory");
        }
    }
    AA_STAMP_0(3);
    // ---- phase 0: c2r pre-processing + chirp + DIF stage 0 (one block of M, stride 256), inputs from the staging area
    {
        const C* raw = work;
        C x[R0];
#if defined(AA_FFT_TRACE_PH0)
        C av[NZ], cv[NZ];
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            const int k = t + q * 256;
            const int j = h - k;
            av[q]       = raw_elem(raw, k);
            cv[q]       = raw_elem(raw, j < 0 ? 0 : j);
        }
        AA_STAMP_0(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        AA_PIN(av, NZ);
        AA_PIN(cv, NZ);
        AA_STAMP_0(5);
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            if (q == 0 && t == 0) {
                av[q].im = 0.;
                cv[q].im = 0.;
            }
            x[q] = fft::cmul(fft::c2r_pre(av[q], fft::cconj(cv[q]), P[q]), Ch[q]);
        }
        AA_PIN(x, NZ);
        AA_STAMP_0(6);
#else
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            // X[k] and X[h-k] straight from the staging area: modes above mmax are zeros there, k >= h (padding of the
            // convolution) reads a slot that exists and is multiplied by the zero chirp of the table padding
            const int k = t + q * 256;
            const int j = h - k;
            C a      = raw_elem(raw, k);   // (fft_core.h / fft_pair.h, by argument type)
            C v      = raw_elem(raw, j < 0 ? 0 : j);
            if (q == 0 && t == 0) {   // k = 0: Im X[0] and Im X[h] do not enter (conventions of row_mode())
                a.im = 0.;
                v.im = 0.;
            }
            x[q] = fft::cmul(fft::c2r_pre(a, fft::cconj(v), P[q]), Ch[q]);
        }
#endif
#pragma unroll
        for (int q = NZ; q < R0; ++q) x[q] = C{0, 0};
        fft::bfly<R0>(x, -1);
        AA_PIN(x, R0);
        AA_STAMP_0(7);
        C w1 = w0;
        w1.im   = -w1.im;
        fft::twiddle_apply<R0>(x, w1);
        AA_PIN(x, R0);
        AA_STAMP_0(8);
        lds_barrier();   // the staging area aliases the work array: everybody has read it
        AA_STAMP_0(9);
        const int pth = lds_far_slot<C, R0>(pt);
#pragma unroll
        for (int q = 0; q < R0; ++q) work[lds_far_pick<C>(q, pt, pth)] = x[q];
#if defined(AA_FFT_TRACE_PH0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        AA_STAMP_0(10);
    }
    // filter spectrum of the first middle butterfly: in flight during phase 1
    // (the widest first butterflies leave phase 1 no room for all 16 values beside its own 16 elements: the last FLT_LATE of them
    // are requested behind phase 1 -- they are also the last ones phase 2 uses -- instead of being spilled and reloaded)
    constexpr int FLT_LATE  = (sizeof(C) == 16 && R0 >= 18) ? AA_CT3_FLT_LATE : 0;
    constexpr bool W0_AGAIN = (sizeof(C) == 16 && R0 == 20);
    C flt[16];
#pragma unroll
    for (int q = 0; q < 16 - FLT_LATE; ++q) flt[q] = r.bhat_t[(q * NMID + t) * AA_ABL(r, 2)];
    AA_SCHED_FENCE();
    AA_STAMP_0(11);
    lds_barrier();
    AA_STAMP_0(12);
    AA_STAMP_N(3);
    // ---- phase 1: DIF level 1 (blocks of 256 = 16 consecutive workers, radix 16, stride 16)
#pragma unroll
    for (int ib = 0; ib < NBM; ++ib) {
        const int b = (ib ? trot : t) + ib * NT;
        if (b < NMID) {
            fft::dif_butterfly_w<16>(work, (b >> 4) * 256 + (b & 15), 16, round_twiddle<NBM>(wm), -1);
        }
    }
#pragma unroll
    for (int q = 16 - FLT_LATE; q < 16; ++q) flt[q] = r.bhat_t[(q * NMID + t) * AA_ABL(r, 2)];
    AA_SCHED_FENCE();
    wave_lds_fence();
    AA_STAMP_N(4);
    AA_STAMP_0(13);
    // ---- phase 2: last DIF stage * filter spectrum * first DIT stage (16 contiguous elements, no twiddles)
#pragma unroll
    for (int ib = 0; ib < NBM; ++ib) {
        const int b = (ib ? trot : t) + ib * NT;
        if (b < NMID) {
            if (ib > 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) flt[q] = r.bhat_t[(q * NMID + b) * AA_ABL(r, 2)];
                AA_SCHED_FENCE();
            }
            C x[16];
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
    AA_STAMP_N(5);
    AA_STAMP_0(14);
    asm volatile("" ::"v"(pf));   // the prefetch requests have returned (see above): their register is free from here
    // chirp of the outputs again (kept from phase 0 it costs 4 NZ registers through the two widest phases: spills)
#pragma unroll
    for (int q = 0; q < NZ; ++q) {
        Ch[q] = r.chirp[(t + q * 256) * AA_ABL(r, 3)];
    }
    // (the widest rows request the stage-0 twiddle again as well instead of keeping it through the middle phases: four registers
    // that phase 1 of M = 5120 does not have)
    C w0b = w0;
    if constexpr (W0_AGAIN) {
        w0b = r.tw[t];
    }
    AA_SCHED_FENCE();
    // ---- phase 3: DIT level 1
#pragma unroll
    for (int ib = 0; ib < NBM; ++ib) {
        const int b = (ib ? trot : t) + ib * NT;
        if (b < NMID) {
            fft::dit_butterfly_w<16>(work, (b >> 4) * 256 + (b & 15), 16, round_twiddle<NBM>(wm), +1);
        }
    }
    lds_barrier();
    AA_STAMP_N(6);
    // ---- phase 4: DIT stage 0 + chirp + store (outputs q >= NZ are padding: their butterfly arithmetic is dead)
    {
        C x[R0];
        const int pth = lds_far_slot<C, R0>(pt);
#pragma unroll
        for (int q = 0; q < R0; ++q) x[q] = work[lds_far_pick<C>(q, pt, pth)];
        fft::twiddle_apply<R0>(x, w0b);
        fft::bfly<R0>(x, +1);
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            x[q]    = fft::cmul(x[q], Ch[q]);
            x[q].re = x[q].re * (Real)io.scale;  // 1/cos(lat) for the wind fields, exactly 1 otherwise
            x[q].im = x[q].im * (Real)io.scale;
        }
        fft::with_store_flavour(io, [&](auto f32c, auto alc) {
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int k = t + q * 256;
                if (k < h) {
                    fft::store_pair_t<decltype(f32c)::value, decltype(alc)::value>(io, (int64_t)k * AA_ABL(r, 4), x[q]);
                }
            }
        });
    }
    AA_STAMP_N(7);
    AA_STAMP_0(15);
}

// Host side of row_ct3's L2 prefetch: its requests are plain loads into a register nobody reads, written as inline assembly so that no
// wait is generated for them -- safe only while the register allocator never moves that register, i.e. in a kernel WITHOUT spills (a
// spilling kernel corrupted results in round 4's native-row experiments).  The instances are built without scratch by the toolchain this
// was developed with (ROCm 7.2.0); should another compiler spill one, the launcher switches the prefetch of that instance off
// (same results, the round-3 speed of the class without its prefetch) and says so once.  (first item, for the Fourier kernels.)
template <auto Kernel>
static bool ct3_prefetch_safe(const char* what) {
    static const bool ok = [what]() {
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(Kernel)) != hipSuccess) {
            (void)hipGetLastError();
            return true;   // no information: the kernel as tested
        }
        if (fa.localSizeBytes != 0) {
            std::fprintf(stderr, "[atlas_amd] a row_ct3 Fourier kernel (%s) was compiled with %zu bytes of scratch: its L2 prefetch is switched off\n",
                         what, (size_t)fa.localSizeBytes);
            return false;
        }
        return true;
    }();
    return ok;
}

}  // namespace trans
}  // namespace atlas_amd
