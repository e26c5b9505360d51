#pragma once
// (Included by fft_native.hip -- the dispatcher -- and by fft_native_i{0..3}.hip, which instantiate the kernel: the eight instances
// (fp64 / fp32 x first-stage radix class x one / two fields per workgroup) take two minutes to compile in one translation unit.)
// Native mixed-radix rows of the longitudinal inverse real FFT on gfx950 (fft_native.h): ONE kernel for every shape -- the
// workgroup reads its row's stage list from a 128-byte record and switches, per stage, into the butterfly instantiated for that
// radix; which elements a butterfly touches and its twiddle come from per-shape tables, so the stage code is loads, the
// butterfly, stores.  One workgroup of 256 workers per (row, field); the row lives in LDS (h + 1 complex elements: 41 KB for the
// longest native row of O1280 where its Bluestein form needs 80 KB), three workgroups per CU.
//
// Reference being replaced: the c2r FFT behind TransLocal::invtrans_fourier_reduced (src/atlas/trans/local/TransLocal.cc:1155-1196;
// FFTW / pocketfft transform every row length natively, src/atlas/linalg/fft/FFTW.cc:38-61).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "../../atlas_amd/csrc/env.h"
#include "../../atlas_amd/csrc/device_structs.h"
#include "../../atlas_amd/csrc/dyn_lds.h"
#include "../../atlas_amd/csrc/fft_device.h"
#include "fft_native.h"

namespace atlas_amd {
namespace trans {

using fft::NAT_MAX_FOLD;
using fft::NAT_MAX_ROUNDS;
using fft::NAT_MAX_STAGES;
using fft::NAT_NT;

// BIGP: the first stage's radix is a prime 17 .. 31 (124 registers of data: three workgroups per CU); else 3 .. 15 (four)
template <bool BIGP, int NF>
constexpr int nat_waves_per_simd() {
    return (BIGP || NF > 1) ? 3 : 4;   // (two fields per workgroup at 128 registers: 300 - 600 spilled)
}

// the work array in LDS (fft_native.h: the butterflies go through an accessor, the host emulation has a plain pointer behind it)
template <class C>
struct NatLdsAccess {
    C* p;
    __device__ __forceinline__ C ld(int i) const { return p[i]; }
    __device__ __forceinline__ void st(int i, C v) const { p[i] = v; }
};

// One workgroup of 256 workers per (row, NF consecutive fields of a field group), NF = 1 or 2.  With two fields the 128-byte
// lines of the intermediate (8 fields x one wavenumber) give the workgroup 32 bytes per visit instead of 16 (the second field's
// request follows the first's at once and finds the line on its way), the row's tables and twiddles are fetched once for both,
// and every barrier interval carries two butterflies per worker: these rows are bound by the line fills of their gather and by
// a job's fixed latencies, not by arithmetic (profiles/r04_fft_native.txt).  LDS: NF work arrays + the prefetch dump area.
// [A form that walks through the 8 fields of a field group with the next field's modes gathered into a staging area of their own
// during the stages of the current one was built and measured: as a loop body the radix switches spill 800 - 2700 registers
// (everything in them becomes loop-invariant and is hoisted), as a function that is not inlined the callee-saved registers are
// restored behind the gather's requests, which return in order: 3.8 ms against 1.6.]
template <bool F32, bool BIGP, int NF>
__global__ void __launch_bounds__(NAT_NT, (nat_waves_per_simd<BIGP, NF>())) fft_rows_nat_kernel(FourierParams p) {
    using C    = std::conditional_t<F32, fft::cplxf, cplx>;
    using Real = typename C::real;
    extern __shared__ double lds_raw[];
    // workgroup -> (row, first field): as fft_block_to_job_index, with FGROUP / NF jobs per field group
    constexpr int JPG = FGROUP / NF;                    // jobs per field group (8 or 4)
    constexpr int JSH = NF == 1 ? 3 : 2;
    const int ngr     = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    int ri, fg;
    {
        const int q = blockIdx.x >> 3;
        if (!fft_unit_to_job(p, ngr, blockIdx.x & 7, q >> JSH, ri, fg)) {
            return;
        }
        fg = fg * FGROUP + (q & (JPG - 1)) * NF;        // first field of the job, relative to f_begin
    }
    const int f   = p.f_begin + fg;
    const int nfl = (p.f_end - f) < NF ? (p.f_end - f) : NF;   // fields of this job
    if (nfl <= 0) {
        return;
    }
    const FftNatDesc d = p.ndesc[ri];   // scalar loads: everything about the row
    const int t        = threadIdx.x;
    const int h        = d.h;
    const int ns       = d.ns;
    const int mmax     = d.mmax;
    NatLdsAccess<C> W[NF];
#pragma unroll
    for (int l = 0; l < NF; ++l) {
        W[l].p = reinterpret_cast<C*>(lds_raw) + l * d.lds_elems;
    }
    const C* table;
    if constexpr (F32) {
        table = p.table_f32;
    }
    else {
        table = p.table;
    }
    const C* __restrict__ tw         = table + d.off_tw;
    const C* __restrict__ pre        = table + d.off_pre;
    const uint32_t* __restrict__ ntb = p.nat_table;

    // dev profiling (atlas_amd__Trans__fft_phase_profile): shader-clock time of worker 0 per phase, slots 16.. of p.prof
    const bool prof          = p.prof != nullptr && t == 0;
    unsigned long long tprev = prof ? clock64() : 0;
    auto stamp               = [&](int slot) {
        if (prof) {
            const unsigned long long tn = clock64();
            atomicAdd(&p.prof[16 + slot], tn - tprev);
            tprev = tn;
        }
    };

    // ---- the kept modes X[0..mmax] of the job's fields into LDS; the staging area of a field aliases its work array.  fp64:
    // LDS-DMA, the requests of the two fields for the same wavenumbers back to back (same 128-byte lines)
#if defined(AA_FFT_ABLATE)
    if (!(p.abl & 32))   // dev: bit 5 leaves the gather out (results wrong): what the phase costs
#endif
    {
        if constexpr (F32 || NF == 1) {
#pragma unroll
            for (int l = 0; l < NF; ++l) {
                if (l < nfl) {
                    gather_modes_to_lds<F32>(p, (long long)(d.row - p.lat0), f + l, mmax, W[l].p, t, NAT_NT);
                }
            }
        }
        else {
            const ModeReaderT<0> rd{p, (long long)(d.row - p.lat0), 2 * f};
            for (int m0 = 0; m0 <= mmax; m0 += NAT_NT) {
                const int m = m0 + t;
                if (m <= mmax) {
                    const double* src = rd.address(m);
#pragma unroll
                    for (int l = 0; l < NF; ++l) {
                        if (l < nfl) {
                            C* dst = W[l].p + m0 + (t & ~63);   // wave-uniform
                            __builtin_amdgcn_global_load_lds(
                                reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src + 2 * l)),
                                reinterpret_cast<__attribute__((address_space(3))) void*>(
                                    static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst))),
                                16, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- table values, requested while the gather is in flight: fold pairs k = t + NAT_NT * i (c2r factor, positions of Z[k] and
    // Z[h-k]); per stage slot and round the entry of this worker's butterfly (first input element | twiddle index << 16).
    // Slots: first (e = 0), up to two middle stages (e = 1, 2), last (e = ns - 1)
    const int half  = h >> 1;
    const int nfold = (half + NAT_NT) / NAT_NT;
    C fw[NAT_MAX_FOLD];
    unsigned pk[NAT_MAX_FOLD], ph[NAT_MAX_FOLD];
#pragma unroll
    for (int i = 0; i < NAT_MAX_FOLD; ++i) {
        if (i < nfold) {
            const int k  = t + NAT_NT * i;
            const int kc = k <= half ? k : half;
            fw[i]        = pre[kc];
            pk[i]        = ntb[d.perm + kc];
            ph[i]        = ntb[d.perm + (kc ? h - kc : 0)];
        }
    }
    const int e_last   = ns - 1;
    const int tab_last = ns == 2 ? d.tab[1] : (ns == 3 ? d.tab[2] : d.tab[3]);
    const int nb_last  = ns == 2 ? d.nb[1] : (ns == 3 ? d.nb[2] : d.nb[3]);
    const int st_last  = ns == 2 ? d.stride[1] : (ns == 3 ? d.stride[2] : d.stride[3]);
    const int rx_last  = ns == 2 ? d.radix[1] : (ns == 3 ? d.radix[2] : d.radix[3]);
    unsigned ent0[NAT_MAX_ROUNDS], ent1[NAT_MAX_ROUNDS], ent2[NAT_MAX_ROUNDS], entl[NAT_MAX_ROUNDS];
    C w1s[NAT_MAX_ROUNDS], w2s[NAT_MAX_ROUNDS], wls[NAT_MAX_ROUNDS];
#pragma unroll
    for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {
        const int b = t + NAT_NT * r;
        ent0[r]     = ntb[d.tab[0] + (b < d.nb[0] ? b : 0)];
        entl[r]     = ntb[tab_last + (b < nb_last ? b : 0)];
        if (e_last > 1) {
            ent1[r] = ntb[d.tab[1] + (b < d.nb[1] ? b : 0)];
        }
        if (e_last > 2) {
            ent2[r] = ntb[d.tab[2] + (b < d.nb[2] ? b : 0)];
        }
    }
    stamp(0);
    __syncthreads();   // the gather has landed (and so have the table values)
    stamp(1);
    // ---- L2 prefetch for a later job of this XCD (as the Bluestein rows do, fft_kernel.hip: PrefetchJob): one request per 128-byte
    // line of the 8 fields x one wavenumber that job's field group will gather; this workgroup requests its share (1 / jobs of
    // the group).  The requests are LDS-DMA loads of 4 bytes into a dump area behind the work arrays (64 elements: one 256-byte
    // granule per wavefront): nothing waits for them and -- unlike a load into a register nobody reads -- they cannot land in a
    // register the allocator has meanwhile given to something else (this kernel spills).  They are older than the twiddles of the
    // last stage, whose wait therefore covers them (vector memory requests return in order): none is in flight when the wave ends.
    if (!F32 && p.pf_dist > 0) {
        const int fgi = fg / FGROUP;
        int ri2, fg2;
        if (fft_unit_to_job(p, ngr, blockIdx.x & 7, (blockIdx.x >> (3 + JSH)) + p.pf_dist, ri2, fg2)) {
            const int left  = p.f_end - p.f_begin - fgi * FGROUP;
            const int j     = (fg - fgi * FGROUP) / NF;
            const int nj    = ((left < FGROUP ? left : FGROUP) + NF - 1) / NF;
            const int mmax2 = p.ndesc[ri2].mmax;
            const ModeReaderT<0> rd{p, (long long)(p.ndesc[ri2].row - p.lat0), 2 * (p.f_begin + fg2 * FGROUP)};
            const int L   = mmax2 + 1;
            const int cnt = (L + nj - 1) / nj;
            const int m0  = j * cnt;
            const int m1  = (m0 + cnt < L) ? m0 + cnt : L;
            C* dump       = W[0].p + NF * d.lds_elems + 16 * (t >> 6);   // wave-uniform
            for (int m = m0 + t; m < m1; m += NAT_NT) {
                const double* a = rd.address(m);
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(a)),
                    reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(dump))),
                    4, 0, 0);
            }
        }
    }
    // ---- fold, field by field: Z[k] and Z[h-k] from X[k] and X[h-k] (zero above mmax); the staging area aliases the work array:
    // all reads of a field, barrier, its writes (and the next field's reads), barrier, ...
    {
        const int mclamp = mmax < 0 ? 0 : mmax;
#pragma unroll
        for (int l = 0; l < NF; ++l) {
            if (l < nfl) {
                C Zk[NAT_MAX_FOLD], Zh[NAT_MAX_FOLD];
#pragma unroll
                for (int i = 0; i < NAT_MAX_FOLD; ++i) {
                    if (i < nfold) {
                        const int k  = t + NAT_NT * i;
                        const int kc = k <= half ? k : half;
                        const int kh = h - kc;
                        C A          = W[l].ld(kc <= mmax ? kc : mclamp);
                        C B          = W[l].ld(kh <= mmax ? kh : mclamp);
                        if (kc > mmax) {
                            A = C{0, 0};
                        }
                        if (kh > mmax) {
                            B = C{0, 0};
                        }
                        fft::nat_fold_pair(A, B, fw[i], kc, Zk[i], Zh[i]);
                    }
                }
                if (l == 0) {
                    stamp(2);
                }
                lds_barrier();
                if (l == 0) {
                    stamp(3);
                }
#pragma unroll
                for (int i = 0; i < NAT_MAX_FOLD; ++i) {
                    if (i < nfold) {
                        const int k = t + NAT_NT * i;
                        if (k <= half) {
                            W[l].st(pk[i], Zk[i]);
                            if (k != 0 && 2 * k != h) {
                                W[l].st(ph[i], Zh[i]);
                            }
                        }
                    }
                }
            }
        }
    }
    // twiddles of the second stage: in flight during the first (requested one stage ahead of their use, not all up front: a
    // radix-31 first stage holds 124 registers of data)
#pragma unroll
    for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {
        if (e_last > 1) {
            w1s[r] = tw[ent1[r] >> 16];
        }
        else {
            wls[r] = tw[entl[r] >> 16];
        }
    }
    AA_SCHED_FENCE();
    stamp(4);
    lds_barrier();
    stamp(5);

    // ---- stage 0: radix RL, groups of RL contiguous elements, no twiddles
#pragma unroll
    for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {
        const int b = t + NAT_NT * r;
        if (b < d.nb[0]) {
            const int base = (int)(ent0[r] & 0xffffu);
            auto first     = [&](auto rc) {
                constexpr int RR = decltype(rc)::value;
#pragma unroll
                for (int l = 0; l < NF; ++l) {
                    if (l < nfl) {
                        const NatLdsAccess<C> wa = W[l];
                        auto sink                = [&](int q, C v) { wa.st(base + q, v); };
                        fft::nat_butterfly<RR, false>(wa, base, 1, C{1, 0}, sink);
                    }
                }
            };
            if constexpr (BIGP) {
                AA_NAT_SWITCH_FIRST_BIG(d.radix[0], (first(std::integral_constant<int, RR>{})))
            }
            else {
                AA_NAT_SWITCH_FIRST_SMALL(d.radix[0], (first(std::integral_constant<int, RR>{})))
            }
        }
    }
    stamp(6);
    lds_barrier();
    stamp(7);
    // ---- middle stages
    auto middle = [&](int rx, int nb, int st, const unsigned* en, const C* ws) {
#pragma unroll
        for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {
            const int b = t + NAT_NT * r;
            if (b < nb) {
                const int base = (int)(en[r] & 0xffffu);
                const C w      = ws[r];
                auto mid       = [&](auto rc) {
                    constexpr int RR = decltype(rc)::value;
#pragma unroll
                    for (int l = 0; l < NF; ++l) {
                        if (l < nfl) {
                            const NatLdsAccess<C> wa = W[l];
                            auto sink                = [&](int q, C v) { wa.st(base + q * st, v); };
                            fft::nat_butterfly<RR, true>(wa, base, st, w, sink);
                        }
                    }
                };
                AA_NAT_SWITCH_TW(rx, (mid(std::integral_constant<int, RR>{})))
            }
        }
        lds_barrier();
    };
    if (e_last > 1) {
#pragma unroll
        for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {   // twiddles of the stage after this one
            if (e_last > 2) {
                w2s[r] = tw[ent2[r] >> 16];
            }
            else {
                wls[r] = tw[entl[r] >> 16];
            }
        }
        AA_SCHED_FENCE();
        middle(d.radix[1], d.nb[1], d.stride[1], ent1, w1s);
    }
    if (e_last > 2) {
#pragma unroll
        for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {
            wls[r] = tw[entl[r] >> 16];
        }
        AA_SCHED_FENCE();
        middle(d.radix[2], d.nb[2], d.stride[2], ent2, w2s);
    }
    stamp(8);
    // ---- last stage: DIT stage fused with the scaling and the store of y[2k], y[2k+1], k = b + q * nb
    // (a row starts on a pair boundary in every field or in none when the points per field are even; else decided per field)
    const long long goff0 = (long long)f * p.npts + d.goff_rel;
    fft::RowOut io;
    io.mmax      = mmax;
    io.y         = nullptr;
    io.aligned16 = ((goff0 & 1) == 0) && ((p.npts & 1) == 0 || nfl == 1);
    io.f32       = F32 ? 1 : 0;
    io.scale     = 1.0;
    fft::with_store_flavour(io, [&](auto f32c, auto alc) {
#pragma unroll
        for (int r = 0; r < NAT_MAX_ROUNDS; ++r) {
            const int b = t + NAT_NT * r;
            if (b < nb_last) {
                const int base = (int)(entl[r] & 0xffffu);
                const C w      = wls[r];
                auto last      = [&](auto rc) {
                    constexpr int RR = decltype(rc)::value;
#pragma unroll
                    for (int l = 0; l < NF; ++l) {
                        if (l < nfl) {
                            const long long goff = goff0 + (long long)l * p.npts;
                            fft::RowOut iol      = io;
                            iol.y = F32 ? reinterpret_cast<double*>(reinterpret_cast<float*>(p.gp) + goff) : p.gp + goff;
                            const Real sc = (Real)((f + l < p.scale_uv_fields) ? d.coslatinv : 1.0);
                            const NatLdsAccess<C> wa = W[l];
                            auto sink = [&](int q, C v) {
                                fft::store_pair_t<decltype(f32c)::value, decltype(alc)::value>(iol, (int64_t)(b + q * nb_last),
                                                                                                C{v.re * sc, v.im * sc});
                            };
                            fft::nat_butterfly<RR, true>(wa, base, st_last, w, sink);
                        }
                    }
                };
                AA_NAT_SWITCH_TW(rx_last, (last(std::integral_constant<int, RR>{})))
            }
        }
    });
    stamp(9);
}

template <bool F32, bool BIGP, int NF>
hipError_t launch_nat_t(FourierParams p, int lds_bytes, hipStream_t stream) {
    if (F32) {
        lds_bytes /= 2;   // 8-byte elements
        if (!p.table_f32) {
            return hipErrorInvalidValue;
        }
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_LDS_PAD")) {  // dev tool: occupancy sensitivity (more LDS per workgroup)
        lds_bytes += atoi(e);
    }
    if (hipError_t e = ensure_dynamic_lds<&fft_rows_nat_kernel<F32, BIGP, NF>>(lds_bytes); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    const int ngr         = (p.f_end - p.f_begin + FGROUP - 1) / FGROUP;
    const long long units = (long long)p.nrows * ngr;
    const unsigned nblk   = (unsigned)((units + 7) / 8 * 8 * (FGROUP / NF));
    static const bool debug = atlas_amd::env_get("ATLAS_AMD_FFT_DEBUG") != nullptr;
    if (debug) {
        int per_cu = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fft_rows_nat_kernel<F32, BIGP, NF>, NAT_NT, lds_bytes);
        hipFuncAttributes fa{};
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&fft_rows_nat_kernel<F32, BIGP, NF>));
        std::fprintf(stderr, "[atlas_amd] fft native f32=%d bigp=%d fields/job=%d lds=%d jobs=%u regs=%d scratch=%zu -> %d workgroups/CU\n",
                     (int)F32, (int)BIGP, NF, lds_bytes, nblk, fa.numRegs, (size_t)fa.localSizeBytes, per_cu);
    }
    p.nvirt = nblk;
    hipLaunchKernelGGL((fft_rows_nat_kernel<F32, BIGP, NF>), dim3(nblk), dim3(NAT_NT), lds_bytes, stream, p);
    return hipGetLastError();
}


}  // namespace trans
}  // namespace atlas_amd
