// NATIVE mixed-radix rows of the longitudinal inverse real FFT: half lengths h = n/2 whose prime factors are small -- any number
// of {2,3,5,7,11,13} and at most ONE prime 17..NAT_MAX_PRIME -- are transformed by a plain inverse DIT of length h with the
// stage list chosen at PLAN time and executed from tables, instead of a Bluestein convolution of length M >= 2h - 1 (about
// 4.5 x the butterflies, 2.15 x the LDS).  Replaces, for those rows, what the reference gets from FFTW / pocketfft, which
// handle every row length natively (src/atlas/linalg/fft/FFTW.cc:38-61, one plan per distinct nx: TransLocal.cc:652-686;
// call site TransLocal.cc:1155-1196).
//
// One kernel, every shape: the radix of a stage is a compile-time template argument chosen by a (wavefront-uniform) switch,
// everything else -- which elements a butterfly touches, its twiddle -- comes from per-shape tables built by the planner
// (fft_plan.cpp: make_native_plan), so a row costs no index arithmetic beyond one add per element:
//   gather : the row's kept modes X[0..mmax] into LDS (LDS-DMA), zeros up to X[h]
//   fold   : pairs (k, h-k), k <= h/2: the c2r pre-processing needs X[k] and X[h-k] for Z[k] AND for Z[h-k] -- one complex
//            product serves both; results held in registers across a barrier (the staging area aliases the work array), then
//            written to the digit-reversed positions perm[k] of the DIT
//   stage 0: radix RL (odd: the largest prime of h, or 15 / 9 / 5 / 3), groups of RL contiguous elements, no twiddles; odd prime
//            radices are dense DFTs by the symmetric split a_q = x_q + x_{P-q}, b_q = x_q - x_{P-q}:
//              y_j, y_{P-j} = x_0 + sum_q cos(2 pi jq/P) a_q  +/-  i sum_q sin(2 pi jq/P) b_q       ((P-1)^2 real FMAs)
//            with literal coefficients; every output pair is written as soon as it is complete (the inputs are in registers:
//            a radix-31 butterfly needs 124 registers of data, not 248)
//   stages : twiddled DIT stages in place, radices from {2,...,13,15,16}
//   last   : DIT stage fused with the scaling and the store of y[2k], y[2k+1]
// LDS layout: position pos of the transform lives at element nat_pos(pos) = (pos / Ls0) * pitch + pos % Ls0, Ls0 = h / r_0 the
// length of the top-level blocks and pitch >= Ls0 ODD: the fold's digit-reversed writes (lane stride Ls0) and the last stage's
// reads then spread over the LDS banks without a swizzle, and inside a top-level block -- where every other stage works --
// addresses stay base + q * stride.
#pragma once
#include <cstdint>
#include <utility>

#include "../../atlas_amd/csrc/fft_core.h"

namespace atlas_amd {
namespace fft {

constexpr int NAT_MAX_STAGES = 4;
constexpr int NAT_MAX_PRIME  = 31;     // largest radix of the first stage (registers: P complex values + the accumulators)
constexpr int NAT_NT         = 256;    // workers per row
constexpr int NAT_MAX_ROUNDS = 2;      // a stage has at most NAT_MAX_ROUNDS * NAT_NT butterflies
constexpr int NAT_MAX_FOLD   = 6;      // fold pairs per worker: h / 2 + 1 <= NAT_MAX_FOLD * NAT_NT
constexpr int NAT_MAX_H      = 2 * NAT_MAX_FOLD * NAT_NT - 2;

template <int P>
struct OddRoots;
#include "fft_roots_odd.inc"

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <class F, int... I>
AA_HD void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
AA_HD void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// y_j = sum_q x_q exp(dir 2 pi i jq / P), P an odd prime; x is destroyed; sink(j, y_j) receives every output exactly once, each pair
// (j, P - j) as soon as it is complete
template <int P, class C, class Sink>
AA_HD void bfly_odd_stream(C* x, int dir, Sink&& sink) {
    using R         = typename C::real;
    constexpr int K = (P - 1) / 2;
    const C x0      = x[0];
    C s0            = x0;
    static_for<K>([&](auto qc) {
        constexpr int q = decltype(qc)::value + 1;
        const C u = x[q], v = x[P - q];
        x[q]     = cadd(u, v);
        x[P - q] = csub(u, v);
        s0       = cadd(s0, x[q]);
    });
    sink(0, s0);
    static_for<K>([&](auto jc) {
        constexpr int j = decltype(jc)::value + 1;
        R ar = x0.re, ai = x0.im, br = 0, bi = 0;
        static_for<K>([&](auto qc) {
            constexpr int q   = decltype(qc)::value + 1;
            constexpr int idx = (j * q) % P;
            constexpr R c     = (R)OddRoots<P>::c[idx];
            constexpr R s     = (R)OddRoots<P>::s[idx];
            ar += c * x[q].re;
            ai += c * x[q].im;
            br += s * x[P - q].re;
            bi += s * x[P - q].im;
        });
        if (dir < 0) {
            br = -br;
            bi = -bi;
        }
        sink(j, C{ar - bi, ai + br});
        sink(P - j, C{ar + bi, ai - br});
    });
}

constexpr bool nat_is_dense_radix(int r) {
    return r == 7 || r == 11 || r == 13 || r == 17 || r == 19 || r == 23 || r == 29 || r == 31 || r == 37 || r == 41 || r == 43 ||
           r == 47;
}

// one butterfly of a native stage: inputs work[base + q * stride] (times w1^q if TW), outputs to sink(q, y_q)
// (WA: accessor of the work array, ld(i) / st(i, v) -- on the device over an LDS-typed pointer, so that the accesses are LDS
// instructions even inside a function that is not inlined into the kernel; on the host a plain pointer)
template <class C>
struct NatHostAccess {
    C* p;
    C ld(int i) const { return p[i]; }
    void st(int i, C v) const { p[i] = v; }
};
template <int R, bool TW, class WA, class C, class Sink>
AA_HD void nat_butterfly(const WA& work, int base, int stride, C w1, Sink&& sink) {
    C x[R];
#pragma unroll
    for (int q = 0; q < R; ++q) x[q] = work.ld(base + q * stride);
    if constexpr (TW) {
        twiddle_apply<R>(x, w1);
    }
    if constexpr (nat_is_dense_radix(R)) {
        bfly_odd_stream<R>(x, +1, sink);
    }
    else {
        bfly<R>(x, +1);
#pragma unroll
        for (int q = 0; q < R; ++q) sink(q, x[q]);
    }
}

// radices of the twiddled stages (middle and last) / of the first stage
#define AA_NAT_CASE(RR_, CALL) case RR_: { constexpr int RR = RR_; CALL; } break;
#define AA_NAT_SWITCH_TW(R, CALL)                                                                                        \
    switch (R) {                                                                                                         \
        AA_NAT_CASE(2, CALL) AA_NAT_CASE(3, CALL) AA_NAT_CASE(4, CALL) AA_NAT_CASE(5, CALL) AA_NAT_CASE(6, CALL)         \
        AA_NAT_CASE(7, CALL) AA_NAT_CASE(8, CALL) AA_NAT_CASE(9, CALL) AA_NAT_CASE(10, CALL) AA_NAT_CASE(11, CALL)       \
        AA_NAT_CASE(12, CALL) AA_NAT_CASE(13, CALL) AA_NAT_CASE(15, CALL) AA_NAT_CASE(16, CALL)                          \
    }
#define AA_NAT_SWITCH_FIRST(R, CALL)                                                                                     \
    switch (R) {                                                                                                         \
        AA_NAT_CASE(3, CALL) AA_NAT_CASE(5, CALL) AA_NAT_CASE(7, CALL) AA_NAT_CASE(9, CALL) AA_NAT_CASE(11, CALL)        \
        AA_NAT_CASE(13, CALL) AA_NAT_CASE(15, CALL) AA_NAT_CASE(17, CALL) AA_NAT_CASE(19, CALL) AA_NAT_CASE(23, CALL)    \
        AA_NAT_CASE(29, CALL) AA_NAT_CASE(31, CALL)                                                                      \
    }
#define AA_NAT_SWITCH_FIRST_SMALL(R, CALL)                                                                               \
    switch (R) {                                                                                                         \
        AA_NAT_CASE(3, CALL) AA_NAT_CASE(5, CALL) AA_NAT_CASE(7, CALL) AA_NAT_CASE(9, CALL) AA_NAT_CASE(11, CALL)        \
        AA_NAT_CASE(13, CALL) AA_NAT_CASE(15, CALL)                                                                      \
    }
#define AA_NAT_SWITCH_FIRST_BIG(R, CALL)                                                                                 \
    switch (R) {                                                                                                         \
        AA_NAT_CASE(17, CALL) AA_NAT_CASE(19, CALL) AA_NAT_CASE(23, CALL) AA_NAT_CASE(29, CALL) AA_NAT_CASE(31, CALL)    \
    }
constexpr bool nat_radix_tw_ok(int r) {
    return (r >= 2 && r <= 13) || r == 15 || r == 16;
}
constexpr bool nat_radix_first_ok(int r) {
    return r == 3 || r == 5 || r == 7 || r == 9 || r == 11 || r == 13 || r == 15 || r == 17 || r == 19 || r == 23 || r == 29 ||
           r == 31;
}

// ---- what the planner hands to the kernel -----------------------------------------------------------------------
// stages in EXECUTION (DIT) order: stage 0 = first (radix RL, stride 1, no twiddles), ..., stage ns-1 = last (radix r_0, fused with
// the store).  Per stage a table of nb entries, entry of butterfly b: (element index of its first input) | (index into the
// twiddle table exp(2 pi i t / h)) << 16.  perm[k]: element the fold writes Z[k] to.
struct NatShape {
    int h;
    int ns;
    int radix[NAT_MAX_STAGES];    // execution order
    int nb[NAT_MAX_STAGES];       // butterflies of the stage (h / radix)
    int stride[NAT_MAX_STAGES];   // elements between the inputs of a butterfly
    int tab[NAT_MAX_STAGES];      // offset of the stage's table in the uint32 table of the plan set
    int perm;                     // offset of perm[0..h) in the same table
    int lds_elems;                // LDS footprint in complex elements (>= h + 1: the staging area holds X[0..h])
    int pitch;                    // elements per top-level block in LDS (>= h / radix[ns-1], odd)
};

// the fold of one pair: A = X[k], Bh = X[h - k], w = exp(2 pi i k / n) -> Z[k], Z[h - k]  (fft_core.h: c2r_pre for both)
template <class C>
AA_HD void nat_fold_pair(C A, C Bh, C w, int k, C& Zk, C& Zh) {
    if (k == 0) {   // Im X[0] and Im X[h] do not enter (conventions of row_mode())
        A.im  = 0;
        Bh.im = 0;
    }
    const C B = cconj(Bh);
    const C S = cadd(A, B);
    const C D = cmul(csub(A, B), w);
    Zk = C{S.re - D.im, S.im + D.re};   // S + i D
    Zh = C{S.re + D.im, D.re - S.im};   // conj(S) + i conj(D)
}

// ---- host execution of one native row with the kernel's own tables and butterflies (planner tests; NOT a product path) ---
// X: modes 0..h (zero above mmax), y: n = 2h reals
inline void nat_execute_row_host(const NatShape& s, const uint32_t* table, const cplx* tw, const cplx* pre, const cplx* X, int mmax,
                                 double* y, double scale = 1.0) {
    const int h = s.h;
    cplx* work  = new cplx[s.lds_elems];
    for (int i = 0; i < s.lds_elems; ++i) work[i] = cplx{0., 0.};
    for (int m = 0; m <= h; ++m) work[m] = m <= mmax ? X[m] : cplx{0., 0.};
    // fold: all reads before all writes
    const uint32_t* perm = table + s.perm;
    const int npairs     = h / 2 + 1;
    cplx* Zk             = new cplx[npairs];
    cplx* Zh             = new cplx[npairs];
    for (int k = 0; k < npairs; ++k) {
        nat_fold_pair(work[k], work[h - k], pre[k], k, Zk[k], Zh[k]);
    }
    for (int k = 0; k < npairs; ++k) {
        work[perm[k]] = Zk[k];
        if (k != 0 && 2 * k != h) {
            work[perm[h - k]] = Zh[k];
        }
    }
    delete[] Zk;
    delete[] Zh;
    for (int i = 0; i < s.ns; ++i) {
        const uint32_t* tb = table + s.tab[i];
        const bool last    = i == s.ns - 1;
        for (int b = 0; b < s.nb[i]; ++b) {
            const int base = (int)(tb[b] & 0xffffu);
            const cplx w1  = tw[tb[b] >> 16];
            const int st   = s.stride[i];
            const NatHostAccess<cplx> wa{work};
            auto sink_lds  = [&](int q, cplx v) { work[base + q * st] = v; };
            auto sink_out  = [&](int q, cplx v) {
                const int k  = b + q * s.nb[i];   // last stage: butterfly b of the single block, output index b + q * Ls0
                y[2 * k]     = v.re * scale;
                y[2 * k + 1] = v.im * scale;
            };
            if (i == 0) {
                AA_NAT_SWITCH_FIRST(s.radix[i], (nat_butterfly<RR, false>(wa, base, st, w1, sink_lds)))
            }
            else if (!last) {
                AA_NAT_SWITCH_TW(s.radix[i], (nat_butterfly<RR, true>(wa, base, st, w1, sink_lds)))
            }
            else {
                AA_NAT_SWITCH_TW(s.radix[i], (nat_butterfly<RR, true>(wa, base, st, w1, sink_out)))
            }
        }
    }
    delete[] work;
}

}  // namespace fft
}  // namespace atlas_amd
