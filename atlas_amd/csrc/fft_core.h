// In-place mixed-radix complex FFT passes shared by the HIP kernel (data in LDS) and by the host planner
// (which runs the identical pass sequence to pre-compute Bluestein filter spectra in the kernel's own
// permuted order, and lets CPU tests exercise the index logic without a GPU).
//
// This replaces the third-party FFT the reference calls through linalg::FFT::inverse_c2r[_many]
// (src/atlas/linalg/fft/FFT.h:27-72, FFTW.cc:38-61, pocketfft.cc:32-60): unnormalised Hermitian c2r.
//
// Conventions
//   * forward DIF  : natural order in  -> "digit-reversed" order out, kernel exp(dir * 2 pi i jk / M)
//   * inverse DIT  : digit-reversed in -> natural order out (mirror image of the DIF stage sequence)
//   * a stage is executed cooperatively: worker `t` of `nt` handles butterflies t, t+nt, ...; stages are
//     separated by a barrier (device: __syncthreads, host: sequential loop over t).
//   * position P after a full DIF holds frequency  freq(P) = sum_i q_i * (r_1 ... r_{i-1})  where q_i are the
//     mixed-radix digits of P, most significant first (digit i has weight M / (r_1 ... r_i)).
//   * radices {2,3,4,5,8,9,16}; the shape builder puts odd radices first so that every later stage works on
//     power-of-two sub-blocks (index arithmetic by shifts), then 16s, then one of 8/4/2.
//   * element i lives at work[PAD(i)], PAD(i) = i + (i >> 4): one pad slot per 16 elements breaks the power-of-two
//     strides that would otherwise put a whole lane group on one LDS bank (measured: 58% of LDS cycles were
//     conflicts without it).
//   * stage twiddles w_L^{jq}, q = 1..R-1, come from ONE table load (w_L^j) and products (depth <= log2 R), which
//     keeps the vector-memory pipeline free for the row data.
#pragma once
#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#define AA_HD __host__ __device__ __forceinline__
#else
#define AA_HD inline
#endif

namespace atlas_amd {
namespace fft {

// complex number of reals R.  The row kernels are written for `cplx` (double); the arithmetic below it (butterflies, twiddles,
// one DIT stage, the direct rows) is templated on the complex type so that the fp32 variant of the transform can run its
// direct rows in fp32 ARITHMETIC as well [r3]: hipcc turns the (re, im) pairs of `cplxf` into packed v_pk_{add,mul,fma}_f32
// (2/3 of the instructions of the fp64 form), LDS and registers per element halve.
template <class R>
struct alignas(2 * sizeof(R)) cplx_t {
    R re, im;
    using real = R;
};
using cplx  = cplx_t<double>;
using cplxf = cplx_t<float>;

template <class C>
AA_HD C cmul(C a, C b) {
    return C{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <class C>
AA_HD C cadd(C a, C b) {
    return C{a.re + b.re, a.im + b.im};
}
template <class C>
AA_HD C csub(C a, C b) {
    return C{a.re - b.re, a.im - b.im};
}
template <class C>
AA_HD C cconj(C a) {
    return C{a.re, -a.im};
}
// multiply by +i (dir=+1) or -i (dir=-1)
template <class C>
AA_HD C cmuli(C a, int dir) {
    return dir > 0 ? C{-a.im, a.re} : C{a.im, -a.re};
}
// multiply by (c + i*dir*s)
template <class C>
AA_HD C cmulw(C a, typename C::real c, typename C::real s, int dir) {
    const typename C::real sd = dir > 0 ? s : -s;
    return C{a.re * c - a.im * sd, a.re * sd + a.im * c};
}

#ifndef AA_FFT_LDS_SWIZZLE
#define AA_FFT_LDS_SWIZZLE 1
#endif
#if AA_FFT_LDS_SWIZZLE
// XOR swizzle of the low 4 index bits with the next 4: a bijection inside every aligned block of 256 elements, so
// the footprint stays exactly M (two M = 5120 rows fit the 160 KiB of a CU) while the power-of-two strides of the
// radix-16/8/4 stages still spread over the 16-byte LDS slots.
#if defined(AA_FFT_LDS_WRAP)
// dev probe (results wrong by construction): every LDS index wrapped into AA_FFT_LDS_WRAP + 1 elements -- the same instruction stream
// and LDS traffic with a fraction of the footprint: the UPPER BOUND of what a smaller exchange window (more jobs per CU) could buy
AA_HD int PAD(int i) {
    return (i ^ ((i >> 4) & 15)) & AA_FFT_LDS_WRAP;
}
#else
AA_HD int PAD(int i) {
    return i ^ ((i >> 4) & 15);
}
#endif
AA_HD constexpr int padded_size(int M) {
    return (M + 255) / 256 * 256;
}
#else
AA_HD int PAD(int i) {
    return i + (i >> 4);
}
AA_HD constexpr int padded_size(int M) {
    return M + (M >> 4) + 1;
}
#endif

constexpr int MAX_STAGES = 12;
#ifndef AA_FFT_KEEP_192
#define AA_FFT_KEEP_192 0
#endif

struct FftShape {           // stage list of an M-point transform
    int M;
    int nstages;
    int radix[MAX_STAGES];  // DIF order
    int lsh[MAX_STAGES];    // log2 of the stage's sub-block length Ls = L/R, or -1 if Ls is not a power of two
};

// ---- radix butterflies: y_q = sum_p x_p exp(dir 2 pi i p q / r) ------------------------------------------------
template <class C>
AA_HD void bfly2(C* x) {
    C a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}
template <class C>
AA_HD void bfly4(C* x, int dir) {
    C a = cadd(x[0], x[2]), b = csub(x[0], x[2]);
    C c = cadd(x[1], x[3]), d = cmuli(csub(x[1], x[3]), dir);
    x[0] = cadd(a, c);
    x[1] = cadd(b, d);
    x[2] = csub(a, c);
    x[3] = csub(b, d);
}
template <class C>
AA_HD void bfly3(C* x, int dir) {
    using R   = typename C::real;
    const R s = (R)0.86602540378443864676372317075294 * dir;  // sin(2pi/3)
    const R h = (R)0.5;
    C t1 = cadd(x[1], x[2]);
    C t2 = C{x[0].re - h * t1.re, x[0].im - h * t1.im};
    C t3 = csub(x[1], x[2]);
    C t4 = C{-s * t3.im, s * t3.re};  // i*s*t3
    x[0] = cadd(x[0], t1);
    x[1] = cadd(t2, t4);
    x[2] = csub(t2, t4);
}
template <class C>
AA_HD void bfly5(C* x, int dir) {
    using R    = typename C::real;
    const R c1 = (R)0.30901699437494742410229341718282;        // cos(2pi/5)
    const R c2 = (R)-0.80901699437494742410229341718282;       // cos(4pi/5)
    const R s1 = (R)0.95105651629515357211643933337938 * dir;  // sin(2pi/5)
    const R s2 = (R)0.58778525229247312916870595463907 * dir;  // sin(4pi/5)
    C a1 = cadd(x[1], x[4]), b1 = csub(x[1], x[4]);
    C a2 = cadd(x[2], x[3]), b2 = csub(x[2], x[3]);
    C x0 = x[0];
    x[0] = C{x0.re + a1.re + a2.re, x0.im + a1.im + a2.im};
    C m1 = C{x0.re + c1 * a1.re + c2 * a2.re, x0.im + c1 * a1.im + c2 * a2.im};
    C m2 = C{x0.re + c2 * a1.re + c1 * a2.re, x0.im + c2 * a1.im + c1 * a2.im};
    C n1 = C{-(s1 * b1.im + s2 * b2.im), s1 * b1.re + s2 * b2.re};
    C n2 = C{-(s2 * b1.im - s1 * b2.im), s2 * b1.re - s1 * b2.re};
    x[1] = cadd(m1, n1);
    x[4] = csub(m1, n1);
    x[2] = cadd(m2, n2);
    x[3] = csub(m2, n2);
}
// radix 8 = 2 x 4:  p = 4 p1 + p0, q = 2 q1 + q0:  w8^{pq} = w2^{p1 q0} w4^{p0 q1} w8^{p0 q0}
template <class C>
AA_HD void bfly8(C* x, int dir) {
    const double r = 0.70710678118654752440084436210485;
    C t0[4], t1[4];
#pragma unroll
    for (int p0 = 0; p0 < 4; ++p0) {
        t0[p0] = cadd(x[p0], x[4 + p0]);
        t1[p0] = csub(x[p0], x[4 + p0]);
    }
    t1[1] = cmulw(t1[1], r, r, dir);
    t1[2] = cmuli(t1[2], dir);
    t1[3] = cmulw(t1[3], -r, r, dir);
    bfly4(t0, dir);
    bfly4(t1, dir);
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) {
        x[2 * q1]     = t0[q1];
        x[2 * q1 + 1] = t1[q1];
    }
}
// radix 16 = 4 x 4:  p = 4 p1 + p0, q = 4 q1 + q0:  w16^{pq} = w4^{p1 q0} w4^{p0 q1} w16^{p0 q0}
template <class C>
AA_HD void bfly16(C* x, int dir) {
    const double c1 = 0.92387953251128675612818318939679;  // cos(pi/8)
    const double s1 = 0.38268343236508977172845998403040;  // sin(pi/8)
    const double r  = 0.70710678118654752440084436210485;
    C t[4][4];  // t[p0][q0]
#pragma unroll
    for (int p0 = 0; p0 < 4; ++p0) {
        C u[4] = {x[p0], x[4 + p0], x[8 + p0], x[12 + p0]};
        bfly4(u, dir);
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) t[p0][q0] = u[q0];
    }
    // internal twiddles w16^{p0 q0}
    t[1][1] = cmulw(t[1][1], c1, s1, dir);
    t[1][2] = cmulw(t[1][2], r, r, dir);
    t[1][3] = cmulw(t[1][3], s1, c1, dir);
    t[2][1] = cmulw(t[2][1], r, r, dir);
    t[2][2] = cmuli(t[2][2], dir);
    t[2][3] = cmulw(t[2][3], -r, r, dir);
    t[3][1] = cmulw(t[3][1], s1, c1, dir);
    t[3][2] = cmulw(t[3][2], -r, r, dir);
    t[3][3] = cmulw(t[3][3], -c1, -s1, dir);
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) {
        C u[4] = {t[0][q0], t[1][q0], t[2][q0], t[3][q0]};
        bfly4(u, dir);
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) x[4 * q1 + q0] = u[q1];
    }
}
// radix 9 = 3 x 3:  p = 3 p1 + p0, q = 3 q1 + q0:  w9^{pq} = w3^{p1 q0} w9^{p0 q0} w3^{p0 q1}
template <class C>
AA_HD void bfly9(C* x, int dir) {
    const double c1 = 0.76604444311897803520239265055542, s1 = 0.64278760968653932632264340990726;   // 40 deg
    const double c2 = 0.17364817766693034885171662676931, s2 = 0.98480775301220805936674302458952;   // 80 deg
    const double c4 = -0.93969262078590838405410927732473, s4 = 0.34202014332566873304409961468226;  // 160 deg
    C t[3][3];  // t[p0][q0]
#pragma unroll
    for (int p0 = 0; p0 < 3; ++p0) {
        C u[3] = {x[p0], x[3 + p0], x[6 + p0]};
        bfly3(u, dir);
#pragma unroll
        for (int q0 = 0; q0 < 3; ++q0) t[p0][q0] = u[q0];
    }
    t[1][1] = cmulw(t[1][1], c1, s1, dir);
    t[1][2] = cmulw(t[1][2], c2, s2, dir);
    t[2][1] = cmulw(t[2][1], c2, s2, dir);
    t[2][2] = cmulw(t[2][2], c4, s4, dir);
#pragma unroll
    for (int q0 = 0; q0 < 3; ++q0) {
        C u[3] = {t[0][q0], t[1][q0], t[2][q0]};
        bfly3(u, dir);
#pragma unroll
        for (int q1 = 0; q1 < 3; ++q1) x[3 * q1 + q0] = u[q1];
    }
}
// exp(dir * 2 pi i k / 360), k in [0,360): the index is a compile-time constant after unrolling, so the entries
// become literal operands
template <class C>
AA_HD C cmul_root360(C a, int k, int dir) {
    constexpr double tab[360][2] = {
#include "fft_roots360.inc"
    };
    if (k == 0) return a;
    if (k == 90) return cmuli(a, dir);
    if (k == 180) return C{-a.re, -a.im};
    if (k == 270) return cmuli(a, -dir);
    return cmulw(a, tab[k][0], tab[k][1], dir);
}

template <int R, class C>
AA_HD void bfly(C* x, int dir);

// composite radix R = R1 * R2 (R a divisor of 360), Cooley-Tukey inside registers:
// p = R2 p1 + p0, q = R1 q1 + q0:  w_R^{pq} = w_R1^{p1 q0} w_R^{p0 q0} w_R2^{p0 q1}
template <int R1, int R2, class C>
AA_HD void bfly_comp(C* x, int dir) {
    constexpr int R = R1 * R2;
    static_assert(360 % R == 0, "root table covers divisors of 360");
    C t[R2][R1];
#pragma unroll
    for (int p0 = 0; p0 < R2; ++p0) {
        C u[R1];
#pragma unroll
        for (int p1 = 0; p1 < R1; ++p1) u[p1] = x[R2 * p1 + p0];
        bfly<R1>(u, dir);
#pragma unroll
        for (int q0 = 0; q0 < R1; ++q0) t[p0][q0] = cmul_root360(u[q0], ((p0 * q0) % R) * (360 / R), dir);
    }
#pragma unroll
    for (int q0 = 0; q0 < R1; ++q0) {
        C u[R2];
#pragma unroll
        for (int p0 = 0; p0 < R2; ++p0) u[p0] = t[p0][q0];
        bfly<R2>(u, dir);
#pragma unroll
        for (int q1 = 0; q1 < R2; ++q1) x[R1 * q1 + q0] = u[q1];
    }
}

template <int R, class C>
AA_HD void bfly(C* x, int dir) {
    if constexpr (R == 2) bfly2(x);
    else if constexpr (R == 3) bfly3(x, dir);
    else if constexpr (R == 4) bfly4(x, dir);
    else if constexpr (R == 5) bfly5(x, dir);
    else if constexpr (R == 8) bfly8(x, dir);
    else if constexpr (R == 9) bfly9(x, dir);
    else if constexpr (R == 16) bfly16(x, dir);
    else if constexpr (R == 6) bfly_comp<3, 2>(x, dir);
    else if constexpr (R == 10) bfly_comp<5, 2>(x, dir);
    else if constexpr (R == 12) bfly_comp<3, 4>(x, dir);
    else if constexpr (R == 20) bfly_comp<5, 4>(x, dir);
    else if constexpr (R == 24) bfly_comp<3, 8>(x, dir);
    else if constexpr (R == 15) bfly_comp<3, 5>(x, dir);
    else if constexpr (R == 18) bfly_comp<9, 2>(x, dir);
    else static_assert(R == 2, "unsupported radix");
}

// powers w^1 .. w^(R-1) by halving products (depth <= log2 R roundings)
template <int R, class C>
AA_HD void twiddle_powers(C w1, C* w) {
    w[0] = C{1, 0};
    w[1] = w1;
#pragma unroll
    for (int q = 2; q < R; ++q) {
        w[q] = cmul(w[q >> 1], w[q - (q >> 1)]);
    }
}

// x[q] *= w1^q, q = 1..R-1, with few live twiddles: q = STEP*j + i walks w1^i * (w1^STEP)^j (chains of at most
// R/STEP products); radix-16/20/24 stages would otherwise hold R twiddles (4 VGPRs each) next to R data elements
template <int R, class C>
AA_HD void twiddle_apply(C* x, C w1) {
    if constexpr (R <= 5) {
        C w[R];
        twiddle_powers<R>(w1, w);
#pragma unroll
        for (int q = 1; q < R; ++q) x[q] = cmul(x[q], w[q]);
    }
    else {
        constexpr int STEP = 4;
        C wi[STEP];
        twiddle_powers<STEP>(w1, wi);
        const C ws = cmul(wi[2], wi[2]);
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
            C t = wi[i];
#pragma unroll
            for (int q = i; q < R; q += STEP) {
                if (q > 0) x[q] = cmul(x[q], t);
                if (q + STEP < R) t = cmul(t, ws);
            }
        }
    }
}

// butterfly index b -> (block, j) for sub-block length Ls (= 1 << lsh when lsh >= 0)
AA_HD void split_index(int b, int Ls, int lsh, int& blk, int& j) {
    if (lsh >= 0) {
        blk = b >> lsh;
        j   = b & (Ls - 1);
    }
    else {
        blk = b / Ls;
        j   = b - blk * Ls;
    }
}

// ---- one DIF stage: blocks of length L, radix R; twiddle table of M entries, w_L^j = tw[j * M/L] ---------------
// (TW: what the table is read through -- a pointer, or the both-lanes accessor of the two-field fp32 form, fft_pair.h)
template <int R, class C, class TW>
AA_HD void dif_stage(C* d, int M, int L, int lsh, TW tw, int dir, int t, int nt) {
    const int Ls  = L / R;
    const int tws = M / L;
    const int nb  = M / R;
    for (int b = t; b < nb; b += nt) {
        int blk, j;
        split_index(b, Ls, lsh, blk, j);
        const int base = blk * L + j;
        C x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = d[PAD(base + q * Ls)];
        bfly<R>(x, dir);
        d[PAD(base)] = x[0];
        if (Ls == 1) {  // j == 0: all twiddles are 1
#pragma unroll
            for (int q = 1; q < R; ++q) d[PAD(base + q)] = x[q];
        }
        else {
            C w1 = tw[j * tws];
            if (dir < 0) w1.im = -w1.im;
            twiddle_apply<R>(x, w1);
#pragma unroll
            for (int q = 1; q < R; ++q) d[PAD(base + q * Ls)] = x[q];
        }
    }
}
// the usual form: a table of the row's own complex type (the explicit table type keeps the __restrict__)
template <int R, class C>
AA_HD void dif_stage(C* d, int M, int L, int lsh, const C* __restrict__ tw, int dir, int t, int nt) {
    dif_stage<R, C, const C* __restrict__>(d, M, L, lsh, tw, dir, t, nt);
}
// ---- one DIT stage (inverse of the DIF stage with the same L, R): twiddle first, then butterfly --------------
template <int R, class C, class TW>
AA_HD void dit_stage(C* d, int M, int L, int lsh, TW tw, int dir, int t, int nt) {
    const int Ls  = L / R;
    const int tws = M / L;
    const int nb  = M / R;
    for (int b = t; b < nb; b += nt) {
        int blk, j;
        split_index(b, Ls, lsh, blk, j);
        const int base = blk * L + j;
        C x[R];
        x[0] = d[PAD(base)];
        if (Ls == 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) x[q] = d[PAD(base + q)];
        }
        else {
            C w1 = tw[j * tws];
            if (dir < 0) w1.im = -w1.im;
#pragma unroll
            for (int q = 1; q < R; ++q) x[q] = d[PAD(base + q * Ls)];
            twiddle_apply<R>(x, w1);
        }
        bfly<R>(x, dir);
#pragma unroll
        for (int q = 0; q < R; ++q) d[PAD(base + q * Ls)] = x[q];
    }
}
// the usual form: a table of the row's own complex type (the explicit table type keeps the __restrict__)
template <int R, class C>
AA_HD void dit_stage(C* d, int M, int L, int lsh, const C* __restrict__ tw, int dir, int t, int nt) {
    dit_stage<R, C, const C* __restrict__>(d, M, L, lsh, tw, dir, t, nt);
}

// the same two stages for one butterfly with the stage twiddle w1 = tw[j * tws] supplied by the caller (the specialised
// device path loads it once per row and keeps it: the DIF and the DIT stage of one level use the same entry)
// dev probes (device builds -DAA_PROBE_LDS2 / -DAA_PROBE_VALU2; results unchanged): the LDS reads, or the butterfly arithmetic, of the
// two wave-local radix-16 levels of the [R0,16,16] rows issued TWICE -- which unit the stage's time follows
#if defined(__HIP_DEVICE_COMPILE__) && defined(AA_PROBE_LDS2)
#define AA_PROBE_EXTRA_READS(R_, d_, base_, Ls_)                                                         \
    do {                                                                                                  \
        C x2_[R_];                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < R_; ++q_) x2_[q_] = d_[PAD((base_) + q_ * (Ls_))];        \
        _Pragma("unroll") for (int q_ = 0; q_ < R_; ++q_) asm volatile("" ::"v"(x2_[q_].re), "v"(x2_[q_].im)); \
    } while (0)
#else
#define AA_PROBE_EXTRA_READS(R_, d_, base_, Ls_) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(AA_PROBE_VALU2)
#define AA_PROBE_EXTRA_BFLY(R_, x_, dir_)                                                                 \
    do {                                                                                                  \
        C y2_[R_];                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < R_; ++q_) { y2_[q_] = x_[q_]; asm volatile("" : "+v"(y2_[q_].re), "+v"(y2_[q_].im)); } \
        bfly<R_>(y2_, dir_);                                                                              \
        _Pragma("unroll") for (int q_ = 0; q_ < R_; ++q_) asm volatile("" ::"v"(y2_[q_].re), "v"(y2_[q_].im)); \
    } while (0)
#else
#define AA_PROBE_EXTRA_BFLY(R_, x_, dir_) ((void)0)
#endif
template <int R, class C>
AA_HD void dif_butterfly_w(C* d, int base, int Ls, C w1, int dir) {
    C x[R];
    AA_PROBE_EXTRA_READS(R, d, base, Ls);
#pragma unroll
    for (int q = 0; q < R; ++q) x[q] = d[PAD(base + q * Ls)];
    AA_PROBE_EXTRA_BFLY(R, x, dir);
    bfly<R>(x, dir);
    d[PAD(base)] = x[0];
    if (dir < 0) w1.im = -w1.im;
    twiddle_apply<R>(x, w1);
#pragma unroll
    for (int q = 1; q < R; ++q) d[PAD(base + q * Ls)] = x[q];
}
template <int R, class C>
AA_HD void dit_butterfly_w(C* d, int base, int Ls, C w1, int dir) {
    C x[R];
    x[0] = d[PAD(base)];
    if (dir < 0) w1.im = -w1.im;
#pragma unroll
    for (int q = 1; q < R; ++q) x[q] = d[PAD(base + q * Ls)];
    AA_PROBE_EXTRA_READS(R, d, base, Ls);
    twiddle_apply<R>(x, w1);
    AA_PROBE_EXTRA_BFLY(R, x, dir);
    bfly<R>(x, dir);
#pragma unroll
    for (int q = 0; q < R; ++q) d[PAD(base + q * Ls)] = x[q];
}

// fused middle of the Bluestein convolution: the last DIF stage and the first DIT stage act on the same contiguous
// groups of R elements (L = R, no twiddles), so forward butterfly, filter multiply and inverse butterfly happen in
// registers with one LDS read and one LDS write.
template <int R>
AA_HD void bluestein_mid(cplx* d, int M, const cplx* __restrict__ bhat, int t, int nt) {
    const int nb = M / R;
    for (int b = t; b < nb; b += nt) {
        cplx x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = d[PAD(b * R + q)];
        bfly<R>(x, -1);
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = cmul(x[q], bhat[b * R + q]);
        bfly<R>(x, +1);
#pragma unroll
        for (int q = 0; q < R; ++q) d[PAD(b * R + q)] = x[q];
    }
}

// composite first radices of the specialised shapes: host only (planner, emulation); the generic device kernel is
// never given a shape that contains them
#if defined(__HIP_DEVICE_COMPILE__)
#define AA_RADIX_SWITCH_COMPOSITE(CALL)
#else
#define AA_RADIX_SWITCH_COMPOSITE(CALL)                  \
        case 6: { constexpr int RR = 6; CALL; } break;   \
        case 10: { constexpr int RR = 10; CALL; } break; \
        case 12: { constexpr int RR = 12; CALL; } break; \
        case 20: { constexpr int RR = 20; CALL; } break; \
        case 24: { constexpr int RR = 24; CALL; } break; \
        case 15: { constexpr int RR = 15; CALL; } break; \
        case 18: { constexpr int RR = 18; CALL; } break;
#endif
#define AA_RADIX_SWITCH(R, CALL)                         \
    switch (R) {                                         \
        case 2: { constexpr int RR = 2; CALL; } break;   \
        case 3: { constexpr int RR = 3; CALL; } break;   \
        case 4: { constexpr int RR = 4; CALL; } break;   \
        case 5: { constexpr int RR = 5; CALL; } break;   \
        case 8: { constexpr int RR = 8; CALL; } break;   \
        case 9: { constexpr int RR = 9; CALL; } break;   \
        case 16: { constexpr int RR = 16; CALL; } break; \
        AA_RADIX_SWITCH_COMPOSITE(CALL)                  \
    }

AA_HD void dif_stage_any(int R, cplx* d, int M, int L, int lsh, const cplx* __restrict__ tw, int dir, int t, int nt) {
    AA_RADIX_SWITCH(R, dif_stage<RR>(d, M, L, lsh, tw, dir, t, nt))
}
AA_HD void dit_stage_any(int R, cplx* d, int M, int L, int lsh, const cplx* __restrict__ tw, int dir, int t, int nt) {
    AA_RADIX_SWITCH(R, dit_stage<RR>(d, M, L, lsh, tw, dir, t, nt))
}
AA_HD void bluestein_mid_any(int R, cplx* d, int M, const cplx* __restrict__ bhat, int t, int nt) {
    AA_RADIX_SWITCH(R, bluestein_mid<RR>(d, M, bhat, t, nt))
}

// frequency held at position P after the full DIF (see header comment)
AA_HD int freq_of_pos(const FftShape& s, int P) {
    int k = 0, w = 1, rem = s.M;
    for (int i = 0; i < s.nstages; ++i) {
        rem /= s.radix[i];
        int q = P / rem;
        P -= q * rem;
        k += q * w;
        w *= s.radix[i];
    }
    return k;
}
// position that holds frequency k after the full DIF (inverse of freq_of_pos)
AA_HD int pos_of_freq(const FftShape& s, int k) {
    int P = 0, rem = s.M;
    for (int i = 0; i < s.nstages; ++i) {
        rem /= s.radix[i];
        int q = k % s.radix[i];
        k /= s.radix[i];
        P += q * rem;
    }
    return P;
}

// ---- c2r pre-processing: half-spectrum X[0..h] of a length n=2h real signal -> Z[0..h) such that
//      z = IDFT_h(Z) (unnormalised, sign +) gives y[2j] = Re z[j], y[2j+1] = Im z[j].
//      wn = exp(+2 pi i k / n).  A = X[k], B = conj(X[h-k]).
template <class C>
AA_HD C c2r_pre(C A, C B, C wn) {
    C s = cadd(A, B);
    C d = cmul(csub(A, B), wn);
    return C{s.re - d.im, s.im + d.re};  // s + i*d
}

// ---- one row (one latitude x one field) of the c2r transform, expressed as barrier-separated phases ----------
// The HIP kernel runs  for (ph...) { row_phase(ph, tid, nthreads, ...); __syncthreads(); }  with `work` in LDS;
// the host emulation (fft_plan.cpp: host_execute_row) loops t sequentially inside each phase.
// `Reader` supplies the input modes: rd(m) -> X[m] (raw; m <= mmax guaranteed by the caller of rd).
constexpr int MAX_PARTS = 16;  // max number of m-owners (multi-GPU m-sharding) the Fourier stage can gather from

struct RowTables {
    int n, h, method;        // method: 0 direct, 1 bluestein
    const FftShape* shape;   // stays in (global / host) memory: indexed at run time
    const cplx* tw;          // [M]
    const cplx* pre;         // [h]
    const cplx* chirp;       // [h] (bluestein)
    const cplx* bhat;        // [M] (bluestein)
};
struct RowOut {
    int mmax;                // highest non-zero mode, mmax <= h
    double* y;               // n reals
    int aligned16;           // the row starts on a pair boundary (16 bytes; 8 bytes for f32): pairs are stored whole
    int f32 = 0;             // y points to float (fp32 variant: fp32 in HBM, the FFT arithmetic stays fp64)
    double scale;            // 1/cos(lat) for the u,v fields of the vor/div path (TransLocal.cc:1443-1469), else 1
    // two-field fp32 form (fft_pair.h): lane x goes to y, lane y to (float*)y + pair_stride if pair_b is set
    long long pair_stride = 0;
    int pair_b            = 0;
    int pair_b_aligned    = 0;
};

struct alignas(8) fpair {
    float x, y;
};
// y[2k], y[2k+1] = z  for either output type
AA_HD void store_pair(const RowOut& io, int64_t k, cplx z) {
    if (io.f32) {
        float* yf = reinterpret_cast<float*>(io.y);
        if (io.aligned16) {
            *reinterpret_cast<fpair*>(yf + 2 * k) = fpair{(float)z.re, (float)z.im};
        }
        else {
            yf[2 * k]     = (float)z.re;
            yf[2 * k + 1] = (float)z.im;
        }
    }
    else if (io.aligned16) {
        *reinterpret_cast<cplx*>(io.y + 2 * k) = z;
    }
    else {
        io.y[2 * k]     = z.re;
        io.y[2 * k + 1] = z.im;
    }
}

// the same with the two (uniform) decisions taken by the caller, outside its element loop
template <bool F32, bool ALIGNED, class C>
AA_HD void store_pair_t(const RowOut& io, int64_t k, C z) {
    if (F32) {
        float* yf = reinterpret_cast<float*>(io.y);
        if (ALIGNED) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AA_FFT_PLAIN_STORE)
            typedef float f2_t __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store(f2_t{(float)z.re, (float)z.im}, reinterpret_cast<f2_t*>(yf + 2 * k));
#else
            *reinterpret_cast<fpair*>(yf + 2 * k) = fpair{(float)z.re, (float)z.im};
#endif
        }
        else {
            yf[2 * k]     = (float)z.re;
            yf[2 * k + 1] = (float)z.im;
        }
    }
    else if (ALIGNED) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AA_FFT_PLAIN_STORE)
        typedef double d2_t __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(d2_t{(double)z.re, (double)z.im}, reinterpret_cast<d2_t*>(io.y + 2 * k));
#else
        *reinterpret_cast<cplx*>(io.y + 2 * k) = cplx{(double)z.re, (double)z.im};
#endif
    }
    else {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AA_FFT_PLAIN_STORE)
        __builtin_nontemporal_store((double)z.re, io.y + 2 * k);
        __builtin_nontemporal_store((double)z.im, io.y + 2 * k + 1);
#else
        io.y[2 * k]     = (double)z.re;
        io.y[2 * k + 1] = (double)z.im;
#endif
    }
}
// calls fn(std::integral_constant<bool, f32>, std::integral_constant<bool, aligned>) for the row's flavour
template <class Fn>
AA_HD void with_store_flavour(const RowOut& io, Fn&& fn) {
    using T = std::true_type;
    using F = std::false_type;
    if (io.f32) {
        if (io.aligned16) fn(T{}, T{});
        else fn(T{}, F{});
    }
    else {
        if (io.aligned16) fn(F{}, T{});
        else fn(F{}, F{});
    }
}

AA_HD int row_num_phases(const RowTables& r) {
    // direct   : load | DIT stages (ns) | store
    // bluestein: load | DIF stages 0..ns-2 | fused [last DIF stage * filter * first DIT stage] | DIT stages (ns-1) | store
    const int ns = r.shape->nstages;
    return r.method == 1 ? 1 + (ns - 1) + 1 + (ns - 1) + 1 : 1 + ns + 1;
}

template <class Reader>
AA_HD cplx row_mode(const Reader& rd, int mmax, int m, int h) {
    // X[m] with the c2r conventions of the reference call site (TransLocal.cc:1166-1178):
    // imaginary part of m=0 is dropped; the Nyquist mode m=h contributes its real part only; modes above mmax are 0.
    // Branch-free (clamped address + select) so that the gathers of one phase can all be in flight together.
    const int mc = m > mmax ? (mmax < 0 ? 0 : mmax) : m;
    cplx v       = rd(mc);
    if (m > mmax || mmax < 0) {
        v.re = 0.;
        v.im = 0.;
    }
    if (m == 0 || m == h) {
        v.im = 0.;
    }
    return v;
}

// the two halves of row_mode() for callers that batch their loads: clamped mode index, then the masks
AA_HD int row_mode_index(int mmax, int m) {
    return m > mmax ? (mmax < 0 ? 0 : mmax) : m;
}
template <class C>
AA_HD C row_mode_mask(C v, int mmax, int m, int h) {
    if (m > mmax || mmax < 0) {
        v.re = 0;
        v.im = 0;
    }
    if (m == 0 || m == h) {
        v.im = 0;
    }
    return v;
}

// scheduling fence (device): instructions are not moved across it
#if defined(__HIP_DEVICE_COMPILE__)
#define AA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define AA_SCHED_FENCE() ((void)0)
#endif

AA_HD int stage_L(const FftShape& s, int i) {
    int L = s.M;
    for (int q = 0; q < i; ++q) L /= s.radix[q];
    return L;
}

template <class Reader>
AA_HD void row_phase(int ph, int t, int nt, const RowTables& r, const Reader& rd, const RowOut& io, cplx* work) {
    const int h  = r.h;
    const int M  = r.shape->M;
    const int ns = r.shape->nstages;
    if (ph == 0) {  // ---- load + c2r pre-processing (+ chirp / zero padding)
        // four elements per sweep, their loads (clamped addresses, masks applied afterwards) issued as one batch
        // ahead of a scheduling fence (see row_phase_ct)
        constexpr int NB = 4;
        for (int k0 = t; k0 < h; k0 += NB * nt) {
            cplx A[NB], B[NB], P[NB], C[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k  = k0 + i * nt;
                const int kc = k < h ? k : h - 1;
                A[i]         = rd(row_mode_index(io.mmax, kc));
                B[i]         = rd(row_mode_index(io.mmax, h - kc));
                P[i]         = r.pre[kc];
                C[i]         = r.method == 1 ? r.chirp[kc] : cplx{1., 0.};
            }
            AA_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k = k0 + i * nt;
                if (k < h) {
                    const cplx a = row_mode_mask(A[i], io.mmax, k, h);
                    const cplx b = cconj(row_mode_mask(B[i], io.mmax, h - k, h));
                    const cplx Z = c2r_pre(a, b, P[i]);
                    if (r.method == 1) {
                        work[PAD(k)] = cmul(Z, C[i]);
                    }
                    else {
                        work[PAD(pos_of_freq(*r.shape, k))] = Z;
                    }
                }
            }
        }
        if (r.method == 1) {
            for (int k = h + t; k < M; k += nt) {
                work[PAD(k)] = cplx{0., 0.};
            }
        }
        return;
    }
    ph -= 1;
    int dit_first = ns - 1;  // index of the first DIT stage still to run
    if (r.method == 1) {
        if (ph < ns - 1) {  // ---- forward DIF, stage ph
            dif_stage_any(r.shape->radix[ph], work, M, stage_L(*r.shape, ph), r.shape->lsh[ph], r.tw, -1, t, nt);
            return;
        }
        ph -= ns - 1;
        if (ph == 0) {  // ---- last DIF stage * filter spectrum * first DIT stage
            bluestein_mid_any(r.shape->radix[ns - 1], work, M, r.bhat, t, nt);
            return;
        }
        ph -= 1;
        dit_first = ns - 2;
    }
    if (ph <= dit_first) {  // ---- inverse DIT, stages in reverse order
        const int i = dit_first - ph;
        dit_stage_any(r.shape->radix[i], work, M, stage_L(*r.shape, i), r.shape->lsh[i], r.tw, +1, t, nt);
        return;
    }
    // ---- store: y[2j] = Re z[j], y[2j+1] = Im z[j]
    for (int j = t; j < h; j += nt) {
        cplx z = work[PAD(j)];
        if (r.method == 1) {
            z = cmul(z, r.chirp[j]);
        }
        store_pair(io, j, z);
    }
}


// ---- odd {3,5}-smooth row lengths (classic reduced Gaussian grids): complex DIT of length n on the Hermitian extension
//      Z[k] = X[k] (k <= mmax), conj(X[n - k]) (n - k <= mmax), 0 otherwise; y[j] = Re z[j].  Phases: load | DIT stages | store
AA_HD int row_num_phases_odd(const RowTables& r) {
    return 1 + r.shape->nstages + 1;
}
template <class Reader>
AA_HD void row_phase_odd(int ph, int t, int nt, const RowTables& r, const Reader& rd, const RowOut& io, cplx* work) {
    const int n  = r.n;
    const int ns = r.shape->nstages;
    if (ph == 0) {
        for (int k = t; k < n; k += nt) {
            cplx z = cplx{0., 0.};
            if (k <= io.mmax) {
                z = rd(k);
                if (k == 0) {
                    z.im = 0.;   // conventions of row_mode(): the imaginary part of the mean is dropped
                }
            }
            else if (n - k <= io.mmax) {
                z = cconj(rd(n - k));
            }
            work[PAD(pos_of_freq(*r.shape, k))] = z;
        }
        return;
    }
    ph -= 1;
    if (ph < ns) {
        const int i = ns - 1 - ph;
        dit_stage_any(r.shape->radix[i], work, n, stage_L(*r.shape, i), r.shape->lsh[i], r.tw, +1, t, nt);
        return;
    }
    for (int j = t; j < n; j += nt) {
        const double v = work[PAD(j)].re * io.scale;
        if (io.f32) {
            reinterpret_cast<float*>(io.y)[j] = (float)v;
        }
        else {
            io.y[j] = v;
        }
    }
}

// ======================================================================================================
// Compile-time specialised Bluestein rows: M = F * 2^K (F in {1,3,5}).  Same stage primitives as above, but the
// stage list is a template parameter so every stride / shift folds to a constant, the load + c2r pre-processing +
// chirp is fused into DIF stage 0 (whose inputs k >= h are the zero padding) and the chirp post-multiply + store is
// fused into the last DIT stage (whose outputs k >= h are not needed): 2*NS-1 phases, 2*NS-2 barriers, 4 LDS round
// trips for a 3-stage length instead of 6.  The filter spectrum is read through a [q][butterfly] transposed copy so
// that the lanes of one load instruction are contiguous.
// stage list of the specialised M = F * 2^K transform (DIF order); 0 past the end.
//   M = R0 * 256 with R0 = F * 2^(K-8) <= 24 :  [R0, 16, 16]   (R0 = 1: [16, 16]); F in {1, 3, 5, 9, 15}
//       one composite first stage (fused with the load; a 3/5-point DFT times a 2/4/8-point DFT in registers),
//       then R0 independent 256-point blocks: their forward stages, the filter multiply and their inverse stages
//       only move data between the 16 lanes that own the block (wave-local, no workgroup barrier)
//   320 = [20, 16], 384 = [24, 16], 640 = [10, 8, 8], 8192 = [16, 16, 16, 2]
AA_HD constexpr int ct_radix(int F, int K, int i) {
    if (F == 5 && K == 7) {
        return i == 0 ? 10 : (i < 3 ? 8 : 0);
    }
    if (K >= 8 && (F << (K - 8)) <= 24) {
        const int r0 = F << (K - 8);
        if (r0 == 1) return i < 2 ? 16 : 0;
        return i == 0 ? r0 : (i < 3 ? 16 : 0);
    }
    const int n16 = K / 4, rem = 1 << (K % 4);
    if (F > 1) {
        if (i == 0) return F * rem;
        return i - 1 < n16 ? 16 : 0;
    }
    if (i < n16) return 16;
    return (i == n16 && rem > 1) ? rem : 0;
}
AA_HD constexpr int ct_nstages(int F, int K) {
    int n = 0;
    while (ct_radix(F, K, n) != 0) ++n;
    return n;
}

template <int F_, int K_>
struct CtShape {
    static constexpr int F  = F_;
    static constexpr int K  = K_;
    static constexpr int M  = F_ << K_;
    static constexpr int NS = ct_nstages(F_, K_);
    // workers per row of the Bluestein kernel: one radix-16 butterfly each in the middle stages.  Never 5 or 6 wavefronts:
    // a workgroup is only launched onto a CU whose every SIMD has registers for ceil(waves / 4) of its wavefronts, so two
    // 5-wavefront workgroups of 168 registers never share a CU although their 10 wavefronts would fit (per-wavefront trace,
    // profiles/r03_fft_trace.txt: one workgroup resident 68-82 % of the time for M = 5120 / 4608, two for M = 4096 / 3840).
    // With 4 wavefronts the butterflies beyond 256 of the middle stages are a second round of the first wavefront(s).
    static constexpr int NT0 = (M / 16 + 63) / 64 * 64 < 64 ? 64 : ((M / 16 + 63) / 64 * 64 > 512 ? 512 : (M / 16 + 63) / 64 * 64);
#if defined(AA_FFT_KEEP_5WAVE)
    static constexpr int NT = NT0;
#else
    // M = 3072 = [12,16,16] (192 middle butterflies, 48 KiB: three workgroups per CU either way): 256 workers as well -- stage 0
    // and the last stage have 256 butterflies, one per worker, the fourth wavefront sits out the middle stages: 0.77 -> 0.70 ms
    // for the class.  Not for M = 2560 / 2304 (40 / 36 KiB: four workgroups of three wavefronts fit, only three of four):
    // 0.31 -> 0.33 ms (tools/fft_classes.sh).
    static constexpr int NT = (NT0 == 320 || NT0 == 384 || (M == 3072 && !AA_FFT_KEEP_192)) ? 256 : NT0;
#endif
    // wavefronts per SIMD the kernel is compiled for: the LDS footprint M * 16 allows two workgroups per CU above
    // 53 KiB (two wavefronts per SIMD at most: 256 registers), three below
    static constexpr int WPS = (M * 16 > 53248) ? 2 : 3;
    static constexpr int radix(int i) { return ct_radix(F_, K_, i); }
    static constexpr int L(int i) {
        int l = M;
        for (int q = 0; q < i; ++q) l /= radix(q);
        return l;
    }
    static constexpr int lsh(int i) {
        int ls = L(i) / radix(i), s = 0;
        while ((1 << s) < ls) ++s;
        return (1 << s) == ls ? s : -1;
    }
    // stages 1.. all have the same radix R and a block (length L(1)) is owned by L(1)/R <= 64 consecutive workers:
    // with a worker count that is a multiple of 64 the owners of a block are lanes of one wavefront in every one
    // of these stages, so the phases between the first and the last only need wavefront-level ordering
    static constexpr bool wave_local_middle() {
        if (NS < 3) return false;
        for (int i = 2; i < NS; ++i)
            if (radix(i) != radix(1)) return false;
        const int owners = L(1) / radix(1);
        return owners <= 64 && 64 % owners == 0;
    }
};

// Dev builds (-DAA_FFT_ABLATE): index multipliers that collapse one class of global accesses onto a single cache line
// (same instructions, no memory traffic), to attribute kernel time; 1 everywhere in normal builds.
#if defined(AA_FFT_ABLATE)
#define AA_ABL(r, bit) (((r).abl >> (bit)) & 1 ? 0 : 1)
#else
#define AA_ABL(r, bit) 1
#endif

// what a row reads its tables through: a pointer to its own complex type, except for the two-field fp32 form (fft_pair.h), whose
// table values are one float complex for both lanes
template <class C>
struct table_ptr_of {
    using type = const C*;
};
template <class C>
struct RowTablesCtT {
    using table_ptr = typename table_ptr_of<C>::type;
    int abl = 0;
    int n, h;
    table_ptr tw;       // [M]
    table_ptr pre;      // [h]
    table_ptr chirp;    // [h]
    table_ptr bhat_t;   // [R_last][M / R_last]  filter spectrum, transposed for the fused middle stage
};
using RowTablesCt = RowTablesCtT<cplx>;

template <class S, int I, class Fn>
AA_HD void ct_stage_dispatch(int i, Fn&& fn) {
    if constexpr (I < S::NS) {
        if (i == I) {
            fn(std::integral_constant<int, I>{});
        }
        else {
            ct_stage_dispatch<S, I + 1>(i, fn);
        }
    }
}

template <class S>
AA_HD constexpr int row_num_phases_ct() {
    return 2 * S::NS - 1;
}

// ---- the row's kept modes are fetched ONCE per row into an LDS staging area `raw` before phase 0, which needs every
//      mode twice (X[k] and X[h-k]).  Device: fft_kernel.hip (LDS-DMA gather); host emulation: plain copy.
// element k of the staging area (the two-field fp32 form stores the intermediate's own order there: fft_pair.h overloads both)
template <class C>
AA_HD C raw_elem(const C* raw, int k) {
    return raw[k];
}
// (RAW: what the staging area is read through -- a pointer, or StridedRaw for rows that share a wavefront)
template <class C>
struct StridedRaw {   // several fields of one row staged side by side: mode m of field `off` is p[m * stride + off]
    const C* p;
    int stride, off;
    AA_HD C operator[](int m) const { return p[m * stride + off]; }
};
template <class C>
AA_HD C ct_raw_mode(StridedRaw<C> raw, int mmax, int m, int h) {
    C v = m <= mmax ? raw[m] : C{0, 0};
    if (m == 0 || m == h) {
        v.im = 0;   // conventions of row_mode()
    }
    return v;
}
template <class C>
AA_HD C ct_raw_mode(const C* raw, int mmax, int m, int h) {
    C v = m <= mmax ? raw[m] : C{0, 0};
    if (m == 0 || m == h) {
        v.im = 0;   // conventions of row_mode()
    }
    return v;
}

// phase 0 of a specialised Bluestein row, butterfly b: c2r pre-processing + chirp + DIF stage 0 (L = M, one block) in
// registers.  M >= 2h-1 and M even: h <= M/2, so the inputs q >= NZ = ceil(R0/2) are zero padding for every b.  The
// table loads are issued in batches of NB elements ahead of a scheduling fence: left alone, the compiler serialises
// them one round trip at a time to save registers.
template <class S, class C, class RAW>
AA_HD void ct_phase0_compute(int b, const RowTablesCtT<C>& r, RAW raw, const RowOut& io, C* x) {
    constexpr int M   = S::M;
    constexpr int R0  = S::radix(0);
    constexpr int Ls0 = M / R0;
    constexpr int NZ  = (R0 + 1) / 2;
    constexpr int NB  = NZ <= 5 ? NZ : (NZ % 5 == 0 ? 5 : (NZ % 4 == 0 ? 4 : (NZ % 3 == 0 ? 3 : 2)));
    const int h       = r.h;
    C w1              = r.tw[b];
#pragma unroll
    for (int q0 = 0; q0 < NZ; q0 += NB) {
        C P[NB], Ch[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (q0 + i < NZ) {
                const int k  = b + (q0 + i) * Ls0;
                const int kc = k < h ? k : h - 1;
                P[i]         = r.pre[kc * AA_ABL(r, 1)];
                Ch[i]        = r.chirp[kc * AA_ABL(r, 1)];
            }
        }
        AA_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (q0 + i < NZ) {
                const int k  = b + (q0 + i) * Ls0;
                const int kc = k < h ? k : h - 1;
                const C a = ct_raw_mode(raw, io.mmax, kc, h);
                const C c = cconj(ct_raw_mode(raw, io.mmax, h - kc, h));
                const C z = cmul(c2r_pre(a, c, P[i]), Ch[i]);
                x[q0 + i] = k < h ? z : C{0, 0};
            }
        }
    }
#pragma unroll
    for (int q = NZ; q < R0; ++q) x[q] = C{0, 0};
    bfly<R0>(x, -1);
    w1.im = -w1.im;
    twiddle_apply<R0>(x, w1);
}

// RAW_ALIASES_WORK: the staging area lives inside `work` (device: LDS is the scarce resource), so every worker
// finishes reading it before anybody writes stage-0 results; needs nt == S::NT.
// NTW: workers of one row (S::NT; fewer where several short rows share a wavefront: fft_kernel.hip, coarse classes)
template <class S, bool RAW_ALIASES_WORK, int NTW = S::NT, class C, class RAW>
AA_HD void row_phase_ct(int ph, int t, int nt, const RowTablesCtT<C>& r, RAW raw, const RowOut& io,
                        C* work) {
    using Real        = typename C::real;
    constexpr int M   = S::M;
    constexpr int NS  = S::NS;
    constexpr int R0  = S::radix(0);
    constexpr int Ls0 = M / R0;
    constexpr int RL  = S::radix(NS - 1);
    const int h       = r.h;
    if (ph == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (RAW_ALIASES_WORK) {
            constexpr int NBUT = (Ls0 + NTW - 1) / NTW;
            C x[NBUT][R0];
#pragma unroll
            for (int ib = 0; ib < NBUT; ++ib) {
                const int b = t + ib * NTW;
                if (b < Ls0) {
                    ct_phase0_compute<S>(b, r, raw, io, x[ib]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int ib = 0; ib < NBUT; ++ib) {
                const int b = t + ib * NTW;
                if (b < Ls0) {
#pragma unroll
                    for (int q = 0; q < R0; ++q) work[PAD(b + q * Ls0)] = x[ib][q];
                }
            }
            return;
        }
#endif
        for (int b = t; b < Ls0; b += nt) {
            C x[R0];
            ct_phase0_compute<S>(b, r, raw, io, x);
#pragma unroll
            for (int q = 0; q < R0; ++q) work[PAD(b + q * Ls0)] = x[q];
        }
        return;
    }
    if (ph < NS - 1) {  // ---- forward DIF stages 1 .. NS-2
        ct_stage_dispatch<S, 1>(ph, [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I >= 1 && I <= NS - 2) {
                dif_stage<S::radix(I)>(work, M, S::L(I), S::lsh(I), r.tw, -1, t, nt);
            }
        });
        return;
    }
    if (ph == NS - 1) {  // ---- fused middle: last DIF stage * filter * first DIT stage (L = RL, no twiddles)
        constexpr int nb = M / RL;
        for (int b = t; b < nb; b += nt) {
            C f[RL];  // filter spectrum: all loads in flight before the LDS reads (see phase 0)
#pragma unroll
            for (int q = 0; q < RL; ++q) f[q] = r.bhat_t[(q * nb + b) * AA_ABL(r, 2)];
            AA_SCHED_FENCE();
            C x[RL];
#pragma unroll
            for (int q = 0; q < RL; ++q) x[q] = work[PAD(b * RL + q)];
            bfly<RL>(x, -1);
#pragma unroll
            for (int q = 0; q < RL; ++q) x[q] = cmul(x[q], f[q]);
            bfly<RL>(x, +1);
#pragma unroll
            for (int q = 0; q < RL; ++q) work[PAD(b * RL + q)] = x[q];
        }
        return;
    }
    if (ph < 2 * NS - 2) {  // ---- inverse DIT stages NS-2 .. 1
        const int i = 2 * NS - 2 - ph;  // ph = NS -> i = NS-2 ; ph = 2NS-3 -> i = 1
        ct_stage_dispatch<S, 1>(i, [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I >= 1 && I <= NS - 2) {
                dit_stage<S::radix(I)>(work, M, S::L(I), S::lsh(I), r.tw, +1, t, nt);
            }
        });
        return;
    }
    // ---- fused: DIT stage 0 + chirp + store (outputs q >= NZ are padding: their butterfly arithmetic is dead)
    constexpr int NZ = (R0 + 1) / 2;
    for (int b = t; b < Ls0; b += nt) {
        C w1 = r.tw[b];
        C c[NZ];
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            const int k = b + q * Ls0;
            c[q]        = r.chirp[(k < h ? k : h - 1) * AA_ABL(r, 3)];
        }
        AA_SCHED_FENCE();
        C x[R0];
#pragma unroll
        for (int q = 0; q < R0; ++q) x[q] = work[PAD(b + q * Ls0)];
        twiddle_apply<R0>(x, w1);
        bfly<R0>(x, +1);
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            x[q]    = cmul(x[q], c[q]);
            x[q].re = x[q].re * (Real)io.scale;  // 1/cos(lat) for the wind fields, exactly 1 otherwise
            x[q].im = x[q].im * (Real)io.scale;
        }
        with_store_flavour(io, [&](auto f32c, auto alc) {
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int k = b + q * Ls0;
                if (k < h) {
                    store_pair_t<decltype(f32c)::value, decltype(alc)::value>(io, (int64_t)k * AA_ABL(r, 4), x[q]);
                }
            }
        });
    }
}

// ======================================================================================================
// Compile-time specialised DIRECT rows: h = n/2 = M = F * 2^K itself is a length of the specialised family (regular
// Gaussian / lon-lat grids: every row; reduced grids: their smooth rows).  One inverse DIT, NS phases:
//   phase 0      : load + c2r pre-processing fused with the first DIT stage (the last radix of the DIF list, L = RL, no
//                  twiddles).  Worker b' loads the modes k = b' + q * (M / RL), q < RL (coalesced over b'); they are the
//                  frequencies held by positions b * RL + q of butterfly b, b = digit reversal of b'.
//   phases 1..   : DIT stages NS-2 .. 1 in LDS
//   last phase   : DIT stage 0 fused with the store of y[2k], y[2k+1], k = b + q * Ls0 (coalesced over b)
template <class S>
AA_HD constexpr int row_num_phases_dct() {
    return S::NS;
}

// the stage list of S reversed: for the direct rows the load is fused with the LAST radix of the list and the store
// with the first, and it is the load phase that is short of registers (3 loads per element in flight), so the small
// composite radix goes there: [16, 16, R0]
template <class S>
struct CtShapeRev {
    static constexpr int F  = S::F;
    static constexpr int K  = S::K;
    static constexpr int M  = S::M;
    static constexpr int NS = S::NS;
    static constexpr int radix(int i) { return S::radix(NS - 1 - i); }
    static constexpr int L(int i) {
        int l = M;
        for (int q = 0; q < i; ++q) l /= radix(q);
        return l;
    }
    static constexpr int lsh(int i) {
        int ls = L(i) / radix(i), s = 0;
        while ((1 << s) < ls) ++s;
        return (1 << s) == ls ? s : -1;
    }
};

// butterfly index of the first DIT stage whose positions b*RL + q hold the frequencies b' + q * M/RL
template <class S0>
AA_HD int dct_first_butterfly(int bp) {
    using S = CtShapeRev<S0>;
    int b = 0;
#pragma unroll
    for (int i = 0; i < S::NS - 1; ++i) {   // b' = sum_i d_i * (R_0 ... R_{i-1});  b = sum_i d_i * (R_{i+1} ... R_{NS-2})
        const int d = bp % S::radix(i);
        bp /= S::radix(i);
        int w = 1;
#pragma unroll
        for (int j = i + 1; j < S::NS - 1; ++j) w *= S::radix(j);
        b += d * w;
    }
    return b;
}

template <class S0, class Reader, class C>
AA_HD void row_phase_dct(int ph, int t, int nt, const RowTablesCtT<C>& r, const Reader& rd, const RowOut& io,
                         C* work) {
    using Real        = typename C::real;
    using S           = CtShapeRev<S0>;
    constexpr int M   = S::M;
    constexpr int NS  = S::NS;
    constexpr int R0  = S::radix(0);
    constexpr int Ls0 = M / R0;
    constexpr int RL  = S::radix(NS - 1);
    constexpr int nbl = M / RL;
    const int h       = r.h;  // == M
    if (ph == 0) {
        constexpr int NB = RL >= 16 ? 2 : (RL % 4 == 0 ? 4 : (RL % 5 == 0 ? 5 : (RL % 3 == 0 ? 3 : (RL % 2 == 0 ? 2 : 1))));
        for (int bp = t; bp < nbl; bp += nt) {
            C x[RL];
#pragma unroll
            for (int q0 = 0; q0 < RL; q0 += NB) {
                C A[NB], B[NB], P[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int k = bp + (q0 + i) * nbl;
                    const cplx a0 = rd(row_mode_index(io.mmax, k));       // (a float -> double -> float round trip of the fp32
                    const cplx b0 = rd(row_mode_index(io.mmax, h - k));   //  form is folded away by the compiler)
                    A[i]          = C{(Real)a0.re, (Real)a0.im};
                    B[i]          = C{(Real)b0.re, (Real)b0.im};
                    P[i]        = r.pre[k];
                }
                AA_SCHED_FENCE();
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int k  = bp + (q0 + i) * nbl;
                    const C a = row_mode_mask(A[i], io.mmax, k, h);
                    const C c = cconj(row_mode_mask(B[i], io.mmax, h - k, h));
                    x[q0 + i]    = c2r_pre(a, c, P[i]);
                }
            }
            bfly<RL>(x, +1);
            const int b = dct_first_butterfly<S0>(bp);
#pragma unroll
            for (int q = 0; q < RL; ++q) work[PAD(b * RL + q)] = x[q];
        }
        return;
    }
    if (ph < NS - 1) {  // DIT stages NS-2 .. 1
        const int i = NS - 1 - ph;
        ct_stage_dispatch<S, 1>(i, [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I >= 1 && I <= NS - 2) {
                dit_stage<S::radix(I)>(work, M, S::L(I), S::lsh(I), r.tw, +1, t, nt);
            }
        });
        return;
    }
    // ---- DIT stage 0 + store
    for (int b = t; b < Ls0; b += nt) {
        const C w1 = r.tw[b];
        C x[R0];
#pragma unroll
        for (int q = 0; q < R0; ++q) x[q] = work[PAD(b + q * Ls0)];
        twiddle_apply<R0>(x, w1);
        bfly<R0>(x, +1);
        with_store_flavour(io, [&](auto f32c, auto alc) {
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                store_pair_t<decltype(f32c)::value, decltype(alc)::value>(
                    io, b + q * Ls0, C{x[q].re * (Real)io.scale, x[q].im * (Real)io.scale});
            }
        });
    }
}

// the (F, K) instances that exist (kernel and host emulation use the same list)
AA_HD constexpr bool ct_supported(int f, int k) {
    return (f == 1 && k >= 8 && k <= 13) || (f == 3 && k >= 7 && k <= 11) || (f == 5 && k >= 6 && k <= 10) ||
           (f == 9 && k >= 8 && k <= 9) || (f == 15 && k == 8);   // 2304, 4608, 3840: tighter Bluestein lengths
}
#define AA_CT_CASE(FF, KK, CALL)                         \
    if (ctf == FF && ctk == KK) {                        \
        using S = ::atlas_amd::fft::CtShape<FF, KK>;     \
        CALL;                                            \
    }
#define AA_CT_DISPATCH(ctf, ctk, CALL)                                                                          \
    AA_CT_CASE(1, 8, CALL) AA_CT_CASE(1, 9, CALL) AA_CT_CASE(1, 10, CALL) AA_CT_CASE(1, 11, CALL)               \
    AA_CT_CASE(1, 12, CALL) AA_CT_CASE(1, 13, CALL) AA_CT_CASE(3, 7, CALL) AA_CT_CASE(3, 8, CALL)               \
    AA_CT_CASE(3, 9, CALL) AA_CT_CASE(3, 10, CALL) AA_CT_CASE(3, 11, CALL) AA_CT_CASE(5, 6, CALL)               \
    AA_CT_CASE(5, 7, CALL) AA_CT_CASE(5, 8, CALL) AA_CT_CASE(5, 9, CALL) AA_CT_CASE(5, 10, CALL)               \
    AA_CT_CASE(9, 8, CALL) AA_CT_CASE(9, 9, CALL) AA_CT_CASE(15, 8, CALL)

// (the HYBRID rows of round 2 -- a dense radix-A stage on the matrix cores -- lost against Bluestein and live in
// tools/experiments/fft_hybrid_core.h; only an experiments build plans, emulates or launches them)
constexpr int HYB_MAX_A = 257;
constexpr int HYB_UPW   = 2;   // dense-stage output tiles (16 x 16) a wavefront accumulates per round

}  // namespace fft
}  // namespace atlas_amd
