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
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define AA_HD __host__ __device__ __forceinline__
#else
#define AA_HD inline
#endif

namespace atlas_amd {
namespace fft {

struct cplx {
    double re, im;
};

AA_HD cplx cmul(cplx a, cplx b) {
    return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
AA_HD cplx cmulc(cplx a, cplx b) {  // a * conj(b)
    return cplx{a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
AA_HD cplx cadd(cplx a, cplx b) {
    return cplx{a.re + b.re, a.im + b.im};
}
AA_HD cplx csub(cplx a, cplx b) {
    return cplx{a.re - b.re, a.im - b.im};
}
AA_HD cplx cconj(cplx a) {
    return cplx{a.re, -a.im};
}
// multiply by +i (dir=+1) or -i (dir=-1)
AA_HD cplx cmuli(cplx a, int dir) {
    return dir > 0 ? cplx{-a.im, a.re} : cplx{a.im, -a.re};
}

constexpr int MAX_STAGES = 10;

struct FftShape {          // stage list of an M-point transform
    int M;
    int nstages;
    int radix[MAX_STAGES];  // DIF order
};

// twiddle w_M^t = exp(+2 pi i t / M) from a table of M entries; dir=-1 conjugates
AA_HD cplx twiddle(const cplx* __restrict__ tw, int t, int dir) {
    cplx w = tw[t];
    if (dir < 0) {
        w.im = -w.im;
    }
    return w;
}

// ---- radix butterflies: y_q = sum_p x_p exp(dir 2 pi i p q / r) ------------------------------------------------
AA_HD void bfly2(cplx* x) {
    cplx a = x[0], b = x[1];
    x[0]   = cadd(a, b);
    x[1]   = csub(a, b);
}
AA_HD void bfly4(cplx* x, int dir) {
    cplx a = cadd(x[0], x[2]), b = csub(x[0], x[2]);
    cplx c = cadd(x[1], x[3]), d = cmuli(csub(x[1], x[3]), dir);
    x[0]   = cadd(a, c);
    x[1]   = cadd(b, d);
    x[2]   = csub(a, c);
    x[3]   = csub(b, d);
}
AA_HD void bfly3(cplx* x, int dir) {
    const double s = 0.86602540378443864676372317075294 * dir;  // sin(2pi/3)
    cplx t1 = cadd(x[1], x[2]);
    cplx t2 = cplx{x[0].re - 0.5 * t1.re, x[0].im - 0.5 * t1.im};
    cplx t3 = csub(x[1], x[2]);
    cplx t4 = cplx{-s * t3.im, s * t3.re};  // i*s*t3
    x[0]    = cadd(x[0], t1);
    x[1]    = cadd(t2, t4);
    x[2]    = csub(t2, t4);
}
AA_HD void bfly5(cplx* x, int dir) {
    const double c1 = 0.30901699437494742410229341718282;   // cos(2pi/5)
    const double c2 = -0.80901699437494742410229341718282;  // cos(4pi/5)
    const double s1 = 0.95105651629515357211643933337938 * dir;  // sin(2pi/5)
    const double s2 = 0.58778525229247312916870595463907 * dir;  // sin(4pi/5)
    cplx a1 = cadd(x[1], x[4]), b1 = csub(x[1], x[4]);
    cplx a2 = cadd(x[2], x[3]), b2 = csub(x[2], x[3]);
    cplx x0 = x[0];
    x[0]    = cplx{x0.re + a1.re + a2.re, x0.im + a1.im + a2.im};
    cplx m1 = cplx{x0.re + c1 * a1.re + c2 * a2.re, x0.im + c1 * a1.im + c2 * a2.im};
    cplx m2 = cplx{x0.re + c2 * a1.re + c1 * a2.re, x0.im + c2 * a1.im + c1 * a2.im};
    // i*(s1*b1 + s2*b2) and i*(s2*b1 - s1*b2)
    cplx n1 = cplx{-(s1 * b1.im + s2 * b2.im), s1 * b1.re + s2 * b2.re};
    cplx n2 = cplx{-(s2 * b1.im - s1 * b2.im), s2 * b1.re - s1 * b2.re};
    x[1]    = cadd(m1, n1);
    x[4]    = csub(m1, n1);
    x[2]    = cadd(m2, n2);
    x[3]    = csub(m2, n2);
}
template <int R>
AA_HD void bfly(cplx* x, int dir) {
    if (R == 2) bfly2(x);
    else if (R == 3) bfly3(x, dir);
    else if (R == 4) bfly4(x, dir);
    else if (R == 5) bfly5(x, dir);
}

// ---- one DIF stage: blocks of length L, radix R; twiddle table of M entries (w_L^t = w_M^{t*M/L}) --------------
template <int R>
AA_HD void dif_stage(cplx* d, int M, int L, const cplx* __restrict__ tw, int dir, int t, int nt) {
    const int Ls  = L / R;        // sub-block length
    const int tws = M / L;        // twiddle stride
    const int nb  = M / R;        // butterflies
    for (int b = t; b < nb; b += nt) {
        const int blk = b / Ls;
        const int j   = b - blk * Ls;
        cplx* p       = d + blk * L + j;
        cplx x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = p[q * Ls];
        bfly<R>(x, dir);
        p[0] = x[0];
#pragma unroll
        for (int q = 1; q < R; ++q) p[q * Ls] = cmul(x[q], twiddle(tw, j * q * tws, dir));
    }
}
// ---- one DIT stage (inverse of the DIF stage with the same L, R): twiddle first, then butterfly --------------
template <int R>
AA_HD void dit_stage(cplx* d, int M, int L, const cplx* __restrict__ tw, int dir, int t, int nt) {
    const int Ls  = L / R;
    const int tws = M / L;
    const int nb  = M / R;
    for (int b = t; b < nb; b += nt) {
        const int blk = b / Ls;
        const int j   = b - blk * Ls;
        cplx* p       = d + blk * L + j;
        cplx x[R];
        x[0] = p[0];
#pragma unroll
        for (int q = 1; q < R; ++q) x[q] = cmul(p[q * Ls], twiddle(tw, j * q * tws, dir));
        bfly<R>(x, dir);
#pragma unroll
        for (int q = 0; q < R; ++q) p[q * Ls] = x[q];
    }
}

AA_HD void dif_stage_any(int R, cplx* d, int M, int L, const cplx* __restrict__ tw, int dir, int t, int nt) {
    switch (R) {
        case 2: dif_stage<2>(d, M, L, tw, dir, t, nt); break;
        case 3: dif_stage<3>(d, M, L, tw, dir, t, nt); break;
        case 4: dif_stage<4>(d, M, L, tw, dir, t, nt); break;
        case 5: dif_stage<5>(d, M, L, tw, dir, t, nt); break;
    }
}
AA_HD void dit_stage_any(int R, cplx* d, int M, int L, const cplx* __restrict__ tw, int dir, int t, int nt) {
    switch (R) {
        case 2: dit_stage<2>(d, M, L, tw, dir, t, nt); break;
        case 3: dit_stage<3>(d, M, L, tw, dir, t, nt); break;
        case 4: dit_stage<4>(d, M, L, tw, dir, t, nt); break;
        case 5: dit_stage<5>(d, M, L, tw, dir, t, nt); break;
    }
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
AA_HD cplx c2r_pre(cplx A, cplx B, cplx wn) {
    cplx s = cadd(A, B);
    cplx d = cmul(csub(A, B), wn);
    return cplx{s.re - d.im, s.im + d.re};  // s + i*d
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
    int aligned16;           // y is 16-byte aligned
};

AA_HD int row_num_phases(const RowTables& r) {
    // load | [DIF stages | pointwise] | DIT stages | store
    const int ns = r.shape->nstages;
    return 1 + (r.method == 1 ? ns + 1 : 0) + ns + 1;
}

template <class Reader>
AA_HD cplx row_mode(const Reader& rd, int mmax, int m, int h) {
    // X[m] with the c2r conventions of the reference call site (TransLocal.cc:1166-1178):
    // imaginary part of m=0 is dropped; the Nyquist mode m=h contributes its real part only.
    if (m > mmax) {
        return cplx{0., 0.};
    }
    cplx v = rd(m);
    if (m == 0 || m == h) {
        v.im = 0.;
    }
    return v;
}

template <class Reader>
AA_HD void row_phase(int ph, int t, int nt, const RowTables& r, const Reader& rd, const RowOut& io, cplx* work) {
    const int h  = r.h;
    const int M  = r.shape->M;
    const int ns = r.shape->nstages;
    if (ph == 0) {  // ---- load + c2r pre-processing (+ chirp / zero padding)
        for (int k = t; k < h; k += nt) {
            cplx A = row_mode(rd, io.mmax, k, h);
            cplx B = cconj(row_mode(rd, io.mmax, h - k, h));
            cplx Z = c2r_pre(A, B, r.pre[k]);
            if (r.method == 1) {
                work[k] = cmul(Z, r.chirp[k]);
            }
            else {
                work[pos_of_freq(*r.shape, k)] = Z;
            }
        }
        if (r.method == 1) {
            for (int k = h + t; k < M; k += nt) {
                work[k] = cplx{0., 0.};
            }
        }
        return;
    }
    ph -= 1;
    if (r.method == 1) {
        if (ph < ns) {  // ---- forward DIF, stage ph
            int L = M;
            for (int i = 0; i < ph; ++i) L /= r.shape->radix[i];
            dif_stage_any(r.shape->radix[ph], work, M, L, r.tw, -1, t, nt);
            return;
        }
        ph -= ns;
        if (ph == 0) {  // ---- pointwise multiply with the filter spectrum (same permuted order)
            for (int p = t; p < M; p += nt) {
                work[p] = cmul(work[p], r.bhat[p]);
            }
            return;
        }
        ph -= 1;
    }
    if (ph < ns) {  // ---- inverse DIT, stages in reverse order
        const int i = ns - 1 - ph;
        int L       = M;
        for (int q = 0; q < i; ++q) L /= r.shape->radix[q];
        dit_stage_any(r.shape->radix[i], work, M, L, r.tw, +1, t, nt);
        return;
    }
    // ---- store: y[2j] = Re z[j], y[2j+1] = Im z[j]
    for (int j = t; j < h; j += nt) {
        cplx z = work[j];
        if (r.method == 1) {
            z = cmul(z, r.chirp[j]);
        }
        if (io.aligned16) {
            *reinterpret_cast<cplx*>(io.y + 2 * (int64_t)j) = z;
        }
        else {
            io.y[2 * (int64_t)j]     = z.re;
            io.y[2 * (int64_t)j + 1] = z.im;
        }
    }
}

}  // namespace fft
}  // namespace atlas_amd
