/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the atlas::trans::TransLocal inverse spherical-harmonics
 * transform (global structured Gaussian / lon-lat grids).
 *
 * This file is a plain-C restatement of the reference ALGORITHM (not its code).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP path against
 * an independent implementation.  Nothing under atlas_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line (ecmwf/atlas 0.44.1) whose behaviour it follows.
 * The oracle is pinned by tests/test_oracle_*.py against the reference's own known-answer tests
 * (analytic spherical harmonics, src/tests/trans/test_transgeneral.cc:80-374,433-449,472-489) and against
 * independent mpmath / numpy.fft (pocketfft) evaluations -- see oracle/README.md.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC translocal_oracle.c -o liboracle.so -lm
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_DEG2RAD (M_PI / 180.)
#define ORC_PIL 3.14159265358979323846264338327950288L
/* TransLocal.cc:49 : latitudes are clamped to +-89.9999999 deg before use */
#define ORC_LATPOLE 89.9999999

/* ------------------------------------------------------------------------------------------------
 * a1  fourier_truncation   (TransLocal.cc:272-300)
 * ---------------------------------------------------------------------------------------------- */
int orc_fourier_truncation(int truncation, int nx, int nxmax, int ndgl, double lat_rad, int fullgrid) {
    (void)nxmax; /* unused in the reference as well (:274) */
    int trc     = truncation;
    int trclin  = ndgl - 1;
    int trcquad = ndgl * 2 / 3 - 1;
    if (truncation >= trclin || fullgrid) {
        trc = (nx - 1) / 2; /* linear */
    }
    else if (truncation >= trcquad) {
        /* quadratic: NB the weight is an INTEGER division in the reference (:287) */
        double weight = (double)(3 * (trclin - truncation) / ndgl);
        double c      = cos(lat_rad);
        double sqcos  = pow(c, 2);
        trc           = (int)((nx - 1) / (2 + weight * sqcos));
    }
    else {
        double c     = cos(lat_rad);
        double sqcos = pow(c, 2);
        trc          = (int)((nx - 1) / (2 + sqcos) - 1); /* cubic */
    }
    return trc < truncation ? trc : truncation;
}

/* num_n (TransLocal.cc:183-187): number of total wavenumbers n in [m, trc] with (n-m) even / odd */
size_t orc_num_n(int trc, int m, int symmetric) {
    int len = (trc - m + (symmetric ? 2 : 1)) / 2;
    return (size_t)(len < 0 ? 0 : len);
}

/* add_padding (TransLocal.cc:236-238): round up to a multiple of 8 doubles */
static size_t orc_pad8(size_t n) {
    return (size_t)(ceil(n / 8.)) * 8;
}

/* ------------------------------------------------------------------------------------------------
 * a4  compute_zfn  (LegendrePolynomials.cc:24-45)  Fourier coefficients of the ordinary Legendre
 *     polynomials (Belousov), IFS normalisation 0.5*int(P^2)=1.   zfn is (trc+1) x (trc+1), row n.
 * ---------------------------------------------------------------------------------------------- */
void orc_compute_zfn(int trc, double* zfn) {
    const size_t ld = (size_t)trc + 1;
    zfn[0]          = 2.;
    for (int n = 1; n <= trc; ++n) {
        double v = zfn[0];
        for (int j = 1; j <= n; ++j) {
            v *= sqrt(1. - 0.25 / ((double)j * (double)j));
        }
        zfn[(size_t)n * ld + n] = v;
        int odd                 = n % 2;
        for (int j = 2; j <= n - odd; j += 2) {
            double num                    = (j - 1.) * (2. * n - j + 2.);
            double den                    = j * (2. * n - j + 1.);
            zfn[(size_t)n * ld + (n - j)] = zfn[(size_t)n * ld + (n - j + 2)] * num / den;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a5  compute_legendre_polynomials_lat (LegendrePolynomials.cc:47-151)
 *     legpol is the packed triangle: index(m,n) = (2*trc+3-m)*m/2 + n-m, 0<=m<=n<=trc.
 *     NB: like the reference (:102) this zeroes zfn(n,0) for odd n as a side effect.
 * ---------------------------------------------------------------------------------------------- */
static inline size_t orc_idxmn(int trc, int m, int n) {
    return (size_t)(2 * trc + 3 - m) * (size_t)m / 2 + (size_t)(n - m);
}

void orc_legendre_lat(int trc, double lat_rad, double* legpol, double* zfn, double* vsin, double* vcos) {
    const size_t ld      = (size_t)trc + 1;
    double theta         = M_PI_2 - lat_rad;
    double x             = cos(theta);
    volatile double sint = sqrt(1. - x * x); /* :61 sin(theta) computed like the IFS trans library */

    legpol[orc_idxmn(trc, 0, 0)] = 1.;
    for (int j = 1; j <= trc; ++j) {
        vsin[j] = sin(j * theta);
        vcos[j] = cos(j * theta);
    }
    double inv_sint = 0.;
    if (fabs(sint) <= sqrt(DBL_EPSILON)) { /* :72 closer than ~1 m to the pole */
        x    = 1.;
        sint = 0.;
    }
    else {
        inv_sint = 1. / sint;
    }
    /* m = 0 and m = 1 columns from the cos / sin series (:85-115) */
    for (int n = 2; n <= trc; n += 2) {
        double p0 = 0.5 * zfn[(size_t)n * ld + 0];
        double p1 = 0.;
        double sq = 1. / sqrt(n * (n + 1.));
        for (int k = 2; k <= n; k += 2) {
            p0 = p0 + zfn[(size_t)n * ld + k] * vcos[k];
            p1 = p1 + sq * zfn[(size_t)n * ld + k] * k * vsin[k];
        }
        legpol[orc_idxmn(trc, 0, n)] = p0;
        legpol[orc_idxmn(trc, 1, n)] = p1;
    }
    for (int n = 1; n <= trc; n += 2) {
        zfn[(size_t)n * ld + 0] = 0.;
        double p0               = 0.;
        double p1               = 0.;
        double sq               = 1. / sqrt(n * (n + 1.));
        for (int k = 1; k <= n; k += 2) {
            p0 = p0 + zfn[(size_t)n * ld + k] * vcos[k];
            p1 = p1 + sq * zfn[(size_t)n * ld + k] * k * vsin[k];
        }
        legpol[orc_idxmn(trc, 0, n)] = p0;
        legpol[orc_idxmn(trc, 1, n)] = p1;
    }
    /* diagonal, Belousov (23) with underflow flush (:122-130) */
    double tiny = inv_sint * DBL_MIN;
    for (int n = 2; n <= trc; ++n) {
        double sq = sqrt((2. * n + 1.) / (2. * n));
        double v  = legpol[orc_idxmn(trc, n - 1, n - 1)] * sint * sq;
        if (fabs(v) < tiny) {
            v = 0.0;
        }
        legpol[orc_idxmn(trc, n, n)] = v;
    }
    /* general recurrence, Belousov (17) (:136-149) */
    for (int n = 3; n <= trc; ++n) {
        for (int m = 2; m < n; ++m) {
            double cn = ((2. * n + 1.) * (n + m - 3.) * (n + m - 1.));
            double cd = ((2. * n - 3.) * (n + m - 2.) * (n + m));
            double dn = ((2. * n + 1.) * (n - m + 1.) * (n + m - 1.));
            double dd = ((2. * n - 1.) * (n + m - 2.) * (n + m));
            double en = ((2. * n + 1.) * (n - m));
            double ed = ((2. * n - 1.) * (n + m));
            legpol[orc_idxmn(trc, m, n)] = sqrt(cn / cd) * legpol[orc_idxmn(trc, m - 2, n - 2)] -
                                           sqrt(dn / dd) * legpol[orc_idxmn(trc, m - 2, n - 1)] * x +
                                           sqrt(en / ed) * legpol[orc_idxmn(trc, m, n - 1)] * x;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a3  table offsets (TransLocal.cc:592-606).  trc_leg = T+1.  begin arrays have trc_leg+2 entries.
 * ---------------------------------------------------------------------------------------------- */
void orc_legendre_offsets(int trc_leg, int nlats_leg, size_t* begin_sym, size_t* begin_asym) {
    size_t ss = 0, sa = 0;
    begin_sym[0]  = 0;
    begin_asym[0] = 0;
    for (int m = 0; m <= trc_leg; ++m) {
        ss += orc_pad8(orc_num_n(trc_leg, m, 1) * (size_t)nlats_leg);
        sa += orc_pad8(orc_num_n(trc_leg, m, 0) * (size_t)nlats_leg);
        begin_sym[m + 1]  = ss;
        begin_asym[m + 1] = sa;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a6  compute_legendre_polynomials (LegendrePolynomials.cc:154-209): scatter each latitude's triangle
 *     into the symmetric / antisymmetric tables, block m column-major K x nlats, n DESCENDING.
 * ---------------------------------------------------------------------------------------------- */
void orc_legendre_tables(int trc, int nlats, const double* lats_rad, double* leg_sym, double* leg_asym,
                         const size_t* begin_sym, const size_t* begin_asym) {
    size_t tri   = (size_t)(trc + 2) * (size_t)(trc + 1) / 2;
    double* zfn0 = (double*)calloc((size_t)(trc + 1) * (trc + 1), sizeof(double));
    orc_compute_zfn(trc, zfn0);
#pragma omp parallel
    {
        double* legpol = (double*)malloc(tri * sizeof(double));
        double* zfn    = (double*)malloc((size_t)(trc + 1) * (trc + 1) * sizeof(double));
        double* vsin   = (double*)malloc((size_t)(trc + 1) * sizeof(double));
        double* vcos   = (double*)malloc((size_t)(trc + 1) * sizeof(double));
        memcpy(zfn, zfn0, (size_t)(trc + 1) * (trc + 1) * sizeof(double));
#pragma omp for schedule(dynamic, 1)
        for (int jlat = 0; jlat < nlats; ++jlat) {
            orc_legendre_lat(trc, lats_rad[jlat], legpol, zfn, vsin, vcos);
            for (int m = 0; m <= trc; ++m) {
                size_t ks = orc_num_n(trc, m, 1), ka = orc_num_n(trc, m, 0);
                size_t is = 0, ia = 0;
                for (int n = trc; n >= m; --n) {
                    if ((n - m) % 2 == 0) {
                        leg_sym[begin_sym[m] + ks * (size_t)jlat + is++] = legpol[orc_idxmn(trc, m, n)];
                    }
                    else {
                        leg_asym[begin_asym[m] + ka * (size_t)jlat + ia++] = legpol[orc_idxmn(trc, m, n)];
                    }
                }
            }
        }
        free(legpol);
        free(zfn);
        free(vsin);
        free(vcos);
    }
    free(zfn0);
}

/* ------------------------------------------------------------------------------------------------
 * Plan: the geometry TransLocal's constructor derives for a GLOBAL structured grid
 * (TransLocal.cc:371-488, 533-558).  Cropped domains are out of scope.
 * ---------------------------------------------------------------------------------------------- */
typedef struct orc_plan {
    int T;          /* truncation_ */
    int nlats;      /* g.ny() == nlatsGlobal_ */
    int nlats_nh;   /* nlatsNH_ */
    int nlats_sh;   /* nlatsSH_ */
    int nlats_leg;  /* nlatsLeg_ = (nlatsGlobal+1)/2 */
    int nlats_legr; /* nlatsLegReduced_ (== nlatsLegDomain_ for a global grid) */
    int regular;    /* RegularGrid(gridGlobal_) */
    int nxmax;
    int64_t npts;
    int* nx;         /* [nlats] */
    double* lat_deg; /* [nlats] */
    int* nlat0;      /* [T+1] */
    size_t* begin_sym;
    size_t* begin_asym; /* [T+3] */
    double* leg_sym;
    double* leg_asym; /* NULL when built without tables */
    double* lats_leg; /* [nlats_leg] radians, clamped */
} orc_plan;

static int orc_approx_zero(double v) { /* eckit::types::is_approximately_equal(lat, 0.) default eps */
    return fabs(v) <= DBL_EPSILON;
}

orc_plan* orc_plan_create(int T, int nlats, const int* nx, const double* lat_deg, int regular, int with_tables) {
    orc_plan* p = (orc_plan*)calloc(1, sizeof(orc_plan));
    p->T        = T;
    p->nlats    = nlats;
    p->regular  = regular;
    p->nx       = (int*)malloc(sizeof(int) * nlats);
    p->lat_deg  = (double*)malloc(sizeof(double) * nlats);
    memcpy(p->nx, nx, sizeof(int) * nlats);
    memcpy(p->lat_deg, lat_deg, sizeof(double) * nlats);
    int neq = 0;
    for (int j = 0; j < nlats; ++j) { /* :371-380 */
        double lat = lat_deg[j];
        if (orc_approx_zero(lat)) neq++;
        else if (lat < 0) p->nlats_sh++;
        else p->nlats_nh++;
        if (nx[j] > p->nxmax) p->nxmax = nx[j];
        p->npts += nx[j];
    }
    if (neq > 0) { /* :381-384 */
        p->nlats_nh++;
        p->nlats_sh++;
    }
    int nlats_leg_domain = p->nlats_nh >= p->nlats_sh ? p->nlats_nh : p->nlats_sh; /* :385-390 */
    p->nlats_leg         = (nlats + 1) / 2;                                         /* :435 */
    int jlat_min_leg     = 0; /* global grid: jlatMin_ = 0 and NH>=SH, so jlatMinLeg_ = 0 (:441-457) */
    p->nlats_legr        = jlat_min_leg + nlats_leg_domain; /* :459 */

    /* nlat0 (:462-488) */
    p->nlat0  = (int*)malloc(sizeof(int) * (T + 1));
    int nmen0 = -1;
    for (int jlat = 0; jlat < nlats / 2; ++jlat) {
        double lat = lat_deg[jlat] * ORC_DEG2RAD;
        int nmen   = orc_fourier_truncation(T, nx[jlat], p->nxmax, nlats, lat, regular);
        if (nmen0 > nmen) nmen = nmen0;
        int ndgluj = jlat_min_leg > jlat ? jlat_min_leg : jlat;
        for (int j = nmen0 + 1; j <= nmen; ++j) p->nlat0[j] = ndgluj;
        nmen0 = nmen;
    }
    for (int j = nmen0 + 1; j <= T; ++j) p->nlat0[j] = p->nlats_leg;

    /* latitudes of the Legendre rows, clamped, radians (:533-545) */
    p->lats_leg = (double*)malloc(sizeof(double) * p->nlats_leg);
    for (int j = 0; j < p->nlats_leg; ++j) {
        double lat = lat_deg[j];
        if (lat > ORC_LATPOLE) lat = ORC_LATPOLE;
        if (lat < -ORC_LATPOLE) lat = -ORC_LATPOLE;
        p->lats_leg[j] = lat * ORC_DEG2RAD;
    }
    p->begin_sym  = (size_t*)malloc(sizeof(size_t) * (T + 3));
    p->begin_asym = (size_t*)malloc(sizeof(size_t) * (T + 3));
    orc_legendre_offsets(T + 1, p->nlats_leg, p->begin_sym, p->begin_asym);
    if (with_tables) {
        p->leg_sym  = (double*)calloc(p->begin_sym[T + 2], sizeof(double));
        p->leg_asym = (double*)calloc(p->begin_asym[T + 2], sizeof(double));
        orc_legendre_tables(T + 1, p->nlats_leg, p->lats_leg, p->leg_sym, p->leg_asym, p->begin_sym, p->begin_asym);
    }
    return p;
}

void orc_plan_destroy(orc_plan* p) {
    if (!p) return;
    free(p->nx);
    free(p->lat_deg);
    free(p->nlat0);
    free(p->begin_sym);
    free(p->begin_asym);
    free(p->leg_sym);
    free(p->leg_asym);
    free(p->lats_leg);
    free(p);
}

/* accessors for ctypes */
int orc_plan_nlat0(const orc_plan* p, int m) { return p->nlat0[m]; }
int orc_plan_nlats_leg(const orc_plan* p) { return p->nlats_leg; }
int orc_plan_nlats_legr(const orc_plan* p) { return p->nlats_legr; }
int64_t orc_plan_npts(const orc_plan* p) { return p->npts; }
size_t orc_plan_size_sym(const orc_plan* p) { return p->begin_sym[p->T + 2]; }
size_t orc_plan_size_asym(const orc_plan* p) { return p->begin_asym[p->T + 2]; }
const double* orc_plan_leg_sym(const orc_plan* p) { return p->leg_sym; }
const double* orc_plan_leg_asym(const orc_plan* p) { return p->leg_asym; }
size_t orc_plan_begin_sym(const orc_plan* p, int m) { return p->begin_sym[m]; }
size_t orc_plan_begin_asym(const orc_plan* p, int m) { return p->begin_asym[m]; }

/* posMethod (TransLocal.h:177-180): layout of the Fourier intermediate */
static inline size_t orc_pos_fourier(const orc_plan* p, int fld, int imag, int jlat, int m) {
    return (size_t)imag + 2 * ((size_t)m + (size_t)(p->T + 1) * ((size_t)jlat + (size_t)p->nlats * (size_t)fld));
}

/* ------------------------------------------------------------------------------------------------
 * a8-a10  invtrans_legendre (TransLocal.cc:939-1097): per zonal wavenumber split the spectra by
 * parity (n descending from T+1), two column-major GEMMs against the tables, merge hemispheres.
 * `trc` is the CALL's truncation (T for scalars, T+1 on the vor/div path); entries are taken only
 * if n <= trc && m < trc (:982).  scl_fourier must be zero-initialised by the caller (:1423-1428).
 * ---------------------------------------------------------------------------------------------- */
/* a9: C(rows x L) += A(rows x K) * B(K x L), all column-major with leading dimensions rows, K, rows
 * (MatrixMultiply.tcc:45-68 -> eckit gemm; "generic" backend order: column of C by column, k outer, r inner).
 * Known-answer test of the reference: src/tests/linalg/test_linalg_dense.cc:117-136. */
void orc_gemm(int rows, size_t K, int L, const double* A, const double* B, double* C) {
    for (int c = 0; c < L; ++c) {
        for (size_t k = 0; k < K; ++k) {
            const double b = B[(size_t)c * K + k];
            for (int r = 0; r < rows; ++r) C[(size_t)c * rows + r] += A[k * rows + r] * b;
        }
    }
}

void orc_invtrans_legendre(const orc_plan* p, int trc, int nf, const double* sp, double* scl_fourier) {
    const int T = p->T;
#pragma omp parallel for schedule(dynamic, 1)
    for (int m = 0; m <= T; ++m) {
        size_t ks        = orc_num_n(T + 1, m, 1);
        size_t ka        = orc_num_n(T + 1, m, 0);
        const int n_imag = m ? 2 : 1;
        int L            = p->nlats_legr - p->nlat0[m];
        int rows         = nf * n_imag;
        if (rows * L > 0) {
            double* a_sym  = (double*)malloc(sizeof(double) * rows * (ks ? ks : 1));
            double* a_asym = (double*)malloc(sizeof(double) * rows * (ka ? ka : 1));
            double* c_sym  = (double*)calloc((size_t)rows * L, sizeof(double));
            double* c_asym = (double*)calloc((size_t)rows * L, sizeof(double));
            /* split (:970-1003) */
            size_t is = 0, ia = 0;
            size_t ioff = (size_t)(2 * trc + 3 - m) * m / 2 * nf * 2;
            for (int n = T + 1; n >= m; --n) {
                for (int imag = 0; imag < n_imag; ++imag) {
                    for (int f = 0; f < nf; ++f) {
                        size_t idx = (size_t)f + (size_t)nf * (imag + 2 * (size_t)(n - m));
                        double v   = (n <= trc && m < trc) ? sp[idx + ioff] : 0.;
                        if ((n - m) % 2 == 0) a_sym[is++] = v;
                        else a_asym[ia++] = v;
                    }
                }
            }
            /* C(rows x L) = A(rows x K) * B(K x L), all column-major (:1007-1023; eckit "generic" order) */
            const double* b_sym  = p->leg_sym + p->begin_sym[m] + (size_t)p->nlat0[m] * ks;
            const double* b_asym = p->leg_asym + p->begin_asym[m] + (size_t)p->nlat0[m] * ka;
            orc_gemm(rows, ks, L, a_sym, b_sym, c_sym);
            orc_gemm(rows, ka, L, a_asym, b_asym, c_asym);
            /* merge hemispheres (:1031-1080); posFourier :955-957 */
            for (int jlat = 0; jlat < p->nlats_nh; ++jlat) {
                int c = L - p->nlats_nh + jlat;
                for (int imag = 0; imag < n_imag; ++imag)
                    for (int f = 0; f < nf; ++f) {
                        double v = 0.;
                        if (c >= 0) {
                            size_t idx = (size_t)f + (size_t)nf * (imag + (size_t)n_imag * c);
                            v          = c_sym[idx] + c_asym[idx];
                        }
                        scl_fourier[orc_pos_fourier(p, f, imag, jlat, m)] = v;
                    }
            }
            for (int jlat = 0; jlat < p->nlats_sh; ++jlat) {
                int c     = L - p->nlats_sh + jlat;
                int jslat = p->nlats - jlat - 1;
                for (int imag = 0; imag < n_imag; ++imag)
                    for (int f = 0; f < nf; ++f) {
                        double v = 0.;
                        if (c >= 0) {
                            size_t idx = (size_t)f + (size_t)nf * (imag + (size_t)n_imag * c);
                            v          = c_sym[idx] - c_asym[idx];
                        }
                        scl_fourier[orc_pos_fourier(p, f, imag, jslat, m)] = v;
                    }
            }
            free(a_sym);
            free(a_asym);
            free(c_sym);
            free(c_asym);
        }
        else {
            for (int jlat = 0; jlat < p->nlats; ++jlat)
                for (int imag = 0; imag < n_imag; ++imag)
                    for (int f = 0; f < nf; ++f) scl_fourier[orc_pos_fourier(p, f, imag, jlat, m)] = 0.;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a13  linalg::FFT::inverse_c2r contract (FFT.h:27-72, FFTW.cc:38-61, pocketfft.cc:32-60):
 *      unnormalised Hermitian c2r, in = n/2+1 complex (interleaved), out = n reals.
 *      The third-party FFT (FFTW3 / pocketfft_hdronly) is absent from /root/reference; the oracle uses
 *      (a) a direct O(n * nmodes) DFT with table twiddles  [orc_c2r_direct], and
 *      (b) a mixed-radix + Bluestein FFT                   [orc_c2r_fft, used by the CPU baseline];
 *      both are pinned against numpy.fft.irfft (pocketfft C) in tests/test_oracle_fft.py.
 * ---------------------------------------------------------------------------------------------- */
void orc_c2r_direct(int n, const double* in, double* out) {
    /* twiddle table in long double, indexed by (m*k) mod n so no phase error accumulates */
    long double* cs = (long double*)malloc(sizeof(long double) * 2 * n);
    for (int j = 0; j < n; ++j) {
        long double ang = 2.0L * ORC_PIL * (long double)j / (long double)n;
        cs[2 * j]       = cosl(ang);
        cs[2 * j + 1]   = sinl(ang);
    }
    int nc = n / 2 + 1;
    /* highest non-zero mode (cheap pruning; zeros contribute nothing) */
    int mtop = nc - 1;
    while (mtop > 0 && in[2 * mtop] == 0. && in[2 * mtop + 1] == 0.) mtop--;
    for (int k = 0; k < n; ++k) {
        long double s = in[0];
        for (int m = 1; m <= mtop; ++m) {
            int j = (int)(((int64_t)m * k) % n);
            if (2 * m == n) {
                s += (long double)in[2 * m] * cs[2 * j]; /* Nyquist: real part only */
            }
            else {
                s += 2.0L * ((long double)in[2 * m] * cs[2 * j] - (long double)in[2 * m + 1] * cs[2 * j + 1]);
            }
        }
        out[k] = (double)s;
    }
    free(cs);
}

/* ---- double-precision c2r FFT: half-length complex transform (c2r pre-processing), recursive mixed radix with
 *      table twiddles for 7-smooth lengths, Bluestein (power-of-two convolution) otherwise; per-length plans are
 *      cached (tables are computed once, in extended precision). ---- */
typedef struct { double re, im; } orc_cplx;

typedef struct orc_fftplan {
    int n, h;        /* real length, complex half length */
    orc_cplx* wn;    /* [h] exp(+2 pi i k / n) */
    orc_cplx* wh;    /* [h] exp(+2 pi i k / h) (direct) */
    int bluestein, M;
    orc_cplx* wM;    /* [M] exp(+2 pi i k / M) */
    orc_cplx* chirp; /* [h] exp(+i pi k^2 / h) */
    orc_cplx* fb;    /* [M] FFT_M(conj chirp, wrapped) / M */
    struct orc_fftplan* next;
} orc_fftplan;

static orc_fftplan* orc_plans = NULL;

static orc_cplx orc_root(int64_t j, int64_t n) {
    j %= n;
    if (j < 0) j += n;
    long double ang = 2.0L * ORC_PIL * (long double)j / (long double)n;
    orc_cplx w      = {(double)cosl(ang), (double)sinl(ang)};
    return w;
}

static int orc_smallest_factor(int n) {
    if (n % 4 == 0) return 4;
    if (n % 2 == 0) return 2;
    for (int f = 3; (int64_t)f * f <= n; f += 2)
        if (n % f == 0) return f;
    return n;
}

static int orc_is_smooth(int n) {
    for (int p = 2; p <= 7; p += (p == 2 ? 1 : 2))
        while (n % p == 0) n /= p;
    return n == 1;
}

/* out[k] = sum_j in[j*stride] exp(sign 2 pi i jk/n); recursive decimation in time.  w: table of N roots, N % n == 0 */
static void orc_fft_rec(int n, int sign, const orc_cplx* in, int stride, orc_cplx* out, orc_cplx* scratch,
                        const orc_cplx* w, int N) {
    if (n == 1) {
        out[0] = in[0];
        return;
    }
    int r = orc_smallest_factor(n);
    int m = n / r;
    for (int q = 0; q < r; ++q) orc_fft_rec(m, sign, in + (size_t)q * stride, stride * r, scratch + (size_t)q * m, out, w, N);
    const int tn = N / n, tr = N / r;
    orc_cplx t[16];
    for (int k = 0; k < m; ++k) {
        for (int q = 0; q < r; ++q) {
            orc_cplx tw = w[(size_t)(((int64_t)q * k) % n) * tn];
            double s    = sign * tw.im;
            orc_cplx v  = scratch[(size_t)q * m + k];
            t[q].re     = v.re * tw.re - v.im * s;
            t[q].im     = v.re * s + v.im * tw.re;
        }
        for (int pp = 0; pp < r; ++pp) {
            double re = 0, im = 0;
            for (int q = 0; q < r; ++q) {
                orc_cplx tw = w[(size_t)((pp * q) % r) * tr];
                double s    = sign * tw.im;
                re += t[q].re * tw.re - t[q].im * s;
                im += t[q].re * s + t[q].im * tw.re;
            }
            out[k + (size_t)pp * m].re = re;
            out[k + (size_t)pp * m].im = im;
        }
    }
}

static orc_fftplan* orc_plan_for(int n) {
    orc_fftplan* p;
#pragma omp critical(orc_plan_cache)
    {
        for (p = orc_plans; p; p = p->next)
            if (p->n == n) break;
        if (!p) {
            p       = (orc_fftplan*)calloc(1, sizeof(orc_fftplan));
            p->n    = n;
            int h   = n / 2;
            p->h    = h;
            p->wn   = (orc_cplx*)malloc(sizeof(orc_cplx) * (h > 0 ? h : 1));
            for (int k = 0; k < h; ++k) p->wn[k] = orc_root(k, n);
            if (h > 0 && orc_is_smooth(h)) {
                p->wh = (orc_cplx*)malloc(sizeof(orc_cplx) * h);
                for (int k = 0; k < h; ++k) p->wh[k] = orc_root(k, h);
            }
            else if (h > 0) {
                p->bluestein = 1;
                int M        = 1;
                while (M < 2 * h - 1) M *= 2;
                p->M     = M;
                p->wM    = (orc_cplx*)malloc(sizeof(orc_cplx) * M);
                p->chirp = (orc_cplx*)malloc(sizeof(orc_cplx) * h);
                p->fb    = (orc_cplx*)malloc(sizeof(orc_cplx) * M);
                for (int k = 0; k < M; ++k) p->wM[k] = orc_root(k, M);
                orc_cplx* bb = (orc_cplx*)calloc(M, sizeof(orc_cplx));
                orc_cplx* sc = (orc_cplx*)malloc(sizeof(orc_cplx) * M);
                for (int k = 0; k < h; ++k) {
                    p->chirp[k] = orc_root(((int64_t)k * k) % (2 * (int64_t)h), 2 * (int64_t)h);
                    bb[k].re    = p->chirp[k].re;
                    bb[k].im    = -p->chirp[k].im;
                    if (k) bb[M - k] = bb[k];
                }
                orc_fft_rec(M, -1, bb, 1, p->fb, sc, p->wM, M);
                for (int k = 0; k < M; ++k) {
                    p->fb[k].re /= M;
                    p->fb[k].im /= M;
                }
                free(bb);
                free(sc);
            }
            p->next   = orc_plans;
            orc_plans = p;
        }
    }
    return p;
}

/* work: at least 3*max(M,h) complex */
static void orc_c2r_fft_work(int n, const double* in, double* out, orc_cplx* work) {
    if (n % 2) { /* odd length: direct sum (never used by Gaussian grids) */
        orc_c2r_direct(n, in, out);
        return;
    }
    orc_fftplan* p = orc_plan_for(n);
    const int h    = p->h;
    const int L    = p->bluestein ? p->M : h;
    orc_cplx *a = work, *b = work + L, *sc = work + 2 * (size_t)L;
    /* c2r pre-processing: Z[k] = (X[k] + conj X[h-k]) + i w_n^k (X[k] - conj X[h-k]);  Im X[0] = Im X[h] = 0 */
    for (int k = 0; k < h; ++k) {
        double ar = in[2 * k], ai = (k == 0) ? 0. : in[2 * k + 1];
        int kk    = h - k;
        double br = in[2 * kk], bi = (kk == h) ? 0. : -in[2 * kk + 1];
        double sr = ar + br, si = ai + bi, dr = ar - br, di = ai - bi;
        double er = dr * p->wn[k].re - di * p->wn[k].im, ei = dr * p->wn[k].im + di * p->wn[k].re;
        a[k].re   = sr - ei;
        a[k].im   = si + er;
    }
    if (!p->bluestein) {
        orc_fft_rec(h, +1, a, 1, b, sc, p->wh, h);
    }
    else {
        const int M = p->M;
        for (int k = 0; k < h; ++k) {
            double re = a[k].re * p->chirp[k].re - a[k].im * p->chirp[k].im;
            double im = a[k].re * p->chirp[k].im + a[k].im * p->chirp[k].re;
            a[k].re   = re;
            a[k].im   = im;
        }
        for (int k = h; k < M; ++k) a[k].re = a[k].im = 0.;
        orc_fft_rec(M, -1, a, 1, b, sc, p->wM, M);
        for (int k = 0; k < M; ++k) {
            double re = b[k].re * p->fb[k].re - b[k].im * p->fb[k].im;
            double im = b[k].re * p->fb[k].im + b[k].im * p->fb[k].re;
            a[k].re   = re;
            a[k].im   = im;
        }
        orc_fft_rec(M, +1, a, 1, b, sc, p->wM, M);
        for (int k = 0; k < h; ++k) {
            double re = b[k].re * p->chirp[k].re - b[k].im * p->chirp[k].im;
            double im = b[k].re * p->chirp[k].im + b[k].im * p->chirp[k].re;
            b[k].re   = re;
            b[k].im   = im;
        }
    }
    for (int j = 0; j < h; ++j) {
        out[2 * j]     = b[j].re;
        out[2 * j + 1] = b[j].im;
    }
}

static size_t orc_fft_work_size(int nmax) {
    size_t M = 1;
    while (M < (size_t)nmax) M *= 2; /* >= 2h-1 for h = nmax/2 */
    return 3 * (M > (size_t)nmax ? M : (size_t)nmax) + 16;
}

void orc_c2r_fft(int n, const double* in, double* out) {
    orc_cplx* work = (orc_cplx*)malloc(sizeof(orc_cplx) * orc_fft_work_size(n));
    orc_c2r_fft_work(n, in, out, work);
    free(work);
}

/* ------------------------------------------------------------------------------------------------
 * a11/a12  invtrans_fourier_reduced / _regular (TransLocal.cc:1101-1196), global grid (jlonMin=0).
 * use_fft = 0: direct DFT (slow, most accurate);  1: FFT.
 * ---------------------------------------------------------------------------------------------- */
void orc_invtrans_fourier(const orc_plan* p, int nf, const double* scl_fourier, double* gp, int use_fft) {
    const int T = p->T;
    int64_t* rowoff = (int64_t*)malloc(sizeof(int64_t) * (p->nlats + 1));
    rowoff[0]       = 0;
    for (int j = 0; j < p->nlats; ++j) rowoff[j + 1] = rowoff[j] + p->nx[j];
#pragma omp parallel
    {
        double* in  = (double*)malloc(sizeof(double) * 2 * (p->nxmax / 2 + 1));
        double* out = (double*)malloc(sizeof(double) * p->nxmax);
        orc_cplx* fw = (orc_cplx*)malloc(sizeof(orc_cplx) * orc_fft_work_size(p->nxmax));
#pragma omp for collapse(2) schedule(dynamic, 8)
        for (int f = 0; f < nf; ++f) {
            for (int jlat = 0; jlat < p->nlats; ++jlat) {
                int n  = p->regular ? p->nxmax : p->nx[jlat];
                int nc = n / 2 + 1;
                in[0]  = scl_fourier[orc_pos_fourier(p, f, 0, jlat, 0)];
                in[1]  = 0.;
                for (int m = 1; m < nc; ++m) {
                    for (int imag = 0; imag < 2; ++imag) {
                        in[2 * m + imag] = (m <= T) ? scl_fourier[orc_pos_fourier(p, f, imag, jlat, m)] : 0.;
                    }
                }
                if (use_fft) orc_c2r_fft_work(n, in, out, fw);
                else orc_c2r_direct(n, in, out);
                double* dst = gp + (size_t)f * p->npts + rowoff[jlat];
                for (int i = 0; i < p->nx[jlat]; ++i) dst[i] = out[i];
            }
        }
        free(in);
        free(out);
        free(fw);
    }
    free(rowoff);
}

/* a7  invtrans_uv, scalar part (TransLocal.cc:1409-1441): zero-fill, Legendre, Fourier */
void orc_invtrans_uv(const orc_plan* p, int trc, int nf, int nb_vordiv, const double* sp, double* gp, int use_fft) {
    if (nf <= 0) return;
    size_t nfour = (size_t)nf * 2 * p->nlats * (p->T + 1);
    double* four = (double*)calloc(nfour, sizeof(double));
    orc_invtrans_legendre(p, trc, nf, sp, four);
    orc_invtrans_fourier(p, nf, four, gp, use_fft);
    /* a14: u,v from U,V (TransLocal.cc:1443-1469) */
    if (nb_vordiv > 0) {
        size_t idx = 0;
        for (int f = 0; f < 2 * nb_vordiv && f < nf; ++f) {
            for (int jlat = 0; jlat < p->nlats; ++jlat) {
                double lat = p->lat_deg[jlat];
                if (lat > ORC_LATPOLE) lat = ORC_LATPOLE;
                if (lat < -ORC_LATPOLE) lat = -ORC_LATPOLE;
                double inv = 1. / cos(lat * ORC_DEG2RAD);
                for (int i = 0; i < p->nx[jlat]; ++i) gp[idx++] *= inv;
            }
        }
    }
    free(four);
}

void orc_invtrans(const orc_plan* p, int nf, const double* sp, double* gp, int use_fft) {
    orc_invtrans_uv(p, p->T, nf, 0, sp, gp, use_fft); /* TransLocal.cc:931-934 */
}

/* expose the Fourier intermediate for stage-level parity tests */
void orc_invtrans_legendre_only(const orc_plan* p, int trc, int nf, const double* sp, double* scl_fourier) {
    memset(scl_fourier, 0, sizeof(double) * (size_t)nf * 2 * p->nlats * (p->T + 1));
    orc_invtrans_legendre(p, trc, nf, sp, scl_fourier);
}

/* ------------------------------------------------------------------------------------------------
 * Row-sampled evaluation for FULL-SIZE parity (TL1279): computes gp rows (jlat in rows[], all nf fields)
 * exactly as the table path would, but evaluating the polynomials of one latitude on the fly, so that
 * no 8 GB table is needed.  Same split / n-descending summation / merge / c2r as above.
 * out is [nrows][nf][nx(row)] packed row after row.
 * ---------------------------------------------------------------------------------------------- */
void orc_invtrans_rows(const orc_plan* p, int trc, int nf, const double* sp, int nrows, const int* rows, double* out,
                       int use_fft) {
    const int T   = p->T;
    const int TL  = T + 1;
    size_t tri    = (size_t)(TL + 2) * (size_t)(TL + 1) / 2;
    double* zfn0  = (double*)calloc((size_t)(TL + 1) * (TL + 1), sizeof(double));
    orc_compute_zfn(TL, zfn0);
    int64_t* outoff = (int64_t*)malloc(sizeof(int64_t) * (nrows + 1));
    outoff[0]       = 0;
    for (int r = 0; r < nrows; ++r) outoff[r + 1] = outoff[r] + (int64_t)nf * p->nx[rows[r]];
#pragma omp parallel
    {
        double* legpol = (double*)malloc(tri * sizeof(double));
        double* zfn    = (double*)malloc((size_t)(TL + 1) * (TL + 1) * sizeof(double));
        double* vsin   = (double*)malloc((size_t)(TL + 1) * sizeof(double));
        double* vcos   = (double*)malloc((size_t)(TL + 1) * sizeof(double));
        double* in     = (double*)malloc(sizeof(double) * 2 * (p->nxmax / 2 + 1));
        double* o      = (double*)malloc(sizeof(double) * p->nxmax);
        double* four   = (double*)malloc(sizeof(double) * 2 * (size_t)(T + 1) * nf);
        memcpy(zfn, zfn0, (size_t)(TL + 1) * (TL + 1) * sizeof(double));
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < nrows; ++r) {
            int jlat  = rows[r];
            int south = jlat >= p->nlats_nh; /* pure southern row (global grid, no equator row) */
            int jleg  = south ? p->nlats - 1 - jlat : jlat;
            orc_legendre_lat(TL, p->lats_leg[jleg], legpol, zfn, vsin, vcos);
            for (int m = 0; m <= T; ++m) {
                const int n_imag = m ? 2 : 1;
                size_t ioff      = (size_t)(2 * trc + 3 - m) * m / 2 * nf * 2;
                for (int f = 0; f < nf; ++f) {
                    for (int imag = 0; imag < 2; ++imag) {
                        double s = 0., a = 0.;
                        if (imag < n_imag && jleg >= p->nlat0[m]) {
                            for (int n = TL; n >= m; --n) {
                                size_t idx = (size_t)f + (size_t)nf * (imag + 2 * (size_t)(n - m));
                                double v   = (n <= trc && m < trc) ? sp[idx + ioff] : 0.;
                                double pl  = legpol[orc_idxmn(TL, m, n)];
                                if ((n - m) % 2 == 0) s += v * pl;
                                else a += v * pl;
                            }
                        }
                        four[((size_t)f * (T + 1) + m) * 2 + imag] = south ? s - a : s + a;
                    }
                }
            }
            int n  = p->regular ? p->nxmax : p->nx[jlat];
            int nc = n / 2 + 1;
            for (int f = 0; f < nf; ++f) {
                const double* ff = four + (size_t)f * (T + 1) * 2;
                in[0]            = ff[0];
                in[1]            = 0.;
                for (int m = 1; m < nc; ++m) {
                    in[2 * m]     = (m <= T) ? ff[2 * m] : 0.;
                    in[2 * m + 1] = (m <= T) ? ff[2 * m + 1] : 0.;
                }
                if (use_fft) orc_c2r_fft(n, in, o);
                else orc_c2r_direct(n, in, o);
                memcpy(out + outoff[r] + (size_t)f * p->nx[jlat], o, sizeof(double) * p->nx[jlat]);
            }
        }
        free(legpol);
        free(zfn);
        free(vsin);
        free(vcos);
        free(in);
        free(o);
        free(four);
    }
    free(zfn0);
    free(outoff);
}

/* ------------------------------------------------------------------------------------------------
 * a15  vor/div path.
 *   extend_truncation  (TransLocal.cc:1496-1519): T -> T+1, new row / column zero.
 *   vd2uv              (VorDivToUVLocal.cc:62-184): U,V*cos(lat) spectra from vorticity / divergence,
 *                      Temperton (1991) eq. 2.12 / 2.13, n-reversed internal storage, scaled by 1/a.
 *   invtrans(ns, sp, nvd, vor, div, gp) (TransLocal.cc:1523-1597): per (m,n,imag) interleave
 *                      [U fields][V fields][scalar fields], then invtrans_uv with truncation T+1.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_EARTH_RADIUS 6371229. /* util::Earth::radius(), src/atlas/util/Earth.h:23 */

void orc_extend_truncation(int old_trc, int nf, const double* old_sp, double* new_sp) {
    int new_trc = old_trc + 1;
    size_t k = 0, ko = 0;
    for (int m = 0; m <= new_trc; ++m)
        for (int n = m; n <= new_trc; ++n)
            for (int imag = 0; imag < 2; ++imag)
                for (int f = 0; f < nf; ++f) {
                    if (m == new_trc || n == new_trc) new_sp[k++] = 0.;
                    else new_sp[k++] = old_sp[ko++];
                }
}

void orc_vd2uv(int trc, int nf, const double* vor, const double* div, double* U, double* V) {
    const double ra = ORC_EARTH_RADIUS;
    int nlei1       = trc + 4 + (trc + 4 + 1) % 2;
    double* repsnm  = (double*)calloc((size_t)(trc + 1) * (trc + 6) / 2 + 8, sizeof(double));
    double* rlapin  = (double*)calloc(trc + 3, sizeof(double));
    size_t idx      = 0;
    for (int m = 0; m <= trc; ++m)
        for (int n = m; n <= trc + 2; ++n, ++idx) repsnm[idx] = sqrt(((double)n * n - (double)m * m) / (4. * n * n - 1.));
    repsnm[0] = 0.;
    for (int n = 1; n <= trc + 2; ++n) rlapin[n] = -ra * ra / (n * (n + 1.));
    rlapin[0] = 0.;
    double* zeps = (double*)calloc(trc + 6, sizeof(double));
    double* zlap = (double*)calloc(trc + 6, sizeof(double));
    double* zn   = (double*)calloc(trc + 6, sizeof(double));
    size_t nint  = (size_t)2 * nf * nlei1;
    double* rvor = (double*)malloc(sizeof(double) * nint);
    double* rdiv = (double*)malloc(sizeof(double) * nint);
    double* ru   = (double*)malloc(sizeof(double) * nint);
    double* rv   = (double*)malloc(sizeof(double) * nint);
    for (int m = 0; m <= trc; ++m) {
        for (int n = m - 1; n <= trc + 2; ++n) { /* reversed order for accuracy (:98-116) */
            int ij = trc + 3 - n;
            if (n >= 0) {
                zlap[ij] = rlapin[n];
                zeps[ij] = (n < m) ? 0. : repsnm[n + (2 * trc - m + 5) * m / 2];
            }
            else {
                zlap[ij] = 0.;
                zeps[ij] = 0.;
            }
            zn[ij] = n;
        }
        zn[0] = trc + 3;
        /* prfi1b (:31-56): spectral data of wavenumber m into the n-reversed internal layout */
        int ilcm = trc + 1 - m, ioff = (2 * trc - m + 3) * m;
        for (int pass = 0; pass < 2; ++pass) {
            const double* src = pass ? div : vor;
            double* dst       = pass ? rdiv : rvor;
            memset(dst, 0, sizeof(double) * nint);
            for (int j = 1; j <= ilcm; ++j) {
                int inm = ioff + (ilcm - j) * 2;
                for (int f = 0; f < nf; ++f) {
                    dst[(size_t)(2 * f) * nlei1 + j + 1]     = src[(size_t)inm * nf + f];
                    dst[(size_t)(2 * f + 1) * nlei1 + j + 1] = src[(size_t)(inm + 1) * nf + f];
                }
            }
        }
        memset(ru, 0, sizeof(double) * nint);
        memset(rv, 0, sizeof(double) * nint);
        for (int f = 0; f < nf; ++f) { /* (:131-156) */
            long ir = (long)2 * f * nlei1 - 1, ii = ir + nlei1;
            for (int ji = 2; ji < trc + 4 - m; ++ji) {
                double psiM1 = zn[ji + 1] * zeps[ji] * zlap[ji + 1];
                double psiP1 = zn[ji - 2] * zeps[ji - 1] * zlap[ji - 1];
                if (m == 0) {
                    ru[ir + ji] = +psiM1 * rvor[ir + ji + 1] - psiP1 * rvor[ir + ji - 1];
                    rv[ir + ji] = -psiM1 * rdiv[ir + ji + 1] + psiP1 * rdiv[ir + ji - 1];
                }
                else {
                    double chiIm = m * zlap[ji];
                    ru[ir + ji]  = -chiIm * rdiv[ii + ji] + psiM1 * rvor[ir + ji + 1] - psiP1 * rvor[ir + ji - 1];
                    ru[ii + ji]  = +chiIm * rdiv[ir + ji] + psiM1 * rvor[ii + ji + 1] - psiP1 * rvor[ii + ji - 1];
                    rv[ir + ji]  = -chiIm * rvor[ii + ji] - psiM1 * rdiv[ir + ji + 1] + psiP1 * rdiv[ir + ji - 1];
                    rv[ii + ji]  = +chiIm * rvor[ir + ji] - psiM1 * rdiv[ii + ji + 1] + psiP1 * rdiv[ii + ji - 1];
                }
            }
        }
        int ilcm2   = trc - m; /* copy back (:160-181) */
        double za_r = 1. / ra;
        for (int j = 0; j <= ilcm2; ++j) {
            int inm = ioff + (ilcm2 - j) * 2;
            for (int f = 0; f < nf; ++f) {
                size_t ir = (size_t)2 * f * nlei1, ii = ir + nlei1;
                size_t k  = (size_t)inm * nf + f;
                U[k]      = ru[ir + j + 2] * za_r;
                V[k]      = rv[ir + j + 2] * za_r;
                k += nf;
                U[k] = ru[ii + j + 2] * za_r;
                V[k] = rv[ii + j + 2] * za_r;
            }
        }
    }
    free(repsnm); free(rlapin); free(zeps); free(zlap); free(zn); free(rvor); free(rdiv); free(ru); free(rv);
}

/* TransLocal::invtrans(ns, sp, nvd, vor, div, gp)  (TransLocal.cc:1523-1597) */
void orc_invtrans_vordiv(const orc_plan* p, int ns, const double* sp, int nvd, const double* vor, const double* div,
                         double* gp, int use_fft) {
    const int T = p->T;
    if (nvd <= 0) {
        if (ns > 0) orc_invtrans_uv(p, T, ns, 0, sp, gp, use_fft);
        return;
    }
    size_t next = (size_t)(T + 2) * (T + 3); /* 2*legendre_size(T+1) */
    double* vor_e = (double*)malloc(sizeof(double) * next * nvd);
    double* div_e = (double*)malloc(sizeof(double) * next * nvd);
    double* U     = (double*)calloc(next * nvd, sizeof(double));
    double* V     = (double*)calloc(next * nvd, sizeof(double));
    orc_extend_truncation(T, nvd, vor, vor_e);
    orc_extend_truncation(T, nvd, div, div_e);
    orc_vd2uv(T + 1, nvd, vor_e, div_e, U, V);
    double* sc_e = NULL;
    if (ns > 0) {
        sc_e = (double*)malloc(sizeof(double) * next * ns);
        orc_extend_truncation(T, ns, sp, sc_e);
    }
    int nall    = 2 * nvd + ns;
    double* all = (double*)malloc(sizeof(double) * next * nall);
    size_t k = 0, i = 0, j = 0, l = 0;
    for (int m = 0; m <= T + 1; ++m)
        for (int n = m; n <= T + 1; ++n)
            for (int imag = 0; imag < 2; ++imag) {
                for (int f = 0; f < nvd; ++f) all[k++] = U[i++];
                for (int f = 0; f < nvd; ++f) all[k++] = V[j++];
                for (int f = 0; f < ns; ++f) all[k++] = sc_e[l++];
            }
    orc_invtrans_uv(p, T + 1, nall, nvd, all, gp, use_fft);
    free(vor_e); free(div_e); free(U); free(V); free(sc_e); free(all);
}

/* Row-sampled twin of orc_invtrans_vordiv for FULL-SIZE parity of the vor/div path (no 8 GB table): the merged spectra
 * [U fields][V fields][scalar fields] at truncation T+1 exactly as above (extend_truncation TransLocal.cc:1496-1519,
 * vd2uv VorDivToUVLocal.cc:62-184, interleave TransLocal.cc:1555-1581), the rows through orc_invtrans_rows with the call's
 * truncation T+1 (so that every m <= T survives and the n = T+1 column of U, V enters, TransLocal.cc:970-997), then the
 * first 2 nvd fields times 1/cos(clamped latitude) (TransLocal.cc:1443-1469).  out is [nrows][2 nvd + ns][nx(row)]. */
void orc_invtrans_vordiv_rows(const orc_plan* p, int ns, const double* sp, int nvd, const double* vor, const double* div,
                              int nrows, const int* rows, double* out, int use_fft) {
    const int T = p->T;
    if (nvd <= 0) {
        if (ns > 0) orc_invtrans_rows(p, T, ns, sp, nrows, rows, out, use_fft);
        return;
    }
    size_t next   = (size_t)(T + 2) * (T + 3);
    double* vor_e = (double*)malloc(sizeof(double) * next * nvd);
    double* div_e = (double*)malloc(sizeof(double) * next * nvd);
    double* U     = (double*)calloc(next * nvd, sizeof(double));
    double* V     = (double*)calloc(next * nvd, sizeof(double));
    orc_extend_truncation(T, nvd, vor, vor_e);
    orc_extend_truncation(T, nvd, div, div_e);
    orc_vd2uv(T + 1, nvd, vor_e, div_e, U, V);
    free(vor_e);
    free(div_e);
    double* sc_e = NULL;
    if (ns > 0) {
        sc_e = (double*)malloc(sizeof(double) * next * ns);
        orc_extend_truncation(T, ns, sp, sc_e);
    }
    int nall    = 2 * nvd + ns;
    double* all = (double*)malloc(sizeof(double) * next * nall);
    size_t k = 0, i = 0, j = 0, l = 0;
    for (int m = 0; m <= T + 1; ++m)
        for (int n = m; n <= T + 1; ++n)
            for (int imag = 0; imag < 2; ++imag) {
                for (int f = 0; f < nvd; ++f) all[k++] = U[i++];
                for (int f = 0; f < nvd; ++f) all[k++] = V[j++];
                for (int f = 0; f < ns; ++f) all[k++] = sc_e[l++];
            }
    free(U);
    free(V);
    free(sc_e);
    orc_invtrans_rows(p, T + 1, nall, all, nrows, rows, out, use_fft);
    free(all);
    size_t off = 0;
    for (int r = 0; r < nrows; ++r) {
        int jlat   = rows[r];
        double lat = p->lat_deg[jlat];
        if (lat > ORC_LATPOLE) lat = ORC_LATPOLE;
        if (lat < -ORC_LATPOLE) lat = -ORC_LATPOLE;
        double inv = 1. / cos(lat * ORC_DEG2RAD);
        size_t nx  = (size_t)p->nx[jlat];
        for (size_t q = 0; q < (size_t)2 * nvd * nx; ++q) out[off + q] *= inv;
        off += (size_t)nall * nx;
    }
}
