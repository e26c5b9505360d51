// Host planner for the longitudinal inverse real FFT (one plan per distinct row length n).
//
// Reference call sites: TransLocal.cc:652-686 (one FFTW/pocketfft plan per distinct row length),
// :1101-1136 (regular grids, inverse_c2r_many) and :1155-1196 (reduced grids, inverse_c2r per row).
//
// Method per row length n (h = n/2):
//   DIRECT    n even, h is {2,3,5}-smooth : half-length complex FFT (c2r pre-processing + in-place DIT)
//   ODD       n odd and {3,5}-smooth (rows of the classic reduced Gaussian grids: 25, 45, 75, ... 3645): complex mixed-radix
//             DIT of length n on the Hermitian extension of the kept modes, real part stored  [r3]
//   HYBRID    n even, h = A*B, B smooth, A = product of the primes > 5 of h, 7 <= A <= HYB_MAX_A :
//                                          half-length complex FFT whose radix-A stage is a dense DFT on the matrix
//                                          cores (fft_core.h: "HYBRID rows"), no Bluestein
//   NATIVE    (experiments build only since round 5: tools/experiments/fft_native.h) n even, h has only small prime factors -- any
//             number of {2,...,13}, at most one prime 17..31 -- and is not a length of the specialised direct family : half-length
//             complex mixed-radix DIT with the stage list chosen at plan time, executed from tables by ONE kernel [r4]
//   BLUESTEIN n even otherwise            : half-length chirp-z with a {2,3,5}-smooth M >= 2h-1
//   DFT       n odd                       : O(n*modes) direct sum (never hit by Gaussian grids)
#pragma once
#include <cstdint>
#include <vector>

#include "fft_core.h"
#if defined(ATLAS_AMD_EXPERIMENTS)
#include "../../tools/experiments/fft_hybrid_core.h"
#include "../../tools/experiments/fft_native.h"
#endif

namespace atlas_amd {
namespace fft {

enum FftMethod : int { FFT_DIRECT = 0, FFT_BLUESTEIN = 1, FFT_DFT = 2, FFT_HYBRID = 3, FFT_ODD = 4, FFT_NATIVE = 5 };

struct FftRowPlan {
    int n;              // row length (number of longitudes of the global row)
    int h;              // n/2 (DIRECT/BLUESTEIN)
    int method;         // FftMethod
    int lds_complex;    // LDS footprint in complex elements (M)
    FftShape shape;     // M-point shape (M = h for DIRECT)
    int64_t off_tw;     // [M]  exp(+2 pi i t / M)
    int64_t off_pre;    // [h]  exp(+2 pi i k / n)        (DFT method: [n] exp(+2 pi i j / n))
    int64_t off_chirp;  // [h]  exp(+i pi k^2 / h)        (BLUESTEIN)
    int64_t off_bhat;   // [M]  DFT_M(conj chirp, wrapped) / M in DIF order (BLUESTEIN)
    int64_t off_bhat_t; // [R_last][M/R_last] the same, transposed for the compile-time specialised kernel
    int ct_f, ct_k;     // M = ct_f << ct_k handled by a specialised kernel instance (ct_k < 0: generic kernel)
    int hyb_A, hyb_B;   // HYBRID: dense radix and smooth part (h = hyb_A * hyb_B)
    int hyb_Mt, hyb_Ks; // HYBRID: tiles of the padded (A+1)/2 x (A+1)/2 cos / sin matrices (16 rows, 4 columns)
    int hyb_raw;        // HYBRID: entries of the LDS staging area for the row's modes (after the padded_size(h) work area)
    int64_t off_cs;     // HYBRID: [Mt][Ks][64] {cos, sin} operand fragments
#if defined(ATLAS_AMD_EXPERIMENTS)
    NatShape nat;       // NATIVE: stage list and table offsets (into FftPlanSet::nat_table)
#endif
};
struct PlanOptions {
    bool specialised_shapes = true;  // compile-time specialised kernel instances where they exist
    bool hybrid             = false; // dense-stage rows (HYBRID): opt-in (ATLAS_AMD_FFT_HYBRID=1) until it beats Bluestein
    int hybrid_max_a        = HYB_MAX_A;
    int hybrid_min_h        = 64;
    int max_mode            = 1 << 30;  // highest wavenumber any row can carry (sizes the staging area of HYBRID rows)
    // Small reduced grids are launch-bound: TL159 -> O160 / 60 fields spends 0.15 of its 0.21 ms in 19 row-class launches of a
    // few microseconds of work each.  With `coarse_classes` every even row takes a Bluestein row of one of a few lengths
    // (256, 512, 1024, 2048: any M >= 2h - 1 is a valid convolution length) -- up to 4 x the butterflies of the tight length,
    // a handful of launches.  Set by Trans for every reduced grid ([r6]; rounds 3 - 5: of at most 704 points per row) -- a property of the GLOBAL grid, so
    // that every decomposition of one grid plans its rows alike); ATLAS_AMD_FFT_COARSE=0/1 overrides.
    bool coarse_classes     = false;
    // native mixed-radix rows (tools/experiments/fft_native.h; experiments build, ATLAS_AMD_FFT_NATIVE=1) for every half length the
    // planner finds a stage list for -- measured at parity with the Bluestein rows they replace (TL1279 -> O1280: 25 % of the
    // points, 1.55 - 1.62 ms against 1.4 - 1.5 ms; the stage 6.60 - 6.65 against 6.61 - 6.67 ms; profiles/r04_fft_native.txt): out
    // of the product library since round 5
    bool native             = false;
    int native_min_h        = 24;
};
int coarse_bluestein_length(int n);   // smallest of 256, 512, 1024, 2048 that is >= n (0: none)

struct FftPlanSet {
    std::vector<FftRowPlan> plans;   // one per distinct n
    std::vector<cplx> table;         // all tables, concatenated (uploaded once)
    std::vector<uint32_t> nat_table; // NATIVE rows: fold permutations and per-stage butterfly tables (fft_native.h: NatShape)
    int plan_index(int n) const;
};

bool is_smooth235(int n);
int next_smooth235(int n);
int next_bluestein_length(int n);  // smallest {1,3,5}*2^k >= n
FftShape make_shape(int M, int max_pow2_radix = 16);  // M must be {2,3,5}-smooth
FftPlanSet make_fft_plans(const std::vector<int>& row_lengths, const PlanOptions& opt);
FftPlanSet make_fft_plans(const std::vector<int>& row_lengths, bool specialised_shapes = true);
int hybrid_dense_radix(int h);  // product of the prime factors > 5 of h
#if defined(ATLAS_AMD_EXPERIMENTS)
// stage list and tables of a native row of half length h (appended to `table`); false: h has no native plan (a prime factor
// above NAT_MAX_PRIME, two primes above 13, a power of two, too long, or no stage list within the per-stage butterfly limit)
bool make_native_shape(int h, NatShape& shape, std::vector<uint32_t>& table);
#endif

// Host execution of one row with exactly the kernel's algorithm (used by CPU tests; NOT a product fallback:
// nothing in the invtrans path calls it).  X: h+1 (or n/2+1) complex modes (zero beyond mmax); y: n reals.
void host_execute_row(const FftPlanSet& ps, int plan, const cplx* X, int mmax, double* y, int nthreads = 256,
                      bool use_specialised = true);

}  // namespace fft
}  // namespace atlas_amd
