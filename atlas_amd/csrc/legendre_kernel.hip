// Inverse Legendre transform on gfx950 (MI355X): batched fp64 MFMA contraction of the associated-Legendre
// table against the spectral coefficients, with the spectral parity split, the two GEMMs per zonal wavenumber and
// the hemisphere merge of the reference fused into one kernel.
//
// Reference being replaced: TransLocal::invtrans_legendre, src/atlas/trans/local/TransLocal.cc:939-1097
//   * "Legendre split"     :970-1003  -> rows of the spectral array are addressed in place (n descending, by parity)
//   * 2 x matrix_multiply  :1007-1023 -> v_mfma_f64_16x16x4_f64, both parities accumulated side by side
//   * "merge spheres"      :1031-1080 -> epilogue: north = sym + asym, south = sym - asym
//
// Work decomposition (trans_plan.h): one workgroup (8 waves) per (m, tile of 64 latitudes); it streams its
// contiguous P block [parity][kpad][64] from HBM exactly once and multiplies it with ALL (field, re/im) columns,
// which it re-reads from L2 (every tile of one m is placed on the same XCD).
//
// MFMA operand roles (layout verified on hardware, tools/probe):
//   A (16 x 4)  = P^T  : lane l holds P[k = 4*ks + (l>>4)][lat = 16*lt + (l&15)]
//   B (4 x 16)  = S    : lane l holds S[k = 4*ks + (l>>4)][r   = 16*rt + (l&15)]     r = 2*field + imag
//   D (16 x 16)        : lane l, reg g holds D[lat = (l>>4) + 4*g][r = l&15]
// so 16 consecutive lanes store 16 consecutive r (128 B) of one latitude.
//
// Output ("Fourier intermediate", own layout -- the reference's scl_fourier is internal to TransLocal):
//   F[(lat * m_cnt + m / m_div) * RP + r],  lat = 0..nlats-1 north->south, r = 2*field + imag, RP = 16*ceil(2*nf/16)
//   (single device: m_div = 1, m_cnt = T+1; multi-GPU m-sharding: this device owns m with m % m_div == part).
// Entries with m above the row's Fourier truncation are never written and never read (fft_kernel.hip).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "device_structs.h"

namespace atlas_amd {
namespace trans {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// scalar type of the transform: double (v_mfma_f64_16x16x4_f64) or float (v_mfma_f32_16x16x4_f32, BASELINE config C5).
// Both MFMAs take A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] from lane l; the result tile differs: the f64 variant
// keeps rows (l>>4) + 4*reg in a lane, the f32 variant rows 4*(l>>4) + reg (tools/probe/probe_f32_gfx950.hip).
template <class Real>
struct RealTraits;
template <>
struct RealTraits<double> {
    typedef d4 acc_t;
    typedef double2 vec_t;                  // 16-byte staging load
    static constexpr int EPL      = 2;      // elements per 16-byte load
    static constexpr int BANK_MOD = 32;     // LDS row strides are == 16 modulo this many elements (conflict-free)
    __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row_of(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct RealTraits<float> {
    typedef f4 acc_t;
    typedef float4 vec_t;
    static constexpr int EPL      = 4;
    static constexpr int BANK_MOD = 64;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

constexpr int KB   = LEG_KB_DEV;  // 8 total wavenumbers per stage
constexpr int BN   = LEG_BN_DEV;  // 64 latitudes per item
constexpr int PSTR = BN + 16;     // LDS row stride of the P stage (== 16 mod 32 doubles: conflict-free ds_read_b64)

// RTW = 16-column tiles per wave, NRG = column groups (of RTW tiles) per workgroup; a workgroup has 4*NRG waves:
// wave w handles latitude tile (w & 3) and column group (w >> 2).
template <int RTW, int NRG, class Real = double>
struct LegLds {
    static constexpr int NTHR   = 256 * NRG;
    static constexpr int SCOLS  = 16 * RTW * NRG;
    static constexpr int BM     = RealTraits<Real>::BANK_MOD;
    static constexpr int SSTR   = SCOLS + ((16 - SCOLS % BM) + BM) % BM;  // smallest stride >= SCOLS that is == 16 mod BM
    static constexpr int P_ELEM = 2 * KB * PSTR;
    static constexpr int S_ELEM = 2 * KB * SSTR;
    static constexpr int STAGE  = P_ELEM + S_ELEM;
    static constexpr int BYTES  = 2 * STAGE * (int)sizeof(Real);  // double buffered
};

template <int RTW, int NRG, class Real>
__global__ void __launch_bounds__(256 * NRG, (NRG >= 3 ? 6 : 2)) legendre_kernel(LegendreParamsT<Real> p) {
    using L  = LegLds<RTW, NRG, Real>;
    using RT = RealTraits<Real>;
    using acc_t = typename RT::acc_t;
    using vec_t = typename RT::vec_t;
    constexpr int NTHR = L::NTHR;
    constexpr int EPL  = RT::EPL;
    extern __shared__ double lds_raw[];
    Real* lds = reinterpret_cast<Real*>(lds_raw);

    // block -> (item, column chunk).  Hardware places block b on XCD b % 8; the chunks of one item are given to
    // blocks b, b+8, b+16, ... (same XCD, dispatched back to back) so that a later chunk finds the item's P block
    // in that XCD's L2, and items keep the XCD their list position implies (trans_plan.cpp).  Speed heuristic only.
    const int nchunks   = p.nchunks_run;
    const int bx        = blockIdx.x & 7;
    const int bq        = blockIdx.x >> 3;
    const int chunk     = p.chunk0 + bq % nchunks;
    const int item_slot = (bq / nchunks) * 8 + bx;
    if (item_slot >= p.nitems) {
        return;
    }
    const LegendreItemDev it = p.items[item_slot];
    if (it.m < 0) {
        return;
    }
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lt   = wave & 3;   // latitude tile of this wave
    const int rg   = wave >> 2;  // r-tile group of this wave
    const int m    = it.m;
    const int T    = p.T;
    const int nf   = p.nf;
    const int trc  = p.trc_in;
    const int r0   = chunk * L::SCOLS;         // first interleaved column of this chunk
    const int TL   = T + 1;                    // truncation of the table
    // largest n <= T+1 of each parity (n-m even: sym)
    const int ntop0 = TL - ((TL - m) & 1);
    const int ntop1 = TL - 1 + ((TL - m) & 1);
    const int nmax  = trc < TL ? trc : TL;  // highest n present in the input spectra
    const bool m_ok = m < trc;              // TransLocal.cc:982  (jm < truncation)
    const long long ioff = (long long)(2 * trc + 3 - m) * m / 2 * nf * 2;
    const Real* __restrict__ sp = p.sp + ioff;
    const Real* __restrict__ Pb = p.P + it.p_off;
    const int nstage = it.kpad / KB;

    acc_t acc[2][RTW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < RTW; ++j) acc[q][j] = acc_t{0, 0, 0, 0};

    // ---- staging: registers for one stage ----
    constexpr int PLOADS = 1024 / EPL;  // 16-byte loads per P stage (1024 elements: 2 parities x 8 k x 64 latitudes)
    constexpr int PPT    = NTHR >= PLOADS ? 1 : PLOADS / NTHR;  // loads per thread
    const bool p_loader  = NTHR <= PLOADS || tid < PLOADS;      // with more threads than loads only the first ones stage P
    vec_t preg[PPT];
    Real sreg[RTW];   // S: RTW elements per thread (16 rows x SCOLS columns per stage)
    // P element ids EPL*(tid + NTHR*i):  parity = e / 512, k = (e % 512) / 64, c = e % 64
    const Real* pg[PPT];
    int plds[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int pe   = (EPL * (tid + NTHR * i)) & 1023;
        const int ppar = pe >> 9, pk = (pe & 511) >> 6, pc = pe & 63;
        pg[i]          = Pb + (long long)ppar * it.kpad * BN + pk * BN + pc;
        plds[i]        = ppar * (KB * PSTR) + pk * PSTR + pc;
    }

    // Spectra operand: element i of this thread is row (parity, k) = q / SCOLS, column q % SCOLS of the stage tile.
    // Its wavenumber n = n0 - 2*KB*s walks down a column of the reference layout, so the address is a running pointer;
    // columns beyond the last field (and every column when m >= truncation) point at a zero in device memory with
    // step 0, which removes the per-stage select.  Only the first/last stages of an item contain rows with n outside
    // [m, nmax] (K padding): those run the bounds-checked path, all others three plain loads.  (The loop used to spend
    // 4.7 VALU instructions per MFMA on rebuilding addresses and masks; MFMA and VALU issue contend.)
    const Real* sptr[RTW];
    long long sstep[RTW];
    int sn0[RTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i) {
        const int q     = tid + NTHR * i;
        const int row   = q / L::SCOLS;  // 0..15 : parity*8 + k
        const int col   = q - row * L::SCOLS;
        const int r     = r0 + col;
        const int f     = r >> 1, im = r & 1;
        const bool fok  = m_ok && f < nf;
        sn0[i]          = ((row >> 3) ? ntop1 : ntop0) - 2 * (row & 7);
        sptr[i]         = fok ? sp + ((long long)(sn0[i] - m) * 2 * nf + im * nf + f) : p.zero;
        sstep[i]        = fok ? (long long)2 * KB * 2 * nf : 0;
    }
    // stages whose 16 rows are all inside [m, nmax]
    const int ntop_hi = ntop0 > ntop1 ? ntop0 : ntop1, ntop_lo = ntop0 < ntop1 ? ntop0 : ntop1;
    const int s_ff    = ntop_hi > nmax ? (ntop_hi - nmax + 2 * KB - 1) / (2 * KB) : 0;
    const int s_lf    = ntop_lo - 2 * (KB - 1) - m >= 0 ? (ntop_lo - 2 * (KB - 1) - m) / (2 * KB) : -1;

    auto load_stage = [&](int s) {  // must be called for s = 0, 1, 2, ... in order
        if (p_loader) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                preg[i] = *reinterpret_cast<const vec_t*>(pg[i]);
                pg[i] += KB * BN;
            }
        }
        if (s >= s_ff && s <= s_lf) {
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                sreg[i] = *sptr[i];
                sptr[i] -= sstep[i];
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                const int n = sn0[i] - 2 * KB * s;
                Real v      = 0;
                if (n >= m && n <= nmax) {
                    v = *sptr[i];
                }
                sreg[i] = v;
                sptr[i] -= sstep[i];
            }
        }
    };
    auto store_stage = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        Real* base        = lds + buf * L::STAGE;
        if (p_loader) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                *reinterpret_cast<vec_t*>(base + plds[i]) = preg[i];
            }
        }
        Real* sb = base + L::P_ELEM;
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
            const int q   = tid + NTHR * i;
            const int row = q / L::SCOLS;
            const int col = q - row * L::SCOLS;
            sb[row * L::SSTR + col] = sreg[i];
        }
    };

    load_stage(0);
    store_stage(std::integral_constant<int, 0>{});
    __syncthreads();

    const int a_off = (lane >> 4) * PSTR + lt * 16 + (lane & 15);
    const int b_off = (lane >> 4) * L::SSTR + rg * RTW * 16 + (lane & 15);
    const bool lat_active = lt * 16 < it.nrows;

    // one stage from LDS buffer `buf` (compile-time: the fragment offsets become instruction immediates)
    auto run_stage = [&](int s, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (s + 1 < nstage) {
            load_stage(s + 1);
        }
        const Real* base = lds + buf * L::STAGE;
        if (lat_active) {  // a wave whose 16 latitudes are all beyond the item's last row only helps with the staging
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const Real* pb = base + par * (KB * PSTR) + a_off;
            const Real* sb = base + L::P_ELEM + par * (KB * L::SSTR) + b_off;
#pragma unroll
            for (int ks = 0; ks < KB / 4; ++ks) {
                const Real a = pb[ks * 4 * PSTR];
#pragma unroll
                for (int j = 0; j < RTW; ++j) {
                    const Real b = sb[ks * 4 * L::SSTR + j * 16];
                    acc[par][j]  = RT::mma(a, b, acc[par][j]);
                }
            }
        }
        }
        if (s + 1 < nstage) {
            store_stage(std::integral_constant<int, buf ^ 1>{});
        }
        __syncthreads();
    };
    for (int s = 0; s < nstage; s += 2) {
        run_stage(s, std::integral_constant<int, 0>{});
        if (s + 1 < nstage) {
            run_stage(s + 1, std::integral_constant<int, 1>{});
        }
    }

    // ---- epilogue: merge hemispheres and store ----
    const int nlats  = p.nlats;
    const int jleg0  = p.nlat0[m] + it.tile * BN;
    const long long RP = p.RP;
    const int ml       = m / p.m_div;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = lt * 16 + RT::row_of(lane, g);
        if (c < it.nrows) {
            const int jn = jleg0 + c;
            const int js = nlats - 1 - jn;
            const bool st_n = jn != js && jn >= p.row_begin && jn < p.row_end;
            const bool st_s = js >= p.row_begin && js < p.row_end;
            Real* fn     = p.F + ((long long)(jn - p.row_begin) * p.m_cnt + ml) * RP;
            Real* fs     = p.F + ((long long)(js - p.row_begin) * p.m_cnt + ml) * RP;
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const int r = r0 + (rg * RTW + j) * 16 + (lane & 15);
                if (r < RP) {
                    Real sy = acc[0][j][g], as = acc[1][j][g];
                    if (m == 0 && (r & 1)) {  // n_imag = 1 for m = 0 (TransLocal.cc:953)
                        sy = 0;
                        as = 0;
                    }
                    if (st_n) {
                        fn[r] = sy + as;
                    }
                    if (st_s) {
                        fs[r] = sy - as;  // for an equator row the southern value wins (TransLocal.cc:1056-1068 runs last)
                    }
                }
            }
        }
    }
}

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt = imm[3:0] + imm[15:14], expcnt imm[6:4], lgkmcnt imm[11:8])
#define AA_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#define AA_WAIT_LGKMCNT0() __builtin_amdgcn_s_waitcnt((15) | (3 << 14) | (7 << 4) | (0 << 8))

// ---- legendre_kernel<3, 2, double> without vector-ALU work in the stage loop ----------------------------------------
// fp64 MFMAs execute on the SIMD's double-precision vector datapath: every VALU instruction a wavefront issues between
// its MFMAs takes that datapath away from them (measured: the MFMA stream alone runs at the part's sustained matrix rate,
// 7.5 ms for the TL1279/O1280/137 launch; legendre_kernel<3, 2> issues 1.75 VALU instructions per MFMA -- 64-bit pointer
// increments, register copies, LDS address arithmetic -- and needs 9.0 ms).  Same tiling, LDS layout, staging scheme and
// summation order as legendre_kernel<3, 2> (results are bit-identical); the differences:
//   * global addresses are a scalar stage base (SALU increments) + a per-lane byte offset fixed before the loop, the
//     loads are inline asm in the saddr form (the compiler turns the same C++ into 64-bit VALU adds per lane);
//   * fragment reads are software-pipelined one MFMA group ahead instead of a whole stage (fewer live registers, no
//     address arithmetic: every LDS offset is an immediate);
//   * columns beyond the last field load the last field instead of a zero word (their products land in the padding
//     columns of F), which removes the per-lane pointer/step pairs.
__global__ void __launch_bounds__(512, 4) legendre_kernel_lean(LegendreParams p) {
    using L  = LegLds<3, 2, double>;
    using RT = RealTraits<double>;
    using acc_t = typename RT::acc_t;
    constexpr int RTW = 3, NTHR = 512;
    // dynamic LDS: L::BYTES, addressed from 0 by the inline asm below

    const int nchunks   = p.nchunks_run;
    const int bx        = blockIdx.x & 7;
    const int bq        = blockIdx.x >> 3;
    const int chunk     = p.chunk0 + bq % nchunks;
    const int item_slot = (bq / nchunks) * 8 + bx;
    if (item_slot >= p.nitems) {
        return;
    }
    const LegendreItemDev it = p.items[item_slot];
    if (it.m < 0) {
        return;
    }
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lt   = wave & 3;
    const int rg   = wave >> 2;
    const int m    = it.m;
    const int nf   = p.nf;
    const int trc  = p.trc_in;
    const int r0   = chunk * L::SCOLS;
    const int TL   = p.T + 1;
    const int ntop0 = TL - ((TL - m) & 1);
    const int ntop1 = TL - 1 + ((TL - m) & 1);
    const int nmax  = trc < TL ? trc : TL;
    const bool m_ok = m < trc;
    const long long ioff = (long long)(2 * trc + 3 - m) * m / 2 * nf * 2;
    const double* __restrict__ sp = p.sp + ioff;
    const double* __restrict__ Pb = p.P + it.p_off;
    const int nstage = it.kpad / KB;
    const int ntop_hi = ntop0 > ntop1 ? ntop0 : ntop1, ntop_lo = ntop0 < ntop1 ? ntop0 : ntop1;

    acc_t acc[2][RTW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < RTW; ++j) acc[q][j] = acc_t{0, 0, 0, 0};

    // ---- staging registers of one stage: one 16-byte table load and three spectra elements per thread ----
    typedef int i4_t __attribute__((ext_vector_type(4)));
    constexpr int NSET = 3;   // register sets: the stage being written to LDS and the ones still in flight
    i4_t preg[NSET];
    double sreg[NSET][RTW];
    // table element pair 2 tid of the stage tile [2 parities][8 k][64 latitudes]
    const int pe = 2 * tid, ppar = pe >> 9, pk = (pe & 511) >> 6, pc = pe & 63;
    const unsigned pbo = 8u * (unsigned)((ppar * it.kpad + pk) * BN + pc);   // bytes from the stage's first table row
    const int plds     = ppar * (KB * PSTR) + pk * PSTR + pc;
    // spectra element q = tid + 512 i of the stage tile [16 rows (parity, k)][96 columns]
    unsigned sbo[RTW];   // bytes from the stage base (lowest wavenumber row of the stage)
    int slds[RTW];
    int sn0[RTW];        // wavenumber of the element in stage 0
    int scolo[RTW];      // im * nf + f
#pragma unroll
    for (int i = 0; i < RTW; ++i) {
        const int q   = tid + NTHR * i;
        const int row = q / L::SCOLS, col = q - row * L::SCOLS;
        const int par = row >> 3, k = row & 7;
        const int r   = r0 + col;
        int f         = r >> 1;
        f             = f < nf ? f : nf - 1;
        const int ntop = par ? ntop1 : ntop0;
        scolo[i] = (r & 1) * nf + f;
        sn0[i]   = ntop - 2 * k;
        sbo[i]   = 8u * (unsigned)((ntop - ntop_lo + 2 * (KB - 1 - k)) * 2 * nf + scolo[i]);
        slds[i]  = L::P_ELEM + row * L::SSTR + col;
    }
    // stages whose 16 rows are all inside [m, nmax]
    const int s_ff = ntop_hi > nmax ? (ntop_hi - nmax + 2 * KB - 1) / (2 * KB) : 0;
    const int s_lf = m_ok ? (ntop_lo - 2 * (KB - 1) - m >= 0 ? (ntop_lo - 2 * (KB - 1) - m) / (2 * KB) : -1) : -1;
    // scalar bases of the next stage to be requested (stages are requested in order)
    const double* pbase = Pb;
    const double* sbase = sp + (long long)(ntop_lo - 2 * (KB - 1) - m) * 2 * nf;
    const long long sstride = (long long)4 * KB * nf;   // 2 KB wavenumbers down

    unsigned szero[NSET] = {0, 0, 0};   // bit i: element i of the staged stage is outside [m, nmax]
    bool sedge[NSET]     = {false, false, false};
    // stage t travels through register set t % 3 into LDS buffer t & 1; it is requested three stages before it is used
    auto load_stage = [&](int s, auto setc) {   // must be called for s = 0, 1, 2, ... in order
        constexpr int SET = decltype(setc)::value;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(preg[SET]) : "v"(pbo), "s"(pbase) : "memory");
        if (s >= s_ff && s <= s_lf) {
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(sreg[SET][i]) : "v"(sbo[i]), "s"(sbase) : "memory");
            }
            sedge[SET] = false;
        }
        else {
            unsigned z = 0;
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                const int n  = sn0[i] - 2 * KB * s;
                const int nc = n < m ? m : (n > nmax ? nmax : n);
                // one load in every case; outside the spectra of this m read the table
                const double* src = m_ok ? sp + (long long)(nc - m) * 2 * nf + scolo[i] : p.P;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sreg[SET][i]) : "v"(src) : "memory");
                z |= (unsigned)(!m_ok || nc != n) << i;
            }
            szero[SET] = z;
            sedge[SET] = true;
        }
        pbase += KB * BN;
        sbase -= sstride;
    };
    const unsigned plds_b = 8u * (unsigned)plds;
    unsigned slds_b[RTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i) {
        slds_b[i] = 8u * (unsigned)slds[i];
    }
    constexpr int LPS = 1 + RTW;   // loads per stage and thread
    auto store_stage = [&](auto bufc, auto setc, int younger_in_flight) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int SET = decltype(setc)::value;
        // the loads above are invisible to the compiler's own s_waitcnt placement: count them
        if (younger_in_flight >= 2) {
            AA_WAIT_VMCNT(2 * LPS);
        }
        else if (younger_in_flight == 1) {
            AA_WAIT_VMCNT(LPS);
        }
        else {
            AA_WAIT_VMCNT(0);
        }
        // stores as asm volatile as well: they stay behind the wait, and their offsets are immediates
        {
            const i4_t pv     = preg[SET];
            const unsigned la = plds_b;
            asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(la), "v"(pv), "n"(buf * L::STAGE * 8) : "memory");
        }
        if (sedge[SET]) {   // uniform; both arms are asm volatile so that this stays a branch (as selects it is 15 VALU per stage)
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                const double v    = ((szero[SET] >> i) & 1) ? 0. : sreg[SET][i];
                const unsigned la = slds_b[i];
                asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(la), "v"(v), "n"(buf * L::STAGE * 8) : "memory");
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                const double v    = sreg[SET][i];
                const unsigned la = slds_b[i];
                asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(la), "v"(v), "n"(buf * L::STAGE * 8) : "memory");
            }
        }
        AA_WAIT_LGKMCNT0();   // the compiler does not see these stores: they must have landed before the barrier
    };

    load_stage(0, std::integral_constant<int, 0>{});
    if (nstage > 1) {
        load_stage(1, std::integral_constant<int, 1>{});
    }
    if (nstage > 2) {
        load_stage(2, std::integral_constant<int, 2>{});
    }
    store_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, (nstage < 3 ? nstage : 3) - 1);
    __syncthreads();

    const int a_off = (lane >> 4) * PSTR + lt * 16 + (lane & 15);
    const int b_off = L::P_ELEM + (lane >> 4) * L::SSTR + rg * RTW * 16 + (lane & 15);
    const bool lat_active = lt * 16 < it.nrows;

    // four steps (parity, 4 wavenumbers) of one A and three B fragments and three MFMAs; the fragments of step t + 1
    // are requested before the MFMAs of step t
    const unsigned a_b = 8u * (unsigned)a_off, b_b = 8u * (unsigned)b_off;
    auto mma_steps = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int NKS = KB / 4;
        double a[2], b[2][RTW];
        // fragment reads as asm with immediate offsets (the compiler pairs them into ds_read2 and pays a VALU add per pair
        // for the base); their completion is counted by hand: 1 + RTW reads per step, one step in flight
        auto fetch = [&](int t, int slot) {
            const int par = t / NKS, ks = t % NKS;
            const unsigned ab = a_b, bb = b_b;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[slot]) : "v"(ab), "n"((buf * L::STAGE + (par * KB + ks * 4) * PSTR) * 8) : "memory");
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                asm volatile("ds_read_b64 %0, %1 offset:%2"
                             : "=v"(b[slot][j])
                             : "v"(bb), "n"((buf * L::STAGE + (par * KB + ks * 4) * L::SSTR + j * 16) * 8)
                             : "memory");
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 2 * NKS; ++t) {
            if (t + 1 < 2 * NKS) {
                fetch(t + 1, (t + 1) & 1);
                __builtin_amdgcn_s_waitcnt((15) | (3 << 14) | (7 << 4) | ((1 + RTW) << 8));   // lgkmcnt(1 + RTW)
            }
            else {
                AA_WAIT_LGKMCNT0();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                acc[t / NKS][j] = RT::mma(a[t & 1], b[t & 1][j], acc[t / NKS][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // stage s is multiplied from LDS buffer s & 1; beside it stage s + 3 is requested into the register set that stage s
    // left (s % 3), and afterwards stage s + 1 goes from its set to the other buffer
    auto run_stage = [&](int s, auto bufc, auto setc) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int SET = decltype(setc)::value;
        if (s + 3 < nstage) {
            load_stage(s + 3, setc);
        }
        if (lat_active) {
            mma_steps(bufc);
        }
        if (s + 1 < nstage) {
            const int last = s + 3 < nstage ? s + 3 : nstage - 1;   // last stage requested so far
            store_stage(std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, (SET + 1) % 3>{}, last - (s + 1));
        }
        __syncthreads();
    };
    for (int s = 0; s < nstage; s += 6) {
        run_stage(s, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if (s + 1 < nstage) {
            run_stage(s + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
        if (s + 2 < nstage) {
            run_stage(s + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        }
        if (s + 3 < nstage) {
            run_stage(s + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        }
        if (s + 4 < nstage) {
            run_stage(s + 4, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        }
        if (s + 5 < nstage) {
            run_stage(s + 5, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        }
    }

    // ---- epilogue: merge hemispheres and store (as legendre_kernel) ----
    const int nlats  = p.nlats;
    const int jleg0  = p.nlat0[m] + it.tile * BN;
    const long long RP = p.RP;
    const int ml       = m / p.m_div;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = lt * 16 + RT::row_of(lane, g);
        if (c < it.nrows) {
            const int jn = jleg0 + c;
            const int js = nlats - 1 - jn;
            const bool st_n = jn != js && jn >= p.row_begin && jn < p.row_end;
            const bool st_s = js >= p.row_begin && js < p.row_end;
            double* fn     = p.F + ((long long)(jn - p.row_begin) * p.m_cnt + ml) * RP;
            double* fs     = p.F + ((long long)(js - p.row_begin) * p.m_cnt + ml) * RP;
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const int r = r0 + (rg * RTW + j) * 16 + (lane & 15);
                if (r < RP) {
                    double sy = acc[0][j][g], as = acc[1][j][g];
                    if ((m == 0 && (r & 1)) || r >= 2 * nf) {   // n_imag = 1 for m = 0; padding columns hold zeros
                        sy = 0;
                        as = 0;
                    }
                    if (st_n) {
                        fn[r] = sy + as;
                    }
                    if (st_s) {
                        fs[r] = sy - as;
                    }
                }
            }
        }
    }
}

static hipError_t launch_lean(LegendreParams p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<3, 2, double>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&legendre_kernel_lean),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
        if (e != hipSuccess) {
            return e;
        }
        attr_set = true;
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL(legendre_kernel_lean, dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES, stream, p);
    return hipGetLastError();
}

// ---- legendre_kernel<3, 2, double> with both operands staged by LDS-DMA -------------------------------------------------
// Same tiling, MFMA roles and summation order as legendre_kernel<3, 2> (bit-identical results).  No staging registers and no
// staging instructions besides two or three global_load_lds_dwordx4 per wavefront and stage:
//   * table: wavefront w moves rows 2w, 2w+1 of the stage tile [2 parities x 8 wavenumbers][64 latitudes];
//   * spectra: the stage tile [16 rows][96 columns] in 16-byte units.  Real and imaginary part of a field lie nf doubles
//     apart in memory, neighbouring fields of the same part are contiguous: the columns of the tile are the real parts of
//     the chunk's 48 fields followed by their imaginary parts, so that consecutive lanes of a DMA instruction read
//     consecutive memory (interleaved runs are not coalesced: 1.2 ms of MFMA time for this launch).  Column group rg of the
//     workgroup then holds part rg, and the epilogue stores every other double of F.  A unit
//     whose second field does not exist (odd nf) reads 8 bytes into the following run, which is inside the array (the last
//     wavenumber's block is never transformed, TransLocal.cc:982) and lands in a padding column.
// Rings of 4 stage slots for each operand (80 KB per workgroup, 2 workgroups per CU), requested three stages ahead, rows of
// exactly 64 / 96 doubles with the 16-element groups of odd rows swapped pairwise (conflict-free fragment reads, see the
// role-split kernel below).  LDS reads are inline asm with immediate offsets: the compiler, which drains vmcnt before any
// LDS access of a wavefront with LDS-DMA in flight, does not see them; waits are counted by hand; barriers are raw.
__global__ void __launch_bounds__(512, 4) legendre_kernel_dma(LegendreParams p) {
    using RT = RealTraits<double>;
    using acc_t = typename RT::acc_t;
    constexpr int RTW = 3, KS = 8, RING = 4, SC = 96;
    constexpr int P_SLOT = 2 * KS * BN;          // doubles
    constexpr int S_SLOT = 2 * KS * SC;
    constexpr int S_BASE = RING * P_SLOT;

    const int nchunks   = p.nchunks_run;
    const int bx        = blockIdx.x & 7;
    const int bq        = blockIdx.x >> 3;
    const int chunk     = p.chunk0 + bq % nchunks;
    const int item_slot = (bq / nchunks) * 8 + bx;
    if (item_slot >= p.nitems) {
        return;
    }
    const LegendreItemDev it = p.items[item_slot];
    if (it.m < 0) {
        return;
    }
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lt   = wave & 3;
    const int rg   = wave >> 2;
    const int m    = it.m;
    const int nf   = p.nf;
    const int trc  = p.trc_in;
    const int r0   = chunk * SC;
    const int TL   = p.T + 1;
    const int ntop0 = TL - ((TL - m) & 1);
    const int ntop1 = TL - 1 + ((TL - m) & 1);
    const int nmax  = trc < TL ? trc : TL;
    const bool m_ok = m < trc;
    const long long ioff = (long long)(2 * trc + 3 - m) * m / 2 * nf * 2;
    const double* __restrict__ sp = p.sp + ioff;
    const double* __restrict__ Pb = p.P + it.p_off;
    const int kpad   = it.kpad;
    const int nstage = kpad / KS;

    // ---- staging: this wavefront's DMA instructions of a stage ----
    // table rows 2 wave, 2 wave + 1: lane l delivers row 2 wave + (l >> 5), LDS doubles 2 (l & 31) .. of it, i.e. latitudes
    // c, c + 1 with c = 2 (l & 31) ^ (16 on the odd row)
    const unsigned pvo = 8u * (unsigned)((lane >> 5) * BN + ((2 * (lane & 31)) ^ ((lane >> 5) << 4)));
    const int ppar = wave >> 2, pk0 = (2 * wave) & 7;
    const double* pbase = Pb + ((long long)ppar * kpad + pk0) * BN;   // + s KS BN per stage (requested in order)
    // spectra instruction d = 3 a + cc covers units 64 d .. 64 d + 63 of [16 rows][48 units]: rows 4 a .. 4 a + 3
    auto spectra_lane_offset = [&](int cc) {   // bytes from the lowest-wavenumber row (local row 3) of the row group
        const int e    = 64 * cc + lane;
        const int rl   = e / 48, up = e - rl * 48;            // local row 0..3, physical unit
        const int cl   = (((up >> 3) ^ (rl & 1)) << 4) + ((2 * up) & 15);   // logical column (even)
        int f          = (r0 >> 1) + cl % 48;                 // columns 0..47: real parts of 48 fields, 48..95: imaginary parts
        const int im   = cl / 48;
        f              = f < nf ? f : nf - 1;
        return 8u * (unsigned)((3 - rl) * 4 * nf + im * nf + f);
    };
    auto spectra_lane_row = [&](int cc) { return (64 * cc + lane) / 48; };
    const int sd0 = wave, sd1 = wave + 8;                     // instructions of this wavefront (sd1 only for wave < 4)
    const bool two = wave < 4;
    const unsigned svo0 = spectra_lane_offset(sd0 % 3), svo1 = spectra_lane_offset(sd1 % 3);
    const int srl0 = spectra_lane_row(sd0 % 3), srl1 = spectra_lane_row(sd1 % 3);
    const long long sstride = (long long)4 * KS * nf;         // doubles per stage (2 KS wavenumbers down)
    // lowest wavenumber of row group a (rows 4a .. 4a+3 of the tile) in stage 0; its row base; requested in order
    auto group_nlow0 = [&](int a) { return ((a >> 1) ? ntop1 : ntop0) - 2 * (4 * (a & 1) + 3); };
    int nlow0 = group_nlow0(sd0 / 3), nlow1 = group_nlow0(sd1 / 3);
    const double* sb0 = sp + (long long)(nlow0 - m) * 2 * nf;
    const double* sb1 = sp + (long long)(nlow1 - m) * 2 * nf;
    int issued = 0;   // stages requested so far

    auto dma_saddr = [&](unsigned voff, const double* base, unsigned lds_bytes) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_bytes) : "memory");
    };
    auto dma_vaddr = [&](const void* addr, unsigned lds_bytes) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(addr), "s"(lds_bytes) : "memory");
    };
    auto spectra_dma = [&](int s, int d, unsigned voff, int rl, const double* base, int nlow_s0, unsigned slot_bytes) {
        const int nlow = nlow_s0 - 2 * KS * s;
        const unsigned dst = slot_bytes + (unsigned)(d * 1024);
        if (m_ok && nlow >= m && nlow + 6 <= nmax) {   // uniform: the four rows of the group lie inside [m, nmax]
            dma_saddr(voff, base, dst);
        }
        else {
            const int n      = nlow + 2 * (3 - rl);
            const bool valid = m_ok && n >= m && n <= nmax;
            const char* a    = valid ? reinterpret_cast<const char*>(base) + voff : reinterpret_cast<const char*>(p.zero);
            dma_vaddr(a, dst);
        }
    };
    auto issue_stage = [&](int s) {   // s == issued
        const unsigned ring = (unsigned)(s & (RING - 1));
        if (!(p.abl & 2)) {
            dma_saddr(pvo, pbase, 8u * (ring * P_SLOT + (unsigned)(2 * wave * BN)));
        }
        if (!(p.abl & 4)) {
            spectra_dma(s, sd0, svo0, srl0, sb0, nlow0, 8u * (S_BASE + ring * S_SLOT));
            if (two) {
                spectra_dma(s, sd1, svo1, srl1, sb1, nlow1, 8u * (S_BASE + ring * S_SLOT));
            }
        }
        pbase += KS * BN;
        sb0 -= sstride;
        sb1 -= sstride;
        issued = s + 1;
    };
    // this wavefront's DMAs of stage t have landed when at most those of the stages requested after t are outstanding
    auto wait_landed = [&](int t) {
        const int later = issued - 1 - t;
        if (two) {
            if (later >= 2) {
                AA_WAIT_VMCNT(6);
            }
            else if (later == 1) {
                AA_WAIT_VMCNT(3);
            }
            else {
                AA_WAIT_VMCNT(0);
            }
        }
        else {
            if (later >= 2) {
                AA_WAIT_VMCNT(4);
            }
            else if (later == 1) {
                AA_WAIT_VMCNT(2);
            }
            else {
                AA_WAIT_VMCNT(0);
            }
        }
    };

    acc_t acc[2][RTW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < RTW; ++j) acc[q][j] = acc_t{0, 0, 0, 0};
    const int x        = (lane >> 4) & 1;
    const unsigned a_b = 8u * (unsigned)((lane >> 4) * BN + ((lt ^ x) << 4) + (lane & 15));
    unsigned b_b[RTW];
#pragma unroll
    for (int j = 0; j < RTW; ++j) {
        b_b[j] = 8u * (unsigned)(S_BASE + (lane >> 4) * SC + (((rg * RTW + j) ^ x) << 4) + (lane & 15));
    }
    const bool lat_active = lt * 16 < it.nrows;

    auto mma_steps = [&](auto slotc) {
        constexpr int SLOT = decltype(slotc)::value;
        constexpr int NKS  = KS / 4;
        double a[2], b[2][RTW];
        auto fetch = [&](int t, int set) {
            const int par = t / NKS, ks = t % NKS;
            const unsigned ab = a_b;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[set]) : "v"(ab), "n"((SLOT * P_SLOT + (par * KS + ks * 4) * BN) * 8) : "memory");
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const unsigned bb = b_b[j];
                asm volatile("ds_read_b64 %0, %1 offset:%2"
                             : "=v"(b[set][j])
                             : "v"(bb), "n"((SLOT * S_SLOT + (par * KS + ks * 4) * SC) * 8)
                             : "memory");
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 2 * NKS; ++t) {
            if (t + 1 < 2 * NKS) {
                fetch(t + 1, (t + 1) & 1);
                __builtin_amdgcn_s_waitcnt((15) | (3 << 14) | (7 << 4) | ((1 + RTW) << 8));   // lgkmcnt(1 + RTW)
            }
            else {
                AA_WAIT_LGKMCNT0();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                acc[t / NKS][j] = RT::mma(a[t & 1], b[t & 1][j], acc[t / NKS][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int t = 0; t < 3 && t < nstage; ++t) {
        issue_stage(t);
    }
    wait_landed(0);
    __builtin_amdgcn_s_barrier();   // stage 0 is in LDS
    auto run_stage = [&](int s, auto slotc) {
        if (s + 3 < nstage) {
            issue_stage(s + 3);     // slot (s + 3) % 4 was read during stage s - 1
        }
        if (lat_active) {
            mma_steps(slotc);
        }
        if (s + 1 < nstage) {
            wait_landed(s + 1);
        }
        __builtin_amdgcn_s_barrier();
    };
    for (int s = 0; s < nstage; s += 4) {
        run_stage(s, std::integral_constant<int, 0>{});
        if (s + 1 < nstage) {
            run_stage(s + 1, std::integral_constant<int, 1>{});
        }
        if (s + 2 < nstage) {
            run_stage(s + 2, std::integral_constant<int, 2>{});
        }
        if (s + 3 < nstage) {
            run_stage(s + 3, std::integral_constant<int, 3>{});
        }
    }

    // ---- epilogue: merge hemispheres and store; column c of the tile is field 2 (c / 4) + (c & 1), part (c >> 1) & 1 ----
    const int nlats  = p.nlats;
    const int jleg0  = p.nlat0[m] + it.tile * BN;
    const long long RP = p.RP;
    const int ml       = m / p.m_div;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = lt * 16 + RT::row_of(lane, g);
        if (c < it.nrows) {
            const int jn = jleg0 + c;
            const int js = nlats - 1 - jn;
            const bool st_n = jn != js && jn >= p.row_begin && jn < p.row_end;
            const bool st_s = js >= p.row_begin && js < p.row_end;
            double* fn     = p.F + ((long long)(jn - p.row_begin) * p.m_cnt + ml) * RP;
            double* fs     = p.F + ((long long)(js - p.row_begin) * p.m_cnt + ml) * RP;
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const int r  = r0 + 2 * (16 * j + (lane & 15)) + rg;   // column group rg holds part rg of fields 16 j + ..
                if (r < RP) {
                    double sy = acc[0][j][g], as = acc[1][j][g];
                    if ((m == 0 && (r & 1)) || r >= 2 * nf) {   // n_imag = 1 for m = 0; padding columns hold zeros
                        sy = 0;
                        as = 0;
                    }
                    if (st_n) {
                        fn[r] = sy + as;
                    }
                    if (st_s) {
                        fs[r] = sy - as;
                    }
                }
            }
        }
    }
}

static hipError_t launch_dma(LegendreParams p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    constexpr int BYTES = 4 * (2 * 8 * BN + 2 * 8 * 96) * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&legendre_kernel_dma),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
        if (e != hipSuccess) {
            return e;
        }
        attr_set = true;
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = std::getenv("ATLAS_AMD_LEG_ABLATE") ? atoi(std::getenv("ATLAS_AMD_LEG_ABLATE")) : 0;
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL(legendre_kernel_dma, dim3(slots * nrun * 8), dim3(512), BYTES, stream, p);
    return hipGetLastError();
}

// ---- role-split variant for the 96-column workgroup (RTW = 3, NRG = 2, fp64) ------------------------------------------
// Same tiling, MFMA roles and summation order as legendre_kernel<3, 2> (results are bit-identical), different division of
// labour: the workgroup has 10 wavefronts, 8 that only read fragments from LDS and issue MFMAs, and 2 that only stage.
// Measured on the 8 + 2 prototypes (profiles/r02_mfma_sustained.txt): MFMA wavefronts that are fed reach the sustained
// matrix rate of the part (7.5 ms for this launch against 9.0 ms of legendre_kernel<3, 2>), whose own wavefronts hold a
// stage of loads in registers while they multiply and so never have more than one stage of memory latency covered.
// Staging wavefront j owns parity j:
//   * table rows by LDS-DMA (global_load_lds_dwordx4, no registers) into a ring of PRING stage slots, three stages ahead;
//   * spectra through registers (real and imaginary part of a field lie nf doubles apart: 8-byte loads; 4-byte LDS-DMA
//     of the same data costs 22 cycles per 256 bytes and CU, 5.4 ms for this launch), two register sets, loaded three
//     stages ahead and written to a double buffer one stage ahead.
// The barriers are raw s_barrier with explicit s_waitcnt: __syncthreads() would wait for every outstanding load.
// LDS layout without padding: rows of exactly 64 latitudes / 96 columns, the 16-element groups of every odd wavenumber
// row swapped pairwise (group g -> g ^ 1).  A half-wavefront of a fragment read covers two consecutive rows x 16
// elements; the swap puts them into different halves of the 64-bank window, which is what the 16-double row padding of
// legendre_kernel does.  Per workgroup PRING x 8 KB + 3 x 12 KB = 68 KB.
// kpad is a multiple of KS = 8.
template <int KS, int PRING>
struct LegLds2 {
    static constexpr int NCOMP  = 8;    // MFMA wavefronts
    static constexpr int NLOAD  = 4;    // staging wavefronts: one for the table, three for the spectra
    static constexpr int NTHR   = 64 * (NCOMP + NLOAD);
    static constexpr int SCOLS  = 96;
    static constexpr int P_ELEM = 2 * KS * BN;
    static constexpr int S_ELEM = 2 * KS * SCOLS;
    static constexpr int S_BASE = PRING * P_ELEM;
    static constexpr int SRING  = 3;    // spectra stage slots = register sets of the spectra wavefronts
    static constexpr int BYTES  = (PRING * P_ELEM + SRING * S_ELEM) * (int)sizeof(double);
};

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;


template <int KS, int PRING>
__global__ void __launch_bounds__(768, 6) legendre_kernel_v2(LegendreParams p) {
    using L  = LegLds2<KS, PRING>;
    using RT = RealTraits<double>;
    using acc_t = typename RT::acc_t;
    constexpr int RTW = 3;
    static_assert(PRING == 4 || PRING == 5, "the waits below are written for three or four stages in flight");
    extern __shared__ double lds_raw[];
    double* lds = lds_raw;

    const int nchunks   = p.nchunks_run;
    const int bx        = blockIdx.x & 7;
    const int bq        = blockIdx.x >> 3;
    const int chunk     = p.chunk0 + bq % nchunks;
    const int item_slot = (bq / nchunks) * 8 + bx;
    if (item_slot >= p.nitems) {
        return;
    }
    const LegendreItemDev it = p.items[item_slot];
    if (it.m < 0) {
        return;
    }
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m    = it.m;
    const int nf   = p.nf;
    const int trc  = p.trc_in;
    const int r0   = chunk * L::SCOLS;
    const int TL   = p.T + 1;
    const int ntop0 = TL - ((TL - m) & 1);
    const int ntop1 = TL - 1 + ((TL - m) & 1);
    const int nmax  = trc < TL ? trc : TL;
    const bool m_ok = m < trc;
    const long long ioff = (long long)(2 * trc + 3 - m) * m / 2 * nf * 2;
    const double* __restrict__ sp = p.sp + ((p.abl & 128) ? 0 : ioff);
    const double* __restrict__ Pb = p.P + ((p.abl & 64) ? 0 : it.p_off);
    const int kpad   = it.kpad;
    const int nstage = kpad / KS;

    if (wave >= L::NCOMP) {
        // ================= staging wavefronts =================
        // fp64 MFMAs run on the SIMD's double-precision VALU datapath: while MFMA wavefronts have work, every VALU
        // instruction of a co-resident wavefront queues behind 64-cycle MFMAs.  The staging wavefronts issue few
        // instructions and each one is on the critical path of a whole stage: they go first.
        if (p.abl & 512) {
            __builtin_amdgcn_s_setprio(1);
        }
        else if (!(p.abl & 256)) {
            __builtin_amdgcn_s_setprio(3);
        }
        if (wave == L::NCOMP) {
            // ---- table wavefront: both parities by LDS-DMA, three stages ahead.  It executes no LDS instruction: the
            // compiler drains vmcnt before any LDS access of a wavefront that has LDS-DMA in flight. ----
            // instruction q moves rows 2q, 2q+1 of the stage tile [2 KS rows][64]; lane l delivers the 16 bytes at LDS
            // position (row 2q + (l >> 5), doubles 2 (l & 31) ..), i.e. latitudes c, c+1 with c = 2 (l & 31) ^ (16 on odd rows)
            const unsigned plane = (unsigned)((lane >> 5) * BN + ((2 * (lane & 31)) ^ ((lane >> 5) << 4)));
            constexpr int DP     = KS;   // DMA instructions per stage
            static_assert(3 * DP <= 63, "vmcnt is a 6-bit counter");
            // requested in stage order: scalar running bases of the two parities, per-lane byte offset fixed; the LDS
            // destination of an instruction is M0 + 16 lane
            const double* pb0 = Pb;
            const double* pb1 = Pb + (long long)kpad * BN;
            const unsigned pbo = plane * 8u;
            auto dma_p = [&](int s) {
                if (p.abl & 2) {
                    return;
                }
                const unsigned slot = (unsigned)((s % PRING) * L::P_ELEM * 8);   // LDS bytes
#pragma unroll
                for (int q = 0; q < DP; ++q) {
                    const int par = (2 * q) / KS, k = (2 * q) % KS;
                    const double* g   = (par ? pb1 : pb0) + k * BN;
                    const unsigned ld = slot + (unsigned)(2 * q * BN * 8);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(pbo), "s"(g), "s"(ld) : "memory");
                }
                pb0 += KS * BN;
                pb1 += KS * BN;
            };
            // stage t + 1 has landed when at most the stages issued after it are outstanding
            auto wait_landed = [&](int t1, int issued_last) {
                const int later = issued_last - t1;
                if (later >= 3) {
                    AA_WAIT_VMCNT(3 * DP);
                }
                else if (later == 2) {
                    AA_WAIT_VMCNT(2 * DP);
                }
                else if (later == 1) {
                    AA_WAIT_VMCNT(DP);
                }
                else {
                    AA_WAIT_VMCNT(0);
                }
            };
            constexpr int AHEAD = PRING - 1;   // stage s + AHEAD is requested beside the MFMAs of stage s
            const int npre = nstage < AHEAD ? nstage : AHEAD;
            for (int t = 0; t < npre; ++t) {
                dma_p(t);
            }
            wait_landed(0, npre - 1);
            __builtin_amdgcn_s_barrier();   // stage 0 is in LDS
            for (int s = 0; s < nstage; ++s) {
                if (s + AHEAD < nstage) {
                    dma_p(s + AHEAD);   // its slot was read during stage s - 1
                }
                if (s + 1 < nstage) {
                    wait_landed(s + 1, s + AHEAD < nstage ? s + AHEAD : nstage - 1);
                }
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
        // ---- spectra wavefronts: through registers (real and imaginary part of a field lie nf doubles apart), three
        // register sets: stage s + 1 is written to LDS while stages s + 2 and s + 3 are in flight, then s + 4 is requested.
        // Instruction g = 8 lw + i (of 24) covers elements 64 g .. 64 g + 63 of the stage tile [2 KS rows][96 columns];
        // element e + 192 is the same column two rows on, so g % 3 fixes (row within the pair, column) and g / 3 the row
        // pair, which lies inside one parity. ----
        constexpr int NSW = L::NLOAD - 1;                 // spectra wavefronts
        constexpr int DS  = 2 * KS * L::SCOLS / 64 / NSW;   // loads per stage and wavefront
        static_assert(DS * NSW * 64 == 2 * KS * L::SCOLS && KS % 2 == 0, "");
        // The wavefronts below must not execute VALU instructions in their stage loop (each one queues behind the MFMAs
        // of the co-resident wavefronts): global addresses are a scalar row base + a per-lane byte offset fixed here,
        // LDS addresses are fixed here for both buffers, everything else is scalar arithmetic and immediates.
        unsigned sbo[3];    // bytes: (row within pair, column) part of the address, for g % 3 = 0, 1, 2
        unsigned slb[1][3]; // LDS byte address of the element in row pair 0 of buffer 0
        int krow[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int e = 64 * c + lane;
            const int k = e / L::SCOLS, col = e - k * L::SCOLS;   // k = 0 / 1
            const int r = r0 + col;
            int f       = r >> 1;
            f           = f < nf ? f : nf - 1;   // columns beyond the last field fetch the last field: their products
            sbo[c]      = 8u * (unsigned)((1 - k) * 4 * nf + (r & 1) * nf + f);   // land in the padding of F
            const int ld = k * L::SCOLS + ((((col >> 4) ^ k) << 4) | (col & 15));
            slb[0][c]   = 8u * (unsigned)(L::S_BASE + ld);
            krow[c]     = k;
        }
        auto spectra_wave = [&](auto lwc) {
            constexpr int LW = decltype(lwc)::value;
            struct SSet {
                double v[DS];
                unsigned zero;   // bit i: element i is outside [m, nmax] and is staged as 0
                bool edge;       // uniform: some row of the stage is not inside [m, nmax]
            };
            SSet set0, set1, set2;
            // lower wavenumber of row pair rp (rows 2 rp, 2 rp + 1 of the stage tile) in stage s
            auto pair_nlo = [&](int s, int rp) {
                const int par = (2 * rp) / KS, kk = (2 * rp) % KS;
                return (par ? ntop1 : ntop0) - 2 * (s * KS + kk + 1);
            };
            // stages in which every row this wavefront stages lies inside [m, nmax]: s_in0 <= s <= s_in1
            int s_in0 = 0, s_in1 = m_ok ? nstage - 1 : -1;
#pragma unroll
            for (int rp = (DS * LW) / 3; rp <= (DS * LW + DS - 1) / 3; ++rp) {
                const int nlo0 = pair_nlo(0, rp);               // nlo(s) = nlo0 - 2 KS s
                // nlo + 2 <= nmax  <=>  s >= (nlo0 + 2 - nmax) / (2 KS) rounded up;   nlo >= m  <=>  s <= (nlo0 - m) / (2 KS)
                const int lo = nlo0 + 2 > nmax ? (nlo0 + 2 - nmax + 2 * KS - 1) / (2 * KS) : 0;
                const int hi = nlo0 >= m ? (nlo0 - m) / (2 * KS) : -1;
                s_in0        = lo > s_in0 ? lo : s_in0;
                s_in1        = hi < s_in1 ? hi : s_in1;
            }
            // row bases of the next stage to be requested (stages are requested in order 0, 1, 2, ...): scalar running
            // pointers, one per row pair this wavefront touches; a stage lower is 2 KS wavenumbers = 4 KS nf doubles down
            constexpr int RP0 = (DS * LW) / 3, NRP = (DS * LW + DS - 1) / 3 - RP0 + 1;
            const double* rb[NRP];
#pragma unroll
            for (int j = 0; j < NRP; ++j) {
                rb[j] = sp + (long long)(pair_nlo(0, RP0 + j) - m) * 2 * nf;
            }
            const long long rstride = (long long)4 * KS * nf;
            auto load_s = [&](int s, SSet& st) {
                if (p.abl & 4) {
                    st.edge = false;
                    return;
                }
                if (s >= s_in0 && s <= s_in1) {
#pragma unroll
                    for (int i = 0; i < DS; ++i) {
                        const int g      = DS * LW + i;
                        const double* rp = rb[g / 3 - RP0];
                        const unsigned bo = sbo[g % 3];
                        // untracked by the compiler's s_waitcnt placement, which would wait for the younger sets too
                        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(st.v[i]) : "v"(bo), "s"(rp) : "memory");
                    }
#pragma unroll
                    for (int j = 0; j < NRP; ++j) {
                        rb[j] -= rstride;
                    }
                    st.edge = false;
                    return;
                }
#pragma unroll
                for (int j = 0; j < NRP; ++j) {
                    rb[j] -= rstride;
                }
                unsigned z = 0;
#pragma unroll
                for (int i = 0; i < DS; ++i) {
                    const int g  = DS * LW + i;
                    const int kr = krow[g % 3];
                    const int n  = pair_nlo(s, g / 3) + 2 * (1 - kr);
                    const int nc = n < m ? m : (n > nmax ? nmax : n);
                    const int colo = (int)(sbo[g % 3] >> 3) - (1 - kr) * 4 * nf;   // im * nf + f
                    // always one load (the waits count instructions); outside the spectra of this m read the table
                    const double* src = m_ok ? sp + (long long)(nc - m) * 2 * nf + colo : p.P;
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(st.v[i]) : "v"(src) : "memory");
                    z |= (unsigned)(!m_ok || nc != n) << i;
                }
                st.zero = z;
                st.edge = true;
            };
            auto store_s = [&](const SSet& st, auto slotc) {   // stage t lives in register set and LDS slot t % 3
                constexpr int SLOT = decltype(slotc)::value;
                if (p.abl & 8) {
                    return;
                }
#pragma unroll
                for (int i = 0; i < DS; ++i) {
                    const int g = DS * LW + i;
                    double v    = st.v[i];
                    if (st.edge) {   // uniform
                        v = ((st.zero >> i) & 1) ? 0. : v;
                    }
                    const unsigned la = slb[0][g % 3];
                    if ((p.abl & 16) && (i & 1)) {
                        continue;
                    }
                    asm volatile("ds_write_b64 %0, %1 offset:%2"
                                 :
                                 : "v"(la), "v"(v), "n"((SLOT * L::S_ELEM + (g / 3) * 2 * L::SCOLS) * 8)
                                 : "memory");
                }
            };
            // every stage is exactly DS loads per wavefront, issued in stage order; the waits below count them
            static_assert(2 * DS <= 63, "vmcnt is a 6-bit counter");
            auto wait_landed = [&](int t, int issued_last) {   // stage t has arrived; later stages may be in flight
                const int later = issued_last - t;
                if (later >= 2) {
                    AA_WAIT_VMCNT(2 * DS);
                }
                else if (later == 1) {
                    AA_WAIT_VMCNT(DS);
                }
                else {
                    AA_WAIT_VMCNT(0);
                }
            };
            // Pipeline: the loads of stage t are issued 5 steps before the MFMAs of stage t, its LDS stores 2 steps before
            // (slot t % 3 was last read during stage t - 3).  The stores of a step are therefore one barrier ahead of the
            // barrier that releases their readers: a late step of this wavefront does not hold up the MFMA wavefronts.
            auto prologue_stage = [&](int t, SSet& st, auto slotc) {   // stages 0, 1: load, wait, store, request stage t + 3
                if (t < nstage) {
                    AA_WAIT_VMCNT(0);
                    store_s(st, slotc);
                    AA_WAIT_LGKMCNT0();
                }
                if (t + 3 < nstage) {
                    load_s(t + 3, st);
                }
            };
            if (0 < nstage) {
                load_s(0, set0);
            }
            if (1 < nstage) {
                load_s(1, set1);
            }
            if (2 < nstage) {
                load_s(2, set2);
            }
            prologue_stage(0, set0, std::integral_constant<int, 0>{});
            prologue_stage(1, set1, std::integral_constant<int, 1>{});
            __builtin_amdgcn_s_barrier();   // stages 0, 1 are in LDS
            // step s (beside the MFMAs of stage s): store stage s + 2, request stage s + 5
            auto step = [&](int s, SSet& st, auto slotc) {
                if (s + 2 < nstage) {
                    wait_landed(s + 2, s + 4 < nstage ? s + 4 : nstage - 1);
                    store_s(st, slotc);
                    AA_WAIT_LGKMCNT0();          // written, and the set is free
                }
                if (s + 5 < nstage) {
                    load_s(s + 5, st);
                }
                __builtin_amdgcn_s_barrier();
            };
            for (int s = 0; s < nstage; s += 3) {
                step(s, set2, std::integral_constant<int, 2>{});
                if (s + 1 < nstage) {
                    step(s + 1, set0, std::integral_constant<int, 0>{});
                }
                if (s + 2 < nstage) {
                    step(s + 2, set1, std::integral_constant<int, 1>{});
                }
            }
        };
        if (wave == L::NCOMP + 1) {
            spectra_wave(std::integral_constant<int, 0>{});
        }
        else if (wave == L::NCOMP + 2) {
            spectra_wave(std::integral_constant<int, 1>{});
        }
        else {
            spectra_wave(std::integral_constant<int, 2>{});
        }
        return;
    }

    // ================= MFMA wavefronts =================
    const int lt = wave & 3;
    const int rg = wave >> 2;
    acc_t acc[2][RTW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < RTW; ++j) acc[q][j] = acc_t{0, 0, 0, 0};
    const int x     = (lane >> 4) & 1;
    const int a_off = (lane >> 4) * BN + ((lt ^ x) << 4) + (lane & 15);
    int b_off[RTW];
#pragma unroll
    for (int j = 0; j < RTW; ++j) {
        b_off[j] = (lane >> 4) * L::SCOLS + (((rg * RTW + j) ^ x) << 4) + (lane & 15);
    }
    const bool lat_active = lt * 16 < it.nrows;

    // 2 NKS steps (parity, 4 wavenumbers) of one A and three B fragments and three MFMAs; the fragments of step t + 1
    // are requested before the MFMAs of step t
    auto mma_steps = [&](const double* pbuf, const double* sbuf) {
        constexpr int NKS = KS / 4;
        double a[2], b[2][RTW];
        auto fetch = [&](int t, int slot) {
            const int par = t / NKS, ks = t % NKS;
            a[slot]       = pbuf[(par * KS + ks * 4) * BN + a_off];
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                b[slot][j] = sbuf[(par * KS + ks * 4) * L::SCOLS + b_off[j]];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 2 * NKS; ++t) {
            if (t + 1 < 2 * NKS) {
                fetch(t + 1, (t + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                acc[t / NKS][j] = RT::mma(a[t & 1], b[t & 1][j], acc[t / NKS][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    __builtin_amdgcn_s_barrier();   // stage 0 is in LDS
    for (int s = 0; s < nstage; ++s) {
        if (lat_active && !(p.abl & 32)) {
            mma_steps(lds + (s % PRING) * L::P_ELEM, lds + L::S_BASE + (s % L::SRING) * L::S_ELEM);
        }
        // every fragment read of this stage has been consumed by an MFMA above, i.e. has returned
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: merge hemispheres and store (as legendre_kernel) ----
    const int nlats  = p.nlats;
    const int jleg0  = p.nlat0[m] + it.tile * BN;
    const long long RP = p.RP;
    const int ml       = m / p.m_div;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = lt * 16 + RT::row_of(lane, g);
        if (c < it.nrows) {
            const int jn = jleg0 + c;
            const int js = nlats - 1 - jn;
            const bool st_n = jn != js && jn >= p.row_begin && jn < p.row_end;
            const bool st_s = js >= p.row_begin && js < p.row_end;
            double* fn     = p.F + ((long long)(jn - p.row_begin) * p.m_cnt + ml) * RP;
            double* fs     = p.F + ((long long)(js - p.row_begin) * p.m_cnt + ml) * RP;
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const int r = r0 + (rg * RTW + j) * 16 + (lane & 15);
                if (r < RP) {
                    double sy = acc[0][j][g], as = acc[1][j][g];
                    if ((m == 0 && (r & 1)) || r >= 2 * nf) {   // n_imag = 1 for m = 0; padding columns hold zeros
                        sy = 0;
                        as = 0;
                    }
                    if (st_n) {
                        fn[r] = sy + as;
                    }
                    if (st_s) {
                        fs[r] = sy - as;
                    }
                }
            }
        }
    }
}

template <int KS, int PRING>
static hipError_t launch_v2(LegendreParams p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds2<KS, PRING>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&legendre_kernel_v2<KS, PRING>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
        if (e != hipSuccess) {
            return e;
        }
        attr_set = true;
        if (std::getenv("ATLAS_AMD_LEG_DEBUG")) {
            int per_cu = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, legendre_kernel_v2<KS, PRING>, L::NTHR, L::BYTES);
            hipFuncAttributes fa{};
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&legendre_kernel_v2<KS, PRING>));
            std::fprintf(stderr, "[atlas_amd] legendre v2 KS=%d ring=%d lds=%d regs=%d scratch=%zu -> %d workgroups/CU\n", KS, PRING, L::BYTES,
                         fa.numRegs, (size_t)fa.localSizeBytes, per_cu);
        }
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = std::getenv("ATLAS_AMD_LEG_ABLATE") ? atoi(std::getenv("ATLAS_AMD_LEG_ABLATE")) : 0;
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL((legendre_kernel_v2<KS, PRING>), dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES, stream, p);
    return hipGetLastError();
}

template <int RTW, int NRG, class Real>
static hipError_t launch_cfg(LegendreParamsT<Real> p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<RTW, NRG, Real>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&legendre_kernel<RTW, NRG, Real>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
        if (e != hipSuccess) {
            return e;
        }
        attr_set = true;
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL((legendre_kernel<RTW, NRG, Real>), dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES, stream, p);
    return hipGetLastError();
}

// tiling for a given number of fields: rt = 16-column tiles; a workgroup covers NRG*RTW of them, the rest goes to
// further column chunks (each chunk its own workgroup, re-reading the item's P block through L2).
void legendre_tiling(int nf, int& rtw, int& nrg, int& nchunks) {
    const int rt = (2 * nf + 15) / 16;
    // measured at TL1279/O1280/137 levels (18 tiles): 6 tiles per 8-wave workgroup (3 per wave) is the fastest of
    // {9x2, 9x1, 6x1, 5x2, 3x3, 3x2, 2x3, 2x2}: 3-4 workgroups share a CU and cover each other's staging bubbles
    nrg          = rt >= 2 ? 2 : 1;
    nchunks      = (rt + 5) / 6;
    rtw          = ((rt + nchunks - 1) / nchunks + nrg - 1) / nrg;
    if (const char* e = std::getenv("ATLAS_AMD_LEG_CFG")) {  // A/B: "rtw,nrg"
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= 9 && b >= 1 && b <= 3 && (b < 3 || a <= 4)) {
            rtw     = a;
            nrg     = b;
            nchunks = (rt + a * b - 1) / (a * b);
        }
    }
}

// chunk0 / nrun: the column chunks [chunk0, chunk0 + nrun) of legendre_tiling(); nrun <= 0: all
template <class Real>
static hipError_t launch_legendre_t(const LegendreParamsT<Real>& p, int nitems, int chunk0, int nrun, hipStream_t stream) {
    int rtw, nrg, nchunks;
    legendre_tiling(p.nf, rtw, nrg, nchunks);
    if (nrun <= 0) {
        chunk0 = 0;
        nrun   = nchunks;
    }
#define LEG_CASE(R)                                                                                    \
    case R:                                                                                            \
        if (nrg == 1) return launch_cfg<R, 1, Real>(p, nitems, nchunks, chunk0, nrun, stream);         \
        if (nrg == 2) return launch_cfg<R, 2, Real>(p, nitems, nchunks, chunk0, nrun, stream);         \
        return launch_cfg<(R <= 4 ? R : 4), 3, Real>(p, nitems, nchunks, chunk0, nrun, stream);
    switch (rtw) {
        LEG_CASE(1) LEG_CASE(2) LEG_CASE(3) LEG_CASE(4) LEG_CASE(5) LEG_CASE(6) LEG_CASE(7) LEG_CASE(8) LEG_CASE(9)
    }
#undef LEG_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_legendre(const LegendreParams& p, int nitems, int chunk0, int nrun, hipStream_t stream) {
    int rtw, nrg, nchunks;
    legendre_tiling(p.nf, rtw, nrg, nchunks);
    if (rtw == 3 && nrg == 2) {
        // the 96-column workgroup (every field count whose 16-column tiles come in sixes, e.g. 137 levels) has two more
        // implementations of the same arithmetic: "lean" (default) and the role-split experiment "split";
        // ATLAS_AMD_LEG_KERNEL=classic selects the generic template
        const char* e       = std::getenv("ATLAS_AMD_LEG_KERNEL");
        const std::string k = e ? e : "lean";
        if (nrun <= 0) {
            chunk0 = 0;
            nrun   = nchunks;
        }
        if (k == "lean") {
            return launch_lean(p, nitems, nchunks, chunk0, nrun, stream);
        }
        if (k == "split") {
            return launch_v2<8, 4>(p, nitems, nchunks, chunk0, nrun, stream);
        }
        if (k == "dma") {
            return launch_dma(p, nitems, nchunks, chunk0, nrun, stream);
        }
    }
    return launch_legendre_t<double>(p, nitems, chunk0, nrun, stream);
}
// fp32 variant: same work list, tiling and table layout (the table converted to float)
hipError_t launch_legendre_f32(const LegendreParamsF32& p, int nitems, int chunk0, int nrun, hipStream_t stream) {
    return launch_legendre_t<float>(p, nitems, chunk0, nrun, stream);
}

}  // namespace trans
}  // namespace atlas_amd
