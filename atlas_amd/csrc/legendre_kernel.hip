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

#include "env.h"
#include "device_structs.h"
#include "dyn_lds.h"

// the Fourier intermediate is written once and read once, by another kernel, after everything else of this launch: streaming stores in
// fp64, where a store instruction's 16 lanes x 8 bytes are a whole 128-byte line.  In the fp32 variant they are HALF a line (16 x 4 bytes;
// the other half comes from the next tile's instruction), and streaming half lines measured slower than letting L2 combine them [r4]:
// TL1279 -> F1280 / O1280 in fp32, Legendre stage 5.88 -> 5.68 / 4.79 -> 4.65 ms; written bytes by the counters 4.88 GB for 3.77 GB
// of intermediate with the streaming form (profiles/r04_legendre_f32_probes.txt).  -DAA_LEG_PLAIN_STORE: plain stores in fp64 as well.
template <class T>
__device__ __forceinline__ void leg_store(T* ptr, T v) {
#if !defined(AA_LEG_PLAIN_STORE)
    if constexpr (sizeof(T) == 8) {
        __builtin_nontemporal_store(v, ptr);
        return;
    }
#endif
    *ptr = v;
}
#define AA_LEG_STORE(ptr, v) leg_store((ptr), (v))

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

#ifndef AA_LEG_F32_TILES_DEFAULT
#define AA_LEG_F32_TILES_DEFAULT 2   // latitude tiles per workgroup of the fp32 lean kernel (2: pairs, legendre_kernel_lean_f32_w2)
#endif
constexpr int KB   = LEG_KB_DEV;  // 8 total wavenumbers per stage
constexpr int BN   = LEG_BN_DEV;  // 64 latitudes per item
constexpr int PSTR = BN + 16;     // LDS row stride of the P stage (== 16 mod 32 doubles: conflict-free ds_read_b64)

// RTW = 16-column tiles per wave, NRG = column groups (of RTW tiles) per workgroup; a workgroup has 4*NRG waves:
// wave w handles latitude tile (w & 3) and column group (w >> 2).
template <int RTW, int NRG, class Real = double, int LTW = 1>
struct LegLds {
    static constexpr int NTHR   = 256 * NRG;
    static constexpr int SCOLS  = 16 * RTW * NRG;
    static constexpr int BM     = RealTraits<Real>::BANK_MOD;
    static constexpr int SSTR   = SCOLS + ((16 - SCOLS % BM) + BM) % BM;  // smallest stride >= SCOLS that is == 16 mod BM
    static constexpr int PSTRL  = BN * LTW + 16;                          // P rows of 64 LTW latitudes (== 16 mod 32 / 64 as well)
    static constexpr int P_ELEM = 2 * KB * PSTRL;
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
    const int r0   = p.col0 + chunk * L::SCOLS;         // first interleaved column of this chunk
    const int TL   = T + 1;                    // truncation of the table
    // largest n <= T+1 of each parity (n-m even: sym)
    const int ntop0 = TL - ((TL - m) & 1);
    const int ntop1 = TL - 1 + ((TL - m) & 1);
    const int nmax  = trc < TL ? trc : TL;  // highest n present in the input spectra
    const bool m_ok = m < trc;              // TransLocal.cc:982  (jm < truncation)
    const long long ioff = p.sp_moff ? p.sp_moff[m] * nf : (long long)(2 * trc + 3 - m) * m / 2 * nf * 2;
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
                        AA_LEG_STORE(fn + r, (Real)(sy + as));
                    }
                    if (st_s) {
                        AA_LEG_STORE(fs + r, (Real)(sy - as));  // for an equator row the southern value wins (TransLocal.cc:1056-1068 runs last)
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
// -DAA_COEX (dev build, tools/coex_probe.py): 136 registers per wavefront, so that a CU takes ONE workgroup of this kernel (two need
// 4 x 136 registers per SIMD) and keeps 240 registers per SIMD and 112 KiB of LDS for a workgroup of the Fourier stage of another
// transform -- the co-residency experiment of DESIGN 3.4
#if defined(AA_COEX)
#define AA_LEAN_WPS 3
#else
#define AA_LEAN_WPS 4
#endif
// The body is written once for both scalar types [r3]: Real = double is the kernel described above; Real = float (BASELINE config C5 and the
// other fp32 calls) is the same kernel with 4-byte elements -- 8-byte table loads, 4-byte spectra loads and fragment reads,
// v_mfma_f32_16x16x4_f32 -- and the summation order of legendre_kernel<3, 2, float> (bit-identical results).
// RTW = 16-column tiles per wavefront (1, 2 or 3; two column groups: 32 / 64 / 96 columns per workgroup) -- whatever
// legendre_tiling() chooses for the field count, as long as the workgroup has its two column groups.
// compile-time loop i = I .. N-1 (the hand-counted waits below are immediates that depend on the step)
template <int I, int N, class Fn>
__device__ __forceinline__ void lean_static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        lean_static_for<I + 1, N>(fn);
    }
}
// LTW = latitude tiles of 16 per wavefront [r6]: 1 = the 64-latitude workgroup tile described above; 2 (fp32 only: the accumulators of a
// 32 x 48 register tile are 48 registers in fp32, 96 in fp64) = a PAIR of consecutive 64-latitude tiles of one wavenumber per
// workgroup (LegendreParams::items2): every spectra fragment read from LDS feeds two MFMAs instead of one and a stage has 24 MFMAs per
// wavefront between its barriers instead of 12 -- the fp32 MFMA takes half the time of the fp64 one, so the per-stage costs around it
// (operand wait, LDS writes, barrier) weigh twice as much in the 64-latitude form (profiles/r04_legendre_f32_probes.txt).
template <int RTW, class Real, int LTW = 1>
__device__ __forceinline__ void legendre_lean_body(const LegendreParamsT<Real> p) {
    using L  = LegLds<RTW, 2, Real, LTW>;
    constexpr int PSTRL = L::PSTRL;   // LDS row stride of the P stage: 64 LTW latitudes + 16
    using RT = RealTraits<Real>;
    using acc_t = typename RT::acc_t;
    constexpr int NTHR = 512;
    constexpr int EB   = (int)sizeof(Real);   // element bytes
    constexpr bool F64 = EB == 8;
    // dynamic LDS: L::BYTES, addressed from absolute offset 0 by the inline asm below.  The kernel must not have static LDS
    // (it would be placed at offset 0 and alias the staging buffers): the extern array is the only declaration.
    extern __shared__ double lean_lds[];
    (void)lean_lds;

    const int nchunks   = p.nchunks_run;
    const int bx        = blockIdx.x & 7;
    const int bq        = blockIdx.x >> 3;
    const int chunk     = p.chunk0 + bq % nchunks;
    const int item_slot = (bq / nchunks) * 8 + bx;
    if (item_slot >= (LTW == 2 ? p.nitems2 : p.nitems)) {
        return;
    }
    const LegendreItemDev it = (LTW == 2 ? p.items2 : p.items)[item_slot];
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
    const int r0   = p.col0 + chunk * L::SCOLS;
    const int TL   = p.T + 1;
    const int ntop0 = TL - ((TL - m) & 1);
    const int ntop1 = TL - 1 + ((TL - m) & 1);
    const int nmax  = trc < TL ? trc : TL;
    const bool m_ok = m < trc;
    const long long ioff = p.sp_moff ? p.sp_moff[m] * nf : (long long)(2 * trc + 3 - m) * m / 2 * nf * 2;
    const Real* __restrict__ sp = p.sp + ioff;
    const Real* __restrict__ Pb = p.P + it.p_off;
    const int nstage = it.kpad / KB;
    const int ntop_hi = ntop0 > ntop1 ? ntop0 : ntop1, ntop_lo = ntop0 < ntop1 ? ntop0 : ntop1;

    acc_t acc[2][LTW][RTW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < LTW; ++i)
#pragma unroll
            for (int j = 0; j < RTW; ++j) acc[q][i][j] = acc_t{0, 0, 0, 0};

    // ---- staging registers of one stage: one 16-byte table load and three spectra elements per thread ----
    typedef int i4_t __attribute__((ext_vector_type(4)));
    typedef int i2_t __attribute__((ext_vector_type(2)));
    using preg_t = std::conditional_t<F64 || LTW == 2, i4_t, i2_t>;   // two table elements (four of a tile pair in fp32)
    static_assert(LTW == 1 || !F64, "the tile pair is the fp32 variant's");
    // register sets: the stage being written to LDS and the ones still in flight = stages of look-ahead of the operand streams.
    // 3 (fp64, 122 registers: a fourth set would cost the fourth wavefront per SIMD).  fp32 [r4]: dev builds -DAA_LEG_F32_NSET=4 / 5
    // measured 5.89 / 6.41 against 5.90 ms on TL1279 -> F1280: the operand streams are not a latency the look-ahead could cover
    // (profiles/r04_legendre_f32_probes.txt)
#ifndef AA_LEG_F32_NSET
#define AA_LEG_F32_NSET 3
#endif
    constexpr int NSET = F64 ? 3 : AA_LEG_F32_NSET;
    static_assert(NSET >= 3 && NSET <= 6 && (NSET - 1) * (1 + RTW) <= 63, "vmcnt is a 6-bit counter");
    preg_t preg[NSET];
    Real sreg[NSET][RTW];
    // table element pair 2 tid of the stage tile [2 parities][8 k][64 latitudes]
    // (LTW = 2: element quadruple 4 tid of [2 parities][8 k][128 latitudes]; latitudes 64 .. 127 come from the second tile's block, which
    // follows the first one's in the table; a workgroup whose item is a single tile reads the first tile twice -- never beyond the table)
    const int pe = 2 * LTW * tid, ppar = pe / (KB * BN * LTW), pk = (pe % (KB * BN * LTW)) / (BN * LTW), pc = pe % (BN * LTW);
    const int ptile = (LTW == 2 && it.nrows > BN) ? pc / BN : 0;
    const unsigned pbo = (unsigned)EB * (unsigned)(((ptile * 2 + ppar) * it.kpad + pk) * BN + (pc % BN));   // bytes from the stage's first table row
    const int plds     = ppar * (KB * PSTRL) + pk * PSTRL + pc;
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
        sbo[i]   = (unsigned)EB * (unsigned)((ntop - ntop_lo + 2 * (KB - 1 - k)) * 2 * nf + scolo[i]);
        slds[i]  = L::P_ELEM + row * L::SSTR + col;
    }
    // stages whose 16 rows are all inside [m, nmax]
    const int s_ff = ntop_hi > nmax ? (ntop_hi - nmax + 2 * KB - 1) / (2 * KB) : 0;
    const int s_lf = m_ok ? (ntop_lo - 2 * (KB - 1) - m >= 0 ? (ntop_lo - 2 * (KB - 1) - m) / (2 * KB) : -1) : -1;
    // scalar bases of the next stage to be requested (stages are requested in order)
    const Real* pbase = Pb;
    const Real* sbase = sp + (long long)(ntop_lo - 2 * (KB - 1) - m) * 2 * nf;
    const long long sstride = (long long)4 * KB * nf;   // 2 KB wavenumbers down

    unsigned szero[NSET] = {};   // bit i: element i of the staged stage is outside [m, nmax]
    bool sedge[NSET]     = {};
    // stage t travels through register set t % NSET into LDS buffer t & 1; it is requested NSET stages before it is used
    auto load_stage = [&](int s, auto setc) {   // must be called for s = 0, 1, 2, ... in order
        constexpr int SET = decltype(setc)::value;
        if constexpr (F64 || LTW == 2) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(preg[SET]) : "v"(pbo), "s"(pbase) : "memory");
        }
        else {
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(preg[SET]) : "v"(pbo), "s"(pbase) : "memory");
        }
        if (s >= s_ff && s <= s_lf) {
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                if constexpr (F64) {
                    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(sreg[SET][i]) : "v"(sbo[i]), "s"(sbase) : "memory");
                }
                else {
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(sreg[SET][i]) : "v"(sbo[i]), "s"(sbase) : "memory");
                }
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
                const Real* src = m_ok ? sp + (long long)(nc - m) * 2 * nf + scolo[i] : p.P;
                if constexpr (F64) {
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sreg[SET][i]) : "v"(src) : "memory");
                }
                else {
                    asm volatile("global_load_dword %0, %1, off" : "=v"(sreg[SET][i]) : "v"(src) : "memory");
                }
                z |= (unsigned)(!m_ok || nc != n) << i;
            }
            szero[SET] = z;
            sedge[SET] = true;
        }
#if defined(AA_LEG_PROBE_FIXED_OPERANDS)
        // dev probe (results wrong): bit 0 -- every stage re-reads the FIRST table tile of the item (cache hits instead of the HBM / L2
        // stream), bit 1 -- likewise the spectra: what the latency of the operand streams costs behind a three-stage look-ahead
        if (!(AA_LEG_PROBE_FIXED_OPERANDS & 1)) pbase += KB * BN;
        if (!(AA_LEG_PROBE_FIXED_OPERANDS & 2)) sbase -= sstride;
#else
        pbase += KB * BN;
        sbase -= sstride;
#endif
    };
    const unsigned plds_b = (unsigned)EB * (unsigned)plds;
    unsigned slds_b[RTW];
#pragma unroll
    for (int i = 0; i < RTW; ++i) {
        slds_b[i] = (unsigned)EB * (unsigned)slds[i];
    }
    constexpr int LPS = 1 + RTW;   // loads per stage and thread
    auto store_stage = [&](auto bufc, auto setc, int younger_in_flight) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int SET = decltype(setc)::value;
        // the loads above are invisible to the compiler's own s_waitcnt placement: count them
        if (NSET > 3 && younger_in_flight >= 3) {
            AA_WAIT_VMCNT((NSET > 3 ? 3 : 2) * LPS);
        }
        else if (younger_in_flight >= 2) {
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
            const preg_t pv   = preg[SET];
            const unsigned la = plds_b;
            if constexpr (F64 || LTW == 2) {
                asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(la), "v"(pv), "n"(buf * L::STAGE * EB) : "memory");
            }
            else {
                asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(la), "v"(pv), "n"(buf * L::STAGE * EB) : "memory");
            }
        }
        if (sedge[SET]) {   // uniform; both arms are asm volatile so that this stays a branch (as selects it is 15 VALU per stage)
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                const Real v      = ((szero[SET] >> i) & 1) ? (Real)0 : sreg[SET][i];
                const unsigned la = slds_b[i];
                if constexpr (F64) {
                    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(la), "v"(v), "n"(buf * L::STAGE * EB) : "memory");
                }
                else {
                    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(la), "v"(v), "n"(buf * L::STAGE * EB) : "memory");
                }
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                const Real v      = sreg[SET][i];
                const unsigned la = slds_b[i];
                if constexpr (F64) {
                    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(la), "v"(v), "n"(buf * L::STAGE * EB) : "memory");
                }
                else {
                    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(la), "v"(v), "n"(buf * L::STAGE * EB) : "memory");
                }
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
    if constexpr (NSET > 3) {
        lean_static_for<3, NSET>([&](auto ic) {
            if (nstage > decltype(ic)::value) {
                load_stage(decltype(ic)::value, ic);
            }
        });
    }
    store_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, (nstage < NSET ? nstage : NSET) - 1);
    __syncthreads();

    const int a_off = (lane >> 4) * PSTRL + lt * 16 + (lane & 15);
    const int b_off = L::P_ELEM + (lane >> 4) * L::SSTR + rg * RTW * 16 + (lane & 15);
    const bool lat_active = lt * 16 < it.nrows;   // (LTW = 2: the first tile of the pair; its second tile is multiplied along -- its rows beyond nrows are not stored)

    // four steps (parity, 4 wavenumbers) of one A and three B fragments and three MFMAs; the fragments of step t + D are requested
    // before the MFMAs of step t, D = 1.  [r4] In fp32 a step's three MFMAs are 96 cycles of matrix pipe (192 in fp64), shorter than an
    // LDS round trip under load -- but two or three steps in flight (-DAA_LEG_FRAG_DEPTH=2 / 3) measured 5.87 -> 5.91 / 6.38 ms on
    // TL1279 -> F1280 / 137 levels: the other five wavefronts of the SIMD already cover the round trip (profiles/r04_fft_native.txt)
    const unsigned a_b = (unsigned)EB * (unsigned)a_off, b_b = (unsigned)EB * (unsigned)b_off;
    auto mma_steps = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int NKS = KB / 4;
        constexpr int NST = 2 * NKS;                 // steps of a stage
#if defined(AA_LEG_FRAG_DEPTH)
        constexpr int D   = AA_LEG_FRAG_DEPTH;       // dev builds: A/B
#else
        constexpr int D   = 1;                       // steps of fragment reads in flight
#endif
        static_assert(D >= 1 && D < NST && D * (LTW + RTW) <= 15, "lgkmcnt is a 4-bit counter");
        Real a_[D + 1][LTW], b_[D + 1][RTW];
        // fragment reads as asm with immediate offsets (the compiler pairs them into ds_read2 and pays a VALU add per pair
        // for the base); their completion is counted by hand: 1 + RTW reads per step, D steps in flight
        auto fetch = [&](auto tc) {
            constexpr int t    = decltype(tc)::value;
            constexpr int slot = t % (D + 1);
            constexpr int par = t / NKS, ks = t % NKS;
            const unsigned ab = a_b, bb = b_b;
            Real(&a)[D + 1][LTW] = a_;   // (named references: clang does not capture an array that a generic lambda only uses as an
            Real(&b)[D + 1][RTW] = b_;   //  asm output operand)
#pragma unroll
            for (int i = 0; i < LTW; ++i) {
                if constexpr (F64) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[slot][i]) : "v"(ab), "n"((buf * L::STAGE + (par * KB + ks * 4) * PSTRL + i * BN) * EB) : "memory");
                }
                else {
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[slot][i]) : "v"(ab), "n"((buf * L::STAGE + (par * KB + ks * 4) * PSTRL + i * BN) * EB) : "memory");
                }
            }
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                if constexpr (F64) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2"
                                 : "=v"(b[slot][j])
                                 : "v"(bb), "n"((buf * L::STAGE + (par * KB + ks * 4) * L::SSTR + j * 16) * EB)
                                 : "memory");
                }
                else {
                    asm volatile("ds_read_b32 %0, %1 offset:%2"
                                 : "=v"(b[slot][j])
                                 : "v"(bb), "n"((buf * L::STAGE + (par * KB + ks * 4) * L::SSTR + j * 16) * EB)
                                 : "memory");
                }
            }
        };
        lean_static_for<0, D>([&](auto tc) { fetch(tc); });
        lean_static_for<0, NST>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t + D < NST) {
                fetch(std::integral_constant<int, t + D>{});
            }
            // younger steps still in flight behind step t: D, fewer at the end of the stage
            constexpr int younger = (NST - 1 - t) < D ? (NST - 1 - t) : D;
            __builtin_amdgcn_s_waitcnt((15) | (3 << 14) | (7 << 4) | ((younger * (LTW + RTW)) << 8));   // lgkmcnt(younger * (LTW + RTW))
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < LTW; ++i) {
#pragma unroll
                for (int j = 0; j < RTW; ++j) {
                    acc[t / NKS][i][j] = RT::mma(a_[t % (D + 1)][i], b_[t % (D + 1)][j], acc[t / NKS][i][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // stage s is multiplied from LDS buffer s & 1; beside it stage s + NSET is requested into the register set that stage s
    // left (s % NSET), and afterwards stage s + 1 goes from its set to the other buffer
    auto run_stage = [&](int s, auto bufc, auto setc) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int SET = decltype(setc)::value;
        if (s + NSET < nstage) {
            load_stage(s + NSET, setc);
        }
        if (lat_active) {
            mma_steps(bufc);
        }
        if (s + 1 < nstage) {
            const int last = s + NSET < nstage ? s + NSET : nstage - 1;   // last stage requested so far
            store_stage(std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, (SET + 1) % NSET>{}, last - (s + 1));
        }
#if defined(AA_LEG_PROBE_NOBARRIER)
        // dev probe (results wrong): the stage loop without its workgroup barrier -- what the synchronisation of the eight wavefronts
        // after every stage costs (the fp32 variant has half the matrix time per stage to hide it behind)
        asm volatile("" ::: "memory");
#else
        __syncthreads();
#endif
    };
    if constexpr (NSET == 3) {   // (written out: the fp64 kernel stays instruction for instruction the one of round 3)
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
    }
    else {
        constexpr int PERIOD = (NSET % 2 == 0) ? NSET : 2 * NSET;   // stages after which (LDS buffer, register set) repeat
        for (int s = 0; s < nstage; s += PERIOD) {
            lean_static_for<0, PERIOD>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (i == 0 || s + i < nstage) {
                    run_stage(s + i, std::integral_constant<int, (i & 1)>{}, std::integral_constant<int, i % NSET>{});
                }
            });
        }
    }

#if defined(AA_LEG_PROBE_NO_EPILOGUE)
    // dev probe (results wrong): no merge, no stores -- what the epilogue of a work item costs
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < LTW; ++i)
#pragma unroll
            for (int j = 0; j < RTW; ++j) asm volatile("" ::"v"(acc[q][i][j]));
    return;
#endif
    // ---- epilogue: merge hemispheres and store (as legendre_kernel) ----
    const int nlats  = p.nlats;
    const int jleg0  = p.nlat0[m] + it.tile * BN;
    const long long RP = p.RP;
    const int ml       = m / p.m_div;
    // (fp32 with the MFMA operands exchanged -- C^T in the accumulators: a lane holds four consecutive columns of one latitude, one
    // 16-byte store per tile and hemisphere instead of four 4-byte stores -- measured 5.98 -> 6.08 ms on TL1279 -> F1280: not kept [r4])
#pragma unroll
    for (int ti = 0; ti < LTW; ++ti)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = ti * BN + lt * 16 + RT::row_of(lane, g);
        if (c < it.nrows) {
            const int jn = jleg0 + c;
            const int js = nlats - 1 - jn;
            const bool st_n = jn != js && jn >= p.row_begin && jn < p.row_end;
            const bool st_s = js >= p.row_begin && js < p.row_end;
            Real* fn       = p.F + ((long long)(jn - p.row_begin) * p.m_cnt + ml) * RP;
            Real* fs       = p.F + ((long long)(js - p.row_begin) * p.m_cnt + ml) * RP;
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const int r = r0 + (rg * RTW + j) * 16 + (lane & 15);
                if (r < RP) {
                    Real sy = acc[0][ti][j][g], as = acc[1][ti][j][g];
                    if ((m == 0 && (r & 1)) || r >= 2 * nf) {   // n_imag = 1 for m = 0; padding columns hold zeros
                        sy = 0;
                        as = 0;
                    }
#if defined(AA_LEG_LAYOUT_PROBE)
                    // dev probe (results unusable by the Fourier stage): what the store side of a layout with several
                    // wavenumbers of one field per 128-byte line costs: p.abl = 1: [2 m][4 fields], 2: [4 m][2 fields], 3: [8 m][1 field]
                    if (p.abl >= 1 && p.abl <= 3) {
                        const int mb   = 1 << p.abl;            // wavenumbers per line
                        const int cw   = 16 >> p.abl;           // doubles of one wavenumber per line
                        const int cb   = r / cw, cr = r - cb * cw;
                        const long long mrow = (p.m_cnt + mb - 1) / mb;
                        const long long on = (((long long)(jn - p.row_begin) * mrow + ml / mb) * (RP / cw) + cb) * 16 + (ml % mb) * cw + cr;
                        const long long os = (((long long)(js - p.row_begin) * mrow + ml / mb) * (RP / cw) + cb) * 16 + (ml % mb) * cw + cr;
                        if (st_n) {
                            AA_LEG_STORE(p.F + on, sy + as);
                        }
                        if (st_s) {
                            AA_LEG_STORE(p.F + os, sy - as);
                        }
                        continue;
                    }
#endif
                    if (st_n) {
                        AA_LEG_STORE(fn + r, sy + as);
                    }
                    if (st_s) {
                        AA_LEG_STORE(fs + r, sy - as);
                    }
                }
            }
        }
    }
#if defined(AA_COEX)
    asm volatile("" ::: "v135");   // register count 136 (see above)
#endif
}

__global__ void __launch_bounds__(512, AA_LEAN_WPS) legendre_kernel_lean(LegendreParams p) {
    legendre_lean_body<3, double>(p);
}
__global__ void __launch_bounds__(512, 4) legendre_kernel_lean_f32(LegendreParamsF32 p) {
    legendre_lean_body<3, float>(p);
}
// [r6] the same on pairs of latitude tiles (LTW = 2): 128 latitudes x 96 columns per workgroup
__global__ void __launch_bounds__(512, 4) legendre_kernel_lean_f32_w2(LegendreParamsF32 p) {
    legendre_lean_body<3, float, 2>(p);
}
// the narrower workgroups (field counts whose tiles come in fours or twos per column chunk)
template <int RTW, class Real>
__global__ void __launch_bounds__(512, 4) legendre_kernel_lean_n(LegendreParamsT<Real> p) {
    legendre_lean_body<RTW, Real>(p);
}
// dev tool (occupancy sensitivity of the lean kernels): ATLAS_AMD_LEG_LDS_PAD=<bytes> more dynamic LDS per workgroup, i.e. fewer
// workgroups per CU (profiles/r04_legendre_f32_probes.txt)
template <auto Kernel>
static int lean_lds_pad(int bytes) {
    static const int pad = atlas_amd::env_get("ATLAS_AMD_LEG_LDS_PAD") ? atoi(atlas_amd::env_get("ATLAS_AMD_LEG_LDS_PAD")) : 0;
    if (pad > 0) {
        (void)ensure_dynamic_lds<Kernel>(bytes + pad);
    }
    return pad > 0 ? pad : 0;
}
template <int RTW, class Real>
static hipError_t launch_lean_n(LegendreParamsT<Real> p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<RTW, 2, Real>;
    if (hipError_t e = ensure_dynamic_lds<&legendre_kernel_lean_n<RTW, Real>>(L::BYTES); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL((legendre_kernel_lean_n<RTW, Real>), dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES, stream, p);
    return hipGetLastError();
}

static hipError_t launch_lean_f32(LegendreParamsF32 p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<3, 2, float>;
    if (hipError_t e = ensure_dynamic_lds<&legendre_kernel_lean_f32>(L::BYTES); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL(legendre_kernel_lean_f32, dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES + lean_lds_pad<&legendre_kernel_lean_f32>(L::BYTES), stream, p);
    return hipGetLastError();
}
// [r6] ... and the narrower workgroups (one or two 16-column tiles per wavefront) of the fp32 variant on tile pairs
template <int RTW>
__global__ void __launch_bounds__(512, 4) legendre_kernel_lean_n_f32_w2(LegendreParamsF32 p) {
    legendre_lean_body<RTW, float, 2>(p);
}
template <int RTW>
static hipError_t launch_lean_n_f32_w2(LegendreParamsF32 p, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<RTW, 2, float, 2>;
    if (hipError_t e = ensure_dynamic_lds<&legendre_kernel_lean_n_f32_w2<RTW>>(L::BYTES); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    p.nitems        = p.nitems2;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
    const int slots = (p.nitems2 + 7) / 8;
    hipLaunchKernelGGL(legendre_kernel_lean_n_f32_w2<RTW>, dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES, stream, p);
    return hipGetLastError();
}
// [r6] the fp32 lean kernel on pairs of latitude tiles (items2)
static hipError_t launch_lean_f32_w2(LegendreParamsF32 p, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<3, 2, float, 2>;
    if (hipError_t e = ensure_dynamic_lds<&legendre_kernel_lean_f32_w2>(L::BYTES); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    p.nitems        = p.nitems2;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
    const int slots = (p.nitems2 + 7) / 8;
    hipLaunchKernelGGL(legendre_kernel_lean_f32_w2, dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES, stream, p);
    return hipGetLastError();
}
static hipError_t launch_lean(LegendreParams p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<3, 2, double>;
    if (hipError_t e = ensure_dynamic_lds<&legendre_kernel_lean>(L::BYTES); e != hipSuccess) {   // dyn_lds.h
        return e;
    }
    p.nitems        = nitems;
    p.nchunks       = nchunks;
    p.chunk0        = chunk0;
    p.nchunks_run   = nrun;
    p.abl           = 0;
#if defined(AA_LEG_LAYOUT_PROBE)
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_LAYOUT_PROBE")) {
        p.abl = atoi(e);
    }
#endif
    const int slots = (nitems + 7) / 8;
    hipLaunchKernelGGL(legendre_kernel_lean, dim3(slots * nrun * 8), dim3(L::NTHR), L::BYTES + lean_lds_pad<&legendre_kernel_lean>(L::BYTES), stream, p);
    return hipGetLastError();
}


#if defined(ATLAS_AMD_EXPERIMENTS)
// [r6] the lean body as one operand stream over the units of a persistent workgroup (ATLAS_AMD_LEG_KERNEL=stream): bit-identical, built
// to hide a unit's prologue and epilogue -- measured 2.5 % SLOWER than the lean kernels (profiles/r06_legendre_stream.txt): experiments build only
#include "../../tools/experiments/legendre_stream.inc"
#include "../../tools/experiments/legendre_kernel_experiments.inc"
#endif

template <int RTW, int NRG, class Real>
static hipError_t launch_cfg(LegendreParamsT<Real> p, int nitems, int nchunks, int chunk0, int nrun, hipStream_t stream) {
    using L = LegLds<RTW, NRG, Real>;
    if (hipError_t e = ensure_dynamic_lds<&legendre_kernel<RTW, NRG, Real>>(L::BYTES); e != hipSuccess) {   // dyn_lds.h
        return e;
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

// The lean kernels request their operands with inline-assembly loads that complete asynchronously and count the waits by hand
// (s_waitcnt vmcnt / lgkmcnt): correct only as long as the compiler never copies or spills one of those registers between the
// request and the wait.  A spill shows as scratch memory: if a toolchain (another hipcc, other flags) gives a lean kernel a
// private segment, it is not used -- the generic template (same arithmetic, same order, compiler-placed waits) takes over, with
// one line on stderr.  Built and tested with ROCm 7.2.0 hipcc (AMD clang 22); the bitwise lean-vs-classic GPU test
// (tests/test_gpu_trans.py::test_legendre_kernel_variants_are_bitwise_equal) runs in the default suite.  
template <auto Kernel>
static bool lean_kernel_usable(const char* name) {
    static const bool ok = [name]() {
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(Kernel)) != hipSuccess) {
            (void)hipGetLastError();
            return true;   // no information: the kernel as tested
        }
        if (fa.localSizeBytes != 0) {
            std::fprintf(stderr, "[atlas_amd] %s was compiled with %zu bytes of scratch (register spills): its hand-counted waits are "
                                 "not safe with this toolchain, using the generic Legendre kernel instead\n", name, (size_t)fa.localSizeBytes);
            return false;
        }
        return true;
    }();
    return ok;
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
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_CFG")) {  // A/B: "rtw,nrg"
        int a = 0, b = 0;
#if defined(ATLAS_AMD_EXPERIMENTS)
        const int amax = 9, bmax = 3;   // the tilings that lost (round 1 sweep, {9x2, 9x1, 6x1, 5x2, 3x3, ...}): experiments build only
#else
        const int amax = 3, bmax = 2;   // the product library instantiates what legendre_tiling() can return by itself
#endif
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && a <= amax && b >= 1 && b <= bmax && (b < 3 || a <= 4)) {
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
#if defined(ATLAS_AMD_EXPERIMENTS)
#define LEG_CASE(R)                                                                                    \
    case R:                                                                                            \
        if (nrg == 1) return launch_cfg<R, 1, Real>(p, nitems, nchunks, chunk0, nrun, stream);         \
        if (nrg == 2) return launch_cfg<R, 2, Real>(p, nitems, nchunks, chunk0, nrun, stream);         \
        return launch_cfg<(R <= 4 ? R : 4), 3, Real>(p, nitems, nchunks, chunk0, nrun, stream);
    switch (rtw) {
        LEG_CASE(1) LEG_CASE(2) LEG_CASE(3) LEG_CASE(4) LEG_CASE(5) LEG_CASE(6) LEG_CASE(7) LEG_CASE(8) LEG_CASE(9)
    }
#else
    // (round 4's library also carried the A/B tilings 4 .. 9 tiles per wavefront and three column groups: 42 kernels nothing launched,
    // among them the only Legendre instances with register spills -- profiles/r05_kernel_resources.txt)
#define LEG_CASE(R)                                                                                    \
    case R:                                                                                            \
        if (nrg == 1) return launch_cfg<R, 1, Real>(p, nitems, nchunks, chunk0, nrun, stream);         \
        return launch_cfg<R, 2, Real>(p, nitems, nchunks, chunk0, nrun, stream);
    switch (rtw) {
        LEG_CASE(1) LEG_CASE(2) LEG_CASE(3)
    }
#endif
#undef LEG_CASE
    return hipErrorInvalidValue;
}

// [r6] Field counts whose 16-column tiles do not come in sixes.  legendre_tiling() gives every column chunk of a call the same width;
// with 13 tiles (97 fields) that is three chunks of six -- the stage costs what 144 fields cost.  Where it is cheaper the call becomes
// TWO launches: the full 96-column chunks, then the remaining one to five tiles with the workgroup width that fits them (LegendreParams::
// col0 = where they start).  Cost model from the TL1279 sweep (tools/probe/nf_sweep.py; ms per chunk of 1 / 2 / 3 tiles per wavefront:
// 1.45 / 2.07 / 3.03 -- the narrow chunks are bound by streaming the table once more): e.g. 97 fields 8.08 -> 7.2 ms, 110 fields
// 8.1 -> 7.2; not for 49 - 64 fields (two chunks of four tiles beat six + two).  Per-column arithmetic does not depend on the chunking.
static bool legendre_mixed_tiling(int nf, int T, int& nfull, int& rem_rtw) {
    if (T < 256) {   // small truncations: a launch more costs what the narrower chunk saves (TL159: 0.03 - 0.04 ms either way)
        return false;
    }
    const int rt = (2 * nf + 15) / 16;
    nfull        = rt / 6;
    const int rem = rt - 6 * nfull;
    if (nfull < 1 || rem < 1 || atlas_amd::env_get("ATLAS_AMD_LEG_CFG")) {
        return false;
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_MIXED")) {   // A/B: 0 = one width per call
        if (atoi(e) == 0) {
            return false;
        }
    }
    static const double cost[4] = {0., 0.48, 0.68, 1.0};
    int rtw, nrg, nchunks;
    legendre_tiling(nf, rtw, nrg, nchunks);
    rem_rtw = (rem + 1) / 2;
    return nfull * cost[3] + cost[rem_rtw] < nchunks * cost[rtw < 1 ? 1 : (rtw > 3 ? 3 : rtw)] - 1e-9;
}

static hipError_t launch_legendre_rtw(const LegendreParams& p, int rtw, int nrg, int nchunks, int nitems, int chunk0, int nrun, hipStream_t stream);

hipError_t launch_legendre(const LegendreParams& p, int nitems, int chunk0, int nrun, hipStream_t stream) {
    int rtw, nrg, nchunks;
    legendre_tiling(p.nf, rtw, nrg, nchunks);
    {
        const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_KERNEL");
        int nfull = 0, rem_rtw = 0;
        if (chunk0 == 0 && nrun <= 0 && (!e || std::string(e) == "lean") && legendre_mixed_tiling(p.nf, p.T, nfull, rem_rtw) &&
            lean_kernel_usable<&legendre_kernel_lean>("legendre_kernel_lean") &&
            lean_kernel_usable<&legendre_kernel_lean_n<1, double>>("legendre_kernel_lean_n<1, double>") &&
            lean_kernel_usable<&legendre_kernel_lean_n<2, double>>("legendre_kernel_lean_n<2, double>")) {
            if (hipError_t err = launch_legendre_rtw(p, 3, 2, nfull, nitems, 0, nfull, stream); err != hipSuccess) {
                return err;
            }
            LegendreParams q = p;
            q.col0           = nfull * 96;
            return launch_legendre_rtw(q, rem_rtw, 2, 1, nitems, 0, 1, stream);
        }
    }
    return launch_legendre_rtw(p, rtw, nrg, nchunks, nitems, chunk0, nrun, stream);
}

static hipError_t launch_legendre_rtw(const LegendreParams& p, int rtw, int nrg, int nchunks, int nitems, int chunk0, int nrun, hipStream_t stream) {
    // [r6] one to eight fields (a single 16-column tile): the two-group lean workgroup with its second column group on padding columns
    // (never stored: r >= RP) instead of the generic one-group template -- the stage is bound by streaming the table, which the lean
    // staging does at 4.9 TB/s where the template reaches 2.9: 2.46 -> 1.43 ms at TL1279 / O1280, 0.33 -> 0.19 at TL639, 0.06 -> 0.04 at
    // TL319 (fp64; the fp32 template is as fast as the lean form there and stays).  Same arithmetic per column: same bits.
    if (rtw == 1 && nrg == 1 && !atlas_amd::env_get("ATLAS_AMD_LEG_CFG")) {
        nrg = 2;
    }
    if (rtw == 3 && nrg == 2) {
        // the 96-column workgroup (every field count whose 16-column tiles come in sixes, e.g. 137 levels) has two more
        // implementations of the same arithmetic: "lean" (default) and, in experiment builds only, "split" and "dma"
        // and "lean2" (tools/experiments/legendre_kernel_experiments.inc); ATLAS_AMD_LEG_KERNEL=classic selects the generic template
        const char* e       = atlas_amd::env_get("ATLAS_AMD_LEG_KERNEL");
        const std::string k = e ? e : "lean";
        if (nrun <= 0) {
            chunk0 = 0;
            nrun   = nchunks;
        }
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (k == "stream" && lean_kernel_usable<&legendre_kernel_stream>("legendre_kernel_stream")) {
            return launch_stream_t<&legendre_kernel_stream, 3, double>(p, nitems, nchunks, chunk0, nrun, stream);
        }
#endif
        if ((k == "lean" || k == "stream") && lean_kernel_usable<&legendre_kernel_lean>("legendre_kernel_lean")) {
            return launch_lean(p, nitems, nchunks, chunk0, nrun, stream);
        }
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (k == "lean2" && p.items2 && p.nitems2 > 0) {
            return launch_lean2(p, nchunks, chunk0, nrun, stream);
        }
        if (k == "split") {
            return launch_v2<8, 4>(p, nitems, nchunks, chunk0, nrun, stream);
        }
        if (k == "dma") {
            return launch_dma(p, nitems, nchunks, chunk0, nrun, stream);
        }
#else
        if (k == "split" || k == "dma" || k == "lean2") {
            return hipErrorNotSupported;   // tools/experiments: needs a library built with -DATLAS_AMD_EXPERIMENTS
        }
#endif
    }
    if (nrg == 2 && (rtw == 1 || rtw == 2)) {
        const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_KERNEL");
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (e && std::string(e) == "stream") {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            if (rtw == 1 && lean_kernel_usable<&legendre_kernel_stream_n<1, double>>("legendre_kernel_stream_n<1, double>")) {
                return launch_stream_t<&legendre_kernel_stream_n<1, double>, 1, double>(p, nitems, nchunks, chunk0, nrun, stream);
            }
            if (rtw == 2 && lean_kernel_usable<&legendre_kernel_stream_n<2, double>>("legendre_kernel_stream_n<2, double>")) {
                return launch_stream_t<&legendre_kernel_stream_n<2, double>, 2, double>(p, nitems, nchunks, chunk0, nrun, stream);
            }
        }
#endif
        if (!e || std::string(e) == "lean" || std::string(e) == "stream") {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            if (rtw == 1 && lean_kernel_usable<&legendre_kernel_lean_n<1, double>>("legendre_kernel_lean_n<1, double>")) {
                return launch_lean_n<1, double>(p, nitems, nchunks, chunk0, nrun, stream);
            }
            if (rtw == 2 && lean_kernel_usable<&legendre_kernel_lean_n<2, double>>("legendre_kernel_lean_n<2, double>")) {
                return launch_lean_n<2, double>(p, nitems, nchunks, chunk0, nrun, stream);
            }
        }
    }
    return launch_legendre_t<double>(p, nitems, chunk0, nrun, stream);
}
// fp32 variant: same work list, tiling and table layout (the table converted to float)
static hipError_t launch_legendre_f32_rtw(const LegendreParamsF32& p, int rtw, int nrg, int nchunks, int nitems, int chunk0, int nrun, hipStream_t stream);

hipError_t launch_legendre_f32(const LegendreParamsF32& p, int nitems, int chunk0, int nrun, hipStream_t stream) {
    int rtw, nrg, nchunks;
    legendre_tiling(p.nf, rtw, nrg, nchunks);
    {
        const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_KERNEL");
        int nfull = 0, rem_rtw = 0;
        if (chunk0 == 0 && nrun <= 0 && (!e || std::string(e) == "lean") && legendre_mixed_tiling(p.nf, p.T, nfull, rem_rtw) &&
            lean_kernel_usable<&legendre_kernel_lean_f32>("legendre_kernel_lean_f32") &&
            lean_kernel_usable<&legendre_kernel_lean_f32_w2>("legendre_kernel_lean_f32_w2") &&
            lean_kernel_usable<&legendre_kernel_lean_n<1, float>>("legendre_kernel_lean_n<1, float>") &&
            lean_kernel_usable<&legendre_kernel_lean_n<2, float>>("legendre_kernel_lean_n<2, float>") &&
            lean_kernel_usable<&legendre_kernel_lean_n_f32_w2<1>>("legendre_kernel_lean_n_f32_w2<1>") &&
            lean_kernel_usable<&legendre_kernel_lean_n_f32_w2<2>>("legendre_kernel_lean_n_f32_w2<2>")) {
            if (hipError_t err = launch_legendre_f32_rtw(p, 3, 2, nfull, nitems, 0, nfull, stream); err != hipSuccess) {
                return err;
            }
            LegendreParamsF32 q = p;
            q.col0              = nfull * 96;
            return launch_legendre_f32_rtw(q, rem_rtw, 2, 1, nitems, 0, 1, stream);
        }
    }
    return launch_legendre_f32_rtw(p, rtw, nrg, nchunks, nitems, chunk0, nrun, stream);
}

static hipError_t launch_legendre_f32_rtw(const LegendreParamsF32& p, int rtw, int nrg, int nchunks, int nitems, int chunk0, int nrun, hipStream_t stream) {
    if (rtw == 3 && nrg == 2) {
        // the 96-column workgroup in its "lean" form for float as well [r3]; ATLAS_AMD_LEG_KERNEL=classic: the generic template
        const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_KERNEL");
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (e && std::string(e) == "stream" && lean_kernel_usable<&legendre_kernel_stream_f32>("legendre_kernel_stream_f32")) {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            return launch_stream_t<&legendre_kernel_stream_f32, 3, float>(p, nitems, nchunks, chunk0, nrun, stream);
        }
#endif
        // [r6] pairs of latitude tiles per workgroup, the default from T = 400 on (measured, Legendre stage, ms: TL1279 -> F1280 5.66 ->
        // 5.13, -> O1280 4.57 -> 4.27, -> N1280 5.22 -> 4.77, the 411-field vor/div call 13.55 -> 12.70; TL639 -> O640 0.77 -> 0.75;
        // TL319 and below: no difference, the 64-latitude form stays).  ATLAS_AMD_LEG_F32_TILES=1 | 2 forces one.  Same bits.
        const char* w = atlas_amd::env_get("ATLAS_AMD_LEG_F32_TILES");
        const bool pairs = w ? atoi(w) == 2 : (AA_LEG_F32_TILES_DEFAULT == 2 && p.T >= 400);
        if (pairs && (!e || std::string(e) == "lean") && p.items2 && p.nitems2 > 0 &&
            lean_kernel_usable<&legendre_kernel_lean_f32_w2>("legendre_kernel_lean_f32_w2")) {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            return launch_lean_f32_w2(p, nchunks, chunk0, nrun, stream);
        }
        if ((!e || std::string(e) == "lean" || std::string(e) == "stream") && lean_kernel_usable<&legendre_kernel_lean_f32>("legendre_kernel_lean_f32")) {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            return launch_lean_f32(p, nitems, nchunks, chunk0, nrun, stream);
        }
    }
    if (nrg == 2 && (rtw == 1 || rtw == 2)) {
        const char* e = atlas_amd::env_get("ATLAS_AMD_LEG_KERNEL");
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (e && std::string(e) == "stream") {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            if (rtw == 1 && lean_kernel_usable<&legendre_kernel_stream_n<1, float>>("legendre_kernel_stream_n<1, float>")) {
                return launch_stream_t<&legendre_kernel_stream_n<1, float>, 1, float>(p, nitems, nchunks, chunk0, nrun, stream);
            }
            if (rtw == 2 && lean_kernel_usable<&legendre_kernel_stream_n<2, float>>("legendre_kernel_stream_n<2, float>")) {
                return launch_stream_t<&legendre_kernel_stream_n<2, float>, 2, float>(p, nitems, nchunks, chunk0, nrun, stream);
            }
        }
#endif
        if (!e || std::string(e) == "lean" || std::string(e) == "stream") {
            if (nrun <= 0) {
                chunk0 = 0;
                nrun   = nchunks;
            }
            const char* w    = atlas_amd::env_get("ATLAS_AMD_LEG_F32_TILES");
            const bool pairs = (w ? atoi(w) == 2 : (AA_LEG_F32_TILES_DEFAULT == 2 && p.T >= 400)) && (!e || std::string(e) == "lean") &&
                               p.items2 && p.nitems2 > 0;
            if (pairs && rtw == 1 && lean_kernel_usable<&legendre_kernel_lean_n_f32_w2<1>>("legendre_kernel_lean_n_f32_w2<1>")) {
                return launch_lean_n_f32_w2<1>(p, nchunks, chunk0, nrun, stream);
            }
            if (pairs && rtw == 2 && lean_kernel_usable<&legendre_kernel_lean_n_f32_w2<2>>("legendre_kernel_lean_n_f32_w2<2>")) {
                return launch_lean_n_f32_w2<2>(p, nchunks, chunk0, nrun, stream);
            }
            if (rtw == 1 && lean_kernel_usable<&legendre_kernel_lean_n<1, float>>("legendre_kernel_lean_n<1, float>")) {
                return launch_lean_n<1, float>(p, nitems, nchunks, chunk0, nrun, stream);
            }
            if (rtw == 2 && lean_kernel_usable<&legendre_kernel_lean_n<2, float>>("legendre_kernel_lean_n<2, float>")) {
                return launch_lean_n<2, float>(p, nitems, nchunks, chunk0, nrun, stream);
            }
        }
    }
    return launch_legendre_t<float>(p, nitems, chunk0, nrun, stream);
}

}  // namespace trans
}  // namespace atlas_amd
