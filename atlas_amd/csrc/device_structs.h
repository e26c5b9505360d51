// Plain structs passed by value to the HIP kernels (shared between the host launch code and the kernels).
#pragma once
#include <cstdint>

#include "fft_core.h"
#include "fft_plan.h"

namespace atlas_amd {
namespace trans {

constexpr int LEG_BN_DEV = 64;  // must equal trans_plan.h: LEG_BN
constexpr int LEG_KB_DEV = 8;   // must equal trans_plan.h: LEG_KB

struct LegendreItemDev {  // mirrors trans_plan.h: LegendreItem
    int m;
    int tile;
    int nrows;
    int kpad;
    long long p_off;
};

template <class Real>
struct LegendreParamsT {
    const Real* P;               // tile-blocked Legendre table
    const Real* sp;              // spectra, layout of TransLocal.cc:970-987 with truncation trc_in
    const long long* sp_moff;    // null: `sp` is the full array (block of wavenumber m at the reference's offset); else [T+2]: `sp`
                                 // holds only this rank's wavenumbers back to back, block m at sp_moff[m] * nf doubles [r3]
    Real* F;                     // Fourier intermediate F[(lat*(T+1)+m)*RP + r]
    const LegendreItemDev* items;  // launch-ordered work items
    const LegendreItemDev* items2; // the same paired: two consecutive tiles of one m per item (legendre_kernel_lean2), or null
    int nitems2;
    const int* nlat0;              // [T+1]
    const Real* zero;            // a 0.0 in device memory (target of the spectra loads of padding columns)
    int T;
    int trc_in;  // truncation of the input layout (T, or T+1 on the vor/div path)
    int nf;
    int RP;
    int nlats;
    int m_div;   // m-sharding: this device owns wavenumbers m with m % m_div == part; local index m / m_div
    int m_cnt;   // number of owned wavenumbers == m-extent of F:  F[(lat*m_cnt + m/m_div)*RP + r]
    int row_begin, row_end;  // latitudes this device stores (band decomposition: its band; else all), F rows are
                             // relative to row_begin
    int nitems;  // filled in by the launcher
    int nchunks; // column chunks per item (filled in by the launcher)
    int chunk0;  // first column chunk of this launch
    int nchunks_run;  // column chunks computed by this launch (pipelined transform: a subset)
    int col0;         // first interleaved column of chunk 0 of this launch [r6]: a call may be two launches of different chunk widths
    int* sched;       // [16] ints, zero between launches: unit counters of the persistent ("stream") kernels per XCD [0..7], workgroups
                      // that have finished [8]; the last workgroup to finish zeroes them again.  null: static unit assignment
    int abl;          // dev switch of the role-split kernel (ATLAS_AMD_LEG_ABLATE): parts left out, results then wrong; 0 in production
};
using LegendreParams    = LegendreParamsT<double>;
using LegendreParamsF32 = LegendreParamsT<float>;   // fp32 variant (BASELINE config C5)

// Everything a workgroup of the specialised Bluestein kernels needs to know about its row, in ONE 64-byte record read with one
// scalar load: the kernels used to walk rows[] -> row_plan[] -> plans[] (+ rowoff[], row_mmax[], coslatinv[]), three dependent
// L2 round trips (~2 us) before the first mode could be requested (per-wavefront trace, profiles/r03_fft_trace.txt) [r3]
struct alignas(64) FftRowDesc {
    int row;                 // latitude row (global index)
    int mmax;                // highest kept wavenumber, already clamped to h
    int h, n;                // half length, length
    long long goff_rel;      // rowoff[row] - rowoff[lat0]: offset of the row inside a field of the local band
    double coslatinv;        // 1 / cos(lat)
    long long off_tw, off_pre, off_chirp, off_bhat_t;   // into FourierParams::table
};

// The same for a native mixed-radix row (tools/experiments/fft_native.h; experiments build): the stage list in execution order and
// the offsets of its tables, 128 bytes
#if !defined(ATLAS_AMD_EXPERIMENTS)
struct FftNatDesc;   // (only ever a null pointer in the product library)
#else
struct alignas(128) FftNatDesc {
    int row;                 // latitude row (global index)
    int mmax;                // highest kept wavenumber, already clamped to h
    int h, n;
    long long goff_rel;      // rowoff[row] - rowoff[lat0]
    double coslatinv;        // 1 / cos(lat)
    long long off_tw, off_pre;   // into FourierParams::table: exp(2 pi i t / h), t < h;  exp(2 pi i k / n), k < h
    int perm;                // into FourierParams::nat_table: element the fold writes Z[k] to, k < h
    int ns;                  // stages
    int radix[fft::NAT_MAX_STAGES], nb[fft::NAT_MAX_STAGES], stride[fft::NAT_MAX_STAGES], tab[fft::NAT_MAX_STAGES];
    int lds_elems;
    int pad_;
};
static_assert(sizeof(FftNatDesc) == 128, "one 128-byte scalar load per workgroup");
#endif

struct FourierParts {
    const double* base[fft::MAX_PARTS];
    const long long* rowoff[fft::MAX_PARTS];
    int cnt[fft::MAX_PARTS];
};

struct FourierParams {
    const FftRowDesc* desc;                   // [nrows] of this launch (specialised Bluestein classes), else null
    const FftNatDesc* ndesc;                  // [nrows] of this launch (native mixed-radix rows), else null
    const uint32_t* nat_table;                // fold permutations and stage tables of the native rows (fft_plan.h: FftPlanSet::nat_table)
    // Fourier intermediate pieces, one per m-owner (see fft_core.h: RowIO).  Piece 0 travels in the kernel arguments; with more
    // than one piece the kernels read FourierParts from device memory (as kernel arguments the 16 x 3 entries sat in scalar
    // registers for the whole kernel and pushed ~200 scalar spill moves per wavefront into the vector ALU) [r3]
    const double* part_base0;
    const long long* part_rowoff0;                 // packed form: offset (doubles) of every local row inside the piece; else unused
    int part_cnt0;
    int packed_cols;                               // packed form: doubles per (row, wavenumber) = 2 * nb_fields; 0: classic layout
    const long long* packed_rowbase;               // packed form, optional: [local row][nparts] offset (doubles, relative to part_base0) of
                                                   // the row's run inside piece `part` -- ONE table read per mode instead of the piece
                                                   // table's base + row-offset pointer + row offset (three dependent reads) [r5]
    int parts_shift;                               // log2(nparts) if nparts is a power of two (m -> piece, index by shift / mask), else -1
    const FourierParts* parts;                     // [nparts > 1] all pieces
    int nparts;
    int lat0;                         // first row of the local latitude band
    double* gp;                       // gp[f*npts + (rowoff[lat]-rowoff[lat0]) + i], npts = points of the local band
    const fft::FftRowPlan* plans;     // device copy of the plan structs
    const fft::cplx* table;           // device copy of all FFT tables
    const fft::cplxf* table_f32;      // the same rounded to float (fp32 variant: the direct rows run in fp32 arithmetic), or null
    const int* row_plan;              // [nlats] plan index of each row
    const int* row_mmax;              // [nlats] highest kept wavenumber of each row (-1: none)
    const long long* rowoff;          // [nlats+1]
    const int* rows;                  // rows handled by this launch (size class)
    int nrows;
    unsigned nvirt;                   // virtual blocks (row, field slots) of the launch, set by the launcher
    int jobs;                         // tools/experiments/fft_kernel_p.hip only (field groups a workgroup walks through); 1
    int pf_dist;                      // L2 prefetch of the modes of the job 8 * pf_dist further on (same XCD); 0: off
    int pf_sectors;                   // requests per 128-byte line of that prefetch (1, 2 or 4)
    int seq_ok;                       // experiments build: every row of this launch keeps at most 1280 wavenumbers (fft_ct_rows_seq.inc)
    int mid_rot;                      // row_ct3 rows with more than 256 middle butterflies: the wavefront that takes the second round rotates
                                      // with the job (0: always the first wavefront -- on the SIMD that holds the first wavefront of both jobs of a CU)
    int row_affinity;                 // FftRowDesc kernels: a row's field groups all on one XCD (fft_device.h: fft_unit_to_job)
    int coarse_n[3];                  // fft_rows_coarse_kernel: rows of the launch's list with Bluestein length 1024 / 512 / 256, in this order
                                      // (the list is sorted by descending row length); all zero: one field per workgroup as in round 3
    int job_group_log2;               // kernels without a row record: log2 of the fields per job group (3; fp32 variant: 4 -- a 128-byte line
                                      // of the intermediate holds 16 fields there)
    int T;
    int RP;
    int nf;
    int f32;                          // fp32 variant: part_base and gp point to float arrays (same element indexing)
    int f_begin, f_end;               // fields transformed by this launch (pipelined transform: a subset)
    long long npts;
    int scale_uv_fields;              // first 2*nb_vordiv fields are multiplied by 1/cos(lat) (TransLocal.cc:1443-1469)
    const double* coslatinv;          // [nlats]
    int abl;                          // dev builds (-DAA_FFT_ABLATE) only: access-ablation bits, see fft_core.h
    unsigned long long* prof;         // optional [64] per-phase cycle accumulators (dev profiling), else null
    unsigned long long* trace;        // dev builds (-DAA_FFT_TRACE) only: 8 words per wavefront (hardware id, s_memtime at the
    unsigned long long trace_cap;     // phase boundaries), slot = (block * waves per workgroup + wave) * 8; capacity in words
};

}  // namespace trans
}  // namespace atlas_amd
