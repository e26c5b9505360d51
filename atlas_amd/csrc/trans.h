// atlas_amd::trans::Trans -- MI355X implementation of the TransLocal inverse transform for global structured
// grids.  Mirrors the subset of atlas::trans::TransImpl that TransLocal implements
// (reference: src/atlas/trans/detail/TransImpl.h:116-181, src/atlas/trans/local/TransLocal.cc:818-934,1409-1597):
//   invtrans(nb_scalar, sp, gp), invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp), invtrans(nb_vordiv, vor, div, gp),
//   truncation(), nb_spectral_coefficients(), grid size;  dirtrans / adjoints are "not implemented" there as well
//   (TransLocal.cc:848-857,899-927,1599-1685).
// Host-pointer overloads stage through device memory; *_device overloads take device pointers and are
// asynchronous on the object's stream.  One in-flight call per object (as for TransLocal, SURVEY 8b "Threading").
//
// Multi-GPU (new capability; TransLocal itself refuses mpi::size()>1, TransLocal.cc:338-340): an object created
// with (nparts, part) owns the zonal wavenumbers m % nparts == part for the Legendre stage and the latitude band
// latitude_bands()[part] .. [part+1] for the Fourier stage; the caller exchanges the Fourier intermediate
// (all-to-all, one contiguous slab per peer) between legendre_device() and fourier_device().
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "fft_plan.h"
#include "gaussian.h"
#include "legendre_host.h"
#include "trans_plan.h"

namespace atlas_amd {
namespace trans {

struct TransConfig {
    bool profile               = false;    // record HIP events around the two stages
    const void* legendre_cache = nullptr;  // optional Legendre cache blob (reference layout: sym ++ asym)
    size_t legendre_cache_size = 0;
    int nparts                 = 1;  // m-sharding / latitude-band decomposition
    int part                   = 0;
    // (Legendre row, latitude in radians): the table row is generated at this latitude instead of the grid's -- RegionalTrans with
    // ATLAS_AMD_REFERENCE_POLES=1 only (the reference's south-pole row, csrc/regional_trans.hip)
    std::vector<std::pair<int, double>> leg_lat_override;
    int row_begin = 0, row_end = 0;        // row_end > row_begin: transform only these latitude rows (a zonal-band
                                           // crop of the global grid, TransLocal.cc:394-470 "nested" case)
    bool by_band               = false;  // nparts > 1: false = wavenumber sharding (all-to-all transposition follows),
                                         // true = latitude-band sharding of both stages (no exchange, 2x Legendre work)
    int ndgl = 0, nxmax = 0;          // the grid is a row subset of a global grid with ndgl latitudes / longest row nxmax
                                      // (mirror-band decomposition): what fourier_truncation must see; 0: the grid's own
    std::vector<int> win_i0, win_n;   // per row of [row_begin, row_end): keep the win_n points starting at index win_i0,
                                      // wrapping around (a RectangularDomain crop, TransLocal.cc:1120-1135); empty: whole rows
    int device_tables          = -1;  // Legendre table computed on the device (1) or on the host and uploaded (0);
                                      // -1: environment variable ATLAS_AMD_TABLES=device|host, default device
};

struct StageTimings {
    double legendre_ms = 0, fourier_ms = 0;
    int legendre_calls = 0, fourier_calls = 0;
    double prepare_ms = 0;   // vor/div calls: spectra_prepare (extend_truncation + vd2uv + field interleave, vd2uv_kernel.hip)
    int prepare_calls = 0;
};

struct FourierParts;
class Trans {
public:
    Trans(const grid::StructuredGrid& grid, int truncation, const TransConfig& cfg = TransConfig());
    ~Trans();
    void release() noexcept;   // frees every device resource (destructor; constructor that throws)
    Trans(const Trans&)            = delete;
    Trans& operator=(const Trans&) = delete;

    int truncation() const { return geo_.T; }
    // points of the output: the rows of the band, or of their longitude windows
    int64_t nb_gridpoints() const { return windowed() ? win_npts_ : band_points(); }
    int64_t band_points() const { return geo_.rowoff[band_end()] - geo_.rowoff[band_begin()]; }
    bool windowed() const { return !cfg_.win_n.empty(); }
    int64_t nb_gridpoints_global() const { return geo_.npts; }
    size_t nb_spectral_coefficients() const { return size_t(geo_.T + 1) * size_t(geo_.T + 2); }  // TransLocal.h:90
    const TransGeometry& geometry() const { return geo_; }
    const LegendreWork& legendre_work() const { return work_; }
    const fft::FftPlanSet& fft_plans() const { return fftplans_; }
    // which kernel a row with this plan is launched with: 0 run-time shaped (fft_rows_kernel), 1 specialised Bluestein
    // (fft_rows_ct_kernel), 2 specialised direct (fft_rows_dct_kernel), 3 dense-stage experiment, 4 native mixed radix
    int fft_row_kernel(const fft::FftRowPlan& pl) const;
    int nparts() const { return cfg_.nparts; }
    int part() const { return cfg_.part; }
    int band_begin() const { return bands_[cfg_.part]; }
    int band_end() const { return bands_[cfg_.part + 1]; }
    // producers the Fourier stage gathers from: the wavenumber owners, or just this device (latitude-band decomposition)
    int fourier_parts() const { return cfg_.by_band ? 1 : cfg_.nparts; }
    const std::vector<int>& bands() const { return bands_; }
    int owned_wavenumbers() const { return m_cnt_; }
    hipStream_t stream() const { return stream_; }
    void set_stream(hipStream_t s);
    void synchronize() const;

    // ---- device-pointer API (asynchronous on stream()) ----
    void invtrans_scalar_device_f32(int nb_fields, const float* sp_dev, float* gp_dev);  // fp32 variant
    void invtrans_device_f32(int nb_scalar, const float* sp_dev, int nb_vordiv, const float* vor_dev, const float* div_dev,
                             float* gp_dev);                                              // ... its vor/div call [r5]
    void invtrans_uv_device_f32(int trc_in, int nb_fields, int nb_vordiv, const float* sp_dev, float* gp_dev);
    void invtrans_scalar_f32(int nb_fields, const float scalar_spectra[], float gp_fields[]);  // host pointers
    // TransLocal::invtrans_uv (TransLocal.cc:1409-1484); the first 2*nb_vordiv fields are scaled by 1/cos(lat)
    void invtrans_uv_device(int trc_in, int nb_fields, int nb_vordiv, const double* sp_dev, double* gp_dev);
    // the two stages separately (multi-GPU driver, stage-level parity tests)
    void legendre_device(int trc_in, int nb_fields, const double* sp_dev, double* fourier_dev);
    // [r3] the same from a spectral array that holds only this object's wavenumbers (m % nparts == part), their blocks in
    // the reference's inner layout back to back in increasing m (spectral_shard_offsets): SURVEY 8(e) scatters the input by m
    void legendre_device_sharded(int nb_fields, const double* sp_shard_dev, double* fourier_dev);
    // offset of wavenumber m's block in that array, in doubles per field (-1: not owned), and the array's size per field
    long long spectral_shard_offsets(std::vector<long long>& moff) const;
    void fourier_device(int nb_fields, int nb_vordiv, const double* fourier_dev, double* gp_dev);
    void fourier_device(int nb_fields, int nb_vordiv, const double* const* part_base, const int* part_cnt,
                        double* gp_dev);
    // packed pieces (dist_trans.h: PackedTransposePlan): part_rowoff_dev[i][r] = offset in doubles of local row r inside piece i
    // rowbase_dev (optional): [local row][nparts] = (part_base[i] - part_base[0]) + part_rowoff_dev[i][r], the combined table the
    // kernels read once per mode instead of walking the piece table
    void fourier_device_packed(int nb_fields, int nb_vordiv, const double* const* part_base,
                               const long long* const* part_rowoff_dev, int cols, double* gp_dev,
                               const long long* rowbase_dev = nullptr);
    size_t fourier_doubles(int nb_fields) const;  // local Fourier intermediate: nlats * owned m * RP
    int fourier_row_pitch(int nb_fields) const;   // RP = 16*ceil(2*nb_fields/16)
    double* fourier_buffer(int nb_fields);        // scratch intermediate owned by the object (grown on demand)

    // TransLocal::invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp) (TransLocal.cc:1523-1597): gp holds
    // [u fields][v fields][scalar fields]; vor/div -> U,V in spectral space, truncation extended to T+1
    void invtrans_device(int nb_scalar_fields, const double* sp_dev, int nb_vordiv_fields, const double* vor_dev,
                         const double* div_dev, double* gp_dev);

    // ---- host-pointer API (synchronous) : TransLocal.cc:931-934, 1486-1490, 1523-1597 ----
    void invtrans(int nb_scalar_fields, const double scalar_spectra[], double gp_fields[]);
    void invtrans(int nb_scalar_fields, const double scalar_spectra[], int nb_vordiv_fields,
                  const double vorticity_spectra[], const double divergence_spectra[], double gp_fields[]);

    // Legendre cache, byte-compatible with TransLocal's write_legendre file (TransLocal.cc:638-647)
    size_t legendre_cache_bytes() const { return (geo_.size_sym() + geo_.size_asym()) * sizeof(double); }
    void export_legendre_cache(void* buffer) const;
    void download_legendre_table(double* out, size_t size_doubles) const;  // test hook
    bool tables_on_device() const { return tables_on_device_; }

    StageTimings timings();
    void reset_timings() {
        collect_timings();
        timings_ = StageTimings();
    }
    void set_profile(bool on) { profile_ = on; }
    bool profile() const { return profile_; }
    // dev profiling: per-phase shader-clock totals of the FFT kernel (thread 0 of every workgroup)
    void enable_phase_profile(bool on);
    void read_phase_profile(unsigned long long out[64]);
    // dev builds (-DAA_FFT_TRACE): out == nullptr allocates and zeroes a trace buffer of `words` 64-bit words (0 frees it);
    // otherwise copies the first `words` words out (layout: device_structs.h, FourierParams::trace)
    void fft_trace(unsigned long long words, unsigned long long* out);

private:
    void upload();
    void collect_timings();
    void timed_begin(int kind, hipStream_t s = nullptr);
    void legendre_chunks(int trc_in, int nb_fields, const double* sp_dev, double* fourier_dev, int chunk0, int nrun,
                         bool sharded_input = false);
    void fourier_fields(int nb_fields, int nb_vordiv, const double* const* part_base, const int* part_cnt,
                        double* gp_dev, int f_begin, int f_end, hipStream_t stream, bool f32 = false,
                        const long long* const* part_rowoff_dev = nullptr, int packed_cols = 0,
                        const long long* packed_rowbase_dev = nullptr);
    void timed_end();

    TransGeometry geo_;
    TransConfig cfg_;
    LegendreWork work_;
    fft::FftPlanSet fftplans_;
    bool fft_coarse_ = false;   // coarse row classes (fft_plan.h: PlanOptions::coarse_classes): small reduced grid, one class stream
    std::vector<int> bands_;
    int m_cnt_          = 0;
    bool profile_       = false;
    bool use_ct_        = true;  // ATLAS_AMD_FFT_GENERIC=1 forces the generic FFT kernel (A/B comparisons)
    bool tables_on_device_ = false;  // the Legendre table was generated by legendre_gen_kernel.hip
    void generate_table_on_device();
    hipStream_t stream_ = nullptr;
    hipStream_t stream2_ = nullptr;            // Fourier stage of the pipelined transform
    hipStream_t ev_stream_ = nullptr;
    std::vector<hipStream_t> side_streams_;     // the row-length classes of the Fourier stage are dealt to several streams
    std::vector<hipEvent_t> side_joins_;
    hipEvent_t side_fork_ = nullptr;
    std::vector<hipEvent_t> pipe_events_;
    int pipeline_ = 1;                          // pieces of the Legendre/Fourier software pipeline (1: off)
    bool own_stream_    = false;

    // device state
    double* d_P_         = nullptr;
    void* d_items_       = nullptr;
    void* d_items2_      = nullptr;  // paired work list (two consecutive tiles of one m per item)
    int nitems2_         = 0;
    int* d_nlat0_        = nullptr;
    long long* d_sp_moff_ = nullptr;   // block offsets of the owned wavenumbers in a sharded spectral array (lazily)
    int* d_leg_sched_    = nullptr;  // 16 ints: work counters of the persistent Legendre kernels (self-resetting)
    double* d_zero_      = nullptr;  // zeros: load target of padding columns in the Legendre kernel
    float* d_P32_        = nullptr;  // fp32 variant: table, zero target and Fourier intermediate in float (lazily)
    float* d_zero32_     = nullptr;
    float* d_fourier32_  = nullptr;
    size_t fourier32_cap_ = 0;
    void* d_fftplans_    = nullptr;
    void* d_ffttable_    = nullptr;
    void* d_ffttable_f32_ = nullptr;   // float copy, uploaded by the first fp32 call
    uint32_t* d_nat_table_ = nullptr;  // fold permutations and stage tables of the native mixed-radix rows (fft_native.h)
    int* d_row_plan_     = nullptr;
    int* d_row_mmax_     = nullptr;
    long long* d_rowoff_ = nullptr;
    double* d_coslatinv_ = nullptr;
    struct SizeClass {
        int lds_bytes;
        int nthreads;
        int ct_f, ct_k;  // specialised kernel instance, or ct_k < 0
        bool direct;     // specialised instance is the direct (no Bluestein) kernel
        bool hybrid = false;  // dense-stage rows (fft_rows_hyb_kernel)
        int max_mmax = -1;    // highest kept wavenumber of any row of the class (specialised Bluestein rows)
        bool coarse_fused = false;   // the coarse Bluestein classes 256 / 512 / 1024 of a small reduced grid in one launch
        int coarse_n[3]   = {0, 0, 0};   // ... rows of Bluestein length 1024 / 512 / 256 in the (sorted) list
        bool native = false;  // native mixed-radix rows (fft_rows_nat_kernel): d_desc holds FftNatDesc records
        bool native_bigp = false;   // ... whose first-stage radix is a prime 17 .. 31 (the kernel instance with 168 registers)
        int native_fpj   = 1;       // ... fields per workgroup (1 or 2)
        int nrows;
        int* d_rows;
        // run-time shaped classes [r6]: the list split into its direct rows (the fp32 variant runs them two fields per job,
        // fft_kernel_pairs.hip: fft_rows_pair_kernel) and the rest (odd lengths, ...); both null where the class has no such split
        int nrows_pair = 0, nrows_rest = 0;
        int* d_rows_pair = nullptr;
        int* d_rows_rest = nullptr;
        void* d_desc = nullptr;   // FftRowDesc[nrows] for the specialised Bluestein kernels
    };
    std::vector<SizeClass> classes_;
    // rows whose transform does not fit a CU's LDS (more than 10 080 complex elements: the four longest row lengths of O2560, regular
    // grids beyond F5040): their c2r sum as a matrix product with a cos / sin table on fp64 MFMA (dft_gemm.h) [r6]; one group per length
    struct GemmRows {
        int n = 0, nrows = 0;
        int* d_rows         = nullptr;   // rows (global index)
        int* d_rowsel       = nullptr;   // 0 .. nrows - 1: row of the dense coefficient array
        long long* d_rowout = nullptr;   // offset of the row inside a field of the band
        double* d_rowscale  = nullptr;   // 1 / cos(lat)
        int* d_rowmmax      = nullptr;   // highest kept wavenumber
        double* d_table     = nullptr;   // [2 (T + 1)][n]
    };
    std::vector<GemmRows> gemm_rows_;
    double* d_gemm_dense_  = nullptr;    // the kept wavenumbers of a group's rows, gathered from the intermediate (any layout) in front of the product
    size_t gemm_dense_cap_ = 0;
public:
    // launches of one Fourier stage (one per row class) and, of those, launches that take two fields per workgroup (native rows)
    void fourier_launch_plan(int out[3]) const {
        out[0] = (int)classes_.size();
        out[1] = out[2] = 0;
        for (const SizeClass& c : classes_) {
            out[1] += c.coarse_fused ? 1 : 0;
            out[2] += (c.native && c.native_fpj == 2) ? 1 : 0;
        }
    }
private:
    double* d_fourier_  = nullptr;
    size_t fourier_cap_ = 0;
    double* d_sp_       = nullptr;
    size_t sp_cap_      = 0;
    double* d_gp_       = nullptr;
    size_t gp_cap_      = 0;
    double* d_gpfull_   = nullptr;  // whole rows of a longitude-window crop, before the window is copied out
    size_t gpfull_cap_  = 0;
    int* d_win_i0_      = nullptr;
    int* d_win_n_       = nullptr;
    long long* d_win_off_ = nullptr;
    int64_t win_npts_   = 0;
    double* d_all_      = nullptr;  // combined (U,V,scalar) spectra of the vor/div path
    size_t all_cap_     = 0;
    double* d_vd_       = nullptr;  // host-API staging of vor ++ div
    unsigned long long* d_prof_ = nullptr;
    unsigned long long* d_trace_ = nullptr;
    std::vector<std::pair<std::vector<unsigned char>, void*>> parts_cache_;   // piece tables (device_structs.h: FourierParts)
    size_t parts_evict_ = 0;                                                  // on the device, by content
    const FourierParts* device_parts(const FourierParts& hp);
public:
    // the piece tables are cached by content (the producers' buffer pointers): a driver that frees those buffers drops them
    void clear_fourier_parts_cache();
private:
    unsigned long long trace_cap_ = 0;
    size_t vd_cap_      = 0;
    void ensure(double*& ptr, size_t& cap, size_t n);
    // host-pointer pipeline (pinned staging)
    // false: the staging buffers could not be allocated, nothing was done -- the caller runs the serial path
    bool invtrans_host_pipelined(int nb_scalar, const double* sp_host, int nb_vordiv, const double* vor_host, const double* div_host,
                                 double* gp_host);
    bool ensure_host_pipeline_buffers(size_t up_doubles, size_t down_doubles);
    void invtrans_scalars_extended_device(int nb_scalar, const double* sp_dev, double* gp_dev);
    double* hp_up_[2]   = {nullptr, nullptr};   // pinned: a chunk's spectra / grid points
    double* hp_down_[2] = {nullptr, nullptr};
    double* hp_dsp_[2]  = {nullptr, nullptr};   // device: the same
    double* hp_dgp_[2]  = {nullptr, nullptr};
    size_t hp_up_cap_ = 0, hp_down_cap_ = 0;    // doubles
    hipEvent_t hp_up_done_[2]   = {nullptr, nullptr};
    hipEvent_t hp_comp_done_[2] = {nullptr, nullptr};
    hipEvent_t hp_down_done_[2] = {nullptr, nullptr};
    hipStream_t hp_up_stream_ = nullptr;
    hipStream_t copy_stream_  = nullptr;
    std::vector<hipEvent_t> events_;  // pairs (begin, end)
    std::vector<int> ev_kind_;        // 0 legendre, 1 fourier, per pair
    size_t ev_used_ = 0;
    StageTimings timings_;
};

}  // namespace trans
}  // namespace atlas_amd
