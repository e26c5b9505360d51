// See trans.h.  Host code of the MI355X TransLocal replacement: builds the plan, uploads the tables once,
// launches the two kernels per call on the object's HIP stream.
#include "env.h"
#include "trans.h"
#include "dft_gemm.h"
#include "host_copy.h"
#include "trace.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <sstream>
#include <stdexcept>

#include "device_structs.h"
#include "legendre_gen_core.h"
#include "legendre_host.h"

namespace atlas_amd {
namespace trans {

hipError_t launch_legendre(const LegendreParams& p, int nitems, int chunk0, int nrun, hipStream_t stream);
hipError_t launch_window_crop(const double* full, double* out, const long long* rowoff, const int* win_i0, const int* win_n,
                              const long long* win_off, int nrows, long long npts_full, long long npts_out, int f_begin,
                              int f_end, hipStream_t stream);
hipError_t launch_legendre_gen(const LegendreGenParams& g, hipStream_t stream);
void legendre_tiling(int nf, int& rtw, int& nrg, int& nchunks);
hipError_t launch_legendre_f32(const LegendreParamsF32& p, int nitems, int chunk0, int nrun, hipStream_t stream);
hipError_t launch_convert_f64_f32(const double* src, float* dst, size_t n, hipStream_t stream);
hipError_t launch_fourier(const FourierParams& p, int lds_bytes, int nthreads, hipStream_t stream);
bool fourier_generic_pairs_usable(const FourierParams& p);   // fft_kernel_pairs.hip
hipError_t launch_fourier_generic_pairs(const FourierParams& p, int lds_bytes, int nthreads, hipStream_t stream);
hipError_t launch_fourier_ct(const FourierParams& p, int ctf, int ctk, int lds_bytes, int nthreads,
                             hipStream_t stream);
hipError_t launch_fourier_hyb(const FourierParams& p, int lds_bytes, int nthreads, hipStream_t stream);
hipError_t launch_fourier_nat(const FourierParams& p, int lds_bytes, int bigp, int fields_per_job, hipStream_t stream);   // fft_native.hip
hipError_t launch_fourier_coarse(const FourierParams& p, int lds_bytes, hipStream_t stream);
hipError_t launch_fourier_dct(const FourierParams& p, int ctf, int ctk, int lds_bytes, int nthreads,
                              hipStream_t stream);
hipError_t launch_spectra_prepare(const double* vor, const double* div, const double* sp, double* out, int T, int nvd,
                                  int ns, hipStream_t stream);
hipError_t launch_spectra_prepare_f32(const float* vor, const float* div, const float* sp, float* out, int T, int nvd, int ns,
                                      hipStream_t stream);

namespace {
void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        std::ostringstream ss;
        ss << "HIP error '" << hipGetErrorString(e) << "' in " << what << " (" << file << ":" << line << ")";
        (void)hipGetLastError();   // the failure is reported here, once: not left sticky for the launch checks of the next call
        throw std::runtime_error(ss.str());
    }
}
#define HIP_CHECK(x) hip_check((x), #x, __FILE__, __LINE__)

template <typename T>
T* dev_upload(const T* host, size_t n) {
    T* d = nullptr;
    if (n == 0) {
        n = 1;
        HIP_CHECK(hipMalloc((void**)&d, sizeof(T)));
        return d;
    }
    HIP_CHECK(hipMalloc((void**)&d, n * sizeof(T)));
    HIP_CHECK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
}  // namespace

Trans::Trans(const grid::StructuredGrid& grid, int truncation, const TransConfig& cfg):
    geo_(make_geometry(grid, truncation, cfg.ndgl, cfg.nxmax)), cfg_(cfg), profile_(cfg.profile) {
    if (cfg.nparts < 1 || cfg.nparts > fft::MAX_PARTS || cfg.part < 0 || cfg.part >= cfg.nparts) {
        throw std::invalid_argument("Trans: invalid (nparts, part)");
    }
    for (const auto& ov : cfg.leg_lat_override) {
        if (ov.first < 0 || ov.first >= (int)geo_.lats_leg.size()) {
            throw std::invalid_argument("Trans: leg_lat_override row out of range");
        }
        geo_.lats_leg[ov.first] = ov.second;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        throw std::runtime_error(
            "atlas_amd::Trans needs a HIP device (MI355X / gfx950); there is no CPU fallback for the transform");
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_PIPELINE")) {
        pipeline_ = std::max(1, atoi(e));
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_GENERIC")) {
        use_ct_ = !(e[0] == '1');
    }
    if (cfg.row_end > cfg.row_begin) {
        // zonal-band crop: the machinery of the latitude-band decomposition with one explicit band
        if (cfg.nparts != 1 || cfg.row_begin < 0 || cfg.row_end > geo_.nlats) {
            throw std::invalid_argument("Trans: rows=j0:j1 needs nparts == 1 and 0 <= j0 < j1 <= ny");
        }
        cfg_.by_band = true;
        bands_       = {cfg.row_begin, cfg.row_end};
    }
    else {
        bands_ = latitude_bands(geo_, cfg.nparts);
    }
    if (!cfg.win_n.empty()) {
        // longitude windows (RectangularDomain crop): per row of rows=j0:j1
        const int nrows = cfg.row_end - cfg.row_begin;
        if (nrows <= 0 || (int)cfg.win_n.size() != nrows || (int)cfg.win_i0.size() != nrows) {
            throw std::invalid_argument("Trans: longitude windows need rows=j0:j1 and one (first index, count) per row");
        }
        win_npts_ = 0;
        for (int r = 0; r < nrows; ++r) {
            const int nx = geo_.nx[cfg.row_begin + r];
            if (cfg.win_i0[r] < 0 || cfg.win_i0[r] >= nx || cfg.win_n[r] < 1 || cfg.win_n[r] > nx) {
                throw std::invalid_argument("Trans: longitude window outside its row");
            }
            win_npts_ += cfg.win_n[r];
        }
    }
    work_ = make_legendre_work(geo_, cfg.nparts, cfg.part, cfg_.by_band, cfg.row_begin, cfg.row_end);
    m_cnt_ = 0;
    for (int m = cfg_.by_band ? 0 : cfg.part; m <= geo_.T; m += cfg_.by_band ? 1 : cfg.nparts) {
        m_cnt_++;
    }
    std::vector<int> lengths;
    if (geo_.regular) {
        lengths.push_back(geo_.nxmax);
    }
    else {
        lengths = geo_.nx;
    }
    {
        fft::PlanOptions po;
        po.specialised_shapes = use_ct_;
        po.max_mode           = geo_.T;
        if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_HYBRID")) {      // A/B switch: 0 = Bluestein for every awkward row
            po.hybrid = atoi(e) != 0;
#if !defined(ATLAS_AMD_EXPERIMENTS)
            if (po.hybrid) {   // the dense-stage kernel lives in tools/experiments: fail here, not at the first launch
                throw std::runtime_error("ATLAS_AMD_FFT_HYBRID=1 needs a library built with -DATLAS_AMD_EXPERIMENTS (make -C atlas_amd/csrc experiments)");
            }
#endif
        }
        if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_NATIVE")) {      // 1: native mixed-radix rows where a stage list exists
            po.native = atoi(e) != 0;
#if !defined(ATLAS_AMD_EXPERIMENTS)
            if (po.native) {   // kernel and planner live in tools/experiments since round 5 (at parity with Bluestein, never the default)
                throw std::runtime_error("ATLAS_AMD_FFT_NATIVE=1 needs a library built with -DATLAS_AMD_EXPERIMENTS (make -C atlas_amd/csrc experiments)");
            }
#endif
        }
        if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_HYB_MAXA")) {    // largest dense radix
            po.hybrid_max_a = atoi(e);
        }
        {
            // small reduced grids: a few coarse row classes instead of one per tight Bluestein length (fft_plan.h)
            std::vector<int> distinct(lengths);
            std::sort(distinct.begin(), distinct.end());
            distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
            // [r6] every reduced grid, not only those of at most 704 points per row: the rows short enough for a coarse length share
            // one launch on every grid (tools/probe/grid_sweep.sh: the Fourier stage of O176 .. O400 spent most of its time in 20 - 25
            // launches of a few microseconds of work: O200 0.32 -> 0.20 ms at 137 fields, O320 at 20 fields 0.18 -> 0.10; O640 -2.5 %,
            // O800 -3 %, O1280 -1 %); the rows beyond the coarse lengths keep their tight classes
            // -- for grids where (nearly) every pair of rows has a length of its own (the octahedral ones: one distinct length per two
            // rows).  The classic N grids beyond 704 points per row keep their tight classes: a few lengths with many rows each, most of
            // them lengths of the specialised direct family, which a coarse Bluestein row would replace at a loss (N320: 0.76 -> 0.90 ms)
            po.coarse_classes = !geo_.regular && distinct.size() >= 24 && (geo_.nxmax <= 704 || distinct.size() * 4 >= (size_t)geo_.nlats);
            if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_COARSE")) {
                po.coarse_classes = atoi(e) != 0;
            }
            fft_coarse_ = po.coarse_classes;
        }
        fftplans_ = fft::make_fft_plans(lengths, po);
    }
    // upload() allocates the table and every plan buffer and can throw (cache size, row length, out of memory): the
    // destructor does not run for a constructor that throws, so release what exists before passing the error on
    try {
        HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        own_stream_ = true;
        upload();
    }
    catch (...) {
        release();
        throw;
    }
}

Trans::~Trans() {
    release();
}

void Trans::release() noexcept {
    if (stream_) {
        (void)hipStreamSynchronize(stream_);
    }
    auto fr = [](auto*& p) {
        if (p) {
            (void)hipFree(p);
            p = nullptr;
        }
    };
    fr(d_P_);
    fr(d_items_);
    fr(d_items2_);
    fr(d_sp_moff_);
    fr(d_nlat0_);
    fr(d_zero_);
    fr(d_leg_sched_);
    fr(d_P32_);
    fr(d_zero32_);
    fr(d_fourier32_);
    fr(d_fftplans_);
    fr(d_ffttable_);
    fr(d_ffttable_f32_);
    fr(d_nat_table_);
    fr(d_gemm_dense_);
    for (GemmRows& g : gemm_rows_) {
        fr(g.d_rows);
        fr(g.d_rowsel);
        fr(g.d_rowout);
        fr(g.d_rowscale);
        fr(g.d_rowmmax);
        fr(g.d_table);
    }
    fr(d_row_plan_);
    fr(d_row_mmax_);
    fr(d_rowoff_);
    fr(d_coslatinv_);
    for (auto& c : classes_) {
        fr(c.d_rows);
        fr(c.d_rows_pair);
        fr(c.d_rows_rest);
        fr(c.d_desc);
    }
    classes_.clear();
    for (auto& e : parts_cache_) {
        fr(e.second);
    }
    parts_cache_.clear();
    fr(d_fourier_);
    fr(d_sp_);
    fr(d_gp_);
    fr(d_all_);
    fr(d_gpfull_);
    fr(d_win_i0_);
    fr(d_win_n_);
    fr(d_win_off_);
    fr(d_prof_);
    fr(d_trace_);
    fr(d_vd_);
    for (auto& e : events_) {
        (void)hipEventDestroy(e);
    }
    events_.clear();
    for (auto& e : pipe_events_) {
        (void)hipEventDestroy(e);
    }
    pipe_events_.clear();
    if (stream2_) {
        (void)hipStreamDestroy(stream2_);
        stream2_ = nullptr;
    }
    if (hp_up_stream_) {
        (void)hipStreamSynchronize(hp_up_stream_);
        (void)hipStreamSynchronize(copy_stream_);
        for (int i = 0; i < 2; ++i) {
            if (hp_up_[i]) (void)hipHostFree(hp_up_[i]);
            if (hp_down_[i]) (void)hipHostFree(hp_down_[i]);
            if (hp_dsp_[i]) (void)hipFree(hp_dsp_[i]);
            if (hp_dgp_[i]) (void)hipFree(hp_dgp_[i]);
            hp_up_[i] = hp_down_[i] = hp_dsp_[i] = hp_dgp_[i] = nullptr;
            (void)hipEventDestroy(hp_up_done_[i]);
            (void)hipEventDestroy(hp_comp_done_[i]);
            (void)hipEventDestroy(hp_down_done_[i]);
        }
        (void)hipStreamDestroy(hp_up_stream_);
        (void)hipStreamDestroy(copy_stream_);
    }
    for (auto st : side_streams_) {
        (void)hipStreamDestroy(st);
    }
    side_streams_.clear();
    for (auto ev : side_joins_) {
        (void)hipEventDestroy(ev);
    }
    side_joins_.clear();
    if (side_fork_) {
        (void)hipEventDestroy(side_fork_);
        side_fork_ = nullptr;
    }
    if (own_stream_ && stream_) {
        (void)hipStreamDestroy(stream_);
    }
    stream_ = nullptr;
}

void Trans::set_stream(hipStream_t s) {
    synchronize();
    if (own_stream_ && stream_) {
        (void)hipStreamDestroy(stream_);
    }
    stream_     = s;
    own_stream_ = false;
}

void Trans::synchronize() const {
    HIP_CHECK(hipStreamSynchronize(stream_));
}

// Legendre table computed on the device (legendre_gen_kernel.hip) from O(T^2) host-prepared inputs: no host
// generation of the O(T^2 N) table, no multi-GB upload
static constexpr size_t kTableSlack = 16 * 64;  // doubles

void Trans::generate_table_on_device() {
    const LegendreGenInputs in = prepare_legendre_gen(geo_, work_);
    const size_t n             = (size_t)work_.table_doubles;
    // + kTableSlack: the deep-stage Legendre kernel stages whole 12 / 16-row stages and may read (never use) up to 8 rows
    // past the last item's block
    HIP_CHECK(hipMalloc((void**)&d_P_, (n + kTableSlack) * sizeof(double)));
    HIP_CHECK(hipMemsetAsync(d_P_, 0, (n + kTableSlack) * sizeof(double), stream_));  // K / latitude padding
    struct Scratch {  // device buffers that only live for the generation, released on every exit path
        std::vector<void*> ptrs;
        void push_back(void* p) { ptrs.push_back(p); }
        ~Scratch() {
            for (void* p : ptrs) {
                (void)hipFree(p);
            }
        }
    } tmp;
    auto up = [&](const auto& v) {
        auto* d = dev_upload(v.data(), v.size());
        tmp.push_back((void*)d);
        return d;
    };
    LegendreGenParams g;
    g.trc = in.trc, g.T = in.T, g.nlats = in.nlats, g.lat_pitch = in.lat_pitch;
    g.zfn = up(in.zfn), g.sq1 = up(in.sq1), g.ca = up(in.ca), g.cb = up(in.cb), g.cc = up(in.cc);
    g.vcos = up(in.vcos), g.vsin = up(in.vsin), g.diag = up(in.diag), g.mu = up(in.mu);
    g.mstop = up(in.mstop);
    g.nlat0 = up(in.nlat0), g.first_item_of_m = up(in.first_item_of_m);
    g.item_p_off = up(in.item_p_off), g.item_kpad = up(in.item_kpad);
    double *col01 = nullptr, *rows = nullptr;
    HIP_CHECK(hipMalloc((void**)&col01, in.col01_doubles() * sizeof(double)));
    tmp.push_back(col01);
    HIP_CHECK(hipMalloc((void**)&rows, in.rows_doubles() * sizeof(double)));
    tmp.push_back(rows);
    g.col01 = col01, g.rows = rows, g.table = d_P_;
    HIP_CHECK(launch_legendre_gen(g, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    tables_on_device_ = true;
}

void Trans::download_legendre_table(double* out, size_t size_doubles) const {
    if (size_doubles != (size_t)work_.table_doubles) {
        throw std::invalid_argument("legendre_table_download: wrong size");
    }
    synchronize();
    if (size_doubles) {
        HIP_CHECK(hipMemcpy(out, d_P_, size_doubles * sizeof(double), hipMemcpyDeviceToHost));
    }
}

// LDS elements of ONE work array of a native row (the staging area of the gathered modes aliases it: whole 64-lane granules); a
// workgroup has one or two of them and the 64-element dump area of the L2 prefetch requests behind them (fft_native.hip)
#if defined(ATLAS_AMD_EXPERIMENTS)
static int native_lds_elems(const fft::FftRowPlan& pl, int row_mmax) {
    const int mmax = std::max(0, std::min(row_mmax, pl.h));
    return std::max(pl.nat.lds_elems, (mmax + 1 + 63) / 64 * 64);
}
#endif

void Trans::upload() {
    // ---- Legendre table (tile-blocked), owned wavenumbers only ----
    bool on_device = cfg_.device_tables == 1;
    if (cfg_.device_tables < 0) {
        // measured at TL1279 / O1280 (gpurun_out/r02_first): device generation 0.9 s, host generation + upload 6.7 s
        const char* e = atlas_amd::env_get("ATLAS_AMD_TABLES");
        on_device     = !(e && std::string(e) == "host");
    }
    if (on_device && !cfg_.legendre_cache) {
        generate_table_on_device();
    }
    else {
        const size_t n = (size_t)work_.table_doubles;
        double* host   = (double*)calloc(std::max<size_t>(n, 1), sizeof(double));
        if (!host) {
            throw std::runtime_error("out of host memory for the Legendre table");
        }
        if (cfg_.legendre_cache) {
            // TransLocal.cc:608-614: sym then asym, size must match exactly
            if (cfg_.legendre_cache_size != legendre_cache_bytes()) {
                free(host);
                throw std::invalid_argument("Legendre cache has the wrong size for this (grid, truncation)");
            }
            const double* sym  = (const double*)cfg_.legendre_cache;
            const double* asym = sym + geo_.size_sym();
            retile_legendre_tables(geo_, work_, sym, asym, host);
        }
        else {
            compute_legendre_table_tiled(geo_, work_, host);
        }
        HIP_CHECK(hipMalloc((void**)&d_P_, (n + kTableSlack) * sizeof(double)));
        HIP_CHECK(hipMemcpy(d_P_, host, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(d_P_ + n, 0, kTableSlack * sizeof(double)));
        free(host);
    }
    {
        std::vector<LegendreItemDev> items(work_.items.size());
        for (size_t i = 0; i < items.size(); ++i) {
            const LegendreItem& it = work_.items[i];
            items[i]               = LegendreItemDev{it.m, it.tile, it.nrows, it.kpad, (long long)it.p_off};
        }
        d_items_ = dev_upload(items.data(), items.size());
        {
        // paired list (legendre_kernel_lean_f32_w2 [r6]; legendre_kernel_lean2 of the experiments build): the launch order interleaves eight per-XCD lists (item i belongs to list i % 8);
        // inside a list the tiles of one m follow each other, and so do their blocks in the table
        std::vector<std::vector<LegendreItemDev>> lists(8);
        for (size_t i = 0; i < items.size(); ++i) {
            if (items[i].m >= 0) {
                lists[i % 8].push_back(items[i]);
            }
        }
        std::vector<std::vector<LegendreItemDev>> pairs(8);
        size_t maxlen = 0;
        for (int x = 0; x < 8; ++x) {
            for (size_t i = 0; i < lists[x].size(); ++i) {
                LegendreItemDev a = lists[x][i];
                if (i + 1 < lists[x].size()) {
                    const LegendreItemDev& b = lists[x][i + 1];
                    if (b.m == a.m && b.tile == a.tile + 1 && a.nrows == LEG_BN_DEV &&
                        b.p_off == a.p_off + 2ll * a.kpad * LEG_BN_DEV) {
                        a.nrows += b.nrows;
                        ++i;
                    }
                }
                pairs[x].push_back(a);
            }
            maxlen = std::max(maxlen, pairs[x].size());
        }
        std::vector<LegendreItemDev> items2;
        for (size_t q = 0; q < maxlen; ++q) {
            for (int x = 0; x < 8; ++x) {
                items2.push_back(q < pairs[x].size() ? pairs[x][q] : LegendreItemDev{-1, 0, 0, 0, 0});
            }
        }
        nitems2_  = (int)items2.size();
        d_items2_ = items2.empty() ? nullptr : dev_upload(items2.data(), items2.size());
        }
    }
    d_nlat0_ = dev_upload(geo_.nlat0.data(), geo_.nlat0.size());
    {
        const double zeros[16] = {0.};
        d_zero_                = dev_upload(zeros, 16);
        const int izeros[16]   = {0};
        d_leg_sched_           = dev_upload(izeros, 16);
    }
    // ---- FFT plans / tables ----
    d_fftplans_ = dev_upload(fftplans_.plans.data(), fftplans_.plans.size());
    d_ffttable_ = dev_upload(fftplans_.table.data(), fftplans_.table.size());
    if (!fftplans_.nat_table.empty()) {
        d_nat_table_ = dev_upload(fftplans_.nat_table.data(), fftplans_.nat_table.size());
    }
    std::vector<int> row_plan(geo_.nlats), row_mmax(geo_.nlats);
    std::vector<double> coslatinv(geo_.nlats);
    for (int j = 0; j < geo_.nlats; ++j) {
        const int n    = geo_.regular ? geo_.nxmax : geo_.nx[j];
        row_plan[j]    = fftplans_.plan_index(n);
        const int jleg = j < geo_.nlatsNH ? j : geo_.nlats - 1 - j;
        row_mmax[j]    = geo_.mmax_leg[jleg];
        double lat     = std::max(std::min(geo_.lat_deg[j], kLatPole), -kLatPole);  // TransLocal.cc:1449-1456
        coslatinv[j]   = 1. / std::cos(lat * (M_PI / 180.));
    }
    d_row_plan_  = dev_upload(row_plan.data(), row_plan.size());
    d_row_mmax_  = dev_upload(row_mmax.data(), row_mmax.size());
    d_coslatinv_ = dev_upload(coslatinv.data(), coslatinv.size());
    std::vector<long long> rowoff(geo_.rowoff.begin(), geo_.rowoff.end());
    d_rowoff_ = dev_upload(rowoff.data(), rowoff.size());
    if (windowed()) {
        std::vector<long long> woff(cfg_.win_n.size() + 1, 0);
        for (size_t r = 0; r < cfg_.win_n.size(); ++r) {
            woff[r + 1] = woff[r] + cfg_.win_n[r];
        }
        d_win_i0_  = dev_upload(cfg_.win_i0.data(), cfg_.win_i0.size());
        d_win_n_   = dev_upload(cfg_.win_n.data(), cfg_.win_n.size());
        d_win_off_ = dev_upload(woff.data(), woff.size());
    }
    // ---- launch classes for the rows of the local latitude band ----
    // Bluestein rows whose length M = F*2^K has a compile-time specialised kernel instance form one class per M;
    // everything else (short rows, {2,3,5}-smooth rows, odd rows) goes to the generic kernel, bucketed by LDS need.
    const int class_M[] = {256, 512, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192, 10080};
    std::map<std::pair<int, int>, std::vector<int>> by_class;  // (specialised ? 1 : 0, M bucket)
    std::map<int, std::vector<int>> gemm_by_n;   // row length -> rows whose transform does not fit a CU's LDS
    int lds_max = fft::padded_size(class_M[sizeof(class_M) / sizeof(class_M[0]) - 1]);
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_LDS_ELEMS")) {   // test hook: small grids through the matrix-product rows
        lds_max = std::min(lds_max, fft::padded_size(std::max(1, atoi(e))));
    }
    for (int j = band_begin(); j < band_end(); ++j) {
        const fft::FftRowPlan& pl = fftplans_.plans[row_plan[j]];
        if (pl.lds_complex > lds_max) {   // [r6] (the four longest row lengths of O2560: Bluestein length 12 288)
            gemm_by_n[pl.n].push_back(j);
            continue;
        }
        if (pl.method == fft::FFT_BLUESTEIN && pl.ct_k >= 0) {
            // small reduced grids: the coarse classes 256 / 512 / 1024 share one launch (fft_kernel.hip: fft_rows_coarse_kernel);
            // ATLAS_AMD_FFT_COARSE_FUSED=0: one launch per class as before (A/B)
            // (read per Trans object, not once per process: the bitwise tests build one object per setting -- ADVICE r4)
            const bool fuse = !(atlas_amd::env_get("ATLAS_AMD_FFT_COARSE_FUSED") && atoi(atlas_amd::env_get("ATLAS_AMD_FFT_COARSE_FUSED")) == 0);
            if (fft_coarse_ && fuse && pl.ct_f == 1 && pl.shape.M <= 1024 && pl.shape.M == fft::coarse_bluestein_length(2 * pl.h - 1)) {
                by_class[{6, 1024}].push_back(j);
                continue;
            }
            by_class[{1, pl.shape.M}].push_back(j);
            continue;
        }
        if (pl.method == fft::FFT_DIRECT && pl.ct_k >= 0) {
            by_class[{2, pl.shape.M}].push_back(j);  // specialised direct rows
            continue;
        }
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (pl.method == fft::FFT_NATIVE) {
            // native mixed-radix rows: one kernel for every shape.  Two register classes (first-stage radix 3 .. 15: four
            // workgroups per CU / prime 17 .. 31: three); launches bucketed by LDS footprint: 40 KiB (four per CU), 52 KiB (three),
            // 80 KiB (two).  ATLAS_AMD_FFT_NATIVE_FPJ=2: two fields per workgroup where two work arrays fit 80 KiB -- measured
            // slower (2.20 against 1.60 ms for the native rows of O1280, profiles/r04_fft_native.txt), bit-identical
            const int nat_fpj = atlas_amd::env_get("ATLAS_AMD_FFT_NATIVE_FPJ") ? atoi(atlas_amd::env_get("ATLAS_AMD_FFT_NATIVE_FPJ")) : 1;
            const int bigp  = pl.nat.radix[0] > 15 ? 1 : 0;
            const int one   = native_lds_elems(pl, row_mmax[j]);
            const int fpj   = (nat_fpj >= 2 && 2 * one + 64 <= 5120) ? 2 : 1;
            const int elems = fpj * one + 64;
            int cls         = 0;
            for (int c : {2560, 3328, 5120}) {
                if (elems <= c && !((bigp || fpj == 2) && c == 2560)) {   // (three workgroups per CU by registers for those)
                    cls = c;
                    break;
                }
            }
            if (cls == 0) {
                throw std::runtime_error("row length " + std::to_string(pl.n) + " does not fit in LDS (160 KiB)");
            }
            by_class[{10 + 2 * bigp + (fpj - 1), cls}].push_back(j);
            continue;
        }
#endif
        if (pl.method == fft::FFT_HYBRID) {
            // dense-stage rows, bucketed by LDS footprint (workgroups per CU: 8, 6, 4, 3, 2, 1)
            const int fp = pl.lds_complex;
            int cls      = 0;
            for (int c : {1280, 1700, 2560, 3400, 5120, 10240}) {
                if (fp <= c) {
                    cls = c;
                    break;
                }
            }
            if (cls == 0) {
                throw std::runtime_error("row length " + std::to_string(pl.n) + " does not fit in LDS (160 KiB)");
            }
            by_class[{3, cls}].push_back(j);
            continue;
        }
        int cls = -1;
        for (int c : class_M) {
            if (pl.lds_complex <= fft::padded_size(c)) {
                cls = c;
                break;
            }
        }
        if (cls < 0) {
            throw std::runtime_error("row length " + std::to_string(pl.n) + " does not fit in LDS (160 KiB)");
        }
        by_class[{0, cls}].push_back(j);
    }
    // [r6] rows beyond the LDS: per row length the table of the c2r sum (FFT.h:22-82: out[k] = X_0 + sum_{m >= 1} 2 (Re X_m cos(2 pi m k / n)
    // - Im X_m sin(2 pi m k / n)), the Nyquist wavenumber with its real part only), unit roots in extended precision, the angle reduced
    // in integers: table[2 m][k], table[2 m + 1][k] = the factors of Re X_m and Im X_m
    for (const auto& kv : gemm_by_n) {
        const int n = kv.first;
        GemmRows gr;
        gr.n     = n;
        gr.nrows = (int)kv.second.size();
        std::vector<int> sel, mm;
        std::vector<long long> out;
        std::vector<double> scl;
        gr.d_rows = dev_upload(kv.second.data(), kv.second.size());
        for (int j : kv.second) {
            sel.push_back((int)sel.size());
            out.push_back((long long)(geo_.rowoff[j] - geo_.rowoff[band_begin()]));
            scl.push_back(coslatinv[j]);
            mm.push_back(row_mmax[j]);
        }
        std::vector<double> rc(n), rs(n);
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (int t = 0; t < n; ++t) {
            const long double a = two_pi * (long double)t / (long double)n;
            rc[t] = (double)cosl(a);
            rs[t] = (double)sinl(a);
        }
        std::vector<double> table((size_t)2 * (geo_.T + 1) * n, 0.);
        for (int m = 0; m <= geo_.T && 2 * m <= n; ++m) {
            const bool edge = m == 0 || 2 * m == n;   // mean and Nyquist wavenumber: real part only, counted once
            double* tc = table.data() + (size_t)(2 * m) * n;
            double* ts = tc + n;
            for (int k = 0; k < n; ++k) {
                const int t = (int)(((long long)m * k) % n);
                tc[k] = edge ? rc[t] : 2. * rc[t];
                ts[k] = edge ? 0. : -2. * rs[t];
            }
        }
        gr.d_rowsel   = dev_upload(sel.data(), sel.size());
        gr.d_rowout   = dev_upload(out.data(), out.size());
        gr.d_rowscale = dev_upload(scl.data(), scl.size());
        gr.d_rowmmax  = dev_upload(mm.data(), mm.size());
        gr.d_table    = dev_upload(table.data(), table.size());
        gemm_rows_.push_back(gr);
    }
    // big classes first (longest blocks start first)
    for (auto it = by_class.rbegin(); it != by_class.rend(); ++it) {
        SizeClass c;
        const int M = it->first.second;
        c.lds_bytes = fft::padded_size(M) * 16;
        int ntdiv   = 16;
        if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_NT_DIV")) {
            ntdiv = std::max(1, atoi(e));
        }
        c.nthreads  = std::min(512, std::max(64, (M / ntdiv + 63) / 64 * 64));
        if (it->first.first == 2 && M >= 256 && !atlas_amd::env_get("ATLAS_AMD_FFT_NT_DIV")) {
            // specialised direct rows [16, 16, R0]: the load phase has M / R0 = 256 butterflies, give it one sweep
            c.nthreads = std::min(512, std::max(256, c.nthreads));
        }
        c.nrows     = (int)it->second.size();
        c.ct_f = c.ct_k = -1;
        c.direct = it->first.first == 2;
        c.hybrid = it->first.first == 3;
        c.coarse_fused = it->first.first == 6;
        c.native = it->first.first >= 10 && it->first.first <= 13;
        c.native_bigp = c.native && ((it->first.first - 10) & 2) != 0;
        c.native_fpj  = c.native ? ((it->first.first - 10) & 1) + 1 : 1;
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (c.native) {
            int lds = 0;
            for (int j : it->second) {
                lds = std::max(lds, c.native_fpj * native_lds_elems(fftplans_.plans[row_plan[j]], row_mmax[j]) + 64);
            }
            c.lds_bytes = lds * 16;
            c.nthreads  = fft::NAT_NT;
        }
#endif
        if (c.hybrid) {
            int lds = 0, nthr = 64;
            for (int j : it->second) {
                const fft::FftRowPlan& pl = fftplans_.plans[row_plan[j]];
                lds                       = std::max(lds, pl.lds_complex);
                // one radix-8 butterfly per worker and stage; enough wavefronts for the dense stage's row tiles
                nthr = std::max(nthr, std::min(512, (pl.h / 8 + 63) / 64 * 64));
                nthr = std::max(nthr, 64 * ((pl.hyb_Mt + fft::HYB_UPW - 1) / fft::HYB_UPW));
            }
            if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_HYB_NT")) {
                nthr = std::max(nthr, atoi(e));
            }
            c.lds_bytes = lds * 16;
            c.nthreads  = nthr;
        }
        if (it->first.first >= 1 && it->first.first <= 3) {
            const fft::FftRowPlan& pl = fftplans_.plans[row_plan[it->second[0]]];
            c.ct_f                    = pl.ct_f;
            c.ct_k                    = pl.ct_k;
        }
        // within a class: longest rows first
        std::sort(it->second.begin(), it->second.end(), [&](int a, int b) {
            const int na = fftplans_.plans[row_plan[a]].n, nb = fftplans_.plans[row_plan[b]].n;
            return na > nb || (na == nb && a < b);
        });
        c.d_rows = dev_upload(it->second.data(), it->second.size());
        if (it->first.first == 0) {   // run-time shaped rows: direct and odd-length ones apart (same order), for the fp32 variant's two-field form
            std::vector<int> direct_rows, rest;
            for (int j : it->second) {
                const int method = fftplans_.plans[row_plan[j]].method;
                (method == fft::FFT_DIRECT || method == fft::FFT_ODD ? direct_rows : rest).push_back(j);
            }
            if (!direct_rows.empty()) {
                c.nrows_pair  = (int)direct_rows.size();
                c.d_rows_pair = dev_upload(direct_rows.data(), direct_rows.size());
                c.nrows_rest  = (int)rest.size();
                c.d_rows_rest = rest.empty() ? nullptr : dev_upload(rest.data(), rest.size());
            }
        }
        c.coarse_n[0] = c.coarse_n[1] = c.coarse_n[2] = 0;
        if (c.coarse_fused) {   // rows per Bluestein length 1024 / 512 / 256: contiguous in the list (sorted by descending row length)
            int last = 1024;
            for (int j : it->second) {
                const int M = fft::coarse_bluestein_length(2 * fftplans_.plans[row_plan[j]].h - 1);
                if (M > last || (M != 1024 && M != 512 && M != 256)) {
                    throw std::logic_error("coarse Fourier classes: row list not sorted by Bluestein length");
                }
                last = M;
                ++c.coarse_n[M == 1024 ? 0 : (M == 512 ? 1 : 2)];
            }
        }
        if (it->first.first == 1 || it->first.first == 6) {   // specialised Bluestein rows: one flat record per row (device_structs.h: FftRowDesc)
            std::vector<FftRowDesc> desc(it->second.size());
            for (size_t i = 0; i < desc.size(); ++i) {
                const int j               = it->second[i];
                const fft::FftRowPlan& pl = fftplans_.plans[row_plan[j]];
                FftRowDesc& d             = desc[i];
                d.row        = j;
                d.mmax       = std::min(row_mmax[j], pl.h);
                c.max_mmax   = std::max(c.max_mmax, d.mmax);
                d.h          = pl.h;
                d.n          = pl.n;
                d.goff_rel   = (long long)(geo_.rowoff[j] - geo_.rowoff[band_begin()]);
                d.coslatinv  = coslatinv[j];
                d.off_tw     = pl.off_tw;
                d.off_pre    = pl.off_pre;
                d.off_chirp  = pl.off_chirp;
                d.off_bhat_t = pl.off_bhat_t;
            }
            c.d_desc = dev_upload(desc.data(), desc.size());
        }
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (c.native) {   // one 128-byte record per row (device_structs.h: FftNatDesc)
            std::vector<FftNatDesc> desc(it->second.size());
            for (size_t i = 0; i < desc.size(); ++i) {
                const int j               = it->second[i];
                const fft::FftRowPlan& pl = fftplans_.plans[row_plan[j]];
                FftNatDesc& d             = desc[i];
                std::memset(&d, 0, sizeof(d));
                d.row       = j;
                d.mmax      = std::min(row_mmax[j], pl.h);
                d.h         = pl.h;
                d.n         = pl.n;
                d.goff_rel  = (long long)(geo_.rowoff[j] - geo_.rowoff[band_begin()]);
                d.coslatinv = coslatinv[j];
                d.off_tw    = pl.off_tw;
                d.off_pre   = pl.off_pre;
                d.perm      = pl.nat.perm;
                d.ns        = pl.nat.ns;
                for (int e = 0; e < fft::NAT_MAX_STAGES; ++e) {
                    d.radix[e]  = e < pl.nat.ns ? pl.nat.radix[e] : 0;
                    d.nb[e]     = e < pl.nat.ns ? pl.nat.nb[e] : 0;
                    d.stride[e] = e < pl.nat.ns ? pl.nat.stride[e] : 0;
                    d.tab[e]    = e < pl.nat.ns ? pl.nat.tab[e] : 0;
                }
                d.lds_elems = pl.nat.lds_elems;
            }
            c.d_desc = dev_upload(desc.data(), desc.size());
        }
#endif
        classes_.push_back(c);
    }
}

int Trans::fft_row_kernel(const fft::FftRowPlan& pl) const {
    if (pl.method == fft::FFT_HYBRID) {
        return 3;
    }
    if (pl.method == fft::FFT_NATIVE) {
        return 4;
    }
    if (use_ct_ && pl.ct_k >= 0 && pl.method == fft::FFT_BLUESTEIN) {
        return 1;
    }
    if (use_ct_ && pl.ct_k >= 0 && pl.method == fft::FFT_DIRECT) {
        return 2;
    }
    return 0;
}

int Trans::fourier_row_pitch(int nb_fields) const {
    return (2 * nb_fields + 15) / 16 * 16;
}

size_t Trans::fourier_doubles(int nb_fields) const {
    const int rows = cfg_.by_band ? band_end() - band_begin() : geo_.nlats;
    return size_t(rows) * size_t(m_cnt_) * size_t(fourier_row_pitch(nb_fields));
}

double* Trans::fourier_buffer(int nb_fields) {
    const size_t need = fourier_doubles(nb_fields);
    if (need > fourier_cap_) {
        synchronize();
        if (d_fourier_) {
            HIP_CHECK(hipFree(d_fourier_));
            d_fourier_ = nullptr;
        }
        HIP_CHECK(hipMalloc((void**)&d_fourier_, need * sizeof(double)));
        fourier_cap_ = need;
    }
    return d_fourier_;
}

static constexpr size_t kMaxPendingTimings = 1024;   // (begin, end) event pairs held before they are folded into timings_

void Trans::timed_begin(int kind, hipStream_t s) {
    if (!profile_) {
        return;
    }
    ev_stream_ = s ? s : stream_;
    if (ev_used_ >= 2 * kMaxPendingTimings) {
        collect_timings();   // a caller that never asks for timings() keeps a bounded number of events
    }
    while (events_.size() < ev_used_ + 2) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        events_.push_back(e);
    }
    ev_kind_.push_back(kind);
    HIP_CHECK(hipEventRecord(events_[ev_used_], ev_stream_));
}

void Trans::timed_end() {
    if (!profile_) {
        return;
    }
    HIP_CHECK(hipEventRecord(events_[ev_used_ + 1], ev_stream_));
    ev_used_ += 2;
}

void Trans::legendre_device(int trc_in, int nb_fields, const double* sp_dev, double* fourier_dev) {
    legendre_chunks(trc_in, nb_fields, sp_dev, fourier_dev, 0, 0);
}

long long Trans::spectral_shard_offsets(std::vector<long long>& moff) const {
    const int T = geo_.T, P = cfg_.by_band ? 1 : cfg_.nparts, part = cfg_.by_band ? 0 : cfg_.part;
    moff.assign(T + 2, -1);
    long long off = 0;
    for (int m = part; m <= T; m += P) {
        moff[m] = off;
        off += 2ll * (T + 1 - m);   // (n = m .. T) x (re, im), per field
    }
    return off;
}

void Trans::legendre_device_sharded(int nb_fields, const double* sp_shard_dev, double* fourier_dev) {
    if (!d_sp_moff_) {
        std::vector<long long> moff;
        spectral_shard_offsets(moff);
        for (long long& v : moff) {
            v = v < 0 ? 0 : v;   // never read for wavenumbers this object does not own
        }
        d_sp_moff_ = dev_upload(moff.data(), moff.size());
    }
    legendre_chunks(geo_.T, nb_fields, sp_shard_dev, fourier_dev, 0, 0, true);
}

void Trans::legendre_chunks(int trc_in, int nb_fields, const double* sp_dev, double* fourier_dev, int chunk0,
                            int nrun, bool sharded_input) {
    if (nb_fields <= 0) {
        return;
    }
    if (trc_in != geo_.T && trc_in != geo_.T + 1) {
        throw std::invalid_argument("legendre_device: input truncation must be T or T+1");
    }
    LegendreParams p;
    p.P      = d_P_;
    p.sp     = sp_dev;
    p.sp_moff = sharded_input ? d_sp_moff_ : nullptr;
    p.F      = fourier_dev;
    p.items  = (const LegendreItemDev*)d_items_;
    p.items2 = (const LegendreItemDev*)d_items2_;
    p.nitems2 = nitems2_;
    p.nlat0  = d_nlat0_;
    p.zero   = d_zero_;
    // dynamic unit assignment of the persistent kernels: one launch at a time per object (the chunked pipeline overlaps launches of two streams)
    p.sched  = (chunk0 == 0 && nrun <= 0) ? d_leg_sched_ : nullptr;
    p.col0   = 0;
    p.T      = geo_.T;
    p.trc_in = trc_in;
    p.nf     = nb_fields;
    p.RP     = fourier_row_pitch(nb_fields);
    p.nlats  = geo_.nlats;
    p.m_div  = cfg_.by_band ? 1 : cfg_.nparts;
    p.m_cnt  = m_cnt_;
    p.row_begin = cfg_.by_band ? band_begin() : 0;
    p.row_end   = cfg_.by_band ? band_end() : geo_.nlats;
    TraceRange trace("Inverse Legendre Transform (GEMM)");   // TransLocal.cc:948
    timed_begin(0);
    if (!work_.items.empty()) {
        HIP_CHECK(launch_legendre(p, (int)work_.items.size(), chunk0, nrun, stream_));
    }
    timed_end();
}

void Trans::fourier_device(int nb_fields, int nb_vordiv, const double* const* part_base, const int* part_cnt,
                           double* gp_dev) {
    fourier_fields(nb_fields, nb_vordiv, part_base, part_cnt, gp_dev, 0, nb_fields, stream_);
}

void Trans::fourier_device_packed(int nb_fields, int nb_vordiv, const double* const* part_base,
                                  const long long* const* part_rowoff_dev, int cols, double* gp_dev,
                                  const long long* rowbase_dev) {
    if (!part_rowoff_dev || (cols != 2 * nb_fields && cols != fourier_row_pitch(nb_fields))) {
        throw std::invalid_argument("fourier_device_packed: row offsets / cols == 2 * nb_fields or the intermediate's pitch");
    }
    fourier_fields(nb_fields, nb_vordiv, part_base, nullptr, gp_dev, 0, nb_fields, stream_, false, part_rowoff_dev, cols, rowbase_dev);
}

// device copy of a piece table: the callers alternate between a few buffer sets (dist_trans.h slots), so a handful of tables
// is kept by content; one that may still be read by kernels in flight is never overwritten without a device synchronisation
void Trans::clear_fourier_parts_cache() {
    synchronize();
    for (auto& e : parts_cache_) {
        if (e.second) {
            (void)hipFree(e.second);
        }
    }
    parts_cache_.clear();
    parts_evict_ = 0;
}

const FourierParts* Trans::device_parts(const FourierParts& hp) {
    const unsigned char* bytes = reinterpret_cast<const unsigned char*>(&hp);
    for (auto& e : parts_cache_) {
        if (std::memcmp(e.first.data(), bytes, sizeof(hp)) == 0) {
            return static_cast<const FourierParts*>(e.second);
        }
    }
    void* d = nullptr;
    if (parts_cache_.size() < 8) {
        HIP_CHECK(hipMalloc(&d, sizeof(FourierParts)));
        parts_cache_.emplace_back(std::vector<unsigned char>(bytes, bytes + sizeof(hp)), d);
    }
    else {
        HIP_CHECK(hipDeviceSynchronize());
        auto& e = parts_cache_[parts_evict_++ % parts_cache_.size()];
        e.first.assign(bytes, bytes + sizeof(hp));
        d = e.second;
    }
    HIP_CHECK(hipMemcpy(d, &hp, sizeof(hp), hipMemcpyHostToDevice));
    return static_cast<const FourierParts*>(d);
}

void Trans::fourier_fields(int nb_fields, int nb_vordiv, const double* const* part_base, const int* part_cnt,
                           double* gp_dev, int f_begin, int f_end, hipStream_t stream, bool f32,
                           const long long* const* part_rowoff_dev, int packed_cols, const long long* packed_rowbase_dev) {
    if (nb_fields <= 0 || f_end <= f_begin) {
        return;
    }
    FourierParams p;
    FourierParts hp;
    std::memset(&hp, 0, sizeof(hp));   // compared by content: no indeterminate padding
    for (int i = 0; i < fft::MAX_PARTS; ++i) {
        hp.base[i]   = i < fourier_parts() ? part_base[i] : nullptr;
        hp.cnt[i]    = (i < fourier_parts() && part_cnt) ? part_cnt[i] : 0;
        hp.rowoff[i] = (i < fourier_parts() && part_rowoff_dev) ? part_rowoff_dev[i] : nullptr;
    }
    p.part_base0   = hp.base[0];
    p.part_cnt0    = hp.cnt[0];
    p.part_rowoff0 = hp.rowoff[0];
    p.parts        = fourier_parts() > 1 ? device_parts(hp) : nullptr;
    p.packed_cols = part_rowoff_dev ? packed_cols : 0;
    p.packed_rowbase  = part_rowoff_dev ? packed_rowbase_dev : nullptr;
    p.nparts          = fourier_parts();
    p.parts_shift     = -1;
    for (int s = 0; s < 8; ++s) {
        if ((1 << s) == p.nparts) {
            p.parts_shift = s;
        }
    }
    p.lat0            = band_begin();
    p.gp              = gp_dev;
    if (windowed()) {
        // whole rows first (TransLocal.cc:1120-1135 does the same: full-length c2r, then the window is copied out)
        if (f32) {
            throw std::logic_error("fp32 invtrans of a longitude-window crop is not implemented");
        }
        const size_t need = (size_t)nb_fields * (size_t)band_points();
        if (need > gpfull_cap_) {
            HIP_CHECK(hipStreamSynchronize(stream));
            if (d_gpfull_) {
                HIP_CHECK(hipFree(d_gpfull_));
                d_gpfull_ = nullptr;
            }
            HIP_CHECK(hipMalloc((void**)&d_gpfull_, need * sizeof(double)));
            gpfull_cap_ = need;
        }
        p.gp = d_gpfull_;
    }
    p.plans           = (const fft::FftRowPlan*)d_fftplans_;
    p.table           = (const fft::cplx*)d_ffttable_;
    if ((long long)(geo_.T + 1) * fourier_row_pitch(nb_fields) >= (1LL << 31) || fourier_row_pitch(nb_fields) >= (1 << 24) ||
        geo_.T + 1 >= (1 << 24)) {   // fft_device.h: ModeReaderT forms wavenumber x record length as a 24-bit product
        throw std::invalid_argument("invtrans: (truncation + 1) x 2 nb_fields exceeds the 31-bit record offset of the Fourier stage");
    }
    if (f32 && !d_ffttable_f32_) {   // float copy of the tables: the fp32 variant's specialised rows run in fp32 arithmetic
        std::vector<fft::cplxf> tf(fftplans_.table.size());
        for (size_t i = 0; i < tf.size(); ++i) {
            tf[i] = fft::cplxf{(float)fftplans_.table[i].re, (float)fftplans_.table[i].im};
        }
        d_ffttable_f32_ = dev_upload(tf.data(), tf.size());
    }
    p.table_f32       = f32 ? (const fft::cplxf*)d_ffttable_f32_ : nullptr;
    p.nat_table       = d_nat_table_;
    p.ndesc           = nullptr;
    p.row_plan        = d_row_plan_;
    p.row_mmax        = d_row_mmax_;
    p.rowoff          = d_rowoff_;
    p.T               = geo_.T;
    p.RP              = fourier_row_pitch(nb_fields);
    p.nf              = nb_fields;
    p.f32             = f32 ? 1 : 0;
    p.f_begin         = f_begin;
    p.f_end           = f_end;
    p.npts            = geo_.rowoff[band_end()] - geo_.rowoff[band_begin()];
    p.scale_uv_fields = std::min(2 * nb_vordiv, nb_fields);
    p.coslatinv       = d_coslatinv_;
    p.prof            = d_prof_;
    p.jobs            = 1;
    p.pf_dist         = 2;   // 16 jobs of the XCD ahead (profiles/r03_fft_experiments.txt, 12.)
    p.pf_sectors      = 1;
    p.row_affinity    = 1;
    p.mid_rot         = 0;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_MIDROT")) {   // A/B
        p.mid_rot = atoi(e);
    }
    p.job_group_log2  = f32 ? 4 : 3;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_GROUP_LOG2")) {   // A/B
        p.job_group_log2 = std::max(3, std::min(4, atoi(e)));
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_ROW_AFFINITY")) {
        p.row_affinity = atoi(e);
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_PREFETCH")) {   // "distance[,requests per line]"
        p.pf_dist = atoi(e);
        if (const char* c = strchr(e, ',')) {
            p.pf_sectors = std::max(1, std::min(4, atoi(c + 1)));
        }
    }
    p.trace           = d_trace_;
    p.trace_cap       = trace_cap_;
    p.abl             = atlas_amd::env_get("ATLAS_AMD_FFT_ABLATE") ? atoi(atlas_amd::env_get("ATLAS_AMD_FFT_ABLATE")) : 0;
    TraceRange trace(geo_.regular ? "Inverse Fourier Transform (mi355x, RegularGrid)"      // TransLocal.cc:1107
                                  : "Inverse Fourier Transform (mi355x, ReducedGrid)");    // TransLocal.cc:1159
    timed_begin(1, stream);
    const int only_m = atlas_amd::env_get("ATLAS_AMD_FFT_ONLY_M") ? atoi(atlas_amd::env_get("ATLAS_AMD_FFT_ONLY_M")) : 0;
    // The row-length classes are independent launches.  Run back to back on one stream each ends in a tail (the last
    // workgroups of a class leave CUs idle); dealt round robin to a few streams the tail of one class is filled by the
    // next and workgroups of different LDS footprints share a CU.  Measured on TL1279/O1280/137 levels (25 classes, in
    // descending row length): 9.3 ms on one stream, 8.9 / 8.6 / 8.0 / 8.3 / 8.6 ms on 2 / 3 / 4 / 5 / 6-8 streams; 150 random
    // orders and stream assignments found nothing below the round robin over four (profiles/r02_fft_streams.txt).
    // ATLAS_AMD_FFT_STREAMS overrides.  All streams fork from and join the caller's stream through events.
    // Small reduced grids (coarse row classes: three or four launches of tens of microseconds): one stream -- forking and joining
    // side streams costs more than the tails (TL159 -> O160, 60 fields: stage 0.133 / 0.110 / 0.100 ms on 4 / 2 / 1 streams).
    const int nstreams_env = atlas_amd::env_get("ATLAS_AMD_FFT_STREAMS") ? atoi(atlas_amd::env_get("ATLAS_AMD_FFT_STREAMS"))
                                                                         : ((fft_coarse_ && geo_.nxmax <= 3300) ? 1 : 4);   // [r6] one stream up to O800 (tools/probe/coarse_ab.sh)
    const int nstreams = std::max(1, std::min(nstreams_env, 8));
    while ((int)side_streams_.size() < nstreams - 1) {
        hipStream_t st;
        hipEvent_t ev;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        side_streams_.push_back(st);
        side_joins_.push_back(ev);
    }
    if (nstreams > 1 && !side_fork_) {
        HIP_CHECK(hipEventCreateWithFlags(&side_fork_, hipEventDisableTiming));
    }
    std::vector<char> used(nstreams, 0);
    bool forked = false;
    int next    = 0;
    for (const SizeClass& c : classes_) {
        p.rows  = c.d_rows;
        p.nrows = c.nrows;
        p.coarse_n[0] = c.coarse_n[0];
        p.coarse_n[1] = c.coarse_n[1];
        p.coarse_n[2] = c.coarse_n[2];
        p.seq_ok = c.max_mmax + 1 <= 1280 ? 1 : 0;
        p.desc  = c.native ? nullptr : (const FftRowDesc*)c.d_desc;
        p.ndesc = c.native ? (const FftNatDesc*)c.d_desc : nullptr;
        if (only_m && c.lds_bytes != fft::padded_size(only_m) * 16) {  // dev tool (tools/fft_phase_prof.py): one class
            continue;
        }
        const int only_native = atlas_amd::env_get("ATLAS_AMD_FFT_ONLY_NATIVE") ? atoi(atlas_amd::env_get("ATLAS_AMD_FFT_ONLY_NATIVE")) : 0;
        if (only_native && !c.native) {   // dev tool: the native mixed-radix rows alone
            continue;
        }
        const int si = next++ % nstreams;
        hipStream_t st = stream;
        if (si > 0) {
            if (!forked) {
                HIP_CHECK(hipEventRecord(side_fork_, stream));
                forked = true;
            }
            st = side_streams_[si - 1];
            if (!used[si]) {
                HIP_CHECK(hipStreamWaitEvent(st, side_fork_, 0));
                used[si] = 1;
            }
        }
        if (c.native) {
            HIP_CHECK(launch_fourier_nat(p, c.lds_bytes, c.native_bigp ? 1 : 0, c.native_fpj, st));
        }
        else if (c.coarse_fused) {
            HIP_CHECK(launch_fourier_coarse(p, c.lds_bytes, st));
        }
        else if (c.hybrid) {
            HIP_CHECK(launch_fourier_hyb(p, c.lds_bytes, c.nthreads, st));
        }
        else if (c.ct_k >= 0 && use_ct_) {
            if (c.direct) {
                HIP_CHECK(launch_fourier_dct(p, c.ct_f, c.ct_k, c.lds_bytes, c.nthreads, st));
            }
            else {
                HIP_CHECK(launch_fourier_ct(p, c.ct_f, c.ct_k, c.lds_bytes, c.nthreads, st));
            }
        }
        else if (c.nrows_pair > 0 && fourier_generic_pairs_usable(p)) {   // fp32 variant: the direct rows two fields per job, then the rest
            p.rows  = c.d_rows_pair;
            p.nrows = c.nrows_pair;
            HIP_CHECK(launch_fourier_generic_pairs(p, c.lds_bytes, c.nthreads, st));
            if (c.nrows_rest > 0) {
                p.rows  = c.d_rows_rest;
                p.nrows = c.nrows_rest;
                HIP_CHECK(launch_fourier(p, c.lds_bytes, c.nthreads, st));
            }
        }
        else {
            HIP_CHECK(launch_fourier(p, c.lds_bytes, c.nthreads, st));
        }
    }
    for (const GemmRows& gr : gemm_rows_) {   // [r6] rows beyond the LDS: dft_gemm.hip, on the caller's stream beside the classes
        // their kept wavenumbers through the stage's own reader (whatever the layout of the intermediate) into a dense array, then the product
        const int nfs    = p.f_end - p.f_begin;
        const int RPd    = 2 * nfs;
        const size_t need = (size_t)gr.nrows * (size_t)(geo_.T + 1) * (size_t)RPd;
        if (need > gemm_dense_cap_) {
            HIP_CHECK(hipStreamSynchronize(stream));
            if (d_gemm_dense_) {
                HIP_CHECK(hipFree(d_gemm_dense_));
                d_gemm_dense_   = nullptr;
                gemm_dense_cap_ = 0;
            }
            HIP_CHECK(hipMalloc((void**)&d_gemm_dense_, need * sizeof(double)));
            gemm_dense_cap_ = need;
        }
        HIP_CHECK(launch_gather_rows_dense(p, gr.d_rows, gr.nrows, d_gemm_dense_, RPd, stream));
        DftGemmArgs ga{};
        ga.F        = d_gemm_dense_;
        ga.f32      = f32 ? 1 : 0;
        ga.rowsel   = gr.d_rowsel;
        ga.table    = gr.d_table;
        ga.out      = p.gp;
        ga.rowout   = gr.d_rowout;
        ga.fstride  = p.npts;
        ga.rowscale = gr.d_rowscale;
        ga.rowmmax  = gr.d_rowmmax;
        ga.T        = geo_.T;
        ga.m_cnt    = geo_.T + 1;
        ga.RP       = RPd;
        ga.nlon     = gr.n;
        ga.nrows    = gr.nrows;
        ga.f0       = 0;
        ga.nf       = nfs;
        ga.nscaled  = p.scale_uv_fields - p.f_begin;   // fields of the dense array are counted from f_begin
        HIP_CHECK(launch_dft_gemm(ga, stream));
    }
    for (int si = 1; si < nstreams; ++si) {
        if (used[si]) {
            HIP_CHECK(hipEventRecord(side_joins_[si - 1], side_streams_[si - 1]));
            HIP_CHECK(hipStreamWaitEvent(stream, side_joins_[si - 1], 0));
        }
    }
    if (windowed()) {
        HIP_CHECK(launch_window_crop(d_gpfull_, gp_dev, d_rowoff_ + band_begin(), d_win_i0_, d_win_n_, d_win_off_,
                                     band_end() - band_begin(), band_points(), win_npts_, f_begin, f_end, stream));
    }
    timed_end();
}

void Trans::fourier_device(int nb_fields, int nb_vordiv, const double* fourier_dev, double* gp_dev) {
    if (fourier_parts() != 1) {
        throw std::logic_error("single-buffer fourier_device needs nparts == 1 or the latitude-band decomposition");
    }
    const double* base[1] = {fourier_dev};
    int cnt[1]            = {m_cnt_};
    fourier_device(nb_fields, nb_vordiv, base, cnt, gp_dev);
}

void Trans::invtrans_uv_device(int trc_in, int nb_fields, int nb_vordiv, const double* sp_dev, double* gp_dev) {
    if (nb_fields <= 0) {
        return;
    }
    TraceRange trace("invtrans_uv structured");   // TransLocal.cc:1418
    if (fourier_parts() != 1) {
        throw std::logic_error("invtrans_uv_device on a wavenumber-sharded Trans: use legendre_device / exchange / fourier_device");
    }
    double* F = fourier_buffer(nb_fields);
    int rtw, nrg, nchunks;
    legendre_tiling(nb_fields, rtw, nrg, nchunks);
    const int cols_per_chunk = 16 * rtw * nrg;  // interleaved (re, im) columns of one chunk
    if (pipeline_ <= 1 || nchunks < 2 || cols_per_chunk % 16 != 0) {
        legendre_device(trc_in, nb_fields, sp_dev, F);
        fourier_device(nb_fields, nb_vordiv, F, gp_dev);
        return;
    }
    // Software pipeline over column chunks: the Fourier stage of chunk c runs on a second stream while the Legendre
    // stage of chunk c+1 occupies the matrix pipes (the two stages stress different units: MFMA + LDS staging vs
    // VALU + LDS exchange + L2 gathers).  Chunks are whole groups of 8 fields, so both stages see aligned tiles.
    if (!stream2_) {
        HIP_CHECK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
    }
    while (pipe_events_.size() < (size_t)nchunks + 1) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pipe_events_.push_back(e);
    }
    const double* base[1] = {F};
    const int cnt[1]      = {m_cnt_};
    const int pieces      = std::min(pipeline_, nchunks);
    int c0                = 0;
    for (int i = 0; i < pieces; ++i) {
        const int c1 = (int)((long long)nchunks * (i + 1) / pieces);
        legendre_chunks(trc_in, nb_fields, sp_dev, F, c0, c1 - c0);
        HIP_CHECK(hipEventRecord(pipe_events_[i], stream_));
        HIP_CHECK(hipStreamWaitEvent(stream2_, pipe_events_[i], 0));
        const int f0 = c0 * cols_per_chunk / 2;
        const int f1 = std::min(nb_fields, c1 * cols_per_chunk / 2);
        fourier_fields(nb_fields, nb_vordiv, base, cnt, gp_dev, f0, f1, stream2_);
        c0 = c1;
    }
    HIP_CHECK(hipEventRecord(pipe_events_[pieces], stream2_));
    HIP_CHECK(hipStreamWaitEvent(stream_, pipe_events_[pieces], 0));  // the call completes on stream()
}

// ---- fp32 variant (BASELINE config C5): fp32 spectra / table / intermediate / grid points, v_mfma_f32_16x16x4_f32 in the
// Legendre stage; the Fourier stage converts on load and store and keeps its arithmetic in fp64.
void Trans::invtrans_scalar_device_f32(int nb_fields, const float* sp_dev, float* gp_dev) {
    invtrans_uv_device_f32(geo_.T, nb_fields, 0, sp_dev, gp_dev);
}

// the vor/div call of the fp32 variant (an extension, like the variant itself): spectra_prepare with float storage (double
// arithmetic), Legendre + Fourier stage at truncation T + 1, 1 / cos(lat) in the Fourier store epilogue -- TransLocal.cc:1523-1597
void Trans::invtrans_device_f32(int nb_scalar, const float* sp_dev, int nb_vordiv, const float* vor_dev, const float* div_dev,
                                float* gp_dev) {
    if (nb_vordiv <= 0) {
        invtrans_uv_device_f32(geo_.T, std::max(nb_scalar, 0), 0, sp_dev, gp_dev);
        return;
    }
    nb_scalar         = std::max(nb_scalar, 0);
    const int T       = geo_.T;
    const int nall    = 2 * nb_vordiv + nb_scalar;
    const size_t nout = size_t(T + 2) * size_t(T + 3) * size_t(nall);
    ensure(d_all_, all_cap_, (nout + 1) / 2);   // (the double buffer of the fp64 path, used as floats)
    timed_begin(2);
    HIP_CHECK(launch_spectra_prepare_f32(vor_dev, div_dev, sp_dev, reinterpret_cast<float*>(d_all_), T, nb_vordiv, nb_scalar, stream_));
    timed_end();
    invtrans_uv_device_f32(T + 1, nall, nb_vordiv, reinterpret_cast<const float*>(d_all_), gp_dev);
}

void Trans::invtrans_uv_device_f32(int trc_in, int nb_fields, int nb_vordiv, const float* sp_dev, float* gp_dev) {
    if (nb_fields <= 0) {
        return;
    }
    if (fourier_parts() != 1) {
        throw std::logic_error("fp32 invtrans on a wavenumber-sharded Trans is not implemented");
    }
    if (!d_P32_) {  // the table, once, in float
        const size_t n = (size_t)work_.table_doubles;
        HIP_CHECK(hipMalloc((void**)&d_P32_, std::max<size_t>(n, 1) * sizeof(float)));
        HIP_CHECK(launch_convert_f64_f32(d_P_, d_P32_, n, stream_));
        HIP_CHECK(hipMalloc((void**)&d_zero32_, 64));
        HIP_CHECK(hipMemsetAsync(d_zero32_, 0, 64, stream_));
    }
    const size_t need = fourier_doubles(nb_fields);  // same element count, float storage
    if (need > fourier32_cap_) {
        synchronize();
        if (d_fourier32_) {
            HIP_CHECK(hipFree(d_fourier32_));
            d_fourier32_ = nullptr;
        }
        HIP_CHECK(hipMalloc((void**)&d_fourier32_, need * sizeof(float)));
        fourier32_cap_ = need;
    }
    LegendreParamsF32 p;
    p.P         = d_P32_;
    p.sp        = sp_dev;
    p.F         = d_fourier32_;
    p.items     = (const LegendreItemDev*)d_items_;
    p.items2    = (const LegendreItemDev*)d_items2_;
    p.nitems2   = nitems2_;
    p.sp_moff   = nullptr;
    p.nlat0     = d_nlat0_;
    p.zero      = d_zero32_;
    p.sched     = d_leg_sched_;
    p.col0      = 0;
    p.T         = geo_.T;
    p.trc_in    = trc_in;
    p.nf        = nb_fields;
    p.RP        = fourier_row_pitch(nb_fields);
    p.nlats     = geo_.nlats;
    p.m_div     = 1;
    p.m_cnt     = m_cnt_;
    p.row_begin = cfg_.by_band ? band_begin() : 0;
    p.row_end   = cfg_.by_band ? band_end() : geo_.nlats;
    timed_begin(0);
    if (!work_.items.empty()) {
        HIP_CHECK(launch_legendre_f32(p, (int)work_.items.size(), 0, 0, stream_));
    }
    timed_end();
    const double* base[1] = {reinterpret_cast<const double*>(d_fourier32_)};
    const int cnt[1]      = {m_cnt_};
    fourier_fields(nb_fields, nb_vordiv, base, cnt, reinterpret_cast<double*>(gp_dev), 0, nb_fields, stream_, true);
}

void Trans::invtrans_scalar_f32(int nb_fields, const float scalar_spectra[], float gp_fields[]) {
    if (nb_fields <= 0) {
        return;
    }
    const size_t nsp = nb_spectral_coefficients() * (size_t)nb_fields;
    const size_t ngp = (size_t)nb_gridpoints() * (size_t)nb_fields;
    // staging buffers shared with the fp64 host API (sized in doubles: twice what the floats need)
    ensure(d_sp_, sp_cap_, (nsp + 1) / 2);
    ensure(d_gp_, gp_cap_, (ngp + 1) / 2);
    HIP_CHECK(hipMemcpyAsync(d_sp_, scalar_spectra, nsp * sizeof(float), hipMemcpyHostToDevice, stream_));
    invtrans_scalar_device_f32(nb_fields, reinterpret_cast<const float*>(d_sp_), reinterpret_cast<float*>(d_gp_));
    HIP_CHECK(hipMemcpyAsync(gp_fields, d_gp_, ngp * sizeof(float), hipMemcpyDeviceToHost, stream_));
    synchronize();
}

void Trans::collect_timings() {
    if (ev_used_ == 0) {
        return;
    }
    synchronize();
    for (size_t i = 0; i + 1 < ev_used_; i += 2) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, events_[i], events_[i + 1]));
        if (ev_kind_[i / 2] == 0) {
            timings_.legendre_ms += ms;
            timings_.legendre_calls++;
        }
        else if (ev_kind_[i / 2] == 2) {
            timings_.prepare_ms += ms;
            timings_.prepare_calls++;
        }
        else {
            timings_.fourier_ms += ms;
            timings_.fourier_calls++;
        }
    }
    ev_used_ = 0;
    ev_kind_.clear();
}

StageTimings Trans::timings() {
    collect_timings();
    return timings_;
}

void Trans::invtrans(int nb_scalar_fields, const double scalar_spectra[], double gp_fields[]) {
    // TransLocal.cc:931-934
    if (nb_scalar_fields <= 0) {
        return;
    }
    if (cfg_.nparts != 1) {
        throw std::logic_error("host invtrans needs nparts == 1");
    }
    const size_t nsp = nb_spectral_coefficients() * (size_t)nb_scalar_fields;
    const size_t ngp = (size_t)nb_gridpoints() * (size_t)nb_scalar_fields;  // all points, or the rows of a crop
    // large transfers go through the field-chunked full-duplex pipeline below (ATLAS_AMD_HOST_PIPELINE=0: one upload, one
    // transform, one download, strictly serial -- 178 ms at TL1279 / O1280 / 137 levels; profiles/r05_bench_host.txt).  Read per
    // call: round 4's bench toggled a process-wide static and measured the same path twice.
    bool pipe = true;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_HOST_PIPELINE")) {
        pipe = atoi(e) != 0;
    }
    if (pipe && nb_scalar_fields >= 32 && ngp * sizeof(double) >= (size_t(256) << 20) && fourier_parts() == 1 && !windowed()) {
        if (invtrans_host_pipelined(nb_scalar_fields, scalar_spectra, 0, nullptr, nullptr, gp_fields)) {
            return;
        }   // staging buffers unavailable: the serial path below
    }
    ensure(d_sp_, sp_cap_, nsp);
    ensure(d_gp_, gp_cap_, ngp);
    HIP_CHECK(hipMemcpyAsync(d_sp_, scalar_spectra, nsp * sizeof(double), hipMemcpyHostToDevice, stream_));
    invtrans_uv_device(geo_.T, nb_scalar_fields, 0, d_sp_, d_gp_);
    HIP_CHECK(hipMemcpyAsync(gp_fields, d_gp_, ngp * sizeof(double), hipMemcpyDeviceToHost, stream_));
    synchronize();
}

// Host arrays in, host arrays out (what atlas__Trans__invtrans_scalar callers pass, TransInterface.h:74-79): 1.8 GB up and
// 7.2 GB down per 137 levels at TL1279 / O1280 against 15 ms of compute -- bound by the link.  PCIe is full duplex (97 GB/s for
// both directions at once against 57 for one, tools/probe/probe_host_link.hip): the transform is cut into CHUNKS OF FIELDS
// (fields are independent in both stages) so that the spectra of later chunks go up, and later chunks are transformed, while
// the grid points of earlier chunks come down:
//     this thread:     gathers the chunk's columns of sp into a pinned buffer (OpenMP; 42 GB/s) and enqueues
//     upload stream:   H2D(c+1) from the pinned buffer
//     Trans stream:    Legendre + Fourier stage of chunk c (nf = the chunk's fields: per-field arithmetic does not depend on the
//                      other fields of a call -- bitwise equal to the one-call device path, tests/test_gpu_trans.py)
//     download thread: D2H(c) into a pinned buffer (copy stream, 57 GB/s) while it drains the pinned buffer of chunk c-1 into the
//                      caller's array (OpenMP)
// Exposed beside the 7.2 GB download: gather, upload and transform of the FIRST chunk and the drain of the last.  Two pinned and
// two device buffers per direction; a buffer is reused two chunks later (events for the device side, a counter for the downloads).
// What round 5 measured on the way (profiles/r05_bench_host.txt, r05_host_link_probe.txt): the host copies must run on a BOUNDED
// team -- with the box's 256 hardware threads an OpenMP memcpy reaches 20 GB/s, with 16 - 32 threads 120 - 165 GB/s (first version
// of this pipeline: 376 ms against 179 serial); the runtime's own pageable path (blocking hipMemcpy from a second thread, 50 GB/s)
// does not overlap the two directions at all (53 GB/s for both together against 97 from pinned memory: 180 ms, no gain).
// ATLAS_AMD_HOST_CHUNK=<fields> (multiple of 8; default 16), ATLAS_AMD_HOST_THREADS=<n> (default 8).
// The staging buffers of the pipeline: two pinned and two device buffers per direction, grown together.  A multi-GB hipHostMalloc
// can fail (memlock limit, memory pressure): the new set is allocated into temporaries and committed only when all eight
// allocations succeeded; on failure the temporaries are freed, the old set is dropped with its pointers nulled and capacities
// zeroed (release() and the next call see a consistent "no buffers" state), the sticky allocation error is cleared and the
// caller falls back to the serial path (ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC=1 forces this branch: tests).
bool Trans::ensure_host_pipeline_buffers(size_t up_doubles, size_t down_doubles) {
    if (up_doubles <= hp_up_cap_ && down_doubles <= hp_down_cap_ && hp_up_[0]) {
        return true;
    }
    synchronize();
    HIP_CHECK(hipStreamSynchronize(hp_up_stream_));
    HIP_CHECK(hipStreamSynchronize(copy_stream_));
    const size_t up_cap   = std::max(hp_up_cap_, up_doubles);
    const size_t down_cap = std::max(hp_down_cap_, down_doubles);
    auto drop = [](double* (&pin_up)[2], double* (&pin_down)[2], double* (&dev_sp)[2], double* (&dev_gp)[2]) {
        for (int i = 0; i < 2; ++i) {
            if (pin_up[i]) (void)hipHostFree(pin_up[i]);
            if (pin_down[i]) (void)hipHostFree(pin_down[i]);
            if (dev_sp[i]) (void)hipFree(dev_sp[i]);
            if (dev_gp[i]) (void)hipFree(dev_gp[i]);
            pin_up[i] = pin_down[i] = dev_sp[i] = dev_gp[i] = nullptr;
        }
    };
    // the old set goes first: both sets together may not fit where the new one alone does
    drop(hp_up_, hp_down_, hp_dsp_, hp_dgp_);
    hp_up_cap_ = hp_down_cap_ = 0;
    double *nu[2] = {nullptr, nullptr}, *nd[2] = {nullptr, nullptr}, *ns[2] = {nullptr, nullptr}, *ng[2] = {nullptr, nullptr};
    bool ok = atlas_amd::env_get("ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC") == nullptr;
    for (int i = 0; i < 2 && ok; ++i) {
        ok = hipHostMalloc((void**)&nu[i], up_cap * sizeof(double), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&nd[i], down_cap * sizeof(double), hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void**)&ns[i], up_cap * sizeof(double)) == hipSuccess &&
             hipMalloc((void**)&ng[i], down_cap * sizeof(double)) == hipSuccess;
    }
    if (!ok) {
        drop(nu, nd, ns, ng);
        (void)hipGetLastError();   // the failed allocation's error code is ours: do not leave it sticky for the caller
        return false;
    }
    for (int i = 0; i < 2; ++i) {
        hp_up_[i]   = nu[i];
        hp_down_[i] = nd[i];
        hp_dsp_[i]  = ns[i];
        hp_dgp_[i]  = ng[i];
    }
    hp_up_cap_   = up_cap;
    hp_down_cap_ = down_cap;
    return true;
}

bool Trans::invtrans_host_pipelined(int nb_scalar, const double* sp_host, int nb_vordiv, const double* vor_host,
                                    const double* div_host, double* gp_host) {
    const size_t ncoef = nb_spectral_coefficients();   // doubles per field
    const size_t npts  = (size_t)nb_gridpoints();
    int C = 16;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_HOST_CHUNK")) {
        C = std::max(8, atoi(e) / 8 * 8);
    }
    // the chunks: groups of C / 2 vor/div pairs (C output fields: u then v), then groups of C scalar fields; output field order
    // of the call: [u 0 .. nvd) [v 0 .. nvd) [scalars] (TransLocal.cc:1555-1581)
    struct Job {
        bool vd;
        int f0, n;
    };
    std::vector<Job> jobs;
    for (int f0 = 0; f0 < nb_vordiv; f0 += C / 2) {
        jobs.push_back(Job{true, f0, std::min(C / 2, nb_vordiv - f0)});
    }
    for (int f0 = 0; f0 < nb_scalar; f0 += C) {
        jobs.push_back(Job{false, f0, std::min(C, nb_scalar - f0)});
    }
    const int nchunks = (int)jobs.size();
    if (!hp_up_stream_) {
        HIP_CHECK(hipStreamCreateWithFlags(&hp_up_stream_, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_CHECK(hipEventCreateWithFlags(&hp_up_done_[i], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&hp_comp_done_[i], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&hp_down_done_[i], hipEventDisableTiming));
        }
    }
    if (!ensure_host_pipeline_buffers((size_t)C * ncoef, (size_t)C * npts)) {
        return false;   // nothing was enqueued: the caller takes the serial pageable path
    }
    // buffers the stages grow on demand (with a synchronisation): before the pipeline starts, not inside it
    (void)fourier_buffer(C);
    if (nb_vordiv > 0) {
        ensure(d_all_, all_cap_, size_t(geo_.T + 2) * size_t(geo_.T + 3) * size_t(C));
    }
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    // ---- the download thread: chunk c leaves the device as soon as its transform has finished
    std::mutex mtx;
    std::condition_variable cv;
    int enqueued = 0, downloaded = 0;   // chunks whose transform is enqueued (event recorded) / whose grid points have left the device
    bool abort = false;
    std::string thread_error;
    std::thread down([&]() {
        try {
            HIP_CHECK(hipSetDevice(device));
            auto drain = [&](int c) {   // pinned -> the caller's array, once the chunk's download has finished
                const Job& j = jobs[c];
                HIP_CHECK(hipEventSynchronize(hp_down_done_[c & 1]));
                {
                    std::lock_guard<std::mutex> lk(mtx);
                    downloaded = c + 1;      // the device buffer of chunk c is free
                }
                cv.notify_all();
                const size_t blk = (size_t)j.n * npts;
                if (j.vd) {
                    bounded_copy(gp_host + (size_t)j.f0 * npts, hp_down_[c & 1], blk * sizeof(double));
                    bounded_copy(gp_host + (size_t)(nb_vordiv + j.f0) * npts, hp_down_[c & 1] + blk, blk * sizeof(double));
                }
                else {
                    bounded_copy(gp_host + (size_t)(2 * nb_vordiv + j.f0) * npts, hp_down_[c & 1], blk * sizeof(double));
                }
            };
            for (int c = 0; c < nchunks; ++c) {
                {
                    std::unique_lock<std::mutex> lk(mtx);
                    cv.wait(lk, [&] { return enqueued > c || abort; });
                    if (abort) {
                        return;
                    }
                }
                const size_t out = (size_t)(jobs[c].vd ? 2 : 1) * jobs[c].n * npts;
                // pinned buffer c & 1 was drained (chunk c - 2) in the previous iteration
                HIP_CHECK(hipStreamWaitEvent(copy_stream_, hp_comp_done_[c & 1], 0));
                HIP_CHECK(hipMemcpyAsync(hp_down_[c & 1], hp_dgp_[c & 1], out * sizeof(double), hipMemcpyDeviceToHost, copy_stream_));
                HIP_CHECK(hipEventRecord(hp_down_done_[c & 1], copy_stream_));
                if (c >= 1) {
                    drain(c - 1);            // beside the download of chunk c
                }
            }
            drain(nchunks - 1);
        }
        catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(mtx);
            thread_error = e.what();
            abort        = true;
            cv.notify_all();
        }
    });
    try {
        for (int c = 0; c < nchunks; ++c) {
            const int b  = c & 1;
            const Job& j = jobs[c];
            if (c >= 2) {
                HIP_CHECK(hipEventSynchronize(hp_up_done_[b]));       // the upload out of this pinned buffer (chunk c-2) has finished
            }
            size_t up = (size_t)j.n * ncoef;
            if (j.vd) {
                gather_field_columns(hp_up_[b], vor_host, ncoef, nb_vordiv, j.f0, j.n);
                gather_field_columns(hp_up_[b] + up, div_host, ncoef, nb_vordiv, j.f0, j.n);
                up *= 2;
            }
            else {
                gather_field_columns(hp_up_[b], sp_host, ncoef, nb_scalar, j.f0, j.n);
            }
            if (c >= 2) {
                HIP_CHECK(hipStreamWaitEvent(hp_up_stream_, hp_comp_done_[b], 0));   // chunk c-2 no longer reads this device buffer
            }
            HIP_CHECK(hipMemcpyAsync(hp_dsp_[b], hp_up_[b], up * sizeof(double), hipMemcpyHostToDevice, hp_up_stream_));
            HIP_CHECK(hipEventRecord(hp_up_done_[b], hp_up_stream_));
            if (c >= 2) {   // the grid points of chunk c-2 must have left device buffer b (the event below is re-recorded, too)
                std::unique_lock<std::mutex> lk(mtx);
                cv.wait(lk, [&] { return downloaded > c - 2 || abort; });
                if (abort) {
                    break;
                }
            }
            HIP_CHECK(hipStreamWaitEvent(stream_, hp_up_done_[b], 0));
            if (j.vd) {
                invtrans_device(0, nullptr, j.n, hp_dsp_[b], hp_dsp_[b] + (size_t)j.n * ncoef, hp_dgp_[b]);
            }
            else {
                // inside a vor/div call the scalars are transformed at truncation T + 1 like the wind fields (TransLocal.cc:1590): the
                // m = T wavenumber, dropped by the scalar-only call, survives -- keep that by going through the same entry point
                if (nb_vordiv > 0) {
                    invtrans_scalars_extended_device(j.n, hp_dsp_[b], hp_dgp_[b]);
                }
                else {
                    invtrans_uv_device(geo_.T, j.n, 0, hp_dsp_[b], hp_dgp_[b]);
                }
            }
            HIP_CHECK(hipEventRecord(hp_comp_done_[b], stream_));
            {
                std::lock_guard<std::mutex> lk(mtx);
                enqueued = c + 1;
            }
            cv.notify_all();
        }
    }
    catch (...) {
        {
            std::lock_guard<std::mutex> lk(mtx);
            abort = true;
        }
        cv.notify_all();
        down.join();
        throw;
    }
    down.join();
    if (!thread_error.empty()) {
        throw std::runtime_error("host pipeline, download thread: " + thread_error);
    }
    synchronize();
    return true;
}

void Trans::enable_phase_profile(bool on) {
    synchronize();
    if (on && !d_prof_) {
        HIP_CHECK(hipMalloc((void**)&d_prof_, 64 * sizeof(unsigned long long)));
    }
    if (on) {
        HIP_CHECK(hipMemset(d_prof_, 0, 64 * sizeof(unsigned long long)));
    }
    else if (d_prof_) {
        HIP_CHECK(hipFree(d_prof_));
        d_prof_ = nullptr;
    }
}

void Trans::fft_trace(unsigned long long words, unsigned long long* out) {
    synchronize();
    if (!out) {   // (re)allocate and zero
        if (d_trace_) {
            HIP_CHECK(hipFree(d_trace_));
            d_trace_ = nullptr;
        }
        trace_cap_ = words;
        if (words) {
            HIP_CHECK(hipMalloc((void**)&d_trace_, words * sizeof(unsigned long long)));
            HIP_CHECK(hipMemset(d_trace_, 0, words * sizeof(unsigned long long)));
        }
        return;
    }
    const unsigned long long n = words < trace_cap_ ? words : trace_cap_;
    if (n) {
        HIP_CHECK(hipMemcpy(out, d_trace_, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
}

void Trans::read_phase_profile(unsigned long long out[64]) {
    synchronize();
    if (!d_prof_) {
        std::memset(out, 0, 64 * sizeof(unsigned long long));
        return;
    }
    HIP_CHECK(hipMemcpy(out, d_prof_, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
}

void Trans::ensure(double*& ptr, size_t& cap, size_t n) {
    if (n > cap) {
        synchronize();
        if (ptr) {
            HIP_CHECK(hipFree(ptr));
            ptr = nullptr;
        }
        HIP_CHECK(hipMalloc((void**)&ptr, n * sizeof(double)));
        cap = n;
    }
}

void Trans::invtrans_device(int nb_scalar, const double* sp_dev, int nb_vordiv, const double* vor_dev,
                            const double* div_dev, double* gp_dev) {
    if (fourier_parts() != 1) {
        throw std::logic_error("invtrans_device on a wavenumber-sharded Trans: use the stage API");
    }
    if (nb_vordiv > 0) {
        const int T       = geo_.T;
        const int nall    = 2 * nb_vordiv + nb_scalar;
        const size_t nout = size_t(T + 2) * size_t(T + 3) * size_t(nall);
        ensure(d_all_, all_cap_, nout);
        timed_begin(2);
        HIP_CHECK(launch_spectra_prepare(vor_dev, div_dev, sp_dev, d_all_, T, nb_vordiv, nb_scalar, stream_));
        timed_end();
        invtrans_uv_device(T + 1, nall, nb_vordiv, d_all_, gp_dev);  // TransLocal.cc:1590
    }
    else if (nb_scalar > 0) {
        invtrans_uv_device(geo_.T, nb_scalar, 0, sp_dev, gp_dev);  // TransLocal.cc:1593-1595
    }
}

// the scalar fields of a vor/div call on their own: zero-extended to truncation T + 1 and transformed there, as inside the combined
// call (TransLocal.cc:1496-1519,1590) -- same bits per field (the host pipeline transforms a call chunk by chunk)
void Trans::invtrans_scalars_extended_device(int nb_scalar, const double* sp_dev, double* gp_dev) {
    const int T = geo_.T;
    ensure(d_all_, all_cap_, size_t(T + 2) * size_t(T + 3) * size_t(nb_scalar));
    timed_begin(2);
    HIP_CHECK(launch_spectra_prepare(nullptr, nullptr, sp_dev, d_all_, T, 0, nb_scalar, stream_));
    timed_end();
    invtrans_uv_device(T + 1, nb_scalar, 0, d_all_, gp_dev);
}

void Trans::invtrans(int nb_scalar, const double sp[], int nb_vordiv, const double vor[], const double div[],
                     double gp[]) {
    if (nb_vordiv <= 0) {
        invtrans(nb_scalar, sp, gp);
        return;
    }
    nb_scalar          = std::max(nb_scalar, 0);
    const size_t ncoef = nb_spectral_coefficients();
    const size_t nvd   = ncoef * (size_t)nb_vordiv;
    const size_t nsp   = ncoef * (size_t)nb_scalar;
    const size_t ngp   = (size_t)nb_gridpoints() * (size_t)(2 * nb_vordiv + nb_scalar);
    {   // large calls: the field-chunked full-duplex pipeline (invtrans_host_pipelined), as for the scalar call
        bool pipe = true;
        if (const char* e = atlas_amd::env_get("ATLAS_AMD_HOST_PIPELINE")) {
            pipe = atoi(e) != 0;
        }
        if (pipe && 2 * nb_vordiv + nb_scalar >= 32 && ngp * sizeof(double) >= (size_t(256) << 20) && fourier_parts() == 1 && !windowed()) {
            if (invtrans_host_pipelined(nb_scalar, sp, nb_vordiv, vor, div, gp)) {
                return;
            }   // staging buffers unavailable: the serial path below
        }
    }
    ensure(d_vd_, vd_cap_, 2 * nvd);
    ensure(d_sp_, sp_cap_, std::max<size_t>(nsp, 1));
    ensure(d_gp_, gp_cap_, ngp);
    HIP_CHECK(hipMemcpyAsync(d_vd_, vor, nvd * sizeof(double), hipMemcpyHostToDevice, stream_));
    HIP_CHECK(hipMemcpyAsync(d_vd_ + nvd, div, nvd * sizeof(double), hipMemcpyHostToDevice, stream_));
    if (nsp) {
        HIP_CHECK(hipMemcpyAsync(d_sp_, sp, nsp * sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    invtrans_device(nb_scalar, d_sp_, nb_vordiv, d_vd_, d_vd_ + nvd, d_gp_);
    HIP_CHECK(hipMemcpyAsync(gp, d_gp_, ngp * sizeof(double), hipMemcpyDeviceToHost, stream_));
    synchronize();
}

void Trans::export_legendre_cache(void* buffer) const {
    double* sym  = (double*)buffer;
    double* asym = sym + geo_.size_sym();
    std::memset(buffer, 0, legendre_cache_bytes());
    compute_legendre_tables_reference_layout(geo_, sym, asym);
}

}  // namespace trans
}  // namespace atlas_amd
