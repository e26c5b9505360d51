// extern "C" layer of include/atlas_amd.h (Grid + Trans part).  No exception crosses this boundary.
#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/atlas_amd.h"
#include "capi_types.h"
#include "domain_crop.h"
#include "equal_regions.h"
#include "legendre_cache_uid.h"
#include "fft_plan.h"
#include "gaussian.h"
#include "legendre_host.h"
#include "env.h"
#include "regional_trans.h"
#include "trans.h"
#include "trans_plan.h"

using namespace atlas_amd;


namespace atlas_amd {
thread_local std::string g_last_note;    // diagnostics that are not errors (e.g. accepted-and-ignored option keys)
thread_local std::string g_last_error;
void set_last_error(const std::string& s) {
    // (a HIP call that failed inside the library clears the runtime's per-thread "last error" where it is caught -- hip_check --,
    // not here: an argument error must not swallow a sticky error of the application's own preceding HIP calls, nor touch the
    // HIP runtime from host-only paths)
    g_last_error = s;
}
}  // namespace atlas_amd

// a HIP call that failed is reported once, here: its error code must not stay in the runtime's per-thread "last error", where the
// launch check of the NEXT call would pick it up as its own
static inline bool hip_failed(hipError_t e) {
    if (e == hipSuccess) {
        return false;
    }
    (void)hipGetLastError();
    return true;
}
#define AA_TRY try {
#define AA_CATCH_INT                                 \
    }                                                \
    catch (const std::exception& e) {                \
        atlas_amd::set_last_error(e.what());         \
        return 1;                                    \
    }                                                \
    catch (...) {                                    \
        atlas_amd::set_last_error("unknown error");  \
        return 1;                                    \
    }                                                \
    return 0;
#define AA_CATCH_PTR                                 \
    }                                                \
    catch (const std::exception& e) {                \
        atlas_amd::set_last_error(e.what());         \
        return nullptr;                              \
    }                                                \
    catch (...) {                                    \
        atlas_amd::set_last_error("unknown error");  \
        return nullptr;                              \
    }

static std::map<std::string, std::string> parse_config(const char* cfg) {
    std::map<std::string, std::string> kv;
    if (!cfg) {
        return kv;
    }
    std::stringstream ss(cfg);
    std::string item;
    while (std::getline(ss, item, ';')) {
        if (item.empty()) {
            continue;
        }
        auto eq = item.find('=');
        if (eq == std::string::npos) {
            throw std::invalid_argument("config item without '=': " + item);
        }
        kv[item.substr(0, eq)] = item.substr(eq + 1);
    }
    return kv;
}

namespace atlas_amd {
namespace trans {
hipError_t launch_vd2uv(const double* vor, const double* div, double* U, double* V, int T, int nf, hipStream_t stream);
}  // namespace trans
}  // namespace atlas_amd


extern "C" {

const char* atlas_amd__last_note(void) {
    return atlas_amd::g_last_note.c_str();
}
const char* atlas_amd__last_error(void) {
    return atlas_amd::g_last_error.c_str();
}
const char* atlas_amd__version(void) {
    return "atlas_amd 0.1.0 (gfx950; TransLocal invtrans + HaloExchange of ecmwf/atlas 0.44.1)";
}
int atlas_amd__set_ignore_env(int on) {
    atlas_amd::env_set_ignore(on != 0);
    return 0;
}
long long atlas_amd__effective_config(char* buf, long long capacity) {
    std::string text;
    int n                      = 0;
    const atlas_amd::EnvSwitch* sw = atlas_amd::env_switches(&n);
    for (int i = 0; i < n; ++i) {
        const bool dev_out = sw[i].cls == atlas_amd::EnvClass::dev && !atlas_amd::env_dev_switches_compiled_in();
        const char* v      = atlas_amd::env_get(sw[i].name);   // (a dev switch set in a product build: one line on stderr, once)
        const char* source = v ? "env" : (dev_out ? "compiled out" : (atlas_amd::env_ignored() && std::getenv(sw[i].name) ? "ignored" : "default"));
        text += std::string(sw[i].name) + "\t" + atlas_amd::env_class_name(sw[i].cls) + "\t" + (v ? v : sw[i].default_value) + "\t" + source +
                "\t" + sw[i].default_value + "\t" + sw[i].what + "\n";
    }
    if (buf && capacity > 0) {
        const size_t k = std::min<size_t>(text.size(), (size_t)capacity - 1);
        std::memcpy(buf, text.data(), k);
        buf[k] = 0;
    }
    return (long long)text.size() + 1;
}
int atlas_amd__device_count(void) {
    int n = 0;
    if (hip_failed(hipGetDeviceCount(&n))) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int atlas_amd__stream_wait_stream(void* waiting_stream, void* signalling_stream) {
    AA_TRY
    if (waiting_stream != signalling_stream) {
        hipEvent_t e;
        if (hip_failed(hipEventCreateWithFlags(&e, hipEventDisableTiming))) {
            throw std::runtime_error("stream_wait_stream: hipEventCreate failed");
        }
        hipError_t r = hipEventRecord(e, (hipStream_t)signalling_stream);
        if (r == hipSuccess) {
            r = hipStreamWaitEvent((hipStream_t)waiting_stream, e, 0);
        }
        (void)hipEventDestroy(e);   // released once the recorded work has completed
        if (hip_failed(r)) {
            throw std::runtime_error(std::string("stream_wait_stream: ") + hipGetErrorString(r));
        }
    }
    AA_CATCH_INT
}

void* atlas_amd__device_malloc(size_t bytes) {
    void* p = nullptr;
    if (hip_failed(hipMalloc(&p, bytes ? bytes : 1))) {
        (void)hipGetLastError();
        atlas_amd::set_last_error("device_malloc: hipMalloc failed");
        return nullptr;
    }
    return p;
}
int atlas_amd__device_free(void* ptr) {
    AA_TRY
    if (ptr && hip_failed(hipFree(ptr))) {
        throw std::runtime_error("device_free: hipFree failed");
    }
    AA_CATCH_INT
}
int atlas_amd__device_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes) {
    AA_TRY
    if (bytes && hip_failed(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice))) {
        throw std::runtime_error("device_memcpy_h2d failed");
    }
    AA_CATCH_INT
}
int atlas_amd__device_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes) {
    AA_TRY
    if (bytes && hip_failed(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost))) {
        throw std::runtime_error("device_memcpy_d2h failed");
    }
    AA_CATCH_INT
}
int atlas_amd__device_synchronize(void) {
    AA_TRY
    if (hip_failed(hipDeviceSynchronize())) {
        throw std::runtime_error("device_synchronize failed");
    }
    AA_CATCH_INT
}
int atlas_amd__set_device(int device) {
    AA_TRY
    if (hip_failed(hipSetDevice(device))) {
        throw std::runtime_error("set_device failed");
    }
    AA_CATCH_INT
}

int atlas_amd__LegendreCacheCreator__uid(const atlas_amd_Grid* grid, int truncation, int flt, char* out, size_t capacity) {
    AA_TRY
    if (!grid || !out || capacity == 0) {
        throw std::invalid_argument("LegendreCacheCreator::uid: null argument");
    }
    const std::string s = atlas_amd::trans::legendre_cache_uid(grid->g, truncation, flt != 0);
    if (s.size() + 1 > capacity) {
        throw std::invalid_argument("LegendreCacheCreator::uid: buffer too small");
    }
    std::memcpy(out, s.c_str(), s.size() + 1);
    AA_CATCH_INT
}
int64_t atlas_amd__LegendreCacheCreator__estimate(int truncation) {
    return atlas_amd::trans::legendre_cache_estimate(truncation);
}
int atlas_amd__LegendreCacheCreator__supported(const atlas_amd_Grid* grid) {
    return grid != nullptr;   // structured grid without projection (LegendreCacheCreatorLocal.cc:140-148): all this library has
}

int atlas_amd__eq_caps(int nb_regions, int capacity, int regions_per_zone[], double zone_colatitudes[], int* nb_zones) {
    AA_TRY
    if (!nb_zones) {
        throw std::invalid_argument("eq_caps: null argument");
    }
    std::vector<int> r;
    std::vector<double> c;
    atlas_amd::grid::eq_caps(nb_regions, r, c);
    *nb_zones = (int)r.size();
    for (int i = 0; i < (int)r.size() && i < capacity; ++i) {
        if (regions_per_zone) {
            regions_per_zone[i] = r[i];
        }
        if (zone_colatitudes) {
            zone_colatitudes[i] = c[i];
        }
    }
    AA_CATCH_INT
}
int atlas_amd__equal_regions_partition(const atlas_amd_Grid* grid, int nb_parts, int partition_out[]) {
    AA_TRY
    if (!grid || !partition_out) {
        throw std::invalid_argument("equal_regions_partition: null argument");
    }
    const std::vector<int> p = atlas_amd::grid::equal_regions_partition(grid->g, nb_parts);
    std::copy(p.begin(), p.end(), partition_out);
    AA_CATCH_INT
}

// ---------------------------------------------------------------- Grid
atlas_amd_Grid* atlas_amd__Grid__new_gaussian(const char* name) {
    AA_TRY
    if (!name) {
        throw std::invalid_argument("grid name is NULL");
    }
    return new atlas_amd_Grid{grid::make_gaussian_grid(name)};
    AA_CATCH_PTR
}
atlas_amd_Grid* atlas_amd__Grid__new_structured(int ny, const int nx[], const double lat_deg[]) {
    AA_TRY
    if (ny <= 0 || !nx || !lat_deg) {
        throw std::invalid_argument("Grid__new_structured: bad arguments");
    }
    for (int j = 0; j < ny; ++j) {
        if (nx[j] < 1) {
            throw std::invalid_argument("Grid__new_structured: every latitude needs at least one point");
        }
        if (!(lat_deg[j] <= 90. && lat_deg[j] >= -90.) || (j > 0 && !(lat_deg[j] < lat_deg[j - 1]))) {
            throw std::invalid_argument(
                "Grid__new_structured: latitudes must lie in [-90, 90] and decrease strictly (north to south)");
        }
    }
    grid::StructuredGrid g;
    g.nx.assign(nx, nx + ny);
    g.y.assign(lat_deg, lat_deg + ny);
    g.regular = true;
    for (int j = 1; j < ny; ++j) {
        g.regular = g.regular && (nx[j] == nx[0]);
    }
    g.name = "structured";
    return new atlas_amd_Grid{g};
    AA_CATCH_PTR
}
void atlas_amd__Grid__delete(atlas_amd_Grid* g) {
    delete g;
}
int atlas_amd__Grid__ny(const atlas_amd_Grid* g) {
    if (!g) {
        atlas_amd::set_last_error("Grid::ny: null handle");
        return -1;
    }
    return g->g.ny();
}
int atlas_amd__Grid__nxmax(const atlas_amd_Grid* g) {
    if (!g) {
        atlas_amd::set_last_error("Grid::nxmax: null handle");
        return -1;
    }
    return g->g.nxmax();
}
int64_t atlas_amd__Grid__size(const atlas_amd_Grid* g) {
    if (!g) {
        atlas_amd::set_last_error("Grid::size: null handle");
        return -1;
    }
    return g->g.size();
}
int atlas_amd__Grid__regular(const atlas_amd_Grid* g) {
    if (!g) {
        atlas_amd::set_last_error("Grid::regular: null handle");
        return -1;
    }
    return g->g.regular ? 1 : 0;
}
int atlas_amd__Grid__nx(const atlas_amd_Grid* g, int nx_out[]) {
    if (!g) {
        atlas_amd::set_last_error("Grid::nx: null handle");
        return -1;
    }
    std::memcpy(nx_out, g->g.nx.data(), sizeof(int) * g->g.nx.size());
    return 0;
}
int atlas_amd__Grid__y(const atlas_amd_Grid* g, double y_out[]) {
    if (!g) {
        atlas_amd::set_last_error("Grid::y: null handle");
        return -1;
    }
    std::memcpy(y_out, g->g.y.data(), sizeof(double) * g->g.y.size());
    return 0;
}
int atlas_amd__gaussian_latitudes_npole_spole(int N, double lats_out[]) {
    AA_TRY
    if (N < 1 || N > grid::kMaxGaussianN || !lats_out) {
        throw std::invalid_argument("gaussian_latitudes_npole_spole: N out of range or NULL output");
    }
    grid::gaussian_latitudes_npole_spole(N, lats_out);
    AA_CATCH_INT
}

// ---------------------------------------------------------------- Trans
atlas_amd_Trans* atlas_amd__Trans__new_config(const atlas_amd_Grid* grid, int truncation, const char* config,
                                              const void* legendre_cache, size_t legendre_cache_size) {
    AA_TRY
    if (!grid) {
        throw std::invalid_argument("grid is NULL");
    }
    trans::TransConfig cfg;
    bool mirror = false;
    atlas_amd::g_last_note.clear();
    // TransLocal's own option keys (src/atlas/option/TransOptions.cc:38-74, read at TransLocal.cc:61-110,326-335): every existing
    // caller of trans::Trans(grid, T, option::type("local") | option::fft("FFTW") | ...) passes some of them.  They select
    // third-party kernels or side files this implementation does not have (its FFT and GEMM are its own kernels, its tables are
    // always precomputed): accepted, validated where the reference validates, and otherwise ignored -- with a note.
    static const char* const ignored_keys[] = {"matrix_multiply", "precompute", "warning", "write_fft", "read_fft", "write_legendre",
                                               "read_legendre", "export_legendre", "global", "split_y", "nproma", "flt",
                                               "scalar_derivatives", "wind_EW_derivatives", "vorticity_divergence_fields"};
    auto note_ignored = [](const std::string& k, const std::string& v) {
        if (!atlas_amd::g_last_note.empty()) {
            atlas_amd::g_last_note += "; ";
        }
        atlas_amd::g_last_note += "option '" + k + "=" + v + "' accepted and ignored (TransLocal key without a counterpart here)";
    };
    for (auto& kv : parse_config(config)) {
        bool ign = false;
        for (const char* k : ignored_keys) {
            ign = ign || kv.first == k;
        }
        if (ign) {
            note_ignored(kv.first, kv.second);
            continue;
        }
        if (kv.first == "fft") {   // TransLocal.cc:90-103: OFF | FFTW | pocketfft, anything else throws
            if (kv.second != "OFF" && kv.second != "FFTW" && kv.second != "pocketfft") {
                throw std::invalid_argument("FFT backend \"" + kv.second + "\" is not one of the supported : OFF, FFTW, pocketfft");
            }
            note_ignored(kv.first, kv.second);   // (OFF selects the reference's dense Fourier matrix: same numbers, own FFT here)
            continue;
        }
        if (kv.first == "profile") {
            cfg.profile = std::stoi(kv.second) != 0;
        }
        else if (kv.first == "nparts") {
            cfg.nparts = std::stoi(kv.second);
        }
        else if (kv.first == "part") {
            cfg.part = std::stoi(kv.second);
        }
        else if (kv.first == "rows") {  // "j0:j1": zonal-band crop
            const size_t colon = kv.second.find(':');
            if (colon == std::string::npos) {
                throw std::invalid_argument("rows must be 'j0:j1'");
            }
            cfg.row_begin = std::stoi(kv.second.substr(0, colon));
            cfg.row_end   = std::stoi(kv.second.substr(colon + 1));
        }
        else if (kv.first == "domain") {  // "west,east,south,north": RectangularDomain crop of the (global) grid
            double b[4];
            size_t pos = 0;
            std::string v = kv.second;
            for (int i = 0; i < 4; ++i) {
                size_t used = 0;
                b[i]        = std::stod(v.substr(pos), &used);
                pos += used;
                if (i < 3) {
                    if (pos >= v.size() || v[pos] != ',') {
                        throw std::invalid_argument("domain must be 'west,east,south,north'");
                    }
                    ++pos;
                }
            }
            const grid::DomainCrop c = grid::crop_to_domain(grid->g, b[0], b[1], b[2], b[3]);
            cfg.row_begin = c.row_begin;
            cfg.row_end   = c.row_end;
            bool whole    = true;
            for (size_t r = 0; r < c.n.size(); ++r) {
                whole = whole && c.i0[r] == 0 && c.n[r] == grid->g.nx[c.row_begin + r];
            }
            if (!whole) {
                cfg.win_i0 = c.i0;
                cfg.win_n  = c.n;
            }
        }
        else if (kv.first == "shard") {
            if (kv.second != "m" && kv.second != "band" && kv.second != "mirror") {
                throw std::invalid_argument(
                    "shard must be 'm' (wavenumbers, exchange follows), 'band' (latitude bands) or 'mirror' (a northern "
                    "band and its mirror image)");
            }
            cfg.by_band = kv.second == "band";
            mirror      = kv.second == "mirror";
        }
        else if (kv.first == "tables") {
            if (kv.second != "host" && kv.second != "device") {
                throw std::invalid_argument("tables must be 'host' or 'device'");
            }
            cfg.device_tables = kv.second == "device" ? 1 : 0;
        }
        else if (kv.first == "type") {
            // atlas option::type: this library IS the "local" implementation (TransLocal.cc:57)
            if (kv.second != "local" && kv.second != "mi355x") {
                throw std::invalid_argument("unsupported trans type '" + kv.second + "'");
            }
        }
        else {
            throw std::invalid_argument("unknown config key '" + kv.first + "'");
        }
    }
    cfg.legendre_cache      = legendre_cache;
    cfg.legendre_cache_size = legendre_cache_size;
    if (mirror) {
        // Mirror-band decomposition: this object transforms the Legendre rows [b0, b1) of the grid in both hemispheres.
        // Every latitude is transformed independently of the others, so this is the zonal-band crop rows = [b0, 2 b1 - b0)
        // of the grid's two polar caps [0, b1) + [ny - b1, ny) taken as a grid of 2 b1 latitudes -- with the Fourier
        // truncation still computed for the full grid (ndgl, nxmax).  Same kernels, same arithmetic per row.
        if (cfg.row_end > cfg.row_begin || legendre_cache) {
            throw std::invalid_argument("shard=mirror cannot be combined with rows= or a Legendre cache");
        }
        const std::vector<int> b = trans::mirror_bands(grid->g, cfg.nparts);
        if (cfg.part < 0 || cfg.part >= cfg.nparts) {
            throw std::invalid_argument("Trans: invalid (nparts, part)");
        }
        const int b0 = b[cfg.part], b1 = b[cfg.part + 1];
        if (b1 <= b0) {
            throw std::invalid_argument("shard=mirror: more parts than the grid has row pairs to give them");
        }
        trans::TransConfig c2 = cfg;
        c2.nparts = 1, c2.part = 0, c2.by_band = false;
        c2.row_begin = b0, c2.row_end = 2 * b1 - b0;
        c2.ndgl = grid->g.ny(), c2.nxmax = grid->g.nxmax();
        auto* t      = new atlas_amd_Trans{new trans::Trans(trans::polar_caps_grid(grid->g, b1), truncation, c2), grid};
        t->mirror_b0 = b0, t->mirror_b1 = b1;
        return t;
    }
    return new atlas_amd_Trans{new trans::Trans(grid->g, truncation, cfg), grid};
    AA_CATCH_PTR
}
atlas_amd_Trans* atlas_amd__Trans__new(const atlas_amd_Grid* grid, int truncation) {
    return atlas_amd__Trans__new_config(grid, truncation, nullptr, nullptr, 0);
}
void atlas_amd__Trans__delete(atlas_amd_Trans* t) {
    if (t) {
        t->dist.reset();   // refers to *t->impl
        delete t->impl;
        delete t;
    }
}
// getters on a null handle (the reference: ATLAS_ASSERT(This != nullptr), TransInterface.cc:43-283): -1 and a message
#define AA_NULL_HANDLE(t, what, ret)                                            \
    if (!(t) || !(t)->impl) {                                                   \
        atlas_amd::set_last_error(std::string(what) + ": null Trans handle");   \
        return ret;                                                             \
    }
int atlas_amd__Trans__truncation(const atlas_amd_Trans* t) {
    AA_NULL_HANDLE(t, "Trans::truncation", -1)
    return t->impl->truncation();
}
int64_t atlas_amd__Trans__nb_gridpoints(const atlas_amd_Trans* t) {
    AA_NULL_HANDLE(t, "Trans::nb_gridpoints", -1)
    return t->impl->nb_gridpoints();
}
// ---------------------------------------------------------------- regional (non-nested) targets
struct atlas_amd_RegionalTrans {
    std::unique_ptr<trans::RegionalTrans> impl;
};
atlas_amd_RegionalTrans* atlas_amd__RegionalTrans__new(int nlon, double west, double dlon, int nlat, const double lats[],
                                                       int truncation) {
    AA_TRY
    if (!lats || nlat < 1) {
        throw std::invalid_argument("RegionalTrans: needs latitudes");
    }
    auto* h = new atlas_amd_RegionalTrans;
    try {
        h->impl.reset(new trans::RegionalTrans(nlon, west, dlon, std::vector<double>(lats, lats + nlat), truncation));
    }
    catch (...) {
        delete h;
        throw;
    }
    return h;
    AA_CATCH_PTR
}
atlas_amd_RegionalTrans* atlas_amd__RegionalTrans__new_unstructured(int npts, const double lons[], const double lats[],
                                                                    int truncation) {
    AA_TRY
    if (!lons || !lats || npts < 1) {
        throw std::invalid_argument("RegionalTrans: needs points");
    }
    auto* h = new atlas_amd_RegionalTrans;
    try {
        h->impl.reset(new trans::RegionalTrans(std::vector<double>(lons, lons + npts), std::vector<double>(lats, lats + npts),
                                               truncation));
    }
    catch (...) {
        delete h;
        throw;
    }
    return h;
    AA_CATCH_PTR
}
void atlas_amd__RegionalTrans__delete(atlas_amd_RegionalTrans* t) {
    delete t;
}
static void need(const atlas_amd_RegionalTrans* t) {
    if (!t || !t->impl) {
        throw std::invalid_argument("RegionalTrans: null handle");
    }
}
int64_t atlas_amd__RegionalTrans__nb_gridpoints(const atlas_amd_RegionalTrans* t) {
    if (!t) {
        atlas_amd::set_last_error("RegionalTrans::nb_gridpoints: null handle");
        return -1;
    }
    return t && t->impl ? t->impl->nb_gridpoints() : -1;
}
int atlas_amd__RegionalTrans__invtrans_scalar(atlas_amd_RegionalTrans* t, int nb_fields, const double scalar_spectra[],
                                              double gp_fields[]) {
    AA_TRY
    if (!t) {
        throw std::invalid_argument("RegionalTrans::invtrans_scalar: null handle");
    }
    need(t);
    t->impl->invtrans(nb_fields, scalar_spectra, gp_fields);
    AA_CATCH_INT
}
int atlas_amd__RegionalTrans__invtrans_scalar_device(atlas_amd_RegionalTrans* t, int nb_fields, const double* sp_dev,
                                                     double* gp_dev) {
    AA_TRY
    if (!t) {
        throw std::invalid_argument("RegionalTrans::invtrans_scalar_device: null handle");
    }
    need(t);
    t->impl->invtrans_scalar_device(nb_fields, sp_dev, gp_dev);
    AA_CATCH_INT
}
int atlas_amd__RegionalTrans__invtrans_vordiv(atlas_amd_RegionalTrans* t, int nb_scalar_fields, const double scalar_spectra[],
                                              int nb_vordiv_fields, const double vorticity_spectra[],
                                              const double divergence_spectra[], double gp_fields[]) {
    AA_TRY
    if (!t) {
        throw std::invalid_argument("RegionalTrans::invtrans_vordiv: null handle");
    }
    need(t);
    t->impl->invtrans(nb_scalar_fields, scalar_spectra, nb_vordiv_fields, vorticity_spectra, divergence_spectra, gp_fields);
    AA_CATCH_INT
}
int atlas_amd__RegionalTrans__synchronize(atlas_amd_RegionalTrans* t) {
    AA_TRY
    if (!t) {
        throw std::invalid_argument("RegionalTrans::synchronize: null handle");
    }
    need(t);
    t->impl->synchronize();
    AA_CATCH_INT
}
void* atlas_amd__RegionalTrans__stream(const atlas_amd_RegionalTrans* t) {
    if (!t) {
        atlas_amd::set_last_error("RegionalTrans::stream: null handle");
        return nullptr;
    }
    return t && t->impl ? (void*)t->impl->stream() : nullptr;
}

int atlas_amd__Grid__crop_to_domain(const atlas_amd_Grid* grid, double west, double east, double south, double north,
                                    int* row_begin, int* row_end, int first_index[], int count[], int capacity) {
    AA_TRY
    if (!grid || !row_begin || !row_end) {
        throw std::invalid_argument("crop_to_domain: NULL argument");
    }
    const grid::DomainCrop c = grid::crop_to_domain(grid->g, west, east, south, north);
    *row_begin = c.row_begin;
    *row_end   = c.row_end;
    if (first_index && count) {
        if (capacity < c.row_end - c.row_begin) {
            throw std::invalid_argument("crop_to_domain: arrays too short for the rows of the domain");
        }
        std::copy(c.i0.begin(), c.i0.end(), first_index);
        std::copy(c.n.begin(), c.n.end(), count);
    }
    AA_CATCH_INT
}
int64_t atlas_amd__Trans__nb_gridpoints_global(const atlas_amd_Trans* t) {
    AA_NULL_HANDLE(t, "Trans::nb_gridpoints_global", -1)
    return t->impl->nb_gridpoints_global();
}
int64_t atlas_amd__Trans__nb_spectral_coefficients(const atlas_amd_Trans* t) {
    AA_NULL_HANDLE(t, "Trans::nb_spectral_coefficients", -1)
    return (int64_t)t->impl->nb_spectral_coefficients();
}

// The reference asserts its arguments (ATLAS_ASSERT -> eckit::Exception); across a C ABI a null handle, a negative field count or a
// missing array of a call that announces fields is an error code, not a crash.  Zero fields of a kind need no array of that kind.
static atlas_amd::trans::Trans& checked_call(atlas_amd_Trans* t, const char* what, int nb_scalar, const void* sp, int nb_vordiv,
                                             const void* vor, const void* div, const void* gp) {
    if (!t || !t->impl) {
        throw std::invalid_argument(std::string(what) + ": null Trans handle");
    }
    if (nb_scalar < 0 || nb_vordiv < 0) {
        throw std::invalid_argument(std::string(what) + ": negative number of fields");
    }
    if ((nb_scalar > 0 && !sp) || (nb_vordiv > 0 && (!vor || !div)) || (nb_scalar + nb_vordiv > 0 && !gp)) {
        throw std::invalid_argument(std::string(what) + ": null array for a call with " + std::to_string(nb_scalar) +
                                    " scalar and " + std::to_string(nb_vordiv) + " vor/div fields");
    }
    return *t->impl;
}
int atlas_amd__Trans__invtrans_scalar(atlas_amd_Trans* t, int nb_fields, const double sp[], double gp[]) {
    AA_TRY
    checked_call(t, "invtrans_scalar", nb_fields, sp, 0, nullptr, nullptr, gp).invtrans(nb_fields, sp, gp);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_scalar_device(atlas_amd_Trans* t, int nb_fields, const double* sp, double* gp) {
    AA_TRY
    auto& tr = checked_call(t, "invtrans_scalar_device", nb_fields, sp, 0, nullptr, nullptr, gp);
    tr.invtrans_uv_device(tr.truncation(), nb_fields, 0, sp, gp);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans(atlas_amd_Trans* t, int nb_scalar, const double sp[], int nb_vordiv,
                               const double vor[], const double div[], double gp[]) {
    AA_TRY
    checked_call(t, "invtrans", nb_scalar, sp, nb_vordiv, vor, div, gp).invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_device(atlas_amd_Trans* t, int nb_scalar, const double* sp, int nb_vordiv,
                                      const double* vor, const double* div, double* gp) {
    AA_TRY
    checked_call(t, "invtrans_device", nb_scalar, sp, nb_vordiv, vor, div, gp)
        .invtrans_device(nb_scalar, sp, nb_vordiv, vor, div, gp);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_vordiv2wind(atlas_amd_Trans* t, int nb_fields, const double vor[],
                                           const double div[], double wind[]) {
    AA_TRY
    // TransLocal.cc:1486-1490: invtrans(0, nullptr, nb_vordiv, vor, div, gp)
    checked_call(t, "invtrans_vordiv2wind", 0, nullptr, nb_fields, vor, div, wind).invtrans(0, nullptr, nb_fields, vor, div, wind);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_scalar_device_f32(atlas_amd_Trans* t, int nb_fields, const float* scalar_spectra,
                                                 float* gp_fields) {
    AA_TRY
    checked_call(t, "invtrans_scalar_device_f32", nb_fields, scalar_spectra, 0, nullptr, nullptr, gp_fields)
        .invtrans_scalar_device_f32(nb_fields, scalar_spectra, gp_fields);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_device_f32(atlas_amd_Trans* t, int nb_scalar_fields, const float* scalar_spectra_dev,
                                          int nb_vordiv_fields, const float* vorticity_spectra_dev,
                                          const float* divergence_spectra_dev, float* gp_fields_dev) {
    AA_TRY
    checked_call(t, "invtrans_device_f32", nb_scalar_fields, scalar_spectra_dev, nb_vordiv_fields, vorticity_spectra_dev,
                 divergence_spectra_dev, gp_fields_dev)
        .invtrans_device_f32(nb_scalar_fields, scalar_spectra_dev, nb_vordiv_fields, vorticity_spectra_dev, divergence_spectra_dev,
                             gp_fields_dev);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_scalar_f32(atlas_amd_Trans* t, int nb_fields, const float scalar_spectra[],
                                          float gp_fields[]) {
    AA_TRY
    checked_call(t, "invtrans_scalar_f32", nb_fields, scalar_spectra, 0, nullptr, nullptr, gp_fields)
        .invtrans_scalar_f32(nb_fields, scalar_spectra, gp_fields);
    AA_CATCH_INT
}
static int not_implemented(const char* what) {
    atlas_amd::set_last_error(std::string("Not implemented: ") + what +
                              " (TransLocal does not implement it either, TransLocal.cc:848-857,899-927,1599-1685)");
    return 2;
}
int atlas_amd__Trans__dirtrans_scalar(atlas_amd_Trans*, int, const double[], double[]) {
    return not_implemented("dirtrans");
}
int atlas_amd__Trans__dirtrans_wind2vordiv(atlas_amd_Trans*, int, const double[], double[], double[]) {
    return not_implemented("dirtrans_wind2vordiv");
}
int atlas_amd__Trans__invtrans_adj_scalar(atlas_amd_Trans*, int, const double[], double[]) {
    return not_implemented("invtrans_adj");
}
int atlas_amd__Trans__invtrans_adj(atlas_amd_Trans*, int, const double[], int, double[], double[], double[]) {
    return not_implemented("invtrans_adj");
}
int atlas_amd__Trans__invtrans_vordiv2wind_adj(atlas_amd_Trans*, int, const double[], double[], double[]) {
    return not_implemented("invtrans_vordiv2wind_adj");
}

// ---- backend registry (Trans.cc:37-48, TransFactory): one implementation under two names
static std::string& current_backend() {
    static std::string b = "local";
    return b;
}
int atlas_amd__Trans__has_backend(const char* backend) {
    return backend && (std::string(backend) == "local" || std::string(backend) == "mi355x");
}
int atlas_amd__Trans__set_backend(const char* backend) {
    AA_TRY
    if (!atlas_amd__Trans__has_backend(backend)) {   // ATLAS_ASSERT(hasBackend(backend)), Trans.cc:42
        throw std::invalid_argument(std::string("no trans backend '") + (backend ? backend : "(null)") + "'");
    }
    current_backend() = backend;
    AA_CATCH_INT
}
int atlas_amd__Trans__backend(char** backend, size_t* size) {
    AA_TRY
    const std::string& s = current_backend();
    *size                = s.size();
    *backend             = (char*)std::malloc(s.size() + 1);
    std::memcpy(*backend, s.c_str(), s.size() + 1);
    AA_CATCH_INT
}
int atlas_amd__Trans__handle(const atlas_amd_Trans* t, int* handle) {
    AA_TRY
    (void)t;
    (void)handle;
    throw std::logic_error("Not implemented: Trans::handle() (TransImpl.cc:20-22; only the IFS backend has one)");
    AA_CATCH_INT
}
const atlas_amd_Spectral* atlas_amd__Trans__spectral(const atlas_amd_Trans* t) {
    if (!t) {
        return nullptr;
    }
    const_cast<atlas_amd_Trans*>(t)->spectral.truncation = t->impl->truncation();
    return &t->spectral;
}
int atlas_amd__Spectral__truncation(const atlas_amd_Spectral* s) {
    if (!s) {
        atlas_amd::set_last_error("Spectral::truncation: null handle");
        return -1;
    }
    return s ? s->truncation : -1;
}
int64_t atlas_amd__Spectral__nb_spectral_coefficients(const atlas_amd_Spectral* s) {
    if (!s) {
        atlas_amd::set_last_error("Spectral::nb_spectral_coefficients: null handle");
        return -1;
    }
    return s ? (int64_t)(s->truncation + 1) * (s->truncation + 2) : 0;
}
int64_t atlas_amd__Spectral__nb_spectral_coefficients_global(const atlas_amd_Spectral* s) {
    return atlas_amd__Spectral__nb_spectral_coefficients(s);
}
const atlas_amd_Grid* atlas_amd__Trans__grid(const atlas_amd_Trans* t) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::grid: null handle");
        return nullptr;
    }
    return t ? t->grid : nullptr;
}

// ---- Field / FieldSet overloads (TransLocal.cc:818-897)
static void require_rank1(const atlas_amd_Field* f, const char* what) {
    if (!f || !f->data) {
        throw std::invalid_argument(std::string(what) + ": field is NULL");
    }
    if (f->rank != 1) {  // ATLAS_ASSERT(field.rank() == 1, ...), TransLocal.cc:821-822,875-876
        throw std::invalid_argument(std::string(what) + ": Only rank-1 fields supported at the moment");
    }
}
int atlas_amd__Trans__invtrans_field(atlas_amd_Trans* t, const atlas_amd_Field* spfield, atlas_amd_Field* gpfield) {
    AA_TRY
    if (!t || !t->impl || !spfield || !gpfield) {
        throw std::invalid_argument("Trans::invtrans_field: null handle");
    }
    require_rank1(spfield, "spfield");
    require_rank1(gpfield, "gpfield");
    // TransLocal.cc:826-831 only prints debug output when the grid-point field is shorter than the grid; the call
    // site's spectral size is not checked there either, but reading past the caller's buffer is not an option here
    if ((size_t)spfield->shape[0] < t->impl->nb_spectral_coefficients() ||
        gpfield->shape[0] < (long)t->impl->nb_gridpoints_global()) {
        throw std::invalid_argument("invtrans(Field, Field): field shorter than the spectral / grid size");
    }
    t->impl->invtrans(1, spfield->data, gpfield->data);
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_fieldset(atlas_amd_Trans* t, const atlas_amd_Field* spfields, int nb_spfields,
                                        atlas_amd_Field* gpfields, int nb_gpfields) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::invtrans_fieldset: null handle");
        return 1;
    }
    if (nb_spfields != nb_gpfields) {  // ATLAS_ASSERT(spfields.size() == gpfields.size()), :840
        atlas_amd::set_last_error("invtrans(FieldSet, FieldSet): spfields.size() != gpfields.size()");
        return 1;
    }
    for (int f = 0; f < nb_spfields; ++f) {
        const int rc = atlas_amd__Trans__invtrans_field(t, spfields + f, gpfields + f);
        if (rc) {
            return rc;
        }
    }
    return 0;
}
int atlas_amd__Trans__invtrans_vordiv2wind_field(atlas_amd_Trans* t, const atlas_amd_Field* spvor,
                                                 const atlas_amd_Field* spdiv, atlas_amd_Field* gpwind) {
    AA_TRY
    if (!t || !t->impl || !spvor || !spdiv) {
        throw std::invalid_argument("Trans::invtrans_vordiv2wind_field: null handle");
    }
    require_rank1(spvor, "spvor");
    require_rank1(spdiv, "spdiv");
    if (!gpwind || !gpwind->data || gpwind->rank != 2) {
        throw std::invalid_argument("gpwind: rank-2 field expected");
    }
    const size_t nspec = t->impl->nb_spectral_coefficients();  // 2 * legendre_size(truncation) * 1, :881
    if ((size_t)spvor->shape[0] != nspec || (size_t)spdiv->shape[0] != nspec) {
        throw std::invalid_argument("invtrans_vordiv2wind: spectral field size != 2 * legendre_size(truncation)");
    }
    const long npts = (long)t->impl->nb_gridpoints_global();
    if (gpwind->shape[0] == 2 && gpwind->shape[1] == npts) {
        t->impl->invtrans(0, nullptr, 1, spvor->data, spdiv->data, gpwind->data);
    }
    else if (gpwind->shape[0] == npts && gpwind->shape[1] == 2) {
        std::vector<double> tmp(size_t(2) * npts);
        t->impl->invtrans(0, nullptr, 1, spvor->data, spdiv->data, tmp.data());
        // gp_transpose(grid().size(), 2, gp_tmp, gp_fields), TransLocal.cc:861-867, taken literally
        for (long jgp = 0; jgp < npts; ++jgp) {
            for (int jfld = 0; jfld < 2; ++jfld) {
                gpwind->data[jfld * npts + jgp] = tmp[jgp * 2 + jfld];
            }
        }
    }
    else {
        return not_implemented("invtrans_vordiv2wind for this wind field shape");
    }
    AA_CATCH_INT
}
int atlas_amd__Trans__invtrans_grad_field(atlas_amd_Trans*, const atlas_amd_Field*, atlas_amd_Field*) {
    return not_implemented("invtrans_grad");
}
int atlas_amd__Trans__invtrans_adj_field(atlas_amd_Trans*, const atlas_amd_Field*, atlas_amd_Field*) {
    return not_implemented("invtrans_adj");
}
int atlas_amd__Trans__invtrans_adj_fieldset(atlas_amd_Trans*, const atlas_amd_Field*, int, atlas_amd_Field*, int) {
    return not_implemented("invtrans_adj");
}
int atlas_amd__Trans__invtrans_grad_adj_field(atlas_amd_Trans*, const atlas_amd_Field*, atlas_amd_Field*) {
    return not_implemented("invtrans_grad_adj");
}
int atlas_amd__Trans__invtrans_vordiv2wind_adj_field(atlas_amd_Trans*, const atlas_amd_Field*, atlas_amd_Field*,
                                                     atlas_amd_Field*) {
    return not_implemented("invtrans_vordiv2wind_adj");
}
int atlas_amd__Trans__dirtrans_field(atlas_amd_Trans*, const atlas_amd_Field*, atlas_amd_Field*) {
    return not_implemented("dirtrans");
}
int atlas_amd__Trans__dirtrans_fieldset(atlas_amd_Trans*, const atlas_amd_Field*, int, atlas_amd_Field*, int) {
    return not_implemented("dirtrans");
}
int atlas_amd__Trans__dirtrans_wind2vordiv_field(atlas_amd_Trans*, const atlas_amd_Field*, atlas_amd_Field*,
                                                 atlas_amd_Field*) {
    return not_implemented("dirtrans_wind2vordiv");
}

// ---- VorDivToUV::execute (VorDivToUVLocal.cc:187-189)
static void vd2uv_check(int truncation, int nb_coeff, int nb_fields) {
    if (truncation < 0 || nb_fields < 0) {
        throw std::invalid_argument("VorDivToUV: negative truncation / field count");
    }
    if (nb_coeff != (truncation + 1) * (truncation + 2)) {
        throw std::invalid_argument("VorDivToUV: nb_coeff != (truncation+1)*(truncation+2)");
    }
}
int atlas_amd__VorDivToUV__execute_device(int truncation, int nb_coeff, int nb_fields, const double* vor,
                                          const double* div, double* U, double* V, void* stream) {
    AA_TRY
    vd2uv_check(truncation, nb_coeff, nb_fields);
    if (nb_fields > 0) {
        hipError_t e = trans::launch_vd2uv(vor, div, U, V, truncation, nb_fields, (hipStream_t)stream);
        if (hip_failed(e)) {
            throw std::runtime_error(std::string("vd2uv launch: ") + hipGetErrorString(e));
        }
    }
    AA_CATCH_INT
}
int atlas_amd__VorDivToUV__execute(int truncation, int nb_coeff, int nb_fields, const double vor[], const double div[],
                                   double U[], double V[]) {
    AA_TRY
    vd2uv_check(truncation, nb_coeff, nb_fields);
    const size_t n = size_t(nb_coeff) * size_t(nb_fields);
    if (n == 0) {
        return 0;
    }
    double* d = nullptr;
    auto ck   = [](hipError_t e) {
        if (hip_failed(e)) {
            throw std::runtime_error(std::string("VorDivToUV: ") + hipGetErrorString(e));
        }
    };
    ck(hipMalloc((void**)&d, 4 * n * sizeof(double)));
    try {
        ck(hipMemcpy(d, vor, n * sizeof(double), hipMemcpyHostToDevice));
        ck(hipMemcpy(d + n, div, n * sizeof(double), hipMemcpyHostToDevice));
        ck(trans::launch_vd2uv(d, d + n, d + 2 * n, d + 3 * n, truncation, nb_fields, nullptr));
        ck(hipMemcpy(U, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost));
        ck(hipMemcpy(V, d + 3 * n, n * sizeof(double), hipMemcpyDeviceToHost));
    }
    catch (...) {
        (void)hipFree(d);
        throw;
    }
    (void)hipFree(d);
    AA_CATCH_INT
}

void* atlas_amd__Trans__stream(atlas_amd_Trans* t) {
    AA_NULL_HANDLE(t, "Trans::stream", nullptr)
    return (void*)t->impl->stream();
}
int atlas_amd__Trans__set_stream(atlas_amd_Trans* t, void* s) {
    AA_NULL_HANDLE(t, "Trans::set_stream", 1)
    AA_TRY
    t->impl->set_stream((hipStream_t)s);
    AA_CATCH_INT
}
int atlas_amd__Trans__synchronize(atlas_amd_Trans* t) {
    AA_NULL_HANDLE(t, "Trans::synchronize", 1)
    AA_TRY
    t->impl->synchronize();
    AA_CATCH_INT
}
size_t atlas_amd__Trans__legendre_cache_size(const atlas_amd_Trans* t) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::legendre_cache_size: null handle");
        return -1;
    }
    return t->impl->legendre_cache_bytes();
}
int atlas_amd__Trans__legendre_cache_export(const atlas_amd_Trans* t, void* buffer, size_t size) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::legendre_cache_export: null handle");
    }
    if (size != t->impl->legendre_cache_bytes()) {
        throw std::invalid_argument("legendre_cache_export: wrong buffer size");
    }
    t->impl->export_legendre_cache(buffer);
    AA_CATCH_INT
}
int atlas_amd__Trans__mirror_rows(const atlas_amd_Trans* t, int out[2]) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::mirror_rows: null handle");
    }
    if (t->mirror_b0 < 0) {
        throw std::invalid_argument("Trans__mirror_rows: the object was not built with shard=mirror");
    }
    out[0] = t->mirror_b0;
    out[1] = t->mirror_b1;
    AA_CATCH_INT
}
int atlas_amd__mirror_bands(const atlas_amd_Grid* grid, int nparts, int bands_out[]) {
    AA_TRY
    if (!grid) {
        throw std::invalid_argument("mirror_bands: null handle");
    }
    const std::vector<int> b = trans::mirror_bands(grid->g, nparts);
    std::memcpy(bands_out, b.data(), sizeof(int) * b.size());
    AA_CATCH_INT
}
int atlas_amd__latitude_bands(const atlas_amd_Grid* grid, int truncation, int nparts, int bands_out[]) {
    AA_TRY
    if (!grid || !bands_out || nparts < 1) {
        throw std::invalid_argument("latitude_bands: bad arguments");
    }
    const std::vector<int> b = trans::latitude_bands(trans::make_geometry(grid->g, truncation), nparts);
    std::memcpy(bands_out, b.data(), sizeof(int) * b.size());
    AA_CATCH_INT
}
int atlas_amd__trans_geometry_probe(const atlas_amd_Grid* grid, int truncation, int caps_rows, int nlat0_out[],
                                    int row_mmax_out[]) {
    AA_TRY
    if (!grid) {
        throw std::invalid_argument("trans_geometry_probe: null handle");
    }
    // geometry of the grid itself (caps_rows == 0) or of its two polar caps of caps_rows latitudes each, seen as part
    // of the full grid (what shard=mirror builds)
    const trans::TransGeometry geo =
        caps_rows > 0 ? trans::make_geometry(trans::polar_caps_grid(grid->g, caps_rows), truncation, grid->g.ny(),
                                             grid->g.nxmax())
                      : trans::make_geometry(grid->g, truncation);
    std::memcpy(nlat0_out, geo.nlat0.data(), sizeof(int) * geo.nlat0.size());
    for (int j = 0; j < geo.nlats; ++j) {
        const int jleg  = j < geo.nlatsLeg ? j : geo.nlats - 1 - j;
        row_mmax_out[j] = geo.mmax_leg[jleg];
    }
    AA_CATCH_INT
}
int atlas_amd__Trans__legendre_table_download(const atlas_amd_Trans* t, double* out, size_t size) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::legendre_table_download: null handle");
    }
    t->impl->download_legendre_table(out, size);
    AA_CATCH_INT
}
int atlas_amd__Trans__fourier_row_pitch(const atlas_amd_Trans* t, int nb_fields) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::fourier_row_pitch: null handle");
        return -1;
    }
    return t->impl->fourier_row_pitch(nb_fields);
}
int64_t atlas_amd__Trans__fourier_size(const atlas_amd_Trans* t, int nb_fields) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::fourier_size: null handle");
        return -1;
    }
    return (int64_t)t->impl->fourier_doubles(nb_fields);
}
int atlas_amd__Trans__owned_wavenumbers(const atlas_amd_Trans* t) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::owned_wavenumbers: null handle");
        return -1;
    }
    return t->impl->owned_wavenumbers();
}
int atlas_amd__Trans__bands(const atlas_amd_Trans* t, int out[]) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::bands: null handle");
        return -1;
    }
    const auto& b = t->impl->bands();
    std::memcpy(out, b.data(), sizeof(int) * b.size());
    return 0;
}
int atlas_amd__Trans__legendre_device(atlas_amd_Trans* t, int trc_in, int nb_fields, const double* sp,
                                      double* fourier) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::legendre_device: null handle");
    }
    t->impl->legendre_device(trc_in, nb_fields, sp, fourier);
    AA_CATCH_INT
}
int atlas_amd__Trans__fourier_device(atlas_amd_Trans* t, int nb_fields, int nb_vordiv,
                                     const double* const part_base[], const int part_cnt[], double* gp) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::fourier_device: null handle");
    }
    t->impl->fourier_device(nb_fields, nb_vordiv, part_base, part_cnt, gp);
    AA_CATCH_INT
}
int atlas_amd__Trans__nlat0(const atlas_amd_Trans* t, int out[]) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::nlat0: null handle");
        return -1;
    }
    const auto& v = t->impl->geometry().nlat0;
    std::memcpy(out, v.data(), sizeof(int) * v.size());
    return 0;
}
int atlas_amd__Trans__fft_row_classes(const atlas_amd_Trans* t, int out[]) {
    AA_TRY
    if (!t || !t->impl || !out) {
        throw std::invalid_argument("fft_row_classes: null argument");
    }
    const auto& geo = t->impl->geometry();
    const auto& ps  = t->impl->fft_plans();
    for (int j = 0; j < geo.nlats; ++j) {
        const int pi = ps.plan_index(geo.regular ? geo.nxmax : geo.nx[j]);
        if (pi < 0) {
            throw std::logic_error("fft_row_classes: row without a plan");
        }
        const fft::FftRowPlan& pl = ps.plans[pi];
        out[3 * j]     = pl.method;
        out[3 * j + 1] = pl.shape.M;
        out[3 * j + 2] = t->impl->fft_row_kernel(pl);
#if defined(ATLAS_AMD_EXPERIMENTS)
        if (pl.method == fft::FFT_NATIVE) {   // every native half length is its own shape: the class is (first radix, stages)
            out[3 * j + 1] = pl.nat.radix[0] * 10 + pl.nat.ns;
        }
#endif
    }
    AA_CATCH_INT
}
double atlas_amd__Trans__legendre_flops(const atlas_amd_Trans* t, int nb_fields) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::legendre_flops: null handle");
        return -1.0;
    }
    return trans::legendre_flops(t->impl->geometry(), nb_fields);
}
int64_t atlas_amd__Trans__legendre_table_bytes(const atlas_amd_Trans* t) {
    if (!t || !t->impl) {
        atlas_amd::set_last_error("Trans::legendre_table_bytes: null handle");
        return -1;
    }
    return t->impl->legendre_work().table_doubles * 8;
}
int atlas_amd__Trans__timings(atlas_amd_Trans* t, double out[4], int reset) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::timings: null handle");
    }
    trans::StageTimings s = t->impl->timings();
    out[0]                = s.legendre_ms;
    out[1]                = s.legendre_calls;
    out[2]                = s.fourier_ms;
    out[3]                = s.fourier_calls;
    if (reset) {
        t->impl->reset_timings();
    }
    AA_CATCH_INT
}
int atlas_amd__Trans__fourier_launch_plan(const atlas_amd_Trans* t, int out[3]) {
    AA_TRY
    if (!t || !t->impl || !out) {
        throw std::invalid_argument("fourier_launch_plan: null argument");
    }
    t->impl->fourier_launch_plan(out);
    AA_CATCH_INT
}
int atlas_amd__Trans__timings_vordiv(atlas_amd_Trans* t, double out[2], int reset) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::timings_vordiv: null handle");
    }
    trans::StageTimings s = t->impl->timings();
    out[0]                = s.prepare_ms;
    out[1]                = s.prepare_calls;
    if (reset) {
        t->impl->reset_timings();
    }
    AA_CATCH_INT
}
int atlas_amd__Trans__set_profile(atlas_amd_Trans* t, int on) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::set_profile: null handle");
    }
    t->impl->synchronize();
    t->impl->set_profile(on != 0);
    AA_CATCH_INT
}

int atlas_amd__Trans__fft_phase_profile(atlas_amd_Trans* t, int enable, unsigned long long out[64]) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("Trans::fft_phase_profile: null handle");
    }
    if (out) {
        t->impl->read_phase_profile(out);
    }
    t->impl->enable_phase_profile(enable != 0);
    AA_CATCH_INT
}

int atlas_amd__Trans__fft_trace(atlas_amd_Trans* t, unsigned long long words, unsigned long long* out) {
    AA_TRY
    if (!t || !t->impl) {
        throw std::invalid_argument("fft_trace: null Trans");
    }
    t->impl->fft_trace(words, out);
    AA_CATCH_INT
}

// ---------------------------------------------------------------- host-only helpers
int atlas_amd__fourier_truncation(int truncation, int nx, int nxmax, int ndgl, double lat_rad, int fullgrid) {
    return trans::fourier_truncation(truncation, nx, nxmax, ndgl, lat_rad, fullgrid != 0);
}
int atlas_amd__legendre_reference_sizes(const atlas_amd_Grid* grid, int truncation, size_t* size_sym,
                                        size_t* size_asym) {
    AA_TRY
    if (!grid) {
        throw std::invalid_argument("legendre_reference_sizes: null handle");
    }
    trans::TransGeometry geo = trans::make_geometry(grid->g, truncation);
    *size_sym                = geo.size_sym();
    *size_asym               = geo.size_asym();
    AA_CATCH_INT
}
int atlas_amd__legendre_reference_tables(const atlas_amd_Grid* grid, int truncation, double* leg_sym,
                                         size_t size_sym, double* leg_asym, size_t size_asym) {
    AA_TRY
    if (!grid) {
        throw std::invalid_argument("legendre_reference_tables: null handle");
    }
    trans::TransGeometry geo = trans::make_geometry(grid->g, truncation);
    if (size_sym != geo.size_sym() || size_asym != geo.size_asym()) {
        throw std::invalid_argument("legendre_reference_tables: wrong sizes");
    }
    std::memset(leg_sym, 0, size_sym * sizeof(double));
    std::memset(leg_asym, 0, size_asym * sizeof(double));
    trans::compute_legendre_tables_reference_layout(geo, leg_sym, leg_asym);
    AA_CATCH_INT
}
int atlas_amd__legendre_gen_host_selfcheck(const atlas_amd_Grid* grid, int truncation, int nparts, int part,
                                           int by_band, long long* table_doubles, long long* mismatches) {
    AA_TRY
    if (!grid) {
        throw std::invalid_argument("legendre_gen_host_selfcheck: null handle");
    }
    if (nparts < 1 || part < 0 || part >= nparts) {
        throw std::invalid_argument("legendre_gen_host_selfcheck: bad nparts / part");
    }
    trans::TransGeometry geo = trans::make_geometry(grid->g, truncation);
    trans::LegendreWork work = trans::make_legendre_work(geo, nparts, part, by_band != 0);
    const size_t n           = (size_t)work.table_doubles;
    std::vector<double> a(n, 0.), b(n, 0.);
    trans::compute_legendre_table_tiled(geo, work, a.data());
    trans::compute_legendre_table_tiled_emulated(geo, work, b.data());
    long long bad = 0;
    for (size_t i = 0; i < n; ++i) {
        bad += std::memcmp(&a[i], &b[i], sizeof(double)) != 0;
    }
    *table_doubles = (long long)n;
    *mismatches    = bad;
    AA_CATCH_INT
}
int atlas_amd__fft_host_row(int n, const double* modes, int mmax, double* out) {
    AA_TRY
    if (n < 1 || !modes || !out) {
        throw std::invalid_argument("fft_host_row: n >= 1 and non-null arrays are required");
    }
    fft::FftPlanSet ps = fft::make_fft_plans({n});
    fft::host_execute_row(ps, 0, reinterpret_cast<const fft::cplx*>(modes), mmax, out);
    AA_CATCH_INT
}
int atlas_amd__fft_host_row_native(int n, const double* modes, int mmax, double* out) {
    AA_TRY
    if (n < 1 || !modes || !out) {
        throw std::invalid_argument("fft_host_row_native: n >= 1 and non-null arrays are required");
    }
#if !defined(ATLAS_AMD_EXPERIMENTS)
    throw std::runtime_error("fft_host_row_native: the native mixed-radix rows live in tools/experiments (make -C atlas_amd/csrc experiments)");
#endif
    fft::PlanOptions po;
    po.native = true;
    fft::FftPlanSet ps = fft::make_fft_plans({n}, po);
    if (ps.plans.at(0).method != fft::FFT_NATIVE) {
        throw std::invalid_argument("fft_host_row_native: row length " + std::to_string(n) + " has no native plan");
    }
    fft::host_execute_row(ps, 0, reinterpret_cast<const fft::cplx*>(modes), mmax, out);
    AA_CATCH_INT
}
int atlas_amd__fft_host_row_bluestein(int n, const double* modes, int mmax, double* out) {
    AA_TRY
    if (n < 1 || !modes || !out) {
        throw std::invalid_argument("fft_host_row_bluestein: n >= 1 and non-null arrays are required");
    }
    fft::PlanOptions po;
    po.native = false;
    fft::FftPlanSet ps = fft::make_fft_plans({n}, po);
    fft::host_execute_row(ps, 0, reinterpret_cast<const fft::cplx*>(modes), mmax, out);
    AA_CATCH_INT
}
int atlas_amd__fft_plan_info(int n, int native, int out[16]) {
    AA_TRY
    if (n < 1 || !out) {
        throw std::invalid_argument("fft_plan_info: n >= 1 and a non-null array are required");
    }
    fft::PlanOptions po;
#if defined(ATLAS_AMD_EXPERIMENTS)
    po.native                  = native != 0;
#else
    if (native != 0) {
        throw std::runtime_error("fft_plan_info: the native mixed-radix rows live in tools/experiments (make -C atlas_amd/csrc experiments)");
    }
#endif
    fft::FftPlanSet ps         = fft::make_fft_plans({n}, po);
    const fft::FftRowPlan& pl = ps.plans.at(0);
    for (int i = 0; i < 16; ++i) {
        out[i] = 0;
    }
    out[0] = pl.method;
    out[1] = pl.shape.M;
    out[2] = pl.lds_complex;
    out[3] = pl.shape.nstages;
    for (int i = 0; i < pl.shape.nstages && i < 8; ++i) {
        out[4 + i] = pl.shape.radix[i];
    }
    out[12] = pl.ct_k >= 0 ? 1 : 0;
#if defined(ATLAS_AMD_EXPERIMENTS)
    out[13] = pl.method == fft::FFT_NATIVE ? pl.nat.pitch : 0;
#endif
    out[14] = (int)ps.nat_table.size();
    AA_CATCH_INT
}
int atlas_amd__fft_host_row_hybrid(int n, const double* modes, int mmax, double* out) {
    AA_TRY
    if (n < 1 || !modes || !out) {
        throw std::invalid_argument("fft_host_row_hybrid: n >= 1 and non-null arrays are required");
    }
#if !defined(ATLAS_AMD_EXPERIMENTS)
    throw std::runtime_error("fft_host_row_hybrid: the dense-stage rows live in tools/experiments (make -C atlas_amd/csrc experiments)");
#endif
    fft::PlanOptions po;
    po.hybrid       = true;
    po.hybrid_min_h = 2;
    fft::FftPlanSet ps = fft::make_fft_plans({n}, po);
    fft::host_execute_row(ps, 0, reinterpret_cast<const fft::cplx*>(modes), mmax, out);
    AA_CATCH_INT
}
int atlas_amd__fft_host_row_coarse(int n, const double* modes, int mmax, double* out) {
    AA_TRY
    if (n < 1 || !modes || !out) {
        throw std::invalid_argument("fft_host_row_coarse: n >= 1 and non-null arrays are required");
    }
    fft::PlanOptions po;
    po.coarse_classes = true;
    fft::FftPlanSet ps = fft::make_fft_plans({n}, po);
    fft::host_execute_row(ps, 0, reinterpret_cast<const fft::cplx*>(modes), mmax, out);
    AA_CATCH_INT
}
int atlas_amd__fft_host_row_generic(int n, const double* modes, int mmax, double* out) {
    AA_TRY
    if (n < 1 || !modes || !out) {
        throw std::invalid_argument("fft_host_row_generic: n >= 1 and non-null arrays are required");
    }
    fft::FftPlanSet ps = fft::make_fft_plans({n}, false);
    fft::host_execute_row(ps, 0, reinterpret_cast<const fft::cplx*>(modes), mmax, out, 256, false);
    AA_CATCH_INT
}

}  // extern "C"
