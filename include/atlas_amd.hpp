// atlas_amd.hpp -- C++ host interface above the C ABI (include/atlas_amd.h), header only.
//
// It mirrors the part of Atlas's C++ API that the hot path is reached through, so that code (and tests) written
// against atlas::trans::Trans / atlas::parallel::HaloExchange / atlas::functionspace::StructuredColumns read the same:
//
//   atlas_amd::StructuredGrid                    atlas::StructuredGrid             src/atlas/grid/StructuredGrid.h
//   atlas_amd::trans::Trans                      atlas::trans::Trans               src/atlas/trans/Trans.h:42-260
//        hasBackend / backend                                                      src/atlas/trans/Trans.cc:37-48
//        invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp) and the two short forms  src/atlas/trans/detail/TransImpl.h:116-141
//        dirtrans / invtrans_adj (throw NotImplemented, as TransLocal does)        src/atlas/trans/local/TransLocal.cc:848-857,899-927
//   atlas_amd::trans::VorDivToUV                 atlas::trans::VorDivToUV          src/atlas/trans/VorDivToUV.h:36-133
//   atlas_amd::parallel::HaloExchange            atlas::parallel::HaloExchange     src/atlas/parallel/HaloExchange.h:44-225
//   atlas_amd::functionspace::StructuredColumns  atlas::functionspace::StructuredColumns
//                                                                                 src/atlas/functionspace/StructuredColumns.h
//
// Error behaviour: where Atlas throws eckit::Exception (ATLAS_ASSERT / ATLAS_NOTIMPLEMENTED), these classes throw
// atlas_amd::Exception / atlas_amd::NotImplemented carrying the library's message.  There is no CPU fallback: creating
// a Trans without a HIP device throws.
//
// Ownership: objects own their C handles (move-only).  Spectral and grid-point arrays stay the caller's, exactly as
// with TransLocal; plain pointers are HOST pointers (synchronous), the *_device members take device pointers and are
// asynchronous on the object's HIP stream.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "atlas_amd.h"

namespace atlas_amd {

class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string& what) : std::runtime_error(what) {}
};
class NotImplemented : public Exception {
public:
    explicit NotImplemented(const std::string& what) : Exception(what) {}
};

namespace detail {
[[noreturn]] inline void raise() {
    const char* m = atlas_amd__last_error();
    const std::string msg(m ? m : "atlas_amd: unknown error");
    if (msg.rfind("Not implemented", 0) == 0) {
        throw NotImplemented(msg);
    }
    throw Exception(msg);
}
inline void check(int rc) {
    if (rc != 0) {
        raise();
    }
}
template <typename T>
struct dtype_code;
template <>
struct dtype_code<int> {
    static constexpr int value = 0;
};
template <>
struct dtype_code<long> {
    static constexpr int value = 1;
};
template <>
struct dtype_code<float> {
    static constexpr int value = 2;
};
template <>
struct dtype_code<double> {
    static constexpr int value = 3;
};
}  // namespace detail

inline int device_count() {
    return atlas_amd__device_count();
}

// ---------------------------------------------------------------------------------------------------------------------
// key=value configuration (stands in for eckit::Configuration / atlas::util::Config on this path)
class Config {
public:
    Config() = default;
    Config(const std::string& key, const std::string& value) { set(key, value); }
    Config(const std::string& key, long value) { set(key, value); }
    Config& set(const std::string& key, const std::string& value) {
        text_ += (text_.empty() ? "" : ";") + key + "=" + value;
        return *this;
    }
    Config& set(const std::string& key, long value) { return set(key, std::to_string(value)); }
    Config operator|(const Config& other) const {
        Config c(*this);
        if (!other.text_.empty()) {
            c.text_ += (c.text_.empty() ? "" : ";") + other.text_;
        }
        return c;
    }
    const std::string& str() const { return text_; }

private:
    std::string text_;
};
// atlas::RectangularDomain({west, east}, {south, north}) in degrees
struct RectangularDomain {
    double west, east, south, north;
};
namespace option {
inline Config type(const std::string& t) {  // atlas::option::type
    return Config("type", t);
}
inline Config halo(long h) {  // atlas::option::halo
    return Config("halo", h);
}
}  // namespace option

// ---------------------------------------------------------------------------------------------------------------------
class StructuredGrid {
public:
    // "F<N>" / "O<N>" Gaussian grids (Atlas: Grid("O1280"))
    explicit StructuredGrid(const std::string& name) : h_(atlas_amd__Grid__new_gaussian(name.c_str())) {
        if (!h_) {
            detail::raise();
        }
    }
    // any global structured grid: points per latitude and latitudes in degrees, north to south
    StructuredGrid(const std::vector<int>& nx, const std::vector<double>& y) {
        if (nx.size() != y.size()) {
            throw Exception("StructuredGrid: nx and y differ in length");
        }
        h_ = atlas_amd__Grid__new_structured(int(nx.size()), nx.data(), y.data());
        if (!h_) {
            detail::raise();
        }
    }
    StructuredGrid(StructuredGrid&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    StructuredGrid& operator=(StructuredGrid&& o) noexcept {
        if (this != &o) {
            reset();
            h_ = std::exchange(o.h_, nullptr);
        }
        return *this;
    }
    StructuredGrid(const StructuredGrid&)            = delete;
    StructuredGrid& operator=(const StructuredGrid&) = delete;
    ~StructuredGrid() { reset(); }

    int ny() const { return atlas_amd__Grid__ny(h_); }
    int nxmax() const { return atlas_amd__Grid__nxmax(h_); }
    std::int64_t size() const { return atlas_amd__Grid__size(h_); }
    bool regular() const { return atlas_amd__Grid__regular(h_) != 0; }
    std::vector<int> nx() const {
        std::vector<int> v(ny());
        detail::check(atlas_amd__Grid__nx(h_, v.data()));
        return v;
    }
    std::vector<double> y() const {
        std::vector<double> v(ny());
        detail::check(atlas_amd__Grid__y(h_, v.data()));
        return v;
    }
    // longitude of point i in a row of n points: global grids start at 0 (Structured.h:308-314)
    static double x(int i, int n) { return 360.0 * i / n; }
    const atlas_amd_Grid* handle() const { return h_; }

private:
    void reset() {
        if (h_) {
            atlas_amd__Grid__delete(h_);
            h_ = nullptr;
        }
    }
    atlas_amd_Grid* h_ = nullptr;
};

inline std::vector<double> gaussian_latitudes_npole_spole(int N) {
    std::vector<double> lats(2 * size_t(N));
    detail::check(atlas_amd__gaussian_latitudes_npole_spole(N, lats.data()));
    return lats;
}

// ---------------------------------------------------------------------------------------------------------------------
namespace trans {

// trans::Trans for a regular longitude-latitude target that is not a crop of a global grid (TransLocal's no_nest branch,
// TransLocal.cc:394-406): latitudes in degrees in row order, longitudes west + i * dlon.  Scalar fields.
class RegionalTrans {
public:
    RegionalTrans(int nlon, double west, double dlon, const std::vector<double>& lats, int truncation) {
        h_ = atlas_amd__RegionalTrans__new(nlon, west, dlon, int(lats.size()), lats.data(), truncation);
        if (!h_) {
            detail::raise();
        }
    }
    RegionalTrans(const RegionalTrans&)            = delete;
    RegionalTrans& operator=(const RegionalTrans&) = delete;
    ~RegionalTrans() { atlas_amd__RegionalTrans__delete(h_); }
    size_t nb_gridpoints() const { return size_t(atlas_amd__RegionalTrans__nb_gridpoints(h_)); }
    // gp_fields[lon + nlon * (lat + nlat * field)]
    void invtrans(int nb_scalar_fields, const double scalar_spectra[], double gp_fields[]) const {
        detail::check(atlas_amd__RegionalTrans__invtrans_scalar(h_, nb_scalar_fields, scalar_spectra, gp_fields));
    }
    void invtrans(int nb_scalar_fields, const double scalar_spectra[], int nb_vordiv_fields, const double vorticity_spectra[],
                  const double divergence_spectra[], double gp_fields[]) const {
        detail::check(atlas_amd__RegionalTrans__invtrans_vordiv(h_, nb_scalar_fields, scalar_spectra, nb_vordiv_fields,
                                                                vorticity_spectra, divergence_spectra, gp_fields));
    }

private:
    atlas_amd_RegionalTrans* h_ = nullptr;
};

class Trans {
public:
    static bool hasBackend(const std::string& backend) { return atlas_amd__Trans__has_backend(backend.c_str()) != 0; }
    static void backend(const std::string& backend) { detail::check(atlas_amd__Trans__set_backend(backend.c_str())); }
    static std::string backend() {
        char* s  = nullptr;
        size_t n = 0;
        detail::check(atlas_amd__Trans__backend(&s, &n));
        std::string out(s, n);
        std::free(s);
        return out;
    }

    Trans(const StructuredGrid& grid, int truncation, const Config& config = Config()) {
        h_ = atlas_amd__Trans__new_config(grid.handle(), truncation, config.str().c_str(), nullptr, 0);
        if (!h_) {
            detail::raise();
        }
    }
    // trans::Trans(global_grid, domain, truncation, config) (Trans.h, TransLocal.cc:394-470): the rows and the longitude
    // window of every row that the domain contains; the output holds those points row by row
    Trans(const StructuredGrid& grid, const RectangularDomain& domain, int truncation, const Config& config = Config()) {
        char text[160];
        std::snprintf(text, sizeof(text), "%.17g,%.17g,%.17g,%.17g", domain.west, domain.east, domain.south, domain.north);
        const Config c = config | Config("domain", std::string(text));
        h_             = atlas_amd__Trans__new_config(grid.handle(), truncation, c.str().c_str(), nullptr, 0);
        if (!h_) {
            detail::raise();
        }
    }
    // with a Legendre cache blob in TransLocal's file layout (trans/Cache.h:98-136)
    Trans(const void* legendre_cache, size_t legendre_cache_size, const StructuredGrid& grid, int truncation,
          const Config& config = Config()) {
        h_ = atlas_amd__Trans__new_config(grid.handle(), truncation, config.str().c_str(), legendre_cache,
                                          legendre_cache_size);
        if (!h_) {
            detail::raise();
        }
    }
    Trans(Trans&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Trans& operator=(Trans&& o) noexcept {
        if (this != &o) {
            reset();
            h_ = std::exchange(o.h_, nullptr);
        }
        return *this;
    }
    Trans(const Trans&)            = delete;
    Trans& operator=(const Trans&) = delete;
    ~Trans() { reset(); }

    int truncation() const { return atlas_amd__Trans__truncation(h_); }
    size_t nb_spectral_coefficients() const { return size_t(atlas_amd__Trans__nb_spectral_coefficients(h_)); }
    size_t nb_spectral_coefficients_global() const { return nb_spectral_coefficients(); }
    size_t nb_gridpoints() const { return size_t(atlas_amd__Trans__nb_gridpoints(h_)); }
    size_t nb_gridpoints_global() const { return size_t(atlas_amd__Trans__nb_gridpoints_global(h_)); }

    // ---- inverse transforms, host arrays (TransImpl.h:116-141) ----
    void invtrans(int nb_scalar_fields, const double scalar_spectra[], int nb_vordiv_fields,
                  const double vorticity_spectra[], const double divergence_spectra[], double gp_fields[]) const {
        detail::check(atlas_amd__Trans__invtrans(h_, nb_scalar_fields, scalar_spectra, nb_vordiv_fields,
                                                 vorticity_spectra, divergence_spectra, gp_fields));
    }
    void invtrans(int nb_scalar_fields, const double scalar_spectra[], double gp_fields[]) const {
        detail::check(atlas_amd__Trans__invtrans_scalar(h_, nb_scalar_fields, scalar_spectra, gp_fields));
    }
    void invtrans(int nb_vordiv_fields, const double vorticity_spectra[], const double divergence_spectra[],
                  double gp_fields[]) const {
        detail::check(atlas_amd__Trans__invtrans_vordiv2wind(h_, nb_vordiv_fields, vorticity_spectra,
                                                             divergence_spectra, gp_fields));
    }
    // fp32 variant (extension; TransLocal is double only)
    void invtrans(int nb_scalar_fields, const float scalar_spectra[], float gp_fields[]) const {
        detail::check(atlas_amd__Trans__invtrans_scalar_f32(h_, nb_scalar_fields, scalar_spectra, gp_fields));
    }
    // ---- the same on device arrays, asynchronous on stream() ----
    void invtrans_device(int nb_scalar_fields, const double* scalar_spectra, int nb_vordiv_fields,
                         const double* vorticity_spectra, const double* divergence_spectra, double* gp_fields) const {
        detail::check(atlas_amd__Trans__invtrans_device(h_, nb_scalar_fields, scalar_spectra, nb_vordiv_fields,
                                                        vorticity_spectra, divergence_spectra, gp_fields));
    }
    void invtrans_device(int nb_scalar_fields, const double* scalar_spectra, double* gp_fields) const {
        detail::check(atlas_amd__Trans__invtrans_scalar_device(h_, nb_scalar_fields, scalar_spectra, gp_fields));
    }
    void invtrans_device(int nb_scalar_fields, const float* scalar_spectra, float* gp_fields) const {
        detail::check(atlas_amd__Trans__invtrans_scalar_device_f32(h_, nb_scalar_fields, scalar_spectra, gp_fields));
    }
    // the vor/div call of the fp32 variant [r5]
    void invtrans_device(int nb_scalar_fields, const float* scalar_spectra, int nb_vordiv_fields, const float* vorticity_spectra,
                         const float* divergence_spectra, float* gp_fields) const {
        detail::check(atlas_amd__Trans__invtrans_device_f32(h_, nb_scalar_fields, scalar_spectra, nb_vordiv_fields,
                                                            vorticity_spectra, divergence_spectra, gp_fields));
    }
    // HIP-event stage times (profile=1): {legendre_ms, legendre_calls, fourier_ms, fourier_calls}, and for vor/div calls the
    // spectra_prepare stage {prepare_ms, prepare_calls}
    void timings(double out[4], bool reset = false) const { detail::check(atlas_amd__Trans__timings(h_, out, reset ? 1 : 0)); }
    void timings_vordiv(double out[2], bool reset = false) const {
        detail::check(atlas_amd__Trans__timings_vordiv(h_, out, reset ? 1 : 0));
    }
    // ---- not implemented by TransLocal: these throw NotImplemented ----
    void dirtrans(int nb_fields, const double scalar_fields[], double scalar_spectra[]) const {
        detail::check(atlas_amd__Trans__dirtrans_scalar(h_, nb_fields, scalar_fields, scalar_spectra));
    }
    void dirtrans(int nb_fields, const double wind_fields[], double vorticity_spectra[],
                  double divergence_spectra[]) const {
        detail::check(
            atlas_amd__Trans__dirtrans_wind2vordiv(h_, nb_fields, wind_fields, vorticity_spectra, divergence_spectra));
    }
    void invtrans_adj(int nb_scalar_fields, const double gp_fields[], double scalar_spectra[]) const {
        detail::check(atlas_amd__Trans__invtrans_adj_scalar(h_, nb_scalar_fields, gp_fields, scalar_spectra));
    }
    void invtrans_adj(int nb_scalar_fields, const double gp_fields[], int nb_vordiv_fields, double vorticity_spectra[],
                      double divergence_spectra[], double scalar_spectra[]) const {
        detail::check(atlas_amd__Trans__invtrans_adj(h_, nb_scalar_fields, gp_fields, nb_vordiv_fields,
                                                     vorticity_spectra, divergence_spectra, scalar_spectra));
    }

    // Config("shard", "mirror"): this object transforms rows [first, second) of the grid and their mirror images
    // [ny - second, ny - first); the output holds the northern rows, then the southern ones
    std::pair<int, int> mirror_rows() const {
        int b[2];
        detail::check(atlas_amd__Trans__mirror_rows(h_, b));
        return {b[0], b[1]};
    }

    // Legendre cache blob in TransLocal's file layout (what LegendreCacheCreatorLocal::create() writes)
    std::vector<char> legendre_cache() const {
        std::vector<char> blob(atlas_amd__Trans__legendre_cache_size(h_));
        detail::check(atlas_amd__Trans__legendre_cache_export(h_, blob.data(), blob.size()));
        return blob;
    }

    void* stream() const { return atlas_amd__Trans__stream(h_); }  // hipStream_t
    void set_stream(void* hip_stream) const { detail::check(atlas_amd__Trans__set_stream(h_, hip_stream)); }
    void synchronize() const { detail::check(atlas_amd__Trans__synchronize(h_)); }
    atlas_amd_Trans* handle() const { return h_; }

private:
    void reset() {
        if (h_) {
            atlas_amd__Trans__delete(h_);
            h_ = nullptr;
        }
    }
    atlas_amd_Trans* h_ = nullptr;
};

class VorDivToUV {
public:
    explicit VorDivToUV(int truncation, const Config& = Config()) : truncation_(truncation) {}
    int truncation() const { return truncation_; }
    // spectral vorticity / divergence -> spectral U = u cos(lat), V = v cos(lat); host arrays of nb_coeff*nb_fields
    void execute(int nb_coeff, int nb_fields, const double vorticity[], const double divergence[], double U[],
                 double V[]) const {
        detail::check(atlas_amd__VorDivToUV__execute(truncation_, nb_coeff, nb_fields, vorticity, divergence, U, V));
    }
    void execute_device(int nb_coeff, int nb_fields, const double* vorticity, const double* divergence, double* U,
                        double* V, void* hip_stream = nullptr) const {
        detail::check(atlas_amd__VorDivToUV__execute_device(truncation_, nb_coeff, nb_fields, vorticity, divergence, U,
                                                            V, hip_stream));
    }

private:
    int truncation_;
};

}  // namespace trans

// ---------------------------------------------------------------------------------------------------------------------
namespace parallel {

class HaloExchange {
public:
    HaloExchange() : h_(atlas_amd__HaloExchange__new()) {
        if (!h_) {
            detail::raise();
        }
    }
    HaloExchange(HaloExchange&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    HaloExchange(const HaloExchange&)            = delete;
    HaloExchange& operator=(const HaloExchange&) = delete;
    ~HaloExchange() {
        if (h_) {
            atlas_amd__HaloExchange__delete(h_);
        }
    }

    // HaloExchange::setup(part, remote_idx, base, parsize[, halo_begin])  (HaloExchange.cc:66-72), one process
    void setup(const int part[], const int remote_idx[], int base, int parsize, int halo_begin = 0) {
        detail::check(atlas_amd__HaloExchange__setup_halo_begin(h_, part, remote_idx, base, parsize, halo_begin));
    }
    // several processes: the local phases around the caller's allToAll / allToAllv (HaloExchange.cc:118,156-159)
    void setup_begin(int nproc, int myproc, const int part[], const int remote_idx[], int base, int parsize,
                     int halo_begin = 0) {
        detail::check(
            atlas_amd__HaloExchange__setup_begin(h_, nproc, myproc, part, remote_idx, base, parsize, halo_begin));
    }
    void setup_finish(const int sendcounts[], const int recv_requests[]) {
        detail::check(atlas_amd__HaloExchange__setup_finish(h_, sendcounts, recv_requests));
    }
    int nproc() const { return atlas_amd__HaloExchange__nproc(h_); }
    int sendcnt() const { return atlas_amd__HaloExchange__sendcnt(h_); }
    int recvcnt() const { return atlas_amd__HaloExchange__recvcnt(h_); }
    // "sendcounts" | "recvcounts" | "senddispls" | "recvdispls" | "sendmap" | "recvmap" | "send_requests"
    std::vector<int> get(const std::string& what) const {
        const bool per_proc = what.find("counts") != std::string::npos || what.find("displs") != std::string::npos;
        std::vector<int> v(per_proc ? nproc() : what == "sendmap" ? sendcnt() : recvcnt());
        detail::check(atlas_amd__HaloExchange__get(h_, what.c_str(), v.data()));
        return v;
    }

    // execute<T>(field, var_strides, var_shape, var_rank): the strided form the Fortran interface uses
    // (HaloExchange.h:433-440); host arrays, one process
    template <typename T>
    void execute(T field[], const int var_strides[], const int var_shape[], int var_rank) const {
        strided<T>(field, var_strides, var_shape, var_rank, false);
    }
    template <typename T>
    void execute_adjoint(T field[], const int var_strides[], const int var_shape[], int var_rank) const {
        strided<T>(field, var_strides, var_shape, var_rank, true);
    }
    // general form (HaloExchange.h:151-290): op 0 execute, 1 adjoint, 2 pack, 3 unpack, 4 pack_adjoint,
    // 5 unpack_adjoint, 6 zero_halos; on_device: device pointers, asynchronous on stream()
    template <typename T>
    void field_op(int op, T* field, int rank, const int shape[], const long long strides[], int parallel_dim,
                  T* buffer, bool on_device) const {
        detail::check(atlas_amd__HaloExchange__field_op(h_, op, detail::dtype_code<T>::value, field, rank, shape,
                                                        strides, parallel_dim, buffer, on_device ? 1 : 0));
    }
    void synchronize() const { detail::check(atlas_amd__HaloExchange__synchronize(h_)); }
    atlas_amd_HaloExchange* handle() const { return h_; }

private:
    template <typename T>
    void strided(T field[], const int var_strides[], const int var_shape[], int var_rank, bool adjoint) const;
    atlas_amd_HaloExchange* h_ = nullptr;
};

#define ATLAS_AMD_HALO_STRIDED(T, NAME)                                                                               \
    template <>                                                                                                       \
    inline void HaloExchange::strided<T>(T field[], const int var_strides[], const int var_shape[], int var_rank,     \
                                         bool adjoint) const {                                                        \
        detail::check(adjoint ? atlas_amd__HaloExchange__execute_adjoint_strided_##NAME(h_, field, var_strides,       \
                                                                                       var_shape, var_rank)           \
                              : atlas_amd__HaloExchange__execute_strided_##NAME(h_, field, var_strides, var_shape,    \
                                                                               var_rank));                            \
    }
ATLAS_AMD_HALO_STRIDED(int, int)
ATLAS_AMD_HALO_STRIDED(long, long)
ATLAS_AMD_HALO_STRIDED(float, float)
ATLAS_AMD_HALO_STRIDED(double, double)
#undef ATLAS_AMD_HALO_STRIDED

}  // namespace parallel

// ---------------------------------------------------------------------------------------------------------------------
namespace functionspace {

class StructuredColumns {
public:
    // blocksize: 1 "equal_bands", nx "regular_bands", 0 "row_bands" (BandsDistribution.h:32-34)
    StructuredColumns(const StructuredGrid& grid, int halo, bool periodic_points = false, int nparts = 1, int part = 0,
                      int blocksize = 1) :
        nparts_(nparts), part_(part) {
        h_ = atlas_amd__StructuredColumns__new(grid.handle(), halo, periodic_points ? 1 : 0, nparts, part, blocksize);
        if (!h_) {
            detail::raise();
        }
    }
    // explicit grid::Distribution: partition[g] of every grid point
    StructuredColumns(const StructuredGrid& grid, int halo, bool periodic_points, int nparts, int part,
                      const std::vector<int>& partition) :
        nparts_(nparts), part_(part) {
        h_ = atlas_amd__StructuredColumns__new_distribution(grid.handle(), halo, periodic_points ? 1 : 0, nparts, part,
                                                            partition.data(), (long long)partition.size());
        if (!h_) {
            detail::raise();
        }
    }
    StructuredColumns(StructuredColumns&& o) noexcept :
        h_(std::exchange(o.h_, nullptr)), nparts_(o.nparts_), part_(o.part_) {}
    StructuredColumns(const StructuredColumns&)            = delete;
    StructuredColumns& operator=(const StructuredColumns&) = delete;
    ~StructuredColumns() {
        if (h_) {
            atlas_amd__StructuredColumns__delete(h_);
        }
    }

    int sizeOwned() const { return atlas_amd__StructuredColumns__size_owned(h_); }
    int sizeHalo() const { return atlas_amd__StructuredColumns__size_halo(h_); }
    int size() const { return sizeHalo(); }
    int j_begin() const { return bounds()[0]; }
    int j_end() const { return bounds()[1]; }
    int j_begin_halo() const { return bounds()[2]; }
    int j_end_halo() const { return bounds()[3]; }
    int i_begin(int j) const { return row_bounds(j)[0]; }
    int i_end(int j) const { return row_bounds(j)[1]; }
    int i_begin_halo(int j) const { return row_bounds(j)[2]; }
    int i_end_halo(int j) const { return row_bounds(j)[3]; }
    int index(int i, int j) const {
        int n = -1;
        detail::check(atlas_amd__StructuredColumns__index(h_, i, j, &n));
        return n;
    }
    // "partition" | "ghost" | "index_i" | "index_j" | "remote_idx"   (0-based)
    std::vector<int> field(const std::string& what) const {
        std::vector<int> v(sizeHalo());
        detail::check(atlas_amd__StructuredColumns__get_int(h_, what.c_str(), v.data()));
        return v;
    }
    std::vector<std::int64_t> global_index() const {  // 1-based
        std::vector<std::int64_t> v(sizeHalo());
        detail::check(atlas_amd__StructuredColumns__global_index(h_, v.data()));
        return v;
    }
    std::vector<double> xy() const {  // [n][2]
        std::vector<double> v(2 * size_t(sizeHalo()));
        detail::check(atlas_amd__StructuredColumns__xy(h_, v.data()));
        return v;
    }
    // HaloExchange::setup(partition, remote_index, base, sizeHalo, sizeOwned)  (StructuredColumns.cc:145-148)
    void setup_halo_exchange(parallel::HaloExchange& hx) const {
        detail::check(atlas_amd__StructuredColumns__setup_halo_exchange(h_, hx.handle(), nparts_, part_));
    }
    atlas_amd_StructuredColumns* handle() const { return h_; }

private:
    struct Four {
        int v[4];
        int operator[](int i) const { return v[i]; }
    };
    Four bounds() const {
        Four b;
        detail::check(atlas_amd__StructuredColumns__bounds(h_, b.v));
        return b;
    }
    Four row_bounds(int j) const {
        Four b;
        detail::check(atlas_amd__StructuredColumns__row_bounds(h_, j, b.v));
        return b;
    }
    atlas_amd_StructuredColumns* h_ = nullptr;
    int nparts_, part_;
};

// functionspace::NodeColumns, the halo-exchange contract of it (src/atlas/functionspace/NodeColumns.cc:101-113: the halo exchange is
// set up from mesh.nodes().partition(), mesh.nodes().remote_index() (base REMOTE_IDX_BASE) and the node count including the halo --
// WITHOUT halo_begin: every node is tested for ghost-ness; :357-459: haloExchange / adjointHaloExchange of a Field of rank 1 .. 4 in
// int / long / float / double, any other rank is "Rank not supported").  Atlas builds the three arrays from a Mesh (out of this
// library's scope); here the caller hands them over, as the adapter does.
class NodeColumns {
public:
    static constexpr int REMOTE_IDX_BASE = 0;   // C++ value; the Fortran interface uses 1
    NodeColumns(const int partition[], const int remote_index[], int nb_nodes, int remote_idx_base = REMOTE_IDX_BASE)
        : nb_nodes_(nb_nodes) {
        hx_.setup(partition, remote_index, remote_idx_base, nb_nodes);   // no halo_begin (NodeColumns.cc:110-111)
    }
    // several processes: the caller's allToAll / allToAllv between the two phases (HaloExchange.cc:118,156-159)
    NodeColumns(int nproc, int myproc, const int partition[], const int remote_index[], int nb_nodes,
                int remote_idx_base = REMOTE_IDX_BASE)
        : nb_nodes_(nb_nodes) {
        hx_.setup_begin(nproc, myproc, partition, remote_index, remote_idx_base, nb_nodes);
    }
    void setup_finish(const int sendcounts[], const int recv_requests[]) { hx_.setup_finish(sendcounts, recv_requests); }
    int nb_nodes() const { return nb_nodes_; }
    const parallel::HaloExchange& halo_exchange() const { return hx_; }
    parallel::HaloExchange& halo_exchange() { return hx_; }

    // field[nb_nodes][shape[1]]..[shape[rank-1]], C order, nodes first (the layout of NodeColumns fields); host or device memory
    template <typename T>
    void haloExchange(T* field, int rank, const int shape[], bool on_device = false) const {
        exchange<T>(0, field, rank, shape, on_device);
    }
    template <typename T>
    void adjointHaloExchange(T* field, int rank, const int shape[], bool on_device = false) const {
        exchange<T>(1, field, rank, shape, on_device);
    }

private:
    template <typename T>
    void exchange(int op, T* field, int rank, const int shape[], bool on_device) const {
        if (rank < 1 || rank > 4) {
            throw Exception("Rank not supported");   // NodeColumns.cc:417,439
        }
        if (shape[0] != nb_nodes_) {
            throw Exception("NodeColumns::haloExchange: the first dimension of the field must be the node count");
        }
        long long strides[4];
        long long st = 1;
        for (int d = rank - 1; d >= 0; --d) {
            strides[d] = st;
            st *= shape[d];
        }
        hx_.field_op<T>(op, field, rank, shape, strides, 0, static_cast<T*>(nullptr), on_device);
        if (!on_device) {
            hx_.synchronize();
        }
    }
    parallel::HaloExchange hx_;
    int nb_nodes_;
};

}  // namespace functionspace
}  // namespace atlas_amd
