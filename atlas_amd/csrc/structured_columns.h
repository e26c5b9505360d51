// Halo index construction of atlas::functionspace::StructuredColumns for GLOBAL structured grids with band
// distributions (host logic; the exchange itself is parallel::HaloExchange).
// Reference: src/atlas/functionspace/detail/StructuredColumns_setup.cc:88-663,
//            StructuredColumns_create_remote_index.cc:37-255, StructuredColumns.cc:104-152 (halo_exchange setup),
//            :732-808 (FixupHaloForVectors), src/atlas/grid/detail/distribution/BandsDistribution.h:32-34.
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "gaussian.h"

namespace atlas_amd {
namespace functionspace {

struct StructuredColumnsConfig {
    int halo             = 0;
    bool periodic_points = false;
    int nparts           = 1;
    int part             = 0;
    int blocksize        = 1;  // bands distribution: 1 = "equal_bands", nx = "regular_bands", 0 = "row_bands" (whole rows)
    // explicit grid::Distribution (partition of every grid point in global order), as Atlas hands one over for its
    // other partitioners (equal_regions, checkerboard, ...); empty: the bands rule above.  As in the reference, the
    // points a part owns in a row must form one contiguous i-range (StructuredColumns_setup.cc:125-226)
    std::vector<int> distribution;
};

class StructuredColumns {
public:
    StructuredColumns(const grid::StructuredGrid& g, const StructuredColumnsConfig& cfg);

    int size_owned() const { return size_owned_; }   // sizeOwned()
    int size_halo() const { return size_halo_; }     // sizeHalo()
    int halo() const { return cfg_.halo; }
    int ny() const { return ny_; }
    int j_begin() const { return j_begin_; }
    int j_end() const { return j_end_; }
    int j_begin_halo() const { return j_begin_halo_; }
    int j_end_halo() const { return j_end_halo_; }
    int i_begin(int j) const { return i_begin_[j]; }
    int i_end(int j) const { return i_end_[j]; }
    int i_begin_halo(int j) const { return i_begin_halo_[j - j_begin_halo_]; }
    int i_end_halo(int j) const { return i_end_halo_[j - j_begin_halo_]; }
    int index(int i, int j) const;                   // StructuredColumns::index(i,j); throws if outside

    const std::vector<int>& partition() const { return partition_; }
    const std::vector<int>& ghost() const { return ghost_; }
    const std::vector<int64_t>& global_index() const { return glb_idx_; }   // 1-based
    const std::vector<int>& index_i() const { return index_i_; }            // 0-based
    const std::vector<int>& index_j() const { return index_j_; }
    const std::vector<double>& xy() const { return xy_; }                   // [size_halo][2]
    const std::vector<int>& remote_index() const { return remote_idx_; }    // base 0
    // nodes of the halo rows beyond the poles (j < 0 || j >= ny): FixupHaloForVectors
    std::vector<int> pole_row_nodes() const;

    int partition_of(int64_t g) const;  // BandsDistribution::function
    int64_t first_global_index(int part) const { return first_of_part_[part]; }

private:
    int compute_j(int j) const;
    int compute_i(int i, int j) const;
    double gx(int i, int j) const { return 0.0 + double(i) * (360.0 / double(nx_[j])); }
    double compute_x(int i, int j) const;
    double compute_x_fast(int i, int jj, int nx) const;
    double compute_y(int j) const;
    int64_t compute_g(int i, int j) const;

    StructuredColumnsConfig cfg_;
    std::vector<int> nx_;
    std::vector<double> y_;
    int ny_ = 0;
    int64_t npts_ = 0;
    std::vector<int64_t> offsets_;
    std::vector<int64_t> first_of_part_;
    int size_owned_ = 0, size_halo_ = 0;
    int j_begin_ = 0, j_end_ = 0, j_begin_halo_ = 0, j_end_halo_ = 0;
    std::vector<int> i_begin_, i_end_, i_begin_halo_, i_end_halo_;
    std::vector<int> partition_, ghost_, index_i_, index_j_, remote_idx_;
    std::vector<int64_t> glb_idx_;
    std::vector<double> xy_;
    // ij -> point: per halo row, offset of the row's first stored i plus a dense table
    std::vector<int> row_imin_, row_start_;
    std::vector<int> ij_table_;
};

}  // namespace functionspace
}  // namespace atlas_amd
