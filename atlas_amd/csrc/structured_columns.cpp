// See structured_columns.h.  Host-only.
#include "structured_columns.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>

namespace atlas_amd {
namespace functionspace {

int StructuredColumns::partition_of(int64_t g) const {
    if (!cfg_.distribution.empty()) {
        return cfg_.distribution[(size_t)g];  // distribution.partition(c), StructuredColumns_setup.cc:141
    }
    // part(g) = ((g / blocksize) * nparts) / nb_blocks   (BandsDistribution.h:32-34)
    if (cfg_.blocksize == 0) {
        // "row_bands": whole latitude rows, a row belongs to the equal_bands part of its FIRST point -- the output
        // decomposition of the multi-GPU transform (trans_plan.cpp: latitude_bands).  For Atlas this is a user-supplied
        // grid::Distribution (array of partitions), which StructuredColumns accepts like any other.
        const int j = int(std::upper_bound(offsets_.begin(), offsets_.end(), g) - offsets_.begin()) - 1;
        return int((offsets_[j] * cfg_.nparts) / npts_);
    }
    const int64_t bs        = cfg_.blocksize;
    const int64_t nb_blocks = (npts_ + bs - 1) / bs;
    return int(((g / bs) * cfg_.nparts) / nb_blocks);
}

int StructuredColumns::compute_j(int j) const {  // StructuredColumns_setup.cc:263-287
    if (j < 0) {
        j = (y_[0] == 90.) ? -j : -j - 1;
    }
    else if (j >= ny_) {
        const int jlast = ny_ - 1;
        j               = (y_[jlast] == -90.) ? jlast - 1 - (j - ny_) : jlast - (j - ny_);
    }
    if (j < 0 || j >= ny_) {
        j = compute_j(j);
    }
    return j;
}

int StructuredColumns::compute_i(int i, int j) const {  // :242-251
    const int nx = nx_[j];
    while (i >= nx) {
        i -= nx;
    }
    while (i < 0) {
        i += nx;
    }
    return i;
}

double StructuredColumns::compute_x(int i, int j) const {  // :290-297
    const int jj   = compute_j(j);
    const int ii   = compute_i(i, jj);
    const int nx   = nx_[jj];
    const double a = (ii - i) / nx;  // integer division, as in the reference
    return gx(ii, jj) - a * gx(nx, jj);
}

double StructuredColumns::compute_x_fast(int i, int jj, int nx) const {  // :299-304
    int ii = i;
    while (ii >= nx) {
        ii -= nx;
    }
    while (ii < 0) {
        ii += nx;
    }
    const double a = (ii - i) / nx;
    return gx(ii, jj) - a * (gx(nx, jj) - gx(0, jj));
}

double StructuredColumns::compute_y(int j) const {  // :312-324
    const int jj = compute_j(j);
    return (j < 0) ? 90. + (90. - y_[jj]) : (j >= ny_) ? -90. + (-90. - y_[jj]) : y_[jj];
}

int64_t StructuredColumns::compute_g(int i, int j) const {  // :331-355
    const int jj = compute_j(j);
    int ii       = compute_i(i, jj);
    if (jj != j) {  // across a pole: shift by 180 degrees
        const int nx = nx_[jj];
        if (nx % 2 == 0) {
            ii = (ii < nx / 2) ? ii + nx / 2 : ii - nx / 2;
        }
        else {
            if (ii < nx / 2 + 1) {
                ii += nx / 2 + 1;
            }
            else {
                ii -= nx / 2 + 1;
            }
        }
    }
    return offsets_[jj] + ii + 1;
}

StructuredColumns::StructuredColumns(const grid::StructuredGrid& g, const StructuredColumnsConfig& cfg): cfg_(cfg) {
    if (cfg.halo < 0 || cfg.nparts < 1 || cfg.part < 0 || cfg.part >= cfg.nparts || cfg.blocksize < 0) {
        throw std::invalid_argument("StructuredColumns: bad configuration");
    }
    nx_ = g.nx;
    y_  = g.y;
    ny_ = g.ny();
    offsets_.assign(ny_ + 1, 0);
    for (int j = 0; j < ny_; ++j) {
        offsets_[j + 1] = offsets_[j] + nx_[j];
    }
    npts_ = offsets_[ny_];
    if (!cfg.distribution.empty()) {
        if ((int64_t)cfg.distribution.size() != npts_) {
            throw std::invalid_argument("StructuredColumns: distribution must hold one partition per grid point");
        }
        for (int p : cfg.distribution) {
            if (p < 0 || p >= cfg.nparts) {
                throw std::invalid_argument("StructuredColumns: distribution entry outside [0, nparts)");
            }
        }
    }
    if (npts_ > std::numeric_limits<int>::max()) {
        throw std::invalid_argument("StructuredColumns: grid too large for 32-bit local indices");
    }
    const double eps = 1.e-12;
    const int halo   = cfg.halo;
    const int BIG    = std::numeric_limits<int>::max();

    // ---- owned bounds (:125-226)
    i_begin_.assign(ny_, BIG);
    i_end_.assign(ny_, std::numeric_limits<int>::min());
    int owned = 0;
    first_of_part_.assign(cfg.nparts, -1);
    if (cfg.nparts == 1) {
        j_begin_ = 0;
        j_end_   = ny_;
        for (int j = 0; j < ny_; ++j) {
            i_begin_[j] = 0;
            i_end_[j]   = nx_[j];
        }
        owned             = (int)npts_;
        first_of_part_[0] = 0;
    }
    else {
        j_begin_  = BIG / 2;
        j_end_    = -(BIG / 2);
        int64_t c = 0;
        for (int j = 0; j < ny_; ++j) {
            for (int i = 0; i < nx_[j]; ++i, ++c) {
                const int p = partition_of(c);
                if (first_of_part_[p] < 0) {
                    first_of_part_[p] = c;
                }
                if (p == cfg.part) {
                    j_begin_    = std::min(j_begin_, j);
                    j_end_      = std::max(j_end_, j + 1);
                    i_begin_[j] = std::min(i_begin_[j], i);
                    i_end_[j]   = std::max(i_end_[j], i + 1);
                    ++owned;
                }
            }
        }
        if (owned == 0) {
            throw std::invalid_argument("StructuredColumns: partition owns no points");
        }
        // The construction below (like the reference's, StructuredColumns_setup.cc:125-226) describes the owned region
        // by one row range and one i-range per row: every row of [j_begin, j_end) must hold owned points and they must
        // be contiguous.  Reject anything else instead of building halos around empty rows.
        int64_t in_ranges = 0;
        for (int j = j_begin_; j < j_end_; ++j) {
            if (i_end_[j] <= i_begin_[j]) {
                throw std::invalid_argument("StructuredColumns: the rows a partition owns must be contiguous (row " +
                                            std::to_string(j) + " inside its row range holds none of its points)");
            }
            in_ranges += i_end_[j] - i_begin_[j];
        }
        if (in_ranges != owned) {
            throw std::invalid_argument(
                "StructuredColumns: the points a partition owns in one row must form one contiguous i-range");
        }
    }
    size_owned_   = owned;
    j_begin_halo_ = j_begin_ - halo;
    j_end_halo_   = j_end_ + halo;
    const int nrows = j_end_halo_ - j_begin_halo_;
    i_begin_halo_.assign(nrows, BIG);
    i_end_halo_.assign(nrows, -BIG);
    auto ibh = [&](int j) -> int& { return i_begin_halo_[j - j_begin_halo_]; };
    auto ieh = [&](int j) -> int& { return i_end_halo_[j - j_begin_halo_]; };

    // ---- halo bounds (:369-455)
    for (int j = j_begin_; j < j_end_; ++j) {
        const int ends[2] = {i_begin_[j], i_end_[j] - 1};
        for (int e = 0; e < 2; ++e) {
            int i = ends[e];
            if (cfg.periodic_points && i == nx_[j] - 1) {
                ++i;  // the periodic point east of the last column (:379-381)
            }
            const double x = gx(i, j), x_next = gx(i + 1, j), x_prev = gx(i - 1, j);
            for (int jj = j - halo; jj <= j + halo; ++jj) {
                const int jjj = compute_j(jj);
                const int nxj = nx_[jjj];
                int last      = nxj - 1;
                if (i == nx_[j]) {
                    ++last;
                }
                const double dx = 360.0 / double(nxj);
                int ii          = int(std::floor((x + eps - 0.0) / dx));  // compute_i_less_equal_x (:306-310)
                while (compute_x_fast(ii - 1, jjj, nxj) > x_prev + eps) {  // ATLAS-186 workaround (:418-420)
                    --ii;
                }
                const int i_minus_halo = ii - halo;
                int iii                = ii;
                while (compute_x_fast(iii + 1, jjj, nxj) < x_next - eps) {
                    ++iii;
                }
                iii                   = std::min(iii, last);
                const int i_plus_halo = iii + halo;
                ibh(jj)               = std::min(ibh(jj), i_minus_halo);
                ieh(jj)               = std::max(ieh(jj), i_plus_halo + 1);
            }
        }
    }

    // ---- point ordering (:469-571): owned row-major, halo rows above, W/E halos of owned rows, halo rows below
    std::vector<int> pi, pj;
    pi.reserve(owned + 4 * halo * 1024);
    pj.reserve(pi.capacity());
    for (int j = j_begin_; j < j_end_; ++j) {
        for (int i = i_begin_[j]; i < i_end_[j]; ++i) {
            pi.push_back(i);
            pj.push_back(j);
        }
    }
    if ((int)pi.size() != owned) {
        throw std::logic_error("StructuredColumns: owned region is not a set of full row segments");
    }
    for (int j = j_begin_halo_; j < j_begin_; ++j) {
        for (int i = ibh(j); i < ieh(j); ++i) {
            pi.push_back(i);
            pj.push_back(j);
        }
    }
    for (int j = j_begin_; j < j_end_; ++j) {
        for (int i = ibh(j); i < i_begin_[j]; ++i) {
            pi.push_back(i);
            pj.push_back(j);
        }
        for (int i = i_end_[j]; i < ieh(j); ++i) {
            pi.push_back(i);
            pj.push_back(j);
        }
    }
    for (int j = j_end_; j < j_end_halo_; ++j) {
        for (int i = ibh(j); i < ieh(j); ++i) {
            pi.push_back(i);
            pj.push_back(j);
        }
    }
    size_halo_ = (int)pi.size();
    index_i_   = pi;
    index_j_   = pj;

    // ij -> point table
    row_imin_.assign(nrows, 0);
    row_start_.assign(nrows + 1, 0);
    for (int j = j_begin_halo_; j < j_end_halo_; ++j) {
        const int r     = j - j_begin_halo_;
        const int w     = std::max(0, ieh(j) - ibh(j));
        row_imin_[r]    = ibh(j);
        row_start_[r + 1] = row_start_[r] + w;
    }
    ij_table_.assign(row_start_[nrows], -1);
    for (int n = 0; n < size_halo_; ++n) {
        const int r = pj[n] - j_begin_halo_;
        ij_table_[row_start_[r] + (pi[n] - row_imin_[r])] = n;
    }

    // ---- owner-side numbering for explicit distributions: every part stores its owned points row by row over its
    // [i_begin, i_end) range (StructuredColumns_setup.cc:591-616), so the index of global point (i, j) on its owner p
    // is row_start[p][j] + (i - i_begin[p][j]).  (The reference obtains it by asking the owner,
    // StructuredColumns_create_remote_index.cc:37-255; every part holds the whole distribution, so it is computed here.)
    std::vector<int> dist_ibeg, dist_rowstart;
    if (!cfg.distribution.empty() && cfg.nparts > 1) {
        const size_t np = (size_t)cfg.nparts;
        std::vector<int> iend(np * ny_, std::numeric_limits<int>::min());
        dist_ibeg.assign(np * ny_, BIG);
        dist_rowstart.assign(np * ny_, 0);
        int64_t c = 0;
        for (int j = 0; j < ny_; ++j) {
            for (int i = 0; i < nx_[j]; ++i, ++c) {
                const size_t o = (size_t)cfg.distribution[(size_t)c] * ny_ + j;
                dist_ibeg[o]   = std::min(dist_ibeg[o], i);
                iend[o]        = std::max(iend[o], i + 1);
            }
        }
        for (size_t p = 0; p < np; ++p) {
            int acc = 0;
            for (int j = 0; j < ny_; ++j) {
                dist_rowstart[p * ny_ + j] = acc;
                if (iend[p * ny_ + j] > dist_ibeg[p * ny_ + j]) {
                    acc += iend[p * ny_ + j] - dist_ibeg[p * ny_ + j];
                }
            }
        }
    }

    // ---- fields (:583-662)
    partition_.assign(size_halo_, 0);
    ghost_.assign(size_halo_, 0);
    glb_idx_.assign(size_halo_, 0);
    xy_.assign(2 * (size_t)size_halo_, 0.);
    remote_idx_.assign(size_halo_, 0);
    for (int n = 0; n < size_halo_; ++n) {
        const int i = pi[n], j = pj[n];
        if (j >= 0 && j < ny_) {
            xy_[2 * (size_t)n]     = gx(i, j);
            xy_[2 * (size_t)n + 1] = y_[j];
        }
        else {
            xy_[2 * (size_t)n]     = compute_x(i, j);
            xy_[2 * (size_t)n + 1] = compute_y(j);
        }
        if (j >= 0 && j < ny_ && i >= 0 && i < nx_[j]) {
            const int64_t k = offsets_[j] + i;
            partition_[n]   = partition_of(k);
            glb_idx_[n]     = k + 1;
        }
        else {
            glb_idx_[n]   = compute_g(i, j);
            partition_[n] = partition_of(glb_idx_[n] - 1);
        }
        ghost_[n] = n >= owned ? 1 : 0;
        // remote index (StructuredColumns_create_remote_index.cc): own index for owned points; for halo points the
        // index of the global point in its owner's owned ordering.  A band owns a contiguous global-index range whose
        // points are stored in global order, hence  remote = g - first_global_index(owner)  without communication.
        if (n < owned) {
            remote_idx_[n] = n;
        }
        else if (!dist_ibeg.empty()) {
            const int64_t g0 = glb_idx_[n] - 1;
            const int jj     = int(std::upper_bound(offsets_.begin(), offsets_.end(), g0) - offsets_.begin()) - 1;
            const size_t o   = (size_t)partition_[n] * ny_ + jj;
            remote_idx_[n]   = dist_rowstart[o] + (int(g0 - offsets_[jj]) - dist_ibeg[o]);
        }
        else {
            remote_idx_[n] = int(glb_idx_[n] - 1 - first_of_part_[partition_[n]]);
        }
    }
}

int StructuredColumns::index(int i, int j) const {
    if (j < j_begin_halo_ || j >= j_end_halo_) {
        throw std::out_of_range("StructuredColumns::index: j outside the halo");
    }
    const int r = j - j_begin_halo_;
    const int o = i - row_imin_[r];
    if (o < 0 || o >= row_start_[r + 1] - row_start_[r] || ij_table_[row_start_[r] + o] < 0) {
        throw std::out_of_range("StructuredColumns::index: i outside the halo");
    }
    return ij_table_[row_start_[r] + o];
}

std::vector<int> StructuredColumns::pole_row_nodes() const {
    std::vector<int> nodes;
    for (int n = 0; n < size_halo_; ++n) {
        if (index_j_[n] < 0 || index_j_[n] >= ny_) {
            nodes.push_back(n);
        }
    }
    return nodes;
}

}  // namespace functionspace
}  // namespace atlas_amd
