// Host-side plan of the TransLocal inverse transform for a GLOBAL structured grid.
//
// TransGeometry   = what TransLocal's constructor derives from (grid, truncation)
//                   (reference: src/atlas/trans/local/TransLocal.cc:272-300, 371-488, 533-558, 592-606).
// LegendreWork    = MI355X work decomposition of the inverse Legendre stage: one work item per
//                   (zonal wavenumber m, tile of LEG_BN latitudes); the P table is stored tile-blocked so that
//                   each item streams one contiguous block from HBM.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "gaussian.h"

namespace atlas_amd {
namespace trans {

constexpr double kLatPole = 89.9999999;  // TransLocal.cc:49
constexpr int LEG_BN      = 64;          // latitudes per Legendre work item (4 MFMA row tiles of 16)
constexpr int LEG_KB      = 8;           // total wavenumbers per LDS stage (2 MFMA k-steps of 4)

int fourier_truncation(int truncation, int nx, int nxmax, int ndgl, double lat_rad, bool fullgrid);

inline int num_n(int trc, int m, bool symmetric) {  // TransLocal.cc:183-187
    int len = (trc - m + (symmetric ? 2 : 1)) / 2;
    return len < 0 ? 0 : len;
}

struct TransGeometry {
    int T          = 0;  // truncation_
    int nlats      = 0;  // ny == nlatsGlobal_
    int nlatsNH    = 0;
    int nlatsSH    = 0;
    int nlatsLeg   = 0;  // (nlatsGlobal+1)/2
    int nlatsLegR  = 0;  // nlatsLegReduced_
    bool regular   = false;
    bool has_equator = false;
    int nxmax      = 0;
    int64_t npts   = 0;
    std::vector<int> nx;
    std::vector<double> lat_deg;
    std::vector<int> nlat0;          // [T+1] first Legendre row that keeps wavenumber m
    std::vector<double> lats_leg;    // [nlatsLeg] radians, clamped to +-kLatPole
    std::vector<size_t> begin_sym;   // [T+3] reference table / cache-file offsets
    std::vector<size_t> begin_asym;  // [T+3]
    std::vector<int64_t> rowoff;     // [nlats+1] prefix sum of nx
    std::vector<int> mmax_leg;       // [nlatsLeg] highest m with nlat0[m] <= jleg (-1 if none)

    size_t size_sym() const { return begin_sym[T + 2]; }
    size_t size_asym() const { return begin_asym[T + 2]; }
    // number of Legendre rows (latitudes) wavenumber m is evaluated on
    int L(int m) const { return nlatsLegR - nlat0[m]; }
};

// ndgl / nxmax: number of latitudes and longest row that fourier_truncation sees (0: those of `g`).  They differ from
// g's when `g` is a row subset of a larger global grid (mirror-band decomposition, below).
TransGeometry make_geometry(const grid::StructuredGrid& g, int truncation, int ndgl = 0, int nxmax = 0);
// Mirror-band decomposition: part q owns the Legendre rows [b[q], b[q+1]) in BOTH hemispheres (a northern band and its
// mirror image), balanced by grid points with the BandsDistribution rule applied to the northern half.  nparts+1
// boundaries over the rows 0 .. ny/2 of a grid that is symmetric about the equator (even ny).
std::vector<int> mirror_bands(const grid::StructuredGrid& g, int nparts);
// the rows [0, b1) and [ny-b1, ny) of g as a grid of 2*b1 latitudes
grid::StructuredGrid polar_caps_grid(const grid::StructuredGrid& g, int b1);

struct LegendreItem {
    int m;
    int tile;       // tile index within m; covers Legendre rows nlat0[m] + tile*LEG_BN ...
    int nrows;      // valid rows in this tile (<= LEG_BN)
    int kpad;       // padded K (multiple of LEG_KB), same for both parities
    int64_t p_off;  // offset (doubles) of this item's P block: [parity][kpad][LEG_BN]
};

struct LegendreWork {
    std::vector<LegendreItem> items;  // launch order (XCD-interleaved, heavy first)
    int64_t table_doubles = 0;        // size of the tile-blocked table
    // first item index of each m in `by_m` order, used by table fill / cache import
    std::vector<int> first_item_of_m;  // [T+2] into items_by_m
    std::vector<LegendreItem> items_by_m;
};

// nparts/part: m-sharding for the multi-GPU decomposition -- only wavenumbers m with m % nparts == part get items
LegendreWork make_legendre_work(const TransGeometry& geo, int nparts = 1, int part = 0, bool by_band = false,
                                int row_begin = 0, int row_end = 0);
// whole-row latitude bands balanced by grid points: row j goes to the band that Atlas's BandsDistribution rule
// part(g) = ((g / blocksize) * nparts) / nb_blocks  (src/atlas/grid/detail/distribution/BandsDistribution.h:32-34,
// blocksize 1 == equal_bands) assigns to the row's FIRST point.  Returns nparts+1 row boundaries.
std::vector<int> latitude_bands(const TransGeometry& geo, int nparts);
// algorithmic flops of the inverse Legendre stage: sum_m 2*(nf*n_imag)*(K_s+K_a)*L(m)
double legendre_flops(const TransGeometry& geo, int nf);

}  // namespace trans
}  // namespace atlas_amd
