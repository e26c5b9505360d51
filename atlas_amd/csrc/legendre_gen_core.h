// Generation of the tile-blocked Legendre table ON THE DEVICE: the latitude-dependent part of
// compute_legendre_polynomials_lat (reference: src/atlas/trans/local/LegendrePolynomials.cc:47-151) arranged so that
// thousands of independent dependency chains run side by side.  Shared by the HIP kernels
// (legendre_gen_kernel.hip) and their host emulation (legendre_host.cpp, CPU tests), like fft_core.h.
//
// What is data-parallel in the reference's per-latitude routine:
//   * the first two columns P(0,n), P(1,n) are Fourier series in the colatitude (:85-115): one (latitude, n) pair
//     per thread, terms added in the reference's order;
//   * the diagonal P(m,m) is a scalar chain per latitude (:122-130): computed on the host (O(T) per latitude);
//   * the three-term recurrence (:136-149) couples (m,n) to (m-2,n-2), (m-2,n-1), (m,n-1): for one latitude the rows
//     m of one parity form one chain, so (latitude, parity of m) pairs are independent -- 2 x nlats chains of
//     ~T^2/4 steps, consecutive latitudes in consecutive lanes (all table/scratch accesses coalesced).
// Everything that involves sqrt, division, sin or cos (zfn, recurrence coefficients, cos/sin of multiples of the
// colatitude, the diagonal) is prepared on the host with the code of legendre_host.cpp, O(T^2) in total; the device
// performs only IEEE multiplications, additions and subtractions in the reference's order with contraction disabled,
// so the table is bit-identical to the host-generated one.
#pragma once

#include <cstddef>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LG_HD __host__ __device__ __forceinline__
#else
#define LG_HD inline
#endif

namespace atlas_amd {
namespace trans {

constexpr int LG_TILE = 64;  // == LEG_BN: latitudes per table tile (and lanes per wavefront)

struct LegendreGenParams {
    int trc;        // truncation of the table = T + 1
    int T;
    int nlats;      // Legendre rows to generate (nlatsLegReduced)
    int lat_pitch;  // >= nlats, multiple of 64; latitude is the fastest index of every per-latitude array below
    // latitude independent
    const double* zfn;  // [(trc+1)^2]  zfn(n, k) at n*(trc+1)+k, zfn(odd n, 0) = 0
    const double* sq1;  // [trc+1]      1/sqrt(n(n+1))
    const double* ca;   // packed (m,n) triangles: idx(m,n) = (2 trc + 3 - m) m / 2 + (n - m)
    const double* cb;
    const double* cc;
    // per latitude, [j][lat_pitch]
    const double* vcos;  // cos(j * colatitude), j = 0..trc
    const double* vsin;
    const double* diag;  // P(m,m), m = 0..trc (m = 0, 1 unused)
    const double* mu;    // [lat_pitch] sin(latitude) as the reference computes it (cos of the colatitude; 1 at a pole)
    const int* mstop;    // [lat_pitch] highest m whose rows include this latitude (-1: none)
    // work arrays
    double* col01;  // [2][trc+1][lat_pitch]   P(0,n), P(1,n)
    double* rows;   // [2 parity][2 buffers][trc+1][lat_pitch]  the rows m-2 and m of each chain
    // destination: tile-blocked table (trans_plan.h)
    double* table;
    const int* nlat0;            // [T+1]
    const int* first_item_of_m;  // [T+2] into the two arrays below (items_by_m order)
    const long long* item_p_off; // -1: tile not held by this device
    const int* item_kpad;
};

LG_HD size_t lg_idx(int trc, int m, int n) {
    return size_t(2 * trc + 3 - m) * size_t(m) / 2 + size_t(n - m);
}

// P(0,jn) and P(1,jn) of one latitude (LegendrePolynomials.cc:85-115).  vcos/vsin point at this latitude, `pitch`
// apart for consecutive j.
LG_HD void legendre_series_point(const double* z /* zfn row of jn */, int jn, double sq, const double* vcos,
                                 const double* vsin, size_t pitch, double& p0, double& p1) {
#pragma clang fp contract(off)
    // value: sum of s(n,k) cos(k colat), half weight on k = 0; derivative term: sum of s(n,k) k sin(k colat) scaled by
    // 1/sqrt(n(n+1)); terms in ascending k, each added to the running sum (the reference's order)
    const int odd  = jn & 1;
    double value   = odd ? 0. : 0.5 * z[0];
    double slope   = 0.0;
    for (int k = 2 - odd; k <= jn; k += 2) {
        value = value + z[k] * vcos[size_t(k) * pitch];
        slope = slope + sq * z[k] * k * vsin[size_t(k) * pitch];
    }
    p0 = value;
    p1 = slope;
}

// one (latitude, n) pair of the series stage; n = 0 holds the constant P(0,0) = 1
LG_HD void legendre_series_store(const LegendreGenParams& g, int lat, int jn) {
    const size_t pitch = size_t(g.lat_pitch);
    double p0 = 1., p1 = 0.;
    if (jn > 0) {
        legendre_series_point(g.zfn + size_t(jn) * size_t(g.trc + 1), jn, g.sq1[jn], g.vcos + lat, g.vsin + lat, pitch,
                              p0, p1);
    }
    g.col01[size_t(jn) * pitch + lat]                           = p0;
    g.col01[(size_t(g.trc + 1) + size_t(jn)) * pitch + lat]     = p1;
}

// where row m of Legendre row `lat` goes in the tile-blocked table (null if this device does not hold the tile):
// P(m,n) -> base[(n-m)&1][((ntop[(n-m)&1] - n) / 2) * LG_TILE]   (n descending within a block, trans_plan.h)
struct LegendreRowDest {
    double* base[2];
    int ntop[2];
};
LG_HD LegendreRowDest legendre_row_dest(const LegendreGenParams& g, int lat, int m) {
    LegendreRowDest d;
    d.base[0] = d.base[1] = nullptr;
    d.ntop[0] = g.trc - ((g.trc - m) & 1);      // largest n <= trc with n - m even (symmetric block)
    d.ntop[1] = g.trc - 1 + ((g.trc - m) & 1);
    if (m > g.T) {
        return d;
    }
    const int c = lat - g.nlat0[m];
    if (c < 0) {
        return d;
    }
    const int item = g.first_item_of_m[m] + c / LG_TILE;
    if (item >= g.first_item_of_m[m + 1]) {
        return d;  // wavenumber of another device
    }
    const long long p_off = g.item_p_off[item];
    if (p_off < 0) {
        return d;  // tile of another latitude band
    }
    d.base[0] = g.table + p_off + (c % LG_TILE);
    d.base[1] = d.base[0] + (long long)g.item_kpad[item] * LG_TILE;
    return d;
}
LG_HD void legendre_row_store(const LegendreRowDest& d, int m, int n, double v) {
    if (d.base[0]) {
        const int par = (n - m) & 1;
        d.base[par][(long long)((d.ntop[par] - n) >> 1) * LG_TILE] = v;
    }
}

// all rows m = parity, parity + 2, ... of one latitude (LegendrePolynomials.cc:136-149, m outer / n inner as in
// legendre_host.cpp)
LG_HD void legendre_chain(const LegendreGenParams& g, int lat, int parity) {
#pragma clang fp contract(off)
    const int trc      = g.trc;
    const size_t pitch = size_t(g.lat_pitch);
    const int mstop    = g.mstop[lat];
    if (mstop < parity) {
        return;
    }
    const double x = g.mu[lat];
    double* buf[2] = {g.rows + size_t(parity * 2 + 0) * size_t(trc + 1) * pitch + lat,
                      g.rows + size_t(parity * 2 + 1) * size_t(trc + 1) * pitch + lat};
    // first row of the chain: the series column
    {
        const double* col        = g.col01 + size_t(parity) * size_t(trc + 1) * pitch + lat;
        const LegendreRowDest d = legendre_row_dest(g, lat, parity);
        for (int n = parity; n <= trc; ++n) {
            const double v            = col[size_t(n) * pitch];
            buf[0][size_t(n) * pitch] = v;
            legendre_row_store(d, parity, n, v);
        }
    }
    int cur = 1;
    for (int m = parity + 2; m <= mstop && m < trc; m += 2, cur ^= 1) {
        double* p       = buf[cur];
        const double* q = buf[cur ^ 1];
        const LegendreRowDest d = legendre_row_dest(g, lat, m);
        double pm1              = g.diag[size_t(m) * pitch + lat];
        p[size_t(m) * pitch]    = pm1;
        legendre_row_store(d, m, m, pm1);
        const size_t base = lg_idx(trc, m, m);
        const int nfirst  = m + 1 > 3 ? m + 1 : 3;
        double q0         = q[size_t(nfirst - 2) * pitch];
        for (int n = nfirst; n <= trc; ++n) {
            const double q1 = q[size_t(n - 1) * pitch];
            const size_t i  = base + size_t(n - m);
            const double v  = g.ca[i] * q0 - g.cb[i] * q1 * x + g.cc[i] * pm1 * x;
            p[size_t(n) * pitch] = v;
            legendre_row_store(d, m, n, v);
            pm1 = v;
            q0  = q1;
        }
    }
}

}  // namespace trans
}  // namespace atlas_amd
