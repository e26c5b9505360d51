// Fourier-series representation of the normalised Legendre polynomial of degree n in the colatitude theta,
//     P_n(cos theta) = sum_{k = n, n-2, ...} s(n, k) cos(k theta)          (half weight on k = 0),
// the starting point of both the Gaussian-latitude Newton iteration (Latitudes.cc:227-273) and the first two columns of
// the Legendre table (LegendrePolynomials.cc:24-45, 85-115).  The leading coefficient is a running product, the others
// follow from it by a downward two-term ratio.  Every product below is evaluated in the order the reference evaluates
// it (tables and latitudes are compared bit for bit), but each degree costs O(n) here: the reference rebuilds the
// leading product from scratch for every n.
#pragma once
#include <cmath>

namespace atlas_amd {

// s(j, j) from s(j-1, j-1); s(0, 0) = 2
inline double legendre_series_lead(double lead_of_previous_degree, int j) {
    return lead_of_previous_degree * std::sqrt(1. - 0.25 / (static_cast<double>(j) * static_cast<double>(j)));
}

// row[k] = s(n, k) for k = n, n-2, ... (the entries of the other parity are not touched)
inline void legendre_series_row(int n, double lead, double* row) {
    row[n]         = lead;
    const int last = n - (n & 1);
    for (int step = 2; step <= last; step += 2) {
        const double num = (step - 1.) * (2. * n - step + 2.);
        const double den = step * (2. * n - step + 1.);
        row[n - step]    = row[n - step + 2] * num / den;
    }
}

}  // namespace atlas_amd
