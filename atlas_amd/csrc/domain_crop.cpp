// See domain_crop.h.
#include "domain_crop.h"

#include <cmath>
#include <stdexcept>
#include <string>

namespace atlas_amd {
namespace grid {

namespace {
constexpr double kBoundTolerance = 1.e-6;   // degrees, RectangularDomain.cc:99
}

bool DomainCrop::whole_rows() const {
    for (size_t r = 0; r < i0.size(); ++r) {
        if (i0[r] != 0) {
            return false;
        }
    }
    return true;
}

long long DomainCrop::size() const {
    long long s = 0;
    for (int v : n) {
        s += v;
    }
    return s;
}

DomainCrop crop_to_domain(const StructuredGrid& g, double west, double east, double south, double north) {
    if (!(east >= west) || !(north >= south)) {
        throw std::invalid_argument("domain: needs west <= east and south <= north");
    }
    DomainCrop c;
    // latitudes: the grid's rows run north -> south, the contained ones are contiguous
    int first = -1, last = -1;
    for (int j = 0; j < g.ny(); ++j) {
        if (g.y[j] >= south - kBoundTolerance && g.y[j] <= north + kBoundTolerance) {
            first = first < 0 ? j : first;
            last  = j;
        }
    }
    if (first < 0) {
        throw std::invalid_argument("domain: no latitude of grid " + g.name + " lies inside [" + std::to_string(south) + ", " +
                                    std::to_string(north) + "]");
    }
    c.row_begin = first;
    c.row_end   = last + 1;
    for (int j = c.row_begin; j < c.row_end; ++j) {
        const int nx    = g.nx[j];
        const double dx = 360. / double(nx);
        // longitudes of the row are k dx for every integer k (period nx); the first one not west of the western bound
        const long long k0 = (long long)std::ceil((west - kBoundTolerance) / dx);
        const double lon0  = double(k0) * dx;
        if (lon0 > east + kBoundTolerance) {
            throw std::invalid_argument("domain: row " + std::to_string(j) + " of grid " + g.name +
                                        " has no point between the western and the eastern bound");
        }
        long long cnt = (long long)std::floor((east + kBoundTolerance - lon0) / dx) + 1;
        cnt           = cnt > nx ? nx : cnt;
        long long i   = k0 % nx;
        c.i0.push_back(int(i < 0 ? i + nx : i));
        c.n.push_back(int(cnt));
    }
    return c;
}

}  // namespace grid
}  // namespace atlas_amd
