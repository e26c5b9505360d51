// Gaussian latitudes and global structured Gaussian grid descriptions (host only).
// Mirrors what TransLocal reads from atlas::StructuredGrid: ny(), nx(j), y(j), nxmax(), regular-ness
// (reference: src/atlas/grid/detail/grid/Structured.h:300-330, Gaussian.cc:86-177).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace atlas_amd {
namespace grid {

struct StructuredGrid {
    std::string name;
    int N = 0;               // Gaussian number (0 if not Gaussian)
    std::vector<int> nx;     // points per latitude, north -> south
    std::vector<double> y;   // latitudes in degrees, north -> south (monotone decreasing)
    bool regular = false;    // RegularGrid(grid): all rows have the same nx
    int ny() const { return (int)y.size(); }
    int nxmax() const {
        int m = 0;
        for (int v : nx) m = v > m ? v : m;
        return m;
    }
    int64_t size() const {
        int64_t s = 0;
        for (int v : nx) s += v;
        return s;
    }
    // Structured.h:308-314 with xmin = 0 (global, west = 0)
    double x(int i, int j) const { return 0. + (double)i * (360. / (double)nx[j]); }
};

// largest Gaussian number accepted by name (the reference tabulates up to N8000; 16000 leaves room for TCo15999 and
// keeps the Newton iteration and the 32-bit row lengths bounded)
constexpr int kMaxGaussianN = 16000;

bool gaussian_latitudes_tabulated(int N);
// points per latitude (north pole -> equator) of the classic reduced Gaussian grid N<N>, if tabulated
bool classic_gaussian_pl(int N, std::vector<int>& pl);
void gaussian_latitudes_npole_equator(int N, double lats[]);
void gaussian_latitudes_npole_spole(int N, double lats[]);

// "F<N>" regular Gaussian, "O<N>" octahedral reduced Gaussian
StructuredGrid make_gaussian_grid(const std::string& name);
// reduced Gaussian grid from an explicit pl array (N or 2N entries), e.g. a classic N<N> grid
StructuredGrid make_reduced_gaussian_grid(int N, const int pl[], int npl);

}  // namespace grid
}  // namespace atlas_amd
