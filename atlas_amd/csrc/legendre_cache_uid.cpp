// atlas::trans::LegendreCacheCreator for the local backend: the identifier of a Legendre cache and its size estimate.
// Reference: src/atlas/trans/local/LegendreCacheCreatorLocal.cc:34-53 (hashes), :66-119 (uid), :140-148 (supported),
// :162-164 (estimate); expected strings: src/tests/trans/test_trans.cc:600-696.  Host only.
#include "legendre_cache_uid.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace atlas_amd {
namespace trans {

namespace {
// MD5 (RFC 1321) -- eckit::MD5 is what the reference hashes with
struct Md5 {
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint64_t bytes = 0;
    unsigned char block[64];
    size_t fill = 0;

    static uint32_t rotl(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }

    void compress(const unsigned char* p) {
        static const int shift[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                                      14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                      4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) {
            w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
                   ((uint32_t)p[4 * i + 3] << 24);
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g;
            if (i < 16) {
                f = (b & c) | (~b & d);
                g = i;
            }
            else if (i < 32) {
                f = (d & b) | (~d & c);
                g = (5 * i + 1) & 15;
            }
            else if (i < 48) {
                f = b ^ c ^ d;
                g = (3 * i + 5) & 15;
            }
            else {
                f = c ^ (b | ~d);
                g = (7 * i) & 15;
            }
            const uint32_t k = (uint32_t)std::floor(std::fabs(std::sin((double)(i + 1))) * 4294967296.0);
            const uint32_t t = d;
            d                = c;
            c                = b;
            b                = b + rotl(a + f + k + w[g], shift[i]);
            a                = t;
        }
        h[0] += a;
        h[1] += b;
        h[2] += c;
        h[3] += d;
    }
    void add(const void* data, size_t n) {
        const unsigned char* p = (const unsigned char*)data;
        bytes += n;
        while (n) {
            const size_t take = n < 64 - fill ? n : 64 - fill;
            std::memcpy(block + fill, p, take);
            fill += take;
            p += take;
            n -= take;
            if (fill == 64) {
                compress(block);
                fill = 0;
            }
        }
    }
    std::string hex() {
        const uint64_t bits = bytes * 8;
        const unsigned char one = 0x80, zero = 0;
        add(&one, 1);
        while (fill != 56) {
            add(&zero, 1);
        }
        unsigned char len[8];
        for (int i = 0; i < 8; ++i) {
            len[i] = (unsigned char)(bits >> (8 * i));
        }
        add(len, 8);
        char out[33];
        for (int i = 0; i < 4; ++i) {
            std::snprintf(out + 8 * i, 9, "%02x%02x%02x%02x", h[i] & 255, (h[i] >> 8) & 255, (h[i] >> 16) & 255,
                          (h[i] >> 24) & 255);
        }
        return std::string(out, 32);
    }
};

std::string first10(Md5& m) {   // truncate(): the first 10 hex digits (LegendreCacheCreatorLocal.cc:34-37)
    return m.hex().substr(0, 10);
}
}  // namespace

std::string legendre_cache_grid_hash(const grid::StructuredGrid& g) {
    Md5 m;
    for (double y : g.y) {
        const int64_t v = (int64_t)std::lround(y * 1.e8);   // latitudes to 1e-8 degrees, each a 64-bit integer
        m.add(&v, sizeof(v));
    }
    return first10(m);
}

std::string legendre_cache_uid(const grid::StructuredGrid& g, int truncation, bool flt) {
    std::string s = "local-T" + std::to_string(truncation) + "-";
    bool named_gaussian = false;
    if (g.name.size() > 1 && (g.name[0] == 'F' || g.name[0] == 'O' || g.name[0] == 'N')) {
        named_gaussian = true;
        for (size_t i = 1; i < g.name.size(); ++i) {
            named_gaussian = named_gaussian && g.name[i] >= '0' && g.name[i] <= '9';
        }
    }
    const int ny = g.ny();
    if (named_gaussian) {
        s += "GaussianN" + std::to_string(std::stoi(g.name.substr(1)));   // one cache for every global Gaussian grid of that N
    }
    else {
        bool uniform = g.regular && ny >= 2 && std::fabs(g.y[0] + g.y[ny - 1]) < 1e-9;
        const double dy = ny >= 2 ? g.y[1] - g.y[0] : 0.;
        for (int j = 1; uniform && j < ny; ++j) {
            const double d = g.y[j] - g.y[j - 1];
            uniform        = std::fabs(d - dy) <= 1e-8 + 1e-5 * std::fabs(dy);
        }
        if (uniform && std::fabs(g.y[0] - 90.) < 1e-9) {
            s += "L-ny" + std::to_string(ny);                                   // regular lon-lat with poles
        }
        else if (uniform && std::fabs(g.y[0] - (90. - 90. / ny)) < 1e-9) {
            s += "S-ny" + std::to_string(ny);                                   // shifted lat
        }
        else {
            s += "grid-" + legendre_cache_grid_hash(g);                         // no reuse across grids
        }
    }
    Md5 opt;
    opt.add("flt", 3);
    const unsigned char b = flt ? 1 : 0;
    opt.add(&b, 1);
    return s + "-OPT" + first10(opt);
}

int64_t legendre_cache_estimate(int truncation) {
    const int64_t T = truncation;
    return (T * T * T) / 2 * 8;
}

}  // namespace trans
}  // namespace atlas_amd
