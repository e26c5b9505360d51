// The fp32 variant of the Fourier stage, TWO FIELDS PER JOB [r4]: a "real number" that is a pair of floats -- lane x belongs to
// field f, lane y to field f + 1 -- under the same complex type, butterflies and stage code as the fp64 rows (fft_core.h is templated
// on the complex type).  Every +, -, * of the row arithmetic becomes one packed v_pk_{add,mul,fma}_f32 that works on both fields, at
// the issue cost of the fp64 instruction it replaces (32 FMA / clock / SIMD either way), with the SAME twiddle for both lanes; an
// element is 16 bytes in registers and LDS, exactly the fp64 element.  In the fp32 Fourier intermediate the two fields of a pair
// sit side by side -- (re f, im f, re f+1, im f+1) = 16 contiguous bytes -- so the gather is the fp64 rows' LDS-DMA request and a
// job moves a 128-byte line for 16 bytes instead of 8.  Against the (re, im)-packed one-field form (fft_core.h: cplxf): half the
// jobs, each with the instruction count of an fp64 job.
//
// Device-only header (clang ext_vector_type); the host emulation of the rows (fft_plan.cpp) does not need it: the lanes of a
// packed instruction are independent IEEE fp32 operations.
#pragma once
#include "fft_core.h"

namespace atlas_amd {
namespace fft {

struct f32x2 {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 v;
    f32x2() = default;
    AA_HD f32x2(v2 q) : v(q) {}
    AA_HD f32x2(float a, float b) : v{a, b} {}
    AA_HD f32x2(float s) : v{s, s} {}
    AA_HD f32x2(double s) : v{(float)s, (float)s} {}
    AA_HD f32x2(int s) : v{(float)s, (float)s} {}
};
AA_HD f32x2 operator+(f32x2 a, f32x2 b) { return f32x2(a.v + b.v); }
AA_HD f32x2 operator-(f32x2 a, f32x2 b) { return f32x2(a.v - b.v); }
AA_HD f32x2 operator*(f32x2 a, f32x2 b) { return f32x2(a.v * b.v); }
AA_HD f32x2 operator-(f32x2 a) { return f32x2(-a.v); }

using cplxp = cplx_t<f32x2>;   // 16 bytes: re = (re of field f, re of field f + 1), im likewise
static_assert(sizeof(cplxp) == 16 && alignof(cplxp) == 16, "an element of a field pair is the 16-byte element of the fp64 rows");

// table value (float complex) -> both lanes
AA_HD cplxp both(cplxf w) {
    return cplxp{f32x2(w.re), f32x2(w.im)};
}

// the row tables of a pair: float complex values, read into both lanes
struct PairTable {
    const cplxf* p;
    AA_HD cplxp operator[](long long i) const { return both(p[i]); }
    AA_HD PairTable operator+(long long o) const { return PairTable{p + o}; }
};
template <>
struct table_ptr_of<cplxp> {
    using type = PairTable;
};

// element k of the staging area: it holds the intermediate's own order (re a, im a, re b, im b)
AA_HD cplxp raw_elem(const cplxp* raw, int k) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 q = *reinterpret_cast<const v4*>(raw + k);
    return cplxp{f32x2(q.x, q.z), f32x2(q.y, q.w)};
}
// mode m of the gathered pair (fft_core.h: ct_raw_mode; this overload is found by argument type)
AA_HD cplxp ct_raw_mode(const cplxp* raw, int mmax, int m, int h) {
    cplxp v = raw_elem(raw, m <= mmax ? m : mmax);
    if (m > mmax) {
        v = cplxp{0, 0};
    }
    if (m == 0 || m == h) {
        v.im = 0;   // conventions of row_mode()
    }
    return v;
}
AA_HD cplxp pair_raw_mode(const cplxp* raw, int mmax, int m, int h) {
    return ct_raw_mode(raw, mmax, m, h);
}

// y[2k], y[2k+1] of both fields (fft_core.h: store_pair_t; RowOut::pair_*).  ALIGNED is field a's flavour.
template <bool F32, bool ALIGNED>
__device__ __forceinline__ void store_pair_t(const RowOut& io, int64_t k, cplxp z) {
    if (!F32) {
        return;   // (with_store_flavour instantiates the fp64 flavours too: never taken, a field pair is the fp32 variant)
    }
    typedef float f2_t __attribute__((ext_vector_type(2)));
    float* ya = reinterpret_cast<float*>(io.y);
    if (ALIGNED) {
        __builtin_nontemporal_store(f2_t{z.re.v.x, z.im.v.x}, reinterpret_cast<f2_t*>(ya + 2 * k));
    }
    else {
        __builtin_nontemporal_store(z.re.v.x, ya + 2 * k);
        __builtin_nontemporal_store(z.im.v.x, ya + 2 * k + 1);
    }
    if (io.pair_b) {
        float* yb = ya + io.pair_stride;
        if (io.pair_b_aligned) {
            __builtin_nontemporal_store(f2_t{z.re.v.y, z.im.v.y}, reinterpret_cast<f2_t*>(yb + 2 * k));
        }
        else {
            __builtin_nontemporal_store(z.re.v.y, yb + 2 * k);
            __builtin_nontemporal_store(z.im.v.y, yb + 2 * k + 1);
        }
    }
}

__device__ __forceinline__ void pin_register(f32x2& x) {   // trace builds (fft_ct_rows.h: AA_PIN)
    asm volatile("" : "+v"(x.v));
}

}  // namespace fft
}  // namespace atlas_amd
