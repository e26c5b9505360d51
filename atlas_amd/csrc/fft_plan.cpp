// See fft_plan.h.  Host-only.
#include "env.h"
#include "fft_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <functional>
#include <stdexcept>

namespace atlas_amd {
namespace fft {

namespace {

const long double kPiL = 3.14159265358979323846264338327950288L;

// exp(+2 pi i t / N), evaluated in extended precision
cplx unit_root(int64_t t, int64_t N) {
    t %= N;
    if (t < 0) {
        t += N;
    }
    long double a = 2.0L * kPiL * (long double)t / (long double)N;
    return cplx{(double)cosl(a), (double)sinl(a)};
}

// full forward DIF on a PADDED array (same stage code as the kernel)
void host_fft_dif(const FftShape& s, cplx* d, const cplx* tw, int dir) {
    int L = s.M;
    for (int i = 0; i < s.nstages; ++i) {
        dif_stage_any(s.radix[i], d, s.M, L, s.lsh[i], tw, dir, 0, 1);
        L /= s.radix[i];
    }
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) {
        ++l;
    }
    return (1 << l) == v ? l : -1;
}

}  // namespace

bool is_smooth235(int n) {
    if (n < 1) {
        return false;
    }
    for (int p : {2, 3, 5}) {
        while (n % p == 0) {
            n /= p;
        }
    }
    return n == 1;
}

int next_smooth235(int n) {
    while (!is_smooth235(n)) {
        ++n;
    }
    return n;
}

// smallest member of {1,3,5} * 2^k that is >= n: Bluestein lengths whose stages are one optional radix-3/5 stage
// followed by radix-16/8/4/2 stages
int next_bluestein_length(int n) {
    int best = 0;
    for (int f : {1, 3, 5}) {
        int m = f;
        while (m < n) {
            m *= 2;
        }
        if (best == 0 || m < best) {
            best = m;
        }
    }
    // the three extra lengths with a specialised [R0, 16, 16] instance: 9*256, 15*256, 18*256
    const bool finer = atlas_amd::env_get("ATLAS_AMD_FFT_FINER_M") ? atoi(atlas_amd::env_get("ATLAS_AMD_FFT_FINER_M")) != 0 : true;
    if (finer) {
        for (int m : {2304, 3840, 4608}) {
            if (m >= n && m < best) {
                best = m;
            }
        }
    }
    return best;
}

int coarse_bluestein_length(int n) {
    for (int m : {256, 512, 1024, 2048}) {
        if (m >= n) {
            return m;
        }
    }
    return 0;
}

FftShape make_shape(int M, int max_pow2_radix) {
    if (!is_smooth235(M)) {
        throw std::invalid_argument("make_shape: M is not {2,3,5}-smooth");
    }
    FftShape s{};
    s.M       = M;
    s.nstages = 0;
    int r     = M;
    auto push = [&](int radix) {
        if (s.nstages >= MAX_STAGES) {
            throw std::runtime_error("make_shape: too many stages");
        }
        s.radix[s.nstages++] = radix;
    };
    // odd radices first: every later stage then works on power-of-two sub-blocks
    while (r % 5 == 0) {
        push(5);
        r /= 5;
    }
    while (r % 9 == 0) {
        push(9);
        r /= 9;
    }
    while (r % 3 == 0) {
        push(3);
        r /= 3;
    }
    while (max_pow2_radix >= 16 && r % 16 == 0) {
        push(16);
        r /= 16;
    }
    while (r % 8 == 0 && r > 8) {
        push(8);
        r /= 8;
    }
    if (r == 8 || r == 4 || r == 2) {
        push(r);
        r = 1;
    }
    if (r != 1) {
        throw std::logic_error("make_shape: factorisation failed");
    }
    int L = M;
    for (int i = 0; i < s.nstages; ++i) {
        s.lsh[i] = ilog2_exact(L / s.radix[i]);
        L /= s.radix[i];
    }
    return s;
}

int FftPlanSet::plan_index(int n) const {
    for (size_t i = 0; i < plans.size(); ++i) {
        if (plans[i].n == n) {
            return (int)i;
        }
    }
    return -1;
}

// stage list of a specialised length (fft_core.h: ct_radix), as a run-time shape for the planner / emulation
static FftShape make_ct_shape(int f, int k) {
    FftShape s{};
    s.M       = f << k;
    s.nstages = ct_nstages(f, k);
    int L     = s.M;
    for (int i = 0; i < s.nstages; ++i) {
        s.radix[i] = ct_radix(f, k, i);
        s.lsh[i]   = ilog2_exact(L / s.radix[i]);
        L /= s.radix[i];
    }
    if (L != 1) {
        throw std::logic_error("make_ct_shape: stage list does not multiply to M");
    }
    return s;
}

int hybrid_dense_radix(int h) {
    for (int p : {2, 3, 5}) {
        while (h % p == 0) {
            h /= p;
        }
    }
    return h;
}

#if defined(ATLAS_AMD_EXPERIMENTS)
#include "../../tools/experiments/fft_native_plan.inc"   // make_native_shape: the native mixed-radix rows [r4], out of the product since round 5
#endif

FftPlanSet make_fft_plans(const std::vector<int>& row_lengths, bool specialised_shapes) {
    PlanOptions opt;
    opt.specialised_shapes = specialised_shapes;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_NATIVE")) {   // native mixed-radix rows: experiments build only
#if defined(ATLAS_AMD_EXPERIMENTS)
        opt.native = atoi(e) != 0;
#else
        if (atoi(e) != 0) {
            throw std::runtime_error("ATLAS_AMD_FFT_NATIVE=1 needs a library built with -DATLAS_AMD_EXPERIMENTS (make -C atlas_amd/csrc experiments)");
        }
#endif
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_FFT_HYBRID")) {
#if defined(ATLAS_AMD_EXPERIMENTS)
        opt.hybrid = atoi(e) != 0;
#else
        if (atoi(e) != 0) {   // the dense-stage kernel lives in tools/experiments: not in this build
            throw std::runtime_error("ATLAS_AMD_FFT_HYBRID=1 needs a library built with -DATLAS_AMD_EXPERIMENTS (make -C atlas_amd/csrc experiments)");
        }
#endif
    }
    return make_fft_plans(row_lengths, opt);
}

FftPlanSet make_fft_plans(const std::vector<int>& row_lengths, const PlanOptions& opt) {
    const bool specialised_shapes = opt.specialised_shapes;
    FftPlanSet ps;
    std::vector<int> ns(row_lengths);
    std::sort(ns.begin(), ns.end());
    ns.erase(std::unique(ns.begin(), ns.end()), ns.end());
    std::map<int, int64_t> tw_of_M;  // twiddle tables are shared between plans with equal M
    auto twiddles = [&](int M) -> int64_t {
        auto it = tw_of_M.find(M);
        if (it != tw_of_M.end()) {
            return it->second;
        }
        int64_t off = (int64_t)ps.table.size();
        for (int t = 0; t < M; ++t) {
            ps.table.push_back(unit_root(t, M));
        }
        tw_of_M[M] = off;
        return off;
    };
    for (int n : ns) {
        FftRowPlan p{};
        p.n    = n;
        p.ct_k = -1;
        if (n % 2 != 0 && n >= 9 && is_smooth235(n)) {
            // odd {3,5}-smooth length: a complex transform of length n itself (no half-length trick for odd n)
            p.method      = FFT_ODD;
            p.h           = n;   // complex points of the transform
            p.shape       = make_shape(n);
            p.lds_complex = padded_size(n);
            p.off_tw      = twiddles(n);
            p.off_pre     = p.off_tw;
            ps.plans.push_back(p);
            continue;
        }
        if (n % 2 != 0) {
            p.method      = FFT_DFT;
            p.h           = 0;
            p.lds_complex = 0;
            p.off_pre     = (int64_t)ps.table.size();
            for (int j = 0; j < n; ++j) {
                ps.table.push_back(unit_root(j, n));
            }
            ps.plans.push_back(p);
            continue;
        }
        const int h = n / 2;
        p.h         = h;
        // a {2,3,5}-smooth h that is NOT a length of the specialised family would take the run-time-shaped direct kernel,
        // which costs 1.4 - 1.6 x a compile-time-shaped Bluestein row of the same length (profiles/r02_fft_rowlen_probe.txt):
        // such rows go through Bluestein as well when a specialised Bluestein instance exists for them [r3] (O1280: 156 rows,
        // 3.3 % of the points, 0.57 -> 0.37 ms; the classic N grids: nearly every row).  ATLAS_AMD_FFT_SMOOTH_DIRECT=1: old rule.
        bool smooth_direct = is_smooth235(h);
        const int Mcoarse  = (opt.coarse_classes && specialised_shapes) ? coarse_bluestein_length(2 * h - 1) : 0;
        if (Mcoarse > 0) {
            smooth_direct = false;   // one of the few coarse Bluestein classes (PlanOptions::coarse_classes)
        }
        if (smooth_direct && specialised_shapes) {
            bool family = false;
            for (int f : {1, 3, 5, 9, 15}) {
                if (h % f == 0) {
                    const int k = ilog2_exact(h / f);
                    family      = family || (k >= 0 && ct_supported(f, k));
                }
            }
            const bool old_rule = atlas_amd::env_get("ATLAS_AMD_FFT_SMOOTH_DIRECT") && atoi(atlas_amd::env_get("ATLAS_AMD_FFT_SMOOTH_DIRECT")) != 0;
            if (!family && !old_rule) {
                const int Mb = next_bluestein_length(2 * h - 1);
                for (int f : {1, 3, 5, 9, 15}) {
                    if (Mb % f == 0) {
                        const int k = ilog2_exact(Mb / f);
                        if (k >= 0 && ct_supported(f, k)) {
                            smooth_direct = false;   // falls through to the Bluestein branch below
                        }
                    }
                }
            }
        }
#if defined(ATLAS_AMD_EXPERIMENTS)
        // native mixed-radix rows [r4] (tools/experiments/fft_native.h): every half length with a stage list that is not a length of
        // the specialised direct family (those kernels are compile-time shaped throughout)
        bool family_direct = false;
        if (specialised_shapes) {
            for (int f : {1, 3, 5, 9, 15}) {
                if (h % f == 0) {
                    const int k   = ilog2_exact(h / f);
                    family_direct = family_direct || (k >= 0 && ct_supported(f, k));
                }
            }
        }
        bool native = false;
        if (opt.native && specialised_shapes && Mcoarse == 0 && !family_direct && h >= opt.native_min_h && !opt.hybrid) {
            native = make_native_shape(h, p.nat, ps.nat_table);
        }
        if (native) {
            p.method      = FFT_NATIVE;
            p.shape       = FftShape{};
            p.shape.M     = h;
            p.shape.nstages = p.nat.ns;
            for (int i = 0; i < p.nat.ns; ++i) {   // DIF order, for introspection (the kernel runs from the tables)
                p.shape.radix[i] = p.nat.radix[p.nat.ns - 1 - i];
                p.shape.lsh[i]   = -1;
            }
            p.lds_complex = p.nat.lds_elems;
            p.off_tw      = twiddles(h);
            p.off_pre     = (int64_t)ps.table.size();
            for (int k = 0; k < h; ++k) {
                ps.table.push_back(unit_root(k, n));
            }
            ps.plans.push_back(p);
            continue;
        }
#endif
        if (smooth_direct) {
            p.method = FFT_DIRECT;
            // h itself a length of the specialised family?  (regular grids: every row)
            if (specialised_shapes) {
                for (int f : {1, 3, 5, 9, 15}) {
                    if (h % f == 0) {
                        const int k = ilog2_exact(h / f);
                        if (k >= 0 && ct_supported(f, k)) {
                            p.ct_f = f;
                            p.ct_k = k;
                        }
                    }
                }
            }
            p.shape = p.ct_k >= 0 ? make_ct_shape(p.ct_f, p.ct_k) : make_shape(h);
        }
#if defined(ATLAS_AMD_EXPERIMENTS)
        else if (Mcoarse == 0 && opt.hybrid && h >= opt.hybrid_min_h && hybrid_dense_radix(h) <= std::min(opt.hybrid_max_a, HYB_MAX_A)) {
            p.method = FFT_HYBRID;
            p.hyb_A  = hybrid_dense_radix(h);
            p.hyb_B  = h / p.hyb_A;
            p.shape  = make_shape(p.hyb_B, 8);   // radix-16 butterflies would cost the kernel a wavefront per SIMD
            if (p.shape.nstages >= MAX_STAGES) {
                throw std::runtime_error("make_fft_plans: too many stages");
            }
            p.shape.M                       = h;
            p.shape.radix[p.shape.nstages]  = p.hyb_A;
            p.shape.nstages += 1;
            int L = h;
            for (int i = 0; i < p.shape.nstages; ++i) {
                p.shape.lsh[i] = ilog2_exact(L / p.shape.radix[i]);
                L /= p.shape.radix[i];
            }
        }
#endif
        else {
            p.method     = FFT_BLUESTEIN;
            const int Mb = Mcoarse > 0 ? Mcoarse : next_bluestein_length(2 * h - 1);
            // M = F * 2^K with a specialised instance?
            if (specialised_shapes) {
                for (int f : {1, 3, 5, 9, 15}) {
                    if (Mb % f == 0) {
                        const int k = ilog2_exact(Mb / f);
                        if (k >= 0 && ct_supported(f, k)) {
                            p.ct_f = f;
                            p.ct_k = k;
                        }
                    }
                }
            }
            p.shape = p.ct_k >= 0 ? make_ct_shape(p.ct_f, p.ct_k) : make_shape(Mb);
        }
        const int M   = p.shape.M;
        p.lds_complex = padded_size(M);
        p.off_tw      = twiddles(M);
        p.off_pre     = (int64_t)ps.table.size();
        for (int k = 0; k < h; ++k) {
            ps.table.push_back(unit_root(k, n));
        }
        // specialised Bluestein rows: the stage-0 butterfly b reads the entries b + q * (M / R0), q < ceil(R0 / 2), some of
        // them beyond h (zero padding of the convolution: the values are not used).  Padding `pre` and `chirp` to that extent
        // (`pre`: copies of entry h - 1, `chirp`: zeros) lets the kernel address them without a clamp: one base register +
        // immediates -- and, the chirp of a padding element being zero, phase 0 needs no select for k >= h (row_ct3).
        int table_pad = 0;
        if (p.method == FFT_BLUESTEIN && p.ct_k >= 0) {
            const int R0 = p.shape.radix[0];
            table_pad    = std::max(0, (R0 + 1) / 2 * (M / R0) - h);
        }
        for (int k = 0; k < table_pad; ++k) {
            ps.table.push_back(unit_root(h - 1, n));
        }
        if (p.method == FFT_HYBRID) {
            const int A = p.hyb_A, Kp = (A + 1) / 2;
            p.hyb_Mt    = (Kp + 15) / 16;
            p.hyb_Ks    = (Kp + 3) / 4;
            p.hyb_raw   = (std::min(h, opt.max_mode) + 1 + 63) / 64 * 64;   // the device gather writes whole 64-lane granules
            p.lds_complex += p.hyb_raw;
            p.off_cs    = (int64_t)ps.table.size();
            for (int mt = 0; mt < p.hyb_Mt; ++mt) {
                for (int ks = 0; ks < p.hyb_Ks; ++ks) {
                    for (int l = 0; l < 64; ++l) {
                        const int j = 16 * mt + (l & 15), q = 4 * ks + (l >> 4);
                        cplx v{0., 0.};
                        if (j < Kp && q < Kp) {
                            v = unit_root((int64_t)j * q, A);
                            if (j == 0 || q == 0) {
                                v.im = 0.;
                            }
                        }
                        ps.table.push_back(v);
                    }
                }
            }
        }
        if (p.method == FFT_BLUESTEIN) {
            // chirp c[k] = exp(+i pi k^2 / h) = exp(2 pi i (k^2 mod 2h) / 2h)
            p.off_chirp = (int64_t)ps.table.size();
            std::vector<cplx> chirp(h);
            for (int k = 0; k < h; ++k) {
                const int64_t q = ((int64_t)k * k) % (2 * (int64_t)h);
                chirp[k]        = unit_root(q, 2 * (int64_t)h);
                ps.table.push_back(chirp[k]);
            }
            for (int k = 0; k < table_pad; ++k) {
                ps.table.push_back(cplx{0., 0.});   // a zero chirp: the padding elements of phase 0 come out as zeros unmasked
            }
            // filter b[d] = conj(c[|d|]) wrapped into M; spectrum via the kernel's own forward DIF (padded layout),
            // stored in position order and scaled by 1/M
            std::vector<cplx> b(padded_size(M), cplx{0., 0.});
            for (int d = 0; d < h; ++d) {
                b[PAD(d)] = cconj(chirp[d]);
                if (d) {
                    b[PAD(M - d)] = b[PAD(d)];
                }
            }
            host_fft_dif(p.shape, b.data(), ps.table.data() + p.off_tw, -1);
            p.off_bhat       = (int64_t)ps.table.size();
            const double inv = 1.0 / M;
            for (int i = 0; i < M; ++i) {
                ps.table.push_back(cplx{b[PAD(i)].re * inv, b[PAD(i)].im * inv});
            }
            // transposed copy [q][butterfly] for the specialised kernel's fused middle stage
            const int RL = p.shape.radix[p.shape.nstages - 1];
            const int nb = M / RL;
            p.off_bhat_t = (int64_t)ps.table.size();
            for (int q = 0; q < RL; ++q) {
                for (int bb = 0; bb < nb; ++bb) {
                    ps.table.push_back(ps.table[p.off_bhat + (int64_t)bb * RL + q]);
                }
            }
        }

        ps.plans.push_back(p);
    }
    return ps;
}

template <class S>
static void row_phase_ct_host(int ph, int t, int nt, const RowTablesCt& r, const cplx* raw, const RowOut& io, cplx* work) {
    row_phase_ct<S, false>(ph, t, nt, r, raw, io, work);   // staging area separate from the work array
}

void host_execute_row(const FftPlanSet& ps, int plan, const cplx* X, int mmax, double* y, int nthreads,
                      bool use_specialised) {
    const FftRowPlan& p = ps.plans.at(plan);
    if (p.method == FFT_ODD) {
        RowTables r;
        r.n = p.n, r.h = p.h, r.method = p.method;
        r.shape = &p.shape;
        r.tw    = ps.table.data() + p.off_tw;
        r.pre = r.chirp = r.bhat = nullptr;
        RowOut io;
        io.mmax = std::min(mmax, (p.n - 1) / 2), io.y = y, io.aligned16 = 0, io.scale = 1.0;
        std::vector<cplx> work(padded_size(p.n));
        auto rd = [&](int m) { return X[m]; };
        const int nph = row_num_phases_odd(r);
        for (int ph = 0; ph < nph; ++ph) {
            for (int t = 0; t < nthreads; ++t) {
                row_phase_odd(ph, t, nthreads, r, rd, io, work.data());
            }
        }
        return;
    }
    if (p.method == FFT_DFT) {
        const cplx* w = ps.table.data() + p.off_pre;
        const int n   = p.n;
        for (int k = 0; k < n; ++k) {
            double s = X[0].re;
            for (int m = 1; m <= std::min(mmax, n / 2); ++m) {
                const cplx t = w[(int64_t)m * k % n];
                s += 2.0 * (X[m].re * t.re - X[m].im * t.im);
            }
            y[k] = s;
        }
        return;
    }
#if defined(ATLAS_AMD_EXPERIMENTS)
    if (p.method == FFT_NATIVE) {
        std::vector<cplx> Xh(p.h + 1, cplx{0., 0.});
        for (int m = 0; m <= std::min(mmax, p.h); ++m) {
            Xh[m] = X[m];
        }
        nat_execute_row_host(p.nat, ps.nat_table.data(), ps.table.data() + p.off_tw, ps.table.data() + p.off_pre, Xh.data(),
                             std::min(mmax, p.h), y);
        return;
    }
    if (p.method == FFT_HYBRID) {
        RowTablesHyb r;
        r.n = p.n, r.h = p.h, r.A = p.hyb_A, r.B = p.hyb_B, r.Kp = (p.hyb_A + 1) / 2;
        r.Mt = p.hyb_Mt, r.Ks = p.hyb_Ks;
        r.shape = &p.shape;
        r.tw    = ps.table.data() + p.off_tw;
        r.pre   = ps.table.data() + p.off_pre;
        r.cs    = ps.table.data() + p.off_cs;
        RowOut io;
        io.mmax      = std::min(mmax, p.h);
        io.y         = y;
        io.aligned16 = 0;
        io.scale     = 1.0;
        auto rd      = [X](int m) { return X[m]; };
        std::vector<cplx> work(padded_size(p.h)), raw(p.h + 1);
        for (int t = 0; t < nthreads; ++t) {
            hyb_gather(r, rd, io, raw.data(), t, nthreads);
        }
        for (int t = 0; t < nthreads; ++t) {
            HybFoldWork fw;
            hyb_fold_prefetch(r, t, nthreads, fw);
            hyb_fold_split(r, io, raw.data(), work.data(), t, nthreads, fw);
        }
        hyb_dense_host(r, work.data());
        for (int i = p.shape.nstages - 2; i >= 0; --i) {
            for (int t = 0; t < nthreads; ++t) {
                hyb_native_phase(i, r, io, work.data(), t, nthreads);
            }
        }
        return;
    }
#endif
    RowTables r;
    r.n      = p.n;
    r.h      = p.h;
    r.method = p.method;
    r.shape  = &p.shape;
    r.tw     = ps.table.data() + p.off_tw;
    r.pre    = ps.table.data() + p.off_pre;
    r.chirp  = ps.table.data() + p.off_chirp;
    r.bhat   = ps.table.data() + p.off_bhat;
    RowOut io;
    io.mmax      = std::min(mmax, p.h);
    io.y         = y;
    io.aligned16 = 0;
    io.scale     = 1.0;
    auto rd      = [X](int m) { return X[m]; };
    std::vector<cplx> work(p.lds_complex);
    if (use_specialised && p.method == FFT_DIRECT && p.ct_k >= 0) {
        RowTablesCt rc;
        rc.n      = p.n;
        rc.h      = p.h;
        rc.tw     = r.tw;
        rc.pre    = r.pre;
        rc.chirp  = nullptr;
        rc.bhat_t = nullptr;
        const int ctf = p.ct_f, ctk = p.ct_k;
        bool done = false;
        AA_CT_DISPATCH(ctf, ctk, {
            for (int ph = 0; ph < row_num_phases_dct<S>(); ++ph) {
                for (int t = 0; t < nthreads; ++t) {
                    row_phase_dct<S>(ph, t, nthreads, rc, rd, io, work.data());
                }
            }
            done = true;
        })
        if (done) {
            return;
        }
    }
    if (use_specialised && p.method == FFT_BLUESTEIN && p.ct_k >= 0) {
        RowTablesCt rc;
        rc.n      = p.n;
        rc.h      = p.h;
        rc.tw     = r.tw;
        rc.pre    = r.pre;
        rc.chirp  = r.chirp;
        rc.bhat_t = ps.table.data() + p.off_bhat_t;
        const int ctf = p.ct_f, ctk = p.ct_k;
        bool done = false;
        AA_CT_DISPATCH(ctf, ctk, {
            std::vector<cplx> rawv(p.h + 1);
            for (int m = 0; m <= io.mmax; ++m) {
                rawv[m] = X[m];
            }
            for (int ph = 0; ph < row_num_phases_ct<S>(); ++ph) {
                for (int t = 0; t < nthreads; ++t) {
                    row_phase_ct_host<S>(ph, t, nthreads, rc, rawv.data(), io, work.data());
                }
            }
            done = true;
        })
        if (done) {
            return;
        }
    }
    const int nph = row_num_phases(r);
    for (int ph = 0; ph < nph; ++ph) {
        for (int t = 0; t < nthreads; ++t) {
            row_phase(ph, t, nthreads, r, rd, io, work.data());
        }
    }
}

}  // namespace fft
}  // namespace atlas_amd
