// HYBRID Fourier rows (round 2; measured 11.7 ms for the stage against 8.9 ms of Bluestein): the host / device
// row code of the dense-stage transform.  Moved out of atlas_amd/csrc/fft_core.h in round 4: the product library does not plan,
// emulate or launch these rows; `make -C atlas_amd/csrc experiments` (-DATLAS_AMD_EXPERIMENTS) includes this header from
// fft_plan.h, and tools/experiments/test_experiments.py holds the tests.
#pragma once
#include "../../atlas_amd/csrc/fft_core.h"

namespace atlas_amd {
namespace fft {

// ======================================================================================================
// HYBRID rows: h = A * B with B {2,3,5}-smooth and A the product of the prime factors > 5 of h (A odd, <= HYB_MAX_A).
// No Bluestein: one inverse DIT whose innermost stage (groups of A contiguous elements, no twiddles) is a DENSE
// A-point DFT evaluated on the matrix cores, followed by the ordinary radix-{2..16} stages of B.
//   DIF stage list: [radices of B ..., A]  (the planner appends A to make_shape(B))
//   phase 0 : the row's kept modes X[0..mmax] are gathered ONCE into an LDS staging area `raw`
//   phase 1 : c2r pre-processing (every Z[k] needs X[k] and X[h-k]: read from `raw`) fused with the symmetric split
//             of the dense stage: for the group g (digit-reversed native digits) and q = 0..(A-1)/2
//                 a_q = Z[k_q] + Z[k_{A-q}],  b_q = Z[k_q] - Z[k_{A-q}],  a_0 = Z[k_0],     k_q = g' + B q
//             stored at the positions of x_q resp. x_{A-q}
//   phase 2 : y_j     = sum_q cos(2 pi j q / A) a_q + i sum_q sin(2 pi j q / A) b_q
//             y_{A-j} = sum_q cos(2 pi j q / A) a_q - i sum_q sin(2 pi j q / A) b_q          (j = 0..(A-1)/2)
//             i.e. two REAL Kp x Kp matrices (Kp = (A+1)/2) applied to the re / im columns of a and b:
//             2 A flop per point instead of 8 A for the complex matrix, on v_mfma_f64_16x16x4_f64
//             (operand A = {cos, sin} fragments streamed from a pre-tiled table, operand B = a / b from LDS).
//             All products are accumulated before anything is written back (in place).
//   phases 3.. : DIT stages of B in LDS; the last one is fused with the store.

struct RowTablesHyb {
    int n, h;
    int A, B, Kp;            // dense radix, smooth part, (A+1)/2
    int Mt, Ks;              // Kp padded: Mt tiles of 16 rows, Ks steps of 4 columns
    const FftShape* shape;   // radices of B (DIF order) followed by A
    const cplx* tw;          // [h]  exp(+2 pi i t / h)
    const cplx* pre;         // [h]  exp(+2 pi i k / n)
    const cplx* cs;          // [Mt][Ks][64] {cos, sin}(2 pi j q / A) at j = 16 mt + (l & 15), q = 4 ks + (l >> 4); 0 beyond Kp
};

AA_HD int hyb_num_phases(const RowTablesHyb& r) {
    return 3 + (r.shape->nstages - 1);   // gather | fold + split | dense | native DIT stages (the last with the store)
}

// mode m of the row with the conventions of row_mode(): staged copy
template <class Reader>
AA_HD void hyb_gather(const RowTablesHyb& r, const Reader& rd, const RowOut& io, cplx* raw, int t, int nt) {
    constexpr int NB = 4;   // loads in flight per worker (clamped addresses, see row_phase_ct)
    for (int m0 = t; m0 <= io.mmax; m0 += NB * nt) {
        cplx v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = m0 + i * nt;
            v[i]        = rd(m <= io.mmax ? m : io.mmax);
        }
        AA_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = m0 + i * nt;
            if (m <= io.mmax) {
                if (m == 0 || m == r.h) {
                    v[i].im = 0.;
                }
                raw[m] = v[i];
            }
        }
    }
}

AA_HD cplx hyb_fold(const RowTablesHyb& r, const RowOut& io, const cplx* raw, int k, cplx pre) {
    const cplx a = ct_raw_mode(raw, io.mmax, k, r.h);
    const cplx b = cconj(ct_raw_mode(raw, io.mmax, r.h - k, r.h));
    return c2r_pre(a, b, pre);
}

// fold + symmetric split.  Worker t owns one native digit combination gf (consecutive workers: consecutive gf, so the
// table loads and the staging reads of a wavefront are contiguous) and every `per`-th q.  The table values of the first
// HYB_FOLD_NB items are requested ahead of time (`hyb_fold_prefetch`, issued next to the mode gather, whose latency then
// hides theirs); the rest in batches of HYB_FOLD_NB, loads first.
constexpr int HYB_FOLD_NB = 4;
struct HybFoldWork {
    int gf, gA, q0, per;            // per == 0: this worker has no items
    cplx p1[HYB_FOLD_NB], p2[HYB_FOLD_NB];
};

AA_HD void hyb_fold_loads(const RowTablesHyb& r, const HybFoldWork& w, int qb, cplx* p1, cplx* p2) {
    const int A = r.A, B = r.B, Kp = r.Kp;
#pragma unroll
    for (int i = 0; i < HYB_FOLD_NB; ++i) {
        int q = qb + i * w.per;
        q     = q < Kp ? q : Kp - 1;
        p1[i] = r.pre[w.gf + B * q];
        p2[i] = r.pre[w.gf + B * (q ? A - q : 0)];
    }
}

AA_HD void hyb_fold_prefetch(const RowTablesHyb& r, int t, int nt, HybFoldWork& w) {
    const int B = r.B;
    w.per       = nt / B;
    w.q0        = w.per ? t / B : 0;
    w.gf        = w.per ? t - w.q0 * B : 0;
    if (w.per == 0 || w.q0 >= w.per) {
        w.per = 0;
        return;
    }
    w.gA = pos_of_freq(*r.shape, w.gf);   // = g * A: position of the group's first element
    hyb_fold_loads(r, w, w.q0, w.p1, w.p2);
}

AA_HD void hyb_fold_item(const RowTablesHyb& r, const RowOut& io, const cplx* raw, cplx* work, int gf, int gA, int q,
                         cplx pre1, cplx pre2) {
    const int A = r.A, B = r.B;
    const cplx z = hyb_fold(r, io, raw, gf + B * q, pre1);
    if (q == 0) {
        work[PAD(gA)] = z;
    }
    else {
        const cplx z2         = hyb_fold(r, io, raw, gf + B * (A - q), pre2);
        work[PAD(gA + q)]     = cadd(z, z2);
        work[PAD(gA + A - q)] = csub(z, z2);
    }
}

AA_HD void hyb_fold_split(const RowTablesHyb& r, const RowOut& io, const cplx* raw, cplx* work, int t, int nt,
                          const HybFoldWork& w) {
    const int B = r.B, Kp = r.Kp;
    if (nt / B == 0) {   // more native digit combinations than workers (tiny A): plain loops
        for (int gf = t; gf < B; gf += nt) {
            const int gA = pos_of_freq(*r.shape, gf);
            for (int q = 0; q < Kp; ++q) {
                hyb_fold_item(r, io, raw, work, gf, gA, q, r.pre[gf + B * q], r.pre[gf + B * (q ? r.A - q : 0)]);
            }
        }
        return;
    }
    if (w.per == 0) {
        return;
    }
    bool first = true;
    for (int qb = w.q0; qb < Kp; qb += HYB_FOLD_NB * w.per) {
        cplx p1[HYB_FOLD_NB], p2[HYB_FOLD_NB];
        if (first) {
#pragma unroll
            for (int i = 0; i < HYB_FOLD_NB; ++i) {
                p1[i] = w.p1[i];
                p2[i] = w.p2[i];
            }
            first = false;
        }
        else {
            hyb_fold_loads(r, w, qb, p1, p2);
            AA_SCHED_FENCE();
        }
#pragma unroll
        for (int i = 0; i < HYB_FOLD_NB; ++i) {
            const int q = qb + i * w.per;
            if (q < Kp) {
                hyb_fold_item(r, io, raw, work, w.gf, w.gA, q, p1[i], p2[i]);
            }
        }
    }
}

// dense stage, host form (one caller does the whole row): the same sums in natural order
inline void hyb_dense_host(const RowTablesHyb& r, cplx* work) {
    const int A = r.A, Kp = r.Kp;
    cplx y[HYB_MAX_A];
    for (int g = 0; g < r.B; ++g) {
        for (int j = 0; j < Kp; ++j) {
            double ur = 0., ui = 0., vr = 0., vi = 0.;
            for (int q = 0; q < Kp; ++q) {
                const cplx cs = r.cs[((j >> 4) * r.Ks + (q >> 2)) * 64 + ((q & 3) << 4) + (j & 15)];
                const cplx a  = work[PAD(g * A + q)];
                const cplx b  = work[PAD(g * A + (q ? A - q : 0))];
                ur += cs.re * a.re;
                ui += cs.re * a.im;
                vr += cs.im * b.re;
                vi += cs.im * b.im;
            }
            y[j] = cplx{ur - vi, ui + vr};
            if (j) {
                y[A - j] = cplx{ur + vi, ui - vr};
            }
        }
        for (int j = 0; j < A; ++j) {
            work[PAD(g * A + j)] = y[j];
        }
    }
}

#if defined(__HIPCC__) && defined(ATLAS_AMD_EXPERIMENTS)
typedef double aa_d4 __attribute__((ext_vector_type(4)));
// dense stage on the matrix cores.  Output tile (mt, ct) = rows j = 16 mt .. +15, groups g = 16 ct .. +15, numbered
// ct * Mt + mt and dealt to the wavefronts round robin, HYB_UPW per wavefront and round.  A round covers whole column
// tiles ct (all their mt): the products of a round are accumulated in registers, and only when every wavefront has
// finished reading the round's operands are the results written back in place; different rounds touch different
// groups g, so there is no ordering between rounds.  Needs (nt / 64) * HYB_UPW >= Mt (planner).
__device__ __forceinline__ void hyb_dense_device(const RowTablesHyb& r, cplx* work, int t, int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int A = r.A, B = r.B, Kp = r.Kp, Mt = r.Mt, Ks = r.Ks;
    const int lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const int Nt        = (B + 15) >> 4;
    const int ntiles    = Mt * Nt;
    const int per_round = (nw * HYB_UPW / Mt) * Mt;
    const int lr = lane & 15, lq = lane >> 4;
    for (int base = 0; base < ntiles; base += per_round) {
        const int end = base + per_round < ntiles ? base + per_round : ntiles;
        aa_d4 acc[HYB_UPW][4];
#pragma unroll
        for (int u = 0; u < HYB_UPW; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[u][c] = aa_d4{0., 0., 0., 0.};
        }
#pragma unroll
        for (int u = 0; u < HYB_UPW; ++u) {
            const int tile = base + wave + u * nw;
            if (tile < end) {   // wave-uniform
                const int mt = tile % Mt, ct = tile / Mt;
                int g = 16 * ct + lr;
                g     = g < B ? g : B - 1;
                const cplx* cst = r.cs + (size_t)mt * Ks * 64 + lane;
                const int gA    = g * A;
                // the coefficient fragments come from global memory (L2): four steps are fetched ahead of their use
                constexpr int PF = 4;
                cplx csn[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i) csn[i] = cst[(i < Ks ? i : Ks - 1) * 64];
                for (int ks0 = 0; ks0 < Ks; ks0 += PF) {
                    cplx csv[PF];
#pragma unroll
                    for (int i = 0; i < PF; ++i) csv[i] = csn[i];
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const int kn = ks0 + PF + i;
                        csn[i]       = cst[(kn < Ks ? kn : Ks - 1) * 64];
                    }
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        if (ks0 + i < Ks) {
                            int q        = 4 * (ks0 + i) + lq;
                            q            = q < Kp ? q : Kp - 1;   // padding columns: finite operand times a zero coefficient
                            const cplx a = work[PAD(gA + q)];
                            const cplx b = work[PAD(gA + (q ? A - q : 0))];
                            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(csv[i].re, a.re, acc[u][0], 0, 0, 0);
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(csv[i].re, a.im, acc[u][1], 0, 0, 0);
                            acc[u][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(csv[i].im, b.re, acc[u][2], 0, 0, 0);
                            acc[u][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(csv[i].im, b.im, acc[u][3], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < HYB_UPW; ++u) {
            const int tile = base + wave + u * nw;
            if (tile < end) {
                const int mt = tile % Mt, ct = tile / Mt;
                const int g  = 16 * ct + lr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 16 * mt + lq + 4 * i;   // result rows of the f64 instruction: (l >> 4) + 4 * reg
                    if (g < B && j < Kp) {
                        const double ur = acc[u][0][i], ui = acc[u][1][i], vr = acc[u][2][i], vi = acc[u][3][i];
                        work[PAD(g * A + j)] = cplx{ur - vi, ui + vr};
                        if (j) {
                            work[PAD(g * A + A - j)] = cplx{ur + vi, ui - vr};
                        }
                    }
                }
            }
        }
    }
#endif
}
#endif

// last DIT stage (stage 0: one block of length M, twiddles w_M^b) fused with the store of y[2k], y[2k+1]
template <int R, bool F32, bool ALIGNED>
AA_HD void dit_stage0_store(cplx* d, int M, const cplx* __restrict__ tw, const RowOut& io, int t, int nt) {
    const int Ls = M / R;
    for (int b = t; b < Ls; b += nt) {
        cplx x[R];
        const cplx w1 = tw[b];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = d[PAD(b + q * Ls)];
        twiddle_apply<R>(x, w1);
        bfly<R>(x, +1);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            store_pair_t<F32, ALIGNED>(io, b + q * Ls, cplx{x[q].re * io.scale, x[q].im * io.scale});
        }
    }
}
template <bool F32, bool ALIGNED>
AA_HD void dit_stage0_store_any(int R, cplx* d, int M, const cplx* __restrict__ tw, const RowOut& io, int t, int nt) {
    AA_RADIX_SWITCH(R, (dit_stage0_store<RR, F32, ALIGNED>(d, M, tw, io, t, nt)))
}

// phases 3.. of a hybrid row (phases 0..2 are driven by the caller: they differ between host and device)
AA_HD void hyb_native_phase(int i, const RowTablesHyb& r, const RowOut& io, cplx* work, int t, int nt) {
    const FftShape& s = *r.shape;
    if (i > 0) {
        dit_stage_any(s.radix[i], work, r.h, stage_L(s, i), s.lsh[i], r.tw, +1, t, nt);
        return;
    }
    with_store_flavour(io, [&](auto f32c, auto alc) {
        dit_stage0_store_any<decltype(f32c)::value, decltype(alc)::value>(s.radix[0], work, r.h, r.tw, io, t, nt);
    });
}


}  // namespace fft
}  // namespace atlas_amd
