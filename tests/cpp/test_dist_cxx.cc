// C++ driver of the distributed paths through the C ABI (include/atlas_amd.h), no Python, no torch:
//   * P ranks as host threads of this process on one GPU ("local" communicator): every rank builds its wavenumber-sharded
//     Trans, runs atlas_amd__Trans__invtrans_distributed[_many]; the bands put together must equal the single-device
//     transform bit for bit (same arithmetic per (m, latitude) and per row);
//   * StructuredColumns partitions + atlas_amd__HaloExchange__setup_comm / execute_comm between the same ranks: every
//     halo node ends up with its owner's value (the pattern of src/tests/functionspace/test_structuredcolumns_haloexchange.cc:
//     38-60: fields filled with the global index);
//   * the same calls over a real RCCL communicator with one rank (what a one-GPU box can host).
// Usage: test_dist_cxx [nranks=2] [grid=O32] [truncation=31] [nfields=3]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "atlas_amd.h"

static int failures = 0;
#define EXPECT(cond)                                                      \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::printf("FAILED %s:%d  %s  [%s]\n", __FILE__, __LINE__, #cond, atlas_amd__last_error()); \
            ++failures;                                                   \
        }                                                                 \
    } while (0)

static std::vector<double> spectra(int T, int nf, unsigned seed) {
    std::vector<double> sp((size_t)(T + 1) * (T + 2) * nf);
    uint64_t s = seed * 2654435761u + 12345;
    size_t i   = 0;
    for (int m = 0; m <= T; ++m) {
        for (int n = m; n <= T; ++n) {
            for (int imag = 0; imag < 2; ++imag) {
                for (int f = 0; f < nf; ++f) {
                    s        = s * 6364136223846793005ULL + 1442695040888963407ULL;
                    double u = ((s >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
                    sp[i++]  = (m == 0 && imag) ? 0. : u * std::pow(1. + n, -5. / 6.);
                }
            }
        }
    }
    return sp;
}

struct RankResult {
    int band_begin = 0, band_end = 0;
    std::vector<double> gp, gp2;   // nf * band points (single call, second transform of the pipelined call)
    bool halo_ok = false;
};

static void run_rank(atlas_amd_Comm* comm, const atlas_amd_Grid* grid, int T, int nf, const std::vector<double>& sp_a,
                     const std::vector<double>& sp_b, const std::vector<double>& ref_a, const std::vector<double>& ref_b,
                     RankResult* out) {
    const int P = atlas_amd__Comm__size(comm), r = atlas_amd__Comm__rank(comm);
    const std::string cfg = "nparts=" + std::to_string(P) + ";part=" + std::to_string(r) + ";shard=m";
    atlas_amd_Trans* t    = atlas_amd__Trans__new_config(grid, T, cfg.c_str(), nullptr, 0);
    EXPECT(t != nullptr);
    if (!t) {
        return;
    }
    std::vector<int> bands(P + 1);
    EXPECT(atlas_amd__Trans__bands(t, bands.data()) == 0);
    out->band_begin   = bands[r];
    out->band_end     = bands[r + 1];
    const int64_t npt = atlas_amd__Trans__nb_gridpoints(t);
    const size_t nsp  = sp_a.size();
    double* d_spa = (double*)atlas_amd__device_malloc(nsp * 8);
    double* d_spb = (double*)atlas_amd__device_malloc(nsp * 8);
    double* d_gpa = (double*)atlas_amd__device_malloc((size_t)npt * nf * 8);
    double* d_gpb = (double*)atlas_amd__device_malloc((size_t)npt * nf * 8);
    EXPECT(d_spa && d_spb && d_gpa && d_gpb);
    EXPECT(atlas_amd__device_memcpy_h2d(d_spa, sp_a.data(), nsp * 8) == 0);
    EXPECT(atlas_amd__device_memcpy_h2d(d_spb, sp_b.data(), nsp * 8) == 0);
    // small messages: the slabs are cut into several pieces
    EXPECT(atlas_amd__Trans__set_max_message_bytes(t, comm, 64 * 1024) == 0);
    EXPECT(atlas_amd__Trans__invtrans_distributed(t, comm, nf, d_spa, d_gpa) == 0);
    EXPECT(atlas_amd__Trans__synchronize(t) == 0);
    out->gp.resize((size_t)npt * nf);
    EXPECT(atlas_amd__device_memcpy_d2h(out->gp.data(), d_gpa, out->gp.size() * 8) == 0);
    // pipelined: three transforms (a, b, a)
    const double* sps[3] = {d_spa, d_spb, d_spa};
    double* gps[3]       = {d_gpa, d_gpb, d_gpa};
    EXPECT(atlas_amd__Trans__invtrans_distributed_many(t, comm, 3, nf, sps, gps) == 0);
    EXPECT(atlas_amd__Trans__synchronize(t) == 0);
    out->gp2.resize((size_t)npt * nf);
    EXPECT(atlas_amd__device_memcpy_d2h(out->gp2.data(), d_gpb, out->gp2.size() * 8) == 0);
    std::vector<double> again((size_t)npt * nf);
    EXPECT(atlas_amd__device_memcpy_d2h(again.data(), d_gpa, again.size() * 8) == 0);
    EXPECT(again == out->gp);

    // ---- halo exchange on the StructuredColumns partition of the same bands ("row_bands": blocksize 0)
    atlas_amd_StructuredColumns* fs = atlas_amd__StructuredColumns__new(grid, 2, 0, P, r, 0);
    EXPECT(fs != nullptr);
    if (fs) {
        const int n = atlas_amd__StructuredColumns__size_halo(fs), nown = atlas_amd__StructuredColumns__size_owned(fs);
        EXPECT(nown == (int)npt);
        std::vector<int> part(n), ridx(n), ghost(n);
        std::vector<int64_t> gidx(n);
        EXPECT(atlas_amd__StructuredColumns__get_int(fs, "partition", part.data()) == 0);
        EXPECT(atlas_amd__StructuredColumns__get_int(fs, "remote_idx", ridx.data()) == 0);
        EXPECT(atlas_amd__StructuredColumns__get_int(fs, "ghost", ghost.data()) == 0);
        EXPECT(atlas_amd__StructuredColumns__global_index(fs, gidx.data()) == 0);
        atlas_amd_HaloExchange* hx = atlas_amd__HaloExchange__new();
        EXPECT(hx && atlas_amd__HaloExchange__setup_comm(hx, comm, part.data(), ridx.data(), 0, n, nown) == 0);
        const int nlev = 3;
        std::vector<double> field((size_t)n * nlev, -1.);
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < nlev; ++k) {
                field[(size_t)i * nlev + k] = ghost[i] ? -1. : (double)gidx[i] * 10 + k;
            }
        }
        double* d_f = (double*)atlas_amd__device_malloc(field.size() * 8);
        EXPECT(atlas_amd__device_memcpy_h2d(d_f, field.data(), field.size() * 8) == 0);
        const int shape[2]           = {n, nlev};
        const long long strides[2]   = {nlev, 1};
        EXPECT(atlas_amd__HaloExchange__execute_comm(hx, comm, 3 /* double */, d_f, 2, shape, strides, 0, 0) == 0);
        EXPECT(atlas_amd__HaloExchange__synchronize(hx) == 0);
        EXPECT(atlas_amd__device_memcpy_d2h(field.data(), d_f, field.size() * 8) == 0);
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            for (int k = 0; k < nlev; ++k) {
                ok = ok && field[(size_t)i * nlev + k] == (double)gidx[i] * 10 + k;
            }
        }
        out->halo_ok = ok;
        EXPECT(ok);
        // ---- three transforms (a, b, a) whose outputs go straight into StructuredColumns fields [size_halo][nf] with
        // their halo exchange, in one pipelined call: owned part = this band's grid points, halo part = the owner's
        // values, i.e. the global transform at the node's global index
        {
            const int64_t npts_g = atlas_amd__Grid__size(grid);
            double* d_gpc = (double*)atlas_amd__device_malloc((size_t)npt * nf * 8);
            double* d_fld[3];
            for (double*& f : d_fld) {
                f = (double*)atlas_amd__device_malloc((size_t)n * nf * 8);
                EXPECT(f != nullptr);
            }
            const double* sps[3] = {d_spa, d_spb, d_spa};
            double* gps[3]       = {d_gpa, d_gpb, d_gpc};
            EXPECT(atlas_amd__Trans__invtrans_distributed_many_halo(t, comm, 3, nf, sps, gps, hx, d_fld) == 0);
            EXPECT(atlas_amd__Trans__synchronize(t) == 0);
            std::vector<double> fld((size_t)n * nf);
            bool fields_ok = true;
            for (int k = 0; k < 3; ++k) {
                const std::vector<double>& ref = k == 1 ? ref_b : ref_a;
                EXPECT(atlas_amd__device_memcpy_d2h(fld.data(), d_fld[k], fld.size() * 8) == 0);
                for (int i = 0; i < n && fields_ok; ++i) {
                    for (int f = 0; f < nf; ++f) {
                        fields_ok = fields_ok && fld[(size_t)i * nf + f] == ref[(size_t)f * npts_g + (gidx[i] - 1)];
                    }
                }
            }
            EXPECT(fields_ok);
            out->halo_ok = out->halo_ok && fields_ok;
            // the same buffer for consecutive transforms is refused
            double* same[2] = {d_gpa, d_gpa};
            EXPECT(atlas_amd__Trans__invtrans_distributed_many_halo(t, comm, 2, nf, sps, same, hx, d_fld) != 0);
            for (double* f : d_fld) {
                atlas_amd__device_free(f);
            }
            atlas_amd__device_free(d_gpc);
        }
        atlas_amd__device_free(d_f);
        atlas_amd__HaloExchange__delete(hx);
        atlas_amd__StructuredColumns__delete(fs);
    }
    atlas_amd__device_free(d_spa);
    atlas_amd__device_free(d_spb);
    atlas_amd__device_free(d_gpa);
    atlas_amd__device_free(d_gpb);
    atlas_amd__Trans__delete(t);
}

int main(int argc, char** argv) {
    const int P           = argc > 1 ? std::atoi(argv[1]) : 2;
    const char* gridname  = argc > 2 ? argv[2] : "O32";
    const int T           = argc > 3 ? std::atoi(argv[3]) : 31;
    const int nf          = argc > 4 ? std::atoi(argv[4]) : 3;
    if (atlas_amd__device_count() < 1) {
        std::printf("no HIP device\n");
        return 2;
    }
    atlas_amd_Grid* grid = atlas_amd__Grid__new_gaussian(gridname);
    EXPECT(grid != nullptr);
    const std::vector<double> sp_a = spectra(T, nf, 1), sp_b = spectra(T, nf, 2);

    // ---- reference: the whole transform on the device, host-pointer entry point
    const int64_t npts = atlas_amd__Grid__size(grid);
    std::vector<double> ref_a((size_t)npts * nf), ref_b((size_t)npts * nf);
    {
        atlas_amd_Trans* t = atlas_amd__Trans__new(grid, T);
        EXPECT(t != nullptr);
        EXPECT(atlas_amd__Trans__invtrans_scalar(t, nf, sp_a.data(), ref_a.data()) == 0);
        EXPECT(atlas_amd__Trans__invtrans_scalar(t, nf, sp_b.data(), ref_b.data()) == 0);
        atlas_amd__Trans__delete(t);
    }
    std::vector<int> nx(atlas_amd__Grid__ny(grid));
    EXPECT(atlas_amd__Grid__nx(grid, nx.data()) == 0);
    std::vector<int64_t> rowoff(nx.size() + 1, 0);
    for (size_t j = 0; j < nx.size(); ++j) {
        rowoff[j + 1] = rowoff[j] + nx[j];
    }
    auto check_bands = [&](const std::vector<RankResult>& res, const char* what) {
        for (const RankResult& rr : res) {
            const int64_t o = rowoff[rr.band_begin], n = rowoff[rr.band_end] - o;
            bool same_a = rr.gp.size() == (size_t)n * nf, same_b = rr.gp2.size() == (size_t)n * nf;
            for (int f = 0; f < nf && same_a && same_b; ++f) {
                same_a = std::memcmp(&rr.gp[(size_t)f * n], &ref_a[(size_t)f * npts + o], n * 8) == 0;
                same_b = std::memcmp(&rr.gp2[(size_t)f * n], &ref_b[(size_t)f * npts + o], n * 8) == 0;
            }
            EXPECT(same_a);
            EXPECT(same_b);
            EXPECT(rr.halo_ok);
        }
        std::printf("%s     %s\n", failures ? "FAILED" : "ok", what);
    };

    // ---- P ranks as threads over the "local" communicator
    {
        atlas_amd_CommHub* hub = atlas_amd__CommHub__new(P);
        EXPECT(hub != nullptr);
        std::vector<atlas_amd_Comm*> comms(P);
        for (int r = 0; r < P; ++r) {
            comms[r] = atlas_amd__Comm__new_local(hub, r);
            EXPECT(comms[r] != nullptr);
        }
        std::vector<RankResult> res(P);
        std::vector<std::thread> th;
        for (int r = 0; r < P; ++r) {
            th.emplace_back(run_rank, comms[r], grid, T, nf, std::cref(sp_a), std::cref(sp_b), std::cref(ref_a), std::cref(ref_b),
                            &res[r]);
        }
        for (auto& t : th) {
            t.join();
        }
        check_bands(res, ("distributed transform + halo exchange, " + std::to_string(P) + " ranks (threads, local communicator)").c_str());
        for (auto* c : comms) {
            atlas_amd__Comm__delete(c);
        }
        atlas_amd__CommHub__delete(hub);
    }
    // ---- one rank over a real RCCL communicator
    {
        std::vector<char> id(atlas_amd__Comm__unique_id_bytes());
        EXPECT(atlas_amd__Comm__get_unique_id(id.data()) == 0);
        atlas_amd_Comm* c = atlas_amd__Comm__new_rccl(id.data(), 1, 0);
        EXPECT(c != nullptr);
        if (c) {
            EXPECT(std::string(atlas_amd__Comm__kind(c)) == "rccl");
            std::vector<RankResult> res(1);
            run_rank(c, grid, T, nf, sp_a, sp_b, ref_a, ref_b, &res[0]);
            check_bands(res, "distributed transform + halo exchange, 1 rank over RCCL");
            atlas_amd__Comm__delete(c);
        }
    }
    // ---- min(P, visible devices) ranks over a real RCCL communicator, ONE DEVICE EACH: only where >= 2 devices are visible
    // (the grouped ncclSend / ncclRecv of csrc/comm.hip with a real peer; the one-GPU boxes of the pool skip it)
    {
        const int ndev = atlas_amd__device_count();
        const int R    = std::min(P, ndev);
        if (R >= 2) {
            std::vector<char> id(atlas_amd__Comm__unique_id_bytes());
            EXPECT(atlas_amd__Comm__get_unique_id(id.data()) == 0);
            std::vector<RankResult> res(R);
            std::vector<std::thread> th;
            for (int r = 0; r < R; ++r) {
                th.emplace_back([&, r]() {
                    EXPECT(atlas_amd__set_device(r) == 0);
                    atlas_amd_Comm* c = atlas_amd__Comm__new_rccl(id.data(), R, r);   // collective: every rank's thread enters
                    EXPECT(c != nullptr);
                    if (c) {
                        EXPECT(std::string(atlas_amd__Comm__kind(c)) == "rccl" && atlas_amd__Comm__size(c) == R);
                        run_rank(c, grid, T, nf, sp_a, sp_b, ref_a, ref_b, &res[r]);
                        atlas_amd__Comm__delete(c);
                    }
                });
            }
            for (auto& t : th) {
                t.join();
            }
            EXPECT(atlas_amd__set_device(0) == 0);
            check_bands(res, ("distributed transform + halo exchange, " + std::to_string(R) + " ranks over RCCL, one device each").c_str());
        }
        else {
            std::printf("skipped     ranks over RCCL on devices of their own: %d visible device(s)\n", ndev);
        }
    }
    atlas_amd__Grid__delete(grid);
    std::printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
