// Distributed inverse transform inside the library: Legendre stage sharded by zonal wavenumber, m -> latitude
// transposition of the Fourier intermediate over a parallel::Comm (RCCL send/recv over xGMI), Fourier stage on the local
// latitude band (Atlas's BandsDistribution rule) -- the decomposition of SURVEY section 8(e).
//
// TransLocal is single-process (it throws if mpi::size() > 1, src/atlas/trans/local/TransLocal.cc:338-340); the
// per-rank stages are those of trans.h on a Trans made with (nparts, part, shard = "m").
//
// Rank p owns the wavenumbers m with m % P == p (local index m / P).  Its intermediate F_p[lat][m_local][RP] holds all
// latitudes; the rows of latitude band q form one contiguous slab, sent to rank q.  Rank q receives, from every p, the
// slab [rows of band q][cnt_p][RP] into R at out_offset[p]; the Fourier kernels gather wavenumber m from piece m % P at
// local index m / P (fft_kernel.hip: ModeReader), no repacking pass.
//
// Streams: stages on the Trans stream, exchanges on a second stream; invtrans_many() pipelines consecutive transforms
// (the exchange of transform i overlaps the Legendre stage of i+1 and the Fourier stage of i-1).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <set>
#include <utility>
#include <vector>

#include "comm.h"
#include "halo_exchange.h"
#include "trans.h"

namespace atlas_amd {
namespace trans {

struct TransposePlan {
    std::vector<int> cnt;               // wavenumbers owned by each rank
    std::vector<int> rows;              // latitude rows of each band
    std::vector<int64_t> out_offsets;   // doubles: where rank p's slab starts in R
    int64_t out_total = 0;              // doubles in R
};
TransposePlan make_transpose_plan(int T, int RP, const std::vector<int>& bands, int nparts, int part);

// the messages of rank `part`: slabs cut by rows into pieces of at most max_message_elems doubles; the cut depends on
// global quantities only, so both ends of a pair cut alike and list the pieces in the same order
struct TransposeMsg {
    int peer;
    int64_t send_begin, send_end;   // doubles, into F (flat)
    int64_t recv_begin, recv_end;   // doubles, into R
};
std::vector<TransposeMsg> transpose_messages(const TransposePlan& plan, const std::vector<int>& bands, int RP, int nparts,
                                             int part, int64_t max_message_elems);

// ---- packed transposition [r3]: only what the Fourier stage reads travels.  Row `lat` keeps the wavenumbers m <= mmax(lat)
// (its Fourier truncation); rank p owns kept(p, lat) = |{m <= mmax(lat) : m % P == p}| of them, each with `cols` = 2 * nb_fields
// live columns (no pitch padding).  Rank p packs its intermediate row by row,
//     S_p[rowoff_p[lat] + ml * cols + c] = F_p[(lat * cnt_p + ml) * RP + c],   ml < kept(p, lat), c < cols,
// the rows of band q are one contiguous run of S_p and go to rank q, which receives the P runs back to back into R; the
// Fourier kernels read wavenumber m of local row r at  R[out_offset[m % P] + rowoff_{m % P}[band_q + r] - rowoff_{m % P}[band_q]
// + (m / P) * cols + 2 f].  At TL1279 / O1280 / 137 fields: 5.07 GB over all ranks instead of 7.55 GB (slabs of allocated rows).
struct PackedTransposePlan {
    int cols = 0;                                  // doubles per (row, wavenumber): 2 * nb_fields
    std::vector<std::vector<int64_t>> rowoff;      // [nparts][nlats + 1] prefix sums of kept(p, lat) * cols (doubles)
    std::vector<int64_t> out_offsets;              // [nparts] where rank p's run starts in this rank's R
    int64_t out_total  = 0;                        // doubles in R
    int64_t send_total = 0;                        // doubles in this rank's S (= rowoff[part][nlats])
};
// row_mmax[lat]: highest kept wavenumber of the row (-1: none), all rows of the grid
PackedTransposePlan make_packed_transpose_plan(const std::vector<int>& row_mmax, int cols, const std::vector<int>& bands,
                                               int nparts, int part);
// messages of rank `part`, cut by rows into pieces of at most max_message_elems doubles (>= 1); the cut depends on global
// quantities only: both ends of a pair cut alike and list the pieces in the same order.  send_* index S, recv_* index R.
std::vector<TransposeMsg> packed_transpose_messages(const PackedTransposePlan& plan, const std::vector<int>& bands, int nparts,
                                                    int part, int64_t max_message_elems);

// measurement aid: one rank's pack kernel alone on the device (ms per launch; bytes packed per launch)
double pack_probe(Trans& trans, int nb_fields, int reps, int64_t* bytes);
// ... and its Fourier stage as the distributed transform runs it (packed runs of P sources, per-row offsets), ms per stage
double fourier_packed_probe(Trans& trans, int nb_fields, int reps);

class DistributedTrans {
public:
    DistributedTrans(Trans& trans, parallel::Comm& comm);
    ~DistributedTrans();
    DistributedTrans(const DistributedTrans&)            = delete;
    DistributedTrans& operator=(const DistributedTrans&) = delete;

    // sp: full spectra (replicated, as for TransLocal); gp: nb_fields * (points of the local band).  Asynchronous on the
    // Trans stream.
    void invtrans(int nb_fields, const double* sp_dev, double* gp_dev);
    void invtrans_many(int ntransforms, int nb_fields, const double* const* sp_dev, double* const* gp_dev);
    // [r3] sharded input: every sp_dev[i] holds only this rank's wavenumbers (Trans::legendre_device_sharded; SURVEY 8(e):
    // the spectra are scattered by m) -- 1/P of the replicated array per rank
    void invtrans_many_sharded(int ntransforms, int nb_fields, const double* const* sp_shard_dev, double* const* gp_dev);
    // invtrans_many + per transform, on the communication stream (i.e. beside the Legendre stage of the transforms that
    // follow): its grid points transposed into the owned part of a StructuredColumns field [size_halo][nb_fields] whose
    // partition is this rank's latitude band, then that field's halo exchange between the ranks (HaloExchange.h:191-219).
    // The Trans stream waits for the last exchange.
    void invtrans_many_halo(int ntransforms, int nb_fields, const double* const* sp_dev, double* const* gp_dev,
                            parallel::HaloExchange& hx, double* const* field_dev);
    hipStream_t comm_stream() const { return comm_stream_; }
    // largest message of the transposition in doubles (default 512 MiB).  COLLECTIVE: called by every rank of the communicator with
    // the same value (both ends of a pair cut their runs alike) -- the ranks compare it inside the call and a mismatch throws on every
    // rank; takes effect at the next transform (the message list is rebuilt).
    void set_max_message_elems(int64_t elems);
    int64_t max_message_elems() const { return max_message_elems_; }
    const PackedTransposePlan& packed_plan() const { return pplan_; }
    // with Trans profiling on: HIP events on the communication stream around the pack kernel and around the send / receive group of
    // every transform (a multi-GPU bench line shows where the curve bends).  The exchange time of a rank includes
    // waiting for its peers to post their side.  bytes_*: per transform, this rank.
    struct ExchangeTimings {
        double pack_ms = 0, exchange_ms = 0;
        int calls = 0;
        int64_t bytes_sent_off_device = 0, bytes_received_off_device = 0, bytes_largest_peer = 0;
        int peers = 0;
    };
    ExchangeTimings exchange_timings(bool reset);

private:
    struct Slot {
        double* F = nullptr;   // this rank's intermediate (all rows, its wavenumbers; Legendre kernel layout)
        double* S = nullptr;   // packed copy that is sent
        double* R = nullptr;   // received runs, one per source rank
        hipEvent_t legendre_done, exchange_done, fourier_done;
        bool used = false;
    };
    void ensure(int nb_fields);
    void check_ranks_agree(int nb_fields);
    bool check_every_call_ = false;               // ATLAS_AMD_DIST_CHECK=always
    std::set<std::pair<int, int64_t>> checked_;   // (field count, message limit) pairs the ranks have compared
    void legendre(int nb_fields, const double* sp_dev, Slot& s);
    bool sharded_input_ = false;   // set for the duration of invtrans_many_sharded
    void poison(Slot& s);
    void exchange(Slot& s);
    void fourier(int nb_fields, Slot& s, double* gp_dev);
    void halo(int nb_fields, Slot& s, const double* gp_dev, parallel::HaloExchange& hx, double* field_dev);

    std::vector<hipEvent_t> xev_;   // triples (before pack, after pack, after the exchange) of the transforms since the last reset
    size_t xev_used_ = 0;
    ExchangeTimings xt_;

    Trans& trans_;
    parallel::Comm& comm_;
    hipStream_t comm_stream_ = nullptr;
    int nf_cap_ = 0, nf_plan_ = 0, RP_ = 0;
    size_t capF_ = 0, capS_ = 0, capR_ = 0;   // doubles allocated for F, S, R of each slot (high-water marks)
    bool clip_noted_ = false;
    int64_t max_message_elems_ = int64_t(1) << 26;
    int64_t msgs_limit_        = 0;     // the limit msgs_ was built with
    bool poison_               = false; // ATLAS_AMD_DIST_POISON=1 (tests): F, S, R are filled with NaN before every transform
    PackedTransposePlan pplan_;
    std::vector<TransposeMsg> msgs_;
    long long* d_rowoff_src_ = nullptr;   // [nlats + 1]: rowoff of this rank as a source (pack kernel)
    long long* d_rowoff_dst_ = nullptr;   // [nparts][rows of my band]: row offsets inside each received run (Fourier kernels)
    long long* d_rowbase_    = nullptr;   // [rows of my band][nparts]: the same combined with the runs' offsets inside R: one read per mode [r5]
    bool use_rowbase_        = true;      // ATLAS_AMD_DIST_ROWBASE=0: the piece-table walk (A/B)
    int* d_kept_             = nullptr;   // [nlats]: kept(part, lat)
    Slot slot_[2];
};

}  // namespace trans
}  // namespace atlas_amd
