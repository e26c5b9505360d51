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
    // invtrans_many + per transform, on the communication stream (i.e. beside the Legendre stage of the transforms that
    // follow): its grid points transposed into the owned part of a StructuredColumns field [size_halo][nb_fields] whose
    // partition is this rank's latitude band, then that field's halo exchange between the ranks (HaloExchange.h:191-219).
    // The Trans stream waits for the last exchange.
    void invtrans_many_halo(int ntransforms, int nb_fields, const double* const* sp_dev, double* const* gp_dev,
                            parallel::HaloExchange& hx, double* const* field_dev);
    hipStream_t comm_stream() const { return comm_stream_; }
    int64_t max_message_elems = int64_t(1) << 26;   // 512 MiB of doubles

private:
    struct Slot {
        double* F = nullptr;
        double* R = nullptr;
        hipEvent_t legendre_done, exchange_done, fourier_done;
        bool used = false;
    };
    void ensure(int nb_fields);
    void legendre(int nb_fields, const double* sp_dev, Slot& s);
    void exchange(Slot& s);
    void fourier(int nb_fields, Slot& s, double* gp_dev);
    void halo(int nb_fields, Slot& s, const double* gp_dev, parallel::HaloExchange& hx, double* field_dev);

    Trans& trans_;
    parallel::Comm& comm_;
    hipStream_t comm_stream_ = nullptr;
    int nf_cap_ = 0, RP_ = 0;
    TransposePlan plan_;
    std::vector<TransposeMsg> msgs_;
    Slot slot_[2];
};

}  // namespace trans
}  // namespace atlas_amd
