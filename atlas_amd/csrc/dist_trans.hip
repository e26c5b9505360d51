// See dist_trans.h.
#include "dist_trans.h"

#include <algorithm>
#include <sstream>
#include <stdexcept>

namespace atlas_amd {
namespace trans {

namespace {
void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        std::ostringstream ss;
        ss << "HIP error '" << hipGetErrorString(e) << "' in " << what << " (" << file << ":" << line << ")";
        throw std::runtime_error(ss.str());
    }
}
#define HIP_CHECK(x) hip_check((x), #x, __FILE__, __LINE__)
}  // namespace

TransposePlan make_transpose_plan(int T, int RP, const std::vector<int>& bands, int nparts, int part) {
    if ((int)bands.size() != nparts + 1 || part < 0 || part >= nparts) {
        throw std::invalid_argument("make_transpose_plan: bands / part");
    }
    TransposePlan p;
    p.cnt.resize(nparts);
    p.rows.resize(nparts);
    p.out_offsets.resize(nparts);
    for (int q = 0; q < nparts; ++q) {
        p.cnt[q]  = q <= T ? (T - q) / nparts + 1 : 0;   // |{m in [0, T] : m % nparts == q}|
        p.rows[q] = bands[q + 1] - bands[q];
    }
    int64_t off = 0;
    for (int q = 0; q < nparts; ++q) {
        p.out_offsets[q] = off;
        off += (int64_t)p.rows[part] * p.cnt[q] * RP;
    }
    p.out_total = off;
    return p;
}

std::vector<TransposeMsg> transpose_messages(const TransposePlan& plan, const std::vector<int>& bands, int RP, int nparts,
                                             int part, int64_t max_message_elems) {
    const int64_t biggest = (int64_t)*std::max_element(plan.rows.begin(), plan.rows.end()) *
                            *std::max_element(plan.cnt.begin(), plan.cnt.end()) * RP;
    int64_t K = std::max<int64_t>(1, (biggest + max_message_elems - 1) / max_message_elems);
    int minrows = 0;
    for (int r : plan.rows) {
        if (r > 0 && (minrows == 0 || r < minrows)) {
            minrows = r;
        }
    }
    K = std::max<int64_t>(1, std::min<int64_t>(K, std::max(minrows, 1)));
    std::vector<TransposeMsg> msgs;
    for (int64_t k = 0; k < K; ++k) {
        for (int peer = 0; peer < nparts; ++peer) {
            const int64_t s0 = plan.rows[peer] * k / K, s1 = plan.rows[peer] * (k + 1) / K;   // rows of the peer's band I send
            const int64_t r0 = plan.rows[part] * k / K, r1 = plan.rows[part] * (k + 1) / K;   // rows of my band I receive
            TransposeMsg m;
            m.peer       = peer;
            m.send_begin = (bands[peer] + s0) * plan.cnt[part] * RP;
            m.send_end   = (bands[peer] + s1) * plan.cnt[part] * RP;
            m.recv_begin = plan.out_offsets[peer] + r0 * plan.cnt[peer] * RP;
            m.recv_end   = plan.out_offsets[peer] + r1 * plan.cnt[peer] * RP;
            msgs.push_back(m);
        }
    }
    return msgs;
}

DistributedTrans::DistributedTrans(Trans& trans, parallel::Comm& comm) : trans_(trans), comm_(comm) {
    if (trans.nparts() != comm.size() || trans.part() != comm.rank()) {
        throw std::invalid_argument("DistributedTrans: the Trans must be made with (nparts, part) = (comm size, comm rank)");
    }
    if (trans.nparts() > 1 && trans.fourier_parts() != trans.nparts()) {
        throw std::invalid_argument("DistributedTrans: the Trans must be sharded by wavenumber (shard = m)");
    }
    HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
    for (Slot& s : slot_) {
        HIP_CHECK(hipEventCreateWithFlags(&s.legendre_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s.exchange_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s.fourier_done, hipEventDisableTiming));
    }
}

DistributedTrans::~DistributedTrans() {
    (void)hipStreamSynchronize(comm_stream_);
    (void)hipStreamSynchronize(trans_.stream());
    for (Slot& s : slot_) {
        if (s.F) {
            (void)hipFree(s.F);
        }
        if (s.R) {
            (void)hipFree(s.R);
        }
        (void)hipEventDestroy(s.legendre_done);
        (void)hipEventDestroy(s.exchange_done);
        (void)hipEventDestroy(s.fourier_done);
    }
    (void)hipStreamDestroy(comm_stream_);
}

void DistributedTrans::ensure(int nb_fields) {
    if (nb_fields <= nf_cap_ && RP_ == trans_.fourier_row_pitch(nb_fields)) {
        return;
    }
    HIP_CHECK(hipStreamSynchronize(comm_stream_));
    trans_.synchronize();
    RP_   = trans_.fourier_row_pitch(nb_fields);
    plan_ = make_transpose_plan(trans_.truncation(), RP_, trans_.bands(), trans_.nparts(), trans_.part());
    msgs_ = transpose_messages(plan_, trans_.bands(), RP_, trans_.nparts(), trans_.part(), max_message_elems);
    for (Slot& s : slot_) {
        if (s.F) {
            HIP_CHECK(hipFree(s.F));
            s.F = nullptr;
        }
        if (s.R) {
            HIP_CHECK(hipFree(s.R));
            s.R = nullptr;
        }
        HIP_CHECK(hipMalloc((void**)&s.F, std::max<size_t>(trans_.fourier_doubles(nb_fields), 1) * sizeof(double)));
        HIP_CHECK(hipMalloc((void**)&s.R, std::max<int64_t>(plan_.out_total, 1) * sizeof(double)));
        s.used = false;
    }
    nf_cap_ = nb_fields;
}

void DistributedTrans::legendre(int nb_fields, const double* sp_dev, Slot& s) {
    trans_.legendre_device(trans_.truncation(), nb_fields, sp_dev, s.F);
    HIP_CHECK(hipEventRecord(s.legendre_done, trans_.stream()));
}

void DistributedTrans::exchange(Slot& s) {
    HIP_CHECK(hipStreamWaitEvent(comm_stream_, s.legendre_done, 0));
    if (s.used) {
        HIP_CHECK(hipStreamWaitEvent(comm_stream_, s.fourier_done, 0));   // R of this slot is still being read
    }
    std::vector<parallel::Msg> sends, recvs;
    for (const TransposeMsg& m : msgs_) {
        if (m.send_end > m.send_begin) {
            sends.push_back(parallel::Msg{m.peer, s.F + m.send_begin, size_t(m.send_end - m.send_begin) * sizeof(double)});
        }
        if (m.recv_end > m.recv_begin) {
            recvs.push_back(parallel::Msg{m.peer, s.R + m.recv_begin, size_t(m.recv_end - m.recv_begin) * sizeof(double)});
        }
    }
    comm_.exchange(sends, recvs, comm_stream_);
    HIP_CHECK(hipEventRecord(s.exchange_done, comm_stream_));
}

void DistributedTrans::fourier(int nb_fields, Slot& s, double* gp_dev) {
    HIP_CHECK(hipStreamWaitEvent(trans_.stream(), s.exchange_done, 0));
    const int P = trans_.nparts();
    std::vector<const double*> base(P);
    for (int p = 0; p < P; ++p) {
        base[p] = s.R + plan_.out_offsets[p];
    }
    trans_.fourier_device(nb_fields, 0, base.data(), plan_.cnt.data(), gp_dev);
    HIP_CHECK(hipEventRecord(s.fourier_done, trans_.stream()));
    s.used = true;
}

void DistributedTrans::invtrans(int nb_fields, const double* sp_dev, double* gp_dev) {
    const double* sp[1] = {sp_dev};
    double* gp[1]       = {gp_dev};
    invtrans_many(1, nb_fields, sp, gp);
}

void DistributedTrans::invtrans_many(int ntransforms, int nb_fields, const double* const* sp_dev, double* const* gp_dev) {
    if (ntransforms <= 0 || nb_fields <= 0) {
        return;
    }
    ensure(nb_fields);
    // L(0) X(0) | L(1) X(1) F(0) | L(2) X(2) F(1) | ... | F(n-1):  the Legendre stage of transform i needs the F buffer of
    // its slot, last sent by exchange i-2, which the Fourier stage of i-2 (earlier on the Trans stream) has waited for.
    for (int i = 0; i < ntransforms; ++i) {
        Slot& s = slot_[i & 1];
        legendre(nb_fields, sp_dev[i], s);
        exchange(s);
        if (i > 0) {
            fourier(nb_fields, slot_[(i - 1) & 1], gp_dev[i - 1]);
        }
    }
    fourier(nb_fields, slot_[(ntransforms - 1) & 1], gp_dev[ntransforms - 1]);
}

hipError_t launch_gp_to_field(const double* gp, double* field, long long npts, int nf, hipStream_t stream);

void DistributedTrans::halo(int nb_fields, Slot& s, const double* gp_dev, parallel::HaloExchange& hx, double* field_dev) {
    HIP_CHECK(hipStreamWaitEvent(comm_stream_, s.fourier_done, 0));
    const long long npts = trans_.nb_gridpoints();
    HIP_CHECK(launch_gp_to_field(gp_dev, field_dev, npts, nb_fields, comm_stream_));
    const int shape[2]         = {hx.plan().parsize, nb_fields};
    const long long strides[2] = {nb_fields, 1};
    hx.execute_comm_on(comm_, parallel::HALO_DOUBLE, field_dev, hx.describe(2, shape, strides, 0), false, comm_stream_);
}

void DistributedTrans::invtrans_many_halo(int ntransforms, int nb_fields, const double* const* sp_dev, double* const* gp_dev,
                                          parallel::HaloExchange& hx, double* const* field_dev) {
    if (ntransforms <= 0 || nb_fields <= 0) {
        return;
    }
    if (!hx.is_setup() || hx.plan().parsize < trans_.nb_gridpoints()) {
        throw std::invalid_argument("invtrans_many_halo: the halo exchange must be set up on a partition that owns this rank's band");
    }
    ensure(nb_fields);
    // Trans stream:  L(0) | L(1) F(0) | L(2) F(1) | ...           communication stream:  X(0) | X(1) H(0) | X(2) H(1) | ...
    // H(i-1) waits for F(i-1), which precedes L(i+1) on the Trans stream: the exchange runs beside that Legendre stage.
    for (int i = 0; i < ntransforms; ++i) {
        Slot& s = slot_[i & 1];
        legendre(nb_fields, sp_dev[i], s);
        exchange(s);
        if (i > 0) {
            fourier(nb_fields, slot_[(i - 1) & 1], gp_dev[i - 1]);
            halo(nb_fields, slot_[(i - 1) & 1], gp_dev[i - 1], hx, field_dev[i - 1]);
        }
    }
    Slot& last = slot_[(ntransforms - 1) & 1];
    fourier(nb_fields, last, gp_dev[ntransforms - 1]);
    halo(nb_fields, last, gp_dev[ntransforms - 1], hx, field_dev[ntransforms - 1]);
    // the call completes on the Trans stream
    HIP_CHECK(hipEventRecord(last.exchange_done, comm_stream_));
    HIP_CHECK(hipStreamWaitEvent(trans_.stream(), last.exchange_done, 0));
}

}  // namespace trans
}  // namespace atlas_amd
