// See dist_trans.h.
#include "env.h"
#include "dist_trans.h"

#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace atlas_amd {
namespace trans {

namespace {
void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        std::ostringstream ss;
        ss << "HIP error '" << hipGetErrorString(e) << "' in " << what << " (" << file << ":" << line << ")";
        (void)hipGetLastError();   // the failure is reported here, once: not left sticky for the launch checks of the next call
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
    if (max_message_elems < 1) {
        throw std::invalid_argument("transpose_messages: max_message_elems must be at least 1");
    }
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

PackedTransposePlan make_packed_transpose_plan(const std::vector<int>& row_mmax, int cols, const std::vector<int>& bands,
                                               int nparts, int part) {
    const int nlats = (int)row_mmax.size();
    if ((int)bands.size() != nparts + 1 || part < 0 || part >= nparts || cols <= 0 || bands[0] != 0 || bands[nparts] != nlats) {
        throw std::invalid_argument("make_packed_transpose_plan: bands / part / cols");
    }
    PackedTransposePlan pl;
    pl.cols = cols;
    pl.rowoff.assign(nparts, std::vector<int64_t>(nlats + 1, 0));
    for (int p = 0; p < nparts; ++p) {
        for (int lat = 0; lat < nlats; ++lat) {
            const int mm     = row_mmax[lat];
            const int kept   = mm >= p ? (mm - p) / nparts + 1 : 0;   // |{m in [0, mmax] : m % nparts == p}|
            pl.rowoff[p][lat + 1] = pl.rowoff[p][lat] + (int64_t)kept * cols;
        }
    }
    pl.out_offsets.resize(nparts);
    int64_t off = 0;
    for (int p = 0; p < nparts; ++p) {
        pl.out_offsets[p] = off;
        off += pl.rowoff[p][bands[part + 1]] - pl.rowoff[p][bands[part]];
    }
    pl.out_total  = off;
    pl.send_total = pl.rowoff[part][nlats];
    return pl;
}

std::vector<TransposeMsg> packed_transpose_messages(const PackedTransposePlan& pl, const std::vector<int>& bands, int nparts,
                                                    int part, int64_t max_message_elems) {
    if (max_message_elems < 1) {
        throw std::invalid_argument("packed_transpose_messages: max_message_elems must be at least 1");
    }
    // K pieces per pair, the same for every pair: from the largest run of any (source, destination) -- global quantities
    int64_t biggest = 0;
    int minrows     = 0;
    for (int q = 0; q < nparts; ++q) {
        const int rows = bands[q + 1] - bands[q];
        if (rows > 0 && (minrows == 0 || rows < minrows)) {
            minrows = rows;
        }
        for (int p = 0; p < nparts; ++p) {
            biggest = std::max(biggest, pl.rowoff[p][bands[q + 1]] - pl.rowoff[p][bands[q]]);
        }
    }
    int64_t K = std::max<int64_t>(1, (biggest + max_message_elems - 1) / max_message_elems);
    K         = std::max<int64_t>(1, std::min<int64_t>(K, std::max(minrows, 1)));
    // the pieces of a run are cut at equal ROW counts and the rows of a band differ in kept wavenumbers (equatorial rows are the
    // longest): raise K until the largest piece of any pair honours the limit, or a piece is a single row [r4].  Global quantities
    // again (the plan holds the row offsets of every rank): both ends of a pair arrive at the same K.
    auto largest_piece = [&](int64_t kk) {
        int64_t worst = 0;
        for (int q = 0; q < nparts; ++q) {
            const int rows = bands[q + 1] - bands[q];
            for (int p = 0; p < nparts; ++p) {
                for (int64_t k = 0; k < kk; ++k) {
                    const int a = (int)(rows * k / kk), b = (int)(rows * (k + 1) / kk);
                    worst       = std::max(worst, pl.rowoff[p][bands[q] + b] - pl.rowoff[p][bands[q] + a]);
                }
            }
        }
        return worst;
    };
    while (K < std::max(minrows, 1) && largest_piece(K) > max_message_elems) {
        ++K;
    }
    std::vector<TransposeMsg> msgs;
    const int myrows = bands[part + 1] - bands[part];
    for (int64_t k = 0; k < K; ++k) {
        for (int peer = 0; peer < nparts; ++peer) {
            const int prow = bands[peer + 1] - bands[peer];
            const int s0 = (int)(prow * k / K), s1 = (int)(prow * (k + 1) / K);       // rows of the peer's band I send
            const int r0 = (int)(myrows * k / K), r1 = (int)(myrows * (k + 1) / K);   // rows of my band I receive
            TransposeMsg m;
            m.peer       = peer;
            m.send_begin = pl.rowoff[part][bands[peer] + s0];
            m.send_end   = pl.rowoff[part][bands[peer] + s1];
            m.recv_begin = pl.out_offsets[peer] + pl.rowoff[peer][bands[part] + r0] - pl.rowoff[peer][bands[part]];
            m.recv_end   = pl.out_offsets[peer] + pl.rowoff[peer][bands[part] + r1] - pl.rowoff[peer][bands[part]];
            msgs.push_back(m);
        }
    }
    return msgs;
}

// Doubles per packed (row, wavenumber) record.  The live columns are 2 nf; padded to the intermediate's own pitch (a multiple of 16
// doubles = 128 bytes) every record starts on a line boundary and the 16-byte pieces of 8 neighbouring fields share ONE line, as in
// the single-device layout -- with 2 nf = 274 a field group's 128 bytes straddle two lines for most wavenumbers and the gather of
// the Fourier rows, which is bound by line fills, pays for it.  Measured per rank (tools/scaling_model.py, round 5): padded records take
// 2 % off the Fourier stage at P = 2 (3.66 -> 3.58 ms), nothing at P = 8, and put 5 % more bytes on the wire -- where P = 2 and 4 are
// bound: the default stays the unpadded record; ATLAS_AMD_DIST_PACK_PAD=1 pads.
static int packed_cols_for(const Trans& trans, int nb_fields) {
    bool pad = false;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_DIST_PACK_PAD")) {
        pad = atoi(e) != 0;
    }
    return pad ? trans.fourier_row_pitch(nb_fields) : 2 * nb_fields;
}

// S[rowoff[lat] + ml * cols + c] = F[(lat * cnt + ml) * RP + c]: one workgroup per row, 16-byte pieces (cols, RP and the
// row offsets are even)
__global__ void __launch_bounds__(256) pack_rows_kernel(const double* __restrict__ F, double* __restrict__ S,
                                                        const long long* __restrict__ rowoff, const int* __restrict__ kept,
                                                        int cnt, int RP, int cols) {
    const int lat   = blockIdx.x;
    const int k     = kept[lat];
    const int c2    = cols >> 1;
    const int total = k * c2;
    const double2* src = reinterpret_cast<const double2*>(F + (long long)lat * cnt * RP);
    double2* dst       = reinterpret_cast<double2*>(S + rowoff[lat]);
    const int rp2      = RP >> 1;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < total; e += blockDim.x * gridDim.y) {
        const int ml = e / c2;
        const int c  = e - ml * c2;
        dst[e]       = src[(long long)ml * rp2 + c];
    }
}

// Measurement aid (tools/scaling_model.py): the pack kernel of ONE rank of a P-rank transform, alone on the device -- the rank's
// intermediate F (uninitialised: the kernel only copies) packed into its send buffer S `reps` times between two events on the
// Trans stream.  No communicator: the plan is host arithmetic.  Returns ms per launch, *bytes = bytes written to S per launch.
double pack_probe(Trans& trans, int nb_fields, int reps, int64_t* bytes) {
    const TransGeometry& geo = trans.geometry();
    std::vector<int> row_mmax(geo.nlats);
    for (int j = 0; j < geo.nlats; ++j) {
        const int jleg = j < geo.nlatsNH ? j : geo.nlats - 1 - j;
        row_mmax[j]    = std::min(geo.mmax_leg[jleg], geo.T);
    }
    const int RP = trans.fourier_row_pitch(nb_fields);
    const PackedTransposePlan plan = make_packed_transpose_plan(row_mmax, packed_cols_for(trans, nb_fields), trans.bands(), trans.nparts(), trans.part());
    std::vector<long long> src(plan.rowoff[trans.part()].begin(), plan.rowoff[trans.part()].end());
    std::vector<int> kept(geo.nlats);
    for (int j = 0; j < geo.nlats; ++j) {
        kept[j] = (int)((src[j + 1] - src[j]) / plan.cols);
    }
    double *F = nullptr, *S = nullptr;
    long long* d_src = nullptr;
    int* d_kept      = nullptr;
    HIP_CHECK(hipMalloc((void**)&F, std::max<size_t>(trans.fourier_doubles(nb_fields), 1) * sizeof(double)));
    HIP_CHECK(hipMalloc((void**)&S, (size_t)std::max<int64_t>(plan.send_total, 1) * sizeof(double)));
    HIP_CHECK(hipMalloc((void**)&d_src, src.size() * sizeof(long long)));
    HIP_CHECK(hipMalloc((void**)&d_kept, kept.size() * sizeof(int)));
    HIP_CHECK(hipMemcpy(d_src, src.data(), src.size() * sizeof(long long), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_kept, kept.data(), kept.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(F, 0, std::max<size_t>(trans.fourier_doubles(nb_fields), 1) * sizeof(double)));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: warm-up
        HIP_CHECK(hipEventRecord(e0, trans.stream()));
        for (int i = 0; i < (pass ? reps : 1); ++i) {
            hipLaunchKernelGGL(pack_rows_kernel, dim3(geo.nlats, 4), dim3(256), 0, trans.stream(), F, S, d_src, d_kept,
                               trans.owned_wavenumbers(), RP, plan.cols);
        }
        HIP_CHECK(hipEventRecord(e1, trans.stream()));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(F);
    (void)hipFree(S);
    (void)hipFree(d_src);
    (void)hipFree(d_kept);
    if (bytes) {
        *bytes = plan.send_total * (int64_t)sizeof(double);
    }
    return ms / std::max(reps, 1);
}

// Measurement aid (tools/scaling_model.py): the Fourier stage of ONE rank of a P-rank transform as the distributed transform runs it
// -- on the rank's latitude band, reading the P packed runs of the receive buffer through the per-row offsets and the piece table --
// alone on the device.  The receive buffer holds zeros (the stage's time does not depend on the values).  ms per stage.
double fourier_packed_probe(Trans& trans, int nb_fields, int reps) {
    const TransGeometry& geo = trans.geometry();
    std::vector<int> row_mmax(geo.nlats);
    for (int j = 0; j < geo.nlats; ++j) {
        const int jleg = j < geo.nlatsNH ? j : geo.nlats - 1 - j;
        row_mmax[j]    = std::min(geo.mmax_leg[jleg], geo.T);
    }
    const int P = trans.nparts(), part = trans.part();
    const PackedTransposePlan plan = make_packed_transpose_plan(row_mmax, packed_cols_for(trans, nb_fields), trans.bands(), P, part);
    const int b0 = trans.bands()[part], b1 = trans.bands()[part + 1];
    const int rows = std::max(b1 - b0, 1);
    std::vector<long long> dst((size_t)P * rows, 0);
    for (int p = 0; p < P; ++p) {
        for (int r = 0; r < b1 - b0; ++r) {
            dst[(size_t)p * rows + r] = plan.rowoff[p][b0 + r] - plan.rowoff[p][b0];
        }
    }
    double *Rbuf = nullptr, *gp = nullptr;
    long long* d_dst = nullptr;
    const size_t nR  = (size_t)std::max<int64_t>(plan.out_total, 1);
    const size_t ngp = (size_t)std::max<int64_t>((int64_t)nb_fields * trans.nb_gridpoints(), 1);
    HIP_CHECK(hipMalloc((void**)&Rbuf, nR * sizeof(double)));
    HIP_CHECK(hipMalloc((void**)&gp, ngp * sizeof(double)));
    HIP_CHECK(hipMalloc((void**)&d_dst, dst.size() * sizeof(long long)));
    HIP_CHECK(hipMemset(Rbuf, 0, nR * sizeof(double)));
    HIP_CHECK(hipMemcpy(d_dst, dst.data(), dst.size() * sizeof(long long), hipMemcpyHostToDevice));
    std::vector<const double*> base(P);
    std::vector<const long long*> rowoff(P);
    for (int p = 0; p < P; ++p) {
        base[p]   = Rbuf + plan.out_offsets[p];
        rowoff[p] = d_dst + (size_t)p * rows;
    }
    std::vector<long long> rowbase((size_t)rows * P, 0);
    for (int r = 0; r < b1 - b0; ++r) {
        for (int p = 0; p < P; ++p) {
            rowbase[(size_t)r * P + p] = (plan.out_offsets[p] - plan.out_offsets[0]) + dst[(size_t)p * rows + r];
        }
    }
    long long* d_rowbase = nullptr;
    HIP_CHECK(hipMalloc((void**)&d_rowbase, rowbase.size() * sizeof(long long)));
    HIP_CHECK(hipMemcpy(d_rowbase, rowbase.data(), rowbase.size() * sizeof(long long), hipMemcpyHostToDevice));
    const char* rb_env    = atlas_amd::env_get("ATLAS_AMD_DIST_ROWBASE");
    const bool use_rowbase = !(rb_env && atoi(rb_env) == 0);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: warm-up
        const int n = pass ? reps : 3;
        HIP_CHECK(hipEventRecord(e0, trans.stream()));
        for (int i = 0; i < n; ++i) {
            trans.fourier_device_packed(nb_fields, 0, base.data(), rowoff.data(), plan.cols, gp, use_rowbase ? d_rowbase : nullptr);
        }
        HIP_CHECK(hipEventRecord(e1, trans.stream()));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    trans.synchronize();
    trans.clear_fourier_parts_cache();   // keyed by the buffers freed below
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(Rbuf);
    (void)hipFree(gp);
    (void)hipFree(d_dst);
    (void)hipFree(d_rowbase);
    return ms / std::max(reps, 1);
}

DistributedTrans::DistributedTrans(Trans& trans, parallel::Comm& comm) : trans_(trans), comm_(comm) {
    if (trans.nparts() != comm.size() || trans.part() != comm.rank()) {
        throw std::invalid_argument("DistributedTrans: the Trans must be made with (nparts, part) = (comm size, comm rank)");
    }
    if (trans.nparts() > 1 && trans.fourier_parts() != trans.nparts()) {
        throw std::invalid_argument("DistributedTrans: the Trans must be sharded by wavenumber (shard = m)");
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_DIST_POISON")) {
        poison_ = atoi(e) != 0;
    }
    // ATLAS_AMD_DIST_CHECK=always (debugging a hang or misplaced rows): the ranks compare (field count, message limit) on EVERY
    // call, not only the first time a pair is used -- ranks that later call with different, individually already-compared field
    // counts are then caught as well; costs a blocking 3-int all-to-all per call, hence not the default
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_DIST_CHECK")) {
        check_every_call_ = std::string(e) == "always";
    }
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_DIST_ROWBASE")) {   // A/B: 0 = the piece-table walk of round 3
        use_rowbase_ = atoi(e) != 0;
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
        if (s.S) {
            (void)hipFree(s.S);
        }
        (void)hipEventDestroy(s.legendre_done);
        (void)hipEventDestroy(s.exchange_done);
        (void)hipEventDestroy(s.fourier_done);
    }
    for (hipEvent_t e : xev_) {
        (void)hipEventDestroy(e);
    }
    (void)hipStreamDestroy(comm_stream_);
    (void)hipFree(d_rowoff_src_);
    (void)hipFree(d_rowoff_dst_);
    (void)hipFree(d_rowbase_);
    (void)hipFree(d_kept_);
}

void DistributedTrans::set_max_message_elems(int64_t elems) {
    // COLLECTIVE: every rank of the communicator calls this, with the same value -- compared here, where all of them are, and not
    // inside the next transform (a check that only the ranks with a changed setting enter pairs its all-to-all with
    // the other ranks' data messages).  An invalid value takes part in the comparison too and is refused AFTER it (a
    // rank that throws before the collective leaves the others waiting in it): the ranks that passed a valid value get the
    // "another message limit" error, this one the argument error, nobody hangs and nobody's limit changes.
    const int64_t previous = max_message_elems_;
    max_message_elems_     = elems;   // ensure() rebuilds the message list when it differs from the one in use
    try {
        check_ranks_agree(0);
    }
    catch (...) {
        max_message_elems_ = previous;
        if (elems >= 1) {
            throw;
        }
    }
    if (elems < 1) {
        max_message_elems_ = previous;
        throw std::invalid_argument("DistributedTrans: the message limit must be at least one element");
    }
}

// the ranks compare (message limit, field count): an all-to-all of three ints; every rank takes part or none
void DistributedTrans::check_ranks_agree(int nb_fields) {
    HIP_CHECK(hipStreamSynchronize(comm_stream_));
    const int P       = comm_.size();
    const int mine[3] = {(int)(max_message_elems_ & 0x7fffffff), (int)(max_message_elems_ >> 31), nb_fields};
    std::vector<int> send((size_t)P * 3), recv((size_t)P * 3, 0);
    for (int p = 0; p < P; ++p) {
        std::copy(mine, mine + 3, send.begin() + (size_t)p * 3);
    }
    comm_.all_to_all(send.data(), recv.data(), 3);
    for (int p = 0; p < P; ++p) {
        if (!std::equal(mine, mine + 3, recv.begin() + (size_t)p * 3)) {
            throw std::runtime_error("DistributedTrans: rank " + std::to_string(p) + " uses another message limit / field count "
                                     "(set_max_message_elems is collective: the same value on every rank)");
        }
    }
    checked_.insert({nb_fields, max_message_elems_});
}

void DistributedTrans::ensure(int nb_fields) {
    if (check_every_call_) {
        check_ranks_agree(nb_fields);
    }
    const bool same_plan = nb_fields == nf_plan_;
    if (same_plan && msgs_limit_ == max_message_elems_) {
        return;
    }
    HIP_CHECK(hipStreamSynchronize(comm_stream_));
    trans_.synchronize();
    if (!same_plan) {
        const TransGeometry& geo = trans_.geometry();
        std::vector<int> row_mmax(geo.nlats);
        for (int j = 0; j < geo.nlats; ++j) {
            const int jleg = j < geo.nlatsNH ? j : geo.nlats - 1 - j;
            row_mmax[j]    = std::min(geo.mmax_leg[jleg], geo.T);
        }
        RP_    = trans_.fourier_row_pitch(nb_fields);
        pplan_ = make_packed_transpose_plan(row_mmax, packed_cols_for(trans_, nb_fields), trans_.bands(), trans_.nparts(), trans_.part());
        // device copies of the offsets: source side (pack kernel), destination side (Fourier kernels).  Their sizes do not
        // depend on the field count: allocated once, rewritten in place (both streams are idle here), so that the pointers the
        // Fourier stage's piece table is keyed by stay the same for every field count (a caller alternating between
        // 137-level and surface fields used to add a piece table per call and, past eight, a device-wide synchronisation)
        const int P = trans_.nparts(), part = trans_.part();
        const int b0 = trans_.bands()[part], b1 = trans_.bands()[part + 1];
        std::vector<long long> src(pplan_.rowoff[part].begin(), pplan_.rowoff[part].end());
        std::vector<int> kept(geo.nlats);
        for (int j = 0; j < geo.nlats; ++j) {
            kept[j] = (int)((src[j + 1] - src[j]) / pplan_.cols);
        }
        std::vector<long long> dst((size_t)P * std::max(b1 - b0, 1), 0);
        for (int p = 0; p < P; ++p) {
            for (int r = 0; r < b1 - b0; ++r) {
                dst[(size_t)p * (b1 - b0) + r] = pplan_.rowoff[p][b0 + r] - pplan_.rowoff[p][b0];
            }
        }
        // [local row][source rank]: where the row's run of that source starts, relative to the run of source 0 (all P runs live in one
        // receive buffer): what the Fourier kernels read once per mode
        std::vector<long long> rowbase((size_t)std::max(b1 - b0, 1) * P, 0);
        for (int r = 0; r < b1 - b0; ++r) {
            for (int p = 0; p < P; ++p) {
                rowbase[(size_t)r * P + p] = (pplan_.out_offsets[p] - pplan_.out_offsets[0]) + dst[(size_t)p * (b1 - b0) + r];
            }
        }
        if (!d_rowoff_src_) {
            HIP_CHECK(hipMalloc((void**)&d_rowoff_src_, src.size() * sizeof(long long)));
            HIP_CHECK(hipMalloc((void**)&d_rowoff_dst_, dst.size() * sizeof(long long)));
            HIP_CHECK(hipMalloc((void**)&d_rowbase_, rowbase.size() * sizeof(long long)));
            HIP_CHECK(hipMalloc((void**)&d_kept_, kept.size() * sizeof(int)));
        }
        HIP_CHECK(hipMemcpy(d_rowbase_, rowbase.data(), rowbase.size() * sizeof(long long), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_rowoff_src_, src.data(), src.size() * sizeof(long long), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_rowoff_dst_, dst.data(), dst.size() * sizeof(long long), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_kept_, kept.data(), kept.size() * sizeof(int), hipMemcpyHostToDevice));
        // F, S and R of both slots at their high-water sizes: reallocated only when a plan needs more
        const size_t needF = std::max<size_t>(trans_.fourier_doubles(nb_fields), 1);
        const size_t needS = (size_t)std::max<int64_t>(pplan_.send_total, 1);
        const size_t needR = (size_t)std::max<int64_t>(pplan_.out_total, 1);
        if (needF > capF_ || needS > capS_ || needR > capR_) {
            capF_ = std::max(capF_, needF);
            capS_ = std::max(capS_, needS);
            capR_ = std::max(capR_, needR);
            for (Slot& s : slot_) {
                for (double** ptr : {&s.F, &s.S, &s.R}) {
                    if (*ptr) {
                        HIP_CHECK(hipFree(*ptr));
                        *ptr = nullptr;
                    }
                }
                HIP_CHECK(hipMalloc((void**)&s.F, capF_ * sizeof(double)));
                HIP_CHECK(hipMalloc((void**)&s.S, capS_ * sizeof(double)));
                HIP_CHECK(hipMalloc((void**)&s.R, capR_ * sizeof(double)));
                s.used = false;
            }
            trans_.clear_fourier_parts_cache();   // keyed by the pointers that were just freed
        }
        nf_cap_  = nb_fields;
        nf_plan_ = nb_fields;
    }
    msgs_       = packed_transpose_messages(pplan_, trans_.bands(), trans_.nparts(), trans_.part(), max_message_elems_);
    msgs_limit_ = max_message_elems_;
    // both ends of every pair must cut their runs alike: the number of pieces per pair follows from the message limit and the
    // field count.  The limit is compared where it is set (set_max_message_elems, collective); a (field count, limit) pair is
    // compared across the ranks the FIRST time it is used -- every rank gets here then, because the transform itself is called by
    // all of them with one field count -- and remembered: a caller alternating between 137-level and surface fields pays the
    // blocking all-to-all twice, not per call
    if (!checked_.count({nb_fields, max_message_elems_})) {
        check_ranks_agree(nb_fields);
    }
    {
        int64_t biggest = 0;
        for (const TransposeMsg& m : msgs_) {
            biggest = std::max<int64_t>(biggest, m.send_end - m.send_begin);
        }
        if (biggest > max_message_elems_ && !clip_noted_) {   // K is capped by the rows of the smallest band
            clip_noted_ = true;
            std::fprintf(stderr, "[atlas_amd] DistributedTrans: message limit of %lld doubles cannot be honoured (largest message %lld): "
                                 "a pair's run is cut into at most as many pieces as the smallest latitude band has rows\n",
                         (long long)max_message_elems_, (long long)biggest);
        }
    }
}

// tests (ATLAS_AMD_DIST_POISON=1): every byte of the three buffers is NaN before the transform writes them, so a read of a
// slot nobody wrote -- or an uninitialised byte on the wire -- shows up in the result
void DistributedTrans::poison(Slot& s) {
    HIP_CHECK(hipMemsetAsync(s.F, 0xFF, capF_ * sizeof(double), trans_.stream()));
    HIP_CHECK(hipMemsetAsync(s.S, 0xFF, capS_ * sizeof(double), trans_.stream()));
    HIP_CHECK(hipMemsetAsync(s.R, 0xFF, capR_ * sizeof(double), trans_.stream()));
}

void DistributedTrans::legendre(int nb_fields, const double* sp_dev, Slot& s) {
    if (poison_) {
        if (s.used) {
            HIP_CHECK(hipStreamWaitEvent(trans_.stream(), s.exchange_done, 0));   // S of this slot may still be on the wire
        }
        poison(s);
    }
    if (sharded_input_) {
        trans_.legendre_device_sharded(nb_fields, sp_dev, s.F);
    }
    else {
        trans_.legendre_device(trans_.truncation(), nb_fields, sp_dev, s.F);
    }
    HIP_CHECK(hipEventRecord(s.legendre_done, trans_.stream()));
}

void DistributedTrans::exchange(Slot& s) {
    HIP_CHECK(hipStreamWaitEvent(comm_stream_, s.legendre_done, 0));
    if (s.used) {
        HIP_CHECK(hipStreamWaitEvent(comm_stream_, s.fourier_done, 0));   // R of this slot is still being read
    }
    const bool prof = trans_.profile();
    if (prof) {
        while (xev_.size() < xev_used_ + 3) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            xev_.push_back(e);
        }
        HIP_CHECK(hipEventRecord(xev_[xev_used_], comm_stream_));
    }
    // pack: the kept wavenumbers and live columns of every row, contiguous per destination band
    if (pplan_.send_total > 0) {
        const int nlats = trans_.geometry().nlats;
        hipLaunchKernelGGL(pack_rows_kernel, dim3(nlats, 4), dim3(256), 0, comm_stream_, s.F, s.S, d_rowoff_src_, d_kept_,
                           trans_.owned_wavenumbers(), RP_, pplan_.cols);
        HIP_CHECK(hipGetLastError());
    }
    if (prof) {
        HIP_CHECK(hipEventRecord(xev_[xev_used_ + 1], comm_stream_));
    }
    std::vector<parallel::Msg> sends, recvs;
    for (const TransposeMsg& m : msgs_) {
        if (m.send_end > m.send_begin) {
            sends.push_back(parallel::Msg{m.peer, s.S + m.send_begin, size_t(m.send_end - m.send_begin) * sizeof(double)});
        }
        if (m.recv_end > m.recv_begin) {
            recvs.push_back(parallel::Msg{m.peer, s.R + m.recv_begin, size_t(m.recv_end - m.recv_begin) * sizeof(double)});
        }
    }
    comm_.exchange(sends, recvs, comm_stream_);
    HIP_CHECK(hipEventRecord(s.exchange_done, comm_stream_));
    if (prof) {
        HIP_CHECK(hipEventRecord(xev_[xev_used_ + 2], comm_stream_));
        xev_used_ += 3;
    }
}

DistributedTrans::ExchangeTimings DistributedTrans::exchange_timings(bool reset) {
    HIP_CHECK(hipStreamSynchronize(comm_stream_));
    for (size_t i = 0; i + 3 <= xev_used_; i += 3) {
        float a = 0, b = 0;
        HIP_CHECK(hipEventElapsedTime(&a, xev_[i], xev_[i + 1]));
        HIP_CHECK(hipEventElapsedTime(&b, xev_[i + 1], xev_[i + 2]));
        xt_.pack_ms += a;
        xt_.exchange_ms += b;
        xt_.calls += 1;
    }
    xev_used_ = 0;
    ExchangeTimings out = xt_;
    // what one transform moves between this rank and its peers (the message list in use)
    std::vector<int64_t> to_peer(comm_.size(), 0);
    for (const TransposeMsg& m : msgs_) {
        if (m.peer != comm_.rank()) {
            to_peer[m.peer] += (m.send_end - m.send_begin) * (int64_t)sizeof(double);
            out.bytes_received_off_device += (m.recv_end - m.recv_begin) * (int64_t)sizeof(double);
        }
    }
    for (int64_t b : to_peer) {
        out.bytes_sent_off_device += b;
        out.bytes_largest_peer = std::max(out.bytes_largest_peer, b);
        out.peers += b > 0;
    }
    if (reset) {
        xt_ = ExchangeTimings();
    }
    return out;
}

void DistributedTrans::fourier(int nb_fields, Slot& s, double* gp_dev) {
    HIP_CHECK(hipStreamWaitEvent(trans_.stream(), s.exchange_done, 0));
    const int P    = trans_.nparts();
    const int rows = trans_.band_end() - trans_.band_begin();
    std::vector<const double*> base(P);
    std::vector<const long long*> rowoff(P);
    for (int p = 0; p < P; ++p) {
        base[p]   = s.R + pplan_.out_offsets[p];
        rowoff[p] = d_rowoff_dst_ + (size_t)p * rows;
    }
    trans_.fourier_device_packed(nb_fields, 0, base.data(), rowoff.data(), pplan_.cols, gp_dev, use_rowbase_ ? d_rowbase_ : nullptr);
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

void DistributedTrans::invtrans_many_sharded(int ntransforms, int nb_fields, const double* const* sp_shard_dev,
                                             double* const* gp_dev) {
    sharded_input_ = true;
    try {
        invtrans_many(ntransforms, nb_fields, sp_shard_dev, gp_dev);
    }
    catch (...) {
        sharded_input_ = false;
        throw;
    }
    sharded_input_ = false;
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
