#include "env.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace atlas_amd {
namespace {
using C = EnvClass;
// clang-format off
const EnvSwitch kSwitches[] = {
    // ---- behaviour: changes results (documented deviations)
    {"ATLAS_AMD_REFERENCE_POLES",   C::behaviour, "0",      "1: a row at latitude -90 of a no_nest / unstructured target reproduces the reference's arithmetic there (INTEGRATION.md, Deviations) instead of the mirror image of the north-pole row"},
    {"ATLAS_AMD_IGNORE_ENV",        C::behaviour, "0",      "non-zero: every other switch of this table reads as unset (same as atlas_amd__set_ignore_env(1), which the Atlas adapter calls)"},
    // ---- tuning: selects between implementations of the same arithmetic (bit-identical unless noted) or sizes a resource
    {"ATLAS_AMD_TABLES",            C::tuning,    "device", "host: Legendre tables generated on the host and uploaded (6.7 s at TL1279) instead of on the device (0.9 s); same bits"},
    {"ATLAS_AMD_PIPELINE",          C::tuning,    "1",      "n > 1: a call's fields in n chunks, Fourier stage of chunk i beside the Legendre stage of chunk i+1 on a second stream (measured no gain: one fp64 datapath)"},
    {"ATLAS_AMD_LEG_KERNEL",        C::tuning,    "lean",   "classic: the generic Legendre template instead of the hand-scheduled lean kernels (same bits); experiments build: stream / lean2 / split / dma"},
    {"ATLAS_AMD_LEG_STREAM_W",      C::dev,       "auto",   "workgroups per XCD of the persistent (stream) Legendre kernels instead of (workgroups per CU by the occupancy query) x 32, rounded down to a multiple of the column chunks"},
    {"ATLAS_AMD_LEG_STREAM_DYNAMIC", C::dev,      "1",      "0: static unit assignment (unit = workgroup + k W) of the persistent Legendre kernels instead of the work counter per XCD"},
    {"ATLAS_AMD_LEG_STREAM_SKEW",   C::dev,       "0",      "n: the second half of the persistent Legendre workgroups of an XCD starts n x 512 cycles late (phase probe)"},
    {"ATLAS_AMD_LEG_F32_TILES",     C::tuning,    "2 from T = 400 on, else 1", "latitude tiles of 64 per workgroup of the fp32 lean Legendre kernel: 2 = pairs (128 latitudes x 96 columns, 24 MFMAs per wavefront and stage), 1 = the 64-latitude form; same bits"},
    {"ATLAS_AMD_LEG_MIXED",         C::tuning,    "1",      "0: every column chunk of a Legendre call has the same width (round 5) instead of full 96-column chunks + one narrower launch for the remaining tiles where that is cheaper"},
    {"ATLAS_AMD_LEG_CFG",           C::tuning,    "auto",   "rtw,nrg: 16-column tiles per wavefront (1-3) and column groups per workgroup (1-2) of the Legendre tiling instead of the planner's choice"},
    {"ATLAS_AMD_FFT_GENERIC",       C::tuning,    "0",      "1: every row through the run-time shaped Fourier kernel (no compile-time shaped rows)"},
    {"ATLAS_AMD_FFT_STREAMS",       C::tuning,    "4 (1 for reduced grids with coarse classes of at most 3300 points per row)", "streams the row-length classes of the Fourier stage are dealt to (1-8)"},
    {"ATLAS_AMD_FFT_PREFETCH",      C::tuning,    "2,1",    "distance[,requests per line] of the L2 prefetch of a later job's modes; 0: off"},
    {"ATLAS_AMD_FFT_ROW_AFFINITY",  C::tuning,    "1",      "0: no row -> XCD affinity of the Fourier jobs"},
    {"ATLAS_AMD_FFT_FAST_M",        C::tuning,    "all",    "M,M,..: only these Bluestein lengths take the whole-row-in-one-function form (row_ct3)"},
    {"ATLAS_AMD_FFT_FINER_M",       C::tuning,    "1",      "0: without the extra Bluestein lengths 2304 / 3840 / 4608"},
    {"ATLAS_AMD_FFT_NT_DIV",        C::tuning,    "16",     "elements per worker that size a Fourier workgroup"},
    {"ATLAS_AMD_FFT_SMOOTH_DIRECT", C::tuning,    "0",      "1: the round-2 rule (direct run-time shaped transform for every {2,3,5}-smooth half length)"},
    {"ATLAS_AMD_FFT_COARSE",        C::tuning,    "auto",   "0 / 1: coarse row classes (the Bluestein lengths 256 / 512 / 1024 / 2048 for every row short enough, the first three in one launch) off / on; auto = on for reduced grids"},
    {"ATLAS_AMD_FFT_COARSE_FUSED",  C::tuning,    "1",      "0: one launch per coarse class instead of one for all three"},
    {"ATLAS_AMD_FFT_COARSE_MULTI",  C::tuning,    "1",      "0: one field per workgroup in the coarse classes instead of several fields of a short row per wavefront"},
    {"ATLAS_AMD_FFT_GROUP_LOG2",    C::tuning,    "3 (fp32: 4)", "log2 of the fields per job group of the record-less Fourier kernels (3 or 4)"},
    {"ATLAS_AMD_FFT_MIDROT",        C::tuning,    "0",      "1: rotated order of the second Bluestein round (measured flat)"},
    {"ATLAS_AMD_FFT_F32_PAIRS",     C::tuning,    "1",      "0: fp32 rows with one field per job instead of two fields in the halves of packed fp32 instructions"},
    {"ATLAS_AMD_PREPARE",           C::tuning,    "auto",   "rows / stream: form of the vor/div preparation kernel (same bits)"},
    {"ATLAS_AMD_GP_TO_FIELD",       C::tuning,    "auto",   "rows / tiles: form of the [field][point] -> [point][field] transposition in front of a halo exchange"},
    {"ATLAS_AMD_HOST_PIPELINE",     C::tuning,    "1",      "0: host-pointer calls as one upload, one transform, one download instead of the full-duplex field-chunk pipeline"},
    {"ATLAS_AMD_HOST_CHUNK",        C::tuning,    "16",     "fields per chunk of the host-pointer pipeline (multiple of 8)"},
    {"ATLAS_AMD_HOST_THREADS",      C::tuning,    "8",      "threads of the host copy teams of the host-pointer pipeline"},
    {"ATLAS_AMD_DIST_ROWBASE",      C::tuning,    "1",      "0: the distributed Fourier stage walks the piece tables (round 3) instead of one combined [row][source] table"},
    {"ATLAS_AMD_DIST_PACK_PAD",     C::tuning,    "0",      "1: packed records of the transposition padded to 16 columns (2 % faster reads at P = 2, 5 % more bytes on the wire)"},
    // ---- test hooks: used by the GPU test-suite on the product library
    {"ATLAS_AMD_DIST_POISON",       C::test_hook, "0",      "1: the buffers of the distributed transform are filled with NaN before every transform"},
    {"ATLAS_AMD_DIST_CHECK",        C::test_hook, "first use", "always: the ranks compare (field count, message limit) on every call (a blocking 3-int all-to-all), not only the first time a pair is used"},
    {"ATLAS_AMD_HOST_PIPELINE_FAIL_ALLOC", C::test_hook, "unset", "set: the staging buffers of the host-pointer pipeline fail to allocate (exercises the serial fallback)"},
    {"ATLAS_AMD_FFT_LDS_ELEMS",     C::test_hook, "10080",  "complex elements of a row transform beyond which a row is evaluated as a matrix product (dft_gemm.hip) instead of an FFT in LDS; read when the object is built: lets small grids exercise the path of O2560's longest rows"},
    // ---- development switches: compiled in only with -DATLAS_AMD_DEV_SWITCHES (make dev, make experiments)
    {"ATLAS_AMD_FFT_ABLATE",        C::dev,       "0",      "access ablations of the Fourier rows (results wrong; also needs -DAA_FFT_ABLATE)"},
    {"ATLAS_AMD_FFT_ONLY_M",        C::dev,       "0",      "M: launch only the Fourier class of this length"},
    {"ATLAS_AMD_FFT_ONLY_NATIVE",   C::dev,       "0",      "1: launch only the native mixed-radix rows"},
    {"ATLAS_AMD_FFT_DEBUG",         C::dev,       "unset",  "set: registers / LDS / workgroups per CU of every Fourier launch on stderr"},
    {"ATLAS_AMD_FFT_LDS_PAD",       C::dev,       "0",      "bytes of extra dynamic LDS per Fourier workgroup (occupancy probe)"},
    {"ATLAS_AMD_LEG_LDS_PAD",       C::dev,       "0",      "bytes of extra dynamic LDS per Legendre workgroup (occupancy probe)"},
    {"ATLAS_AMD_LEG_ABLATE",        C::dev,       "0",      "parts of the role-split / dma Legendre kernels left out (tools/experiments; results wrong)"},
    {"ATLAS_AMD_LEG_DEBUG",         C::dev,       "unset",  "set: launch geometry of the experimental Legendre kernels on stderr"},
    {"ATLAS_AMD_LEG_LAYOUT_PROBE",  C::dev,       "0",      "store side of intermediate layouts with several wavenumbers per line (results unusable; also needs -DAA_LEG_LAYOUT_PROBE)"},
    {"ATLAS_AMD_FFT_HYBRID",        C::dev,       "0",      "1: dense-stage hybrid rows (tools/experiments)"},
    {"ATLAS_AMD_FFT_HYB_MAXA",      C::dev,       "plan",   "largest dense radix of the hybrid rows"},
    {"ATLAS_AMD_FFT_HYB_NT",        C::dev,       "plan",   "threads of a hybrid-row workgroup"},
    {"ATLAS_AMD_FFT_NATIVE",        C::dev,       "0",      "1: native mixed-radix rows where a stage list exists (tools/experiments)"},
    {"ATLAS_AMD_FFT_NATIVE_FPJ",    C::dev,       "1",      "2: two fields per workgroup of the native rows"},
    {"ATLAS_AMD_FFT_HALFWIN",       C::dev,       "0",      "1: Bluestein rows with LDS as a half-row window (tools/experiments)"},
    {"ATLAS_AMD_FFT_SEQ",           C::dev,       "0",      "1: two jobs per workgroup in sequence (tools/experiments)"},
};
// clang-format on
std::atomic<int> g_ignore{0};

const EnvSwitch* find(const char* name) {
    for (const EnvSwitch& s : kSwitches) {
        if (std::strcmp(s.name, name) == 0) {
            return &s;
        }
    }
    return nullptr;
}
}  // namespace

const EnvSwitch* env_switches(int* count) {
    *count = (int)(sizeof(kSwitches) / sizeof(kSwitches[0]));
    return kSwitches;
}

const char* env_class_name(EnvClass c) {
    switch (c) {
        case C::tuning: return "tuning";
        case C::behaviour: return "behaviour";
        case C::test_hook: return "test hook";
        default: return "dev";
    }
}

bool env_dev_switches_compiled_in() {
#if defined(ATLAS_AMD_DEV_SWITCHES) || defined(ATLAS_AMD_EXPERIMENTS)
    return true;
#else
    return false;
#endif
}

void env_set_ignore(bool on) {
    g_ignore.store(on ? 1 : 0);
}

bool env_ignored() {
    if (g_ignore.load()) {
        return true;
    }
    const char* e = std::getenv("ATLAS_AMD_IGNORE_ENV");
    return e && *e && std::strcmp(e, "0") != 0;
}

const char* env_get(const char* name) {
    const EnvSwitch* s = find(name);
    if (!s) {
        std::fprintf(stderr, "[atlas_amd] internal error: environment switch %s is not in the table of csrc/env.cpp\n", name);
        return nullptr;
    }
    if (std::strcmp(name, "ATLAS_AMD_IGNORE_ENV") == 0) {
        return std::getenv(name);
    }
    if (env_ignored()) {
        return nullptr;
    }
    if (s->cls == C::dev && !env_dev_switches_compiled_in()) {
        if (std::getenv(name)) {   // say so once per name: a developer who set it should know why nothing changed
            static std::mutex mtx;
            static std::set<std::string> told;
            std::lock_guard<std::mutex> lk(mtx);
            if (told.insert(name).second) {
                std::fprintf(stderr, "[atlas_amd] %s is a development switch: ignored by this build (make -C atlas_amd/csrc dev | experiments)\n", name);
            }
        }
        return nullptr;
    }
    return std::getenv(name);
}

}  // namespace atlas_amd
