// Host-side copies of the host-pointer pipeline (csrc/trans.hip: invtrans_host_pipelined), compiled by g++ with OpenMP.
#pragma once
#include <cstddef>

namespace atlas_amd {
// threads of a copy team: ATLAS_AMD_HOST_THREADS, default 8 (two teams run at once: the gather and the drain; sweep of 8 / 12 / 16 /
// 24 threads x chunks of 16 / 24 / 32 / 40 fields in profiles/r05_bench_host.txt).  Bounded on purpose: on the GPU box (2 x 64 cores, 256 hardware
// threads) an OpenMP memcpy over all threads reaches 20 GB/s, over 16 - 32 threads 120 - 165 GB/s (profiles/r05_host_link_probe.txt)
int host_copy_threads();
void bounded_copy(void* dst, const void* src, size_t bytes);
// dst[r * n + j] = src[r * nf + f0 + j], r < nrows: the fields [f0, f0 + n) of every spectral coefficient (fields are the fastest index)
void gather_field_columns(double* dst, const double* src, size_t nrows, int nf, int f0, int n);
}  // namespace atlas_amd
