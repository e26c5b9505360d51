// Probe (GPU box): what the host side of the host-pointer entry points can sustain -- host memcpy rates (one thread / OpenMP),
// the strided field gather, PCIe rates from pinned and from pageable memory, one direction and both at once.
//   hipcc -O2 -fopenmp --offload-arch=gfx950 tools/probe/probe_host_link.hip -o tools/probe/probe_host_link_gfx950
#include <hip/hip_runtime.h>
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s\n", hipGetErrorString(e), #x); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pcopy(void* d, const void* s, size_t n) {
    const size_t blk = size_t(4) << 20;
    const long long nb = (long long)((n + blk - 1) / blk);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) std::memcpy((char*)d + b * blk, (const char*)s + b * blk, std::min<size_t>(blk, n - (size_t)b * blk));
}
int main() {
    const size_t N = size_t(2) << 30;   // bytes per buffer
    std::printf("omp_get_max_threads %d, hardware_concurrency %u\n", omp_get_max_threads(), std::thread::hardware_concurrency());
    char *pg1 = (char*)std::malloc(N), *pg2 = (char*)std::malloc(N), *pin1, *pin2, *d1, *d2;
    std::memset(pg1, 1, N);
    std::memset(pg2, 2, N);
    CK(hipHostMalloc((void**)&pin1, N, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&pin2, N, hipHostMallocDefault));
    std::memset(pin1, 3, N);
    std::memset(pin2, 4, N);
    CK(hipMalloc((void**)&d1, N));
    CK(hipMalloc((void**)&d2, N));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto rate = [&](const char* what, auto&& fn, double bytes) {
        fn();
        double best = 1e9;
        for (int r = 0; r < 3; ++r) {
            const double t0 = now();
            fn();
            best = std::min(best, now() - t0);
        }
        std::printf("%-58s %7.1f GB/s (%.1f ms)\n", what, bytes / best / 1e9, best * 1e3);
    };
    rate("memcpy pageable -> pinned, one thread", [&] { std::memcpy(pin1, pg1, N); }, (double)N);
    rate("memcpy pageable -> pinned, OpenMP", [&] { pcopy(pin1, pg1, N); }, (double)N);
    rate("memcpy pinned -> pageable, OpenMP", [&] { pcopy(pg2, pin2, N); }, (double)N);
    rate("memcpy pageable -> pageable, OpenMP", [&] { pcopy(pg2, pg1, N); }, (double)N);
    for (int nt : {2, 4, 8, 16, 32, 64}) {
        omp_set_num_threads(nt);
        char what[96];
        std::snprintf(what, sizeof what, "memcpy pinned -> pageable, %d threads", nt);
        rate(what, [&] { pcopy(pg2, pin2, N); }, (double)N);
        std::snprintf(what, sizeof what, "D2H pinned + drain of the other pinned buffer, %d threads", nt);
        rate(what, [&] {
            CK(hipMemcpyAsync(pin2, d2, N, hipMemcpyDeviceToHost, s2));
            pcopy(pg2, pin1, N);
            CK(hipStreamSynchronize(s2));
        }, (double)N);
    }
    omp_set_num_threads(omp_get_num_procs());
    {   // field gather: rows of 137 doubles, 24 of them taken
        const int nf = 137, n = 24;
        const size_t rows = N / (nf * 8);
        rate("gather 24 of 137 doubles per row, pageable -> pinned, OpenMP", [&] {
#pragma omp parallel for schedule(static)
            for (long long r = 0; r < (long long)rows; ++r) std::memcpy(pin1 + r * n * 8, pg1 + r * nf * 8, n * 8);
        }, (double)rows * n * 8);
    }
    rate("H2D pinned", [&] { CK(hipMemcpyAsync(d1, pin1, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); }, (double)N);
    rate("D2H pinned", [&] { CK(hipMemcpyAsync(pin2, d2, N, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); }, (double)N);
    rate("H2D + D2H pinned at once (sum of both)", [&] {
        CK(hipMemcpyAsync(d1, pin1, N, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(pin2, d2, N, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
    }, 2.0 * N);
    rate("H2D pageable", [&] { CK(hipMemcpy(d1, pg1, N, hipMemcpyHostToDevice)); }, (double)N);
    rate("D2H pageable", [&] { CK(hipMemcpy(pg2, d2, N, hipMemcpyDeviceToHost)); }, (double)N);
    rate("H2D + D2H pageable from two host threads (sum of both)", [&] {
        std::thread a([&] { CK(hipMemcpy(d1, pg1, N, hipMemcpyHostToDevice)); });
        std::thread b([&] { CK(hipMemcpy(pg2, d2, N, hipMemcpyDeviceToHost)); });
        a.join();
        b.join();
    }, 2.0 * N);
    rate("D2H pinned + OpenMP drain of the other pinned buffer at once", [&] {
        CK(hipMemcpyAsync(pin2, d2, N, hipMemcpyDeviceToHost, s2));
        pcopy(pg2, pin1, N);
        CK(hipStreamSynchronize(s2));
    }, (double)N);
    return 0;
}
