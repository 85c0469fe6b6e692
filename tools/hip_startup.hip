// hip_startup.hip -- where the ~95 ms between main() and the first usable device context go (end to end).
// hipcc --offload-arch=gfx950 -O2 tools/hip_startup.hip -o tools/hip_startup ; tools/hip_startup [GiB of filter]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

__global__ void k_nop(unsigned *p) { if (p) p[0] = 1; }

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    double t = now_ms(), t0 = t;
    auto lap = [&](const char *what) { const double n = now_ms(); printf("%-44s %8.2f ms  (at %7.2f)\n", what, n - t, n - t0); t = n; };
    (void)hipInit(0); lap("hipInit");
    int n = 0; (void)hipGetDeviceCount(&n); lap("hipGetDeviceCount");
    (void)hipSetDevice(0); lap("hipSetDevice");
    (void)hipFree(nullptr); lap("hipFree(0) (primary context)");
    hipStream_t s; (void)hipStreamCreate(&s); lap("hipStreamCreate");
    hipEvent_t ev[32]; for (auto &e : ev) (void)hipEventCreate(&e); lap("32 x hipEventCreate");
    void *small = nullptr; (void)hipMalloc(&small, 4096); lap("hipMalloc 4 KiB");
    void *filt = nullptr; (void)hipMalloc(&filt, gib << 30); lap("hipMalloc filter");
    void *buf = nullptr; (void)hipMalloc(&buf, (size_t)30 << 30); lap("hipMalloc 30 GiB");
    void *pin = nullptr; (void)hipHostMalloc(&pin, 64 << 20, hipHostMallocDefault); lap("hipHostMalloc 64 MiB");
    (void)hipMemsetAsync(filt, 0, gib << 30, s); lap("hipMemsetAsync filter (submit)");
    (void)hipStreamSynchronize(s); lap("  ... sync");
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, (unsigned *)small); lap("first kernel launch (submit)");
    (void)hipStreamSynchronize(s); lap("  ... sync");
    (void)hipMemcpyAsync(small, pin, 4096, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); lap("first H2D copy 4 KiB");
    (void)hipMemcpyAsync(buf, pin, 64 << 20, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); lap("H2D 64 MiB pinned");
    (void)hipMemcpyAsync(pin, buf, 64 << 20, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); lap("D2H 64 MiB pinned");
    printf("total %.2f ms\n", now_ms() - t0);
    return 0;
}
