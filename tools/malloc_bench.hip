// malloc_bench.hip -- what a GiB of device memory costs to GET on this system, by the way it is asked for (the CLI
// holds 60 GB for a 78 MB input and CreateEnumerator at configs[4]'s shape spends 5.7 s in hipMalloc before its first kernel).
//   hipcc --offload-arch=gfx950 -O2 tools/malloc_bench.hip -o tools/malloc_bench && tools/malloc_bench [GiB ...]
// Per size: one hipMalloc; the same in 1 GiB pieces; hipMallocAsync from the default pool (first use, and again after a free: the pool
// keeps the memory); the virtual-memory API (hipMemAddressReserve + hipMemCreate / hipMemMap of 1 GiB handles + hipMemSetAccess).
// Every allocation is touched by a memset of its first and last MiB (a lazily mapped allocation would look free otherwise) and freed
// before the next way is timed.  Run twice in a row to see what a process pays that starts right after another one's exit.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("    %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); return -1.0; } } while (0)

static void touch(void *p, size_t bytes)
{
    (void)hipMemset(p, 0, 1 << 20);
    (void)hipMemset((char *)p + bytes - (1 << 20), 0, 1 << 20);
    (void)hipDeviceSynchronize();
}

static double t_malloc(size_t bytes)
{
    void *p = nullptr;
    const double t0 = now_ms();
    CK(hipMalloc(&p, bytes));
    touch(p, bytes);
    const double t1 = now_ms();
    (void)hipFree(p);
    return t1 - t0;
}

static double t_pieces(size_t bytes)
{
    std::vector<void *> ps;
    const double t0 = now_ms();
    for (size_t o = 0; o < bytes; o += (size_t)1 << 30) {
        void *p = nullptr;
        CK(hipMalloc(&p, (size_t)1 << 30));
        touch(p, (size_t)1 << 30);
        ps.push_back(p);
    }
    const double t1 = now_ms();
    for (void *p : ps) (void)hipFree(p);
    return t1 - t0;
}

static double t_async(size_t bytes, bool keep)
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipMemPool_t pool;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = keep ? ~0ull : 0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    void *p = nullptr;
    const double t0 = now_ms();
    CK(hipMallocAsync(&p, bytes, s));
    CK(hipStreamSynchronize(s));
    touch(p, bytes);
    const double t1 = now_ms();
    CK(hipFreeAsync(p, s));
    CK(hipStreamSynchronize(s));
    (void)hipStreamDestroy(s);
    return t1 - t0;
}

static double t_vmm(size_t bytes)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t piece = (size_t)1 << 30;
    void *va = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    const double t0 = now_ms();
    CK(hipMemAddressReserve(&va, bytes, gran, nullptr, 0));
    for (size_t o = 0; o < bytes; o += piece) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, piece, &prop, 0));
        CK(hipMemMap((char *)va + o, piece, 0, h, 0));
        hs.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    touch(va, bytes);
    const double t1 = now_ms();
    (void)hipMemUnmap(va, bytes);
    for (auto h : hs) (void)hipMemRelease(h);
    (void)hipMemAddressFree(va, bytes);
    std::printf("    (granularity %zu KiB)\n", gran >> 10);
    return t1 - t0;
}

// Back-to-back processes (a run started right after another one's exit waits seconds in its first large allocation):
//   malloc_bench dirty N          allocate N GiB, write all of it, exit (the driver then clears it in the background)
//   malloc_bench probe N [B]      right after that: [B GiB of ballast allocated first and never touched,] then N GiB allocated and touched
//                                 at both ends -- hipMalloc and the first touch timed apart
static int back_to_back(int argc, char **argv)
{
    const size_t n = (size_t)atoll(argv[2]) << 30;
    const double t0 = now_ms();
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    const double t1 = now_ms();
    if (argv[1][0] == 'd') {
        void *p = nullptr;
        if (hipMalloc(&p, n) != hipSuccess) { std::printf("dirty: hipMalloc failed\n"); return 1; }
        (void)hipMemset(p, 0x5A, n);
        (void)hipDeviceSynchronize();
        std::printf("dirty: %zu GiB written, exiting without freeing (init %.0f ms, alloc + fill %.0f ms)\n", n >> 30, t1 - t0, now_ms() - t1);
        return 0;
    }
    const size_t ballast = argc > 3 ? (size_t)atoll(argv[3]) << 30 : 0;
    const bool threaded = argc > 4;  // probe N B t: the ballast is allocated by a SECOND thread, this one waits 50 ms and goes on
    void *b = nullptr, *p = nullptr;
    double tb = 0;
    std::thread side;
    if (ballast && threaded) {
        side = std::thread([&]() { (void)hipSetDevice(0); const double a0 = now_ms(); if (hipMalloc(&b, ballast) != hipSuccess) std::printf("probe: ballast failed\n"); tb = now_ms() - a0; });
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    } else if (ballast) { const double a0 = now_ms(); if (hipMalloc(&b, ballast) != hipSuccess) std::printf("probe: ballast failed\n"); tb = now_ms() - a0; }
    const double a0 = now_ms();
    if (hipMalloc(&p, n) != hipSuccess) { std::printf("probe: hipMalloc failed\n"); return 1; }
    const double a1 = now_ms();
    touch(p, n);
    const double a2 = now_ms();
    (void)hipMemset(p, 0, n);
    (void)hipDeviceSynchronize();
    const double a3 = now_ms();
    if (side.joinable()) side.join();
    std::printf("probe: init %.0f ms; ballast %zu GiB%s hipMalloc %.1f ms; %zu GiB: hipMalloc %.1f ms, first touch (2 MiB) %.1f ms, memset of all of it %.1f ms\n", t1 - t0, ballast >> 30,
                threaded ? " (second thread)" : "", tb, n >> 30, a1 - a0, a2 - a1, a3 - a2);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 2 && (argv[1][0] == 'd' || argv[1][0] == 'p')) return back_to_back(argc, argv);
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back((size_t)atoll(argv[i]));
    if (sizes.empty()) sizes = {8, 32, 64, 128};
    const double t0 = now_ms();
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    std::printf("hipInit + context: %.1f ms; free %.1f of %.1f GiB\n", now_ms() - t0, fr / 1073741824.0, tot / 1073741824.0);
    for (size_t g : sizes) {
        const size_t bytes = g << 30;
        std::printf("%zu GiB\n", g);
        double t;
        t = t_malloc(bytes);        std::printf("  hipMalloc, one piece        %9.1f ms  %6.2f ms/GiB\n", t, t / g);
        t = t_malloc(bytes);        std::printf("  hipMalloc, again            %9.1f ms  %6.2f ms/GiB\n", t, t / g);
        t = t_pieces(bytes);        std::printf("  hipMalloc, 1 GiB pieces     %9.1f ms  %6.2f ms/GiB\n", t, t / g);
        t = t_async(bytes, true);   std::printf("  hipMallocAsync (pool keeps) %9.1f ms  %6.2f ms/GiB\n", t, t / g);
        t = t_async(bytes, false);  std::printf("  hipMallocAsync, again       %9.1f ms  %6.2f ms/GiB\n", t, t / g);
        t = t_vmm(bytes);           std::printf("  hipMemCreate + hipMemMap    %9.1f ms  %6.2f ms/GiB\n", t, t / g);
    }
    return 0;
}
