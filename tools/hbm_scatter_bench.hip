// hbm_scatter_bench.hip -- what the memory system does with the write pattern of the binning kernels: whole aligned
// granules of G bytes (what one bin flush stores) going to many independent output streams, nothing else in the
// kernel.  Patterns:
//   random   every granule goes to a pseudo-random granule-aligned address of the buffer
//   streams  65536 regions (256 workgroups x 256 bins, as in k_q_hash); each workgroup walks its 256 regions in a
//            pseudo-random order and appends one granule to the region it picked
//   linear   the same number of granules written back to back (the streaming ceiling of the same code)
// and the mirror-image read patterns (granule loads summed into a sink).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_scatter_bench.hip -o tools/hbm_scatter_bench && tools/hbm_scatter_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// PATTERN 0 random, 1 streams, 2 linear.  LPG lanes of 16 bytes make one granule.  READ: load instead of store.
template <int PATTERN, int LPG, bool READ>
__global__ void __launch_bounds__(1024) k_scatter(uint4 *buf, uint64_t n_granules_total, uint64_t granules_per_wg, uint64_t region_granules, uint32_t *sink)
{
    const uint32_t tid = threadIdx.x, l = tid % LPG, grp = tid / LPG;
    constexpr uint32_t GPW = 1024 / LPG;  // granules per workgroup step
    const uint64_t g0 = (uint64_t)blockIdx.x * granules_per_wg;
    uint32_t acc = 0;
    for (uint64_t i = grp; i < granules_per_wg; i += GPW) {
        uint64_t g;  // granule index in the buffer
        if (PATTERN == 0) {
            const uint64_t r = ((uint64_t)mix((uint32_t)(g0 + i)) << 32 | mix((uint32_t)(g0 + i) ^ 0x9E3779B9u));
            g = r % n_granules_total;
        } else if (PATTERN == 1) {
            // step s of this workgroup: GPW granules, each to a different pseudo-random bin; the position inside
            // the region advances with the step
            const uint64_t s = i / GPW;
            const uint32_t bin = (mix((uint32_t)s * 40503u + blockIdx.x) + grp * 37u) & 255u;
            const uint64_t pos = (s * GPW) / 256u;  // about this many granules are in each region by now
            g = ((uint64_t)blockIdx.x * 256u + bin) * region_granules + (pos + grp / 256u) % region_granules;
        } else {
            g = g0 + i;
        }
        uint4 *p = buf + g * LPG + l;
        if (READ) { const uint4 v = *p; acc += v.x ^ v.y ^ v.z ^ v.w; }
        else *p = make_uint4((uint32_t)i, tid, 0u, 0u);
    }
    if (READ && acc == 0x12345678u) sink[0] = acc;
}

template <int PATTERN, int LPG, bool READ>
int run(const char *name, uint4 *buf, uint64_t buf_bytes, uint64_t total_bytes, uint32_t *sink)
{
    const uint64_t G = (uint64_t)LPG * 16;
    const uint64_t n_total = buf_bytes / G;
    const uint32_t nwg = 256;
    const uint64_t per_wg = total_bytes / G / nwg;
    const uint64_t region_granules = n_total / (256ull * 256ull);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_scatter<PATTERN, LPG, READ>), dim3(nwg), dim3(1024), 0, 0, buf, n_total, per_wg, region_granules, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-8s %-7s granule %5llu B: %8.3f ms  %6.2f TB/s\n", READ ? "read" : "write", name, (unsigned long long)G, ms, (double)per_wg * nwg * G / ms / 1e9);
    return 0;
}

int main()
{
    const uint64_t buf_bytes = 16ull << 30, total = 15ull << 30;
    uint4 *buf; uint32_t *sink;
    CK(hipMalloc(&buf, buf_bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, buf_bytes));
#define ALL(P, NAME, RD) \
    run<P, 4, RD>(NAME, buf, buf_bytes, total, sink); run<P, 8, RD>(NAME, buf, buf_bytes, total, sink); run<P, 16, RD>(NAME, buf, buf_bytes, total, sink); \
    run<P, 32, RD>(NAME, buf, buf_bytes, total, sink); run<P, 64, RD>(NAME, buf, buf_bytes, total, sink);
    ALL(2, "linear", false)
    ALL(0, "random", false)
    ALL(1, "streams", false)
    ALL(2, "linear", true)
    ALL(0, "random", true)
    ALL(1, "streams", true)
    CK(hipFree(buf)); CK(hipFree(sink));
    return 0;
}
