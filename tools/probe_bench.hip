// probe_bench.hip -- random 4-byte Bloom probes against large filters: uniformly random over the whole filter (what k_q_verify
// does) vs random inside a window that moves with the block index (what probes sorted into coarse address buckets would do).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_bench.hip -o tools/probe_bench && tools/probe_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// window_words = footprint: uniform.  Otherwise block b probes window (b * n_windows / gridDim.x).
template <int PER>
__global__ void __launch_bounds__(256) k_probe(const uint32_t *buf, uint64_t words, uint64_t window_words, uint64_t salt, unsigned long long *sink)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_windows = words / window_words;
    const uint64_t base = (uint64_t)blockIdx.x * n_windows / gridDim.x * window_words;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint64_t a = base + mix((t * PER + i) * 0x9E3779B97F4A7C15ull + salt) % window_words;
        acc += buf[a];
    }
    if (acc == 0x12345) atomicAdd(sink, 1ull);
}

int main()
{
    unsigned long long *sink;
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int loggib : { 3, 5, 7 }) {
        const uint64_t bytes = 1ull << (30 + loggib);
        uint32_t *buf = nullptr;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("no %d GiB\n", 1 << loggib); (void)hipGetLastError(); continue; }
        CK(hipMemset(buf, 0, bytes));
        const int PER = 8;
        const uint64_t nthreads = 1ull << 26;
        for (unsigned long long wbytes : { (unsigned long long)bytes, 4ull << 30, 1ull << 30, 256ull << 20, 64ull << 20, 16ull << 20 }) {
            if (wbytes > bytes) continue;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_probe<PER>), dim3((unsigned)(nthreads / 256)), dim3(256), 0, 0, buf, bytes / 4, wbytes / 4, (uint64_t)rep, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("filter %4d GiB, probes random within a %6.0f MiB window: %8.2f ms  %6.1f G probes/s\n", 1 << loggib, wbytes / 1048576.0, ms, nthreads * PER / ms / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
