// microbench.hip -- calibration of the scattered-access rates that bound the Bloom passes on
// MI355X: random atomicOr / random 4-byte load / test-then-set vs footprint, LDS atomics, memset.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench && tools/microbench
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// mode 0: atomicOr no return; 1: load; 2: test-then-set; 3: atomicOr with return
template <int MODE, int PER>
__global__ void __launch_bounds__(256) k_rand(uint32_t *buf, uint64_t bits_mask, uint64_t salt, unsigned long long *sink)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint64_t a = mix((t * PER + i) * 0x9E3779B97F4A7C15ull + salt) & bits_mask;
        const uint32_t bit = 1u << (a & 31);
        if (MODE == 0) atomicOr(&buf[a >> 5], bit);
        else if (MODE == 1) acc += buf[a >> 5] & bit;
        else if (MODE == 2) { if (!(buf[a >> 5] & bit)) atomicOr(&buf[a >> 5], bit); }
        else acc += atomicOr(&buf[a >> 5], bit) & bit;
    }
    if (acc == 0x12345) atomicAdd(sink, 1ull);
}

template <int PER>
__global__ void __launch_bounds__(256) k_lds(uint32_t *out, uint64_t salt)
{
    __shared__ uint32_t s[32768];  // 128 KiB
    for (int i = threadIdx.x; i < 32768; i += 256) s[i] = 0;
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < PER; i++) {
        const uint64_t a = mix((t * PER + i) * 0x9E3779B97F4A7C15ull + salt) & ((1u << 20) - 1);
        atomicOr(&s[a >> 5], 1u << (a & 31));
    }
    __syncthreads();
    uint32_t x = 0;
    for (int i = threadIdx.x; i < 32768; i += 256) x ^= s[i];
    if (x == 0x12345) out[0] = x;
}

__global__ void k_stream_write(uint4 *p, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = make_uint4(i, 1, 2, 3);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d MHz, mem %.1f GB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.totalGlobalMem / 1e9);
    uint32_t *buf;
    const uint64_t maxbytes = 8ull << 30;
    CK(hipMalloc(&buf, maxbytes + 64));
    unsigned long long *sink;
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // memset
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0)); CK(hipMemsetAsync(buf, 0, maxbytes, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("memset 8 GiB: %.3f ms = %.0f GB/s\n", ms, maxbytes / ms / 1e6);
    }
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_stream_write, dim3(4096), dim3(256), 0, 0, (uint4 *)buf, maxbytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream write 8 GiB (uint4): %.3f ms = %.0f GB/s\n", ms, maxbytes / ms / 1e6);
    const int PER = 16;
    const uint64_t nthreads = 1ull << 26;  // x16 = 2^30 ops
    const double nops = (double)nthreads * PER;
    const char *names[4] = { "atomicOr(noret)", "load4B", "test-then-set", "atomicOr(ret)" };
    for (int logbytes : { 22, 25, 28, 30, 33 }) {
        const uint64_t bits_mask = (8ull << logbytes) - 1;
        for (int mode = 0; mode < 4; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0 || mode == 2 || mode == 3) CK(hipMemsetAsync(buf, 0, 1ull << logbytes, 0));
                CK(hipEventRecord(e0));
                dim3 g((unsigned)(nthreads / 256)), b(256);
                const uint64_t salt = rep * 7919 + mode;
                if (mode == 0) hipLaunchKernelGGL((k_rand<0, PER>), g, b, 0, 0, buf, bits_mask, salt, sink);
                if (mode == 1) hipLaunchKernelGGL((k_rand<1, PER>), g, b, 0, 0, buf, bits_mask, salt, sink);
                if (mode == 2) hipLaunchKernelGGL((k_rand<2, PER>), g, b, 0, 0, buf, bits_mask, salt, sink);
                if (mode == 3) hipLaunchKernelGGL((k_rand<3, PER>), g, b, 0, 0, buf, bits_mask, salt, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 1) printf("footprint 2^%d B  %-16s %8.3f ms  %7.2f Gops/s  (x64B = %.0f GB/s)\n", logbytes, names[mode], ms, nops / ms / 1e6, nops * 64 / ms / 1e6);
            }
        }
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_lds<256>), dim3(8192), dim3(256), 0, 0, buf, (uint64_t)rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("LDS atomicOr random (128 KiB slice/WG): %.3f ms %.2f Gops/s\n", ms, 8192.0 * 256 * 256 / ms / 1e6);
    }
    return 0;
}
