// lds_bench.hip -- LDS operation rates that bound the write-combining bins (tpc_bins.h) on MI355X:
// returning / non-returning LDS atomics over a few hundred counters, random ds_or over a slice,
// random ring stores, and the whole push pattern (claim + head read + ring store).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_bench.hip -o tools/lds_bench && tools/lds_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0 atomicAdd returning over NB counters      1 atomicAdd no return over NB counters
//      2 atomicOr no return over 32768 words       3 store b32 random over 32768 words
//      4 store b64 random over 16384               5 load b32 random over 32768 words
//      6 push pattern u32 entries: claim + head read + ring store   7 push pattern u64 entries
//      8 atomicAdd returning on u64 counters (claim packs)          9 load b64 random (table lookups)
template <int MODE, int THREADS, int UNROLL>
__global__ void __launch_bounds__(THREADS) k_lds(uint32_t *out, int iters, int NB, uint32_t salt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s[];  // 128 KiB data + counters
    uint32_t *cnt = s + 32768;
    uint32_t *head = cnt + 1024;
    for (int i = threadIdx.x; i < 32768 + 2048; i += THREADS) s[i] = 0;
    __syncthreads();
    uint32_t rng = (blockIdx.x * THREADS + threadIdx.x) * 2654435761u + salt;
    uint32_t acc = 0;
    const uint32_t nbm = (uint32_t)NB - 1u;
    const int logcap = MODE == 7 ? 14 - (31 - __builtin_clz(NB)) : 15 - (31 - __builtin_clz(NB));
    for (int it = 0; it < iters; it++) {
        uint32_t r[UNROLL], v[UNROLL], h[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) r[u] = lcg(rng);
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) v[u] = atomicAdd(&cnt[r[u] & nbm], 1u);
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += v[u];
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) __hip_atomic_fetch_add(&cnt[r[u] & nbm], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) __hip_atomic_fetch_or(&s[(r[u] >> 5) & 32767u], 1u << (r[u] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) s[r[u] & 32767u] = r[u];
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) reinterpret_cast<uint64_t *>(s)[r[u] & 16383u] = ((uint64_t)r[u] << 32) | it;
        } else if (MODE == 5) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) v[u] = s[r[u] & 32767u];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += v[u];
        } else if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) { v[u] = atomicAdd(&cnt[r[u] & nbm], 1u); h[u] = head[r[u] & nbm]; }
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (v[u] - h[u] < 0x40000000u) s[((r[u] & nbm) << logcap) + (v[u] & ((1u << logcap) - 1u))] = r[u];
        } else if (MODE == 7) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) { v[u] = atomicAdd(&cnt[r[u] & nbm], 1u); h[u] = head[r[u] & nbm]; }
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (v[u] - h[u] < 0x40000000u) reinterpret_cast<uint64_t *>(s)[((r[u] & nbm) << logcap) + (v[u] & ((1u << logcap) - 1u))] = ((uint64_t)r[u] << 32) | v[u];
        } else if (MODE == 8) {
            unsigned long long *c64 = reinterpret_cast<unsigned long long *>(cnt);
            unsigned long long w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) w[u] = atomicAdd(&c64[r[u] & nbm & 511u], 1ull);
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += (uint32_t)w[u];
        } else if (MODE == 9) {
            uint64_t w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) w[u] = reinterpret_cast<uint64_t *>(s)[r[u] & 63u];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += (uint32_t)w[u];
        }
    }
    __syncthreads();
    uint32_t x = acc;
    for (int i = threadIdx.x; i < 32768 + 2048; i += THREADS) x ^= s[i];
    if (x == 0x12345) out[0] = x;
}

template <int MODE, int THREADS, int UNROLL>
int run(const char *name, uint32_t *out, int NB)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (32768 + 2048) * 4;
    CK(hipFuncSetAttribute((const void *)k_lds<MODE, THREADS, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int iters = 2048 / UNROLL, blocks = 1024;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_lds<MODE, THREADS, UNROLL>), dim3(blocks), dim3(THREADS), lds, 0, out, iters, NB, (uint32_t)rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double ops = (double)blocks * THREADS * iters * UNROLL;
    printf("%-44s threads %4d unroll %2d NB %4d: %8.3f ms %9.1f Gops/s  (%.2f lanes/clk/CU)\n", name, THREADS, UNROLL, NB, ms, ops / ms / 1e6,
           ops / ms / 1e6 / 256 / 2.4);
    return 0;
}

int main()
{
    uint32_t *out;
    CK(hipMalloc(&out, 64));
    for (int NB : {256, 512}) {
        run<0, 1024, 8>("atomicAdd returning, NB counters", out, NB);
        run<0, 1024, 2>("atomicAdd returning, NB counters", out, NB);
        run<0, 512, 8>("atomicAdd returning, NB counters", out, NB);
        run<0, 256, 8>("atomicAdd returning, NB counters", out, NB);
        run<1, 1024, 8>("atomicAdd no return, NB counters", out, NB);
        run<8, 1024, 8>("atomicAdd u64 returning, NB counters", out, NB);
        run<6, 1024, 8>("push u32: claim + head read + ring store", out, NB);
        run<6, 1024, 5>("push u32: claim + head read + ring store", out, NB);
        run<7, 1024, 8>("push u64: claim + head read + ring store", out, NB);
        run<7, 1024, 4>("push u64: claim + head read + ring store", out, NB);
    }
    run<2, 1024, 8>("atomicOr no return, 32768 words", out, 256);
    run<2, 256, 8>("atomicOr no return, 32768 words", out, 256);
    run<2, 256, 1>("atomicOr no return, 32768 words", out, 256);
    run<3, 1024, 8>("store b32 random, 32768 words", out, 256);
    run<4, 1024, 8>("store b64 random, 16384 dwords", out, 256);
    run<5, 1024, 8>("load b32 random, 32768 words", out, 256);
    run<9, 1024, 8>("load b64 random, 64 entries (tables)", out, 256);
    return 0;
}
