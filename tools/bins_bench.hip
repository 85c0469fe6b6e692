// bins_bench.hip -- the LDS write-combining bins on their own: synthetic entries (random bins), no hashing.
// Compares tpc_bins.h:Bins (barrier flush per round) with tpc_rbins.h:RBins (barrier-free rings) at the
// entry counts of the M2 workload and verifies that every entry reaches its region exactly once.
//   hipcc --offload-arch=gfx950 -O3 -Itwopaco_amd/csrc -Iinclude tools/bins_bench.hip -o tools/bins_bench
#include "tpc_rbins.h"
#include "tpc_bins3.h"
#include "tpc_binsp.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

template <class T> __device__ __forceinline__ T make_val(uint32_t r, uint32_t id);
template <> __device__ __forceinline__ uint32_t make_val<uint32_t>(uint32_t r, uint32_t) { return r & 0x0FFFFFFFu; }
template <> __device__ __forceinline__ uint64_t make_val<uint64_t>(uint32_t r, uint32_t id) { return ((uint64_t)id << 31) | (r & 0x0FFFFFFFu); }

// V2 = false: Bins with a flush every `ppr` steps; true: RBins.  N entries per thread per step.
template <class T, int N, bool V2, int THREADS>
__global__ void __launch_bounds__(THREADS) k_bins(int LOG_NB, int steps, int ppr, int filler, T *buf, uint32_t *cnt, uint64_t cap,
                                                   unsigned long long *sums)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    const uint32_t wg = blockIdx.x;
    auto reg = [buf, cap, wg, NB](uint32_t b) { return PtRegion<T>{buf + ((uint64_t)wg * NB + b) * cap, cap}; };
    unsigned long long *lostc = sums + 2;
    auto lost = [lostc](uint32_t, T) { atomicAdd(lostc, 1ull); };
    uint32_t rng = (blockIdx.x * THREADS + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long sum = 0;
    uint32_t acc = 0;
    if constexpr (V2) {
        RBins<T, THREADS> bins;
        bins.carve(smem, LOG_NB);
        bins.init();
        __syncthreads();
        for (int s = 0; s < steps; s++) {
            uint32_t b[N];
            T val[N];
            bool ok[N];
#pragma unroll
            for (int i = 0; i < N; i++) {
                uint32_t r = lcg(rng);
                for (int f = 0; f < (filler >= 1000 ? 0 : filler); f++) r = r * 1664525u + (r >> 13);  // stands in for the hashing
                b[i] = (r >> 8) & (uint32_t)(NB - 1);
                val[i] = make_val<T>(r, (uint32_t)s);
                ok[i] = true;
                sum += (unsigned long long)val[i] ^ ((unsigned long long)b[i] << 40);
            }
            bins.template push_batch<N>(b, val, ok, reg, lost);
        }
        bins.flush(true, reg, lost);
        bins.store_counts(cnt + (uint64_t)wg * NB, reg, [](uint32_t b) { return b; });
    } else {
        Bins<T, THREADS> bins;
        bins.carve(smem, LOG_NB);
        bins.init();
        __syncthreads();
        if (filler < 0) {  // stagger the workgroups: phase * D
            const unsigned long long t0 = wall_clock64(), d = (unsigned long long)(blockIdx.x % 4) * (unsigned long long)(-filler);
            while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
            filler = 0;
        }
        for (int s0 = 0; s0 < steps; s0 += ppr) {
            for (int s = s0; s < min(steps, s0 + ppr); s++) {
                uint32_t b[N];
                T val[N];
                bool ok[N];
#pragma unroll
                for (int i = 0; i < N; i++) {
                    uint32_t r = lcg(rng);
                    for (int f = 0; f < (filler >= 1000 ? 0 : filler); f++) r = r * 1664525u + (r >> 13);
                    b[i] = (r >> 8) & (uint32_t)(NB - 1);
                    if (filler >= 1000 && ((s >> 4) & 1)) b[i] = (uint32_t)((i * 5 + (s >> 5)) % 3) & (uint32_t)(NB - 1);
                    val[i] = make_val<T>(r, (uint32_t)s);
                    ok[i] = true;
                    sum += (unsigned long long)val[i] ^ ((unsigned long long)b[i] << 40);
                }
                bins.template push_batch<N>(b, val, ok, lost);
            }
            bins.flush(false, reg, lost);
        }
        bins.flush(true, reg, lost);
        bins.store_counts(cnt + (uint64_t)wg * NB, reg, [](uint32_t b) { return b; });
        bins.dump(sums + 8);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[0], sum);
    if (acc == 0x12345) sums[3] = 1;
}

template <class T> __global__ void k_check(const T *buf, const uint32_t *cnt, uint64_t cap, int NB, unsigned long long *sums);
// Bins3 (tpc_bins3.h): every wave flushes the ring groups it owns
template <class T, int N, int THREADS, int DEBUG>
__global__ void __launch_bounds__(THREADS) k_bins3(int LOG_NB, int steps, int ppr, int filler, T *buf, uint32_t *cnt, uint64_t cap, unsigned long long *sums)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    const uint32_t wg = blockIdx.x;
    unsigned lost_n = 0;
    auto lost = [&lost_n](uint32_t, T) { lost_n++; };
    uint32_t rng = (blockIdx.x * THREADS + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long sum = 0;
    Bins3<T, THREADS, DEBUG> bins;
    bins.carve(smem, LOG_NB);
    bins.init(buf, [=](uint32_t b) { return make_uint2((uint32_t)((((uint64_t)wg * NB + b) * cap) / Bins3<T, THREADS>::GROUP), (uint32_t)cap); });
    __syncthreads();
    for (int s0 = 0; s0 < steps; s0 += ppr) {
        for (int s = s0; s < min(steps, s0 + ppr); s++) {
            uint32_t b[N];
            T val[N];
            bool ok[N];
#pragma unroll
            for (int i = 0; i < N; i++) {
                uint32_t r = lcg(rng);
                for (int f = 0; f < (filler >= 1000 ? 0 : filler); f++) r = r * 1664525u + (r >> 13);
                b[i] = ((r ^ (r >> 15)) * 0x2c1b3c6du >> 12) & (uint32_t)(NB - 1);
                if (filler >= 1000 && ((s >> 4) & 1)) b[i] = (uint32_t)((i * 5 + (s >> 5)) % 3) & (uint32_t)(NB - 1);  // poly-A-like: every lane hits the same few bins for 16 steps
                val[i] = make_val<T>(r, (uint32_t)s);
                ok[i] = true;
                sum += (unsigned long long)val[i] ^ ((unsigned long long)b[i] << 40);
            }
            bins.template push_batch<N>(b, val, ok, lost);
        }
        bins.template flush<false>(lost);
    }
    bins.template flush<true>(lost);
    bins.store_counts(cnt + (uint64_t)wg * NB, [](uint32_t b) { return b; });
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[0], sum);
    if (lost_n) atomicAdd(&sums[2], (unsigned long long)lost_n);
}


// two workgroups per CU: 512 threads, 64 KiB of rings, 64-byte flush granules
template <class T, int N>
__global__ void __launch_bounds__(512) k_bins_half(int LOG_NB, int steps, T *buf, uint32_t *cnt, uint64_t cap, unsigned long long *sums)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    const uint32_t wg = blockIdx.x;
    auto reg = [buf, cap, wg, NB](uint32_t b) { return PtRegion<T>{buf + ((uint64_t)wg * NB + b) * cap, cap}; };
    unsigned lost_n = 0;
    auto lost = [&lost_n](uint32_t, T) { lost_n++; };
    uint32_t rng = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long sum = 0;
    Bins<T, 512, 65536, 64> bins;
    bins.carve(smem, LOG_NB);
    bins.init();
    __syncthreads();
    for (int s = 0; s < steps; s++) {
        uint32_t b[N];
        T val[N];
        bool ok[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t r = lcg(rng);
            b[i] = (r >> 8) & (uint32_t)(NB - 1);
            val[i] = make_val<T>(r, (uint32_t)s);
            ok[i] = true;
            sum += (unsigned long long)val[i] ^ ((unsigned long long)b[i] << 40);
        }
        bins.template push_batch<N>(b, val, ok, lost);
        bins.flush(false, reg, lost);
    }
    bins.flush(true, reg, lost);
    bins.store_counts(cnt + (uint64_t)wg * NB, reg, [](uint32_t b) { return b; });
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[0], sum);
    if (lost_n) atomicAdd(&sums[2], (unsigned long long)lost_n);
}

template <class T, int N>
int run_half(const char *name, int log_nb, int steps)
{
    const int nwg = 512, NB = 1 << log_nb;
    const uint64_t per_bin = (uint64_t)steps * 512 * N / NB;
    const uint64_t cap = ((uint64_t)(per_bin * 1.2) + 256 + 31) & ~31ull;
    T *buf; uint32_t *cnt; unsigned long long *sums;
    CK(hipMalloc(&buf, (size_t)nwg * NB * cap * sizeof(T)));
    CK(hipMalloc(&cnt, (size_t)nwg * NB * 4));
    CK(hipMalloc(&sums, 256));
    const size_t lds = Bins<T, 512, 65536, 64>::lds_bytes(log_nb);
    CK(hipFuncSetAttribute((const void *)k_bins_half<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(sums, 0, 256));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_bins_half<T, N>), dim3(nwg), dim3(512), lds, 0, log_nb, steps, buf, cnt, cap, sums);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    hipLaunchKernelGGL((k_check<T>), dim3(nwg * NB), dim3(256), 0, 0, buf, cnt, cap, NB, sums);
    unsigned long long h[32];
    CK(hipMemcpy(h, sums, 256, hipMemcpyDeviceToHost));
    const double n = (double)nwg * 512 * steps * N;
    printf("%-28s NB %3d N %d (2 WG/CU, lds %zu): %8.3f ms %8.1f G entries/s   (entries %.0f, found %llu, lost %llu) %s\n", name, NB, N, lds, ms, n / ms / 1e6, n, h[4], h[2],
           h[4] + h[2] == (unsigned long long)n ? "count OK" : "COUNT MISMATCH");
    CK(hipFree(buf)); CK(hipFree(cnt)); CK(hipFree(sums));
    return 0;
}

template <class T>
__global__ void k_check(const T *buf, const uint32_t *cnt, uint64_t cap, int NB, unsigned long long *sums)
{   // one workgroup per region
    const uint64_t r = blockIdx.x;
    const uint32_t b = (uint32_t)(r % NB);
    const uint32_t n = cnt[r];
    unsigned long long sum = 0, c = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const T v = buf[r * cap + i];
        if (v != (T)~(T)0) { sum += (unsigned long long)v ^ ((unsigned long long)b << 40); c++; }
    }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); c += __shfl_down(c, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[1], sum); atomicAdd(&sums[4], c); }
}

template <class T, int N, bool V2>
int run(const char *name, int log_nb, int steps, int ppr, int filler)
{
    constexpr int THREADS = 1024;
    const int nwg = 256, NB = 1 << log_nb;
    const uint64_t per_bin = (uint64_t)steps * THREADS * N / NB;
    const uint64_t cap = ((uint64_t)(per_bin * 1.2) + 256 + 31) & ~31ull;
    T *buf;
    uint32_t *cnt;
    unsigned long long *sums;
    CK(hipMalloc(&buf, (size_t)nwg * NB * cap * sizeof(T)));
    CK(hipMalloc(&cnt, (size_t)nwg * NB * 4));
    CK(hipMalloc(&sums, 256));
    const size_t lds = V2 ? RBins<T, THREADS>::lds_bytes(log_nb) : Bins<T, THREADS>::lds_bytes(log_nb);
    CK(hipFuncSetAttribute((const void *)k_bins<T, N, V2, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(sums, 0, 256));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_bins<T, N, V2, THREADS>), dim3(nwg), dim3(THREADS), lds, 0, log_nb, steps, ppr, filler, buf, cnt, cap, sums);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    hipLaunchKernelGGL((k_check<T>), dim3(nwg * NB), dim3(256), 0, 0, buf, cnt, cap, NB, sums);
    unsigned long long h[32];
    CK(hipMemcpy(h, sums, 256, hipMemcpyDeviceToHost));
#ifdef TPC_PROFILE_PHASES
    if (!V2) printf("    phases (wall_clock ticks summed over WGs): push %llu  bookkeeping %llu  copy %llu  rounds %llu\n", h[8], h[9], h[10], h[12]);
#endif
    const double n = (double)nwg * THREADS * steps * N;
    printf("%-28s NB %3d N %d filler %2d: %8.3f ms %8.1f G entries/s %6.2f TB/s written   %s (entries %.0f, found %llu, lost %llu)\n", name, NB, N, filler, ms,
           n / ms / 1e6, n * sizeof(T) / ms / 1e9, (h[0] == h[1] && h[4] + h[2] == (unsigned long long)n && h[2] == 0) ? "OK" : "MISMATCH", n, h[4], h[2]);
    CK(hipFree(buf)); CK(hipFree(cnt)); CK(hipFree(sums));
    return 0;
}

template <class T, int N, int DEBUG = 0>
int run3(const char *name, int log_nb, int steps, int ppr, int filler)
{
    constexpr int THREADS = 1024;
    const int nwg = 256, NB = 1 << log_nb;
    const uint64_t per_bin = (uint64_t)steps * THREADS * N / NB;
    const uint64_t cap = ((uint64_t)(per_bin * 1.2) + 256 + 31) & ~31ull;
    T *buf; uint32_t *cnt; unsigned long long *sums;
    CK(hipMalloc(&buf, (size_t)nwg * NB * cap * sizeof(T)));
    CK(hipMalloc(&cnt, (size_t)nwg * NB * 4));
    CK(hipMalloc(&sums, 256));
    const size_t lds = Bins3<T, THREADS>::lds_bytes(log_nb);
    CK(hipFuncSetAttribute((const void *)k_bins3<T, N, THREADS, DEBUG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(sums, 0, 256));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_bins3<T, N, THREADS, DEBUG>), dim3(nwg), dim3(THREADS), lds, 0, log_nb, steps, ppr, filler, buf, cnt, cap, sums);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    hipLaunchKernelGGL((k_check<T>), dim3(nwg * NB), dim3(256), 0, 0, buf, cnt, cap, NB, sums);
    unsigned long long h[32];
    CK(hipMemcpy(h, sums, 256, hipMemcpyDeviceToHost));
    const double n = (double)nwg * THREADS * steps * N;
    printf("%-28s NB %3d N %d ppr %d filler %2d: %8.3f ms %8.1f G entries/s %6.2f TB/s written   %s (entries %.0f, found %llu, lost %llu)\n", name, NB, N, ppr, filler, ms,
           n / ms / 1e6, n * sizeof(T) / ms / 1e9, (h[0] == h[1] + 0 && h[4] + h[2] == (unsigned long long)n) ? (h[2] ? "count OK (some lost)" : "OK") : "MISMATCH", n, h[4], h[2]);
    CK(hipFree(buf)); CK(hipFree(cnt)); CK(hipFree(sums));
    return 0;
}

// BinsP (tpc_binsp.h): planar lines of 21 x 48-bit / 42 x 24-bit entries
template <class F> __device__ __forceinline__ typename F::T make_pval(uint32_t r, uint32_t id);
template <> __device__ __forceinline__ uint64_t make_pval<PFmt6>(uint32_t r, uint32_t id) { return ((uint64_t)(id & 0xFFFFFu) << 28) | (r & 0x0FFFFFFFu); }
template <> __device__ __forceinline__ uint32_t make_pval<PFmt3>(uint32_t r, uint32_t) { return r & 0xFFFFFFu; }

template <class F, int N, int THREADS>
__global__ void __launch_bounds__(THREADS) k_binsp(int LOG_NB, int steps, int ppr, int filler, unsigned char *buf, uint32_t *cnt, uint64_t cap_lines, unsigned long long *sums)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using T = typename F::T;
    const int NB = 1 << LOG_NB;
    const uint32_t wg = blockIdx.x;
    unsigned lost_n = 0;
    auto lost = [&lost_n](uint32_t, T) { lost_n++; };
    uint32_t rng = (blockIdx.x * THREADS + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long sum = 0;
    BinsP<F, THREADS> bins;
    bins.carve(smem, LOG_NB);
    bins.init(buf, [=](uint32_t b) { return make_uint2((uint32_t)(((uint64_t)wg * NB + b) * cap_lines), (uint32_t)cap_lines); });
    __syncthreads();
    for (int s0 = 0; s0 < steps; s0 += ppr) {
        for (int s = s0; s < min(steps, s0 + ppr); s++) {
            uint32_t b[N];
            T val[N];
            bool ok[N];
#pragma unroll
            for (int i = 0; i < N; i++) {
                uint32_t r = lcg(rng);
                for (int f = 0; f < (filler >= 1000 ? 0 : filler); f++) r = r * 1664525u + (r >> 13);
                b[i] = ((r ^ (r >> 15)) * 0x2c1b3c6du >> 12) & (uint32_t)(NB - 1);
                if (filler >= 1000 && ((s >> 4) & 1)) b[i] = (uint32_t)((i * 5 + (s >> 5)) % 3) & (uint32_t)(NB - 1);
                val[i] = make_pval<F>(r, (uint32_t)s);
                ok[i] = true;
                sum += (unsigned long long)val[i] ^ ((unsigned long long)b[i] << 50);
            }
            bins.template push_batch<N>(b, val, ok, lost);
        }
        bins.template flush<false>(lost);
    }
    bins.template flush<true>(lost);
    bins.store_counts(cnt + (uint64_t)wg * NB, [](uint32_t b) { return b; });
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[0], sum);
    if (lost_n) atomicAdd(&sums[2], (unsigned long long)lost_n);
}

template <class F>
__global__ void k_check_p(const unsigned char *buf, const uint32_t *cnt, uint64_t cap_lines, int NB, unsigned long long *sums)
{   // one workgroup per region
    const uint64_t r = blockIdx.x;
    const uint32_t b = (uint32_t)(r % NB);
    const uint32_t n = cnt[r];
    unsigned long long sum = 0, c = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t line = i / F::GROUP, e = i % F::GROUP;
        const typename F::T v = F::load(buf + (r * cap_lines + line) * 128, e);
        sum += (unsigned long long)v ^ ((unsigned long long)b << 50); c++;
    }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); c += __shfl_down(c, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[1], sum); atomicAdd(&sums[4], c); }
}

template <class F, int N>
int runp(const char *name, int log_nb, int steps, int ppr, int filler)
{
    constexpr int THREADS = 1024;
    const int nwg = 256, NB = 1 << log_nb;
    const uint64_t per_bin = (uint64_t)steps * THREADS * N / NB;
    const uint64_t cap_lines = ((uint64_t)(per_bin * 1.2) + 256) / F::GROUP + 1;
    unsigned char *buf; uint32_t *cnt; unsigned long long *sums;
    CK(hipMalloc(&buf, (size_t)nwg * NB * cap_lines * 128));
    CK(hipMalloc(&cnt, (size_t)nwg * NB * 4));
    CK(hipMalloc(&sums, 256));
    const size_t lds = BinsP<F, THREADS>::lds_bytes(log_nb);
    CK(hipFuncSetAttribute((const void *)k_binsp<F, N, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(sums, 0, 256));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_binsp<F, N, THREADS>), dim3(nwg), dim3(THREADS), lds, 0, log_nb, steps, ppr, filler, buf, cnt, cap_lines, sums);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    hipLaunchKernelGGL((k_check_p<F>), dim3(nwg * NB), dim3(256), 0, 0, buf, cnt, cap_lines, NB, sums);
    unsigned long long h[32];
    CK(hipMemcpy(h, sums, 256, hipMemcpyDeviceToHost));
    const double n = (double)nwg * THREADS * steps * N;
    printf("%-28s NB %3d N %d ppr %d filler %2d: %8.3f ms %8.1f G entries/s %6.2f TB/s written   %s (entries %.0f, found %llu, lost %llu)\n", name, NB, N, ppr, filler, ms,
           n / ms / 1e6, n * 128.0 / F::GROUP / ms / 1e9, (h[0] == h[1] + 0 && h[4] + h[2] == (unsigned long long)n) ? (h[2] ? "count OK (some lost)" : "OK") : "MISMATCH", n, h[4], h[2]);
    CK(hipFree(buf)); CK(hipFree(cnt)); CK(hipFree(sums));
    return 0;
}

int main()
{
    if (getenv("BINSP_ONLY")) {
        for (int filler : {0, 40}) {
            run3<uint64_t, 6>("Bins3 u64", 8, 1184, 1, filler);
            runp<PFmt6, 6>("BinsP 48-bit", 8, 1184, 1, filler);
            run3<uint32_t, 5>("Bins3 u32", 8, 1184, 3, filler);
            runp<PFmt3, 5>("BinsP 24-bit ppr 3", 8, 1184, 3, filler);
            runp<PFmt3, 5>("BinsP 24-bit ppr 4", 8, 1184, 4, filler);
        }
        run3<uint64_t, 4>("Bins3 u64 split-like", 8, 1776, 1, 0);
        runp<PFmt6, 4>("BinsP 48-bit split-like", 8, 1776, 1, 0);
        runp<PFmt6, 5>("BinsP 48-bit split-like N5", 8, 1420, 1, 0);
        run3<uint32_t, 8>("Bins3 u32 split-like N8x2", 8, 1184, 2, 0);
        runp<PFmt3, 8>("BinsP 24-bit split-like N8x2", 8, 1184, 2, 0);
        runp<PFmt3, 8>("BinsP 24-bit split-like N8x3", 8, 1184, 3, 0);
        run3<uint64_t, 6>("Bins3 u64 512 bins", 9, 1184, 1, 0);
        runp<PFmt6, 6>("BinsP 48-bit 512 bins", 9, 1184, 1, 0);
        runp<PFmt6, 3>("BinsP 48-bit 512 bins N3", 9, 2368, 1, 0);
        runp<PFmt6, 4>("BinsP 48-bit 512 bins N4", 9, 1776, 1, 0);
        runp<PFmt6, 6>("BinsP 48-bit 64 bins", 6, 1184, 1, 0);
        runp<PFmt6, 6>("BinsP 48-bit 16 bins ppr4", 4, 1184, 4, 0);
        runp<PFmt3, 5>("BinsP 24-bit 16 bins ppr8", 4, 1184, 8, 0);
        runp<PFmt6, 6>("BinsP 48-bit 64 bins skew", 6, 1184, 1, 1000);
        runp<PFmt3, 5>("BinsP 24-bit 512 bins", 9, 1184, 1, 0);
        return 0;
    }
    if (getenv("BINS3_ONLY")) {
        for (int filler : {0, 40}) {
            run<uint64_t, 6, false>("Bins  u64 (flush/round)", 8, 1184, 1, filler);
            run3<uint64_t, 6>("Bins3 u64", 8, 1184, 1, filler);
            run<uint32_t, 5, false>("Bins  u32 (flush/3 steps)", 8, 1184, 3, filler);
            run3<uint32_t, 5>("Bins3 u32", 8, 1184, 3, filler);
        }
        run<uint64_t, 6, false>("Bins  u64 8 bins skew", 3, 1184, 1, 1000);
        run3<uint64_t, 6>("Bins3 u64 8 bins skew", 3, 1184, 1, 1000);
        run<uint32_t, 2, false>("Bins  u32 8 bins skew", 3, 1184, 9, 1000);
        run3<uint32_t, 2>("Bins3 u32 8 bins skew", 3, 1184, 9, 1000);
        run<uint64_t, 6, false>("Bins  u64 64 bins skew", 6, 1184, 1, 1000);
        run3<uint64_t, 6>("Bins3 u64 64 bins skew", 6, 1184, 1, 1000);
        run<uint32_t, 2, false>("Bins  u32 8 bins ppr9", 3, 1184, 9, 0);
        run3<uint32_t, 2>("Bins3 u32 8 bins ppr9 (waits)", 3, 1184, 9, 0);
        run3<uint32_t, 2>("Bins3 u32 8 bins ppr10 (waits)", 3, 1184, 10, 0);
        run3<uint32_t, 4>("Bins3 u32 16 bins ppr9 (waits)", 4, 1184, 9, 0);
        run3<uint64_t, 6, 1>("Bins3 u64 stores->L2", 8, 1184, 1, 0);
        run3<uint64_t, 6, 2>("Bins3 u64 no copy", 8, 1184, 1, 0);
        run3<uint32_t, 5, 1>("Bins3 u32 stores->L2", 8, 1184, 3, 0);
        run3<uint32_t, 5, 2>("Bins3 u32 no copy", 8, 1184, 3, 0);
        run3<uint32_t, 5, 2>("Bins3 u32 no copy", 8, 1184, 2, 0);
        run<uint64_t, 4, false>("Bins  u64 split-like", 8, 1776, 1, 0);
        run3<uint64_t, 4>("Bins3 u64 split-like", 8, 1776, 1, 0);
        run3<uint32_t, 5>("Bins3 u32 64 bins", 6, 1184, 4, 0);
        run3<uint32_t, 5>("Bins3 u32 16 bins", 4, 1184, 4, 0);
        run3<uint32_t, 5>("Bins3 u32 2 bins (multi)", 1, 1184, 2, 0);
        run3<uint64_t, 6>("Bins3 u64 4 bins (multi)", 2, 1184, 1, 0);
        run3<uint64_t, 6>("Bins3 u64 8 bins (multi)", 3, 1184, 1, 0);
        run3<uint32_t, 5>("Bins3 u32 8 bins (multi)", 3, 1184, 3, 0);
        run3<uint64_t, 6>("Bins3 u64 128 bins", 7, 1184, 1, 0);
        run3<uint64_t, 6>("Bins3 u64 128 bins", 7, 1184, 2, 0);
        run3<uint32_t, 5>("Bins3 u32 512 bins", 9, 1184, 1, 0);
        run3<uint32_t, 5>("Bins3 u32 256 bins", 8, 1184, 2, 0);
        run3<uint32_t, 5>("Bins3 u32 256 bins", 8, 1184, 4, 0);
        run3<uint64_t, 6>("Bins3 u64 512 bins", 9, 1184, 1, 0);
        return 0;
    }
    run_half<uint64_t, 6>("Bins u64 half-size WGs", 8, 1184);
    run_half<uint32_t, 5>("Bins u32 half-size WGs", 8, 1184);
    // M2: query 1.86 G uint64 entries (6 per position), insert 1.55 G uint32 entries (5 per position)
    for (int filler : {0, 40}) {
        run<uint64_t, 6, false>("Bins  u64 (flush/round)", 8, 1184, 1, filler);
        run<uint64_t, 6, true>("RBins u64 (barrier-free)", 8, 1184, 1, filler);
        run<uint32_t, 5, false>("Bins  u32 (flush/3 steps)", 8, 1184, 3, filler);
        run<uint32_t, 5, true>("RBins u32 (barrier-free)", 8, 1184, 3, filler);
    }
    for (int d : {30, 60, 110, 200}) {
        run<uint64_t, 6, false>("Bins  u64 staggered x4", 8, 1184, 1, -d);
        run<uint32_t, 5, false>("Bins  u32 staggered x4", 8, 1184, 3, -d);
    }
    run<uint64_t, 4, false>("Bins  u64 split-like", 8, 1776, 1, 0);
    run<uint64_t, 4, true>("RBins u64 split-like", 8, 1776, 1, 0);
    run<uint64_t, 6, false>("Bins  u64 512 bins", 9, 1184, 1, 0);
    run<uint64_t, 6, true>("RBins u64 512 bins", 9, 1184, 1, 0);
    run<uint32_t, 5, true>("RBins u32 512 bins", 9, 1184, 1, 0);
    run<uint32_t, 5, true>("RBins u32 64 bins", 6, 1184, 1, 0);
    return 0;
}
