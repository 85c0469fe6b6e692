#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    hipFree(0);
    size_t tot = 0;
    for (size_t gb : {8, 9, 12, 4, 2, 13, 19, 4, 22, 31}) {
        void *p = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, gb << 30);
        auto t1 = std::chrono::steady_clock::now();
        tot += gb;
        printf("+%zu GiB (total %zu): malloc %.1f ms (%s)\n", gb, tot, std::chrono::duration<double, std::milli>(t1 - t0).count(), hipGetErrorString(e));
    }
    return 0;
}
