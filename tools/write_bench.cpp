// write_bench.cpp -- how fast can 528 MB (M2's junction stream) reach a fresh file?  DESIGN.md section 7, end to end.
// g++ -O2 -pthread tools/write_bench.cpp -o tools/write_bench ; tools/write_bench /tmp/wb.bin [MB] [threads]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <class F> static void parallel(int threads, F f)
{
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([=]() { f(t); });
    for (auto &th : pool) th.join();
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/tmp/wb.bin";
    const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 528) << 20;
    const int threads = argc > 3 ? atoi(argv[3]) : 16;
    std::vector<char> src(bytes);
    for (size_t i = 0; i < bytes; i += 64) src[i] = (char)i;  // touched
    const size_t chunk = 4 << 20;
    const size_t n_chunks = (bytes + chunk - 1) / chunk;
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            unlink(path);
            const double t0 = now_ms();
            int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0644);
            if (fd < 0) { perror("open"); return 1; }
            const char *name = "";
            if (mode == 0) {
                name = "pwrite, 1 thread, 4 MiB chunks";
                for (size_t c = 0; c < n_chunks; c++) { const size_t o = c * chunk, n = std::min(chunk, bytes - o); if (pwrite(fd, src.data() + o, n, o) != (ssize_t)n) return 2; }
            } else if (mode == 1) {
                name = "pwrite, N threads, interleaved 4 MiB chunks";
                parallel(threads, [&](int t) { for (size_t c = t; c < n_chunks; c += threads) { const size_t o = c * chunk, n = std::min(chunk, bytes - o); if (pwrite(fd, src.data() + o, n, o) != (ssize_t)n) _exit(2); } });
            } else if (mode == 2) {
                name = "fallocate + pwrite, N threads";
                if (posix_fallocate(fd, 0, bytes) != 0) perror("fallocate");
                parallel(threads, [&](int t) { for (size_t c = t; c < n_chunks; c += threads) { const size_t o = c * chunk, n = std::min(chunk, bytes - o); if (pwrite(fd, src.data() + o, n, o) != (ssize_t)n) _exit(2); } });
            } else if (mode == 3 || mode == 4) {
                name = mode == 3 ? "ftruncate + mmap + memcpy, N threads" : "ftruncate + mmap(MAP_POPULATE) + memcpy, N threads";
                if (ftruncate(fd, bytes) != 0) perror("ftruncate");
                char *m = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 4 ? MAP_POPULATE : 0), fd, 0);
                if (m == MAP_FAILED) { perror("mmap"); return 3; }
                parallel(threads, [&](int t) { for (size_t c = t; c < n_chunks; c += threads) { const size_t o = c * chunk, n = std::min(chunk, bytes - o); memcpy(m + o, src.data() + o, n); } });
                munmap(m, bytes);
            } else {
                name = "N files' worth: pwrite into N separate files (upper bound without the per-file lock)";
                close(fd);
                parallel(threads, [&](int t) {
                    char p[512]; snprintf(p, sizeof p, "%s.%d", path, t);
                    unlink(p);
                    int f = open(p, O_CREAT | O_TRUNC | O_RDWR, 0644);
                    for (size_t c = t; c < n_chunks; c += threads) { const size_t o = c * chunk, n = std::min(chunk, bytes - o); if (write(f, src.data() + o, n) != (ssize_t)n) _exit(2); }
                    close(f);
                });
                fd = open(path, O_RDWR);
            }
            close(fd);
            const double t1 = now_ms();
            printf("%-90s %7.1f ms  %6.2f GB/s\n", name, t1 - t0, bytes / (t1 - t0) / 1e6);
            if (mode == 5) for (int t = 0; t < threads; t++) { char p[512]; snprintf(p, sizeof p, "%s.%d", path, t); unlink(p); }
        }
    }
    unlink(path);
    return 0;
}
