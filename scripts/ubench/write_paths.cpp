// Micro-benchmark: how fast can formatted SAM chunks reach a file?  (a) one thread write(), (b) T threads pwrite() at known offsets,
// (c) ftruncate + mmap of the batch's region + T threads memcpy + munmap.   usage: write_paths <file> <batches> <MB per batch> <threads>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char ** argv)
{
    const std::string path = argv[1];
    const int batches = atoi(argv[2]); const size_t mb = (size_t)atoi(argv[3]); const unsigned T = (unsigned)atoi(argv[4]);
    const size_t bytes = mb << 20;
    std::vector<char> src(bytes);
    for (size_t i = 0; i < bytes; ++i) src[i] = (char)('A' + i % 23);
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int mode = 0; mode < 3; ++mode)
    {
        int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
        const auto t0 = now();
        for (int b = 0; b < batches; ++b)
        {
            const size_t base = (size_t)b * bytes;
            if (mode == 0) { size_t at = 0; while (at < bytes) { ssize_t w = pwrite(fd, src.data() + at, bytes - at, (off_t)(base + at)); if (w <= 0) return 1; at += (size_t)w; } }
            else if (mode == 1)
            {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = bytes * t / T, hi = bytes * (t + 1) / T; while (lo < hi) { ssize_t w = pwrite(fd, src.data() + lo, hi - lo, (off_t)(base + lo)); if (w <= 0) return; lo += (size_t)w; } });
                for (auto & x : th) x.join();
            }
            else
            {
                if (ftruncate(fd, (off_t)(base + bytes)) != 0) return 2;
                char * m = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)base);
                if (m == MAP_FAILED) return 3;
                std::vector<std::thread> th;
                for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] { size_t lo = bytes * t / T, hi = bytes * (t + 1) / T; memcpy(m + lo, src.data() + lo, hi - lo); });
                for (auto & x : th) x.join();
                munmap(m, bytes);
            }
        }
        const double dt = std::chrono::duration<double>(now() - t0).count();
        close(fd);
        printf("%s: %.3f s, %.2f GB/s, %.1f ms per batch\n", mode == 0 ? "one thread pwrite" : mode == 1 ? "threads pwrite" : "mmap + threads memcpy", dt, batches * (double)bytes / dt / 1e9, 1e3 * dt / batches);
        unlink(path.c_str());
    }
    return 0;
}
