// io_probe.cpp -- how fast this host takes a 350-MB append into /dev/shm: one pwrite, parallel pwrites, parallel stores through a shared mapping
// (the sink of the BLOW5 writer's stored-block mode); build: g++ -O2 -o tools/bin/io_probe tools/io_probe.cpp -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t n = (size_t)350 << 20; const int nth = argc > 1 ? atoi(argv[1]) : 16; const int reps = 8;
    std::vector<char> buf(n, 1);
    for (int nf : {2, 4, 8}) {                                       // one pwrite stream per FILE, nf files side by side
        std::vector<int> fds;
        for (int f = 0; f < nf; f++) { char pth[64]; snprintf(pth, sizeof pth, "/dev/shm/wtest.%d.bin", f); unlink(pth); fds.push_back(open(pth, O_RDWR | O_CREAT | O_TRUNC, 0644)); }
        double t0 = now();
        for (int r = 0; r < reps; r++) {
            std::vector<std::thread> th;
            for (int f = 0; f < nf; f++) th.emplace_back([&, f, r] { size_t lo = n * f / nf, hi = n * (f + 1) / nf; off_t at = (off_t)r * (hi - lo) + 100; while (lo < hi) { ssize_t k = pwrite(fds[f], buf.data() + lo, hi - lo, at); lo += k; at += k; } });
            for (auto& t : th) t.join();
        }
        double dt = now() - t0;
        printf("%d files, one pwrite stream each: %.2f GB/s\n", nf, reps * n / dt / 1e9);
        for (int f = 0; f < nf; f++) { close(fds[f]); char pth[64]; snprintf(pth, sizeof pth, "/dev/shm/wtest.%d.bin", f); unlink(pth); }
    }
    for (int mode = 0; mode < 4; mode++) {
        const char* path = "/dev/shm/wtest.bin"; unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now(); off_t base = 100;
        for (int r = 0; r < reps; r++) {
            if (mode == 0) { size_t lo = 0; while (lo < n) lo += pwrite(fd, buf.data() + lo, n - lo, base + lo); }
            else if (mode == 1) { std::vector<std::thread> th; for (int t = 0; t < nth; t++) th.emplace_back([&, t] { size_t lo = n * t / nth, hi = n * (t + 1) / nth; while (lo < hi) lo += pwrite(fd, buf.data() + lo, hi - lo, base + lo); }); for (auto& t : th) t.join(); }
            else {
                ftruncate(fd, base + n);
                off_t m0 = base & ~4095L; size_t ml = n + (base - m0);
                char* m = (char*)mmap(nullptr, ml, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 3 ? MAP_POPULATE : 0), fd, m0);
                std::vector<std::thread> th; for (int t = 0; t < nth; t++) th.emplace_back([&, t] { size_t lo = n * t / nth, hi = n * (t + 1) / nth; memcpy(m + (base - m0) + lo, buf.data() + lo, hi - lo); }); for (auto& t : th) t.join();
                munmap(m, ml);
            }
            base += n;
        }
        double dt = now() - t0;
        printf("mode %d (%s), %d threads: %.2f GB/s\n", mode, mode == 0 ? "one pwrite" : mode == 1 ? "parallel pwrite" : mode == 2 ? "mmap + parallel memcpy" : "mmap populate + memcpy", nth, reps * n / dt / 1e9);
        close(fd); unlink(path);
    }
}
