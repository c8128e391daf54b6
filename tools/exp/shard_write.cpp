// How fast can ONE file be written from pinned-like memory by N threads?  (`main compress` on incompressible data is bound by the
// shard's writers.)  pwrite from N threads (the kernel serialises buffered writes to one inode), against a shared mapping of the
// file filled by N threads (page faults in parallel), against one thread.   g++ -O2 -pthread -o shard_write.bin shard_write.cpp; ./shard_write.bin <dir> [GB] [threads]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)(atof(argc > 2 ? argv[2] : "2.6") * 1e9) & ~(size_t)0xfffff;
    const int T = argc > 3 ? atoi(argv[3]) : 16;
    const size_t piece = 8 << 20;
    std::vector<char> src(piece * T);
    for (size_t i = 0; i < src.size(); i++) src[i] = (char)(i * 2654435761u >> 24);
    auto run = [&](const char* name, int threads, int mode) {
        const std::string path = dir + "/zwz_shard_write_test.bin";
        unlink(path.c_str());
        const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
        const double t0 = now();
        char* map = nullptr;
        if (mode == 1) { if (posix_fallocate(fd, 0, (off_t)total) != 0) { perror("fallocate"); } map = (char*)mmap(nullptr, total, PROT_WRITE, MAP_SHARED, fd, 0); }
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t] {
                for (size_t off = (size_t)t * piece; off < total; off += (size_t)threads * piece) {
                    const size_t n = std::min(piece, total - off);
                    if (mode == 1) memcpy(map + off, src.data() + (size_t)t * piece, n);
                    else { size_t k = 0; while (k < n) { ssize_t w = pwrite(fd, src.data() + (size_t)t * piece + k, n - k, (off_t)(off + k)); if (w <= 0) break; k += (size_t)w; } }
                }
            });
        for (auto& x : th) x.join();
        if (map) munmap(map, total);
        const double t1 = now();
        close(fd);
        printf("%-28s %2d threads: %.3f s = %.2f GB/s\n", name, threads, t1 - t0, total / (t1 - t0) / 1e9);
        unlink(path.c_str());
    };
    run("pwrite", 1, 0); run("pwrite", T, 0); run("mmap + memcpy (fallocate)", T, 1); run("mmap + memcpy (fallocate)", 1, 1);
    return 0;
}
