// How fast can N files of S bytes be created in ONE directory by T threads (fopen/fwrite/fclose as decode_whole_shard's tasks do)?
// `main decompress` of 10 000 x 256 KiB files waits 1.1 s for its sixteen writers: are they waiting for the directory's lock?
//   g++ -O2 -pthread -o file_create.bin file_create.cpp; ./file_create.bin <dir> [files] [bytes]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const std::string base = std::string(argc > 1 ? argv[1] : "/tmp") + "/zwz_file_create_test";
    const int n = argc > 2 ? atoi(argv[2]) : 10000; const size_t sz = argc > 3 ? (size_t)atol(argv[3]) : 262144;
    std::vector<char> buf(sz, 'x');
    for (int mode = 0; mode < 2; mode++)
    for (int T : {1, 2, 4, 8, 16, 32}) {
        (void)system(("rm -rf " + base).c_str()); mkdir(base.c_str(), 0777);
        if (mode) for (int d = 0; d < 64; d++) mkdir((base + "/d" + std::to_string(d)).c_str(), 0777);
        std::atomic<int> next{0};
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&] {
            for (;;) { const int i = next.fetch_add(1); if (i >= n) return;
                const std::string p = base + (mode ? "/d" + std::to_string(i % 64) : "") + "/f" + std::to_string(i);
                const int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
                for (size_t k = 0; k < sz;) { ssize_t w = write(fd, buf.data() + k, std::min<size_t>(65535, sz - k)); if (w <= 0) break; k += (size_t)w; }
                close(fd); } });
        for (auto& x : th) x.join();
        const double dt = now() - t0;
        printf("%s %2d threads: %d files x %zu B in %.3f s = %.1f us a file, %.2f GB/s\n", mode ? "64 directories," : "one directory, ", T, n, sz, dt, dt / n * 1e6, n * sz / dt / 1e9);
    }
    (void)system(("rm -rf " + base).c_str());
}
