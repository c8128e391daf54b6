// Does an initialised HIP runtime make open() slow?  usage: open_hip <dir with files> <0|1: init HIP first>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dirent.h>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char** argv) {
    if (argc < 3) return 1;
    if (atoi(argv[2])) { void* p; hipMalloc(&p, 1 << 20); hipFree(p); }
    std::vector<std::string> names;
    DIR* d = opendir(argv[1]); while (dirent* e = readdir(d)) if (e->d_name[0] != '.') names.push_back(std::string(argv[1]) + "/" + e->d_name); closedir(d);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<int> fds(names.size());
    for (size_t i = 0; i < names.size(); i++) fds[i] = open(names[i].c_str(), O_RDONLY);
    auto t1 = std::chrono::steady_clock::now();
    for (int f : fds) close(f);
    auto t2 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int w = 0; w < 16; w++) th.emplace_back([&, w] { for (size_t i = w; i < names.size(); i += 16) fds[i] = open(names[i].c_str(), O_RDONLY); });
    for (auto& t : th) t.join();
    auto t3 = std::chrono::steady_clock::now();
    auto us = [&](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count() / names.size(); };
    printf("hip=%s files=%zu: open %.1f us, close %.1f us, open on 16 threads %.1f us per file\n", argv[2], names.size(), us(t0, t1), us(t1, t2), us(t2, t3));
    return 0;
}
