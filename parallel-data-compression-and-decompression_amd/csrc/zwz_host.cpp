// zwz_host.cpp -- the reference's per-rank pipeline around the codec, rebuilt on the batch API:
//   sort_files_by_size / count_non_empty_lines / md5_of_file (file_sort.cpp, file_tools.cpp,
//   verification.cpp).  The per-rank compress / decompress pipelines live in zwz_pipeline.cpp.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <atomic>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "zwz_api_internal.h"
#include "zwz_md5.h"

namespace fs = std::filesystem;
using namespace zwz;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

namespace {

bool read_lines(const std::string& path, std::vector<std::string>& lines) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string s;
    while (std::getline(f, s)) lines.push_back(s);
    return true;
}

bool blank(const std::string& s) {
    return std::all_of(s.begin(), s.end(), [](unsigned char ch) { return std::isspace(ch) != 0; });
}

bool verbose() { static int v = getenv("ZWZ_VERBOSE") ? 1 : 0; return v != 0; }

}  // namespace

extern "C" {

int zwz_sort_files_by_size(const char* src_dir, char* record_path_out, size_t cap) {
    if (!src_dir) return ZWZ_E_INVALID;
    try {
        // same enumeration + comparator as file_sort.cpp:14-31, so ties come out in the same
        // (readdir + std::sort) order on the same box
        struct Entry { std::string rel; off_t size; };
        std::vector<Entry> files;
        fs::path base(src_dir);
        // fs::relative() canonicalises both paths -- several lstat() calls per file, 11 us of the 12 us an entry cost at
        // 370 000 files (BASELINE configs[3]: 4 s before a byte was read).  The iterator does not descend into symlinked
        // directories, so for every entry that is not itself a symlink the canonical path is the base's canonical path plus
        // the entry's own tail: the relative path is that tail, taken lexically.  Symlinks keep the reference's call.
        const std::string base_str = base.string();
        const size_t skip = base_str.size() + (base_str.empty() || base_str.back() == '/' ? 0 : 1);
        auto take = [&](const fs::directory_entry& e, std::vector<Entry>& into) {
            if (!e.is_regular_file()) return;
            const std::string& full = e.path().native();
            const bool plain = !e.is_symlink() && full.size() > skip && full.compare(0, base_str.size(), base_str) == 0;
            into.push_back({plain ? full.substr(skip) : fs::relative(e.path(), base).string(), static_cast<off_t>(e.file_size())});
        };
        // The walk is the reference's -- recursive_directory_iterator: a directory's entries in readdir order, a subdirectory's contents right
        // behind its own entry -- but the subtrees under the top directory are walked by a few threads at once and spliced back in that order
        // (370 000 files in 997 x 13 directories: 1.0 s of readdir + stat on one thread, a third of `main compress`).  The sequence handed
        // to std::sort is the same, so ties come out the same.
        struct Sub { size_t at; fs::path dir; std::vector<Entry> found; std::string error; };
        std::vector<Sub> subs;
        for (const auto& e : fs::directory_iterator(base)) {
            if (e.is_directory() && !e.is_symlink()) subs.push_back({files.size(), e.path(), {}, {}});     // (the iterator does not descend into symlinked directories)
            else take(e, files);
        }
        if (!subs.empty()) {
            unsigned hc = std::thread::hardware_concurrency();
            const unsigned nthreads = (unsigned)std::min<size_t>(subs.size(), hc <= 16u ? std::max(1u, hc) : std::min(32u, hc / 4u));
            std::atomic<size_t> next{0};
            auto work = [&] {
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= subs.size()) return;
                    try { for (const auto& e : fs::recursive_directory_iterator(subs[i].dir)) take(e, subs[i].found); }
                    catch (const std::exception& ex) { subs[i].error = ex.what(); }
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nthreads; t++) { try { pool.emplace_back(work); } catch (const std::system_error&) { break; } }
            work();
            for (auto& t : pool) t.join();
            std::vector<Entry> all;
            size_t from = 0;
            for (Sub& sb : subs) {
                if (!sb.error.empty()) { set_error("%s", sb.error.c_str()); return ZWZ_E_IO; }
                all.insert(all.end(), std::make_move_iterator(files.begin() + from), std::make_move_iterator(files.begin() + sb.at));
                all.insert(all.end(), std::make_move_iterator(sb.found.begin()), std::make_move_iterator(sb.found.end()));
                from = sb.at;
            }
            all.insert(all.end(), std::make_move_iterator(files.begin() + from), std::make_move_iterator(files.end()));
            files.swap(all);
        }
        std::sort(files.begin(), files.end(), [](const Entry& a, const Entry& b) { return a.size > b.size; });
        fs::path out = base.parent_path() / "sorted_files_by_size.txt";   // file_sort.cpp:33
        std::ofstream f(out);
        if (!f.is_open()) { set_error("cannot write %s", out.c_str()); return ZWZ_E_IO; }
        for (const auto& e : files) f << e.rel << "\n";
        if (record_path_out && cap) snprintf(record_path_out, cap, "%s", out.string().c_str());
        return ZWZ_OK;
    } catch (const std::exception& ex) {
        set_error("%s", ex.what());
        return ZWZ_E_IO;
    }
}

int zwz_count_non_empty_lines(const char* file_path) {
    std::vector<std::string> lines;
    if (!file_path || !read_lines(file_path, lines)) return -1;   // file_tools.cpp:8-11
    int n = 0;
    for (const auto& s : lines) if (!s.empty() && !blank(s)) n++;
    return n;
}

int zwz_md5_of_file(const char* path, char hex_out[33]) {
    if (!path || !hex_out) return ZWZ_E_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) { hex_out[0] = 0; set_error("cannot open %s", path); return ZWZ_E_IO; }   // verification.cpp:8-11 returns ""
    Md5 m;
    std::vector<uint8_t> buf(1 << 20);
    size_t k;
    while ((k = fread(buf.data(), 1, buf.size(), f)) > 0) m.update(buf.data(), k);
    fclose(f);
    m.hex(hex_out);
    return ZWZ_OK;
}

}  // extern "C"
