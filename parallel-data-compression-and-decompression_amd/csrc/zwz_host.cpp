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
#include <string>
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
        for (const auto& e : fs::recursive_directory_iterator(base)) {
            if (!e.is_regular_file()) continue;
            const std::string& full = e.path().native();
            const bool plain = !e.is_symlink() && full.size() > skip && full.compare(0, base_str.size(), base_str) == 0;
            files.push_back({plain ? full.substr(skip) : fs::relative(e.path(), base).string(), static_cast<off_t>(e.file_size())});
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
