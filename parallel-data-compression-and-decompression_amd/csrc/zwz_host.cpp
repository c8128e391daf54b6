// zwz_host.cpp -- the reference's per-rank pipeline around the codec, rebuilt on the batch API:
//   producer()+ConcurrenceQueue+consumer()+data_writer()  (compression.cpp:24-148)  ->  compress_dir
//   decompress_zwz()+do_decompression()                   (decompression.cpp:45-178) ->  decompress_dir
//   sort_files_by_size / count_non_empty_lines / md5_of_file (file_sort.cpp, file_tools.cpp,
//   verification.cpp)
// The queue of 65.5 KB Chunk copies becomes a pinned staging buffer of 65536-byte slots filled
// straight from the files; one H2D + one kernel pipeline + one D2H per slice of chunks; records are
// emitted in the reference's order (files in list order, chunks ascending).
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

namespace fs = std::filesystem;
using namespace zwz;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

namespace {

// ---- MD5 (RFC 1321), streaming --------------------------------------------------------------
struct Md5 {
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint64_t total = 0;
    uint8_t buf[64];
    uint32_t fill = 0;

    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        uint32_t w[16];
        memcpy(w, p, 64);  // little-endian host
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define R1(a, b, c, d, k, s, t) a = b + rol(a + ((b & c) | (~b & d)) + w[k] + t, s)
#define R2(a, b, c, d, k, s, t) a = b + rol(a + ((b & d) | (c & ~d)) + w[k] + t, s)
#define R3(a, b, c, d, k, s, t) a = b + rol(a + (b ^ c ^ d) + w[k] + t, s)
#define R4(a, b, c, d, k, s, t) a = b + rol(a + (c ^ (b | ~d)) + w[k] + t, s)
        R1(a,b,c,d,0,7,0xd76aa478u); R1(d,a,b,c,1,12,0xe8c7b756u); R1(c,d,a,b,2,17,0x242070dbu); R1(b,c,d,a,3,22,0xc1bdceeeu);
        R1(a,b,c,d,4,7,0xf57c0fafu); R1(d,a,b,c,5,12,0x4787c62au); R1(c,d,a,b,6,17,0xa8304613u); R1(b,c,d,a,7,22,0xfd469501u);
        R1(a,b,c,d,8,7,0x698098d8u); R1(d,a,b,c,9,12,0x8b44f7afu); R1(c,d,a,b,10,17,0xffff5bb1u); R1(b,c,d,a,11,22,0x895cd7beu);
        R1(a,b,c,d,12,7,0x6b901122u); R1(d,a,b,c,13,12,0xfd987193u); R1(c,d,a,b,14,17,0xa679438eu); R1(b,c,d,a,15,22,0x49b40821u);
        R2(a,b,c,d,1,5,0xf61e2562u); R2(d,a,b,c,6,9,0xc040b340u); R2(c,d,a,b,11,14,0x265e5a51u); R2(b,c,d,a,0,20,0xe9b6c7aau);
        R2(a,b,c,d,5,5,0xd62f105du); R2(d,a,b,c,10,9,0x02441453u); R2(c,d,a,b,15,14,0xd8a1e681u); R2(b,c,d,a,4,20,0xe7d3fbc8u);
        R2(a,b,c,d,9,5,0x21e1cde6u); R2(d,a,b,c,14,9,0xc33707d6u); R2(c,d,a,b,3,14,0xf4d50d87u); R2(b,c,d,a,8,20,0x455a14edu);
        R2(a,b,c,d,13,5,0xa9e3e905u); R2(d,a,b,c,2,9,0xfcefa3f8u); R2(c,d,a,b,7,14,0x676f02d9u); R2(b,c,d,a,12,20,0x8d2a4c8au);
        R3(a,b,c,d,5,4,0xfffa3942u); R3(d,a,b,c,8,11,0x8771f681u); R3(c,d,a,b,11,16,0x6d9d6122u); R3(b,c,d,a,14,23,0xfde5380cu);
        R3(a,b,c,d,1,4,0xa4beea44u); R3(d,a,b,c,4,11,0x4bdecfa9u); R3(c,d,a,b,7,16,0xf6bb4b60u); R3(b,c,d,a,10,23,0xbebfbc70u);
        R3(a,b,c,d,13,4,0x289b7ec6u); R3(d,a,b,c,0,11,0xeaa127fau); R3(c,d,a,b,3,16,0xd4ef3085u); R3(b,c,d,a,6,23,0x04881d05u);
        R3(a,b,c,d,9,4,0xd9d4d039u); R3(d,a,b,c,12,11,0xe6db99e5u); R3(c,d,a,b,15,16,0x1fa27cf8u); R3(b,c,d,a,2,23,0xc4ac5665u);
        R4(a,b,c,d,0,6,0xf4292244u); R4(d,a,b,c,7,10,0x432aff97u); R4(c,d,a,b,14,15,0xab9423a7u); R4(b,c,d,a,5,21,0xfc93a039u);
        R4(a,b,c,d,12,6,0x655b59c3u); R4(d,a,b,c,3,10,0x8f0ccc92u); R4(c,d,a,b,10,15,0xffeff47du); R4(b,c,d,a,1,21,0x85845dd1u);
        R4(a,b,c,d,8,6,0x6fa87e4fu); R4(d,a,b,c,15,10,0xfe2ce6e0u); R4(c,d,a,b,6,15,0xa3014314u); R4(b,c,d,a,13,21,0x4e0811a1u);
        R4(a,b,c,d,4,6,0xf7537e82u); R4(d,a,b,c,11,10,0xbd3af235u); R4(c,d,a,b,2,15,0x2ad7d2bbu); R4(b,c,d,a,9,21,0xeb86d391u);
#undef R1
#undef R2
#undef R3
#undef R4
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        if (fill) {
            size_t k = std::min<size_t>(64 - fill, n);
            memcpy(buf + fill, p, k); fill += (uint32_t)k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) { memcpy(buf, p, n); fill = (uint32_t)n; }
    }
    void hex(char out[33]) {
        uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (fill < 56 ? 56 : 120) - fill;
        uint64_t saved = total;
        update(pad, padlen);
        uint8_t lenb[8];
        for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
        update(lenb, 8);
        total = saved;
        static const char* dig = "0123456789abcdef";
        for (int i = 0; i < 16; i++) {
            uint8_t v = (uint8_t)(h[i >> 2] >> (8 * (i & 3)));
            out[2 * i] = dig[v >> 4]; out[2 * i + 1] = dig[v & 15];
        }
        out[32] = 0;
    }
};

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

struct PendingRecord { uint32_t file; int32_t seq; uint8_t last; };

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
        for (const auto& e : fs::recursive_directory_iterator(base))
            if (e.is_regular_file()) files.push_back({fs::relative(e.path(), base).string(), static_cast<off_t>(e.file_size())});
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

int zwz_compress_dir(zwz_ctx* c, const char* src_dir, const char* dst_dir, const char* file_record, int rank, int nranks) {
    if (!c || !src_dir || !dst_dir || !file_record || rank < 0 || nranks <= 0) return ZWZ_E_INVALID;
    std::vector<std::string> lines;
    if (!read_lines(file_record, lines)) { set_error("cannot open file record %s", file_record); return ZWZ_E_IO; }
    int non_empty = 0;
    for (const auto& s : lines) if (!s.empty() && !blank(s)) non_empty++;
    if (rank >= non_empty) return ZWZ_OK;   // main.cpp:47-51: this rank has nothing to do and creates no shard

    std::vector<std::string> mine;           // compression.cpp:35-41
    for (size_t i = (size_t)rank; i < lines.size(); i += (size_t)nranks) mine.push_back(lines[i]);

    const std::string out_path = (fs::path(dst_dir) / ("compressed_" + std::to_string(rank) + ".zwz")).string();
    FILE* dest = fopen(out_path.c_str(), "wb");
    if (!dest) { set_error("cannot create %s", out_path.c_str()); return ZWZ_E_IO; }
    std::vector<char> iobuf(8 << 20);
    setvbuf(dest, iobuf.data(), _IOFBF, iobuf.size());

    HIPCHK(hipSetDevice(c->device));
    const uint32_t slice = c->max_batch;
    int rc = ensure_staging(c, slice);
    if (rc) { fclose(dest); return rc; }
    StageView v = stage_view(c, slice);

    std::vector<PendingRecord> recs;
    std::vector<std::string> md5s(mine.size());
    recs.reserve(slice);
    uint32_t truncated = 0;

    auto flush = [&]() -> int {
        const uint32_t m = (uint32_t)recs.size();
        if (!m) return ZWZ_OK;
        HIPCHK(hipMemcpyAsync(v.d_in, v.h_in, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(v.d_off, v.h_off, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(v.d_len, v.h_len, m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        int r = zwz_deflate_batch_dev(c, v.d_in, v.d_off, v.d_len, m, v.d_out, ZWZ_DEV_STRIDE, v.d_olen);
        if (r) return r;
        HIPCHK(hipMemcpyAsync(v.h_out, v.d_out, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(v.h_olen, v.d_olen, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < m; i++) {   // data_writer(), compression.cpp:73-104
            const PendingRecord& r2 = recs[i];
            const std::string& rel = mine[r2.file];
            const int32_t path_len = (int32_t)rel.size();
            const int32_t payload = (int32_t)v.h_olen[i];
            const int32_t total = 4 + path_len + 4 + 1 + payload;
            fwrite(&total, 4, 1, dest);
            fwrite(&path_len, 4, 1, dest);
            fwrite(rel.data(), 1, (size_t)path_len, dest);
            fwrite(&r2.seq, 4, 1, dest);
            fwrite(&r2.last, 1, 1, dest);
            fwrite(v.h_out + (size_t)i * ZWZ_DEV_STRIDE, 1, (size_t)payload, dest);
            if (payload == (int32_t)ZWZ_CHUNK_SIZE && v.h_len[i] >= 65510u) truncated++;   // may have been cut
            if (r2.last) fwrite(md5s[r2.file].data(), 1, md5s[r2.file].size(), dest);
        }
        recs.clear();
        return ferror(dest) ? ZWZ_E_IO : ZWZ_OK;
    };

    for (uint32_t fi = 0; fi < mine.size() && rc == ZWZ_OK; fi++) {
        const std::string full = (fs::path(src_dir) / mine[fi]).string();
        int fd = open(full.c_str(), O_RDONLY);
        if (fd < 0) {   // compression.cpp:45-48: log and skip
            fprintf(stderr, "Error opening source file: \"%s\"\n", full.c_str());
            continue;
        }
        Md5 md5;
        int32_t seq = 0;
        for (;;) {      // compression.cpp:52-64: a short read is what ends the file
            if (recs.size() == slice) { rc = flush(); if (rc) break; }
            const uint32_t slot = (uint32_t)recs.size();
            uint8_t* dstp = v.h_in + (size_t)slot * ZWZ_DEV_STRIDE;
            size_t got = 0;
            while (got < ZWZ_CHUNK_SIZE) {
                ssize_t k = read(fd, dstp + got, ZWZ_CHUNK_SIZE - got);
                if (k < 0 && errno == EINTR) continue;
                if (k <= 0) break;
                got += (size_t)k;
            }
            md5.update(dstp, got);
            const uint8_t last = got < ZWZ_CHUNK_SIZE;
            v.h_off[slot] = (uint64_t)slot * ZWZ_DEV_STRIDE;
            v.h_len[slot] = (uint32_t)got;
            recs.push_back({fi, seq++, last});
            if (last) break;
        }
        close(fd);
        char hex[33];
        md5.hex(hex);
        md5s[fi] = hex;   // verification.cpp:24-27 rendering
    }
    if (rc == ZWZ_OK) rc = flush();
    if (fclose(dest) != 0 && rc == ZWZ_OK) rc = ZWZ_E_IO;
    if (truncated && verbose())
        fprintf(stderr, "zwz: %u chunk payload(s) reached the reference's 65535-byte cap (lossy, like the reference)\n", truncated);
    return rc;
}

int zwz_decompress_dir(zwz_ctx* c, const char* src_dir, const char* dst_dir, int* md5_mismatches) {
    if (!c || !src_dir || !dst_dir) return ZWZ_E_INVALID;
    if (md5_mismatches) *md5_mismatches = 0;
    std::vector<std::string> shards;
    try {
        for (const auto& e : fs::directory_iterator(src_dir))   // decompression.cpp:168-172
            if (e.path().extension() == ".zwz") shards.push_back(e.path().string());
    } catch (const std::exception& ex) { set_error("%s", ex.what()); return ZWZ_E_IO; }

    HIPCHK(hipSetDevice(c->device));
    const uint32_t slice = c->max_batch;
    int rc = ensure_staging(c, slice);
    if (rc) return rc;
    StageView v = stage_view(c, slice);
    int mismatches = 0;

    for (const std::string& shard : shards) {
        // ---- pass 1: parse records (decompression.cpp:65-92) and replay the reference's
        // per-path sequencing (expected id + pending heap, :119-153) to get, per output file
        // instance, the ordered list of records that get decoded into it.
        int fd = open(shard.c_str(), O_RDONLY);
        if (fd < 0) { fprintf(stderr, "Error opening file: %s\n", shard.c_str()); continue; }
        struct stat sb;
        fstat(fd, &sb);
        std::vector<uint8_t> blob((size_t)sb.st_size);
        size_t got = 0;
        while (got < blob.size()) { ssize_t k = read(fd, blob.data() + got, blob.size() - got); if (k <= 0) break; got += (size_t)k; }
        close(fd);
        blob.resize(got);

        struct Rec { uint32_t off, len; int32_t seq; uint8_t last; };
        struct FileInst { std::string rel; std::vector<Rec> order; std::multimap<int32_t, Rec> pending; int32_t expected = 0; std::string md5; bool finalised = false; };
        std::vector<FileInst> insts;
        std::map<std::string, size_t> open_inst;
        size_t p = 0;
        while (p + 4 <= blob.size()) {
            int32_t total, path_len, seq;
            memcpy(&total, &blob[p], 4); p += 4;
            if (p + 4 > blob.size()) break;
            memcpy(&path_len, &blob[p], 4); p += 4;
            if (path_len < 0 || p + (size_t)path_len + 5 > blob.size()) break;
            std::string rel(reinterpret_cast<const char*>(&blob[p]), (size_t)path_len); p += (size_t)path_len;
            memcpy(&seq, &blob[p], 4); p += 4;
            const uint8_t last = blob[p++];
            const int64_t plen = (int64_t)total - (4 + path_len + 4 + 1);
            if (plen < 0 || plen > (int64_t)ZWZ_CHUNK_SIZE || p + (size_t)plen > blob.size()) { rc = ZWZ_E_FORMAT; break; }
            Rec r{(uint32_t)p, (uint32_t)plen, seq, last};
            p += (size_t)plen;
            std::string md5;
            if (last) { if (p + ZWZ_MD5_HEX_LEN > blob.size()) { rc = ZWZ_E_FORMAT; break; } md5.assign(reinterpret_cast<const char*>(&blob[p]), ZWZ_MD5_HEX_LEN); p += ZWZ_MD5_HEX_LEN; }
            auto it = open_inst.find(rel);
            if (it == open_inst.end()) { insts.push_back(FileInst{}); insts.back().rel = rel; it = open_inst.emplace(rel, insts.size() - 1).first; }
            FileInst& fi = insts[it->second];
            if (last) fi.md5 = md5;
            if (fi.expected == seq) {
                fi.order.push_back(r); fi.expected++;
                for (auto pit = fi.pending.find(fi.expected); pit != fi.pending.end(); pit = fi.pending.find(fi.expected)) {
                    fi.order.push_back(pit->second); fi.pending.erase(pit); fi.expected++;
                }
                if (last && fi.expected == seq + 1 && fi.pending.empty()) { fi.finalised = true; open_inst.erase(it); }
            } else {
                fi.pending.emplace(seq, r);
            }
        }
        if (rc) { set_error("malformed shard %s", shard.c_str()); return rc; }

        // ---- pass 2: decode every scheduled record on the GPU, slice by slice, appending to files
        struct Job { uint32_t inst; Rec r; };
        std::vector<Job> jobs;
        for (uint32_t i = 0; i < insts.size(); i++) for (const Rec& r : insts[i].order) jobs.push_back({i, r});
        FILE* out = nullptr;
        uint32_t out_inst = UINT32_MAX;
        Md5 md5;
        std::vector<char> iobuf(4 << 20);
        auto finish_file = [&]() {
            if (!out) return;
            fclose(out); out = nullptr;
            FileInst& fi = insts[out_inst];
            if (fi.finalised) {   // decompression.cpp:132-149
                char hex[33];
                md5.hex(hex);
                const std::string file_path = std::string(dst_dir) + "/" + fi.rel;
                if (fi.md5 != hex) {
                    mismatches++;
                    fprintf(stderr, "MD5 mismatch for file: %s\n", file_path.c_str());
                    if (verbose()) printf("Expected MD5: %s\nCalculated MD5: %s\n", fi.md5.c_str(), hex);
                } else if (verbose()) printf("MD5 match for file: %s\n", file_path.c_str());
            } else if (!fi.pending.empty()) {
                fprintf(stderr, "Warning: pending chunks remaining for file: %s\n", fi.rel.c_str());
            }
        };
        // files that received no decodable record still get created (reference opens on first record)
        std::vector<bool> created(insts.size(), false);
        auto open_file = [&](uint32_t inst) -> bool {
            const std::string file_path = std::string(dst_dir) + "/" + insts[inst].rel;
            std::error_code ec;
            fs::path dir = fs::path(file_path).parent_path();
            if (!dir.empty() && !fs::exists(dir, ec)) fs::create_directories(dir, ec);
            out = fopen(file_path.c_str(), "wb");
            if (!out) { fprintf(stderr, "Error creating output file: %s\n", file_path.c_str()); return false; }
            setvbuf(out, iobuf.data(), _IOFBF, iobuf.size());
            out_inst = inst; md5 = Md5(); created[inst] = true;
            return true;
        };
        for (size_t done = 0; done < jobs.size(); done += slice) {
            const uint32_t m = (uint32_t)std::min<size_t>(slice, jobs.size() - done);
            for (uint32_t i = 0; i < m; i++) {
                const Rec& r = jobs[done + i].r;
                memcpy(v.h_in + (size_t)i * ZWZ_DEV_STRIDE, &blob[r.off], r.len);
                v.h_off[i] = (uint64_t)i * ZWZ_DEV_STRIDE; v.h_len[i] = r.len;
            }
            HIPCHK(hipMemcpyAsync(v.d_in, v.h_in, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(v.d_off, v.h_off, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(v.d_len, v.h_len, m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            int r2 = zwz_inflate_batch_dev(c, v.d_in, v.d_off, v.d_len, m, v.d_out, ZWZ_DEV_STRIDE, v.d_olen, v.d_status);
            if (r2) return r2;
            HIPCHK(hipMemcpyAsync(v.h_out, v.d_out, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(v.h_olen, v.d_olen, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t inst = jobs[done + i].inst;
                if (inst != out_inst || !out) { finish_file(); if (!open_file(inst)) { out_inst = inst; continue; } }
                const uint8_t* src = v.h_out + (size_t)i * ZWZ_DEV_STRIDE;
                fwrite(src, 1, v.h_olen[i], out);
                md5.update(src, v.h_olen[i]);
            }
        }
        finish_file();
        for (uint32_t i = 0; i < insts.size(); i++)
            if (!created[i]) { if (open_file(i)) { fclose(out); out = nullptr; } }
    }
    if (md5_mismatches) *md5_mismatches = mismatches;
    return ZWZ_OK;
}

}  // extern "C"
