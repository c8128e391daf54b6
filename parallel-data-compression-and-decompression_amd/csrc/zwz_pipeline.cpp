// zwz_pipeline.cpp -- the reference's per-rank pipelines, rebuilt around the batch codec:
//   producer() + ConcurrenceQueue<Chunk> + consumer() + data_writer()   (compression.cpp:24-148)
//   decompress_zwz() + do_decompression()                                (decompression.cpp:45-178)
//
// The reference moves every chunk through a mutex queue as a 65.5 KB struct copy (twice), calls
// zlib once per chunk on one thread and computes each file's MD5 inside the writer lock.  Here:
//   * chunks are pread() straight into 65536-byte slots of a pinned staging buffer by a small
//     thread pool (slot positions are known up front from the file sizes);
//   * a slice of chunks costs one H2D, one kernel pipeline and one D2H on the context's stream;
//   * slices are double-buffered: while the GPU works on slice s the pool reads slice s+1 and the
//     caller's thread writes the records of slice s-1;
//   * MD5 runs on the pool, off the critical path (a file that fits in one slice is hashed from the
//     staging buffer right after it was read; larger files are re-read, like md5_of_file does).
// Record order is the reference's: files in list order restricted to this rank, chunks ascending
// (SURVEY.md Appendix A), so shards stay byte-identical.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <filesystem>
#include <fstream>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include "zwz_api_internal.h"
#include "zwz_md5.h"

namespace fs = std::filesystem;
using namespace zwz;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

namespace {

bool verbose() { static int v = getenv("ZWZ_VERBOSE") ? 1 : 0; return v != 0; }
// ZWZ_TIMELINE=1: the phase timelines alone (stderr), without ZWZ_VERBOSE's per-file messages -- 10 000 "MD5 match" lines through a pipe
// are a load of their own (tools/e2e.py measures with this)
bool timeline() { static int v = (getenv("ZWZ_TIMELINE") || getenv("ZWZ_VERBOSE")) ? 1 : 0; return v != 0; }

// Chunks per staging slice.  A slice costs 2 x 64 KiB of pinned host memory per chunk, twice (double buffering), and pinning is
// slow (0.2 s for 4096-chunk slices, a quarter of a 2 GB job); 2048 chunks still give every CU eight chunks a launch.
constexpr uint32_t kSliceChunks = 2048;

bool blank(const std::string& s) {
    return std::all_of(s.begin(), s.end(), [](unsigned char ch) { return std::isspace(ch) != 0; });
}

// Fixed pool of host workers with countable task groups.
class Pool {
public:
    struct Group { std::atomic<int> pending{0}; };
    explicit Pool(unsigned n) {
        for (unsigned i = 0; i < n; i++) {
            const bool prefers_bg = n >= 4 && i % 4 == 3;
            try { workers_.emplace_back([this, prefers_bg] { run(prefers_bg); }); }
            catch (const std::system_error&) { if (workers_.empty()) throw; break; }   // thread limit: run with what we have
        }
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // background = true: work nothing downstream waits for soon (whole-file MD5 of big files).  Workers take foreground
    // tasks first, so a slice's reads never queue behind every big file's hash (each ~1.5 s per GB); one worker in four
    // prefers background work, so hashing always makes progress.
    void submit(Group& g, std::function<void()> fn, bool background = false) {
        g.pending.fetch_add(1);
        { std::lock_guard<std::mutex> l(m_); (background ? bg_ : q_).push_back({&g, std::move(fn)}); }
        cv_.notify_one();
    }
    void wait(Group& g) {
        std::unique_lock<std::mutex> l(m_);
        done_cv_.wait(l, [&] { return g.pending.load() == 0; });
    }
    bool failed(std::string* why = nullptr) {
        std::lock_guard<std::mutex> l(m_);
        if (why) *why = failure_;
        return has_failure_;
    }
private:
    void note_failure(const char* what) {
        std::lock_guard<std::mutex> l(m_);
        if (!has_failure_) { has_failure_ = true; failure_ = what ? what : ""; }
    }
    bool has_failure_ = false;
    std::string failure_;
    struct Task { Group* g; std::function<void()> fn; };
    void run(bool prefers_bg) {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || !q_.empty() || !bg_.empty(); });
                if (q_.empty() && bg_.empty()) return;
                std::deque<Task>& from = (q_.empty() || (prefers_bg && !bg_.empty())) ? bg_ : q_;
                t = std::move(from.front()); from.pop_front();
            }
            // a task that throws (std::bad_alloc in a vector, a std::filesystem error) must not unwind out of the thread:
            // that is std::terminate -> abort() with nothing but a line on stderr.  It is recorded and the job fails.
            try { t.fn(); } catch (const std::exception& ex) { note_failure(ex.what()); } catch (...) { note_failure("unknown exception"); }
            if (t.g->pending.fetch_sub(1) == 1) { std::lock_guard<std::mutex> l(m_); done_cv_.notify_all(); }
        }
    }
    std::vector<std::thread> workers_;
    std::deque<Task> q_, bg_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    bool stop_ = false;
};

unsigned host_threads() {
    if (const char* v = getenv("ZWZ_HOST_THREADS")) return (unsigned)std::max(1, atoi(v));
    // Sixteen at most.  Measured on the GPU box (256 logical CPUs, overlay file system), BASELINE configs[1-3] through `main`: with 64 workers every
    // stage that touches the file system got SLOWER -- the shard is one file (its writers queue on the inode lock), 10 000 output files land
    // in one directory (their creates queue on the directory's), stat() of 370 000 files 0.25 -> 0.32 s -- slices of text 0.42 -> 0.51 s,
    // decompress 0.77 -> 1.41 s.  The workers mostly wait for the kernel's locks, and more of them wait less efficiently.
    unsigned hc = std::thread::hardware_concurrency();
    return std::min(16u, std::max(2u, hc));
}

// Two staging slices (in / out slots, offsets, lengths, status), pinned on the host and mirrored on
// the device; reuses the context's staging allocation.
struct Slices {
    uint32_t cap = 0;      // chunks per slice
    uint8_t *h_in[2], *h_out[2], *d_in[2], *d_out[2];
    uint64_t *h_off[2], *d_off[2];
    uint32_t *h_len[2], *h_olen[2], *h_st[2], *d_len[2], *d_olen[2], *d_st[2];
    hipEvent_t done[2] = {nullptr, nullptr};
    // Three queues a slice passes through -- copy in (s_in), the codec's own stream, copy out (s_out) -- so that slice s + 1 arrives and
    // slice s - 1 leaves while slice s is in the kernels (PCIe is full duplex; on ONE stream a text slice took 16.8 ms for ~5 ms of kernels).
    // ev_in[b]: buffer pair b has arrived; ev_k[b]: its kernels are done (its d_in may be overwritten, its d_out copied out).
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr};
    // GPU MD5 of the files that lie whole inside a slice (zwz_md5_files_dev): (first slot, slots) per file in, 16 bytes out
    uint32_t *h_files[2] = {nullptr, nullptr}, *d_files[2] = {nullptr, nullptr};
    uint8_t *h_dig[2] = {nullptr, nullptr}, *d_dig[2] = {nullptr, nullptr};
    uint32_t n_md5[2] = {0, 0};
    void* md5_host = nullptr; void* md5_dev = nullptr;
};

int make_slices(zwz_ctx* c, uint32_t cap, Slices& s) {
    int rc = ensure_staging(c, 2 * cap);
    if (rc) return rc;
    s.cap = cap;
    auto carve = [&](uint8_t* base, int i, uint8_t*& in, uint8_t*& out, uint64_t*& off, uint32_t*& len, uint32_t*& olen, uint32_t*& st) {
        const size_t total = c->stage_chunks;           // layout of stage_view(): [in | out | off | len | olen | st]
        uint8_t* in0 = base; uint8_t* out0 = in0 + total * ZWZ_DEV_STRIDE;
        uint64_t* off0 = reinterpret_cast<uint64_t*>(out0 + total * ZWZ_DEV_STRIDE);
        uint32_t* len0 = reinterpret_cast<uint32_t*>(off0 + total); uint32_t* olen0 = len0 + total; uint32_t* st0 = olen0 + total;
        const size_t o = (size_t)i * cap;
        in = in0 + o * ZWZ_DEV_STRIDE; out = out0 + o * ZWZ_DEV_STRIDE; off = off0 + o; len = len0 + o; olen = olen0 + o; st = st0 + o;
    };
    for (int i = 0; i < 2; i++) {
        carve(static_cast<uint8_t*>(c->h_stage), i, s.h_in[i], s.h_out[i], s.h_off[i], s.h_len[i], s.h_olen[i], s.h_st[i]);
        carve(static_cast<uint8_t*>(c->d_stage), i, s.d_in[i], s.d_out[i], s.d_off[i], s.d_len[i], s.d_olen[i], s.d_st[i]);
        hipError_t e = hipEventCreateWithFlags(&s.done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_in[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_k[i], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
    }
    {
        hipError_t e = hipStreamCreateWithFlags(&s.s_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.s_out, hipStreamNonBlocking);
        if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
    }
    const size_t per = (size_t)cap * (8 + 16);                // a slice holds at most `cap` files
    hipError_t e = hipHostMalloc(&s.md5_host, 2 * per, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&s.md5_dev, 2 * per);
    if (e != hipSuccess) return hip_fail(e, "MD5 staging allocation");
    for (int i = 0; i < 2; i++) {
        uint8_t* hb = static_cast<uint8_t*>(s.md5_host) + i * per; uint8_t* db = static_cast<uint8_t*>(s.md5_dev) + i * per;
        s.h_dig[i] = hb; s.h_files[i] = reinterpret_cast<uint32_t*>(hb + (size_t)cap * 16);
        s.d_dig[i] = db; s.d_files[i] = reinterpret_cast<uint32_t*>(db + (size_t)cap * 16);
    }
    return ZWZ_OK;
}

void free_slices(Slices& s) {                 // (idempotent: scope guards call it again behind the timed call of the normal path)
    if (s.s_in) { (void)hipStreamSynchronize(s.s_in); (void)hipStreamDestroy(s.s_in); s.s_in = nullptr; }
    if (s.s_out) { (void)hipStreamSynchronize(s.s_out); (void)hipStreamDestroy(s.s_out); s.s_out = nullptr; }
    for (auto& e : s.done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : s.ev_in) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : s.ev_k) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (s.md5_host) { (void)hipHostFree(s.md5_host); s.md5_host = nullptr; }
    if (s.md5_dev) { (void)hipFree(s.md5_dev); s.md5_dev = nullptr; }
}

// Waits for every listed task group when its scope is left, however that happens: tasks capture the enclosing function's
// locals by reference, so nothing may still be running when those locals go (declared last = destroyed first).
struct Drain {
    Pool& pool; std::vector<Pool::Group*> groups;
    ~Drain() { for (Pool::Group* g : groups) pool.wait(*g); }
};

// Raw bytes per Chunk.  process.hpp:12 fixes 65535 and the shards are bit-exact only with it (the default).  SURVEY.md
// section 8 f4, opt-in and never default: ZWZ_LOSSLESS=1 cuts files every 65509 bytes instead -- the largest size whose level-6
// stream always fits the reference's 65535-byte payload buffer (a chunk of fewer than 65533 bytes has at most 65532
// symbols, i.e. at most 4 blocks; stored worst case: 2 + 4 x 5 + 65509 + 4 = 65535), so no chunk is ever truncated
// and the round trip is exact; ZWZ_CHUNK_SIZE=<n <= 65535> sets any other size.  The container is
// unchanged and the reference's own decoder reads such shards (it takes any chunk that decodes to <= 65535 bytes).
uint32_t chunk_bytes_for(const zwz_ctx* c) {
    if (c->chunk_bytes) return c->chunk_bytes;
    if (const char* v = getenv("ZWZ_CHUNK_SIZE")) { const long n = atol(v); if (n >= 1 && n <= (long)ZWZ_CHUNK_SIZE) return (uint32_t)n; }
    if (const char* v = getenv("ZWZ_LOSSLESS")) if (*v && *v != '0') return ZWZ_LOSSLESS_CHUNK_SIZE;
    return ZWZ_CHUNK_SIZE;
}

int compress_dir_impl(zwz_ctx* c, const char* src_dir, const char* dst_dir, const char* file_record, int rank, int nranks);
int decompress_dir_impl(zwz_ctx* c, const char* src_dir, const char* dst_dir, int rank, int nranks, zwz_allgather_u64_fn exchange,
                        void* user, int* md5_mismatches);

// No exception crosses the C ABI (include/zwz.h): anything thrown below an entry point becomes a status code.
template <class F> int guarded(const char* what, F&& body) {
    try { return body(); }
    catch (const std::bad_alloc&) { set_error("%s: out of host memory", what); return ZWZ_E_NOMEM; }
    catch (const std::exception& ex) { set_error("%s: %s", what, ex.what()); return ZWZ_E_IO; }
    catch (...) { set_error("%s: unknown exception", what); return ZWZ_E_IO; }
}

}  // namespace

extern "C" {

int zwz_compress_dir(zwz_ctx* c, const char* src_dir, const char* dst_dir, const char* file_record, int rank, int nranks) {
    if (!c || !src_dir || !dst_dir || !file_record || rank < 0 || nranks <= 0) return ZWZ_E_INVALID;
    return guarded("zwz_compress_dir", [&] { return compress_dir_impl(c, src_dir, dst_dir, file_record, rank, nranks); });
}

int zwz_decompress_dir(zwz_ctx* c, const char* src_dir, const char* dst_dir, int* md5_mismatches) {
    return zwz_decompress_dir_ranked(c, src_dir, dst_dir, 0, 1, nullptr, nullptr, md5_mismatches);
}

int zwz_decompress_dir_ranked(zwz_ctx* c, const char* src_dir, const char* dst_dir, int rank, int nranks, zwz_allgather_u64_fn exchange,
                              void* user, int* md5_mismatches) {
    if (!c || !src_dir || !dst_dir || rank < 0 || nranks <= 0 || rank >= nranks) return ZWZ_E_INVALID;
    if (md5_mismatches) *md5_mismatches = 0;
    return guarded("zwz_decompress_dir", [&] { return decompress_dir_impl(c, src_dir, dst_dir, rank, nranks, exchange, user, md5_mismatches); });
}

int zwz_ctx_set_chunk_size(zwz_ctx* c, uint32_t bytes) {
    if (!c || bytes > ZWZ_CHUNK_SIZE) return ZWZ_E_INVALID;
    c->chunk_bytes = bytes;
    return ZWZ_OK;
}

}  // extern "C"

namespace {

int compress_dir_impl(zwz_ctx* c, const char* src_dir, const char* dst_dir, const char* file_record, int rank, int nranks) {
    const uint32_t chunk_bytes = chunk_bytes_for(c);
    const auto t_entry = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {          // ZWZ_VERBOSE: the pipeline's own timeline
        if (timeline()) fprintf(stderr, "zwz: [%.3f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count(), what);
    };
    std::vector<std::string> lines;
    {
        std::ifstream f(file_record);
        if (!f.is_open()) { set_error("cannot open file record %s", file_record); return ZWZ_E_IO; }
        std::string s;
        while (std::getline(f, s)) lines.push_back(s);
    }
    int non_empty = 0;
    for (const auto& s : lines) if (!s.empty() && !blank(s)) non_empty++;
    if (rank >= non_empty) return ZWZ_OK;   // main.cpp:47-51: this rank has nothing to do and creates no shard

    // this rank's files (compression.cpp:35-41); a file that cannot be opened is logged and skipped (:45-48)
    struct File { std::string rel, full; bool ok; uint64_t size; uint32_t first_chunk, nchunks; std::string md5; std::atomic<int>* md5_ready;
                  int32_t gpu_md5 = -1; };   // index into its slice's GPU digest list, or -1: hashed on the host
    std::vector<File> files;
    uint64_t total_chunks = 0;
    mark("file list read");
    Pool pool(host_threads());
    // The pinned staging (0.1 s for 2 x 2 048 chunks: hipHostMalloc pins page by page) is allocated by a helper thread while this one sizes the
    // files; its size only needs an upper bound of the chunk count, which the list's length gives for the common case (cap is re-derived below).
    HIPCHK(hipSetDevice(c->device));
    std::thread staging_thread;
    int staging_rc = ZWZ_OK;
    const uint32_t early_cap = std::max(1u, std::min(c->max_batch, kSliceChunks));
    const bool early_staging = lines.size() / (size_t)nranks >= early_cap;        // at least a slice's worth of files: the full-size staging is what will be asked for
    if (early_staging) staging_thread = std::thread([&] { (void)hipSetDevice(c->device); staging_rc = ensure_staging(c, 2 * early_cap); });
    struct JoinStaging { std::thread& t; ~JoinStaging() { if (t.joinable()) t.join(); } } join_staging{staging_thread};
    {   // Sizes come from stat(), on the pool; a file is opened only by the task that reads it and closed right after.
        // (Holding every source file open needed one descriptor per file -- 370 k of them in BASELINE's config 4 -- and
        // grew the descriptor table step by step, each step a synchronize_rcu() in a process the HIP runtime has made
        // multi-threaded: 80 us per open(), 0.7 s for 8 000 files, more than the rest of the job.  tools/exp/open_hip.cpp)
        std::vector<File> cand;
        for (size_t i = (size_t)rank; i < lines.size(); i += (size_t)nranks) {
            File f;
            f.rel = lines[i];
            f.full = (fs::path(src_dir) / f.rel).string();
            f.ok = false; f.size = 0; f.md5_ready = nullptr;
            cand.push_back(std::move(f));
        }
        Pool::Group stat_group;
        const size_t stat_per = std::max<size_t>(32, std::min<size_t>(256, cand.size() / (4 * host_threads()) + 1));
        for (size_t i0 = 0; i0 < cand.size(); i0 += stat_per)
            pool.submit(stat_group, [&cand, i0, stat_per] {
                for (size_t i = i0; i < std::min(cand.size(), i0 + stat_per); i++) {
                    File& f = cand[i];
                    struct stat sb;
                    if (stat(f.full.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && access(f.full.c_str(), R_OK) == 0) { f.ok = true; f.size = (uint64_t)sb.st_size; }
                }
            });
        pool.wait(stat_group);
        for (File& f : cand) {                     // in list order, like the reference's producer
            if (!f.ok) { fprintf(stderr, "Error opening source file: \"%s\"\n", f.full.c_str()); continue; }
            f.nchunks = (uint32_t)(f.size / chunk_bytes) + 1;   // a short (possibly empty) read ends the file (:52-58)
            f.first_chunk = (uint32_t)total_chunks;
            total_chunks += f.nchunks;
            files.push_back(std::move(f));
        }
    }
    if (total_chunks > 0xffffffffull) { set_error("too many chunks"); return ZWZ_E_INVALID; }
    mark("source files sized");
    std::vector<std::atomic<int>> ready(files.size());
    for (size_t i = 0; i < files.size(); i++) { ready[i].store(0); files[i].md5_ready = &ready[i]; }

    const std::string out_path = (fs::path(dst_dir) / ("compressed_" + std::to_string(rank) + ".zwz")).string();
    const int dest = open(out_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (dest < 0) { set_error("cannot create %s", out_path.c_str()); return ZWZ_E_IO; }

    const uint32_t T = (uint32_t)total_chunks;
    const uint32_t cap = std::max(1u, std::min(c->max_batch, std::min(kSliceChunks, (T + 1) / 2 + 1)));
    Slices sl;
    const double t_alloc0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (staging_thread.joinable()) staging_thread.join();
    if (staging_rc) { close(dest); return staging_rc; }
    int rc = make_slices(c, cap, sl);
    if (timeline()) fprintf(stderr, "zwz: staging for 2 x %u chunks ready after another %.3f s (pinned beside the file sizing)\n", cap,
                           std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_alloc0);
    if (rc) { free_slices(sl); close(dest); return rc; }
    const uint32_t nslices = (T + cap - 1) / cap;

    // Everything the pool's tasks reach by reference is declared BEFORE the Drain (destroyed after it has waited), the staging and
    // the shard's descriptor under guards of their own: an exception on this thread (an allocation in a vector) then unwinds
    // through a function whose tasks have finished and whose resources are released.
    struct Rec { uint64_t off; uint32_t slot; int32_t path_len, seq, payload; uint8_t last; const std::string* rel; const char* md5; };
    std::vector<Rec> recs;
    std::vector<char> hexes;                     // digests that came back from the GPU as 16 raw bytes, as text
    std::atomic<int> io_error{0}, write_error{0};
    uint32_t truncated = 0;
    uint64_t out_pos = 0;                        // shard bytes laid out so far
    struct Release { Slices& sl; const int& fd; ~Release() { free_slices(sl); if (fd >= 0) close(fd); } };
    int dest_open = dest;                        // -1 once the normal path has closed it
    Release release{sl, dest_open};
    Pool::Group md5_group, read_group[2], write_group;
    Drain drain{pool, {&md5_group, &read_group[0], &read_group[1], &write_group}};
    // chunk g -> (file, chunk index in file): files are laid out back to back
    auto file_of = [&](uint32_t g) -> uint32_t {
        uint32_t lo = 0, hi = (uint32_t)files.size() - 1;
        while (lo < hi) { uint32_t mid = (lo + hi + 1) / 2; if (files[mid].first_chunk <= g) lo = mid; else hi = mid - 1; }
        return lo;
    };
    auto hash_whole_file = [&](uint32_t fi) {     // md5_of_file(): re-read (verification.cpp:6-30)
        Md5 m;
        std::vector<uint8_t> buf(1 << 20);
        uint64_t off = 0;
        const int fd = open(files[fi].full.c_str(), O_RDONLY);
        if (fd < 0) io_error.store(1);
        for (; fd >= 0;) {
            ssize_t k = pread(fd, buf.data(), buf.size(), (off_t)off);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) break;
            m.update(buf.data(), (size_t)k); off += (uint64_t)k;
        }
        if (fd >= 0) close(fd);
        char hex[33]; m.hex(hex);
        files[fi].md5 = hex; files[fi].md5_ready->store(1);
    };
    // a file that lies whole inside a slice is hashed on the GPU from the chunk slots the codec
    // reads anyway (one lane per file); files cut by a slice boundary or read in pieces are hashed by host workers
    // like before.  ZWZ_HOST_MD5=1 keeps everything on the host.
    const bool gpu_md5 = getenv("ZWZ_HOST_MD5") == nullptr;
    auto start_read = [&](uint32_t s) {
        const int b = (int)(s & 1u);
        const uint32_t g0 = s * cap, g1 = std::min(T, g0 + cap);
        uint32_t g = g0;
        sl.n_md5[b] = 0;
        // A unit = up to 64 chunks of one file (big files are read by several workers); a task = a run of units worth about
        // 64 chunks or 32 files, whichever comes first (370 000 one-chunk files as 370 000 tasks spent more time in the
        // pool's queue than in pread()).
        struct Unit { uint32_t fi, u0, u1; bool hash_here; };
        const uint32_t task_chunks = std::max(16u, std::min(64u, cap / (2u * host_threads())));   // every worker gets a couple of tasks a slice
        std::vector<Unit> batch;
        uint32_t batch_chunks = 0;
        auto flush = [&] {
            if (batch.empty()) return;
            pool.submit(read_group[b], [&, b, g0, units = std::move(batch)] {
              for (const Unit& un : units) {
                const uint32_t fi = un.fi, u0 = un.u0, u1 = un.u1;
                const bool hash_here = un.hash_here;
                {
                    const File& ff = files[fi];
                    Md5 m;
                    const int fd = open(ff.full.c_str(), O_RDONLY);
                    if (fd < 0) io_error.store(1);
                    for (uint32_t ci = u0; ci < u1; ci++) {
                        const uint32_t slot = ff.first_chunk + ci - g0;
                        uint8_t* dstp = sl.h_in[b] + (size_t)slot * ZWZ_DEV_STRIDE;
                        const uint64_t off = (uint64_t)ci * chunk_bytes;
                        const size_t want = off < ff.size ? (size_t)std::min<uint64_t>(chunk_bytes, ff.size - off) : 0;
                        size_t got = 0;
                        while (got < want && fd >= 0) {
                            ssize_t k = pread(fd, dstp + got, want - got, (off_t)(off + got));
                            if (k < 0 && errno == EINTR) continue;
                            if (k <= 0) break;
                            got += (size_t)k;
                        }
                        if (got != want) io_error.store(1);       // the file shrank under us
                        sl.h_off[b][slot] = (uint64_t)slot * ZWZ_DEV_STRIDE;
                        sl.h_len[b][slot] = (uint32_t)got;
                        if (hash_here) m.update(dstp, got);
                    }
                    if (fd >= 0) close(fd);
                    if (hash_here) { char hex[33]; m.hex(hex); files[fi].md5 = hex; files[fi].md5_ready->store(1); }
                }
              }
            });
            batch.clear(); batch_chunks = 0;
        };
        while (g < g1) {
            const uint32_t fi = file_of(g);
            File& f = files[fi];
            const uint32_t c0 = g - f.first_chunk, c1 = std::min(f.nchunks, c0 + (g1 - g));
            const bool whole = c0 == 0 && c1 == f.nchunks;
            for (uint32_t u0 = c0; u0 < c1; u0 += 64) {
                const uint32_t u1 = std::min(c1, u0 + 64);
                const bool one_unit = whole && u0 == 0 && u1 == c1;
                if (one_unit && gpu_md5) {
                    f.gpu_md5 = (int32_t)sl.n_md5[b];
                    sl.h_files[b][2 * sl.n_md5[b]] = f.first_chunk - g0; sl.h_files[b][2 * sl.n_md5[b] + 1] = f.nchunks;
                    sl.n_md5[b]++;
                }
                batch.push_back({fi, u0, u1, one_unit && !gpu_md5});
                batch_chunks += u1 - u0;
                if (batch_chunks >= task_chunks || batch.size() >= 32) flush();
            }
            g += c1 - c0;
        }
        flush();
    };
    // a whole file read as ONE unit is hashed by its reader (or the GPU); every other file -- cut by a slice boundary, or
    // read in pieces -- by hash_whole_file.  Those tasks only need the source file, and MD5 is one sequential stream per
    // file (~0.65 GB/s), so they all go out NOW: submitted when the pipeline reached the file, three 384 MiB files were
    // hashed one after the other with the record writer waiting for each digest in turn.
    for (uint32_t fi = 0; fi < (uint32_t)files.size(); fi++) {
        const File& f = files[fi];
        const bool whole = f.first_chunk / cap == (f.first_chunk + f.nchunks - 1) / cap;
        if (!whole || f.nchunks > 64) pool.submit(md5_group, [&, fi] { hash_whole_file(fi); }, /*background=*/true);
    }
    auto launch_gpu = [&](uint32_t s) -> int {
        const int b = (int)(s & 1u);
        const uint32_t m = std::min(T, (s + 1) * cap) - s * cap;
        // copy in (its own queue): the pair's device buffers were last read by the kernels of slice s - 2
        if (s >= 2) HIPCHK(hipStreamWaitEvent(sl.s_in, sl.ev_k[b], 0));
        HIPCHK(hipMemcpyAsync(sl.d_in[b], sl.h_in[b], (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, sl.s_in));
        HIPCHK(hipMemcpyAsync(sl.d_len[b], sl.h_len[b], m * sizeof(uint32_t), hipMemcpyHostToDevice, sl.s_in));
        HIPCHK(hipMemcpyAsync(sl.d_off[b], sl.h_off[b], m * sizeof(uint64_t), hipMemcpyHostToDevice, sl.s_in));
        if (sl.n_md5[b]) HIPCHK(hipMemcpyAsync(sl.d_files[b], sl.h_files[b], (size_t)sl.n_md5[b] * 8, hipMemcpyHostToDevice, sl.s_in));
        HIPCHK(hipEventRecord(sl.ev_in[b], sl.s_in));
        // the kernels (the codec's stream): behind the copy in, and behind the copy out of slice s - 2, which read d_out[b]
        HIPCHK(hipStreamWaitEvent(c->stream, sl.ev_in[b], 0));
        if (s >= 2) HIPCHK(hipStreamWaitEvent(c->stream, sl.done[b], 0));
        // offsets are relative to this slice's d_in
        int r = zwz_deflate_batch_dev(c, sl.d_in[b], sl.d_off[b], sl.d_len[b], m, sl.d_out[b], ZWZ_DEV_STRIDE, sl.d_olen[b]);
        if (r) return r;
        if (sl.n_md5[b]) {
            r = zwz_md5_files_dev(c, sl.d_in[b], sl.d_off[b], sl.d_len[b], sl.d_files[b], sl.n_md5[b], sl.d_dig[b]);
            if (r) return r;
        }
        HIPCHK(hipEventRecord(sl.ev_k[b], c->stream));
        // copy out (its own queue)
        HIPCHK(hipStreamWaitEvent(sl.s_out, sl.ev_k[b], 0));
        if (sl.n_md5[b]) HIPCHK(hipMemcpyAsync(sl.h_dig[b], sl.d_dig[b], (size_t)sl.n_md5[b] * 16, hipMemcpyDeviceToHost, sl.s_out));
        HIPCHK(hipMemcpyAsync(sl.h_out[b], sl.d_out[b], (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, sl.s_out));
        HIPCHK(hipMemcpyAsync(sl.h_olen[b], sl.d_olen[b], m * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.s_out));
        HIPCHK(hipEventRecord(sl.done[b], sl.s_out));
        return ZWZ_OK;
    };
    // data_writer(), compression.cpp:73-104.  The shard is one file, but every record's place in it is known once the
    // slice's payload lengths are back: the caller's thread lays the slice out (and waits for the digests it needs),
    // then the pool writes it, each worker its own run of records with pwritev straight from the pinned output slots
    // -- a single buffered writer copied every payload twice and was the whole pipeline's pace (1.9 GB/s).
    const uint32_t consumers = [] { const char* v = getenv("ZWZ_CONSUMERS"); const long k = v ? atol(v) : 1; return (uint32_t)(k < 1 ? 1 : k > 64 ? 64 : k); }();
    auto write_records = [&](uint32_t s) {
        const int b = (int)(s & 1u);
        const uint32_t g0 = s * cap, g1 = std::min(T, g0 + cap);
        recs.clear(); recs.reserve(g1 - g0);
        hexes.resize((size_t)cap * 32);          // (sl.n_md5[b] already counts the slice being read into this buffer pair)
        // SURVEY.md section 8 f4, opt-in and never default: ZWZ_CONSUMERS=k writes a slice's records the way k of the reference's
        // consumers (process.hpp:13 NUM_CONSUMERS, shipped as 1) might finish them -- every run of k consecutive chunks of a file
        // back to front -- except a file's last chunk, which stays behind all its others: the reference's reader finalises a file
        // only when its last record arrives in turn (decompression.cpp:132).  Readers put the chunks back by sequence_id.
        std::vector<uint32_t> order(g1 - g0), file_idx(g1 - g0);
        {
            uint32_t fi = file_of(g0);
            for (uint32_t g = g0; g < g1; g++) {
                while (g >= files[fi].first_chunk + files[fi].nchunks) fi++;
                order[g - g0] = g; file_idx[g - g0] = fi;
            }
            if (consumers > 1)
                for (uint32_t i = 0; i < g1 - g0;) {
                    const uint32_t fi2 = file_idx[i];
                    uint32_t j = i;
                    while (j < g1 - g0 && file_idx[j] == fi2 && order[j] + 1u != files[fi2].first_chunk + files[fi2].nchunks && j - i < consumers) j++;
                    if (j == i) { i++; continue; }                          // a file's last chunk: where it is
                    std::reverse(order.begin() + i, order.begin() + j);
                    i = j;
                }
        }
        for (uint32_t e = 0; e < g1 - g0; e++) {
            const uint32_t g = order[e];
            const File& f = files[file_idx[g - g0]];
            Rec r;
            r.slot = g - g0; r.seq = (int32_t)(g - f.first_chunk); r.last = r.seq + 1 == (int32_t)f.nchunks;
            r.path_len = (int32_t)f.rel.size(); r.payload = (int32_t)sl.h_olen[b][r.slot]; r.rel = &f.rel; r.md5 = nullptr;
            r.off = out_pos;
            out_pos += 4u + 4u + (uint64_t)r.path_len + 4u + 1u + (uint64_t)r.payload;
            if (r.last) {
                if (f.gpu_md5 >= 0) {
                    static const char* dig = "0123456789abcdef";
                    const uint8_t* d = sl.h_dig[b] + (size_t)f.gpu_md5 * 16;
                    char* hex = hexes.data() + (size_t)f.gpu_md5 * 32;
                    for (int i = 0; i < 16; i++) { hex[2 * i] = dig[d[i] >> 4]; hex[2 * i + 1] = dig[d[i] & 15]; }
                    r.md5 = hex;
                } else {
                    while (!f.md5_ready->load()) std::this_thread::yield();
                    r.md5 = f.md5.data();          // 32 hex characters (verification.cpp:24-27)
                }
                out_pos += 32u;
            }
            recs.push_back(r);
        }
        // ONE writer, this thread.  The shard is one file and the kernel serialises buffered writes to an inode: sixteen workers with
        // pwritev on their own runs of records (rounds 1-4) wrote it at 4 - 8 GB/s, each waiting for the others' lock; one thread writing
        // the records in order reaches 8.5 (tmpfs) to 13 GB/s (the box's overlay file system) -- tools/exp/shard_write.cpp.  The GPU works on
        // the next slice and the pool reads the one after it meanwhile.
        {
            const size_t n = recs.size();
            std::vector<int32_t> hdr;  hdr.reserve(4 * n);                   // total, path_len | seq per record
            std::vector<struct iovec> iov; iov.reserve(6 * n);
            for (size_t i = 0; i < n; i++) {
                const Rec& r = recs[i];
                hdr.push_back(4 + r.path_len + 4 + 1 + r.payload); hdr.push_back(r.path_len); hdr.push_back(r.seq); hdr.push_back(0);
            }
            for (size_t i = 0; i < n; i++) {
                const Rec& r = recs[i];
                int32_t* h = hdr.data() + 4 * i;
                iov.push_back({h, 8});
                iov.push_back({const_cast<char*>(r.rel->data()), (size_t)r.path_len});
                iov.push_back({h + 2, 4});
                iov.push_back({const_cast<uint8_t*>(&r.last), 1});
                iov.push_back({sl.h_out[b] + (size_t)r.slot * ZWZ_DEV_STRIDE, (size_t)r.payload});
                if (r.last) iov.push_back({const_cast<char*>(r.md5), 32});
            }
            uint64_t off = n ? recs[0].off : 0;
            size_t k = 0;
            while (k < iov.size()) {                                         // pwritev: at most IOV_MAX pieces a call, and it may stop short
                if (iov[k].iov_len == 0) { k++; continue; }
                const int cnt = (int)std::min<size_t>(1024, iov.size() - k);
                ssize_t w = pwritev(dest, iov.data() + k, cnt, (off_t)off);
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) { write_error.store(1); break; }
                off += (uint64_t)w;
                size_t left = (size_t)w;
                while (left) {
                    if (left >= iov[k].iov_len) { left -= iov[k].iov_len; k++; }
                    else { iov[k].iov_base = static_cast<char*>(iov[k].iov_base) + left; iov[k].iov_len -= left; left = 0; }
                }
            }
        }
        pool.wait(write_group);                 // the slice's buffers go back to the GPU next
    };

    auto count_truncated = [&](uint32_t s) {      // payloads that reached the 65535-byte cap: lossy, like the reference
        const int b = (int)(s & 1u);
        const uint32_t m = std::min(T, (s + 1) * cap) - s * cap;
        for (uint32_t i = 0; i < m; i++) if (sl.h_olen[b][i] == ZWZ_CHUNK_SIZE && sl.h_len[b][i] >= 65510u) truncated++;
    };
    // (ZWZ_VERBOSE: where the caller's thread spent its time)
    double t_read = 0, t_gpu = 0, t_write = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    mark("first slice's reads queued");
    if (nslices) start_read(0);
    for (uint32_t s = 0; s < nslices && rc == ZWZ_OK; s++) {
        double t0 = now();
        pool.wait(read_group[s & 1]);
        t_read += now() - t0;
        rc = launch_gpu(s);
        if (rc) break;
        t0 = now();
        if (s >= 1) { hipError_t e = hipEventSynchronize(sl.done[(s - 1) & 1]); if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; } }
        t_gpu += now() - t0;
        if (s >= 1) count_truncated(s - 1);           // (reads h_len, which the next slice's readers are about to overwrite)
        if (s + 1 < nslices) start_read(s + 1);       // its buffers were last used by slice s-1, now complete
        t0 = now();
        if (s >= 1) write_records(s - 1);
        t_write += now() - t0;
    }
    if (rc == ZWZ_OK && nslices) {
        double t0 = now();
        hipError_t e = hipEventSynchronize(sl.done[(nslices - 1) & 1]);
        t_gpu += now() - t0; t0 = now();
        if (e != hipSuccess) rc = hip_fail(e, "hipEventSynchronize"); else { count_truncated(nslices - 1); write_records(nslices - 1); }
        t_write += now() - t0;
    }
    if (timeline())
        fprintf(stderr, "zwz: %u slices of <= %u chunks in %.3f s: waiting for reads %.3f s, for the GPU %.3f s, writing %.3f s\n",
                nslices, cap, now() - t_begin, t_read, t_gpu, t_write);
    mark("last slice written");
    pool.wait(read_group[0]); pool.wait(read_group[1]); pool.wait(md5_group);
    (void)hipStreamSynchronize(c->stream);
    mark("workers and stream idle");
    const double t_free0 = now();
    free_slices(sl);
    if (timeline()) fprintf(stderr, "zwz: staging released in %.3f s\n", now() - t_free0);
    if (write_error.load() && rc == ZWZ_OK) { set_error("writing %s failed", out_path.c_str()); rc = ZWZ_E_IO; }
    dest_open = -1;
    if (close(dest) != 0 && rc == ZWZ_OK) rc = ZWZ_E_IO;
    mark("shard closed");
    if (io_error.load() && rc == ZWZ_OK) { set_error("a source file changed or vanished while it was being read"); rc = ZWZ_E_IO; }
    { std::string why; if (pool.failed(&why) && rc == ZWZ_OK) { set_error("a pipeline task failed: %s", why.c_str()); rc = ZWZ_E_IO; } }
    if (truncated && timeline())
        fprintf(stderr, "zwz: %u chunk payload(s) reached the reference's 65535-byte cap (lossy, like the reference)\n", truncated);
    return rc;
}

// ---- .zwz reader (decompression.cpp:45-163) ---------------------------------------------------------------------------
struct Mapped {            // the shard is mapped, not read: records are parsed in place and payloads copied from the mapping
    const uint8_t* p = nullptr; size_t n = 0;   // straight into pinned staging slots
    ~Mapped() { if (p) munmap(const_cast<uint8_t*>(p), n); }
    size_t size() const { return n; }
    const uint8_t& operator[](size_t i) const { return p[i]; }
};
struct Rec { uint64_t off; uint32_t len; int32_t seq; uint8_t last; uint32_t avail; };   // avail < len: the shard ends inside this payload
struct FileInst {          // one output file instance: the records that get decoded into it, in decode order
    std::string rel; std::vector<Rec> order; std::multimap<int32_t, Rec> pending; int32_t expected = 0; std::string md5; bool finalised = false;
};

// Pass 1: parse the records (decompression.cpp:65-92) and replay the reference's per-path sequencing (expected id +
// pending heap, :119-153).  A shard that ends inside a record is damaged: everything in front of the damage is kept and
// decoded as the reference's reader would have (it processes record after record until its stream fails): a record cut
// inside its payload is decoded from the bytes that are there followed by zeros up to its declared length (the
// reference reads into a zero-filled vector of that length, decompression.cpp:82-84), one cut inside its MD5 keeps the
// partial digest (and so fails verification).  Returns false if the shard was damaged.
bool parse_shard(const Mapped& blob, std::vector<FileInst>& insts) {
    std::map<std::string, size_t> open_inst;
    bool intact = true;
    size_t p = 0;
    while (p < blob.size()) {
        int32_t total, path_len, seq;
        if (p + 8 > blob.size()) { intact = false; break; }
        memcpy(&total, &blob[p], 4); memcpy(&path_len, &blob[p + 4], 4); p += 8;
        if (path_len < 0 || p + (size_t)path_len + 5 > blob.size()) { intact = false; break; }
        std::string rel(reinterpret_cast<const char*>(&blob[p]), (size_t)path_len); p += (size_t)path_len;
        memcpy(&seq, &blob[p], 4); p += 4;
        const uint8_t last = blob[p++];
        int64_t plen = (int64_t)total - (4 + path_len + 4 + 1);
        if (plen < 0 || plen > (int64_t)ZWZ_CHUNK_SIZE) { intact = false; break; }     // (the reference's CompressedChunk holds 65535 bytes)
        uint32_t avail = (uint32_t)plen;
        if (p + (size_t)plen > blob.size()) { intact = false; avail = (uint32_t)(blob.size() - p); }
        Rec r{(uint64_t)p, (uint32_t)plen, seq, last, avail};
        p += (size_t)avail;
        std::string md5;
        if (last) {
            const size_t have = std::min<size_t>(ZWZ_MD5_HEX_LEN, blob.size() - p);
            md5.assign(reinterpret_cast<const char*>(&blob[p]), have);
            md5.resize(ZWZ_MD5_HEX_LEN, '\0');
            if (have < ZWZ_MD5_HEX_LEN) intact = false;
            p += have;
        }
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
        if (!intact) break;
    }
    return intact;
}

int map_shard(const std::string& shard, Mapped& blob, bool& opened) {
    opened = false;
    const int fd = open(shard.c_str(), O_RDONLY);
    if (fd < 0) { fprintf(stderr, "Error opening file: %s\n", shard.c_str()); return ZWZ_OK; }   // decompression.cpp:47-50
    opened = true;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); set_error("cannot stat %s", shard.c_str()); return ZWZ_E_IO; }
    if (sb.st_size > 0) {
        void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); set_error("cannot map %s", shard.c_str()); return ZWZ_E_IO; }
        blob.p = static_cast<const uint8_t*>(m); blob.n = (size_t)sb.st_size;
        (void)madvise(m, blob.n, MADV_WILLNEED);
    }
    close(fd);
    return ZWZ_OK;
}

struct Job { uint32_t inst; Rec r; };

// What both decode modes share: output files, MD5 verification, the reference's messages.
struct DecodeSink {
    const char* dst_dir; std::vector<FileInst>& insts; std::atomic<int>& mismatches; std::mutex& log_mutex;
    std::string path_of(uint32_t inst) const { return std::string(dst_dir) + "/" + insts[inst].rel; }
    void make_parent(const std::string& file_path) const {
        std::error_code ec;
        fs::path dir = fs::path(file_path).parent_path();
        if (!dir.empty() && !fs::exists(dir, ec)) fs::create_directories(dir, ec);
    }
    void verdict(uint32_t inst, const char* hex) const {         // decompression.cpp:132-149
        const FileInst& fi = insts[inst];
        const std::string file_path = path_of(inst);
        std::lock_guard<std::mutex> l(log_mutex);
        if (fi.finalised) {
            if (fi.md5 != hex) {
                mismatches.fetch_add(1);
                fprintf(stderr, "MD5 mismatch for file: %s\n", file_path.c_str());
                if (verbose()) printf("Expected MD5: %s\nCalculated MD5: %s\n", fi.md5.c_str(), hex);
            } else if (verbose()) printf("MD5 match for file: %s\n", file_path.c_str());
        } else if (!fi.pending.empty()) {
            fprintf(stderr, "Warning: pending chunks remaining for file: %s\n", fi.rel.c_str());
        }
    }
    void hash_from_disk(uint32_t inst) const {                   // md5_of_file(), verification.cpp:6-30
        Md5 m;
        const std::string file_path = path_of(inst);
        const int fd = open(file_path.c_str(), O_RDONLY);
        std::vector<uint8_t> buf(1 << 20);
        for (uint64_t off = 0; fd >= 0;) {
            ssize_t k = pread(fd, buf.data(), buf.size(), (off_t)off);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) break;
            m.update(buf.data(), (size_t)k); off += (uint64_t)k;
        }
        if (fd >= 0) close(fd);
        char hex[33]; m.hex(hex);
        verdict(inst, hex);
    }
    // The same digest, taken WHILE the file is being written: the hash trails the writers -- it reads what `durable` says is on disk,
    // waits when it has caught up, and ends when the file is `complete` and read to its end.  One MD5 stream is ~0.8 GB/s (an 8 GiB
    // file: 10.9 s, tools/one_file_e2e.py), the decode of the same file 1.2 s: started behind the last record, the hash ADDED its time;
    // started with the first slice it hides the decode.
    void hash_trailing(uint32_t inst, const std::atomic<uint64_t>& durable, const std::atomic<int>& complete) const {
        Md5 m;
        const std::string file_path = path_of(inst);
        const int fd = open(file_path.c_str(), O_RDONLY);
        std::vector<uint8_t> buf(4 << 20);
        uint64_t off = 0;
        while (fd >= 0) {
            const int fin = complete.load(std::memory_order_acquire);        // (read BEFORE durable: a complete file's durable is final)
            const uint64_t upto = durable.load(std::memory_order_acquire);
            if (off < upto) {
                const ssize_t k = pread(fd, buf.data(), (size_t)std::min<uint64_t>(buf.size(), upto - off), (off_t)off);
                if (k < 0 && errno == EINTR) continue;
                if (k <= 0) break;                                            // (a write that failed: the file is short and fails its check, as before)
                m.update(buf.data(), (size_t)k); off += (uint64_t)k;
            } else if (fin) break;
            else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (fd >= 0) close(fd);
        char hex[33]; m.hex(hex);
        verdict(inst, hex);
    }
};

// pwritev of a run of decoded chunks at a known file offset
void pwrite_chunks(int fd, uint64_t off, std::vector<struct iovec>& iov) {
    size_t k = 0;
    while (k < iov.size()) {
        if (iov[k].iov_len == 0) { k++; continue; }
        ssize_t w = pwritev(fd, iov.data() + k, (int)std::min<size_t>(512, iov.size() - k), (off_t)off);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) break;                            // (a short file then fails its MD5 check)
        off += (uint64_t)w;
        size_t left = (size_t)w;
        while (left) {
            if (left >= iov[k].iov_len) { left -= iov[k].iov_len; k++; }
            else { iov[k].iov_base = static_cast<char*>(iov[k].iov_base) + left; iov[k].iov_len -= left; left = 0; }
        }
    }
}

// ---- one whole shard on this rank: all scheduled records, slice by slice, double-buffered; files are written and hashed by
// the pool, one task per file per slice (slices of a file run in order)
int decode_whole_shard(zwz_ctx* c, Pool& pool, const Mapped& blob, std::vector<FileInst>& insts, DecodeSink& sink) {
    std::vector<Job> jobs;
    for (uint32_t i = 0; i < insts.size(); i++) for (const Rec& r : insts[i].order) jobs.push_back({i, r});
    struct OutState { FILE* f = nullptr; int fd = -1; uint64_t written = 0; Md5 md5; uint32_t remaining = 0; bool failed = false; bool deferred = false;
                      int32_t gpu_md5 = -1;         // index into its slice's GPU digest list, or -1: hashed by the task that writes it
                      std::atomic<uint64_t> durable{0}; std::atomic<int> complete{0}; bool hashing = false; };   // a spanning file's trailing hash (DecodeSink::hash_trailing)
    std::vector<OutState> outs(insts.size());
    // SURVEY.md section 8 f1, second call site (decompression.cpp:136): a file decoded whole within one slice (and no longer than the
    // 64 chunks a lane is asked to hash on the compress side) sits in consecutive output slots on the device -- its MD5 is taken
    // there by md5_files_kernel, one lane per file, before the bytes come back; the writing task then only writes.  Files that
    // span slices are hashed from disk as before; ZWZ_HOST_MD5=1 keeps every digest on the host.
    const bool gpu_md5 = getenv("ZWZ_HOST_MD5") == nullptr;
    // A path may occur as two instances in one shard (finalised, then seen again: duplicate lines in the file list).  The reference
    // handles records strictly in shard order, so the second instance re-opens (truncates) the file after the first is complete;
    // here instances are written by concurrent tasks, so instances that share a path are taken out of the concurrency: each is
    // written on this thread, behind everything submitted before it.
    std::vector<bool> shared_path(insts.size(), false);
    {
        std::unordered_map<std::string, uint32_t> first;
        for (uint32_t i = 0; i < insts.size(); i++) {
            auto it = first.find(insts[i].rel);
            if (it == first.end()) first.emplace(insts[i].rel, i);
            else { shared_path[i] = true; shared_path[it->second] = true; }
        }
    }
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_gpu_wait = 0, t_write_wait = 0;
    for (uint32_t i = 0; i < insts.size(); i++) outs[i].remaining = (uint32_t)insts[i].order.size();
    const uint32_t T = (uint32_t)jobs.size();
    const uint32_t cap = std::max(1u, std::min(c->max_batch, std::min(kSliceChunks, (T + 1) / 2 + 1)));
    Slices sl;
    int rc = make_slices(c, cap, sl);
    if (rc) { free_slices(sl); return rc; }
    const uint32_t nslices = (T + cap - 1) / cap;
    struct FreeSlices { Slices& s; ~FreeSlices() { free_slices(s); } } free_sl{sl};      // behind the Drain: released once the tasks are done, whatever the way out
    Pool::Group fill_group[2], write_group, md5_group;
    Drain drain{pool, {&fill_group[0], &fill_group[1], &write_group, &md5_group}};
    // MD5 is one sequential stream per file (~0.65 GB/s).  A file decoded within one slice is hashed by the task that
    // writes it, from the staging buffer; a file that spans slices would hold every slice back for its hash (three
    // 384 MiB files: 2.1 s, slower than the reference), so it is hashed afterwards, from the file just written, by a
    // task of its own that runs beside the slices of the files behind it.
    {
        uint32_t g = 0;
        for (uint32_t i = 0; i < insts.size(); i++) {
            const uint32_t k = (uint32_t)insts[i].order.size();
            if (k) outs[i].deferred = g / cap != (g + k - 1) / cap;
            g += k;
        }
    }
    auto open_out = [&](uint32_t inst) {
        const std::string file_path = sink.path_of(inst);
        sink.make_parent(file_path);
        outs[inst].f = fopen(file_path.c_str(), "wb");
        if (!outs[inst].f) { std::lock_guard<std::mutex> l(sink.log_mutex); fprintf(stderr, "Error creating output file: %s\n", file_path.c_str()); outs[inst].failed = true; }
    };
    auto finish_out = [&](uint32_t inst, const uint8_t* digest /* 16 bytes from the GPU, or null */) {
        OutState& o = outs[inst];
        if (o.f) { fclose(o.f); o.f = nullptr; }
        char hex[33];
        if (digest) { static const char* dig = "0123456789abcdef"; for (int i = 0; i < 16; i++) { hex[2 * i] = dig[digest[i] >> 4]; hex[2 * i + 1] = dig[digest[i] & 15]; } hex[32] = 0; }
        else o.md5.hex(hex);
        sink.verdict(inst, hex);
    };
    auto start_fill = [&](uint32_t s) {
        const int b = (int)(s & 1u);
        const uint32_t g0 = s * cap, g1 = std::min(T, g0 + cap);
        sl.n_md5[b] = 0;
        if (gpu_md5)
            for (uint32_t g = g0; g < g1;) {
                uint32_t e = g;
                while (e < g1 && jobs[e].inst == jobs[g].inst) e++;
                OutState& o = outs[jobs[g].inst];
                if (!o.deferred && !shared_path[jobs[g].inst] && e - g <= 64u && e - g == (uint32_t)insts[jobs[g].inst].order.size()) {
                    o.gpu_md5 = (int32_t)sl.n_md5[b];
                    sl.h_files[b][2 * sl.n_md5[b]] = g - g0; sl.h_files[b][2 * sl.n_md5[b] + 1] = e - g;
                    sl.n_md5[b]++;
                }
                g = e;
            }
        const uint32_t per_task = std::max(32u, std::min(256u, cap / (2u * host_threads())));
        for (uint32_t u0 = g0; u0 < g1; u0 += per_task) {
            const uint32_t u1 = std::min(g1, u0 + per_task);
            pool.submit(fill_group[b], [&, b, g0, u0, u1] {
                for (uint32_t g = u0; g < u1; g++) {
                    const Rec& r = jobs[g].r;
                    uint8_t* slot = sl.h_in[b] + (size_t)(g - g0) * ZWZ_DEV_STRIDE;
                    memcpy(slot, &blob[r.off], r.avail);
                    if (r.avail < r.len) memset(slot + r.avail, 0, r.len - r.avail);
                    sl.h_off[b][g - g0] = (uint64_t)(g - g0) * ZWZ_DEV_STRIDE; sl.h_len[b][g - g0] = r.len;
                }
            });
        }
    };
    auto launch_gpu = [&](uint32_t s) -> int {
        const int b = (int)(s & 1u);
        const uint32_t m = std::min(T, (s + 1) * cap) - s * cap;
        // three queues, as on the compress side (see Slices)
        if (s >= 2) HIPCHK(hipStreamWaitEvent(sl.s_in, sl.ev_k[b], 0));
        HIPCHK(hipMemcpyAsync(sl.d_in[b], sl.h_in[b], (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, sl.s_in));
        HIPCHK(hipMemcpyAsync(sl.d_len[b], sl.h_len[b], m * sizeof(uint32_t), hipMemcpyHostToDevice, sl.s_in));
        HIPCHK(hipMemcpyAsync(sl.d_off[b], sl.h_off[b], m * sizeof(uint64_t), hipMemcpyHostToDevice, sl.s_in));
        if (sl.n_md5[b]) HIPCHK(hipMemcpyAsync(sl.d_files[b], sl.h_files[b], (size_t)sl.n_md5[b] * 8, hipMemcpyHostToDevice, sl.s_in));
        HIPCHK(hipEventRecord(sl.ev_in[b], sl.s_in));
        HIPCHK(hipStreamWaitEvent(c->stream, sl.ev_in[b], 0));
        if (s >= 2) HIPCHK(hipStreamWaitEvent(c->stream, sl.done[b], 0));
        int r = zwz_inflate_batch_dev(c, sl.d_in[b], sl.d_off[b], sl.d_len[b], m, sl.d_out[b], ZWZ_DEV_STRIDE, sl.d_olen[b], sl.d_st[b]);
        if (r) return r;
        if (sl.n_md5[b]) {      // (output slot k sits at k * 65536, which is what d_off holds for the input slots)
            r = zwz_md5_files_dev(c, sl.d_out[b], sl.d_off[b], sl.d_olen[b], sl.d_files[b], sl.n_md5[b], sl.d_dig[b]);
            if (r) return r;
        }
        HIPCHK(hipEventRecord(sl.ev_k[b], c->stream));
        HIPCHK(hipStreamWaitEvent(sl.s_out, sl.ev_k[b], 0));
        if (sl.n_md5[b]) HIPCHK(hipMemcpyAsync(sl.h_dig[b], sl.d_dig[b], (size_t)sl.n_md5[b] * 16, hipMemcpyDeviceToHost, sl.s_out));
        HIPCHK(hipMemcpyAsync(sl.h_out[b], sl.d_out[b], (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, sl.s_out));
        HIPCHK(hipMemcpyAsync(sl.h_olen[b], sl.d_olen[b], m * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.s_out));
        HIPCHK(hipEventRecord(sl.done[b], sl.s_out));
        return ZWZ_OK;
    };
    // One task per file that has records in this slice -- a file decoded within one slice is written (stdio) and hashed
    // by it.  A file that spans slices is written with pwritev at offsets laid out here, 128 records a task, and
    // hashed afterwards from the file (see `deferred` above).
    auto write_slice = [&](uint32_t s) {
        const int b = (int)(s & 1u);
        const uint32_t g0 = s * cap, g1 = std::min(T, g0 + cap);
        std::vector<uint32_t> finished;          // spanning files whose last record is in this slice
        uint32_t g = g0;
        while (g < g1) {
            uint32_t e = g;
            while (e < g1 && jobs[e].inst == jobs[g].inst) e++;
            const uint32_t inst = jobs[g].inst, a0 = g, a1 = e;
            OutState& os = outs[inst];
            if (shared_path[inst]) {            // rare: in shard order, nothing else of this path in flight (see shared_path)
                pool.wait(write_group); pool.wait(md5_group);
                if (!os.f && !os.failed && os.remaining == (uint32_t)insts[inst].order.size()) open_out(inst);
                for (uint32_t k = a0; k < a1; k++) {
                    const uint8_t* src = sl.h_out[b] + (size_t)(k - g0) * ZWZ_DEV_STRIDE;
                    const uint32_t n = sl.h_olen[b][k - g0];
                    if (os.f) fwrite(src, 1, n, os.f);
                    os.md5.update(src, n);
                }
                os.remaining -= a1 - a0;
                if (os.remaining == 0) finish_out(inst, nullptr);
            } else if (!os.deferred) {
                pool.submit(write_group, [&, b, g0, inst, a0, a1] {
                    OutState& o = outs[inst];
                    if (!o.f && !o.failed) open_out(inst);
                    for (uint32_t k = a0; k < a1; k++) {
                        const uint8_t* src = sl.h_out[b] + (size_t)(k - g0) * ZWZ_DEV_STRIDE;
                        const uint32_t n = sl.h_olen[b][k - g0];
                        if (o.f) fwrite(src, 1, n, o.f);
                        if (o.gpu_md5 < 0) o.md5.update(src, n);
                    }
                    o.remaining -= a1 - a0;
                    if (o.remaining == 0) finish_out(inst, o.gpu_md5 >= 0 ? sl.h_dig[b] + (size_t)o.gpu_md5 * 16 : nullptr);
                });
            } else {
                if (os.fd < 0 && !os.failed) {
                    const std::string file_path = sink.path_of(inst);
                    sink.make_parent(file_path);
                    os.fd = open(file_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
                    if (os.fd < 0) { std::lock_guard<std::mutex> l(sink.log_mutex); fprintf(stderr, "Error creating output file: %s\n", file_path.c_str()); os.failed = true; }
                }
                for (uint32_t r0 = a0; r0 < a1 && os.fd >= 0; r0 += 128) {
                    const uint32_t r1 = std::min(a1, r0 + 128);
                    const uint64_t off0 = os.written;
                    uint64_t bytes = 0;
                    for (uint32_t k = r0; k < r1; k++) bytes += sl.h_olen[b][k - g0];
                    os.written += bytes;
                    const int fd = os.fd;
                    pool.submit(write_group, [&, b, g0, fd, r0, r1, off0] {
                        std::vector<struct iovec> iov;
                        for (uint32_t k = r0; k < r1; k++)
                            if (sl.h_olen[b][k - g0]) iov.push_back({sl.h_out[b] + (size_t)(k - g0) * ZWZ_DEV_STRIDE, (size_t)sl.h_olen[b][k - g0]});
                        pwrite_chunks(fd, off0, iov);
                    });
                }
                os.remaining -= a1 - a0;
                if (os.remaining == 0) finished.push_back(inst);
            }
            g = e;
        }
        { const double t0 = since(); pool.wait(write_group); t_write_wait += since() - t0; }   // the next slice of a file must follow this one
        // what this slice wrote of the spanning files is on disk now: their hashes (started with a file's first slice) may read on
        for (uint32_t g2 = g0; g2 < g1;) {
            uint32_t e2 = g2;
            while (e2 < g1 && jobs[e2].inst == jobs[g2].inst) e2++;
            OutState& o = outs[jobs[g2].inst];
            if (o.deferred && !shared_path[jobs[g2].inst] && o.fd >= 0) {
                o.durable.store(o.written, std::memory_order_release);
                if (!o.hashing) { o.hashing = true; const uint32_t inst = jobs[g2].inst; pool.submit(md5_group, [&sink, &outs, inst] { sink.hash_trailing(inst, outs[inst].durable, outs[inst].complete); }, /*background=*/true); }
            }
            g2 = e2;
        }
        for (uint32_t inst : finished) {
            OutState& o = outs[inst];
            if (o.fd >= 0) { close(o.fd); o.fd = -1; }
            if (o.hashing) o.complete.store(1, std::memory_order_release);
            else pool.submit(md5_group, [&sink, inst] { sink.hash_from_disk(inst); }, /*background=*/true);     // (a file that could not be created: reports as before)
        }
    };

    if (nslices && rc == ZWZ_OK) start_fill(0);
    for (uint32_t s = 0; s < nslices && rc == ZWZ_OK; s++) {
        pool.wait(fill_group[s & 1]);
        rc = launch_gpu(s);
        if (rc) break;
        if (s >= 1) { const double t0 = since(); hipError_t e = hipEventSynchronize(sl.done[(s - 1) & 1]); t_gpu_wait += since() - t0; if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; } }
        if (s + 1 < nslices) start_fill(s + 1);
        if (s >= 1) write_slice(s - 1);
    }
    if (rc == ZWZ_OK && nslices) {
        const double t0 = since();
        hipError_t e = hipEventSynchronize(sl.done[(nslices - 1) & 1]);
        t_gpu_wait += since() - t0;
        if (e != hipSuccess) rc = hip_fail(e, "hipEventSynchronize"); else write_slice(nslices - 1);
    }
    const double t_slices = since();
    pool.wait(fill_group[0]); pool.wait(fill_group[1]); pool.wait(md5_group);
    if (timeline()) fprintf(stderr, "zwz: decode: %u records of %zu files in %u slices done at %.3f s (waited %.3f s for the GPU, %.3f s for the writers); "
                                   "spanning files hashed by %.3f s\n", T, insts.size(), nslices, t_slices, t_gpu_wait, t_write_wait, since());
    (void)hipStreamSynchronize(c->stream);
    free_slices(sl);
    // instances that never received a decodable record still get created (the reference opens on the first record of a path)
    for (uint32_t i = 0; i < insts.size(); i++)
        if (insts[i].order.empty()) { open_out(i); finish_out(i, nullptr); }
    return rc;
}

// ---- one shard split over all ranks (the reference decodes a shard serially on one thread,
// decompression.cpp:65-154).  Every rank parses the headers (cheap) and so holds the same schedule; rank r takes the r-th
// contiguous range of it, inflates the range into device memory, and only then learns where its bytes go: a chunk's place in
// its file is the sum of the DECODED lengths in front of it (truncated chunks decode short), so the ranks all-gather, per
// rank, the bytes it decoded for the first and for the last file of its range -- the only files it can share with a
// neighbour.  Then every rank writes its range at its offsets; a second exchange is the barrier after which the rank
// holding a shared file's last record verifies its MD5.  <dst> must be one file system for all ranks (as for the
// reference, whose ranks share everything).
// rc_in: what went wrong on this rank before it got here (the shard could not be mapped): it then decodes nothing but still
// takes part in both exchanges with its failure flag -- its peers are waiting there (a collective has no time-out).
// Device memory: a rank's range need not fit (BASELINE configs[4]: 64 GiB over few ranks).  Phase 1 inflates the range slice by
// slice and keeps what comes back whole only for the first `res` chunks -- as many slices as the budget allows (free device
// memory less a margin, or ZWZ_MAX_RANGE_CHUNKS) -- and for the rest keeps the decoded LENGTHS alone (4 bytes a chunk), which is
// all the exchange needs; phase 2 copies the resident slices out and inflates the others a second time into the slice's own
// output buffer (the reference streams by construction, decompression.cpp:65-154; inflating twice costs less than it reads).
// No way out of this function skips an exchange: everything between them runs under a catch-all that turns an exception
// (std::bad_alloc from a vector or a pool task's closure, a std::filesystem error) into this rank's failure flag.
int decode_shard_split(zwz_ctx* c, Pool& pool, const Mapped& blob, std::vector<FileInst>& insts, DecodeSink& sink, int rank, int nranks,
                       zwz_allgather_u64_fn exchange, void* user, int rc_in) {
    constexpr uint32_t kFields = 5;
    const uint64_t kNoFile = ~0ull;
    // (the exchanges' buffers first: once they exist, nothing below can keep this rank from joining its peers)
    std::vector<uint64_t> all((size_t)nranks * kFields), flags((size_t)nranks);
    int rc = rc_in;
    auto shielded = [&](const char* what, auto&& body) {
        try { body(); }
        catch (const std::bad_alloc&) { if (rc == ZWZ_OK) { set_error("out of host memory while %s", what); rc = ZWZ_E_NOMEM; } }
        catch (const std::exception& e) { if (rc == ZWZ_OK) { set_error("%s: %s", what, e.what()); rc = ZWZ_E_IO; } }
        catch (...) { if (rc == ZWZ_OK) { set_error("unknown exception while %s", what); rc = ZWZ_E_IO; } }
    };
    std::vector<Job> jobs;
    shielded("listing the shard's records", [&] { if (rc_in == ZWZ_OK) for (uint32_t i = 0; i < insts.size(); i++) for (const Rec& r : insts[i].order) jobs.push_back({i, r}); });
    if (rc != ZWZ_OK) jobs.clear();
    const uint64_t T = jobs.size();
    const uint32_t j0 = (uint32_t)(T * (uint64_t)rank / (uint64_t)nranks), j1 = (uint32_t)(T * (uint64_t)(rank + 1) / (uint64_t)nranks);
    const uint32_t n = j1 - j0;
    uint8_t* d_big = nullptr; uint32_t* d_lens = nullptr; uint32_t* d_stat = nullptr;
    std::vector<uint32_t> lens, stats;      // decoded length and end-of-stream status of every chunk of the range (phase 1)
    Slices sl;
    bool have_slices = false;
    const uint32_t cap = std::max(1u, std::min(c->max_batch, std::min(kSliceChunks, n ? (n + 1) / 2 + 1 : 1u)));
    const uint32_t nslices = (n + cap - 1) / cap;
    uint32_t res = 0;                            // chunks [0, res) of the range stay in d_big between the phases (whole slices)
    auto cleanup = [&] {                         // (idempotent: also run by the guard below on any way out)
        (void)hipStreamSynchronize(c->stream);
        if (have_slices) free_slices(sl);
        if (d_big) { (void)hipFree(d_big); d_big = nullptr; }
        if (d_lens) { (void)hipFree(d_lens); d_lens = nullptr; }
        if (d_stat) { (void)hipFree(d_stat); d_stat = nullptr; }
    };
    struct Cleanup { decltype(cleanup)& f; ~Cleanup() { f(); } } cleanup_guard{cleanup};
    // (declared before the Drain -- destroyed after it has waited: the write tasks reach fds and cursor by reference)
    std::vector<int> fds;
    struct CloseFds { std::vector<int>& v; ~CloseFds() { for (int& fd : v) if (fd >= 0) { close(fd); fd = -1; } } } close_fds{fds};
    std::vector<uint64_t> cursor;
    Pool::Group fill_group[2], write_group;
    Drain drain{pool, {&fill_group[0], &fill_group[1], &write_group}};
    auto start_fill = [&](uint32_t s) {
        const int b = (int)(s & 1u);
        const uint32_t g0 = s * cap, g1 = std::min(n, g0 + cap);
        const uint32_t per_task = std::max(32u, std::min(256u, cap / (2u * host_threads())));
        for (uint32_t u0 = g0; u0 < g1; u0 += per_task) {
            const uint32_t u1 = std::min(g1, u0 + per_task);
            pool.submit(fill_group[b], [&, b, g0, u0, u1] {
                for (uint32_t g = u0; g < u1; g++) {
                    const Rec& r = jobs[j0 + g].r;
                    uint8_t* slot = sl.h_in[b] + (size_t)(g - g0) * ZWZ_DEV_STRIDE;
                    memcpy(slot, &blob[r.off], r.avail);
                    if (r.avail < r.len) memset(slot + r.avail, 0, r.len - r.avail);
                    sl.h_off[b][g - g0] = (uint64_t)(g - g0) * ZWZ_DEV_STRIDE; sl.h_len[b][g - g0] = r.len;
                }
            });
        }
    };
    // slice s: payloads to the device, inflated into `target` (cap slots), decoded lengths to d_len_out
    auto inflate_slice = [&](uint32_t s, uint8_t* target, uint32_t* d_len_out, uint32_t* d_stat_out) -> int {
        const int b = (int)(s & 1u);
        const uint32_t g0 = s * cap, m = std::min(n, g0 + cap) - g0;
        hipError_t e = hipMemcpyAsync(sl.d_in[b], sl.h_in[b], (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sl.d_len[b], sl.h_len[b], m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sl.d_off[b], sl.h_off[b], m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync");
        return zwz_inflate_batch_dev(c, sl.d_in[b], sl.d_off[b], sl.d_len[b], m, target, ZWZ_DEV_STRIDE, d_len_out, d_stat_out);
    };
    uint64_t mine[kFields] = {kNoFile, 0, kNoFile, 0, 0};
    shielded("inflating a record range", [&] {
        lens.resize(n); stats.resize(n); fds.assign(insts.size(), -1); cursor.assign(insts.size(), 0);
        if (n && rc == ZWZ_OK) {
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_lens), (size_t)n * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_stat), (size_t)n * sizeof(uint32_t));
            // (no early return on any of these: the failure travels through the exchanges below)
            if (e != hipSuccess) { (void)hipGetLastError(); set_error("no device memory for the lengths of %u chunks: %s", n, hipGetErrorString(e)); rc = ZWZ_E_NOMEM; }
            else { rc = make_slices(c, cap, sl); have_slices = rc == ZWZ_OK; }
            if (rc == ZWZ_OK) {
                // what stays resident between the phases: whole slices, within the budget; a failed allocation halves it, down to nothing
                uint64_t budget = n;
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = free_b > ((size_t)1 << 30) ? (free_b - ((size_t)1 << 30)) / ZWZ_DEV_STRIDE : 0;
                if (const char* v = getenv("ZWZ_MAX_RANGE_CHUNKS")) budget = (uint64_t)std::max(0L, atol(v));
                uint32_t want = (uint32_t)std::min<uint64_t>(nslices, budget / cap);               // slices
                while (want) {
                    const uint32_t chunks = std::min(n, want * cap);
                    if (hipMalloc(reinterpret_cast<void**>(&d_big), (size_t)chunks * ZWZ_DEV_STRIDE) == hipSuccess) { res = chunks; break; }
                    (void)hipGetLastError(); d_big = nullptr; want /= 2;
                }
                if (timeline()) fprintf(stderr, "zwz: split decode: rank %d holds %u of its %u chunks on the device between the phases (%u slices of %u)\n", rank, res, n, nslices, cap);
            }
        }
        // ---- phase 1: payloads in, range inflated; the first `res` chunks stay, of the others only the decoded lengths come back
        if (nslices && rc == ZWZ_OK) start_fill(0);
        for (uint32_t s = 0; s < nslices && rc == ZWZ_OK; s++) {
            const int b = (int)(s & 1u);
            const uint32_t g0 = s * cap, g1 = std::min(n, g0 + cap);
            pool.wait(fill_group[b]);
            rc = inflate_slice(s, g1 <= res ? d_big + (size_t)g0 * ZWZ_DEV_STRIDE : sl.d_out[b], d_lens + g0, d_stat + g0);
            if (rc) break;
            hipError_t e = hipEventRecord(sl.done[b], c->stream);
            if (e == hipSuccess && s >= 1) e = hipEventSynchronize(sl.done[(s - 1) & 1]);      // the other buffer pair is free again
            if (e != hipSuccess) { rc = hip_fail(e, "hipEventSynchronize"); break; }
            if (s + 1 < nslices) start_fill(s + 1);
        }
        pool.wait(fill_group[0]); pool.wait(fill_group[1]);
        if (rc == ZWZ_OK && n) {
            hipError_t e = hipMemcpyAsync(lens.data(), d_lens, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(stats.data(), d_stat, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = hip_fail(e, "decoded lengths");
        }
        // Files are created (and truncated) by the rank that holds their first record, BEFORE the exchange: a rank that joins a
        // file further in opens it only afterwards, so no write can be lost to a late O_TRUNC.
        for (uint32_t g = j0; g < j1 && rc == ZWZ_OK; g++) {
            const uint32_t inst = jobs[g].inst;
            if (g != 0 && jobs[g - 1].inst == inst) continue;
            const std::string file_path = sink.path_of(inst);
            sink.make_parent(file_path);
            fds[inst] = open(file_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            if (fds[inst] < 0) { std::lock_guard<std::mutex> l(sink.log_mutex); fprintf(stderr, "Error creating output file: %s\n", file_path.c_str()); }
        }
        if (n && rc == ZWZ_OK) {
            mine[0] = jobs[j0].inst; mine[2] = jobs[j1 - 1].inst;
            for (uint32_t g = 0; g < n && jobs[j0 + g].inst == mine[0]; g++) mine[1] += lens[g];
            for (uint32_t g = n; g-- > 0 && jobs[j0 + g].inst == mine[2];) mine[3] += lens[g];
        }
    });
    pool.wait(fill_group[0]); pool.wait(fill_group[1]);      // (an exception may have left fills in flight)
    // ---- exchange: (first file, bytes decoded into it, last file, bytes decoded into it, status) per rank.  Every rank
    // takes part even after a local failure, so that nobody waits for a rank that has given up.
    if (rc != ZWZ_OK) { mine[0] = kNoFile; mine[1] = 0; mine[2] = kNoFile; mine[3] = 0; }
    mine[4] = (uint64_t)(rc != ZWZ_OK);
    if (exchange(user, mine, all.data(), kFields) != 0) { cleanup(); set_error("rank exchange failed"); return ZWZ_E_IO; }
    for (int r = 0; r < nranks; r++) if (all[(size_t)r * kFields + 4]) { if (rc == ZWZ_OK) { set_error("rank %d failed while decoding its record range", r); rc = ZWZ_E_IO; } }
    // ---- phase 2: this rank's bytes to their places
    shielded("writing a record range", [&] {
        if (!(rc == ZWZ_OK && n)) return;
        for (int r = 0; r < rank; r++)                      // what earlier ranks decoded into the file my range starts in
            if (all[(size_t)r * kFields + 2] == mine[0]) cursor[mine[0]] += all[(size_t)r * kFields + 3];
        const uint32_t first_again = (res + cap - 1) / cap;            // slices from here on are inflated a second time
        if (first_again < nslices) start_fill(first_again);
        for (uint32_t s = 0; s < nslices; s++) {
            const int b = (int)(s & 1u);
            const uint32_t g0 = s * cap, g1 = std::min(n, g0 + cap);
            // (h_out[b] was the slice before last's; its writes were waited for in the last trip, before that trip's own were submitted -- a wait
            //  here would also wait for the LAST slice's writes and serialise this slice's copy / second inflate behind them: ADVICE r4)
            hipError_t e = hipSuccess;
            if (g1 <= res) e = hipMemcpyAsync(sl.h_out[b], d_big + (size_t)g0 * ZWZ_DEV_STRIDE, (size_t)(g1 - g0) * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, c->stream);
            else {
                pool.wait(fill_group[b]);
                rc = inflate_slice(s, sl.d_out[b], sl.d_olen[b], sl.d_st[b]);
                if (rc) break;
                e = hipMemcpyAsync(sl.h_out[b], sl.d_out[b], (size_t)(g1 - g0) * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(sl.h_olen[b], sl.d_olen[b], (size_t)(g1 - g0) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(sl.h_st[b], sl.d_st[b], (size_t)(g1 - g0) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { rc = hip_fail(e, "decoded range to host"); break; }
            if (g1 > res) {
                // the second decode must be the first: same length, same status (how the stream ended) for every chunk
                for (uint32_t g = g0; g < g1; g++) {
                    if (sl.h_olen[b][g - g0] != lens[g]) { set_error("a chunk decoded to %u bytes the second time, %u the first", sl.h_olen[b][g - g0], lens[g]); rc = ZWZ_E_IO; }
                    else if (sl.h_st[b][g - g0] != stats[g]) { set_error("a chunk's stream ended with status %u the second time, %u the first", sl.h_st[b][g - g0], stats[g]); rc = ZWZ_E_IO; }
                }
                if (rc) break;
                if (s + 1 < nslices) { pool.wait(fill_group[(s + 1) & 1u]); start_fill(s + 1); }     // (h_in of the other pair is free: its slice is on the device or done)
            }
            if (s >= 1) pool.wait(write_group);             // (two buffer pairs: the work above overlapped the previous slice's writes)
            for (uint32_t r0 = g0; r0 < g1;) {
                const uint32_t inst = jobs[j0 + r0].inst;
                uint32_t r1 = r0;
                uint64_t bytes = 0;
                while (r1 < g1 && r1 - r0 < 128 && jobs[j0 + r1].inst == inst) bytes += lens[r1++];
                const bool joins = r0 == 0 && j0 != 0 && jobs[j0 - 1].inst == inst;     // a file begun by an earlier rank
                if (fds[inst] < 0 && joins) fds[inst] = open(sink.path_of(inst).c_str(), O_WRONLY, 0666);
                const uint64_t off0 = cursor[inst];
                cursor[inst] += bytes;
                const int fd = fds[inst];
                if (fd >= 0)
                    pool.submit(write_group, [&, b, g0, fd, r0, r1, off0] {
                        std::vector<struct iovec> iov;
                        for (uint32_t k = r0; k < r1; k++) if (lens[k]) iov.push_back({sl.h_out[b] + (size_t)(k - g0) * ZWZ_DEV_STRIDE, (size_t)lens[k]});
                        pwrite_chunks(fd, off0, iov);
                    });
                r0 = r1;
            }
        }
        pool.wait(write_group);
    });
    pool.wait(fill_group[0]); pool.wait(fill_group[1]); pool.wait(write_group);
    for (int& fd : fds) if (fd >= 0) { close(fd); fd = -1; }
    cleanup();
    // ---- barrier, then verification by whoever holds a file's last record
    uint64_t flag[1] = {(uint64_t)(rc != ZWZ_OK)};
    if (exchange(user, flag, flags.data(), 1) != 0) { set_error("rank exchange failed"); return ZWZ_E_IO; }
    for (int r = 0; r < nranks; r++) if (flags[(size_t)r] && rc == ZWZ_OK) { set_error("rank %d failed while writing its record range", r); rc = ZWZ_E_IO; }
    if (rc == ZWZ_OK) {
        Pool::Group md5_group;
        Drain d2{pool, {&md5_group}};
        uint32_t prev = 0xffffffffu;
        for (uint32_t g = j0; g < j1; g++) {
            const uint32_t inst = jobs[g].inst;
            const bool last_of_file = g + 1 == T || jobs[g + 1].inst != inst;
            if (last_of_file && inst != prev) pool.submit(md5_group, [&sink, inst] { sink.hash_from_disk(inst); });
            if (last_of_file) prev = inst;
        }
        if (rank == 0)       // files without a decodable record are still created (the reference opens on the first record of a path)
            for (uint32_t i = 0; i < insts.size(); i++)
                if (insts[i].order.empty()) {
                    const std::string file_path = sink.path_of(i);
                    sink.make_parent(file_path);
                    if (FILE* f = fopen(file_path.c_str(), "wb")) fclose(f);
                    Md5 m; char hex[33]; m.hex(hex);
                    sink.verdict(i, hex);
                }
        pool.wait(md5_group);
    }
    return rc;
}

int decompress_dir_impl(zwz_ctx* c, const char* src_dir, const char* dst_dir, int rank, int nranks, zwz_allgather_u64_fn exchange,
                        void* user, int* md5_mismatches) {
    std::vector<std::string> shards;
    for (const auto& e : fs::directory_iterator(src_dir))   // decompression.cpp:168-172
        if (e.path().extension() == ".zwz") shards.push_back(e.path().string());
    std::sort(shards.begin(), shards.end());                // every rank must see the same order
    // Shard j -> rank j mod N when there is a shard for every rank (the reference: one OpenMP thread per shard,
    // decompression.cpp:174-177); fewer shards than ranks (BASELINE config 5: ONE shard) are each split over all ranks.
    const bool split = nranks > 1 && shards.size() < (size_t)nranks;
    if (split && !exchange) { set_error("%zu shard(s) for %d ranks needs the rank exchange callback", shards.size(), nranks); return ZWZ_E_INVALID; }

    HIPCHK(hipSetDevice(c->device));
    Pool pool(host_threads());
    std::atomic<int> mismatches{0};
    std::mutex log_mutex;
    int rc = ZWZ_OK;
    bool damaged = false;
    for (size_t j = 0; j < shards.size(); j++) {
        if (!split && (int)(j % (size_t)nranks) != rank) continue;
        Mapped blob;
        bool opened = false;
        rc = map_shard(shards[j], blob, opened);
        if (rc && split) {      // the other ranks are on their way into this shard's exchanges: join them, failure flag up
            std::vector<FileInst> none;
            DecodeSink sink{dst_dir, none, mismatches, log_mutex};
            (void)decode_shard_split(c, pool, blob, none, sink, rank, nranks, exchange, user, rc);
        }
        if (rc) break;
        if (!opened && !split) continue;
        std::vector<FileInst> insts;
        if (!parse_shard(blob, insts)) { damaged = true; if (rank == 0 || !split) fprintf(stderr, "Malformed shard (damaged from some record on): %s\n", shards[j].c_str()); }
        DecodeSink sink{dst_dir, insts, mismatches, log_mutex};
        // A path that occurs as two instances (duplicate lines in the file list) must be written in shard order, the later instance
        // over the earlier (decode_whole_shard does that; the record-range split writes instances concurrently from several
        // ranks).  Such a shard is not split: rank 0 decodes it whole, and every rank still goes through the split's two exchanges
        // -- with nothing of its own -- so that a rank that could not even map the shard (above) meets the others there.
        bool repeated_path = false;
        if (split) {
            std::unordered_set<std::string> seen;
            for (const FileInst& fi : insts) if (!seen.insert(fi.rel).second) { repeated_path = true; break; }
        }
        if (split && repeated_path) {
            int rc0 = rank == 0 ? decode_whole_shard(c, pool, blob, insts, sink) : ZWZ_OK;
            std::vector<FileInst> none;
            DecodeSink nothing{dst_dir, none, mismatches, log_mutex};
            rc = decode_shard_split(c, pool, blob, none, nothing, rank, nranks, exchange, user, rc0);
        } else
            rc = split ? decode_shard_split(c, pool, blob, insts, sink, rank, nranks, exchange, user, ZWZ_OK) : decode_whole_shard(c, pool, blob, insts, sink);
        if (rc) break;
    }
    if (md5_mismatches) *md5_mismatches = mismatches.load();
    { std::string why; if (pool.failed(&why) && rc == ZWZ_OK) { set_error("a pipeline task failed: %s", why.c_str()); rc = ZWZ_E_IO; } }
    if (rc == ZWZ_OK && damaged) { set_error("a shard in %s ends inside a record; what precedes the damage was decoded", src_dir); rc = ZWZ_E_FORMAT; }
    return rc;
}

}  // namespace
