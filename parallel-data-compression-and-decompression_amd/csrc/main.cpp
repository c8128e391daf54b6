// main.cpp -- `main compress|decompress <src> <dst>`: the reference's command line (main.cpp:78-159)
// over the GPU pipeline.  Same argv, same output names (<dst>/compressed_<rank>.zwz, the side file
// <parent(src)>/sorted_files_by_size.txt), same final banner.  One process drives one GPU; rank and
// world size come from the launcher's environment (ZWZ_RANK/ZWZ_NRANKS, else RANK/WORLD_SIZE, else
// OMPI/PMI variables, else 0/1) where the reference asked MPI.
//
// What the reference does with MPI (two broadcasts of the record PATH, main.cpp:27,35; three barriers, :41,131,144)
// this binary does through small marker files in <dst> -- the reference likewise assumes one file system for the list
// and the shards (compression.cpp:25).  The torch.distributed launcher (python -m ... cli) uses RCCL for the same steps.
// The markers cannot be confused with those of an earlier run that crashed and left its own behind:
//   * every name carries a run id (ZWZ_RUN_ID, else the launcher's pid + start time: ranks of one launch share a parent);
//   * every rank announces itself with a nonce (pid + start time); rank 0 publishes the list path together with its own
//     nonce and every nonce it has seen, atomically (write + rename), and keeps adding late arrivals; a rank only accepts
//     a publication that names its nonce, i.e. one written by a rank 0 that has seen THIS process;
//   * completion markers carry rank 0's nonce and the rank's return code; rank 0 fails if a rank failed or never arrived.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/zwz.h"

namespace {

int env_int(const char* const* names, int dflt) {
    for (; *names; names++) if (const char* v = getenv(*names)) return atoi(v);
    return dflt;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::string start_time_of(long pid) {          // field 22 of /proc/<pid>/stat: start time in clock ticks since boot
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%ld/stat", pid);
    FILE* f = fopen(path, "r");
    if (!f) return "0";
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char* p = strrchr(buf, ')');          // the command name may contain spaces
    if (!p) return "0";
    int field = 2;
    for (p++; *p; p++) if (*p == ' ' && ++field == 22) { p++; break; }
    std::string s;
    for (; *p && *p != ' '; p++) s += *p;
    return s.empty() ? "0" : s;
}

bool write_atomic(const std::string& path, const std::string& content) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "w");
    if (!f) return false;
    const bool ok = fwrite(content.data(), 1, content.size(), f) == content.size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); return false; }
    return true;
}

bool read_all(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    out.clear();
    char buf[4096];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, k);
    fclose(f);
    return true;
}

std::vector<std::string> names_with_prefix(const std::string& dir, const std::string& prefix) {
    std::vector<std::string> out;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) if (!strncmp(e->d_name, prefix.c_str(), prefix.size()) && !strstr(e->d_name, ".tmp")) out.push_back(e->d_name);
        closedir(d);
    }
    std::sort(out.begin(), out.end());
    return out;
}

std::vector<std::string> lines_of(const std::string& s) {
    std::vector<std::string> out;
    size_t p = 0;
    while (p <= s.size()) {
        const size_t e = s.find('\n', p);
        if (e == std::string::npos) { if (p < s.size()) out.push_back(s.substr(p)); break; }
        out.push_back(s.substr(p, e - p));
        p = e + 1;
    }
    return out;
}

// The ranks of one launch, meeting through files named <dir>/<stem>_*.
struct Rendezvous {
    std::string dir, stem;          // stem = ".zwz_<op>_<run id>"
    int rank = 0, world = 1;
    std::string nonce, nonce0;      // mine; rank 0's (known once published / accepted)
    std::string payload;            // what rank 0 publishes beside the nonces (the record path)
    std::vector<std::string> seen;  // rank 0: nonces published so far
    double timeout_s = 600;
    uint32_t round = 0;

    std::string path(const std::string& tail) const { return dir + "/" + stem + "_" + tail; }

    // a rank announces itself once its codec context exists (or has failed): the second line says whether it could join an RCCL communicator
    void hello(bool healthy) { if (rank != 0) write_atomic(path("here_" + std::to_string(rank) + "_" + nonce), nonce + "\n" + (healthy ? "ok" : "bad") + "\n"); }

    // rank 0, before it publishes anything: wait for every rank's announcement (MPI_Barrier, main.cpp:41) and learn whether all of them
    // are healthy.  The choice between RCCL and marker files is made HERE, once, for the whole job -- a rank that decided alone
    // left the others waiting inside ncclCommInitRank or a collective.  false: some rank never showed up (then: files).
    bool gather(bool& all_healthy) {
        const double t0 = now_s();
        for (;;) {
            std::vector<bool> have((size_t)world, false);
            have[0] = true; all_healthy = true;
            for (const std::string& n : names_with_prefix(dir, stem + "_here_")) {
                const std::string rest = n.substr(stem.size() + 6);               // "<rank>_<nonce>"
                const size_t us = rest.find('_');
                if (us == std::string::npos) continue;
                const int r = atoi(rest.substr(0, us).c_str());
                if (r <= 0 || r >= world) continue;
                std::string content;
                if (!read_all(dir + "/" + n, content)) continue;
                const std::vector<std::string> ls = lines_of(content);
                have[(size_t)r] = true;
                if (ls.size() < 2 || ls[1] != "ok") all_healthy = false;
            }
            if (std::all_of(have.begin(), have.end(), [](bool b) { return b; })) return true;
            if (now_s() - t0 > timeout_s) { all_healthy = false; return false; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }

    // rank 0: (re)publish payload + every announced nonce; true once every rank has announced itself at least once
    bool publish() {
        std::vector<std::string> nonces;
        std::vector<bool> have((size_t)world, false);
        have[0] = true;
        for (const std::string& n : names_with_prefix(dir, stem + "_here_")) {
            const std::string rest = n.substr(stem.size() + 6);               // "<rank>_<nonce>"
            const size_t us = rest.find('_');
            if (us == std::string::npos) continue;
            const int r = atoi(rest.substr(0, us).c_str());
            if (r > 0 && r < world) { have[(size_t)r] = true; nonces.push_back(rest.substr(us + 1)); }
        }
        if (nonces != seen) {
            std::string content = nonce + "\n" + payload + "\n";
            for (const std::string& n : nonces) content += n + "\n";
            if (write_atomic(path("list_" + nonce), content)) seen = nonces;
        }
        return std::all_of(have.begin(), have.end(), [](bool b) { return b; });
    }

    // rank > 0: wait for a publication that names this process
    bool accept() {
        const double t0 = now_s();
        for (;;) {
            for (const std::string& n : names_with_prefix(dir, stem + "_list_")) {
                std::string content;
                if (!read_all(dir + "/" + n, content)) continue;
                const std::vector<std::string> ls = lines_of(content);
                if (ls.size() >= 2 && std::find(ls.begin() + 2, ls.end(), nonce) != ls.end()) { nonce0 = ls[0]; payload = ls[1]; return true; }
            }
            if (now_s() - t0 > timeout_s) return false;
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }

    // all-gather of `count` 64-bit values per rank (zwz_allgather_u64_fn); also a barrier
    int allgather(const uint64_t* mine, uint64_t* all, uint32_t count) {
        const std::string tag = "x" + std::to_string(round++) + "_";
        std::string content;
        for (uint32_t i = 0; i < count; i++) content += std::to_string((unsigned long long)mine[i]) + "\n";
        if (!write_atomic(path(tag + std::to_string(rank) + "_" + nonce0), content)) return -1;
        const double t0 = now_s();
        for (int r = 0; r < world; r++) {
            std::string got;
            while (!read_all(path(tag + std::to_string(r) + "_" + nonce0), got)) {
                if (rank == 0) publish();                                       // late arrivals still need the publication
                if (now_s() - t0 > timeout_s) return -1;
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
            const std::vector<std::string> ls = lines_of(got);
            if (ls.size() < count) return -1;
            for (uint32_t i = 0; i < count; i++) all[(size_t)r * count + i] = strtoull(ls[i].c_str(), nullptr, 10);
        }
        return 0;
    }

    // MPI_Barrier + status (main.cpp:144): every rank reports its return code; rank 0 collects them and cleans up
    int finish(int rc) {
        if (world <= 1) return rc;
        if (!nonce0.empty()) write_atomic(path("done_" + std::to_string(rank) + "_" + nonce0), std::to_string(rc) + "\n");
        if (rank != 0) return rc;
        const double t0 = now_s();
        int worst = rc;
        for (int r = 1; r < world; r++) {
            std::string got;
            while (!read_all(path("done_" + std::to_string(r) + "_" + nonce0), got)) {
                publish();
                if (now_s() - t0 > (getenv("ZWZ_RENDEZVOUS_TIMEOUT") ? timeout_s : 3600.0)) { fprintf(stderr, "rank 0: rank %d never reported completion\n", r); return worst ? worst : 3; }
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            const int rr = atoi(got.c_str());
            if (rr != 0) { fprintf(stderr, "rank 0: rank %d failed with status %d\n", r, rr); if (!worst) worst = rr; }
        }
        for (const std::string& n : names_with_prefix(dir, stem + "_")) unlink((dir + "/" + n).c_str());
        return worst;
    }
};

// RCCL over xGMI for the ranks' few collectives (north star): the list CONTENTS broadcast from rank 0
// (the reference broadcasts the path and reads a shared file system, main.cpp:24-39), the completion status all-reduced
// (MPI_Barrier, main.cpp:144), and zwz_decompress_dir_ranked's all-gather.  The communicator's 128-byte id is the one
// thing that still travels through the marker-file handshake above (an MPI launcher would carry it out of band).
// Used when every rank has a GPU of its own; ranks that share a device (RCCL refuses those) keep to the files.
struct Rccl {
    ncclComm_t comm = nullptr; hipStream_t stream = nullptr; int rank = 0, world = 1; bool on = false;
    static std::string hex(const void* p, size_t n) {
        static const char* d = "0123456789abcdef"; std::string s; const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; i++) { s += d[b[i] >> 4]; s += d[b[i] & 15]; }
        return s;
    }
    static bool unhex(const std::string& s, void* p, size_t n) {
        if (s.size() != 2 * n) return false;
        for (char c : s) if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f'))) return false;
        auto v = [](char c) { return c >= 'a' ? c - 'a' + 10 : c - '0'; };
        for (size_t i = 0; i < n; i++) static_cast<unsigned char*>(p)[i] = (unsigned char)(v(s[2 * i]) << 4 | v(s[2 * i + 1]));
        return true;
    }
    bool init(const ncclUniqueId& id, int r, int w) {
        rank = r; world = w;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        on = ncclCommInitRank(&comm, w, id, r) == ncclSuccess;
        return on;
    }
    template <class F> bool with_dev(size_t bytes, F&& f) {          // a device scratch buffer for one collective
        void* d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) return false;
        const bool ok = f(d) && hipStreamSynchronize(stream) == hipSuccess;
        (void)hipFree(d);
        return ok;
    }
    bool broadcast(std::string& bytes) {                            // rank 0's bytes on every rank
        uint64_t n = rank == 0 ? bytes.size() : 0;
        if (!with_dev(8, [&](void* d) {
                return hipMemcpy(d, &n, 8, hipMemcpyHostToDevice) == hipSuccess && ncclBroadcast(d, d, 8, ncclChar, 0, comm, stream) == ncclSuccess &&
                       hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(&n, d, 8, hipMemcpyDeviceToHost) == hipSuccess; })) return false;
        bytes.resize((size_t)n);
        return with_dev((size_t)n, [&](void* d) {
            if (rank == 0 && n && hipMemcpy(d, bytes.data(), (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return false;
            if (n && ncclBroadcast(d, d, (size_t)n, ncclChar, 0, comm, stream) != ncclSuccess) return false;
            return hipStreamSynchronize(stream) == hipSuccess && (!n || hipMemcpy(&bytes[0], d, (size_t)n, hipMemcpyDeviceToHost) == hipSuccess); });
    }
    int allgather(const uint64_t* mine, uint64_t* all, uint32_t count) {
        const size_t one = (size_t)count * 8;
        return with_dev(one * (size_t)(world + 1), [&](void* d) {
            char* send = static_cast<char*>(d) + one * (size_t)world;
            return hipMemcpy(send, mine, one, hipMemcpyHostToDevice) == hipSuccess && ncclAllGather(send, d, count, ncclUint64, comm, stream) == ncclSuccess &&
                   hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(all, d, one * (size_t)world, hipMemcpyDeviceToHost) == hipSuccess; }) ? 0 : -1;
    }
    // "a final gather of per-shard .zwz blobs" (north star): the protocol is the library's (zwz_gather_shards, csrc/zwz_gather.cpp -- the same
    // code cli.py drives over torch.distributed); here are its hooks over RCCL.  A piece goes host -> device staging -> ncclSend on the sender,
    // ncclRecv -> device staging -> host on rank 0: ONE device buffer and one pinned host buffer of a piece's size per rank, whatever the
    // shards' sizes (round 3 allocated every peer's whole shard on rank 0 at once).
    void* g_dev = nullptr; void* g_pin = nullptr; uint64_t g_piece = 0;
    static int hook_allgather(void* u, const uint64_t* mine, uint64_t* all, uint32_t count) { return static_cast<Rccl*>(u)->allgather(mine, all, count); }
    static int hook_prepare(void* u, uint64_t piece) {
        Rccl* r = static_cast<Rccl*>(u);
        r->g_piece = piece;
        return hipMalloc(&r->g_dev, piece) == hipSuccess && hipHostMalloc(&r->g_pin, piece, hipHostMallocDefault) == hipSuccess ? 0 : -1;
    }
    static void hook_release(void* u) {
        Rccl* r = static_cast<Rccl*>(u);
        if (r->g_dev) (void)hipFree(r->g_dev);
        if (r->g_pin) (void)hipHostFree(r->g_pin);
        r->g_dev = r->g_pin = nullptr; r->g_piece = 0;
    }
    static int hook_send(void* u, const void* buf, uint64_t n, int to) {
        Rccl* r = static_cast<Rccl*>(u);
        if (!r->g_dev || n > r->g_piece) return -1;
        memcpy(r->g_pin, buf, (size_t)n);
        (void)hipMemcpyAsync(r->g_dev, r->g_pin, (size_t)n, hipMemcpyHostToDevice, r->stream);     // (a failed copy still sends the piece: the receiver is waiting; its write then fails the MD5 later)
        return ncclSend(r->g_dev, (size_t)n, ncclChar, to, r->comm, r->stream) == ncclSuccess && hipStreamSynchronize(r->stream) == hipSuccess ? 0 : -1;
    }
    static int hook_recv(void* u, void* buf, uint64_t n, int from) {
        Rccl* r = static_cast<Rccl*>(u);
        if (!r->g_dev || n > r->g_piece) return -1;
        if (ncclRecv(r->g_dev, (size_t)n, ncclChar, from, r->comm, r->stream) != ncclSuccess) return -1;
        const bool copied = hipMemcpyAsync(r->g_pin, r->g_dev, (size_t)n, hipMemcpyDeviceToHost, r->stream) == hipSuccess;
        if (hipStreamSynchronize(r->stream) != hipSuccess) return -1;
        if (copied) memcpy(buf, r->g_pin, (size_t)n); else memset(buf, 0, (size_t)n);
        return 0;
    }
    bool gather_files(const std::string& my_path, const std::string& out_dir) {
        zwz_gather_hooks hk{this, hook_allgather, hook_send, hook_recv, hook_prepare, hook_release};
        return zwz_gather_shards(rank, world, my_path.c_str(), out_dir.c_str(), 0, &hk) == 1;
    }
    int worst_status(int rc) {                                      // barrier + every rank learns whether any rank failed
        int64_t v = rc != 0;
        const bool ok = with_dev(8, [&](void* d) {
            return hipMemcpy(d, &v, 8, hipMemcpyHostToDevice) == hipSuccess && ncclAllReduce(d, d, 1, ncclInt64, ncclSum, comm, stream) == ncclSuccess &&
                   hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(&v, d, 8, hipMemcpyDeviceToHost) == hipSuccess; });
        return !ok ? 3 : (rc ? rc : (v ? 2 : 0));
    }
    void done() { if (comm) ncclCommDestroy(comm); if (stream) (void)hipStreamDestroy(stream); comm = nullptr; stream = nullptr; on = false; }
};

struct Exchange { Rendezvous* rv; Rccl* nc; };
int exchange_cb(void* user, const uint64_t* mine, uint64_t* all, uint32_t count) {
    Exchange* x = static_cast<Exchange*>(user);
    return x->nc->on ? x->nc->allgather(mine, all, count) : x->rv->allgather(mine, all, count);
}

}  // namespace

int main(int argc, char* argv[]) {
    const auto start = std::chrono::steady_clock::now();
    static const char* const rank_vars[] = {"ZWZ_RANK", "RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", nullptr};
    static const char* const size_vars[] = {"ZWZ_NRANKS", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", nullptr};
    static const char* const dev_vars[] = {"ZWZ_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", nullptr};
    const int world_rank = env_int(rank_vars, 0), world_size = env_int(size_vars, 1);

    if (argc < 4) {   // main.cpp:88-92
        fprintf(stderr, "Usage: %s <compress/decompress> <source directory path> <output directory path>\n", argv[0]);
        return 1;
    }
    std::string operation = argv[1], source_path = argv[2], output_path = argv[3];
    if (!source_path.empty() && source_path.back() == '/') source_path.pop_back();   // main.cpp:72-76
    if (!output_path.empty() && output_path.back() == '/') output_path.pop_back();
    printf("source_path: %s\noutput_path: %s\n", source_path.c_str(), output_path.c_str());

    if (operation != "compress" && operation != "decompress") {   // main.cpp:138-142
        fprintf(stderr, "Invalid operation: %s. Please use 'compress' or 'decompress'.\n", operation.c_str());
        return 1;
    }
    if (world_rank < 0 || world_size < 1 || world_rank >= world_size) { fprintf(stderr, "bad rank %d of %d\n", world_rank, world_size); return 1; }
    if (world_rank == 0) {   // main.cpp:105-129
        struct stat st {};
        if (stat(source_path.c_str(), &st) != 0) { fprintf(stderr, "Source path does not exist.\n"); return 1; }
        if (stat(output_path.c_str(), &st) != 0) {
            if (mkdir(output_path.c_str(), 0777) == -1) { perror("Failed to create output directory"); return 1; }
        } else if (!S_ISDIR(st.st_mode)) { fprintf(stderr, "Output path is not a directory.\n"); return 1; }
    }

    Rendezvous rv;
    rv.dir = output_path; rv.rank = world_rank; rv.world = world_size;
    {
        const char* id = getenv("ZWZ_RUN_ID");
        const long parent = (long)getppid();
        rv.stem = ".zwz_" + operation + "_" + (id && *id ? std::string(id) : "p" + std::to_string(parent) + "s" + start_time_of(parent));
        rv.nonce = std::to_string((long)getpid()) + "s" + start_time_of((long)getpid());
        if (world_rank == 0) rv.nonce0 = rv.nonce;
        if (const char* t = getenv("ZWZ_RENDEZVOUS_TIMEOUT")) rv.timeout_s = atof(t);
    }
    if (world_size > 1 && world_rank != 0) {          // (<dst> exists once rank 0 is past its checks: MPI_Barrier, main.cpp:131)
        const double t0 = now_s();
        struct stat st {};
        while (stat(output_path.c_str(), &st) != 0 && now_s() - t0 < rv.timeout_s) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }

    const bool trace = getenv("ZWZ_VERBOSE") != nullptr || getenv("ZWZ_TIMELINE") != nullptr;
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count(); };
    // Rank 0's file sort (file_sort.cpp:24-43) needs no GPU: it runs beside the context's creation (HIP start-up, the kernels' self-tests:
    // 0.1 - 0.3 s; the sort of BASELINE configs[3]'s 370 000 files as long again).
    char sorted_record[4096] = "";
    int sort_rc = ZWZ_OK;
    std::thread sort_thread;
    const bool sort_here = operation == "compress" && world_rank == 0 && !getenv("ZWZ_FILE_RECORD");
    if (sort_here) sort_thread = std::thread([&] { sort_rc = zwz_sort_files_by_size(source_path.c_str(), sorted_record, sizeof sorted_record);
                                                    if (trace) fprintf(stderr, "zwz: file list sorted at %.3f s\n", since()); });
    int device_count = 0;
    zwz_device_count(&device_count);
    zwz_ctx* ctx = nullptr;
    const int dev = device_count > 0 ? env_int(dev_vars, world_rank) % device_count : 0;
    int rc = zwz_ctx_create(dev, 0, &ctx);
    if (trace) fprintf(stderr, "zwz: context ready at %.3f s\n", since());
    if (sort_thread.joinable()) sort_thread.join();
    if (rc != ZWZ_OK) fprintf(stderr, "zwz: %s (%s)\n", zwz_strerror(rc), zwz_last_error());

    // ---- RCCL communicator (see struct Rccl): its id rides on the handshake's publication -- and only if EVERY rank can join:
    // each rank says so in its announcement, rank 0 reads them all before it publishes (Rendezvous::gather), so either all
    // ranks find an id and enter ncclCommInitRank, or none does and all keep to the marker files.
    Rccl nc;
    const char* comm_env = getenv("ZWZ_COMM");
    const bool want_rccl = rc == ZWZ_OK && (world_size > 1 ? device_count >= world_size : false) && !(comm_env && !strcmp(comm_env, "files"));
    const bool force_rccl = rc == ZWZ_OK && comm_env && !strcmp(comm_env, "rccl");          // (also with one rank: exercises the path on a 1-GPU box)
    if (world_size > 1 && world_rank != 0) rv.hello(want_rccl || force_rccl);
    bool all_here = true, all_healthy = true;
    if (world_size > 1 && world_rank == 0) all_here = rv.gather(all_healthy);               // (the ranks wait in accept() meanwhile, as they did for the file sort)
    std::string nccl_hex;
    auto make_id = [&] {        // rank 0, at the moment it publishes: its own state counts too (a failed file sort means no communicator)
        if (world_rank == 0 && rc == ZWZ_OK && (want_rccl || force_rccl) && all_here && all_healthy) {
            ncclUniqueId id;
            if (ncclGetUniqueId(&id) == ncclSuccess) nccl_hex = Rccl::hex(&id, sizeof id);
        }
        if (world_rank == 0 && world_size > 1 && trace) fprintf(stderr, "zwz: rank 0: %s for this job (%s)\n", nccl_hex.empty() ? "marker files" : "RCCL",
                                                                 !all_here ? "a rank never announced itself" : !all_healthy ? "a rank cannot join a communicator" : rc != ZWZ_OK ? "rank 0 failed" : "every rank is healthy");
    };

    if (operation == "compress") {
        char record[4096] = "";
        if (const char* fr = getenv("ZWZ_FILE_RECORD")) snprintf(record, sizeof record, "%s", fr);
        if (world_rank == 0) {
            printf("Compressing folder: %s\n", source_path.c_str());
            if (sort_here) { snprintf(record, sizeof record, "%s", sorted_record); if (rc == ZWZ_OK) rc = sort_rc; }
            printf("File record saved location: %s\n", record);
            if (trace) fprintf(stderr, "zwz: file list ready at %.3f s\n", since());
            make_id();
            if (world_size > 1) { rv.payload = std::string(record) + "\t" + nccl_hex; rv.publish(); if (!all_here) fprintf(stderr, "rank 0: not every rank announced itself; going on\n"); }
        } else if (world_size > 1) {   // the reference broadcasts the record path (main.cpp:24-39)
            if (!rv.accept()) { fprintf(stderr, "rank %d: no file list from rank 0\n", world_rank); if (rc == ZWZ_OK) rc = 3; }
            else {
                const size_t tab = rv.payload.find('\t');
                snprintf(record, sizeof record, "%s", rv.payload.substr(0, tab).c_str());
                if (tab != std::string::npos) nccl_hex = rv.payload.substr(tab + 1);
            }
        }
        if ((world_size > 1 || force_rccl) && !nccl_hex.empty() && rc == ZWZ_OK) {
            ncclUniqueId id;
            if (!Rccl::unhex(nccl_hex, &id, sizeof id) || !nc.init(id, world_rank, world_size)) { fprintf(stderr, "rank %d: RCCL communicator not available, using marker files\n", world_rank); nc.done(); }
        }
        std::string private_list;
        if (nc.on && rc == ZWZ_OK) {
            // the list's CONTENTS over RCCL: every rank works from a private copy, so the list needs no shared file system
            std::string listing;
            if (world_rank == 0) { FILE* f = fopen(record, "rb"); if (f) { char buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) listing.append(buf, k); fclose(f); } }
            if (!nc.broadcast(listing)) { fprintf(stderr, "rank %d: RCCL broadcast of the file list failed\n", world_rank); rc = ZWZ_E_IO; }
            else if (world_rank != 0) {
                char tmpl[] = "/tmp/zwz_list_XXXXXX";                         // (created exclusively: nothing in a shared /tmp can stand in its place)
                const int fd = mkstemp(tmpl);
                bool ok = fd >= 0;
                for (size_t off = 0; ok && off < listing.size();) { const ssize_t k = write(fd, listing.data() + off, listing.size() - off); if (k <= 0) ok = false; else off += (size_t)k; }
                if (fd >= 0) { close(fd); private_list = tmpl; }
                if (!ok) { fprintf(stderr, "rank %d: cannot write a private copy of the file list under /tmp\n", world_rank); rc = ZWZ_E_IO; }
                else snprintf(record, sizeof record, "%s", private_list.c_str());
            }
            if (trace) fprintf(stderr, "zwz: rank %d has the list (%zu bytes) over RCCL\n", world_rank, listing.size());
        }
        // ZWZ_GATHER=1 (with RCCL): ranks other than 0 write their shard into a private directory and hand it to rank 0 over the
        // communicator; rank 0 writes it to <dst> -- <dst> then only has to exist on rank 0's host.
        const bool gather = nc.on && getenv("ZWZ_GATHER") && !strcmp(getenv("ZWZ_GATHER"), "1");
        std::string shard_dir = output_path;
        if (gather && world_rank != 0) { char tmpl[] = "/tmp/zwz_shard_XXXXXX"; if (mkdtemp(tmpl)) shard_dir = tmpl; else { fprintf(stderr, "rank %d: cannot create a private shard directory\n", world_rank); rc = ZWZ_E_IO; } }
        if (rc == ZWZ_OK) {
            printf("file_record: %s\n", record);
            if (world_rank < zwz_count_non_empty_lines(record)) rc = zwz_compress_dir(ctx, source_path.c_str(), shard_dir.c_str(), record, world_rank, world_size);
            else printf("Rank: %d - No file to compress\n", world_rank);
        }
        if (gather) {       // every rank of the communicator, whatever its own status (a failed or idle rank contributes nothing)
            const std::string mine_path = shard_dir + "/compressed_" + std::to_string(world_rank) + ".zwz";
            const bool gathered = nc.gather_files(rc == ZWZ_OK && world_rank != 0 ? mine_path : std::string(), output_path);
            if (trace) fprintf(stderr, "zwz: rank %d: shard gather over RCCL %s\n", world_rank, gathered ? "done" : "FAILED");
            if (!gathered) {
                // the shard stays where the rank wrote it: nothing is lost, and the message says where it is
                if (world_rank != 0 && rc == ZWZ_OK) fprintf(stderr, "rank %d: RCCL gather of the shards failed; this rank's shard is kept at %s\n", world_rank, mine_path.c_str());
                else fprintf(stderr, "rank %d: RCCL gather of the shards failed\n", world_rank);
                if (rc == ZWZ_OK) rc = ZWZ_E_IO;
            } else if (world_rank != 0 && shard_dir != output_path) { unlink(mine_path.c_str()); rmdir(shard_dir.c_str()); }
        }
        if (!private_list.empty()) unlink(private_list.c_str());
    } else {
        // The reference decodes on rank 0 only (main.cpp:61-68).  Here every rank takes its share: whole shards round-robin,
        // or record ranges of a shard when there are fewer shards than ranks (zwz_decompress_dir_ranked).
        if (world_size == 1) make_id();                                      // (ZWZ_COMM=rccl with one rank: a one-rank communicator)
        if (world_size > 1) {
            if (world_rank == 0) { make_id(); rv.payload = "\t" + nccl_hex; rv.publish(); if (!all_here) { fprintf(stderr, "rank 0: not every rank announced itself\n"); if (rc == ZWZ_OK) rc = 3; } }
            else if (!rv.accept()) { fprintf(stderr, "rank %d: rank 0 never showed up\n", world_rank); if (rc == ZWZ_OK) rc = 3; }
            else { const size_t tab = rv.payload.find('\t'); if (tab != std::string::npos) nccl_hex = rv.payload.substr(tab + 1); }
        }
        if ((world_size > 1 || force_rccl) && !nccl_hex.empty() && rc == ZWZ_OK) {
            ncclUniqueId id;
            if (!Rccl::unhex(nccl_hex, &id, sizeof id) || !nc.init(id, world_rank, world_size)) { fprintf(stderr, "rank %d: RCCL communicator not available, using marker files\n", world_rank); nc.done(); }
        }
        int bad = 0;
        Exchange ex{&rv, &nc};
        if (rc == ZWZ_OK) rc = zwz_decompress_dir_ranked(ctx, source_path.c_str(), output_path.c_str(), world_rank, world_size, world_size > 1 ? exchange_cb : nullptr, &ex, &bad);
    }
    if (trace) fprintf(stderr, "zwz: %s done at %.3f s\n", operation.c_str(), since());
    if (rc != ZWZ_OK && rc != 3) fprintf(stderr, "zwz: %s (%s)\n", zwz_strerror(rc), zwz_last_error());
    zwz_ctx_destroy(ctx);
    if (trace) fprintf(stderr, "zwz: context destroyed at %.3f s\n", since());

    int my_rc = rc == ZWZ_OK ? 0 : 2;
    if (nc.on) { my_rc = nc.worst_status(my_rc); if (trace) fprintf(stderr, "zwz: rank %d: job status %d over RCCL\n", world_rank, my_rc); }
    nc.done();
    const int job_rc = rv.finish(my_rc);   // MPI_Barrier (main.cpp:144); also removes the handshake's files
    if (world_rank == 0) {   // main.cpp:148-155
        const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        printf("========================================\nOperation: %s\nProcessor Count: %d\nTime Taken: %g seconds\n"
               "========================================\n", operation.c_str(), world_size, total);
    }
    return job_rc;
}
