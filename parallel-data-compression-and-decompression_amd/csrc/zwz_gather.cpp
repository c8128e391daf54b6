// zwz_gather.cpp -- "a final gather of per-shard .zwz blobs" (north star; the reference leaves every rank's shard where the rank wrote it,
// compression.cpp:151-170): ONE statement of the protocol for both launchers -- csrc/main.cpp drives it over RCCL (ncclSend / ncclRecv
// through a device staging buffer), cli.py over torch.distributed -- so that a fix to it lands once (rounds 3-4 fixed the bounded
// pieces and the failure flags twice).  No GPU call in here: the transport is the caller's.
//
//   sizes        all-gather of every rank's shard size (0: nothing to contribute)
//   readiness    every rank prepares its staging (hooks->prepare) and the verdicts are all-gathered: either every rank enters the
//                transfers or none does -- an unmatched send never completes and RCCL has no time-out
//   transfers    shard by shard, piece by piece (<= piece_bytes), exactly one send for one receive.  Two kinds of failure are kept
//                apart: the TRANSPORT failing ends the loops (nothing further can be matched); an I/O failure -- the sender
//                cannot read its shard, rank 0 cannot create or write <out_dir>/compressed_<r>.zwz -- marks that shard bad and the
//                pieces go on: rank 0 keeps receiving and discarding, the sender keeps sending, every later shard is still stored
//   verdict      all-gather of "every shard I touched is fine": only a unanimous yes returns 1, and only then may a sender delete its
//                private copy.  Rank 0 writes through <name>.part and renames.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/zwz.h"

extern "C" int zwz_gather_shards(int rank, int world, const char* my_shard_path, const char* out_dir, uint64_t piece_bytes, const zwz_gather_hooks* hk) {
    if (!hk || !hk->allgather_u64 || !hk->send || !hk->recv || !out_dir || rank < 0 || world < 1 || rank >= world) return 0;
    if (piece_bytes == 0) piece_bytes = (uint64_t)64 << 20;
    const std::string my_path = my_shard_path ? my_shard_path : "";
    bool comm_ok = true, io_failed = false;
    FILE* f = rank != 0 && !my_path.empty() ? fopen(my_path.c_str(), "rb") : nullptr;
    uint64_t my_size = 0;
    if (f) { struct stat st {}; if (fstat(fileno(f), &st) == 0) my_size = (uint64_t)st.st_size; else io_failed = true; }
    else if (rank != 0 && !my_path.empty()) io_failed = true;      // a shard this rank says it has, and cannot open: never to be taken for transferred
    if (io_failed) { my_size = 0; fprintf(stderr, "rank %d: cannot read its shard %s\n", rank, my_path.c_str()); }
    std::vector<uint64_t> sizes((size_t)world);
    comm_ok = hk->allgather_u64(hk->user, &my_size, sizes.data(), 1) == 0;
    uint64_t biggest = 0;
    for (int r = 1; r < world; r++) biggest = std::max(biggest, sizes[(size_t)r]);
    const uint64_t piece = std::min<uint64_t>(biggest, piece_bytes);
    std::vector<uint8_t> buf;
    const bool need = comm_ok && piece && (rank == 0 || my_size);
    uint64_t ready = 0;
    if (comm_ok) {
        ready = 1;
        if (need) {
            try { buf.resize((size_t)piece); } catch (const std::bad_alloc&) { ready = 0; }
            if (ready && hk->prepare && hk->prepare(hk->user, piece) != 0) ready = 0;
        }
    }
    std::vector<uint64_t> all_ready((size_t)world, 0);
    if (comm_ok) comm_ok = hk->allgather_u64(hk->user, &ready, all_ready.data(), 1) == 0;      // (a rank whose first all-gather failed cannot be helped: the transport is gone)
    bool transfer = comm_ok && piece != 0;
    for (int r = 0; r < world && comm_ok; r++)
        if (!all_ready[(size_t)r]) { transfer = false; io_failed = true; if (rank == 0) fprintf(stderr, "rank 0: rank %d has no staging memory for the shard gather\n", r); }
    if (transfer) {
        for (int r = 1; r < world && comm_ok; r++) {
            const uint64_t total = sizes[(size_t)r];
            if (!total || (rank != 0 && rank != r)) continue;
            FILE* out = nullptr;
            const std::string final_path = std::string(out_dir) + "/compressed_" + std::to_string(r) + ".zwz", tmp_path = final_path + ".part";
            if (rank == 0 && !(out = fopen(tmp_path.c_str(), "wb"))) fprintf(stderr, "rank 0: cannot create %s\n", tmp_path.c_str());
            bool shard_ok = rank != 0 || out != nullptr;
            for (uint64_t off = 0; off < total && comm_ok; off += piece) {
                const size_t k = (size_t)std::min<uint64_t>(piece, total - off);
                if (rank == r) {
                    if (shard_ok && fread(buf.data(), 1, k, f) != k) { shard_ok = false; memset(buf.data(), 0, k); }
                    comm_ok = hk->send(hk->user, buf.data(), k, 0) == 0;             // (sent whatever the read did: rank 0 is waiting for this piece)
                } else {
                    comm_ok = hk->recv(hk->user, buf.data(), k, r) == 0;
                    if (comm_ok && shard_ok && fwrite(buf.data(), 1, k, out) != k) shard_ok = false;      // (a full disk: this shard is lost, the next ones are still received)
                }
            }
            if (out) { if (fclose(out) != 0) shard_ok = false; if (comm_ok && shard_ok) shard_ok = rename(tmp_path.c_str(), final_path.c_str()) == 0; else unlink(tmp_path.c_str()); }
            if (!shard_ok) { fprintf(stderr, "rank %d: I/O error while %s shard %d\n", rank, rank == 0 ? "writing" : "reading", r); io_failed = true; }
        }
    }
    if (f) fclose(f);
    if (hk->release) hk->release(hk->user);
    // every rank learns whether every transfer and every write succeeded (a sender must not delete a shard rank 0 could not store).
    // A rank-local read failure is the sender's own flag; rank 0's write failure is rank 0's: the all-gather carries both to everybody.
    if (!comm_ok) return 0;
    uint64_t fine = io_failed ? 0 : 1;
    std::vector<uint64_t> all_fine((size_t)world, 0);
    if (hk->allgather_u64(hk->user, &fine, all_fine.data(), 1) != 0) return 0;
    for (int r = 0; r < world; r++) if (!all_fine[(size_t)r]) return 0;
    return 1;
}
