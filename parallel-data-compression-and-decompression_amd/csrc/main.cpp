// main.cpp -- `main compress|decompress <src> <dst>`: the reference's command line (main.cpp:78-159)
// over the GPU pipeline.  Same argv, same output names (<dst>/compressed_<rank>.zwz, the side file
// <parent(src)>/sorted_files_by_size.txt), same final banner.  One process drives one GPU; rank and
// world size come from the launcher's environment (ZWZ_RANK/ZWZ_NRANKS, else RANK/WORLD_SIZE, else
// OMPI/PMI variables, else 0/1) where the reference asked MPI.  Ranks meet through marker files in
// <dst> (the reference likewise assumes a shared file system for the list, compression.cpp:25);
// the torch.distributed launcher (python -m ... cli) uses RCCL for the same two steps.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/zwz.h"

static int env_int(const char* const* names, int dflt) {
    for (; *names; names++) if (const char* v = getenv(*names)) return atoi(v);
    return dflt;
}

static bool wait_for(const std::string& path, double seconds) {
    auto t0 = std::chrono::steady_clock::now();
    struct stat sb;
    while (stat(path.c_str(), &sb) != 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    return true;
}

static void touch(const std::string& path) { if (FILE* f = fopen(path.c_str(), "w")) fclose(f); }

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
    if (world_rank == 0) {   // main.cpp:105-129
        struct stat st {};
        if (stat(source_path.c_str(), &st) != 0) { fprintf(stderr, "Source path does not exist.\n"); return 1; }
        if (stat(output_path.c_str(), &st) != 0) {
            if (mkdir(output_path.c_str(), 0777) == -1) { perror("Failed to create output directory"); return 1; }
        } else if (!S_ISDIR(st.st_mode)) { fprintf(stderr, "Output path is not a directory.\n"); return 1; }
    }

    int device_count = 0;
    zwz_device_count(&device_count);
    zwz_ctx* ctx = nullptr;
    const int dev = device_count > 0 ? env_int(dev_vars, world_rank) % device_count : 0;
    int rc = zwz_ctx_create(dev, 0, &ctx);
    const bool trace = getenv("ZWZ_VERBOSE") != nullptr;
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count(); };
    if (trace) fprintf(stderr, "zwz: context ready at %.3f s\n", since());
    if (rc != ZWZ_OK) { fprintf(stderr, "zwz: %s (%s)\n", zwz_strerror(rc), zwz_last_error()); return 2; }

    const std::string tag = output_path + "/.zwz_" + operation;
    if (operation == "compress") {
        char record[4096] = "";
        if (const char* fr = getenv("ZWZ_FILE_RECORD")) snprintf(record, sizeof record, "%s", fr);
        if (world_rank == 0) {
            printf("Compressing folder: %s\n", source_path.c_str());
            if (!record[0]) rc = zwz_sort_files_by_size(source_path.c_str(), record, sizeof record);
            printf("File record saved location: %s\n", record);
            if (trace) fprintf(stderr, "zwz: file list ready at %.3f s\n", since());
            if (world_size > 1) { FILE* f = fopen((tag + "_list").c_str(), "w"); if (f) { fputs(record, f); fclose(f); } touch(tag + "_list_ready"); }
        } else {   // the reference broadcasts the record path (main.cpp:24-39)
            if (!wait_for(tag + "_list_ready", 600)) { fprintf(stderr, "rank %d: no file list from rank 0\n", world_rank); return 3; }
            FILE* f = fopen((tag + "_list").c_str(), "r");
            if (!f || !fgets(record, sizeof record, f)) { fprintf(stderr, "rank %d: cannot read list path\n", world_rank); return 3; }
            fclose(f);
        }
        if (rc == ZWZ_OK) {
            printf("file_record: %s\n", record);
            if (world_rank < zwz_count_non_empty_lines(record)) rc = zwz_compress_dir(ctx, source_path.c_str(), output_path.c_str(), record, world_rank, world_size);
            else printf("Rank: %d - No file to compress\n", world_rank);
        }
    } else if (world_rank == 0) {   // main.cpp:61-68: decompression is a single-rank job
        if (world_size > 1) printf("Decompression is not supported in MPI parallel mode.\nOnly use one process to decompress.\n");
        int bad = 0;
        rc = zwz_decompress_dir(ctx, source_path.c_str(), output_path.c_str(), &bad);
    }
    if (trace) fprintf(stderr, "zwz: %s done at %.3f s\n", operation.c_str(), since());
    zwz_ctx_destroy(ctx);
    if (trace) fprintf(stderr, "zwz: context destroyed at %.3f s\n", since());
    if (rc != ZWZ_OK) fprintf(stderr, "zwz: %s (%s)\n", zwz_strerror(rc), zwz_last_error());

    if (world_size > 1) {   // MPI_Barrier (main.cpp:144)
        touch(tag + "_done_" + std::to_string(world_rank));
        if (world_rank == 0) {
            for (int r = 0; r < world_size; r++) wait_for(tag + "_done_" + std::to_string(r), 3600);
            for (int r = 0; r < world_size; r++) unlink((tag + "_done_" + std::to_string(r)).c_str());
            unlink((tag + "_list").c_str()); unlink((tag + "_list_ready").c_str());
        }
    }
    if (world_rank == 0) {   // main.cpp:148-155
        const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        printf("========================================\nOperation: %s\nProcessor Count: %d\nTime Taken: %g seconds\n"
               "========================================\n", operation.c_str(), world_size, total);
    }
    return rc == ZWZ_OK ? 0 : 2;
}
