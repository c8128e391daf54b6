// zwz_api_internal.h -- context object behind include/zwz.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/zwz.h"
#include "zwz_kernels.h"

struct zwz_ctx {
    int device = 0;
    uint32_t max_batch = 0;
    hipStream_t stream = nullptr;
    void* workspace = nullptr;
    uint32_t ws_chunks = 0;          // chunks the workspace is currently sized for
    uint4* inf_order = nullptr;      // inflate's launch order ((offset, length, chunk) by payload length), grown to the largest batch seen
    uint32_t inf_order_cap = 0;
    // staging for the host-buffer entry points and the directory pipeline
    void* d_stage = nullptr;
    void* h_stage = nullptr;
    uint32_t stage_chunks = 0;
    uint32_t cu_count = 0;
    uint32_t chunk_bytes = 0;        // raw bytes per Chunk for zwz_compress_dir; 0 = default (see chunk_bytes_for)
    // Switches (zwz_ctx_set_option; defaults from ZWZ_MATCH / ZWZ_PLAN / ZWZ_INFLATE_HEADER read ONCE at zwz_ctx_create, or forced by a
    // failed self-test there).  Every setting produces the same bytes; they differ in which kernels run.
    uint32_t match_mode = 0;             // zwz::kMatchAuto | kMatchWalk | kMatchBand
    uint32_t plan_serial = 0;            // 1: lane-serial block flush
    uint32_t inflate_serial_header = 0;  // 1: block headers and tables on lane 0
    // Kernel forms this device failed a self-test of at zwz_ctx_create (kForbid*): zwz_ctx_set_option refuses to switch them back on
    uint32_t forbidden = 0;
    bool profiling = false;
    hipEvent_t ev[zwz::kNumDeflateStages + 1] = {};
    hipEvent_t ev_inf[2] = {};
    float stage_ms[ZWZ_NUM_STAGES] = {};
};

namespace zwz {

enum : uint32_t { kForbidLinks = 1u, kForbidSort = 2u, kForbidPlanWave = 4u, kForbidInflateWave = 8u };   // zwz_ctx::forbidden

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
void carve_workspace(zwz_ctx* c, DeflateArgs& a);
int ensure_staging(zwz_ctx* c, uint32_t chunks);

// One staging slice: input slots, output slots, offsets/lengths/status; same layout on host and device.
struct StageView {
    uint8_t *h_in, *h_out; uint64_t* h_off; uint32_t *h_len, *h_olen, *h_status;
    uint8_t *d_in, *d_out; uint64_t* d_off; uint32_t *d_len, *d_olen, *d_status;
};

inline size_t stage_bytes(uint32_t m) {
    return 2 * (size_t)m * ZWZ_DEV_STRIDE + (size_t)m * (sizeof(uint64_t) + 3 * sizeof(uint32_t)) + 1024;
}

inline StageView stage_view(zwz_ctx* c, uint32_t /*m*/) {
    const size_t cap = c->stage_chunks;
    StageView v;
    auto carve = [&](uint8_t* base, uint8_t*& in, uint8_t*& out, uint64_t*& off, uint32_t*& len, uint32_t*& olen, uint32_t*& st) {
        in = base; out = in + cap * ZWZ_DEV_STRIDE;
        off = reinterpret_cast<uint64_t*>(out + cap * ZWZ_DEV_STRIDE);
        len = reinterpret_cast<uint32_t*>(off + cap); olen = len + cap; st = olen + cap;
    };
    carve(static_cast<uint8_t*>(c->h_stage), v.h_in, v.h_out, v.h_off, v.h_len, v.h_olen, v.h_status);
    carve(static_cast<uint8_t*>(c->d_stage), v.d_in, v.d_out, v.d_off, v.d_len, v.d_olen, v.d_status);
    return v;
}

}  // namespace zwz
