// zwz_api.cpp -- C ABI (include/zwz.h) over the kernel pipeline: context, workspace, batch slicing,
// pinned staging for host buffers.  No CPU codec lives here: without a GPU every entry point fails.
#include "zwz_api_internal.h"

#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace zwz {

thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ZWZ_E_HIP;
}

}  // namespace zwz

using namespace zwz;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

namespace {

// Known-answer test of lz_links, run once per context.  The kernel's feeders keep input loads in flight across several
// hand-overs in registers the compiler merely promises not to touch, and its inserter is one block of hand-scheduled
// assembly: a toolchain change could break either without a diagnostic, and wrong links mean shards that are no longer
// the reference's.  Three chunks -- text-like with long chains, incompressible, a ragged short one -- go through the
// kernel and every link is compared with zlib's insert restated on the host (prev[p] = head[h]; head[h] = p).
// Returns ZWZ_OK with *bad = 0, or bit 0 set (lz_links' links are wrong) / bit 1 set (lz_sort's order is wrong) and the text of the
// first finding in zwz_last_error(); a HIP failure is returned as such.
int links_self_test(zwz_ctx* c, uint32_t* bad) {
    *bad = 0;
    // three workgroups for eight chunks: each one runs on from chunk to chunk (full, short, one-block, empty and tiny ones)
    constexpr uint32_t K = 8;
    const uint32_t lens[K] = {65535u, 40000u, 0u, 2049u + 37u, 5u, 6144u, 65535u, 2u};
    std::vector<uint8_t> h_in((size_t)K * ZWZ_DEV_STRIDE, 0);
    uint32_t rng = 0x2545f491u;
    uint64_t offs[K];
    for (uint32_t k = 0; k < K; k++) {
        uint8_t* d = h_in.data() + (size_t)k * ZWZ_DEV_STRIDE;
        offs[k] = (uint64_t)k * ZWZ_DEV_STRIDE;
        for (uint32_t i = 0; i < lens[k]; i++) {
            rng = rng * 1664525u + 1013904223u;
            if (k & 1u) d[i] = (uint8_t)(rng >> 24);                                   // incompressible
            else d[i] = (rng >> 28) < 3 && i >= 7 ? d[i - 7 + ((rng >> 20) & 3u)] : (uint8_t)(97u + ((rng >> 16) % 6u));   // six letters and short copies: long chains
        }
    }
    uint8_t* d_in = nullptr; uint64_t* d_off = nullptr; uint32_t* d_len = nullptr; uint16_t* d_links = nullptr; uint32_t* d_stat = nullptr;
    uint32_t *d_sorted = nullptr, *d_list = nullptr, *d_tickets = nullptr;
    auto cleanup = [&] { (void)hipFree(d_in); (void)hipFree(d_off); (void)hipFree(d_len); (void)hipFree(d_links); (void)hipFree(d_stat);
                         (void)hipFree(d_sorted); (void)hipFree(d_list); (void)hipFree(d_tickets); };
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), h_in.size());
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_off), sizeof offs);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_len), sizeof lens);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_links), K * (size_t)kLinkStride * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_stat), K * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_sorted), K * (size_t)kSortedStride * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_list), K * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_tickets), kTicketBytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_tickets, 0, kTicketBytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_sorted, 0xa5, K * (size_t)kSortedStride * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, h_in.data(), h_in.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offs, sizeof offs, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_len, lens, sizeof lens, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_links, 0xa5, K * (size_t)kLinkStride * sizeof(uint16_t), c->stream);
    DeflateArgs a{};
    a.in = d_in; a.in_off = d_off; a.in_len = d_len; a.n = K; a.links = d_links; a.link_stat = d_stat; a.cu_count = 3;
    a.sorted = d_sorted; a.dense_list = d_list; a.sparse_list = d_list; a.tickets = d_tickets;   // (every chunk goes onto the dense list here)
    if (e == hipSuccess) e = launch_links_only(a, c->stream);
    std::vector<uint16_t> got(K * (size_t)kLinkStride);
    uint32_t stat[K] = {};
    if (e == hipSuccess) e = hipMemcpyAsync(got.data(), d_links, got.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(stat, d_stat, sizeof stat, hipMemcpyDeviceToHost, c->stream);
    // lz_sort over the same chunks (it overwrites their link arrays, read back above): its ranking pass stands on the lane
    // order of the returning LDS add
    std::vector<uint32_t> srt(K * (size_t)kSortedStride);
    if (e == hipSuccess) e = launch_dense_list(a, c->stream, 2u);
    if (e == hipSuccess) e = launch_sort(a, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(srt.data(), d_sorted, srt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return hip_fail(e, "lz_links / lz_sort self-test");
    std::vector<uint16_t> head(32768);
    for (uint32_t k = 0; k < K; k++) {
        const uint8_t* d = h_in.data() + (size_t)k * ZWZ_DEV_STRIDE;
        std::fill(head.begin(), head.end(), (uint16_t)0);
        uint32_t linked = 0;
        for (uint32_t p = 0; p + kMinMatch <= lens[k]; p++) {
            const uint32_t h = hash3(d[p], d[p + 1], d[p + 2]);
            const uint16_t want = head[h];
            head[h] = (uint16_t)p;
            linked += want != 0;
            if (got[(size_t)k * kLinkStride + p] != want) {
                if (!(*bad & 1u)) set_error("zwz_ctx_create: lz_links self-test failed (chunk %u, position %u: link %u, expected %u) -- the kernel's hand-scheduled "
                                            "code does not survive this toolchain", k, p, (unsigned)got[(size_t)k * kLinkStride + p], (unsigned)want);
                *bad |= 1u;
                break;
            }
        }
        for (uint32_t p = lens[k] >= kMinMatch ? lens[k] - (kMinMatch - 1u) : 0u; p < lens[k]; p++) {   // the last positions have no trigram: NIL
            if (got[(size_t)k * kLinkStride + p] != 0 && !(*bad & 1u)) { set_error("zwz_ctx_create: lz_links self-test failed (chunk %u: a link behind the last trigram)", k); *bad |= 1u; }
        }
        if (stat[k] != linked && !(*bad & 1u)) { set_error("zwz_ctx_create: lz_links self-test failed (chunk %u: %u linked positions counted, %u expected)", k, stat[k], linked); *bad |= 1u; }
        // the sorted array: a stable counting sort of the positions by bucket
        const uint32_t n = lens[k] >= kMinMatch ? lens[k] - (kMinMatch - 1u) : 0u;
        std::vector<uint32_t> start(32769, 0);
        for (uint32_t p = 0; p < n; p++) start[hash3(d[p], d[p + 1], d[p + 2]) + 1]++;
        for (uint32_t h = 0; h < 32768; h++) start[h + 1] += start[h];
        for (uint32_t p = 0; p < n; p++) {
            const uint32_t h = hash3(d[p], d[p + 1], d[p + 2]), want = start[h]++;
            const uint32_t got_d = reinterpret_cast<const uint16_t*>(srt.data() + (size_t)k * kSortedStride)[p];
            if (got_d != want) {
                if (!(*bad & 2u)) set_error("zwz_ctx_create: lz_sort self-test failed (chunk %u, position %u: sorted index %u, expected %u) -- the returning LDS add does not "
                                            "serve same-address lanes in lane order on this device", k, p, got_d, want);
                *bad |= 2u;
                break;
            }
        }
    }
    return ZWZ_OK;
}

// Known-answer test of the two other users of the ordered LDS add: the wave form of the block flush (zwz_plan.hip:
// next_code[len]++ in symbol order) and inflate's wave-built decoding tables (wave_build_decode_table: a symbol's place in sorted[] and
// hence its code).  Three small chunks with dynamic-Huffman blocks -- text-like, a geometric byte distribution whose tree is deep, one
// with long matches and far distances -- are deflated with the serial and the wave flush and inflated with serial and wave headers:
// the serial forms must reproduce the input (else the device cannot run the codec at all), and a wave form that disagrees with its
// serial form is switched off for this context.  A few milliseconds per context.
int codec_self_test(zwz_ctx* c) {
    constexpr uint32_t K = 3;
    const uint32_t lens[K] = {24000u, 30000u, 40000u};
    std::vector<uint8_t> in;
    uint64_t offs[K];
    uint32_t rng = 0x9e3779b9u;
    auto next = [&] { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (uint32_t k = 0; k < K; k++) {
        offs[k] = in.size();
        const size_t base = in.size();
        for (uint32_t i = 0; i < lens[k]; i++) {
            uint8_t b;
            if (k == 0) b = (next() & 15u) < 3 && i >= 9 ? in[base + i - 9 + (next() & 3u)] : (uint8_t)(97u + next() % 7u);
            else if (k == 1) { uint32_t r = next() | 1u << 23, z = 0; while (!(r & 1u)) { r >>= 1; z++; } b = (uint8_t)(32u + z); }   // P(32 + z) = 2^-(z + 1): a tree as deep as its alphabet
            else b = i >= 20000u && ((i / 37u) & 1u) ? in[base + i - 20000u] : (uint8_t)next();                                       // far copies in noise: long distance codes
            in.push_back(b);
        }
    }
    const uint32_t keep_plan = c->plan_serial, keep_hdr = c->inflate_serial_header;
    std::vector<uint8_t> p_serial((size_t)K * ZWZ_CHUNK_SIZE), p_wave((size_t)K * ZWZ_CHUNK_SIZE), back((size_t)K * ZWZ_CHUNK_SIZE);
    uint32_t l_serial[K], l_wave[K], l_back[K], st[K];
    uint64_t poffs[K];
    for (uint32_t k = 0; k < K; k++) poffs[k] = (uint64_t)k * ZWZ_CHUNK_SIZE;
    auto same_as_input = [&] {
        for (uint32_t k = 0; k < K; k++) if (st[k] != ZWZ_INF_END || l_back[k] != lens[k] || memcmp(back.data() + (size_t)k * ZWZ_CHUNK_SIZE, in.data() + offs[k], lens[k])) return false;
        return true;
    };
    c->plan_serial = 1;
    int rc = zwz_deflate_batch(c, in.data(), offs, lens, K, p_serial.data(), l_serial);
    c->inflate_serial_header = 1;
    if (rc == ZWZ_OK) rc = zwz_inflate_batch(c, p_serial.data(), poffs, l_serial, K, back.data(), l_back, st);
    if (rc != ZWZ_OK) return rc;
    if (!same_as_input()) { set_error("zwz_ctx_create: codec self-test failed -- three chunks do not survive deflate + inflate on this device even in the kernels' order-free forms"); return ZWZ_E_NO_DEVICE; }
    if (!keep_plan) {
        c->plan_serial = 0;
        rc = zwz_deflate_batch(c, in.data(), offs, lens, K, p_wave.data(), l_wave);
        if (rc != ZWZ_OK) return rc;
        bool same = true;
        for (uint32_t k = 0; k < K; k++) same = same && l_wave[k] == l_serial[k] && !memcmp(p_wave.data() + (size_t)k * ZWZ_CHUNK_SIZE, p_serial.data() + (size_t)k * ZWZ_CHUNK_SIZE, l_serial[k]);
        if (!same) { c->plan_serial = 1; c->forbidden |= kForbidPlanWave; fprintf(stderr, "zwz: the wave form of the block flush disagrees with the lane-serial one on this device: using the lane-serial kernel\n"); }
    }
    if (!keep_hdr) {
        c->inflate_serial_header = 0;
        rc = zwz_inflate_batch(c, p_serial.data(), poffs, l_serial, K, back.data(), l_back, st);
        if (rc != ZWZ_OK) return rc;
        if (!same_as_input()) { c->inflate_serial_header = 1; c->forbidden |= kForbidInflateWave; fprintf(stderr, "zwz: inflate's wave-built tables decode wrongly on this device: block headers go to lane 0\n"); }
    }
    return ZWZ_OK;
}

}  // namespace

extern "C" {

const char* zwz_strerror(int s) {
    switch (s) {
        case ZWZ_OK: return "ok";
        case ZWZ_E_INVALID: return "invalid argument";
        case ZWZ_E_HIP: return "HIP runtime error";
        case ZWZ_E_NO_DEVICE: return "no usable GPU (this library has no CPU fallback)";
        case ZWZ_E_IO: return "I/O error";
        case ZWZ_E_NOMEM: return "out of memory";
        case ZWZ_E_FORMAT: return "malformed .zwz shard";
        default: return "unknown status";
    }
}

const char* zwz_last_error(void) { return g_err; }

int zwz_device_count(int* count) {
    if (!count) return ZWZ_E_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return ZWZ_E_NO_DEVICE; }
    *count = n;
    return ZWZ_OK;
}

int zwz_ctx_create(int device, uint32_t max_batch, zwz_ctx** out) {
    if (!out) return ZWZ_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (zwz_device_count(&n) != ZWZ_OK || n <= 0) { set_error("hipGetDeviceCount found no device"); return ZWZ_E_NO_DEVICE; }
    if (device < 0 || device >= n) return ZWZ_E_INVALID;
    HIPCHK(hipSetDevice(device));
    zwz_ctx* c = new (std::nothrow) zwz_ctx();
    if (!c) return ZWZ_E_NOMEM;
    c->device = device;
    c->max_batch = max_batch ? max_batch : 8192u;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cu_count = (uint32_t)cus; }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = configure_kernels();
    for (int i = 0; i < kNumDeflateStages + 1 && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreate(&c->ev_inf[i]);
    if (e != hipSuccess) { int rc = hip_fail(e, "zwz_ctx_create"); zwz_ctx_destroy(c); return rc; }
    if (const uint32_t xk = exp_flags_kernels(), xb = exp_flags_band(); xk | xb)     // never silent: an instrumented build must not pass for the product
        fprintf(stderr, "zwz: EXPERIMENT BUILD (ZWZ_MATCH_EXP=%u ZWZ_PARSE_EXP=%u ZWZ_ENC_EXP=%u ZWZ_INF_EXP=%u ZWZ_BAND_EXP=%u): timings only, not the product library\n",
                xk & 255u, (xk >> 8) & 255u, (xk >> 16) & 255u, xk >> 24, xb);
    // Test switches, read ONCE per context (a stray variable is still honoured, but no launch re-reads the environment: ADVICE r3)
    // (a value that is not understood is said so: ZWZ_MATCH=bnad used to be ignored without a word)
    auto from_env = [&](const char* var, const char* option) {
        if (const char* v = getenv(var)) if (zwz_ctx_set_option(c, option, v) != ZWZ_OK) fprintf(stderr, "zwz: %s=%s is not a value of option \"%s\": ignored\n", var, v, option);
    };
    from_env("ZWZ_MATCH", "match"); from_env("ZWZ_PLAN", "plan"); from_env("ZWZ_INFLATE_HEADER", "inflate_header");
    const bool tl = getenv("ZWZ_TIMELINE") != nullptr || getenv("ZWZ_VERBOSE") != nullptr;
    const auto t_create = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) { if (tl) fprintf(stderr, "zwz: context: %s at +%.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create).count()); };
    // Three kernels stand on LDS behaviour the ISA manual does not promise: lz_links on the lane order of
    // ds_wrxchg_rtn, lz_sort / the wave plan / inflate's wave-built tables on that of ds_add_rtn.  Each is checked here against a
    // host restatement or its own order-free form; a failed check selects the form that does not need the property, and only a
    // device on which no search kernel is left gets no context.
    {
        bool xchg_ok = false;
        e = probe_exchange_order(c->stream, &xchg_ok);
        if (e != hipSuccess) { int rc = hip_fail(e, "zwz_ctx_create: exchange-order probe"); zwz_ctx_destroy(c); return rc; }
        mark("exchange-order probe done");
        uint32_t bad = 0;
        const int rc = links_self_test(c, &bad);
        mark("lz_links / lz_sort known-answer tests done");
        if (rc != ZWZ_OK) { zwz_ctx_destroy(c); return rc; }
        const bool links_ok = xchg_ok && !(bad & 1u), sort_ok = !(bad & 2u);
        if (!links_ok && !sort_ok) {
            if (bad == 2u || !bad) set_error("zwz_ctx_create: neither ds_wrxchg_rtn_b32 nor ds_add_rtn_u32 serves same-address lanes in lane order on this device: no match search can run on it");
            zwz_ctx_destroy(c);
            return ZWZ_E_NO_DEVICE;
        }
        // (a test hook: ZWZ_FORCE_SELFTEST_FAIL=links|sort pretends that check failed, so that the fallback configurations run through the
        //  oracle comparison once -- tests/test_gpu_codec.py)
        const char* force = getenv("ZWZ_FORCE_SELFTEST_FAIL");
        const bool f_links = force && strstr(force, "links"), f_sort = force && strstr(force, "sort");
        if (!links_ok || f_links) c->forbidden |= kForbidLinks;
        if (!sort_ok || f_sort) c->forbidden |= kForbidSort | kForbidPlanWave | kForbidInflateWave;
        if (f_links && !f_sort) { c->match_mode = kMatchBand; fprintf(stderr, "zwz: (forced) lz_links counts as failed: every chunk takes the sort + band search\n"); }
        if (f_sort) { c->match_mode = kMatchWalk; c->plan_serial = 1; c->inflate_serial_header = 1; fprintf(stderr, "zwz: (forced) lz_sort counts as failed: chain walk, lane-serial block flush and inflate headers\n"); }
        if (!links_ok) { c->match_mode = kMatchBand; fprintf(stderr, "zwz: lz_links' exchange order does not hold on this device: every chunk takes the sort + band search (%s)\n", zwz_last_error()); }
        if (!sort_ok) {
            c->match_mode = kMatchWalk; c->plan_serial = 1; c->inflate_serial_header = 1;
            fprintf(stderr, "zwz: the returning LDS add is not served in lane order on this device: chain walk, lane-serial block flush and lane-serial inflate headers (%s)\n", zwz_last_error());
        }
    }
    { const int rc = codec_self_test(c); if (rc != ZWZ_OK) { zwz_ctx_destroy(c); return rc; } }
    mark("codec known-answer test done");
    *out = c;
    return ZWZ_OK;
}

int zwz_ctx_set_option(zwz_ctx* c, const char* name, const char* value) {
    if (!c || !name || !value) return ZWZ_E_INVALID;
    const std::string n = name, v = value;
    // A form this device failed its self-test of stays off (include/zwz.h promises the same bytes from every setting, which a
    // failing kernel form would break): asking for it is ZWZ_E_NO_DEVICE, and "auto" / "wave" keep meaning "what this device can run".
    const bool no_links = c->forbidden & kForbidLinks, no_sort = c->forbidden & kForbidSort;
    auto refuse = [&](const char* what) { set_error("zwz_ctx_set_option: %s=%s needs a kernel form that failed its self-test on this device", what, v.c_str()); return ZWZ_E_NO_DEVICE; };
    if (n == "match") {
        uint32_t m;
        if (v == "auto" || v.empty()) m = kMatchAuto; else if (v == "walk") m = kMatchWalk; else if (v == "band") m = kMatchBand; else if (v == "lazy") m = kMatchLazy;
        else if (v == "autoband") m = kMatchAutoBand; else if (v == "autolazy") m = kMatchAutoLazy; else return ZWZ_E_INVALID;
        const bool needs_links = m == kMatchAuto || m == kMatchWalk || m == kMatchAutoBand || m == kMatchAutoLazy;
        const bool needs_sort = m == kMatchAuto || m == kMatchBand || m == kMatchLazy || m == kMatchAutoBand || m == kMatchAutoLazy;
        if (m == kMatchAuto && (no_links || no_sort)) m = no_sort ? kMatchWalk : kMatchBand;          // "auto" on a device with one search left: that one
        else if ((needs_links && no_links) || (needs_sort && no_sort)) return refuse("match");
        c->match_mode = m;
    } else if (n == "plan") {
        if (v == "wave" || v.empty()) { if (c->forbidden & kForbidPlanWave) { if (v.empty()) return ZWZ_OK; return refuse("plan"); } c->plan_serial = 0; }
        else if (v == "serial") c->plan_serial = 1; else return ZWZ_E_INVALID;
    } else if (n == "inflate_header") {
        if (v == "wave" || v.empty()) { if (c->forbidden & kForbidInflateWave) { if (v.empty()) return ZWZ_OK; return refuse("inflate_header"); } c->inflate_serial_header = 0; }
        else if (v == "serial") c->inflate_serial_header = 1; else return ZWZ_E_INVALID;
    } else return ZWZ_E_INVALID;
    return ZWZ_OK;
}

void zwz_ctx_destroy(zwz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->workspace) (void)hipFree(c->workspace);
    if (c->inf_order) (void)hipFree(c->inf_order);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_inf) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* zwz_ctx_stream(zwz_ctx* c) { return c ? (void*)c->stream : nullptr; }

int zwz_ctx_sync(zwz_ctx* c) {
    if (!c) return ZWZ_E_INVALID;
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZWZ_OK;
}

int zwz_ctx_set_profiling(zwz_ctx* c, int on) {
    if (!c) return ZWZ_E_INVALID;
    c->profiling = on != 0;
    return ZWZ_OK;
}

int zwz_ctx_stage_ms(zwz_ctx* c, float* ms, int reset) {
    if (!c || !ms) return ZWZ_E_INVALID;
    for (int i = 0; i < ZWZ_NUM_STAGES; i++) ms[i] = c->stage_ms[i];
    if (reset) for (auto& v : c->stage_ms) v = 0.f;
    return ZWZ_OK;
}

int zwz_deflate_batch_dev(zwz_ctx* c, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                          uint8_t* d_out, uint64_t out_stride, uint32_t* d_out_len) {
    if (!c || (n && (!d_in || !d_in_off || !d_in_len || !d_out || !d_out_len))) return ZWZ_E_INVALID;
    if (out_stride < ZWZ_DEV_STRIDE || (out_stride & 15u) || ((uintptr_t)d_in & 15u) || ((uintptr_t)d_out & 15u)) return ZWZ_E_INVALID;
    HIPCHK(hipSetDevice(c->device));
    {   // the workspace grows to the largest slice seen (never past max_batch): small jobs and pure
        // decompression never pay for the full ~5.6 GB
        const uint32_t want = n < c->max_batch ? n : c->max_batch;
        if (want > c->ws_chunks) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->workspace) { (void)hipFree(c->workspace); c->workspace = nullptr; c->ws_chunks = 0; }
            HIPCHK(hipMalloc(&c->workspace, (size_t)want * kWorkspaceBytesPerChunk + kTicketBytes + 8192));
            c->ws_chunks = want;
        }
    }
    for (uint32_t done = 0; done < n; done += c->max_batch) {
        const uint32_t m = n - done < c->max_batch ? n - done : c->max_batch;
        DeflateArgs a;
        a.in = d_in; a.in_off = d_in_off + done; a.in_len = d_in_len + done; a.n = m;
        a.out = d_out + (size_t)done * out_stride; a.out_stride = out_stride; a.out_len = d_out_len + done;
        carve_workspace(c, a);
        a.match_mode = c->match_mode; a.plan_serial = c->plan_serial;
        HIPCHK(launch_deflate(a, c->stream, c->profiling ? c->ev : nullptr));
        if (c->profiling) {   // profiling serialises slices: stage times are read back per slice
            HIPCHK(hipEventSynchronize(c->ev[kNumDeflateStages]));
            for (int i = 0; i < kNumDeflateStages; i++) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
                c->stage_ms[i] += ms;
            }
        }
    }
    return ZWZ_OK;
}

int zwz_md5_files_dev(zwz_ctx* c, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, const uint32_t* d_files,
                      uint32_t n_files, uint8_t* d_digests) {
    if (!c || (n_files && (!d_in || !d_in_off || !d_in_len || !d_files || !d_digests))) return ZWZ_E_INVALID;
    if ((uintptr_t)d_digests & 3u) return ZWZ_E_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_md5_files(d_in, d_in_off, d_in_len, d_files, n_files, reinterpret_cast<uint32_t*>(d_digests), c->stream));
    return ZWZ_OK;
}

int zwz_inflate_batch_dev(zwz_ctx* c, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                          uint8_t* d_out, uint64_t out_stride, uint32_t* d_out_len, uint32_t* d_status) {
    if (!c || (n && (!d_in || !d_in_off || !d_in_len || !d_out || !d_out_len || !d_status))) return ZWZ_E_INVALID;
    if (out_stride < ZWZ_CHUNK_SIZE || ((uintptr_t)d_in & 15u)) return ZWZ_E_INVALID;
    HIPCHK(hipSetDevice(c->device));
    if (n > c->inf_order_cap) {     // scratch for the launch order (longest payloads first)
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->inf_order) { (void)hipFree(c->inf_order); c->inf_order = nullptr; c->inf_order_cap = 0; }
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->inf_order), (size_t)n * sizeof(uint4)));
        c->inf_order_cap = n;
    }
    InflateArgs a{d_in, d_in_off, d_in_len, n, d_out, out_stride, d_out_len, d_status, c->inf_order, c->inflate_serial_header};
    if (c->profiling) HIPCHK(hipEventRecord(c->ev_inf[0], c->stream));
    HIPCHK(launch_inflate(a, c->stream));
    if (c->profiling) {
        HIPCHK(hipEventRecord(c->ev_inf[1], c->stream));
        HIPCHK(hipEventSynchronize(c->ev_inf[1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_inf[0], c->ev_inf[1]));
        c->stage_ms[kNumDeflateStages] += ms;
    }
    return ZWZ_OK;
}

// ---- host-buffer variants: pack chunks into 65536-byte slots of a pinned buffer, one H2D, run,
// one D2H, unpack at the reference's 65535-byte stride.
int zwz_deflate_batch(zwz_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n, uint8_t* out,
                      uint32_t* out_len) {
    if (!c || (n && (!in || !in_off || !in_len || !out || !out_len))) return ZWZ_E_INVALID;
    for (uint32_t i = 0; i < n; i++) if (in_len[i] > ZWZ_CHUNK_SIZE) return ZWZ_E_INVALID;
    HIPCHK(hipSetDevice(c->device));
    const uint32_t slice = c->max_batch;
    for (uint32_t done = 0; done < n; done += slice) {
        const uint32_t m = n - done < slice ? n - done : slice;
        int rc = ensure_staging(c, m);
        if (rc) return rc;
        StageView v = stage_view(c, m);
        for (uint32_t i = 0; i < m; i++) {
            memcpy(v.h_in + (size_t)i * ZWZ_DEV_STRIDE, in + in_off[done + i], in_len[done + i]);
            v.h_off[i] = (uint64_t)i * ZWZ_DEV_STRIDE;
            v.h_len[i] = in_len[done + i];
        }
        HIPCHK(hipMemcpyAsync(v.d_in, v.h_in, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(v.d_off, v.h_off, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(v.d_len, v.h_len, m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        rc = zwz_deflate_batch_dev(c, v.d_in, v.d_off, v.d_len, m, v.d_out, ZWZ_DEV_STRIDE, v.d_olen);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(v.h_out, v.d_out, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(v.h_olen, v.d_olen, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < m; i++) {
            out_len[done + i] = v.h_olen[i];
            memcpy(out + (size_t)(done + i) * ZWZ_CHUNK_SIZE, v.h_out + (size_t)i * ZWZ_DEV_STRIDE, v.h_olen[i]);
        }
    }
    return ZWZ_OK;
}

int zwz_inflate_batch(zwz_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n, uint8_t* out,
                      uint32_t* out_len, uint32_t* status) {
    if (!c || (n && (!in || !in_off || !in_len || !out || !out_len || !status))) return ZWZ_E_INVALID;
    for (uint32_t i = 0; i < n; i++) if (in_len[i] > ZWZ_CHUNK_SIZE) return ZWZ_E_INVALID;
    HIPCHK(hipSetDevice(c->device));
    const uint32_t slice = c->max_batch;
    for (uint32_t done = 0; done < n; done += slice) {
        const uint32_t m = n - done < slice ? n - done : slice;
        int rc = ensure_staging(c, m);
        if (rc) return rc;
        StageView v = stage_view(c, m);
        for (uint32_t i = 0; i < m; i++) {
            memcpy(v.h_in + (size_t)i * ZWZ_DEV_STRIDE, in + in_off[done + i], in_len[done + i]);
            v.h_off[i] = (uint64_t)i * ZWZ_DEV_STRIDE;
            v.h_len[i] = in_len[done + i];
        }
        HIPCHK(hipMemcpyAsync(v.d_in, v.h_in, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(v.d_off, v.h_off, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(v.d_len, v.h_len, m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        rc = zwz_inflate_batch_dev(c, v.d_in, v.d_off, v.d_len, m, v.d_out, ZWZ_DEV_STRIDE, v.d_olen, v.d_status);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(v.h_out, v.d_out, (size_t)m * ZWZ_DEV_STRIDE, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(v.h_olen, v.d_olen, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(v.h_status, v.d_status, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < m; i++) {
            out_len[done + i] = v.h_olen[i];
            status[done + i] = v.h_status[i];
            memcpy(out + (size_t)(done + i) * ZWZ_CHUNK_SIZE, v.h_out + (size_t)i * ZWZ_DEV_STRIDE, v.h_olen[i]);
        }
    }
    return ZWZ_OK;
}

}  // extern "C"

namespace zwz {

void carve_workspace(zwz_ctx* c, DeflateArgs& a) {
    uint8_t* p = static_cast<uint8_t*>(c->workspace);
    const size_t n = c->ws_chunks;
    auto take = [&](size_t bytes) { uint8_t* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    a.entries = reinterpret_cast<uint2*>(take(n * kEntryStride * sizeof(uint2)));
    a.links = reinterpret_cast<uint16_t*>(take(n * kLinkStride * sizeof(uint16_t)));
    a.has128 = reinterpret_cast<uint64_t*>(take(n * kMaskWords * 8));
    a.sym = reinterpret_cast<uint64_t*>(take(n * kMaskWords * 8));
    a.mst = reinterpret_cast<uint64_t*>(take(n * kMaskWords * 8));
    a.perm = reinterpret_cast<uint16_t*>(take(n * kTile * sizeof(uint16_t)));
    a.link_stat = reinterpret_cast<uint32_t*>(take(n * sizeof(uint32_t)));
    a.tickets = reinterpret_cast<uint32_t*>(take(kTicketBytes));
    a.sorted = reinterpret_cast<uint32_t*>(take(n * kSortedStride * sizeof(uint32_t)));
    a.dense_list = reinterpret_cast<uint32_t*>(take(n * sizeof(uint32_t)));
    a.sparse_list = reinterpret_cast<uint32_t*>(take(n * sizeof(uint32_t)));
    a.cu_count = c->cu_count;
    a.info = reinterpret_cast<ChunkInfo*>(take(n * sizeof(ChunkInfo)));
    a.blocks = reinterpret_cast<BlockInfo*>(take(n * kMaxBlocks * sizeof(BlockInfo)));
    a.plans = reinterpret_cast<BlockOut*>(take(n * kMaxBlocks * sizeof(BlockOut)));
    static_assert(kMaxBlocks * sizeof(BlockProbe) <= kLinkStride * sizeof(uint16_t), "probes alias the link array");
    a.probes = reinterpret_cast<BlockProbe*>(a.links);
}

int ensure_staging(zwz_ctx* c, uint32_t m) {
    if (m <= c->stage_chunks) return ZWZ_OK;
    if (c->d_stage) { (void)hipFree(c->d_stage); c->d_stage = nullptr; }
    if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; }
    c->stage_chunks = 0;
    const size_t bytes = stage_bytes(m);
    hipError_t e = hipMalloc(&c->d_stage, bytes);
    if (e == hipSuccess) e = hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return hip_fail(e, "staging allocation");
    c->stage_chunks = m;
    return ZWZ_OK;
}

}  // namespace zwz
