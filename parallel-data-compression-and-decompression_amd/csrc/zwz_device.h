// zwz_device.h -- wave-level helpers shared by the kernel files (device code only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zwz {

static __device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
static __device__ __forceinline__ uint64_t lanes_below() { return (1ull << lane_id()) - 1ull; }
// set bits of a wave mask below this lane (v_mbcnt pair: two instructions against four for popc(m & lanes_below()))
static __device__ __forceinline__ uint32_t rank_in(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Inclusive add-scan over the wave with DPP row shifts and broadcasts: six VALU instructions.  (`__shfl_up` is a
// ds_bpermute -- an LDS-pipeline operation -- and the encoder scans once per 64 symbols with 32 waves on the CU.)
static __device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

static __device__ __forceinline__ uint32_t wave_scan_max_incl(uint32_t v) {          // as wave_scan_incl, with max (0 = identity)
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

// Workgroup copy of n 16-byte vectors global -> LDS with eight loads in flight per thread.
static __device__ __forceinline__ void copy_vec16(uint4* dst, const uint4* src, uint32_t n) {
    const uint32_t T = blockDim.x;
    uint32_t i = threadIdx.x;
    for (; i + 7 * T < n; i += 8 * T) {
        uint4 r[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) r[u] = src[i + u * T];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) dst[i + u * T] = r[u];
    }
    for (; i < n; i += T) dst[i] = src[i];
}

}  // namespace zwz
