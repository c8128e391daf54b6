// Does one ds_wrxchg_rtn_b32 whose lanes hit the same address chain them in ascending lane order?
// (lane i gets what lane i-1 of its group wrote, the lowest gets the old value, the highest's value stays.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out, const uint32_t* addr_of_lane, uint32_t rounds) {
    __shared__ uint32_t tab[256];
    const uint32_t lane = threadIdx.x;
    uint32_t bad = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t i = lane; i < 256; i += 64) tab[i] = 0xdead0000u + i;
        __syncthreads();
        const uint32_t a = addr_of_lane[r * 64 + lane] & 255u;
        typedef __attribute__((address_space(3))) uint32_t* lp;
        const uint32_t la = (uint32_t)(uintptr_t)(lp)&tab[a];
        uint32_t old;
        const uint32_t mine = 0x1000u + lane;
        asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(la), "v"(mine) : "memory");
        __syncthreads();
        // expected: nearest lower lane with the same address, else the initial value
        uint32_t want = 0xdead0000u + a;
        for (uint32_t j = 0; j < lane; j++) if ((addr_of_lane[r * 64 + j] & 255u) == a) want = 0x1000u + j;
        uint32_t last = lane;
        for (uint32_t j = lane + 1; j < 64; j++) if ((addr_of_lane[r * 64 + j] & 255u) == a) last = j;
        if (old != want) bad++;
        if (tab[a] != 0x1000u + last) bad++;
        __syncthreads();
    }
    out[lane] = bad;
}
int main() {
    const uint32_t rounds = 2000;
    uint32_t* h = new uint32_t[rounds * 64];
    uint64_t s = 12345;
    for (uint32_t i = 0; i < rounds * 64; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; const uint32_t r = i / 64;
        const uint32_t spread = r % 5 == 0 ? 1 : r % 5 == 1 ? 4 : r % 5 == 2 ? 16 : r % 5 == 3 ? 64 : 256; h[i] = (uint32_t)(s >> 33) % spread; }
    uint32_t *d, *o; hipMalloc(&d, rounds * 64 * 4); hipMalloc(&o, 256); hipMemcpy(d, h, rounds * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d, rounds);
    uint32_t res[64]; hipMemcpy(res, o, 256, hipMemcpyDeviceToHost);
    uint32_t tot = 0; for (int i = 0; i < 64; i++) tot += res[i];
    printf("ds_wrxchg_rtn_b32 same-address lanes chain in ascending lane order: %s (%u violations over %u rounds)\n", tot ? "NO" : "yes", tot, rounds);
    return 0;
}
