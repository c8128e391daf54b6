// How fast does the LDS retire returning atomics (ds_wrxchg_rtn_b32 / ds_mskor_rtn_b32), and is the limit per CU or per wave?
// One workgroup per CU with W waves, each wave issuing R exchanges on pseudo-random addresses of its own table slice.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int kOp>
__global__ void k(uint32_t* out, uint32_t rounds) {
    extern __shared__ uint32_t tab[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 16384u; i += blockDim.x) tab[i] = i;
    __syncthreads();
    typedef __attribute__((address_space(3))) uint32_t* lp;
    const uint32_t base = (uint32_t)(uintptr_t)(lp)&tab[0];
    uint32_t rng = 0x9e3779b9u * (threadIdx.x + 1u) + blockIdx.x, acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t a[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { rng = rng * 1664525u + 1013904223u; a[u] = base + 4u * ((rng >> 10) & 16383u); }
        if (kOp == 0) asm volatile("ds_wrxchg_rtn_b32 %0, %4, %8\n\tds_wrxchg_rtn_b32 %1, %5, %8\n\tds_wrxchg_rtn_b32 %2, %6, %8\n\tds_wrxchg_rtn_b32 %3, %7, %8\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(r) : "memory");
        if (kOp == 1) asm volatile("ds_mskor_rtn_b32 %0, %4, %9, %8\n\tds_mskor_rtn_b32 %1, %5, %9, %8\n\tds_mskor_rtn_b32 %2, %6, %9, %8\n\tds_mskor_rtn_b32 %3, %7, %9, %8\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(r & 0xffffu), "v"(0xffffu) : "memory");
        if (kOp == 2) asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
        if (kOp == 3) asm volatile("ds_add_rtn_u32 %0, %4, %8\n\tds_add_rtn_u32 %1, %5, %8\n\tds_add_rtn_u32 %2, %6, %8\n\tds_add_rtn_u32 %3, %7, %8\n\ts_waitcnt lgkmcnt(0)"
                                  : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(r) : "memory");
        acc += o[0] ^ o[1] ^ o[2] ^ o[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    (void)wave; (void)lane;
}
// ordering check for ds_mskor_rtn_b32 on 16-bit halves: lanes naming the same half chain in ascending lane order?
__global__ void order_mskor(uint32_t* out, uint32_t rounds) {
    __shared__ uint32_t tab[64];
    __shared__ uint32_t addr[64];
    const uint32_t lane = threadIdx.x;
    uint32_t bad = 0, rng = 0x9e3779b9u * (lane + 1u);
    for (uint32_t r = 0; r < rounds; r++) {
        tab[lane] = 0xd000d000u + lane * 0x10001u;
        rng = rng * 1664525u + 1013904223u;
        const uint32_t spread = 1u << (2u * (r % 4u));                  // 1, 4, 16, 64 half-words
        const uint32_t a = (rng >> 12) & (spread - 1u);                 // half-word index
        addr[lane] = a;
        __syncthreads();
        typedef __attribute__((address_space(3))) uint32_t* lp;
        const uint32_t la = (uint32_t)(uintptr_t)(lp)&tab[a >> 1];
        const uint32_t sh = (a & 1u) * 16u, mask = 0xffffu << sh, val = (0x1000u + lane) << sh;
        uint32_t old;
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(la), "v"(mask), "v"(val) : "memory");
        __syncthreads();
        uint32_t want = (0xd000u + (a >> 1)) & 0xffffu, last = lane;
        for (uint32_t j = 0; j < 64; j++) { if (addr[j] != a) continue; if (j < lane) want = 0x1000u + j; if (j > lane) last = j; }
        bad += (uint32_t)(((old >> sh) & 0xffffu) != want) + (uint32_t)(((tab[a >> 1] >> sh) & 0xffffu) != 0x1000u + last);
        __syncthreads();
    }
    out[lane] = bad;
}
template <int kOp> static void run(const char* name, int waves) {
    uint32_t* o; hipMalloc(&o, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t rounds = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<kOp>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<kOp>, dim3(256), dim3(64 * waves), 65536, 0, o, 100u);
    hipEventRecord(e0); hipLaunchKernelGGL(k<kOp>, dim3(256), dim3(64 * waves), 65536, 0, o, rounds); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)rounds * 4 * waves;     // wave-instructions per CU
    printf("%-18s waves/CU=%d  %.3f ms  -> %.1f ns per wave-op per CU = %.1f cycles @2.4GHz\n", name, waves, ms, ms * 1e6 / ops, ms * 1e6 / ops * 2.4);
    hipFree(o);
}
int main() {
    for (int w : {1, 2, 4, 8}) run<0>("ds_wrxchg_rtn_b32", w);
    for (int w : {1, 2, 4, 8}) run<1>("ds_mskor_rtn_b32", w);
    for (int w : {1, 2, 4, 8}) run<3>("ds_add_rtn_u32", w);
    for (int w : {1, 2, 4, 8}) run<2>("ds_read_b32", w);
    uint32_t* o; hipMalloc(&o, 256); hipLaunchKernelGGL(order_mskor, dim3(1), dim3(64), 0, 0, o, 2000u);
    uint32_t res[64]; hipMemcpy(res, o, 256, hipMemcpyDeviceToHost); uint32_t tot = 0; for (int i = 0; i < 64; i++) tot += res[i];
    printf("ds_mskor_rtn_b32 on 16-bit halves chains same-half lanes in ascending lane order: %s (%u violations)\n", tot ? "NO" : "yes", tot);
    return 0;
}
