// Random-gather throughput of the LDS by access width (lz_match's chain walk is made of these).
// 16 waves per CU (as lz_match), every lane reads a pseudo-random address of a 96 KiB region each trip.
// Prints LDS cycles per wave-instruction per CU (all waves' instructions over the kernel's cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ __launch_bounds__(1024) void k(uint32_t* out, uint32_t iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < 98304 / 4; i += 1024) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = (x >> 8) % 98000u;
            if (MODE == 0) acc += *reinterpret_cast<const uint16_t*>(lds + (a & ~1u));
            if (MODE == 1) acc += *reinterpret_cast<const uint32_t*>(lds + (a & ~3u));
            if (MODE == 2) { const uint2 v = *reinterpret_cast<const uint2*>(lds + (a & ~7u)); acc += v.x ^ v.y; }
            if (MODE == 3) { const uint32_t* w = reinterpret_cast<const uint32_t*>(lds + (a & ~3u)); acc += w[0] ^ w[1]; }   // two dwords (the round-1 filter read)
            if (MODE == 4) acc += lds[a];
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int MODE> void run(const char* name) {
    uint32_t* d; hipMalloc(&d, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const uint32_t iters = 2000, grid = 256 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 98304, 0, d, 10u);
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 98304, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_cu = (double)grid / 256.0 * 16.0 * iters * 4.0;      // gathers issued per CU
    printf("%-28s %.3f ms  -> %.2f cycles per wave-gather per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
}
int main() { run<0>("ds_read_u16"); run<1>("ds_read_b32"); run<2>("ds_read_b64 (8-aligned)"); run<3>("2 x ds_read_b32 (adjacent)"); run<4>("ds_read_u8"); return 0; }
