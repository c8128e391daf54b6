#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) uint8_t s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    typedef __attribute__((address_space(3))) uint8_t* lds_ptr;
    uint32_t base = (uint32_t)(uintptr_t)(lds_ptr)s;
    uint32_t a = base + threadIdx.x * 5 + 1, v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x] = v;
    // timing: dependent chain of unaligned vs aligned reads
    long long t0 = clock64();
    uint32_t x = a;
    for (int i = 0; i < 256; i++) { uint32_t r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(x) : "memory"); x = base + ((r + x) & 1023u & ~0u) % 1000u; }
    long long t1 = clock64();
    uint32_t y = base + threadIdx.x * 4;
    for (int i = 0; i < 256; i++) { uint32_t r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(y) : "memory"); y = base + (((r + y) & 1023u) % 1000u & ~3u); }
    long long t2 = clock64();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    out[64 + threadIdx.x] = x + y;
}
int main() {
    uint32_t* d; long long* c; hipMalloc(&d, 512 * 4); hipMalloc(&c, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
    uint32_t h[128]; long long hc[2]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) {
        uint32_t o = t * 5 + 1, e = 0;
        for (int j = 0; j < 4; j++) e |= (uint32_t)(uint8_t)((o + j) * 7 + 3) << (8 * j);
        if (e != h[t]) { bad++; if (bad < 5) printf("lane %d off %u got %08x want %08x\n", t, o, h[t], e); }
    }
    printf("unaligned ds_read_b32: %s (%d bad); cycles unaligned %lld aligned %lld\n", bad ? "WRONG" : "OK", bad, hc[0], hc[1]);
    return 0;
}
