#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
    uint32_t lo = 0x33221100u, hi = 0x77665544u, sh = threadIdx.x, r;
    asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "v"(sh));
    out[threadIdx.x] = r;
    // physical-register pair + ds_read2
    __shared__ uint32_t s[128];
    s[threadIdx.x] = threadIdx.x * 0x01010101u; s[threadIdx.x + 64] = 0xabcd0000u + threadIdx.x; __syncthreads();
    typedef __attribute__((address_space(3))) uint8_t* lds_ptr;
    uint32_t a = (uint32_t)(uintptr_t)(lds_ptr)(uint8_t*)s + threadIdx.x * 4u, x;
    uint32_t t0, t1;
    asm volatile("ds_read2_b32 v[126:127], %3 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_alignbyte_b32 %0, v127, v126, %4" : "=v"(x), "=&{v126}"(t0), "=&{v127}"(t1) : "v"(a), "v"(threadIdx.x) : "memory");
    out[64 + threadIdx.x] = x;
}
int main() {
    uint32_t* d; hipMalloc(&d, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int t = 0; t < 10; t++) printf("sh=%d -> %08x   pair: %08x\n", t, h[t], h[64 + t]);
    printf("sh=35 -> %08x sh=63 -> %08x\n", h[35], h[63]);
    return 0;
}
