// zwz_md5.h -- streaming RFC 1321 MD5 (verification.cpp:13-27 renders the digest as 32 lowercase hex chars).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>

// The 64 steps of one block on the working variables a, b, c, d and the sixteen message words w[0..15];
// rol must be in scope.  Shared by the host class below and the device kernel (zwz_kernels.hip: md5_files).
#define ZWZ_MD5_R1(a, b, c, d, k, s, t) a = b + rol(a + ((b & c) | (~b & d)) + w[k] + t, s)
#define ZWZ_MD5_R2(a, b, c, d, k, s, t) a = b + rol(a + ((b & d) | (c & ~d)) + w[k] + t, s)
#define ZWZ_MD5_R3(a, b, c, d, k, s, t) a = b + rol(a + (b ^ c ^ d) + w[k] + t, s)
#define ZWZ_MD5_R4(a, b, c, d, k, s, t) a = b + rol(a + (c ^ (b | ~d)) + w[k] + t, s)
#define ZWZ_MD5_STEPS \
    ZWZ_MD5_R1(a,b,c,d,0,7,0xd76aa478u); ZWZ_MD5_R1(d,a,b,c,1,12,0xe8c7b756u); ZWZ_MD5_R1(c,d,a,b,2,17,0x242070dbu); ZWZ_MD5_R1(b,c,d,a,3,22,0xc1bdceeeu); \
    ZWZ_MD5_R1(a,b,c,d,4,7,0xf57c0fafu); ZWZ_MD5_R1(d,a,b,c,5,12,0x4787c62au); ZWZ_MD5_R1(c,d,a,b,6,17,0xa8304613u); ZWZ_MD5_R1(b,c,d,a,7,22,0xfd469501u); \
    ZWZ_MD5_R1(a,b,c,d,8,7,0x698098d8u); ZWZ_MD5_R1(d,a,b,c,9,12,0x8b44f7afu); ZWZ_MD5_R1(c,d,a,b,10,17,0xffff5bb1u); ZWZ_MD5_R1(b,c,d,a,11,22,0x895cd7beu); \
    ZWZ_MD5_R1(a,b,c,d,12,7,0x6b901122u); ZWZ_MD5_R1(d,a,b,c,13,12,0xfd987193u); ZWZ_MD5_R1(c,d,a,b,14,17,0xa679438eu); ZWZ_MD5_R1(b,c,d,a,15,22,0x49b40821u); \
    ZWZ_MD5_R2(a,b,c,d,1,5,0xf61e2562u); ZWZ_MD5_R2(d,a,b,c,6,9,0xc040b340u); ZWZ_MD5_R2(c,d,a,b,11,14,0x265e5a51u); ZWZ_MD5_R2(b,c,d,a,0,20,0xe9b6c7aau); \
    ZWZ_MD5_R2(a,b,c,d,5,5,0xd62f105du); ZWZ_MD5_R2(d,a,b,c,10,9,0x02441453u); ZWZ_MD5_R2(c,d,a,b,15,14,0xd8a1e681u); ZWZ_MD5_R2(b,c,d,a,4,20,0xe7d3fbc8u); \
    ZWZ_MD5_R2(a,b,c,d,9,5,0x21e1cde6u); ZWZ_MD5_R2(d,a,b,c,14,9,0xc33707d6u); ZWZ_MD5_R2(c,d,a,b,3,14,0xf4d50d87u); ZWZ_MD5_R2(b,c,d,a,8,20,0x455a14edu); \
    ZWZ_MD5_R2(a,b,c,d,13,5,0xa9e3e905u); ZWZ_MD5_R2(d,a,b,c,2,9,0xfcefa3f8u); ZWZ_MD5_R2(c,d,a,b,7,14,0x676f02d9u); ZWZ_MD5_R2(b,c,d,a,12,20,0x8d2a4c8au); \
    ZWZ_MD5_R3(a,b,c,d,5,4,0xfffa3942u); ZWZ_MD5_R3(d,a,b,c,8,11,0x8771f681u); ZWZ_MD5_R3(c,d,a,b,11,16,0x6d9d6122u); ZWZ_MD5_R3(b,c,d,a,14,23,0xfde5380cu); \
    ZWZ_MD5_R3(a,b,c,d,1,4,0xa4beea44u); ZWZ_MD5_R3(d,a,b,c,4,11,0x4bdecfa9u); ZWZ_MD5_R3(c,d,a,b,7,16,0xf6bb4b60u); ZWZ_MD5_R3(b,c,d,a,10,23,0xbebfbc70u); \
    ZWZ_MD5_R3(a,b,c,d,13,4,0x289b7ec6u); ZWZ_MD5_R3(d,a,b,c,0,11,0xeaa127fau); ZWZ_MD5_R3(c,d,a,b,3,16,0xd4ef3085u); ZWZ_MD5_R3(b,c,d,a,6,23,0x04881d05u); \
    ZWZ_MD5_R3(a,b,c,d,9,4,0xd9d4d039u); ZWZ_MD5_R3(d,a,b,c,12,11,0xe6db99e5u); ZWZ_MD5_R3(c,d,a,b,15,16,0x1fa27cf8u); ZWZ_MD5_R3(b,c,d,a,2,23,0xc4ac5665u); \
    ZWZ_MD5_R4(a,b,c,d,0,6,0xf4292244u); ZWZ_MD5_R4(d,a,b,c,7,10,0x432aff97u); ZWZ_MD5_R4(c,d,a,b,14,15,0xab9423a7u); ZWZ_MD5_R4(b,c,d,a,5,21,0xfc93a039u); \
    ZWZ_MD5_R4(a,b,c,d,12,6,0x655b59c3u); ZWZ_MD5_R4(d,a,b,c,3,10,0x8f0ccc92u); ZWZ_MD5_R4(c,d,a,b,10,15,0xffeff47du); ZWZ_MD5_R4(b,c,d,a,1,21,0x85845dd1u); \
    ZWZ_MD5_R4(a,b,c,d,8,6,0x6fa87e4fu); ZWZ_MD5_R4(d,a,b,c,15,10,0xfe2ce6e0u); ZWZ_MD5_R4(c,d,a,b,6,15,0xa3014314u); ZWZ_MD5_R4(b,c,d,a,13,21,0x4e0811a1u); \
    ZWZ_MD5_R4(a,b,c,d,4,6,0xf7537e82u); ZWZ_MD5_R4(d,a,b,c,11,10,0xbd3af235u); ZWZ_MD5_R4(c,d,a,b,2,15,0x2ad7d2bbu); ZWZ_MD5_R4(b,c,d,a,9,21,0xeb86d391u);

namespace zwz {

struct Md5 {
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint64_t total = 0;
    uint8_t buf[64];
    uint32_t fill = 0;

    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        uint32_t w[16];
        memcpy(w, p, 64);  // little-endian host
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
        ZWZ_MD5_STEPS
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        if (fill) {
            size_t k = std::min<size_t>(64 - fill, n);
            memcpy(buf + fill, p, k); fill += (uint32_t)k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) { memcpy(buf, p, n); fill = (uint32_t)n; }
    }
    void hex(char out[33]) {
        uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (fill < 56 ? 56 : 120) - fill;
        uint64_t saved = total;
        update(pad, padlen);
        uint8_t lenb[8];
        for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
        update(lenb, 8);
        total = saved;
        static const char* dig = "0123456789abcdef";
        for (int i = 0; i < 16; i++) {
            uint8_t v = (uint8_t)(h[i >> 2] >> (8 * (i & 3)));
            out[2 * i] = dig[v >> 4]; out[2 * i + 1] = dig[v & 15];
        }
        out[32] = 0;
    }
};


}  // namespace zwz
