// crc32_fast.hpp — the gzip CRC-32 of a BGZF block by carry-less multiplication (PCLMULQDQ folding, Gopal et al.,
// "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009), with zlib's crc32 for the
// tail and for hosts without the instruction.  Every BGZF block is summed once when it is read and once when it is
// written; zlib 1.2.11's table CRC runs at ~1 GB/s per core, which was 15 % of the host pipeline's CPU time.
// Same interface and results as zlib's crc32(crc, buf, len); selftest/deflate_selftest.cpp compares the two.
#pragma once
#include <cstddef>
#include <cstdint>
#include <immintrin.h>
#include <zlib.h>

namespace htsl {

// Folds n bytes (n >= 64, n % 16 == 0) into the running remainder `crc` (zlib's internal, inverted, form).
// Constants: x^k mod P for the reflected polynomial 0xEDB88320 — k1,k2 fold by 512 bits, k3,k4 by 128, k5 by 64;
// `poly` holds P' and the Barrett constant mu.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(uint32_t crc, const uint8_t *p, size_t n) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
#define FADE_CRC_LD(q) _mm_loadu_si128(reinterpret_cast<const __m128i *>(q))
// acc folded forward by 128 bits onto next (k = k3:k4)
#define FADE_CRC_FOLD128(acc, next) \
    _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k, 0x11), next), _mm_clmulepi64_si128(acc, k, 0x00))
    __m128i x1 = FADE_CRC_LD(p), x2 = FADE_CRC_LD(p + 16), x3 = FADE_CRC_LD(p + 32), x4 = FADE_CRC_LD(p + 48);
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    __m128i k = _mm_load_si128(reinterpret_cast<const __m128i *>(k1k2));
    p += 64;
    n -= 64;
    for (; n >= 64; p += 64, n -= 64) {  // four independent 128-bit lanes, each folded forward by 512 bits
        const __m128i l1 = _mm_clmulepi64_si128(x1, k, 0x00), l2 = _mm_clmulepi64_si128(x2, k, 0x00);
        const __m128i l3 = _mm_clmulepi64_si128(x3, k, 0x00), l4 = _mm_clmulepi64_si128(x4, k, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, l1), FADE_CRC_LD(p));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, l2), FADE_CRC_LD(p + 16));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, l3), FADE_CRC_LD(p + 32));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, l4), FADE_CRC_LD(p + 48));
    }
    k = _mm_load_si128(reinterpret_cast<const __m128i *>(k3k4));
    x1 = FADE_CRC_FOLD128(x1, x2);
    x1 = FADE_CRC_FOLD128(x1, x3);
    x1 = FADE_CRC_FOLD128(x1, x4);
    for (; n >= 16; p += 16, n -= 16) x1 = FADE_CRC_FOLD128(x1, FADE_CRC_LD(p));
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    k = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(k5k0));
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k, 0x00), t);
    // Barrett reduction 64 -> 32 bits
    k = _mm_load_si128(reinterpret_cast<const __m128i *>(poly));
    t = _mm_and_si128(x1, mask32);
    t = _mm_clmulepi64_si128(t, k, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, k, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
#undef FADE_CRC_LD
#undef FADE_CRC_FOLD128
}

inline bool crc32_have_clmul() {
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return have;
}

// zlib's crc32(crc, buf, len): start from 0, feed the previous return value to continue
inline uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n) {
    if (n >= 64 && crc32_have_clmul()) {
        const size_t m = n & ~(size_t)15;
        crc = ~crc32_fold(~crc, p, m);
        p += m;
        n -= m;
    }
    while (n) {  // (uInt lengths)
        const size_t c = n < (1u << 30) ? n : (1u << 30);
        crc = (uint32_t)crc32(crc, p, (uInt)c);
        p += c;
        n -= c;
    }
    return crc;
}

}  // namespace htsl
