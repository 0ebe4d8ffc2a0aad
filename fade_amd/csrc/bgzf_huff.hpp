// bgzf_huff.hpp — the serial pieces of a DEFLATE (RFC 1951) encoder, written once for host and device: minimum-redundancy
// code lengths, the 15- / 7-bit length limit, canonical codes, the run-length coded code-length header, the length and
// distance symbol maps and the CRC-32 combination arithmetic.  The gfx950 BGZF compressor (bgzf_deflate.hpp: one
// workgroup per BGZF block) calls them from single lanes on LDS arrays; host/selftest/gpu_deflate_model.cpp calls the
// same functions on the CPU and checks the resulting streams with zlib's inflate (no GPU needed for the format logic).
//
// What it replaces: htslib's bgzf_write -> zlib deflate behind `SAMWriter(..., SAMWriterTypes.BAM)` (source/util.d:65-76),
// i.e. the write at source/anno.d:47-49.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FADE_HD __host__ __device__ inline
#else
#define FADE_HD inline
#endif

namespace fadehip {
namespace bgzf {

constexpr int NUM_LITLEN = 286, NUM_DIST = 30, NUM_CL = 19;
constexpr int MAX_LITLEN_BITS = 15, MAX_CL_BITS = 7;

// Arrays are reached through accessors (get / set) so that the same code runs on plain memory (host, MemArr) and on
// arrays spread over the lanes of a wavefront (device: WaveArr in bgzf_deflate.hpp, v_readlane / v_writelane — a
// dependent LDS round trip costs ~130 clocks, a lane access ~10, and these loops are chains of dependent accesses).
struct MemArr {
    uint32_t *p;
    FADE_HD uint32_t get(int i) const { return p[i]; }
    FADE_HD void set(int i, uint32_t v) { p[i] = v; }
};
template <class T>
struct MemArrT {
    T *p;
    FADE_HD uint32_t get(int i) const { return (uint32_t)p[i]; }
    FADE_HD void set(int i, uint32_t v) { p[i] = (T)v; }
};

// ---- minimum-redundancy code lengths in place (Moffat & Katajainen 1995).  A[0..m) = frequencies in ASCENDING order
// (all > 0) on entry, code lengths (descending) on return.
template <class Arr>
FADE_HD void mr_code_lengths_t(Arr &A, int m) {
    if (m == 1) { A.set(0, 1); return; }
    A.set(0, A.get(0) + A.get(1));
    int root = 0, leaf = 2;
    for (int next = 1; next < m - 1; next++) {
        uint32_t v;
        if (leaf >= m || A.get(root) < A.get(leaf)) { v = A.get(root); A.set(root++, (uint32_t)next); }
        else v = A.get(leaf++);
        if (leaf >= m || (root < next && A.get(root) < A.get(leaf))) { v += A.get(root); A.set(root++, (uint32_t)next); }
        else v += A.get(leaf++);
        A.set(next, v);
    }
    A.set(m - 2, 0);
    for (int next = m - 3; next >= 0; next--) A.set(next, A.get((int)A.get(next)) + 1);
    int avbl = 1, used = 0, depth = 0;
    root = m - 2;
    int next = m - 1;
    while (avbl > 0) {
        while (root >= 0 && (int)A.get(root) == depth) { used++; root--; }
        while (avbl > used) { A.set(next--, (uint32_t)depth); avbl--; }
        avbl = 2 * used;
        depth++;
        used = 0;
    }
}
FADE_HD void mr_code_lengths(uint32_t *A, int m) {
    MemArr a{A};
    mr_code_lengths_t(a, m);
}

// ---- length limit.  A[0..m) = optimal lengths of the symbols in ascending order of frequency (so lengths descend).
// Counts per length with everything beyond max_bits folded into max_bits; while the Kraft sum exceeds 1, one code of
// max_bits is taken away and the deepest shorter code becomes two codes one bit longer (the number of codes stays, the
// sum drops by 2^-max_bits); then the lengths are handed out again, longest to the rarest.  bl[0..max_bits] is scratch.
template <class Arr, class Bl>
FADE_HD void limit_code_lengths_t(Arr &A, int m, int max_bits, Bl &bl) {
    if (m < 2 || (int)A.get(0) <= max_bits) return;
    for (int b = 0; b <= max_bits; b++) bl.set(b, 0);
    for (int i = 0; i < m; i++) {
        const int l = (int)A.get(i) > max_bits ? max_bits : (int)A.get(i);
        bl.set(l, bl.get(l) + 1);
    }
    uint32_t total = 0;
    for (int b = max_bits; b > 0; b--) total += bl.get(b) << (max_bits - b);
    while (total != (1u << max_bits)) {
        bl.set(max_bits, bl.get(max_bits) - 1);
        for (int b = max_bits - 1; b > 0; b--)
            if (bl.get(b)) { bl.set(b, bl.get(b) - 1); bl.set(b + 1, bl.get(b + 1) + 2); break; }
        total--;
    }
    int i = 0;
    for (int bits = max_bits; bits >= 1; bits--)
        for (uint32_t k = 0; k < bl.get(bits); k++) A.set(i++, (uint32_t)bits);
}
FADE_HD void limit_code_lengths(uint32_t *A, int m, int max_bits, uint32_t *bl) {
    MemArr a{A}, b{bl};
    limit_code_lengths_t(a, m, max_bits, b);
}

FADE_HD uint32_t bit_reverse(uint32_t v, int n) {  // the low n bits of v, reversed
    uint32_t r = 0;
    for (int k = 0; k < n; k++) r |= ((v >> k) & 1u) << (n - 1 - k);
    return r;
}

// ---- canonical codes from lengths (RFC 1951 §3.2.2), bit-reversed: DEFLATE packs Huffman codes starting with their
// most significant bit into a stream filled from the least significant bit.  code[s] is meaningful where len[s] > 0.
FADE_HD void canonical_codes(const uint8_t *len, int n, int max_bits, uint16_t *code) {
    uint32_t bl_count[16], next_code[16];
    for (int b = 0; b <= max_bits; b++) bl_count[b] = 0;
    for (int s = 0; s < n; s++) bl_count[len[s]]++;
    bl_count[0] = 0;
    uint32_t c = 0;
    next_code[0] = 0;
    for (int b = 1; b <= max_bits; b++) {
        c = (c + bl_count[b - 1]) << 1;
        next_code[b] = c;
    }
    for (int s = 0; s < n; s++)
        if (len[s]) code[s] = (uint16_t)bit_reverse(next_code[len[s]]++, len[s]);
}

// ---- length / distance symbols (RFC 1951 §3.2.5)
struct Sym { uint32_t sym, ebits, eval; };
FADE_HD Sym length_symbol(uint32_t len) {  // 3..258
    Sym s;
    const uint32_t l = len - 3;
    if (l < 8) { s.sym = 257 + l; s.ebits = 0; s.eval = 0; }
    else if (len == 258) { s.sym = 285; s.ebits = 0; s.eval = 0; }
    else {
        const uint32_t n = 31u - (uint32_t)__builtin_clz(l);  // floor(log2 l) >= 3
        const uint32_t e = n - 2;
        s.sym = 261 + 4 * e + ((l >> e) & 3u);
        s.ebits = e;
        s.eval = l & ((1u << e) - 1u);
    }
    return s;
}
FADE_HD Sym dist_symbol(uint32_t dist) {  // 1..32768
    Sym s;
    const uint32_t d = dist - 1;
    if (d < 4) { s.sym = d; s.ebits = 0; s.eval = 0; }
    else {
        const uint32_t n = 31u - (uint32_t)__builtin_clz(d);  // floor(log2 d) >= 2
        s.sym = 2 * n + ((d >> (n - 1)) & 1u);
        s.ebits = n - 1;
        s.eval = d & ((1u << (n - 1)) - 1u);
    }
    return s;
}

// ---- bit sink over 32-bit words: whole words are stored, never read back (single lane / single thread use)
struct WordSink {
    uint32_t *w;
    uint64_t acc = 0;
    int cnt = 0;
    uint32_t wi = 0;
    FADE_HD void put(uint32_t v, int n) {  // n <= 16
        acc |= (uint64_t)v << cnt;
        cnt += n;
        if (cnt >= 32) {
            w[wi++] = (uint32_t)acc;
            acc >>= 32;
            cnt -= 32;
        }
    }
    FADE_HD uint32_t finish() {  // bits written; the last, partial word is stored too
        if (cnt) w[wi] = (uint32_t)acc;
        return 32u * wi + (uint32_t)cnt;
    }
};

// ---- run-length coding of the hlit + hdist code lengths (RFC 1951 §3.2.7): 16 = repeat the previous length 3-6 times,
// 17 = 3-10 zeros, 18 = 11-138 zeros.  emit(symbol, extra) is called per code-length symbol, in order.
template <class LL, class DL, class F>
FADE_HD void cl_rle(const LL &ll, const DL &dl, int hlit, int hdist, F &&emit) {
    const int n = hlit + hdist;
    int i = 0;
    while (i < n) {
        const int v = i < hlit ? (int)ll.get(i) : (int)dl.get(i - hlit);
        int run = 1;
        while (i + run < n && ((i + run) < hlit ? (int)ll.get(i + run) : (int)dl.get(i + run - hlit)) == v) run++;
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int r = left > 138 ? 138 : left; emit(18, r - 11); left -= r; }
            if (left >= 3) { emit(17, left - 3); left = 0; }
            while (left-- > 0) emit(0, 0);
        } else {
            emit(v, 0);
            left--;
            while (left >= 3) { const int r = left > 6 ? 6 : left; emit(16, r - 3); left -= r; }
            while (left-- > 0) emit(v, 0);
        }
        i += run;
    }
}

FADE_HD int cl_order(int k) {  // the order in which the code-length code lengths are sent
    const uint8_t order[NUM_CL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return order[k];
}

// ---- the dynamic-block header: BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN, the code-length code and the run-length
// coded lengths of the two alphabets (RFC 1951 §3.2.7).  ll (286) / dl (30) hold the code lengths; freq, sf, ss, cll,
// clc (19 entries each) and bl (8) are scratch arrays.  The run-length coding is done twice (count, then emit) instead of
// being stored.
template <class Sink, class LL, class DL, class A>
FADE_HD void write_dynamic_header_t(Sink &out, const LL &ll, const DL &dl, A &freq, A &sf, A &ss, A &cll, A &clc, A &bl) {
    int hlit = NUM_LITLEN, hdist = NUM_DIST;
    while (hlit > 257 && ll.get(hlit - 1) == 0) hlit--;
    while (hdist > 1 && dl.get(hdist - 1) == 0) hdist--;
    for (int s = 0; s < NUM_CL; s++) { freq.set(s, 0); cll.set(s, 0); }
    cl_rle(ll, dl, hlit, hdist, [&](int s, int) { freq.set(s, freq.get(s) + 1); });
    // code lengths of the code-length alphabet (limit 7): the used symbols sorted by (frequency, symbol)
    int m = 0;
    for (int s = 0; s < NUM_CL; s++) {
        const uint32_t f = freq.get(s);
        if (f) {
            int j = m++;
            while (j > 0 && sf.get(j - 1) > f) { sf.set(j, sf.get(j - 1)); ss.set(j, ss.get(j - 1)); j--; }
            sf.set(j, f);
            ss.set(j, (uint32_t)s);
        }
    }
    if (m == 1) cll.set((int)ss.get(0), 1);
    else {
        mr_code_lengths_t(sf, m);
        limit_code_lengths_t(sf, m, MAX_CL_BITS, bl);
        for (int k = 0; k < m; k++) cll.set((int)ss.get(k), sf.get(k));
    }
    // canonical codes of the 19 symbols: bl = count per length, then sf = next code per length
    for (int b = 0; b <= MAX_CL_BITS; b++) bl.set(b, 0);
    for (int s = 0; s < NUM_CL; s++)
        if (cll.get(s)) bl.set((int)cll.get(s), bl.get((int)cll.get(s)) + 1);
    uint32_t c = 0;
    sf.set(0, 0);
    for (int b = 1; b <= MAX_CL_BITS; b++) {
        c = (c + (b > 1 ? bl.get(b - 1) : 0u)) << 1;
        sf.set(b, c);
    }
    for (int s = 0; s < NUM_CL; s++) {
        const int l = (int)cll.get(s);
        if (l) {
            clc.set(s, bit_reverse(sf.get(l), l));
            sf.set(l, sf.get(l) + 1);
        }
    }
    int hclen = NUM_CL;
    while (hclen > 4 && cll.get(cl_order(hclen - 1)) == 0) hclen--;
    out.put(1, 1);  // BFINAL
    out.put(2, 2);  // BTYPE = dynamic
    out.put((uint32_t)(hlit - 257), 5);
    out.put((uint32_t)(hdist - 1), 5);
    out.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; k++) out.put(cll.get(cl_order(k)), 3);
    cl_rle(ll, dl, hlit, hdist, [&](int s, int e) {
        out.put(clc.get(s), (int)cll.get(s));
        if (s == 16) out.put((uint32_t)e, 2);
        else if (s == 17) out.put((uint32_t)e, 3);
        else if (s == 18) out.put((uint32_t)e, 7);
    });
}

// ---- CRC-32 (the gzip polynomial, reflected) arithmetic for combining the CRCs of pieces: crc(A || B) =
// mulmod(x^(8 |B|), crc(A)) ^ crc(B), the identity zlib's crc32_combine uses.
constexpr uint32_t CRC_POLY = 0xedb88320u;
FADE_HD uint32_t crc_mulmod(uint32_t a, uint32_t b) {  // a(x) * b(x) mod P, bit 31 = x^0
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8 n) mod P; x2n[k] = x^(2^k) mod P for k = 0..31 (crc_x2n_table)
FADE_HD uint32_t crc_x8n(uint32_t n_bytes, const uint32_t *x2n) {
    uint32_t p = 1u << 31;  // x^0
    uint32_t n = n_bytes;
    int k = 3;              // bytes -> bits
    while (n) {
        if (n & 1u) p = crc_mulmod(x2n[k & 31], p);
        n >>= 1;
        k++;
    }
    return p;
}
FADE_HD void crc_x2n_table(uint32_t *x2n) {
    uint32_t p = 1u << 30;  // x^1
    x2n[0] = p;
    for (int k = 1; k < 32; k++) x2n[k] = p = crc_mulmod(p, p);
}
FADE_HD uint32_t crc_table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
    return c;
}

}  // namespace bgzf
}  // namespace fadehip
