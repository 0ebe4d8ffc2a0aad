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

// ---- minimum-redundancy code lengths in place (Moffat & Katajainen 1995).  A[0..m) = frequencies in ASCENDING order
// (all > 0) on entry, code lengths (descending) on return.  m >= 2.
FADE_HD void mr_code_lengths(uint32_t *A, int m) {
    if (m == 1) { A[0] = 1; return; }
    A[0] += A[1];
    int root = 0, leaf = 2;
    for (int next = 1; next < m - 1; next++) {
        if (leaf >= m || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint32_t)next; }
        else A[next] = A[leaf++];
        if (leaf >= m || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint32_t)next; }
        else A[next] += A[leaf++];
    }
    A[m - 2] = 0;
    for (int next = m - 3; next >= 0; next--) A[next] = A[A[next]] + 1;
    int avbl = 1, used = 0, depth = 0;
    root = m - 2;
    int next = m - 1;
    while (avbl > 0) {
        while (root >= 0 && (int)A[root] == depth) { used++; root--; }
        while (avbl > used) { A[next--] = (uint32_t)depth; avbl--; }
        avbl = 2 * used;
        depth++;
        used = 0;
    }
}

// ---- length limit.  A[0..m) = optimal lengths of the symbols in ascending order of frequency (so lengths descend).
// Counts per length with everything beyond max_bits folded into max_bits; while the Kraft sum exceeds 1, one code of
// max_bits is taken away and the deepest shorter code becomes two codes one bit longer (the number of codes stays, the
// sum drops by 2^-max_bits); then the lengths are handed out again, longest to the rarest.  bl[0..32] is scratch.
FADE_HD void limit_code_lengths(uint32_t *A, int m, int max_bits, uint32_t *bl) {
    if (m < 2 || (int)A[0] <= max_bits) return;
    for (int b = 0; b <= max_bits; b++) bl[b] = 0;
    for (int i = 0; i < m; i++) bl[(int)A[i] > max_bits ? max_bits : (int)A[i]]++;
    uint32_t total = 0;
    for (int b = max_bits; b > 0; b--) total += bl[b] << (max_bits - b);
    while (total != (1u << max_bits)) {
        bl[max_bits]--;
        for (int b = max_bits - 1; b > 0; b--)
            if (bl[b]) { bl[b]--; bl[b + 1] += 2; break; }
        total--;
    }
    int i = 0;
    for (int bits = max_bits; bits >= 1; bits--)
        for (uint32_t k = 0; k < bl[bits]; k++) A[i++] = (uint32_t)bits;
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

// ---- a little bit writer over 32-bit words the caller zeroed (single lane / single thread use)
struct BitW {
    uint32_t *w;
    uint32_t pos;  // bits written
    FADE_HD void put(uint32_t v, int n) {  // n <= 16
        if (!n) return;
        const uint32_t at = pos >> 5, sh = pos & 31u;
        w[at] |= v << sh;
        if (sh + (uint32_t)n > 32u) w[at + 1] |= v >> (32u - sh);
        pos += (uint32_t)n;
    }
};

// ---- the dynamic-block header: BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN, the code-length code and the run-length
// coded lengths of the two alphabets (RFC 1951 §3.2.7).  ll[0..286) / dl[0..30) are the code lengths; cl_sym / cl_ext
// (room for 320 entries) and the small arrays are scratch.  Returns the number of header bits written to bw.
FADE_HD uint32_t write_dynamic_header(BitW &bw, const uint8_t *ll, const uint8_t *dl, uint8_t *cl_sym, uint8_t *cl_ext,
                                      uint32_t *sortbuf /* >= 2 * 19 + 8 */) {
    int hlit = NUM_LITLEN, hdist = NUM_DIST;
    while (hlit > 257 && ll[hlit - 1] == 0) hlit--;
    while (hdist > 1 && dl[hdist - 1] == 0) hdist--;
    const int n = hlit + hdist;
    auto at = [&](int k) -> int { return k < hlit ? ll[k] : dl[k - hlit]; };
    // run-length code the n lengths: 16 = repeat previous 3-6, 17 = 3-10 zeros, 18 = 11-138 zeros
    int nt = 0;
    uint32_t freq[NUM_CL];
    for (int k = 0; k < NUM_CL; k++) freq[k] = 0;
    for (int i = 0; i < n;) {
        const int v = at(i);
        int run = 1;
        while (i + run < n && at(i + run) == v) run++;
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int r = left > 138 ? 138 : left; cl_sym[nt] = 18; cl_ext[nt++] = (uint8_t)(r - 11); freq[18]++; left -= r; }
            if (left >= 3) { cl_sym[nt] = 17; cl_ext[nt++] = (uint8_t)(left - 3); freq[17]++; left = 0; }
            while (left-- > 0) { cl_sym[nt] = 0; cl_ext[nt++] = 0; freq[0]++; }
        } else {
            cl_sym[nt] = (uint8_t)v; cl_ext[nt++] = 0; freq[v]++; left--;
            while (left >= 3) { const int r = left > 6 ? 6 : left; cl_sym[nt] = 16; cl_ext[nt++] = (uint8_t)(r - 3); freq[16]++; left -= r; }
            while (left-- > 0) { cl_sym[nt] = (uint8_t)v; cl_ext[nt++] = 0; freq[v]++; }
        }
        i += run;
    }
    // code lengths of the code-length alphabet (limit 7): sort the used symbols by (frequency, symbol)
    uint32_t *sf = sortbuf, *ss = sortbuf + NUM_CL, *bl = sortbuf + 2 * NUM_CL;
    int m = 0;
    for (int s = 0; s < NUM_CL; s++)
        if (freq[s]) {
            int j = m++;
            while (j > 0 && sf[j - 1] > freq[s]) { sf[j] = sf[j - 1]; ss[j] = ss[j - 1]; j--; }
            sf[j] = freq[s];
            ss[j] = (uint32_t)s;
        }
    uint8_t cll[NUM_CL];
    uint16_t clc[NUM_CL];
    for (int s = 0; s < NUM_CL; s++) cll[s] = 0;
    if (m == 1) cll[ss[0]] = 1;
    else {
        mr_code_lengths(sf, m);
        limit_code_lengths(sf, m, MAX_CL_BITS, bl);
        for (int k = 0; k < m; k++) cll[ss[k]] = (uint8_t)sf[k];
    }
    canonical_codes(cll, NUM_CL, MAX_CL_BITS, clc);
    const uint8_t order[NUM_CL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = NUM_CL;
    while (hclen > 4 && cll[order[hclen - 1]] == 0) hclen--;
    const uint32_t p0 = bw.pos;
    bw.put(1, 1);  // BFINAL
    bw.put(2, 2);  // BTYPE = dynamic
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; k++) bw.put(cll[order[k]], 3);
    for (int k = 0; k < nt; k++) {
        const int s = cl_sym[k];
        bw.put(clc[s], cll[s]);
        if (s == 16) bw.put(cl_ext[k], 2);
        else if (s == 17) bw.put(cl_ext[k], 3);
        else if (s == 18) bw.put(cl_ext[k], 7);
    }
    return bw.pos - p0;
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
