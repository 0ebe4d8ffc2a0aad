// deflate_fast.hpp — a DEFLATE (RFC 1951) compressor for BGZF blocks.
//
// once the alignment kernels run at hundreds of millions of reads per second the end-to-end
// rate of `fade annotate` is set by the BGZF codec (the reference gets it from htslib + zlib, util.d:65-76).  A BGZF
// block is an independent <= 64 KiB DEFLATE stream, so the compressor can be small: positions fit 16 bits, the match
// table needs no window sliding, one dynamic-Huffman block per BGZF block.  BAM payloads are dominated by base
// qualities and packed bases (entropy-coded, few matches), so the match finder is built to fail cheaply (see
// `find`), a profitability filter drops the short far matches zlib takes at a loss, and the Huffman encoder is
// table-driven.  Output is standard DEFLATE: any inflater (zlib, htslib, libdeflate) reads it.
//
// Not derived from zlib/libdeflate sources; follows RFC 1951 only.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>

namespace htsl {

class FastDeflate {
public:
    static constexpr size_t MAX_IN = 65535;
    // worst case: a stored block (5 bytes of framing); the caller's buffer must hold n + 16 bytes
    static constexpr size_t bound(size_t n) { return n + 16; }

    // effort: 1 = greedy parse that stops probing early in match-free stretches, 2 = lazy parse with a milder skip
    // (default: output smaller than zlib level 6's on BAM payloads), both over the 16-bit one-candidate table
    // (parse_fast); 3 = lazy parse probing every position, 4 = lazy parse over 4 candidates per hash (parse).
    // skip_after / skip_cap >= 0 override the effort's skip rule.
    static constexpr int MAX_EFFORT = 4;
    explicit FastDeflate(int effort = 2, int skip_after = -1, int skip_cap = -1) : effort_(std::max(1, std::min(MAX_EFFORT, effort))) {
        lazy_ = effort_ >= 2;
        skip_after_ = effort_ == 1 ? 16 : effort_ == 2 ? 48 : 0;
        skip_cap_ = effort_ <= 2 ? 7 : 0;
        if (skip_after >= 0) skip_after_ = (size_t)skip_after;
        if (skip_cap >= 0) skip_cap_ = (size_t)skip_cap;
        init_static();
    }

    // What the caller knows about the layout of the bytes (efforts 1 and 2): from input offset `pos` on, the parser's count
    // of probes in vain is `miss` — a large value where a stretch without repeats begins (a BAM record's bases and
    // qualities: the skip-ahead starts at once instead of after skip_after probes), 0 where structure resumes (its tags and
    // the next record's fixed fields: every position probed again, and no skip reaches across).  Sorted by pos.
    struct Hint { uint32_t pos, miss; };
    static constexpr uint32_t HINT_SKIP = 1u << 20;  // nothing to find here (packed bases)
    static constexpr uint32_t HINT_MILD = 1;         // probably nothing, but runs happen (qualities): the first skip step only

    // Compresses in[0, n), n <= MAX_IN, into one final DEFLATE block at out; returns the byte count (<= bound(n)).
    size_t compress(const uint8_t *in, size_t n, uint8_t *out, const Hint *hints = nullptr, size_t n_hints = 0) {
        hints_ = hints;
        n_hints_ = hints ? n_hints : 0;
#ifdef FADE_DEFLATE_TIMING  // selftest only: where a block's time goes
        const auto t0 = std::chrono::steady_clock::now();
#endif
        if (effort_ >= 4) parse<4, 15>(in, n);
        else if (effort_ == 3) parse<1, 14>(in, n);
        else parse_fast<14>(in, n);
#ifdef FADE_DEFLATE_TIMING
        const auto t1 = std::chrono::steady_clock::now();
        const size_t r = encode(in, n, out);
        t_parse += std::chrono::duration<double>(t1 - t0).count();
        t_encode += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        return r;
#else
        return encode(in, n, out);
#endif
    }
#ifdef FADE_DEFLATE_TIMING
    double t_parse = 0, t_encode = 0;
#endif

private:
    // ------------------------------------------------------------------ LZ77 parse
    // Match finder: a hash of 5 bytes selects a bucket holding the WAYS most recent positions with that hash (one
    // 64-bit word, a FIFO by shifting).  The candidates are checked without data-dependent branches — on qualities
    // and packed bases nearly every probe fails, and a chained search there pays a branch miss plus an exposed
    // cache-miss chain per position (measured: 45 cycles/position against 20).  Only a 5-byte hit branches into the
    // extension code.  Minimum match 5: shorter ones rarely pay for their distance bits.
    // Measured on BAM payloads (selftest/deflate_selftest <file>; uniform qualities / binned qualities with runs):
    //   1 way,  14 bits: 0.5701 / 0.4082 of the input at 70 MB/s       (effort 3)
    //   4 ways, 15 bits: 0.5684 / 0.40   at 29 MB/s                    (effort 4)
    //   zlib level 6   : 0.5908 / 0.3957 at 8-14 MB/s on the same core
    // The parse is 78 % of a block's time and nearly all of it is probes that find nothing, so efforts 1 and 2 probe
    // four positions per round and skip ahead in match-free stretches (parse_fast).  On an idle core of the GPU box's
    // host, uniform qualities (tools/deflate_where.sh): effort 1 564 MB/s 0.5954, effort 2 462 MB/s 0.5729 (with the BAM
    // writer's layout hints: 590 MB/s 0.5709 and 540 MB/s 0.5704), effort 3 201 MB/s 0.5709, effort 4 86 MB/s 0.5691;
    // zlib level 1 88 MB/s 0.6123, zlib level 6 40 MB/s 0.5915.
    static constexpr int MIN_MATCH = 5, MAX_MATCH = 258;
    static constexpr uint64_t MASK5 = 0xffffffffffull;
    static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
    template <int HB> static inline uint32_t hash5(uint64_t v5) { return (uint32_t)((v5 * 0x9E3779B185EBCA87ull) >> (64 - HB)); }

    static inline int match_len(const uint8_t *a, const uint8_t *b, int maxlen) {
        int l = 0;
        while (l + 8 <= maxlen) {
            const uint64_t x = load64(a + l) ^ load64(b + l);
            if (x) return l + (__builtin_ctzll(x) >> 3);
            l += 8;
        }
        while (l < maxlen && a[l] == b[l]) l++;
        return l;
    }
    // a match pays for itself only if it is long enough for its distance (extra distance bits vs literal bits)
    static inline bool worth(int len, int dist) {
        return len >= 7 || (len == 6 && dist <= 16384) || (len == 5 && dist <= 4096) || (len == 4 && dist <= 512);
    }

    // positions p with p + 8 <= n can be hashed (the 8-byte load stays inside the block)
    template <int HB> inline void insert(const uint8_t *in, size_t p) {
        const uint32_t h = hash5<HB>(load64(in + p) & MASK5);
        tab_[h] = (tab_[h] << 16) | (uint64_t)(p + 1);
    }
    // longest profitable match at p (which is inserted); returns length or 0
    template <int WAYS, int HB> inline int find(const uint8_t *in, size_t n, size_t p, int &dist_out) {
        const uint64_t v = load64(in + p) & MASK5;
        const uint32_t h = hash5<HB>(v);
        const uint64_t b = tab_[h];
        tab_[h] = (b << 16) | (uint64_t)(p + 1);
        uint32_t hits = 0;
        size_t cpos[WAYS];
#pragma GCC unroll 4
        for (int w = 0; w < WAYS; w++) {
            const uint32_t cw = (uint32_t)(b >> (16 * w)) & 0xffffu;
            const size_t c = cw ? cw - 1 : p;  // an empty way points at p itself and is masked out below
            cpos[w] = c;
            const uint32_t ok = ((load64(in + c) & MASK5) == v) & (cw != 0) & (p - c <= 32768);
            hits |= ok << w;
        }
        dist_out = 0;
        if (__builtin_expect(hits == 0, 1)) return 0;
        int best = 0;
        const int maxlen = (int)std::min<size_t>(MAX_MATCH, n - p);
        for (int w = 0; w < WAYS; w++) {
            if (!(hits >> w & 1)) continue;
            const int dist = (int)(p - cpos[w]);
            const int l = match_len(in + cpos[w], in + p, maxlen);
            if (l > best && worth(l, dist)) {
                best = l;
                dist_out = dist;
            }
        }
        return best;
    }

    template <int WAYS, int HB> void parse(const uint8_t *in, size_t n) {
        memset(tab_, 0, sizeof(uint64_t) << HB);
        memset(lfreq_, 0, sizeof lfreq_);
        memset(dfreq_, 0, sizeof dfreq_);
        nsym_ = 0;
        extra_bits_ = 0;
        const size_t hash_end = n >= 8 ? n - 7 : 0;  // positions < hash_end can be hashed
        size_t p = 0;
        size_t miss = 0;  // positions probed in vain since the last match
        while (p < n) {
            int len = 0, dist = 0;
            if (p < hash_end) len = find<WAYS, HB>(in, n, p, dist);
            if (len >= MIN_MATCH) {
                miss = 0;
                bool probed = false;  // whether p + 1 is already in the table
                if (lazy_) {
                    // defer while the next position starts a strictly longer match
                    while (len < 40 && p + 1 < hash_end) {
                        int d2 = 0;
                        const int l2 = find<WAYS, HB>(in, n, p + 1, d2);
                        if (l2 <= len) { probed = true; break; }
                        put_literal(in[p]);
                        p++;
                        len = l2;
                        dist = d2;
                    }
                }
                put_match(len, dist);
                const size_t stop = std::min(p + (size_t)len, hash_end);
                for (size_t q = p + 1 + (probed ? 1 : 0); q < stop; q++) insert<HB>(in, q);
                p += (size_t)len;
            } else {
                put_literal(in[p]);
                p++;
                if (skip_after_) {
                    // Qualities and packed bases: nothing to find for hundreds of bytes, and every probe costs ~10 cycles.
                    // After skip_after_ probes in vain, each further one is followed by (misses / skip_after_, at most
                    // skip_cap_) literals that are neither probed nor entered into the table.
                    miss++;
                    size_t k = std::min<size_t>(miss / skip_after_, skip_cap_);
                    for (; k && p < n; k--) put_literal(in[p++]);
                }
            }
        }
        count_literals();
        lfreq_[256] = 1;  // end of block
    }

    // ---- efforts 1 and 2: one candidate per hash in a 16-bit table (32 KB: stays in L1), four positions probed per
    // round from ONE 8-byte load (its four 5-byte windows), one branch per round while nothing is found.  An empty
    // table entry is position 0, a candidate like any other: the 5-byte compare and the distance test (0 < dist <=
    // 32768 as one unsigned compare) decide.  The positions of a round do not see each other's insertions, so a match
    // at distance < 4 is found one round late (a run loses at most 3 bytes of its first match).
    template <int HB> inline int find16(const uint8_t *in, size_t n, size_t p, int &dist_out) {
        const uint64_t v = load64(in + p) & MASK5;
        const uint32_t h = hash5<HB>(v);
        const size_t c = tab16_[h];
        tab16_[h] = (uint16_t)p;
        dist_out = 0;
        if (__builtin_expect(((load64(in + c) & MASK5) != v) | (p - c - 1 >= 32768), 1)) return 0;
        const int l = match_len(in + c, in + p, (int)std::min<size_t>(MAX_MATCH, n - p));
        if (!worth(l, (int)(p - c))) return 0;
        dist_out = (int)(p - c);
        return l;
    }
    template <int HB> void parse_fast(const uint8_t *in, size_t n) {
        memset(tab16_, 0, sizeof(uint16_t) << HB);
        memset(lfreq_, 0, sizeof lfreq_);
        memset(dfreq_, 0, sizeof dfreq_);
        memset(lhist_, 0, sizeof lhist_);
        nsym_ = 0;
        extra_bits_ = 0;
        const size_t hash_end = n >= 8 ? n - 7 : 0;  // positions < hash_end can be hashed
        size_t p = 0, miss = 0, lit_start = 0;  // lit_start: first position after the last match
        size_t hi = 0, next_hint = n_hints_ ? hints_[0].pos : (size_t)-1, region_miss = 0;
        while (p < n) {
            // rounds of four probes while nothing is found
            while (p + 4 <= hash_end) {
                while (p >= next_hint) {  // the caller's layout hints
                    miss = region_miss = hints_[hi].miss == HINT_MILD ? (uint32_t)skip_after_ : hints_[hi].miss;
                    hi++;
                    next_hint = hi < n_hints_ ? hints_[hi].pos : (size_t)-1;
                }
                const uint64_t w = load64(in + p);
                const uint64_t v0 = w & MASK5, v1 = (w >> 8) & MASK5, v2 = (w >> 16) & MASK5, v3 = w >> 24;
                const uint32_t h0 = hash5<HB>(v0), h1 = hash5<HB>(v1), h2 = hash5<HB>(v2), h3 = hash5<HB>(v3);
                const size_t c0 = tab16_[h0], c1 = tab16_[h1], c2 = tab16_[h2], c3 = tab16_[h3];
                const uint32_t hit = (uint32_t)(((load64(in + c0) & MASK5) == v0) & (p - c0 - 1 < 32768)) |
                                     (uint32_t)(((load64(in + c1) & MASK5) == v1) & (p - c1 < 32768)) << 1 |
                                     (uint32_t)(((load64(in + c2) & MASK5) == v2) & (p + 1 - c2 < 32768)) << 2 |
                                     (uint32_t)(((load64(in + c3) & MASK5) == v3) & (p + 2 - c3 < 32768)) << 3;
                if (__builtin_expect(hit != 0, 0)) {
                    // the positions before the first hit are literals; the hit itself goes through find16 below
                    const int first = __builtin_ctz(hit);
                    const uint32_t hs[3] = {h0, h1, h2};
                    for (int j = 0; j < first; j++) {
                        tab16_[hs[j]] = (uint16_t)(p + (size_t)j);
                        put_literal_counted(in[p + (size_t)j], p + (size_t)j);
                    }
                    p += (size_t)first;
                    break;
                }
                tab16_[h0] = (uint16_t)p;
                tab16_[h1] = (uint16_t)(p + 1);
                tab16_[h2] = (uint16_t)(p + 2);
                tab16_[h3] = (uint16_t)(p + 3);
                // four literals: symbols as one 64-bit store, one histogram per position of the round
                const uint64_t s4 = (w & 0xff) | ((w & 0xff00) << 8) | ((w & 0xff0000) << 16) | ((w & 0xff000000ull) << 24);
                memcpy(sym_ + nsym_, &s4, 8);
                nsym_ += 4;
                lhist_[p & 3][w & 0xff]++;
                lhist_[(p + 1) & 3][(w >> 8) & 0xff]++;
                lhist_[(p + 2) & 3][(w >> 16) & 0xff]++;
                lhist_[(p + 3) & 3][(w >> 24) & 0xff]++;
                p += 4;
                if (skip_after_) {
                    // Qualities and packed bases: nothing to find for hundreds of bytes.  After skip_after_ probes in
                    // vain every probe is followed by (misses / skip_after_, at most skip_cap_) literals that are
                    // neither probed nor entered into the table.
                    miss += 4;
                    size_t k = 4 * std::min<size_t>(miss / skip_after_, skip_cap_);
                    k = std::min(k, n - p);
                    if (next_hint != (size_t)-1) k = std::min(k, next_hint > p ? next_hint - p : 0);  // never across a hint
                    for (size_t j = 0; j < k; j++) put_literal_counted(in[p + j], p + j);
                    p += k;
                }
            }
            if (p >= n) break;
            int len = 0, dist = 0;
            if (p < hash_end) len = find16<HB>(in, n, p, dist);
            if (len >= MIN_MATCH) {
                miss = region_miss = 0;  // (a repeat inside a hinted stretch: the hint was wrong for this one, e.g. runs of equal qualities)
                bool probed = false;  // whether p + 1 is already in the table
                if (lazy_) {
                    // defer while the next position starts a strictly longer match
                    while (len < 40 && p + 1 < hash_end) {
                        int d2 = 0;
                        const int l2 = find16<HB>(in, n, p + 1, d2);
                        if (l2 <= len) { probed = true; break; }
                        put_literal_counted(in[p], p);
                        p++;
                        len = l2;
                        dist = d2;
                    }
                }
                // catch up: a probe that landed inside a repeat (after skipped positions) extends the match backwards
                // over the literals already emitted since the last match
                size_t back = 0;
                while (p - back > lit_start && p - back > (size_t)dist && len + (int)back < MAX_MATCH &&
                       in[p - back - 1] == in[p - back - 1 - (size_t)dist]) back++;
                for (size_t j = 1; j <= back; j++) lhist_[(p - j) & 3][in[p - j]]--;
                nsym_ -= back;
                put_match(len + (int)back, dist);
                const size_t stop = std::min(p + (size_t)len, hash_end);
                for (size_t q = p + 1 + (probed ? 1 : 0); q < stop; q++) tab16_[hash5<HB>(load64(in + q) & MASK5)] = (uint16_t)q;
                p += (size_t)len;
                lit_start = p;
            } else {
                put_literal_counted(in[p], p);
                p++;
                miss++;
            }
        }
        for (int c = 0; c < 256; c++) lfreq_[c] = lhist_[0][c] + lhist_[1][c] + lhist_[2][c] + lhist_[3][c];
        lfreq_[256] = 1;  // end of block
    }
    inline void put_literal_counted(uint8_t c, size_t pos) {  // the histogram is chosen by position, so that catch-up can undo it
        sym_[nsym_++] = c;
        lhist_[pos & 3][c]++;
    }

    // symbol stream: a literal is its byte value; a match is 0x8000 | length followed by distance - 1.
    // Literal frequencies are counted afterwards (count_literals), off the parser's critical path.
    inline void put_literal(uint8_t c) { sym_[nsym_++] = c; }
    inline void put_match(int len, int dist) {
        sym_[nsym_++] = (uint16_t)(0x8000u | (uint32_t)len);
        sym_[nsym_++] = (uint16_t)(dist - 1);
        const int ls = len_sym_[len], ds = dist_sym(dist);
        lfreq_[257 + ls]++;
        dfreq_[ds]++;
        extra_bits_ += len_xbits_[ls] + dist_xbits_[ds];
    }
    void count_literals() {
        uint32_t h[4][256];
        memset(h, 0, sizeof h);
        size_t i = 0;
        int lane = 0;
        while (i < nsym_) {
            const uint32_t v = sym_[i];
            if (v & 0x8000u) { i += 2; continue; }
            h[lane & 3][v]++;
            lane++;
            i++;
        }
        for (int c = 0; c < 256; c++) lfreq_[c] = h[0][c] + h[1][c] + h[2][c] + h[3][c];
    }

    // ------------------------------------------------------------------ Huffman code construction
    // Code lengths (<= maxbits, complete prefix code) for freq[0, n); at least two symbols get a code.
    static void build_lengths(const uint32_t *freq_in, int n, int maxbits, uint8_t *lens) {
        uint32_t freq[288];
        int order[288];
        int m = 0;
        for (int i = 0; i < n; i++) {
            freq[i] = freq_in[i];
            lens[i] = 0;
        }
        for (int i = 0; i < n; i++)
            if (freq[i]) order[m++] = i;
        // decoders reject incomplete codes (except one 1-bit code): always give two symbols a code
        for (int i = 0; m < 2 && i < n; i++)
            if (!freq[i]) {
                freq[i] = 1;
                order[m++] = i;
            }
        std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        // two-queue Huffman: leaves 0..m-1 (ascending), internal nodes m..2m-2 are created in ascending weight order
        uint64_t w[576];
        int parent[576];
        for (int i = 0; i < m; i++) w[i] = freq[order[i]];
        int leaf = 0, inode = m, next = m;
        auto take = [&]() {
            if (leaf < m && (inode >= next || w[leaf] <= w[inode])) return leaf++;
            return inode++;
        };
        while (next < 2 * m - 1) {
            const int a = take(), b = take();
            w[next] = w[a] + w[b];
            parent[a] = parent[b] = next;
            next++;
        }
        int depth[576];
        depth[2 * m - 2] = 0;
        for (int i = 2 * m - 3; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
        // length limit: clamp, then repair the Kraft sum (leaf i = i-th least frequent symbol)
        int len[288];
        uint32_t kraft = 0;
        const uint32_t full = 1u << maxbits;
        for (int i = 0; i < m; i++) {
            len[i] = std::min(depth[i], maxbits);
            kraft += full >> len[i];
        }
        while (kraft > full) {  // over-subscribed: lengthen the rarest symbol that still can be
            int i = 0;
            while (len[i] >= maxbits) i++;
            kraft -= full >> (len[i] + 1);
            len[i]++;
        }
        while (kraft < full) {  // incomplete: shorten, most frequent first, whatever fits
            bool moved = false;
            for (int i = m - 1; i >= 0 && kraft < full; i--) {
                if (len[i] > 1 && kraft + (full >> len[i]) <= full) {
                    kraft += full >> len[i];
                    len[i]--;
                    moved = true;
                }
            }
            if (!moved) break;
        }
        for (int i = 0; i < m; i++) lens[order[i]] = (uint8_t)len[i];
    }

    // canonical codes (RFC 1951 §3.2.2), stored bit-reversed for LSB-first output
    static void assign_codes(const uint8_t *lens, int n, uint16_t *codes) {
        int bl_count[16] = {0};
        for (int i = 0; i < n; i++) bl_count[lens[i]]++;
        bl_count[0] = 0;
        uint32_t next_code[16];
        uint32_t code = 0;
        for (int b = 1; b <= 15; b++) {
            code = (code + (uint32_t)bl_count[b - 1]) << 1;
            next_code[b] = code;
        }
        for (int i = 0; i < n; i++) {
            const int l = lens[i];
            if (!l) { codes[i] = 0; continue; }
            uint32_t c = next_code[l]++, r = 0;
            for (int k = 0; k < l; k++) { r = (r << 1) | (c & 1); c >>= 1; }
            codes[i] = (uint16_t)r;
        }
    }

    // ------------------------------------------------------------------ static tables
    void init_static() {
        static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t LX[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t DX[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (int s = 0; s < 29; s++) {
            len_base_[s] = LBASE[s];
            len_xbits_[s] = LX[s];
            const int hi = s == 28 ? 258 : LBASE[s + 1] - 1;
            for (int l = LBASE[s]; l <= hi && l <= 258; l++) len_sym_[l] = (uint8_t)s;
        }
        len_sym_[258] = 28;
        for (int s = 0; s < 30; s++) {
            dist_base_[s] = DBASE[s];
            dist_xbits_[s] = DX[s];
        }
        // distance symbol lookup: d-1 < 256 direct, else by (d-1) >> 7
        for (int d = 1; d <= 32768; d++) {
            int s = 29;
            while (DBASE[s] > d) s--;
            if (d <= 256) dsym_lo_[d - 1] = (uint8_t)s;
            else dsym_hi_[(d - 1) >> 7] = (uint8_t)s;
        }
        for (int i = 0; i < 288; i++) fix_llen_[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
        for (int i = 0; i < 32; i++) fix_dlen_[i] = 5;
        assign_codes(fix_llen_, 288, fix_lcode_);
        assign_codes(fix_dlen_, 32, fix_dcode_);
    }
    inline int dist_sym(int dist) const { return dist <= 256 ? dsym_lo_[dist - 1] : dsym_hi_[(dist - 1) >> 7]; }

    // ------------------------------------------------------------------ bit output
    struct Bits {
        uint8_t *p;
        uint64_t buf = 0;
        int cnt = 0;
        inline void add(uint32_t v, int n) {
            buf |= (uint64_t)v << cnt;
            cnt += n;
            if (cnt >= 32) {
                const uint32_t w = (uint32_t)buf;
                memcpy(p, &w, 4);
                p += 4;
                buf >>= 32;
                cnt -= 32;
            }
        }
        // whole bytes out; the partial byte (tail_bits < 8 bits) stays in `tail` for the caller to continue from
        uint64_t tail = 0;
        int tail_bits = 0;
        inline void finish() {
            while (cnt >= 8) {
                *p++ = (uint8_t)buf;
                buf >>= 8;
                cnt -= 8;
            }
            tail = buf;
            tail_bits = cnt;
            buf = 0;
            cnt = 0;
        }
    };

    // ------------------------------------------------------------------ block encoding
    size_t encode(const uint8_t *in, size_t n, uint8_t *out) {
        uint8_t llen[288], dlen[32];
        build_lengths(lfreq_, 286, 15, llen);
        build_lengths(dfreq_, 30, 15, dlen);
        llen[286] = llen[287] = 0;
        dlen[30] = dlen[31] = 0;
        int hlit = 286, hdist = 30;
        while (hlit > 257 && llen[hlit - 1] == 0) hlit--;
        while (hdist > 1 && dlen[hdist - 1] == 0) hdist--;

        // code-length sequence, run-length coded with 16 / 17 / 18 (RFC 1951 §3.2.7)
        uint8_t seq[320];
        const int nseq = hlit + hdist;
        memcpy(seq, llen, (size_t)hlit);
        memcpy(seq + hlit, dlen, (size_t)hdist);
        uint8_t rl_sym[320], rl_x[320];
        int nrl = 0;
        uint32_t cfreq[19] = {0};
        for (int i = 0; i < nseq;) {
            const int v = seq[i];
            int run = 1;
            while (i + run < nseq && seq[i + run] == v) run++;
            i += run;
            if (v == 0) {
                while (run >= 11) {
                    const int r = std::min(run, 138);
                    rl_sym[nrl] = 18; rl_x[nrl++] = (uint8_t)(r - 11); cfreq[18]++;
                    run -= r;
                }
                if (run >= 3) {
                    rl_sym[nrl] = 17; rl_x[nrl++] = (uint8_t)(run - 3); cfreq[17]++;
                    run = 0;
                }
            } else {
                rl_sym[nrl] = (uint8_t)v; rl_x[nrl++] = 0; cfreq[v]++;
                run--;
                while (run >= 3) {
                    const int r = std::min(run, 6);
                    rl_sym[nrl] = 16; rl_x[nrl++] = (uint8_t)(r - 3); cfreq[16]++;
                    run -= r;
                }
            }
            while (run-- > 0) {
                rl_sym[nrl] = (uint8_t)v; rl_x[nrl++] = 0; cfreq[v]++;
            }
        }
        uint8_t clen[19];
        uint16_t ccode[19];
        build_lengths(cfreq, 19, 7, clen);
        assign_codes(clen, 19, ccode);
        static const uint8_t CORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && clen[CORDER[hclen - 1]] == 0) hclen--;

        // costs in bits
        uint64_t dyn = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen + extra_bits_;
        for (int i = 0; i < nrl; i++) dyn += clen[rl_sym[i]] + (rl_sym[i] == 16 ? 2 : rl_sym[i] == 17 ? 3 : rl_sym[i] == 18 ? 7 : 0);
        uint64_t fix = 3 + extra_bits_;
        for (int i = 0; i < 286; i++) {
            dyn += (uint64_t)lfreq_[i] * llen[i];
            fix += (uint64_t)lfreq_[i] * fix_llen_[i];
        }
        for (int i = 0; i < 30; i++) {
            dyn += (uint64_t)dfreq_[i] * dlen[i];
            fix += (uint64_t)dfreq_[i] * 5;
        }
        const uint64_t stored = 8 * ((uint64_t)n + 5);

        if (stored <= dyn && stored <= fix) {
            out[0] = 1;  // BFINAL = 1, BTYPE = 00, then byte-aligned LEN / NLEN
            const uint16_t l = (uint16_t)n, nl = (uint16_t)~l;
            memcpy(out + 1, &l, 2);
            memcpy(out + 3, &nl, 2);
            memcpy(out + 5, in, n);
            return n + 5;
        }
        Bits bw;
        bw.p = out;
        const uint8_t *ll;
        const uint16_t *lc, *dc;
        const uint8_t *dl;
        uint16_t lcode[288], dcode[32];
        if (fix <= dyn) {
            bw.add(1 | (1 << 1), 3);
            ll = fix_llen_; lc = fix_lcode_; dl = fix_dlen_; dc = fix_dcode_;
        } else {
            bw.add(1 | (2 << 1), 3);
            bw.add((uint32_t)(hlit - 257), 5);
            bw.add((uint32_t)(hdist - 1), 5);
            bw.add((uint32_t)(hclen - 4), 4);
            for (int i = 0; i < hclen; i++) bw.add(clen[CORDER[i]], 3);
            for (int i = 0; i < nrl; i++) {
                const int s = rl_sym[i];
                bw.add(ccode[s], clen[s]);
                if (s == 16) bw.add(rl_x[i], 2);
                else if (s == 17) bw.add(rl_x[i], 3);
                else if (s == 18) bw.add(rl_x[i], 7);
            }
            assign_codes(llen, 288, lcode);
            assign_codes(dlen, 32, dcode);
            ll = llen; lc = lcode; dl = dlen; dc = dcode;
        }
        bw.finish();  // the header, to a byte boundary at most 7 bits short: the symbol loop has its own 64-bit writer
        // Symbols: code and length of a literal / length symbol come from one 32-bit table entry; the bit buffer is
        // flushed without a branch (8 bytes stored, the pointer advanced by the whole bytes), after every three literals
        // (<= 45 bits) or one match (<= 48 bits) on top of the <= 7 bits left over.
        uint32_t lt[288];
        for (int i = 0; i < 288; i++) lt[i] = (uint32_t)lc[i] | ((uint32_t)ll[i] << 16);
        uint8_t *op = bw.p;
        uint64_t buf = bw.tail;
        int cnt = bw.tail_bits;
        auto flush = [&]() {
            memcpy(op, &buf, 8);
            op += cnt >> 3;
            buf >>= (cnt & ~7);
            cnt &= 7;
        };
        sym_[nsym_] = 0x8000u | 0x7fffu;  // sentinel: stops the literal pairing at the end
        for (size_t i = 0; i < nsym_;) {
            const uint32_t v = sym_[i];
            if (!(v & 0x8000u)) {
                const uint32_t e0 = lt[v];
                buf |= (uint64_t)(e0 & 0xffffu) << cnt;
                cnt += (int)(e0 >> 16);
                i++;
                const uint32_t v1 = sym_[i];
                if (!(v1 & 0x8000u)) {
                    const uint32_t e1 = lt[v1];
                    buf |= (uint64_t)(e1 & 0xffffu) << cnt;
                    cnt += (int)(e1 >> 16);
                    i++;
                    const uint32_t v2 = sym_[i];
                    if (!(v2 & 0x8000u)) {
                        const uint32_t e2 = lt[v2];
                        buf |= (uint64_t)(e2 & 0xffffu) << cnt;
                        cnt += (int)(e2 >> 16);
                        i++;
                    }
                }
                flush();
            } else {
                const int len = (int)(v & 0x7fffu), dist = (int)sym_[i + 1] + 1;
                i += 2;
                const int ls = len_sym_[len];
                const uint32_t e = lt[257 + ls];
                buf |= (uint64_t)(e & 0xffffu) << cnt;
                cnt += (int)(e >> 16);
                buf |= (uint64_t)(uint32_t)(len - len_base_[ls]) << cnt;
                cnt += len_xbits_[ls];
                const int ds = dist_sym(dist);
                buf |= (uint64_t)dc[ds] << cnt;
                cnt += dl[ds];
                buf |= (uint64_t)(uint32_t)(dist - dist_base_[ds]) << cnt;
                cnt += dist_xbits_[ds];
                flush();
            }
        }
        buf |= (uint64_t)lc[256] << cnt;
        cnt += ll[256];
        flush();
        if (cnt) *op++ = (uint8_t)buf;
        return (size_t)(op - out);
    }

    const Hint *hints_ = nullptr;
    size_t n_hints_ = 0;
    size_t skip_after_ = 0, skip_cap_ = 0;  // 0: every position is probed
    int effort_;
    bool lazy_;
    uint64_t tab_[1 << 15];
    uint16_t tab16_[1 << 14];
    uint32_t lhist_[4][256];
    uint16_t sym_[65536 + 8];
    size_t nsym_ = 0;
    uint64_t extra_bits_ = 0;
    uint32_t lfreq_[288], dfreq_[32];
    uint16_t len_base_[29], dist_base_[30];
    uint8_t len_xbits_[29], dist_xbits_[30];
    uint8_t len_sym_[259];
    uint8_t dsym_lo_[256], dsym_hi_[256];
    uint8_t fix_llen_[288], fix_dlen_[32];
    uint16_t fix_lcode_[288], fix_dcode_[32];
};

}  // namespace htsl
