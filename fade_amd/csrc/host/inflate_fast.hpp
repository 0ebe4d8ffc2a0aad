// inflate_fast.hpp — a DEFLATE (RFC 1951) decoder for BGZF blocks, the counterpart of deflate_fast.hpp.
//
// A BGZF block is a complete <= 64 KiB DEFLATE stream whose uncompressed size is known up front (ISIZE), so the
// decoder writes straight into the caller's buffer with exact bounds: 64-bit bit buffer refilled eight bytes at a
// time, 11-bit primary / secondary tables for the literal-length code, 8-bit for the distance code, word-wise match
// copies.  A primary entry whose index holds TWO whole literal codes yields both at once (BAM payload is literals mostly —
// packed bases and qualities at 4-6 bits a code — so a look-up, the unit of the decoder's dependent chain, then moves two
// bytes).  Every table index, back-reference and output position is checked: corrupt input returns false, it never
// reads or writes outside [in, in + in_len) / [out, out + out_len) (the caller's BgzfIn also checks the CRC32).
//
// Not derived from zlib/libdeflate sources; follows RFC 1951 only.
#pragma once
#include <cstdint>
#include <cstring>

namespace htsl {

class FastInflate {
public:
    // Decodes one DEFLATE stream into exactly out_len bytes.  false: malformed stream or size mismatch.
    bool inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
        begin(in, in_len, out, out_len);
        for (;;) {
            const int r = next_block();
            if (r < 0) return false;
            if (r == 0) return op_ == oend_;
            if (!decode_block()) return false;
        }
    }

    // Two independent streams at once (two BGZF blocks).  Huffman decoding is a chain of dependent table look-ups
    // (shift, mask, load: ~8 cycles per symbol whatever the core's width); while both streams are inside a Huffman block
    // their literals are decoded in lockstep, so the two chains overlap.  Same checks and results as two inflate() calls.
    static bool inflate2(FastInflate &A, const uint8_t *in_a, size_t in_len_a, uint8_t *out_a, size_t out_len_a,
                         FastInflate &B, const uint8_t *in_b, size_t in_len_b, uint8_t *out_b, size_t out_len_b) {
        A.begin(in_a, in_len_a, out_a, out_len_a);
        B.begin(in_b, in_len_b, out_b, out_len_b);
        int ra = A.next_block(), rb = B.next_block();
        while (ra == 1 && rb == 1) {
            const int f = pair_loop(A, B);
            if (f & F_ERROR) return false;
            if (f & F_A_EOB) ra = A.next_block();
            else if (f & F_A_CAREFUL) ra = A.decode_block() ? A.next_block() : -1;
            if (f & F_B_EOB) rb = B.next_block();
            else if (f & F_B_CAREFUL) rb = B.decode_block() ? B.next_block() : -1;
        }
        while (ra == 1) ra = A.decode_block() ? A.next_block() : -1;
        while (rb == 1) rb = B.decode_block() ? B.next_block() : -1;
        return ra == 0 && rb == 0 && A.op_ == A.oend_ && B.op_ == B.oend_;
    }

private:
    void begin(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
        ip_ = in;
        iend_ = in + in_len;
        bb_ = 0;
        bc_ = 0;
        out_ = op_ = out;
        oend_ = out + out_len;
        final_seen_ = false;
    }
    // Reads block headers (and copies stored blocks) up to the next Huffman block: 1 = inside one, its tables in lit_ /
    // dist_; 0 = the final block is behind us; -1 = malformed.
    int next_block() {
        for (;;) {
            if (final_seen_) return 0;
            refill();
            if (bc_ < 3) return -1;
            const uint32_t bfinal = (uint32_t)bb_ & 1u, btype = ((uint32_t)bb_ >> 1) & 3u;
            drop(3);
            final_seen_ = bfinal != 0;
            if (btype == 0) {
                drop(bc_ & 7);  // to the byte boundary
                refill();
                if (bc_ < 32) return -1;
                const uint32_t len = (uint32_t)bb_ & 0xffffu, nlen = ((uint32_t)(bb_ >> 16)) & 0xffffu;
                if ((len ^ nlen) != 0xffffu) return -1;
                drop(32);
                ip_ -= bc_ >> 3;  // whole bytes still in the bit buffer go back to the byte stream
                bb_ = 0;
                bc_ = 0;
                if ((size_t)(iend_ - ip_) < len || (size_t)(oend_ - op_) < len) return -1;
                memcpy(op_, ip_, len);
                op_ += len;
                ip_ += len;
            } else if (btype == 1) {
                if (!fixed_built_) {
                    uint8_t l[288 + 32];
                    for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
                    for (int i = 0; i < 32; i++) l[288 + i] = 5;
                    if (!build(l, 288, LIT_BITS, fix_lit_, FIX_LIT_SIZE, true) || !build(l + 288, 32, DIST_BITS, fix_dist_, FIX_DIST_SIZE, false))
                        return -1;
                    fixed_built_ = true;
                }
                lit_ = fix_lit_;
                dist_ = fix_dist_;
                return 1;
            } else if (btype == 2) {
                if (!read_dynamic()) return -1;
                lit_ = dyn_lit_;
                dist_ = dyn_dist_;
                return 1;
            } else
                return -1;
        }
    }

    static constexpr int LIT_BITS = 11, DIST_BITS = 8;
    // entry: bits 0-7 code length to consume (sub-table pointer: primary bits), 8-11 kind, 12-15 extra bits or
    // sub-table index bits (a literal entry: how many literals it holds, 1 or 2), 16-31 literal value(s, the first in the
    // low byte) / base / sub-table offset
    static constexpr uint32_t K_LIT = 1u << 8, K_LEN = 2u << 8, K_EOB = 3u << 8, K_SUB = 4u << 8, K_DIST = 5u << 8, K_MASK = 15u << 8;
    static constexpr int DYN_LIT_SIZE = (1 << LIT_BITS) + 32 * 288, DYN_DIST_SIZE = (1 << DIST_BITS) + 128 * 32;
    static constexpr int FIX_LIT_SIZE = (1 << LIT_BITS) + 32, FIX_DIST_SIZE = (1 << DIST_BITS) + 32;

    static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
    // the literal(s) of entry e: both bytes are stored (the second is scratch when the entry holds one; callers keep room
    // for it), the pointer moves by the entry's count
    static inline __attribute__((always_inline)) void put2(uint8_t *&op, uint32_t e) {
        const uint16_t v = (uint16_t)(e >> 16);
        memcpy(op, &v, 2);
        op += (e >> 12) & 3u;
    }
    inline void refill() {
        if (iend_ - ip_ >= 8) {
            bb_ |= load64(ip_) << bc_;
            ip_ += (63 - bc_) >> 3;
            bc_ |= 56;
        } else {
            while (bc_ <= 56 && ip_ < iend_) {
                bb_ |= (uint64_t)*ip_++ << bc_;
                bc_ += 8;
            }
        }
    }
    inline void drop(int n) {
        bb_ >>= n;
        bc_ -= n;
    }
    inline uint32_t take(int n) {  // n <= 16, caller guarantees bc_ >= n (checked through need())
        const uint32_t v = (uint32_t)bb_ & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    inline bool need(int n) {
        if (bc_ < n) refill();
        return bc_ >= n;
    }

    static uint32_t rev(uint32_t c, int n) {
        uint32_t r = 0;
        for (int i = 0; i < n; i++) { r = (r << 1) | (c & 1); c >>= 1; }
        return r;
    }

    // Decoding table for the canonical code with lengths l[0, n).  Codes longer than `primary` go through sub-tables
    // of (longest - primary) bits.  Rejects over-subscribed and (unless it is the lone-code case) incomplete codes.
    bool build(const uint8_t *l, int n, int primary, uint32_t *tab, int cap, bool litlen) {
        static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t LX[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t DX[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        int count[16] = {0};
        int longest = 0, used = 0;
        for (int i = 0; i < n; i++) {
            count[l[i]]++;
            if (l[i]) { used++; if (l[i] > longest) longest = l[i]; }
        }
        const int psize = 1 << primary;
        for (int i = 0; i < psize; i++) tab[i] = 0;  // 0 = invalid entry
        if (used == 0) return !litlen;               // a block without matches may declare no distance code
        // Kraft sum
        long left = 1;
        for (int b = 1; b <= 15; b++) {
            left <<= 1;
            left -= count[b];
            if (left < 0) return false;
        }
        if (left > 0 && !(used == 1 && longest == 1)) return false;  // incomplete (a lone 1-bit code is allowed)
        const int sub_bits = longest > primary ? longest - primary : 0;
        int next = psize;
        uint32_t code = 0;
        for (int len = 1; len <= 15; len++) {
            for (int sym = 0; sym < n; sym++) {
                if (l[sym] != len) continue;
                uint32_t e;
                if (litlen) {
                    if (sym < 256) e = ((uint32_t)sym << 16) | (1u << 12) | K_LIT;
                    else if (sym == 256) e = K_EOB;
                    else if (sym < 286) e = ((uint32_t)LBASE[sym - 257] << 16) | ((uint32_t)LX[sym - 257] << 12) | K_LEN;
                    else e = 0;  // 286, 287 never appear in valid data: decoding them fails
                } else {
                    e = sym < 30 ? (((uint32_t)DBASE[sym] << 16) | ((uint32_t)DX[sym] << 12) | K_DIST) : 0;
                }
                const uint32_t r = rev(code, len);
                if (len <= primary) {
                    if (e) e |= (uint32_t)len;
                    for (uint32_t i = r; i < (uint32_t)psize; i += 1u << len) tab[i] = e;
                } else {
                    const uint32_t pre = r & (uint32_t)(psize - 1);
                    if ((tab[pre] & K_MASK) != K_SUB) {
                        if (next + (1 << sub_bits) > cap) return false;
                        tab[pre] = ((uint32_t)next << 16) | ((uint32_t)sub_bits << 12) | K_SUB | (uint32_t)primary;
                        for (int i = 0; i < (1 << sub_bits); i++) tab[next + i] = 0;
                        next += 1 << sub_bits;
                    }
                    const uint32_t off = tab[pre] >> 16, hi = r >> primary;
                    if (e) e |= (uint32_t)(len - primary);  // the primary bits were consumed with the pointer
                    for (uint32_t i = hi; i < (1u << sub_bits); i += 1u << (len - primary)) tab[off + i] = e;
                }
                code++;
            }
            code <<= 1;
        }
        if (litlen) {
            // two literals per entry where the index holds both codes whole: entry i = literal of l1 bits, and the bits
            // behind it (i >> l1, whose top l1 bits are not the stream's) select a literal of at most primary - l1 bits
            uint32_t *const two = pair_tmp_;
            for (int i = 0; i < psize; i++) {
                const uint32_t e = tab[i];
                two[i] = e;
                if ((e & K_MASK) != K_LIT) continue;
                const uint32_t l1 = e & 255u;
                if (l1 >= (uint32_t)primary) continue;
                const uint32_t f = tab[(uint32_t)i >> l1];
                if ((f & K_MASK) != K_LIT || (f & 255u) + l1 > (uint32_t)primary) continue;
                two[i] = (((f >> 16) & 255u) << 24) | (e & 0x00ff0000u) | (2u << 12) | K_LIT | (l1 + (f & 255u));
            }
            for (int i = 0; i < psize; i++) tab[i] = two[i];
        }
        return true;
    }

    bool read_dynamic() {
        if (!need(14)) return false;
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) {
            if (!need(3)) return false;
            cl[ORDER[i]] = (uint8_t)take(3);
        }
        uint32_t ctab[1 << 7];
        if (!build_small(cl, ctab)) return false;
        uint8_t lens[286 + 30 + 138];
        int i = 0;
        const int total = hlit + hdist;
        while (i < total) {
            if (!need(7 + 7)) { if (bc_ < 1) return false; }
            const uint32_t e = ctab[(uint32_t)bb_ & 127u];
            const int clen = (int)(e & 15u), sym = (int)(e >> 4);
            if (clen == 0 || clen > bc_) return false;
            drop(clen);
            if (sym < 16) lens[i++] = (uint8_t)sym;
            else {
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0 || bc_ < 2) return false;
                    val = lens[i - 1];
                    rep = 3 + (int)take(2);
                } else if (sym == 17) {
                    if (bc_ < 3) return false;
                    rep = 3 + (int)take(3);
                } else {
                    if (bc_ < 7) return false;
                    rep = 11 + (int)take(7);
                }
                if (i + rep > total) return false;
                while (rep--) lens[i++] = (uint8_t)val;
            }
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        return build(lens, hlit, LIT_BITS, dyn_lit_, DYN_LIT_SIZE, true) && build(lens + hlit, hdist, DIST_BITS, dyn_dist_, DYN_DIST_SIZE, false);
    }
    // code-length code: at most 7 bits, one flat table; entry = sym << 4 | length (0 = invalid)
    static bool build_small(const uint8_t *cl, uint32_t *tab) {
        int count[8] = {0}, used = 0;
        for (int i = 0; i < 19; i++) {
            if (cl[i] > 7) return false;
            count[cl[i]]++;
            if (cl[i]) used++;
        }
        for (int i = 0; i < 128; i++) tab[i] = 0;
        if (!used) return false;
        long left = 1;
        for (int b = 1; b <= 7; b++) {
            left <<= 1;
            left -= count[b];
            if (left < 0) return false;
        }
        if (left > 0 && !(used == 1 && count[1] == 1)) return false;
        uint32_t code = 0;
        for (int len = 1; len <= 7; len++) {
            for (int sym = 0; sym < 19; sym++)
                if (cl[sym] == len) {
                    const uint32_t r = rev(code, len);
                    for (uint32_t i = r; i < 128; i += 1u << len) tab[i] = ((uint32_t)sym << 4) | (uint32_t)len;
                    code++;
                }
            code <<= 1;
        }
        return true;
    }

    // the rest of the current Huffman block, from wherever the stream stands
    bool decode_block() {
        uint8_t *op = op_;
        uint8_t *const out = out_, *const oend = oend_;
        const uint32_t *lit = lit_, *dist = dist_;
        // Fast loop while a whole worst-case step fits on both sides: >= 8 input bytes for the word refill, room for
        // three literals and a 258-byte match with its 8-byte copy overshoot.  One refill (>= 56 bits) serves up to
        // three literals (<= 45 bits); a length / distance pair (<= 48 bits) refills first if fewer are left.
        while (iend_ - ip_ >= 8 && oend - op >= 7 + 258 + 8) {
            bb_ |= load64(ip_) << bc_;
            ip_ += (63 - bc_) >> 3;
            bc_ |= 56;
            uint32_t e = lit[(uint32_t)bb_ & ((1u << LIT_BITS) - 1u)];
            if ((e & K_MASK) == K_LIT) {  // (one or two literals: both bytes are stored, the pointer moves by the count)
                drop((int)(e & 255u));
                put2(op, e);
                e = lit[(uint32_t)bb_ & ((1u << LIT_BITS) - 1u)];
                if ((e & K_MASK) == K_LIT) {
                    drop((int)(e & 255u));
                    put2(op, e);
                    e = lit[(uint32_t)bb_ & ((1u << LIT_BITS) - 1u)];
                    if ((e & K_MASK) == K_LIT) {
                        drop((int)(e & 255u));
                        put2(op, e);
                        continue;
                    }
                }
            }
            if ((e & K_MASK) == K_SUB) {
                drop(LIT_BITS);
                e = lit[(e >> 16) + ((uint32_t)bb_ & ((1u << ((e >> 12) & 15u)) - 1u))];
            }
            const int cl = (int)(e & 255u);
            if (cl == 0) return false;
            drop(cl);
            const uint32_t kind = e & K_MASK;
            if (kind == K_LIT) {  // (a sub-table literal: always one)
                put2(op, e);
                continue;
            }
            if (kind == K_EOB) {
                op_ = op;
                return true;
            }
            if (kind != K_LEN) return false;
            if (bc_ < 48) {  // here ip_ may have < 8 bytes left: refill() picks the right path
                refill();
            }
            const int lx = (int)((e >> 12) & 15u);
            if (lx > bc_) return false;
            const size_t len = (e >> 16) + ((uint32_t)bb_ & ((1u << lx) - 1u));
            drop(lx);
            uint32_t d = dist[(uint32_t)bb_ & ((1u << DIST_BITS) - 1u)];
            if ((d & K_MASK) == K_SUB) {
                drop(DIST_BITS);
                d = dist[(d >> 16) + ((uint32_t)bb_ & ((1u << ((d >> 12) & 15u)) - 1u))];
            }
            const int dl = (int)(d & 255u);
            if (dl == 0 || dl > bc_ || (d & K_MASK) != K_DIST) return false;
            drop(dl);
            const int dx = (int)((d >> 12) & 15u);
            if (dx > bc_) return false;
            const size_t distance = (d >> 16) + ((uint32_t)bb_ & ((1u << dx) - 1u));
            drop(dx);
            if (distance > (size_t)(op - out)) return false;
            const uint8_t *src = op - distance;
            uint8_t *dst = op;
            const uint8_t *const stop = op + len;
            if (distance >= 8) {
                do {
                    memcpy(dst, src, 8);
                    dst += 8;
                    src += 8;
                } while (dst < stop);
            } else if (distance == 1) {
                memset(dst, *src, len);
            } else {
                do { *dst++ = *src++; } while (dst < stop);
            }
            op += len;
        }
        for (;;) {
            refill();
            uint32_t e = lit[(uint32_t)bb_ & ((1u << LIT_BITS) - 1u)];
            if ((e & K_MASK) == K_SUB) {
                drop(LIT_BITS);
                e = lit[(e >> 16) + ((uint32_t)bb_ & ((1u << ((e >> 12) & 15u)) - 1u))];
            }
            const int cl = (int)(e & 255u);
            if (cl == 0 || cl > bc_) return false;  // invalid code or input exhausted
            drop(cl);
            const uint32_t kind = e & K_MASK;
            if (kind == K_LIT) {
                if (op == oend) return false;
                *op++ = (uint8_t)(e >> 16);
                if (((e >> 12) & 3u) == 2u) {
                    if (op == oend) return false;
                    *op++ = (uint8_t)(e >> 24);
                }
                continue;
            }
            if (kind == K_EOB) break;
            if (kind != K_LEN) return false;
            const int lx = (int)((e >> 12) & 15u);
            if (lx > bc_) return false;
            const size_t len = (e >> 16) + ((uint32_t)bb_ & ((1u << lx) - 1u));
            drop(lx);
            if (bc_ < 32) refill();  // a distance code takes up to 15 + 13 bits
            uint32_t d = dist[(uint32_t)bb_ & ((1u << DIST_BITS) - 1u)];
            if ((d & K_MASK) == K_SUB) {
                drop(DIST_BITS);
                d = dist[(d >> 16) + ((uint32_t)bb_ & ((1u << ((d >> 12) & 15u)) - 1u))];
            }
            const int dl = (int)(d & 255u);
            if (dl == 0 || dl > bc_ || (d & K_MASK) != K_DIST) return false;
            drop(dl);
            const int dx = (int)((d >> 12) & 15u);
            if (dx > bc_) return false;
            const size_t distance = (d >> 16) + ((uint32_t)bb_ & ((1u << dx) - 1u));
            drop(dx);
            if (distance > (size_t)(op - out) || len > (size_t)(oend - op)) return false;
            const uint8_t *src = op - distance;
            if (distance >= 8 && (size_t)(oend - op) >= len + 8) {
                uint8_t *dst = op;
                const uint8_t *const stop = op + len;
                do {  // may write up to 7 bytes past `len`, inside the output (checked above), later overwritten
                    memcpy(dst, src, 8);
                    dst += 8;
                    src += 8;
                } while (dst < stop);
            } else {
                for (size_t k = 0; k < len; k++) op[k] = src[k];
            }
            op += len;
        }
        op_ = op;
        return true;
    }

    // ---- two streams in lockstep (inflate2)
    enum { F_A_EOB = 1, F_B_EOB = 2, F_A_CAREFUL = 4, F_B_CAREFUL = 8, F_ERROR = 16 };
    struct Cur {  // a stream's hot state in registers
        uint64_t bb;
        int bc;
        const uint8_t *ip;
        uint8_t *op;
    };
    // one symbol whose primary entry is e (not yet consumed), with everything checked as in decode_block's fast loop:
    // 0 = done, 1 = end of block, -1 = malformed
    static inline __attribute__((always_inline)) int pair_symbol(Cur &c, uint32_t e, const uint32_t *lit, const uint32_t *dist,
                                                                const uint8_t *out, const uint8_t *iend) {
        if ((e & K_MASK) == K_SUB) {
            c.bb >>= LIT_BITS;
            c.bc -= LIT_BITS;
            e = lit[(e >> 16) + ((uint32_t)c.bb & ((1u << ((e >> 12) & 15u)) - 1u))];
        }
        const int cl = (int)(e & 255u);
        if (cl == 0) return -1;
        c.bb >>= cl;
        c.bc -= cl;
        const uint32_t kind = e & K_MASK;
        if (kind == K_LIT) {
            put2(c.op, e);
            return 0;
        }
        if (kind == K_EOB) return 1;
        if (kind != K_LEN) return -1;
        if (c.bc < 48) {  // (the pair loop guarantees >= 8 input bytes at its top; a refill moves ip by at most 7)
            if (iend - c.ip >= 8) {
                c.bb |= load64(c.ip) << c.bc;
                c.ip += (63 - c.bc) >> 3;
                c.bc |= 56;
            } else {
                while (c.bc <= 56 && c.ip < iend) {
                    c.bb |= (uint64_t)*c.ip++ << c.bc;
                    c.bc += 8;
                }
            }
        }
        const int lx = (int)((e >> 12) & 15u);
        if (lx > c.bc) return -1;
        const size_t len = (e >> 16) + ((uint32_t)c.bb & ((1u << lx) - 1u));
        c.bb >>= lx;
        c.bc -= lx;
        uint32_t d = dist[(uint32_t)c.bb & ((1u << DIST_BITS) - 1u)];
        if ((d & K_MASK) == K_SUB) {
            c.bb >>= DIST_BITS;
            c.bc -= DIST_BITS;
            d = dist[(d >> 16) + ((uint32_t)c.bb & ((1u << ((d >> 12) & 15u)) - 1u))];
        }
        const int dl = (int)(d & 255u);
        if (dl == 0 || dl > c.bc || (d & K_MASK) != K_DIST) return -1;
        c.bb >>= dl;
        c.bc -= dl;
        const int dx = (int)((d >> 12) & 15u);
        if (dx > c.bc) return -1;
        const size_t distance = (d >> 16) + ((uint32_t)c.bb & ((1u << dx) - 1u));
        c.bb >>= dx;
        c.bc -= dx;
        if (distance > (size_t)(c.op - out)) return -1;
        const uint8_t *src = c.op - distance;
        uint8_t *dst = c.op;
        const uint8_t *const stop = c.op + len;
        if (distance >= 8) {
            do {
                memcpy(dst, src, 8);
                dst += 8;
                src += 8;
            } while (dst < stop);
        } else if (distance == 1) {
            memset(dst, *src, len);
        } else {
            do { *dst++ = *src++; } while (dst < stop);
        }
        c.op += len;
        return 0;
    }
    // Runs while both streams have a whole worst-case step of room on both sides (as decode_block's fast loop); returns
    // which stream reached its end of block or has to continue in decode_block's careful loop.
    static int pair_loop(FastInflate &A, FastInflate &B) {
        Cur a{A.bb_, A.bc_, A.ip_, A.op_}, b{B.bb_, B.bc_, B.ip_, B.op_};
        const uint32_t *const la = A.lit_, *const lb = B.lit_;
        constexpr uint32_t M = (1u << LIT_BITS) - 1u;
        int flags = 0;
        for (;;) {
            if (!(A.iend_ - a.ip >= 8 && A.oend_ - a.op >= 7 + 258 + 8)) flags |= F_A_CAREFUL;
            if (!(B.iend_ - b.ip >= 8 && B.oend_ - b.op >= 7 + 258 + 8)) flags |= F_B_CAREFUL;
            if (flags) break;
            a.bb |= load64(a.ip) << a.bc;
            a.ip += (63 - a.bc) >> 3;
            a.bc |= 56;
            b.bb |= load64(b.ip) << b.bc;
            b.ip += (63 - b.bc) >> 3;
            b.bc |= 56;
            uint32_t ea = la[(uint32_t)a.bb & M], eb = lb[(uint32_t)b.bb & M];
            // up to three look-ups (six literals) from each stream per refill (<= 3 x 15 of >= 56 bits), the two chains side by side
            if (((ea & K_MASK) == K_LIT) & ((eb & K_MASK) == K_LIT)) {
                a.bb >>= (ea & 63u); a.bc -= (int)(ea & 255u); put2(a.op, ea);
                b.bb >>= (eb & 63u); b.bc -= (int)(eb & 255u); put2(b.op, eb);
                ea = la[(uint32_t)a.bb & M];
                eb = lb[(uint32_t)b.bb & M];
                if (((ea & K_MASK) == K_LIT) & ((eb & K_MASK) == K_LIT)) {
                    a.bb >>= (ea & 63u); a.bc -= (int)(ea & 255u); put2(a.op, ea);
                    b.bb >>= (eb & 63u); b.bc -= (int)(eb & 255u); put2(b.op, eb);
                    ea = la[(uint32_t)a.bb & M];
                    eb = lb[(uint32_t)b.bb & M];
                    if (((ea & K_MASK) == K_LIT) & ((eb & K_MASK) == K_LIT)) {
                        a.bb >>= (ea & 63u); a.bc -= (int)(ea & 255u); put2(a.op, ea);
                        b.bb >>= (eb & 63u); b.bc -= (int)(eb & 255u); put2(b.op, eb);
                        continue;
                    }
                }
            }
            // at least one of the two is not a literal: one symbol of each the long way
            const int sa = pair_symbol(a, ea, la, A.dist_, A.out_, A.iend_);
            const int sb = pair_symbol(b, eb, lb, B.dist_, B.out_, B.iend_);
            if ((sa < 0) | (sb < 0)) { flags = F_ERROR; break; }
            if (sa) flags |= F_A_EOB;
            if (sb) flags |= F_B_EOB;
            if (flags) break;
        }
        A.bb_ = a.bb; A.bc_ = a.bc; A.ip_ = a.ip; A.op_ = a.op;
        B.bb_ = b.bb; B.bc_ = b.bc; B.ip_ = b.ip; B.op_ = b.op;
        return flags;
    }

    const uint8_t *ip_ = nullptr, *iend_ = nullptr;
    uint8_t *out_ = nullptr, *op_ = nullptr, *oend_ = nullptr;
    bool final_seen_ = false;
    uint64_t bb_ = 0;
    int bc_ = 0;
    const uint32_t *lit_ = nullptr, *dist_ = nullptr;
    bool fixed_built_ = false;
    uint32_t pair_tmp_[1 << LIT_BITS];
    uint32_t dyn_lit_[DYN_LIT_SIZE], dyn_dist_[DYN_DIST_SIZE];
    uint32_t fix_lit_[FIX_LIT_SIZE], fix_dist_[FIX_DIST_SIZE];
};

}  // namespace htsl
