// Round-trip and size/speed check of deflate_fast.hpp and inflate_fast.hpp against zlib:  deflate_selftest [file]
// Every block is compressed by FastDeflate (all effort levels), inflated by zlib and compared; with a file argument
// the file is cut into 0xff00-byte BGZF-sized blocks and the sizes and rates of both compressors are printed.
#include <zlib.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "../crc32_fast.hpp"
#include "../deflate_fast.hpp"
#include "../inflate_fast.hpp"

using namespace htsl;

static bool roundtrip(FastDeflate &fd, const uint8_t *in, size_t n, size_t *clen_out) {
    std::vector<uint8_t> out(FastDeflate::bound(n) + 16), back(n + 1);
    const size_t clen = fd.compress(in, n, out.data());
    if (clen > FastDeflate::bound(n)) { fprintf(stderr, "bound exceeded: %zu > %zu\n", clen, FastDeflate::bound(n)); return false; }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, -15);
    zs.next_in = out.data();
    zs.avail_in = (uInt)clen;
    zs.next_out = back.data();
    zs.avail_out = (uInt)back.size();
    const int rc = inflate(&zs, Z_FINISH);
    const size_t got = zs.total_out, used = zs.total_in;
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || got != n || used != clen || (n && memcmp(back.data(), in, n) != 0)) {
        fprintf(stderr, "round trip failed: rc=%d (%s) got=%zu want=%zu used=%zu clen=%zu\n", rc, zs.msg ? zs.msg : "", got, n, used, clen);
        return false;
    }
    if (clen_out) *clen_out = clen;
    // layout hints (what the BGZF writer passes for BAM records) change which positions are probed, never the bytes that
    // come back: random stretches marked as free of repeats, right or wrong
    {
        static uint64_t lcg = 424242;
        std::vector<FastDeflate::Hint> hints;
        uint32_t pos = 0;
        bool skip = true;
        while (pos < n) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            pos += (uint32_t)((lcg >> 33) % 400);
            if (pos >= n) break;
            hints.push_back({pos, skip ? FastDeflate::HINT_SKIP : 0u});
            skip = !skip;
        }
        std::vector<uint8_t> out2(FastDeflate::bound(n) + 16), back2(n + 1);
        const size_t c2 = fd.compress(in, n, out2.data(), hints.data(), hints.size());
        z_stream z3;
        memset(&z3, 0, sizeof z3);
        inflateInit2(&z3, -15);
        z3.next_in = out2.data();
        z3.avail_in = (uInt)c2;
        z3.next_out = back2.data();
        z3.avail_out = (uInt)back2.size();
        const int r3 = inflate(&z3, Z_FINISH);
        const size_t got3 = z3.total_out;
        inflateEnd(&z3);
        if (c2 > FastDeflate::bound(n) || r3 != Z_STREAM_END || got3 != n || (n && memcmp(back2.data(), in, n) != 0)) {
            fprintf(stderr, "round trip with layout hints failed (n=%zu, %zu hints)\n", n, hints.size());
            return false;
        }
    }
    // the same stream, and zlib's own streams at several settings, through FastInflate
    static FastInflate fi;
    std::vector<uint8_t> mine(n + 1);
    if (!fi.inflate(out.data(), clen, mine.data(), n) || (n && memcmp(mine.data(), in, n) != 0)) {
        fprintf(stderr, "FastInflate failed on a FastDeflate stream (n=%zu clen=%zu)\n", n, clen);
        return false;
    }
    // the pair decoder: this stream next to the previous case's, in both orders, whole and damaged
    {
        static std::vector<uint8_t> prev_stream, prev_plain;
        static bool have_prev = false;
        static FastInflate fa, fb;
        if (have_prev) {
            const size_t pn = prev_plain.size();
            for (int order = 0; order < 2; order++) {
                std::vector<uint8_t> o1(pn + 1, 0xa5), o2(n + 1, 0xa5);
                const bool ok = order == 0 ? FastInflate::inflate2(fa, prev_stream.data(), prev_stream.size(), o1.data(), pn, fb, out.data(), clen, o2.data(), n)
                                           : FastInflate::inflate2(fa, out.data(), clen, o2.data(), n, fb, prev_stream.data(), prev_stream.size(), o1.data(), pn);
                if (!ok || (pn && memcmp(o1.data(), prev_plain.data(), pn) != 0) || (n && memcmp(o2.data(), in, n) != 0) || o1[pn] != 0xa5 || o2[n] != 0xa5) {
                    fprintf(stderr, "inflate2 failed (order %d, n=%zu, previous n=%zu)\n", order, n, pn);
                    return false;
                }
            }
            // a wrong size for either stream is refused; damaged streams stay inside the buffers (ASan watches)
            std::vector<uint8_t> o1(pn + 1), o2(n + 1);
            if (n && FastInflate::inflate2(fa, prev_stream.data(), prev_stream.size(), o1.data(), pn, fb, out.data(), clen, o2.data(), n - 1)) { fprintf(stderr, "inflate2: short output accepted\n"); return false; }
            if (pn && FastInflate::inflate2(fa, prev_stream.data(), prev_stream.size(), o1.data(), pn - 1, fb, out.data(), clen, o2.data(), n)) { fprintf(stderr, "inflate2: short output accepted\n"); return false; }
            if (clen > 4) {
                std::vector<uint8_t> bad(out.begin(), out.begin() + (std::ptrdiff_t)clen);
                static uint64_t lcg2 = 1234567;
                for (int rep = 0; rep < 3; rep++) {
                    lcg2 = lcg2 * 6364136223846793005ull + 1442695040888963407ull;
                    bad[(size_t)(lcg2 >> 33) % clen] ^= (uint8_t)(1u << ((lcg2 >> 20) & 7));
                    (void)FastInflate::inflate2(fa, prev_stream.data(), prev_stream.size(), o1.data(), pn, fb, bad.data(), clen, o2.data(), n);
                    (void)FastInflate::inflate2(fa, bad.data(), clen / 2, o2.data(), n, fb, prev_stream.data(), prev_stream.size(), o1.data(), pn);
                }
            }
        }
        prev_stream.assign(out.begin(), out.begin() + (std::ptrdiff_t)clen);
        prev_plain.assign(in, in + n);
        have_prev = true;
    }
    static const int settings[][2] = {{0, Z_DEFAULT_STRATEGY}, {1, Z_DEFAULT_STRATEGY}, {6, Z_DEFAULT_STRATEGY}, {9, Z_DEFAULT_STRATEGY},
                                      {6, Z_FIXED}, {6, Z_HUFFMAN_ONLY}, {6, Z_RLE}};
    std::vector<uint8_t> zbuf(n + n / 8 + 1024);
    for (const auto &st : settings) {
        z_stream z2;
        memset(&z2, 0, sizeof z2);
        deflateInit2(&z2, st[0], Z_DEFLATED, -15, 8, st[1]);
        z2.next_in = const_cast<uint8_t *>(in);
        z2.avail_in = (uInt)n;
        z2.next_out = zbuf.data();
        z2.avail_out = (uInt)zbuf.size();
        const int r2 = deflate(&z2, Z_FINISH);
        const size_t zl = z2.total_out;
        deflateEnd(&z2);
        if (r2 != Z_STREAM_END) { fprintf(stderr, "zlib deflate failed\n"); return false; }
        std::fill(mine.begin(), mine.end(), 0xa5);
        if (!fi.inflate(zbuf.data(), zl, mine.data(), n) || (n && memcmp(mine.data(), in, n) != 0) || mine[n] != 0xa5) {
            fprintf(stderr, "FastInflate failed on a zlib stream (level %d strategy %d, n=%zu)\n", st[0], st[1], n);
            return false;
        }
        {  // a zlib stream and the FastDeflate stream of the same bytes as a pair
            static FastInflate fa, fb;
            std::vector<uint8_t> o2(n + 1, 0xa5);
            std::fill(mine.begin(), mine.end(), 0xa5);
            if (!FastInflate::inflate2(fa, zbuf.data(), zl, mine.data(), n, fb, out.data(), clen, o2.data(), n) ||
                (n && (memcmp(mine.data(), in, n) != 0 || memcmp(o2.data(), in, n) != 0)) || mine[n] != 0xa5 || o2[n] != 0xa5) {
                fprintf(stderr, "inflate2 failed on a zlib stream (level %d strategy %d, n=%zu)\n", st[0], st[1], n);
                return false;
            }
        }
        // wrong sizes and damaged streams must be refused or at least stay inside the buffers (ASan watches)
        if (n && fi.inflate(zbuf.data(), zl, mine.data(), n - 1)) { fprintf(stderr, "short output accepted\n"); return false; }
        if (zl > 4) {
            std::vector<uint8_t> bad(zbuf.begin(), zbuf.begin() + (std::ptrdiff_t)zl);
            static uint64_t lcg = 88172645463325252ull;
            for (int rep = 0; rep < 3; rep++) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                bad[(size_t)(lcg >> 33) % zl] ^= (uint8_t)(1u << ((lcg >> 20) & 7));
                (void)fi.inflate(bad.data(), zl, mine.data(), n);
                (void)fi.inflate(bad.data(), zl / 2, mine.data(), n);
            }
        }
    }
    return true;
}

static size_t zlib_block(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<uint8_t *>(in);
    zs.avail_in = (uInt)n;
    zs.next_out = out;
    zs.avail_out = (uInt)cap;
    deflate(&zs, Z_FINISH);
    const size_t r = zs.total_out;
    deflateEnd(&zs);
    return r;
}

int main(int argc, char **argv) {
    std::mt19937_64 rng(12345);
    int fails = 0, cases = 0;
    const bool speed_only = getenv("DEFLATE_SPEED_ONLY") != nullptr;  // with a file: efforts 1 and 2 only, no unit cases
    for (int effort = 1; effort <= FastDeflate::MAX_EFFORT + 1 && !(speed_only && argc > 1); effort++) {
        // (the last round: a skip rule far beyond the shipped ones)
        FastDeflate fd = effort <= FastDeflate::MAX_EFFORT ? FastDeflate(effort) : FastDeflate(2, 1, 300);
        auto check = [&](const std::vector<uint8_t> &v, const char *what) {
            cases++;
            if (!roundtrip(fd, v.data(), v.size(), nullptr)) { fprintf(stderr, "  case: %s, n=%zu, effort=%d\n", what, v.size(), effort); fails++; }
        };
        // sizes around every boundary the parser has
        for (size_t n : {0u, 1u, 2u, 3u, 4u, 5u, 7u, 8u, 9u, 15u, 16u, 17u, 257u, 258u, 259u, 260u, 300u, 4096u, 32767u, 32768u, 32769u, 65279u, 65280u, 65535u}) {
            std::vector<uint8_t> v(n);
            for (auto &c : v) c = (uint8_t)rng();
            check(v, "random bytes (stored)");
            for (auto &c : v) c = 'A';
            check(v, "one byte repeated (dist 1, len 258)");
            for (size_t i = 0; i < n; i++) v[i] = "ACGT"[rng() & 3];
            check(v, "4-letter text");
            for (size_t i = 0; i < n; i++) v[i] = (uint8_t)(i % 251);
            check(v, "period 251");
            for (size_t i = 0; i < n; i++) v[i] = (uint8_t)((i * i) >> 3);
            check(v, "quadratic");
        }
        // skewed alphabets: Fibonacci-like frequencies push the unrestricted Huffman depth past 15 (and past 7 for
        // the code-length code)
        {
            std::vector<uint8_t> v;
            uint64_t a = 1, b = 1;
            for (int s = 0; s < 40 && v.size() < 60000; s++) {
                for (uint64_t k = 0; k < a && v.size() < 60000; k++) v.push_back((uint8_t)s);
                const uint64_t c = a + b; a = b; b = c;
            }
            std::shuffle(v.begin(), v.end(), rng);
            check(v, "fibonacci literal frequencies");
        }
        // far matches: the same 300-byte unit at distances up to 32768 and beyond
        for (size_t gap : {1000u, 32000u, 32468u, 32469u, 33000u}) {
            std::vector<uint8_t> v(gap + 600);
            for (auto &c : v) c = (uint8_t)rng();
            memcpy(v.data() + gap + 300, v.data(), 300);
            check(v, "far repeat");
        }
        // structured records (BAM-like): fixed fields that repeat, names that count up, random payload
        for (int rep = 0; rep < 20; rep++) {
            std::vector<uint8_t> v;
            while (v.size() < 60000) {
                const uint32_t pos = (uint32_t)(rng() % 100000000);
                const uint8_t fixed[12] = {1, 0, 0, 0, 0, 0, 0, 0, 60, 0, 99, 0};
                v.insert(v.end(), fixed, fixed + 12);
                v.insert(v.end(), (const uint8_t *)&pos, (const uint8_t *)&pos + 4);
                const std::string nm = "read" + std::to_string(v.size() / 300);
                v.insert(v.end(), nm.begin(), nm.end());
                v.push_back(0);
                for (int k = 0; k < 75; k++) v.push_back((uint8_t)rng());
                for (int k = 0; k < 150; k++) v.push_back((uint8_t)(20 + rng() % 21));
            }
            v.resize(std::min<size_t>(v.size(), 65280));
            check(v, "record-like");
        }
        // random mixtures
        for (int rep = 0; rep < 300; rep++) {
            const size_t n = (size_t)(rng() % 65536);
            std::vector<uint8_t> v(n);
            const int alpha = 1 + (int)(rng() % 64);
            const int copy_pct = (int)(rng() % 60);
            for (size_t i = 0; i < n;) {
                if (i > 8 && (int)(rng() % 100) < copy_pct) {
                    const size_t d = 1 + rng() % std::min<size_t>(i, 40000), l = 3 + rng() % 300;
                    for (size_t k = 0; k < l && i < n; k++, i++) v[i] = v[i - d];
                } else v[i++] = (uint8_t)(rng() % alpha);
            }
            check(v, "mixture");
        }
    }
    // crc32_fast against zlib's crc32: every length up to 5000, random lengths and offsets beyond, and continuation
    {
        std::vector<uint8_t> v(1 << 17);
        for (auto &c : v) c = (uint8_t)rng();
        for (int it = 0; it < 30000; it++) {
            size_t n = it < 5000 ? (size_t)it : (size_t)(rng() % 70000);
            const size_t off = (size_t)(rng() % 64);
            n = std::min(n, v.size() - off);
            const size_t cut = n ? (size_t)(rng() % n) : 0;
            const uint32_t want = (uint32_t)crc32(0L, v.data() + off, (uInt)n);
            cases++;
            if (crc32_fast(0, v.data() + off, n) != want ||
                crc32_fast(crc32_fast(0, v.data() + off, cut), v.data() + off + cut, n - cut) != want) {
                fprintf(stderr, "crc32_fast differs from zlib: n=%zu off=%zu cut=%zu\n", n, off, cut);
                fails++;
            }
        }
    }
    printf("deflate_selftest: %d cases, %d failures\n", cases, fails);
    if (fails) return 1;

    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { perror(argv[1]); return 2; }
        std::vector<uint8_t> data;
        uint8_t buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + k);
        fclose(f);
        const size_t B = 0xff00;
        std::vector<uint8_t> out(B + 1024);
        // DEFLATE_HINTS=2: the payload is a BAM stream; give the compressor each record's layout as the BGZF writer does
        // (packed bases: nothing to find; qualities: mild; tags and the next record's fixed fields: structure).  =1: bases
        // and qualities as one stretch without repeats
        std::vector<FastDeflate::Hint> hints;
        if (getenv("DEFLATE_HINTS") && data.size() > 12 && memcmp(data.data(), "BAM\1", 4) == 0) {
            auto rd32 = [&](size_t o) { int32_t v; memcpy(&v, data.data() + o, 4); return v; };
            size_t o = 8 + (size_t)rd32(4);
            const int nref = rd32(o);
            o += 4;
            for (int k = 0; k < nref; k++) o += 8 + (size_t)rd32(o);
            while (o + 36 <= data.size()) {
                const size_t bs = (size_t)rd32(o), rec = o + 4;
                if (rec + bs > data.size()) break;
                const size_t lname = data[rec + 8], ncig = (size_t)(data[rec + 12] | (data[rec + 13] << 8)), lseq = (size_t)rd32(rec + 16);
                const size_t seq0 = rec + 32 + lname + 4 * ncig, qend = seq0 + (lseq + 1) / 2 + lseq;
                hints.push_back({(uint32_t)seq0, FastDeflate::HINT_SKIP});
                if (getenv("DEFLATE_HINTS")[0] != '1') hints.push_back({(uint32_t)(seq0 + (lseq + 1) / 2), FastDeflate::HINT_MILD});
                hints.push_back({(uint32_t)qend, 0});
                o = rec + bs;
            }
            printf("%zu layout hints\n", hints.size());
        }
        std::vector<FastDeflate::Hint> bh;
        auto block_hints = [&](size_t o, size_t n) {  // the hints of [o, o + n), block-relative; a block that starts inside a stretch opens with it
            bh.clear();
            auto it = std::lower_bound(hints.begin(), hints.end(), o, [](const FastDeflate::Hint &h, size_t v) { return h.pos < v; });
            if (it != hints.begin() && std::prev(it)->miss) bh.push_back({0, std::prev(it)->miss});
            for (; it != hints.end() && it->pos < o + n; ++it) bh.push_back({(uint32_t)(it->pos - o), it->miss});
        };
        const int NE = FastDeflate::MAX_EFFORT;
        for (int mode = 0; mode < (speed_only ? 2 : NE + 2); mode++) {  // FastDeflate effort 1..NE, then zlib 1 and zlib 6
            int sk_a = -1, sk_c = -1;  // DEFLATE_SKIP=after,cap overrides the effort's skip rule (experiments)
            if (const char *e = getenv("DEFLATE_SKIP")) sscanf(e, "%d,%d", &sk_a, &sk_c);
            FastDeflate fd(std::min(mode + 1, NE), sk_a, sk_c);
            size_t total = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t o = 0; o < data.size(); o += B) {
                const size_t n = std::min(B, data.size() - o);
                if (mode < NE) {
                    size_t c = 0;
                    if (speed_only && !hints.empty()) {
                        block_hints(o, n);
                        c = fd.compress(data.data() + o, n, out.data(), bh.data(), bh.size());
                    } else if (speed_only) c = fd.compress(data.data() + o, n, out.data());
                    else if (!roundtrip(fd, data.data() + o, n, &c)) return 1;
                    total += c;
                } else total += zlib_block(data.data() + o, n, mode == NE ? 1 : 6, out.data(), out.size());
            }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (mode < NE) {  // time compression alone (the loop above also inflates)
                const auto t1 = std::chrono::steady_clock::now();
                for (size_t o = 0; o < data.size(); o += B) {
                    if (!hints.empty()) block_hints(o, std::min(B, data.size() - o));
                    fd.compress(data.data() + o, std::min(B, data.size() - o), out.data(), hints.empty() ? nullptr : bh.data(), bh.size());
                }
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            }
#ifdef FADE_DEFLATE_TIMING
            if (mode < NE) printf("    parse %.3f s, encode %.3f s (both loops)\n", fd.t_parse, fd.t_encode);
#endif
            printf("%-22s %10zu -> %10zu bytes (%.4f)  %7.1f MB/s\n",
                   mode < NE ? (std::string("FastDeflate effort ") + std::to_string(mode + 1)).c_str() : (mode == NE ? "zlib level 1" : "zlib level 6"),
                   data.size(), total, (double)total / (double)data.size(), (double)data.size() / dt / 1e6);
        }
    }
    if (argc > 1 && !speed_only) {
        FILE *f = fopen(argv[1], "rb");
        std::vector<uint8_t> data;
        uint8_t buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + k);
        fclose(f);
        const size_t B = 0xff00;
        FastDeflate fd(2);
        FastInflate fi;
        std::vector<std::vector<uint8_t>> blocks;
        std::vector<size_t> sizes;
        for (size_t o = 0; o < data.size(); o += B) {
            const size_t n = std::min(B, data.size() - o);
            std::vector<uint8_t> c(FastDeflate::bound(n) + 8);
            c.resize(fd.compress(data.data() + o, n, c.data()) + 8);
            blocks.push_back(c);
            sizes.push_back(n);
        }
        std::vector<uint8_t> out(B + 16), out2(B + 16);
        FastInflate fi2;
        for (int mode = 0; mode < 3; mode++) {
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t b = 0; b < blocks.size(); b++) {
                if (mode == 2) {
                    if (b + 1 < blocks.size()) {
                        if (!FastInflate::inflate2(fi, blocks[b].data(), blocks[b].size() - 8, out.data(), sizes[b],
                                                   fi2, blocks[b + 1].data(), blocks[b + 1].size() - 8, out2.data(), sizes[b + 1])) return 1;
                        b++;
                    } else if (!fi.inflate(blocks[b].data(), blocks[b].size() - 8, out.data(), sizes[b])) return 1;
                } else if (mode == 0) {
                    if (!fi.inflate(blocks[b].data(), blocks[b].size() - 8, out.data(), sizes[b])) return 1;
                } else {
                    z_stream zs;
                    memset(&zs, 0, sizeof zs);
                    inflateInit2(&zs, -15);
                    zs.next_in = blocks[b].data();
                    zs.avail_in = (uInt)(blocks[b].size() - 8);
                    zs.next_out = out.data();
                    zs.avail_out = (uInt)sizes[b];
                    inflate(&zs, Z_FINISH);
                    inflateEnd(&zs);
                }
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%-22s %7.1f MB/s (output bytes)\n", mode == 0 ? "FastInflate" : (mode == 1 ? "zlib inflate" : "FastInflate, pairs"), (double)data.size() / dt / 1e6);
        }
    }
    return 0;
}
