// hts_selftest.cpp — corrupt-input cases of host/hts_lite.hpp under AddressSanitizer + UBSan (CPU only):
// BAM records whose aux area is truncated or lies about its lengths, BGZF blocks whose BSIZE / ISIZE are impossible.
// Every case must be REJECTED (exception or layout_ok() == false) without touching memory outside the record.
//   make -C fade_amd/csrc build/hts_selftest && fade_amd/csrc/build/hts_selftest
#include "../hts_lite.hpp"
#include <thread>

#include <cstdio>

using namespace htsl;

static int failures = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static Rec make_rec(const std::vector<uint8_t> &aux) {
    Rec r;
    const char qname[] = "r1";
    const int l_seq = 4;
    r.d.assign(32, 0);
    r.wr<int32_t>(0, 0);
    r.wr<int32_t>(4, 100);
    r.d[8] = sizeof qname;
    r.wr<uint16_t>(12, 1);
    r.wr<uint16_t>(14, 0);
    r.wr<int32_t>(16, l_seq);
    r.d.insert(r.d.end(), qname, qname + sizeof qname);
    const uint32_t cig = (4u << 4) | 0u;
    r.d.insert(r.d.end(), (const uint8_t *)&cig, (const uint8_t *)&cig + 4);
    r.d.insert(r.d.end(), {0x12, 0x48});          // ACGT packed
    r.d.insert(r.d.end(), {30, 30, 30, 30});      // quals
    r.d.insert(r.d.end(), aux.begin(), aux.end());
    return r;
}

static std::vector<uint8_t> bgzf_block(const std::vector<uint8_t> &payload, int bsize_override = -1, int64_t isize_override = -1) {
    std::vector<uint8_t> out;
    bgzf_compress_block(payload.data(), payload.size(), 6, out);
    if (bsize_override >= 0) {
        const uint16_t bs = (uint16_t)bsize_override;
        memcpy(out.data() + 16, &bs, 2);
    }
    if (isize_override >= 0) {
        const uint32_t v = (uint32_t)isize_override;
        memcpy(out.data() + out.size() - 4, &v, 4);
    }
    return out;
}

static bool reader_rejects(const std::vector<uint8_t> &bytes) {
    const char *path = "/tmp/hts_selftest.bin";
    FILE *f = fopen(path, "wb");
    fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
    try {
        Pool pool(2);
        Reader rd(path, &pool);
        std::vector<Rec> recs;
        while (rd.read_chunk(recs, 1000)) {}
    } catch (const std::exception &) {
        return true;
    }
    return false;
}

int main() {
    // ---- well-formed aux area: accepted, tags found, updates work
    {
        Rec r = make_rec({'N', 'M', 'C', 3, 'r', 's', 'C', 1, 'X', 'Z', 'Z', 'h', 'i', 0, 'B', 'B', 'B', 's', 2, 0, 0, 0, 1, 0, 2, 0});
        CHECK(r.layout_ok());
        CHECK(r.aux_exists("rs") && r.aux_exists("BB") && !r.aux_exists("SA"));
        r.aux_update_uint("rs", 37);
        r.aux_update_str("XZ", "a longer string than before");
        r.aux_update_uint("NM", 70000);  // widens C -> I in place
        CHECK(r.layout_ok());
        Header h;
        h.names = {"chr1"};
        h.lens = {1000};
        std::string s;
        sam_format(r, h, s);
        CHECK(s.find("rs:i:37") != std::string::npos && s.find("BB:B:s,1,2") != std::string::npos);
    }
    // ---- (a) a 'B' array whose 32-bit count runs far past the record (sam_format used to loop over it)
    CHECK(!make_rec({'B', 'B', 'B', 'i', 0xff, 0xff, 0xff, 0x7f, 1, 0, 0, 0}).layout_ok());
    CHECK(!make_rec({'B', 'B', 'B', 'c', 0xff, 0xff, 0xff, 0xff}).layout_ok());
    // ---- (b) an rs tag cut short: type says 4 bytes, 1 is there (aux_update_uint used to memcpy past the end)
    CHECK(!make_rec({'r', 's', 'I', 1}).layout_ok());
    CHECK(!make_rec({'r', 's'}).layout_ok());
    CHECK(!make_rec({'r'}).layout_ok());
    // ---- (c) a Z string without its NUL, an unknown type byte, a B array of an unknown element type
    CHECK(!make_rec({'X', 'Z', 'Z', 'a', 'b', 'c'}).layout_ok());
    CHECK(!make_rec({'X', 'Q', '?', 0, 0}).layout_ok());
    CHECK(!make_rec({'B', 'B', 'B', 'Z', 1, 0, 0, 0, 0}).layout_ok());
    // ---- fixed fields that overrun the record
    {
        Rec r = make_rec({});
        r.wr<int32_t>(16, 1 << 20);  // l_seq
        CHECK(!r.layout_ok());
        Rec q = make_rec({});
        q.wr<uint16_t>(12, 60000);  // n_cigar
        CHECK(!q.layout_ok());
    }
    // ---- through the reader: a BAM whose one record has a truncated aux field is refused as a whole
    {
        std::vector<uint8_t> raw = {'B', 'A', 'M', 1, 0, 0, 0, 0, 1, 0, 0, 0, 5, 0, 0, 0, 'c', 'h', 'r', '1', 0, 0xe8, 3, 0, 0};
        Rec bad = make_rec({'r', 's', 'I', 1});
        const uint32_t bs = (uint32_t)bad.d.size();
        raw.insert(raw.end(), (const uint8_t *)&bs, (const uint8_t *)&bs + 4);
        raw.insert(raw.end(), bad.d.begin(), bad.d.end());
        std::vector<uint8_t> file = bgzf_block(raw);
        file.insert(file.end(), BGZF_EOF, BGZF_EOF + sizeof BGZF_EOF);
        CHECK(reader_rejects(file));
        // the same file with a sound record is read
        std::vector<uint8_t> raw2(raw.begin(), raw.begin() + 25);
        Rec good = make_rec({'r', 's', 'C', 1});
        const uint32_t bs2 = (uint32_t)good.d.size();
        raw2.insert(raw2.end(), (const uint8_t *)&bs2, (const uint8_t *)&bs2 + 4);
        raw2.insert(raw2.end(), good.d.begin(), good.d.end());
        std::vector<uint8_t> file2 = bgzf_block(raw2);
        file2.insert(file2.end(), BGZF_EOF, BGZF_EOF + sizeof BGZF_EOF);
        CHECK(!reader_rejects(file2));
    }
    // ---- BGZF framing: BSIZE smaller than header + trailer (the ISIZE read used to land in front of the block),
    //      BSIZE < 3 (underflow), ISIZE beyond the format's 64 KiB
    {
        std::vector<uint8_t> payload(1000, 'A');
        for (int bsz : {0, 1, 2, 10, 24}) {
            std::vector<uint8_t> blk = bgzf_block(payload, bsz);
            blk.resize(std::max<size_t>(blk.size(), 64));
            CHECK(reader_rejects(blk));
        }
        CHECK(reader_rejects(bgzf_block(payload, -1, 65537)));
        CHECK(reader_rejects(bgzf_block(payload, -1, 0x7fffffff)));
        CHECK(reader_rejects(bgzf_block(payload, -1, 999)));  // ISIZE that the stream does not inflate to
    }
    // ---- Pool: jobs from several threads at once; every index runs exactly once; an exception reaches its caller only
    {
        Pool pool(4);
        std::atomic<long> sums[3];
        for (auto &v : sums) v = 0;
        std::atomic<int> caught{0};
        std::vector<std::thread> callers;
        for (int c = 0; c < 3; c++)
            callers.emplace_back([&, c] {
                for (int rep = 0; rep < 200; rep++) {
                    std::vector<std::atomic<int>> seen(97 + (size_t)c);
                    for (auto &v : seen) v = 0;
                    pool.parallel_for(seen.size(), [&](size_t i) {
                        seen[i]++;
                        sums[c] += (long)i;
                    });
                    for (auto &v : seen)
                        if (v != 1) failures++;
                    if (c == 2 && rep % 50 == 7) {
                        try {
                            pool.parallel_for(64, [&](size_t i) { if (i == 13) throw std::runtime_error("task 13"); });
                        } catch (const std::runtime_error &) {
                            caught++;
                        }
                    }
                }
            });
        for (auto &t : callers) t.join();
        for (int c = 0; c < 3; c++) {
            const long n = 97 + c;
            CHECK(sums[c] == 200 * (n * (n - 1) / 2));
        }
        CHECK(caught == 4);
    }
    // ---- RawBuf: recycled buffers keep nothing of their previous size; reserve never shrinks the contents
    {
        for (int rep = 0; rep < 6; rep++) {
            RawBuf b;
            b.reserve((size_t)9 << 20);
            CHECK(b.size() == 0 && b.capacity() >= ((size_t)9 << 20));
            b.resize(100);
            memset(b.data(), rep, 100);
            b.resize((size_t)20 << 20);  // grows: the first 100 bytes survive
            CHECK(b.data()[0] == rep && b.data()[99] == rep);
            b.data()[b.size() - 1] = 1;
        }
    }
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("hts_selftest: all corrupt-input cases rejected cleanly\n");
    return 0;
}
