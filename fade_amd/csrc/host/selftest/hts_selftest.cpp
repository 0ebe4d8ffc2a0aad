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
    // ---- block reading equals record reading: SAM text and the BAM written from it (records that span BGZF blocks and
    // gulps, batches that end in the middle of a block), and a block written back equals the records written back
    {
        const char *sam_path = "/tmp/hts_selftest.sam", *bam_path = "/tmp/hts_selftest.bam", *bam2_path = "/tmp/hts_selftest2.bam";
        {
            FILE *f = fopen(sam_path, "w");
            fprintf(f, "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:100000\n@SQ\tSN:chr2\tLN:5000\n");
            uint64_t lcg = 99;
            auto rnd = [&](uint32_t m) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(lcg >> 33) % m; };
            for (int i = 0; i < 30000; i++) {
                const int lq = 20 + (int)rnd(180);
                std::string seq, qual;
                for (int k = 0; k < lq; k++) { seq += "ACGTN"[rnd(5)]; qual += (char)(33 + rnd(41)); }
                const int clip = (int)rnd(3) ? 0 : 1 + (int)rnd(10);
                char cig[64];
                if (clip && clip < lq) snprintf(cig, sizeof cig, "%dS%dM", clip, lq - clip);
                else snprintf(cig, sizeof cig, "%dM", lq);
                fprintf(f, "read%d\t%d\tchr%d\t%u\t60\t%s\t*\t0\t0\t%s\t%s\tNM:i:%u\tXS:Z:tag%u\n", i, rnd(2) ? 16 : 0, 1 + (int)rnd(2),
                        1 + rnd(4000), cig, seq.c_str(), qual.c_str(), rnd(5), rnd(1000));
            }
            fclose(f);
        }
        auto read_all = [&](const char *path, bool blocks, size_t batch, Header *hdr_out) {
            Pool pool(3);
            Reader rd(path, &pool);
            if (hdr_out) *hdr_out = rd.header();
            std::vector<std::vector<uint8_t>> recs;
            if (blocks) {
                RecordBlock blk;
                while (rd.read_block(blk, batch))
                    for (size_t i = 0; i < blk.size(); i++) recs.emplace_back(blk.view(i).bytes(), blk.view(i).bytes() + blk.view(i).nbytes());
            } else {
                std::vector<Rec> v;
                while (rd.read_chunk(v, batch)) {
                    for (auto &r : v) recs.push_back(r.d);
                    v.clear();
                }
            }
            return recs;
        };
        Header hdr;
        const auto sam_recs = read_all(sam_path, false, 7001, &hdr);
        const auto sam_blocks = read_all(sam_path, true, 4999, nullptr);
        CHECK(sam_recs.size() == 30000 && sam_recs == sam_blocks);
        {  // SAM -> BAM through the record writer, BAM -> BAM through the block writer
            Pool pool(3);
            FILE *fo = fopen(bam_path, "wb");
            {
                Reader rd(sam_path, &pool);
                Writer w(fo, OutFmt::BAM, hdr, &pool);
                std::vector<Rec> v;
                while (rd.read_chunk(v, 6000)) {
                    w.write(v);
                    v.clear();
                }
                w.close();
            }
            fclose(fo);
            fo = fopen(bam2_path, "wb");
            {
                Reader rd(bam_path, &pool);
                Writer w(fo, OutFmt::BAM, hdr, &pool);
                RecordBlock blk;
                Writer::BlockOut none;
                while (rd.read_block(blk, 3333)) w.write_block(blk, none);
                w.close();
            }
            fclose(fo);
        }
        const auto bam_recs = read_all(bam_path, false, 5000, nullptr);
        const auto bam_blocks = read_all(bam_path, true, 1234, nullptr);
        const auto bam2_blocks = read_all(bam2_path, true, 100000, nullptr);
        CHECK(bam_recs == sam_recs);
        CHECK(bam_blocks == sam_recs);
        CHECK(bam2_blocks == sam_recs);
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
