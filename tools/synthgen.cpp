// synthgen.cpp — fast generator of the synthetic workload of SURVEY.md §8(d) (MEASUREMENT TOOL, not product code).
//
// fade_amd/synth.py is the generator the tests and the golden fixtures use (numpy, ~3 s per million reads); bench.py
// needs 10 M reads per rank plus a 10 M-read BAM for its end-to-end leg, and that took half a minute of a run whose timed
// region is a fifth of a second.  This is the same workload — same laws for fragments, clips, planted artifacts
// (reverse-complement copies of a window segment), substitutions, unmapped and SA-tagged reads — drawn from a
// counter-based generator so that every read can be made by any thread, in C++ on all host threads.  The reads are NOT
// bit-identical to synth.py's (another random stream); nothing compares the two.
//
//   g++ -O2 -std=c++17 -shared -fPIC -o tools/libsynthgen.so tools/synthgen.cpp -lz -lpthread
#include "../fade_amd/csrc/host/hts_lite.hpp"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

using namespace htsl;

namespace {

inline uint64_t mix(uint64_t z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct Rng {  // one independent stream per (seed, object, purpose)
    uint64_t s;
    Rng(uint64_t seed, uint64_t idx, uint64_t stream) : s(mix(mix(seed) ^ mix(idx * 0x2545f4914f6cdd1dull + stream))) {}
    uint64_t next() { return s = mix(s); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)(next() % (uint64_t)(hi - lo + 1)); }  // inclusive
    double normal() {
        const double u1 = std::max(uniform(), 1e-300), u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

template <class F>
void parallel(int threads, int64_t n, F f) {
    threads = std::max(1, threads);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([=] { f(n * t / threads, n * (t + 1) / threads); });
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" {

struct SgCfg {
    int32_t n_contigs;
    int64_t contig_len;
    int32_t read_len, window;
    double p_sc;
    int32_t clip_min, clip_max;
    double insert_mu, insert_sd, p_unmapped, p_sa, p_sub, p_planted;
};

// genome: base codes 0..3 (A, C, G, T), contigs concatenated
void sg_genome(uint64_t seed, int64_t n_bases, uint8_t *codes, int threads) {
    const int64_t n_words = (n_bases + 31) / 32;
    parallel(threads, n_words, [=](int64_t lo, int64_t hi) {
        for (int64_t w = lo; w < hi; w++) {
            uint64_t v = mix(mix(seed) ^ mix((uint64_t)w + 0x51ed27u));
            const int64_t e = std::min<int64_t>(n_bases, 32 * (w + 1));
            for (int64_t i = 32 * w; i < e; i++, v >>= 2) codes[i] = (uint8_t)(v & 3);
        }
    });
}

struct SgPlan {  // per read, from the first pass
    int64_t span, segL, segR_end;
    int32_t clipL, clipR;
    uint8_t plantL, plantR, unmapped, n_ops;
};

// Reads idx0 .. idx0 + n - 1 of the stream `seed` (pairs: read 2 p and 2 p + 1 share a fragment).  Output arrays as in
// fade_amd/synth.make_reads: cigar_ops has room for 3 n entries, seq_packed n * ((L + 1) / 2), qual n * L.
// Returns the number of CIGAR ops written.
int64_t sg_reads(const uint8_t *codes, const SgCfg *cfg, int64_t n, uint64_t seed, int64_t idx0, int32_t *tid, int32_t *pos,
                 uint16_t *flag, uint8_t *has_sa, int32_t *l_seq, uint32_t *cigar_off, uint32_t *cigar_ops, uint32_t *seq_off,
                 uint8_t *seq_packed, uint8_t *qual, int threads) {
    const SgCfg c = *cfg;
    const int Lq = c.read_len, W = c.window, nbytes = (Lq + 1) / 2;
    const int64_t clen = c.contig_len;
    std::vector<SgPlan> plan((size_t)n);
    parallel(threads, n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            const int64_t gi = idx0 + i, pair = gi >> 1;
            Rng rp(seed, (uint64_t)pair, 1), rr(seed, (uint64_t)gi, 2);
            const int32_t t = (int32_t)rp.range(0, c.n_contigs - 1);
            int64_t ins = (int64_t)std::llround(c.insert_mu + c.insert_sd * rp.normal());
            ins = std::min<int64_t>(std::max<int64_t>(ins, Lq), clen);
            const int64_t fs = (int64_t)(rp.uniform() * (double)(clen - ins + 1));
            SgPlan p;
            p.span = (gi & 1) ? fs + ins - Lq : fs;
            const bool has_clip = rr.uniform() < c.p_sc;
            const double side = rr.uniform();
            const bool left = has_clip && (side < 0.475 || side >= 0.95), right = has_clip && side >= 0.475;
            int64_t cl = rr.range(c.clip_min, c.clip_max), cr = rr.range(c.clip_min, c.clip_max);
            if (!left) cl = 0;
            if (!right) cr = 0;
            if (cl + cr >= Lq) cr = 0;
            bool pl = left && rr.uniform() < c.p_planted, pr = right && rr.uniform() < c.p_planted;
            const int64_t dL = rr.range(0, W / 2), dR = rr.range(0, W / 2);
            const int64_t A = Lq - cl - cr, ps = p.span + cl;
            const int64_t ws = std::max<int64_t>(ps - W, 0), we = std::min<int64_t>(ps + A + W, clen);
            p.segL = ws + dL;
            p.segR_end = we - dR;
            pl = pl && (p.segL + cl <= we) && cl > 0;
            pr = pr && (p.segR_end - cr >= ws) && cr > 0;
            p.unmapped = rr.uniform() < c.p_unmapped;
            has_sa[i] = rr.uniform() < c.p_sa;
            p.clipL = (int32_t)cl;
            p.clipR = (int32_t)cr;
            p.plantL = pl;
            p.plantR = pr;
            p.n_ops = p.unmapped ? 0 : (uint8_t)(1 + (cl > 0) + (cr > 0));
            plan[(size_t)i] = p;
            tid[i] = p.unmapped ? -1 : t;
            pos[i] = p.unmapped ? -1 : (int32_t)ps;
            flag[i] = (uint16_t)(((gi & 1) ? 147 : 99) | (p.unmapped ? 4 : 0));
            l_seq[i] = Lq;
            seq_off[i] = (uint32_t)((uint64_t)i * (uint64_t)nbytes);
            // bases and qualities
            Rng rb(seed, (uint64_t)gi, 3);
            const uint8_t *g = codes + (int64_t)t * clen;
            uint8_t *sp = seq_packed + (int64_t)i * nbytes, *ql = qual + (int64_t)i * Lq;
            static const uint8_t nt16[4] = {1, 2, 4, 8};
            uint64_t bits = 0;
            int nb = 0;
            auto rnd = [&](int k) -> uint32_t {  // k fresh bits
                if (nb < k) { bits = rb.next(); nb = 64; }
                const uint32_t v = (uint32_t)(bits & ((1u << k) - 1));
                bits >>= k;
                nb -= k;
                return v;
            };
            // substitutions: positions by geometric skips (p_sub is small), not one draw per base
            int64_t next_sub = c.p_sub > 0 ? (int64_t)(std::log(std::max(rb.uniform(), 1e-300)) / std::log(1.0 - c.p_sub)) : (int64_t)1 << 40;
            uint8_t cur = 0;
            for (int k = 0; k < Lq; k++) {
                uint32_t base;
                const bool inL = k < cl, inR = k >= Lq - cr;
                if (inL && pl) base = 3u - g[std::min<int64_t>(std::max<int64_t>(p.segL + cl - 1 - k, 0), clen - 1)];
                else if (inR && pr) base = 3u - g[std::min<int64_t>(std::max<int64_t>(p.segR_end - 1 - (k - (Lq - cr)), 0), clen - 1)];
                else if (inL || inR) base = rnd(2);
                else base = g[std::min<int64_t>(std::max<int64_t>(p.span + k, 0), clen - 1)];
                if (k == next_sub) {
                    base = (base + 1u + rnd(2) % 3u) & 3u;
                    next_sub += 1 + (int64_t)(std::log(std::max(rb.uniform(), 1e-300)) / std::log(1.0 - c.p_sub));
                }
                if (k & 1) sp[k >> 1] = (uint8_t)(cur | nt16[base]);
                else cur = (uint8_t)(nt16[base] << 4);
                ql[k] = (uint8_t)(20 + rb.next() % 21);
            }
            if (Lq & 1) sp[Lq >> 1] = cur;
        }
    });
    int64_t at = 0;
    for (int64_t i = 0; i < n; i++) {
        const SgPlan &p = plan[(size_t)i];
        cigar_off[i] = (uint32_t)at;
        if (!p.unmapped) {
            if (p.clipL) cigar_ops[at++] = ((uint32_t)p.clipL << 4) | 4u;
            cigar_ops[at++] = (uint32_t)(c.read_len - p.clipL - p.clipR) << 4;
            if (p.clipR) cigar_ops[at++] = ((uint32_t)p.clipR << 4) | 4u;
        }
    }
    cigar_off[n] = (uint32_t)at;
    seq_off[n] = (uint32_t)((uint64_t)n * (uint64_t)nbytes);
    return at;
}

// Drop the bases of the records the device never aligns (unmapped, or no S op; anno.d:61): what a packing thread does
// while it fills a block.  seq_out has room for the input's bytes; returns the bytes kept, *n_with_seq the records that
// keep theirs, *span the longest cigar.alignedLength among them.
int64_t sg_compact(int64_t n, const uint16_t *flag, const uint32_t *cigar_off, const uint32_t *cigar_ops, const uint32_t *seq_off,
                   const uint8_t *seq_packed, uint32_t *seq_off_out, uint8_t *seq_out, int32_t *n_with_seq, int32_t *span) {
    int64_t at = 0;
    int32_t cnt = 0, sp = 1;
    for (int64_t i = 0; i < n; i++) {
        seq_off_out[i] = (uint32_t)at;
        if (flag[i] & 4) continue;
        bool soft = false;
        int64_t al = 0;
        for (uint32_t k = cigar_off[i]; k < cigar_off[i + 1]; k++) {
            const uint32_t op = cigar_ops[k] & 15u;
            soft |= op == 4;
            if ((0x18Du >> op) & 1u) al += cigar_ops[k] >> 4;  // include/fadehip.h FADEHIP_REF_CONSUMING_OPS
        }
        if (!soft) continue;
        const uint32_t nb = seq_off[i + 1] - seq_off[i];
        memcpy(seq_out + at, seq_packed + seq_off[i], nb);
        at += nb;
        cnt++;
        sp = std::max<int32_t>(sp, (int32_t)std::min<int64_t>(al, INT32_MAX));
    }
    seq_off_out[n] = (uint32_t)at;
    *n_with_seq = cnt;
    *span = sp;
    return at;
}

// FASTA of the genome (line width 60), contigs chr1..chrN
int sg_write_fasta(const char *path, const uint8_t *codes, int n_contigs, int64_t contig_len) {
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    std::vector<char> line(61);
    std::vector<char> buf;
    buf.reserve((size_t)(contig_len + contig_len / 60 + 64));
    for (int c = 0; c < n_contigs; c++) {
        buf.clear();
        char name[32];
        const int nl = snprintf(name, sizeof name, ">chr%d\n", c + 1);
        buf.insert(buf.end(), name, name + nl);
        const uint8_t *g = codes + (int64_t)c * contig_len;
        for (int64_t i = 0; i < contig_len; i += 60) {
            const int64_t e = std::min<int64_t>(contig_len, i + 60);
            for (int64_t k = i; k < e; k++) buf.push_back("ACGT"[g[k]]);
            buf.push_back('\n');
        }
        if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); return 1; }
    }
    return fclose(f) != 0;
}

struct SgBam {
    FILE *f = nullptr;
    Pool *pool = nullptr;
    Writer *w = nullptr;
    Header hdr;
};

void *sg_bam_open(const char *path, int n_contigs, int64_t contig_len, int threads) {
    try {
        SgBam *b = new SgBam();
        b->f = fopen(path, "wb");
        if (!b->f) { delete b; return nullptr; }
        b->pool = new Pool(std::max(1, threads));
        b->hdr.text = "@HD\tVN:1.6\tSO:unsorted\n";
        for (int c = 0; c < n_contigs; c++) {
            const std::string nm = "chr" + std::to_string(c + 1);
            b->hdr.names.push_back(nm);
            b->hdr.lens.push_back(contig_len);
            b->hdr.text += "@SQ\tSN:" + nm + "\tLN:" + std::to_string(contig_len) + "\n";
        }
        b->hdr.text += "@PG\tID:synth\tPN:tools/synthgen\n";
        b->w = new Writer(b->f, OutFmt::BAM, b->hdr, b->pool);
        return b;
    } catch (...) {
        return nullptr;
    }
}

// the reads of one sg_reads batch as BAM records r<name_base + i/2>, MAPQ 60, SA:Z on the reads that have one
int sg_bam_write(void *h, int64_t n, int64_t name_base, const int32_t *tid, const int32_t *pos, const uint16_t *flag, const uint8_t *has_sa,
                 const int32_t *l_seq, const uint32_t *cigar_off, const uint32_t *cigar_ops, const uint32_t *seq_off,
                 const uint8_t *seq_packed, const uint8_t *qual) {
    SgBam *b = (SgBam *)h;
    try {
        static const char sa[] = "SAZchr1,1,+,50M,60,0;";
        RecordBlock blk;
        blk.off.resize((size_t)n);
        blk.len.resize((size_t)n);
        std::vector<std::string> names((size_t)n);
        size_t at = 0;
        int64_t qoff = 0;
        std::vector<int64_t> qo((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            names[(size_t)i] = "r" + std::to_string(name_base + i / 2);
            const size_t nc = cigar_off[i + 1] - cigar_off[i];
            const size_t len = 32 + names[(size_t)i].size() + 1 + 4 * nc + ((size_t)l_seq[i] + 1) / 2 + (size_t)l_seq[i] + (has_sa[i] ? sizeof sa : 0);
            at += 4;  // (room for the block_size a BAM stream carries in front; the writer puts its own)
            blk.off[(size_t)i] = (uint32_t)at;
            blk.len[(size_t)i] = (uint32_t)len;
            at += len;
            qo[(size_t)i] = qoff;
            qoff += l_seq[i];
        }
        if (at >= 0xffffffffull) return 2;
        blk.buf.resize(at);
        b->pool->parallel_for((size_t)b->pool->size() * 4, [&](size_t t) {
            const size_t nt = (size_t)b->pool->size() * 4;
            for (size_t i = (size_t)n * t / nt; i < (size_t)n * (t + 1) / nt; i++) {
                uint8_t *d = blk.buf.data() + blk.off[i];
                const std::string &nm = names[i];
                const uint32_t nc = cigar_off[i + 1] - cigar_off[i];
                int64_t reflen = 0;
                for (uint32_t k = cigar_off[i]; k < cigar_off[i + 1]; k++)
                    if ((0x18Du >> (cigar_ops[k] & 15u)) & 1u) reflen += cigar_ops[k] >> 4;
                auto w32 = [&](size_t o, int32_t v) { memcpy(d + o, &v, 4); };
                auto w16 = [&](size_t o, uint16_t v) { memcpy(d + o, &v, 2); };
                w32(0, tid[i]);
                w32(4, pos[i]);
                d[8] = (uint8_t)(nm.size() + 1);
                d[9] = tid[i] >= 0 ? 60 : 0;
                w16(10, (uint16_t)(tid[i] >= 0 ? reg2bin(pos[i], pos[i] + std::max<int64_t>(reflen, 1)) : 4680));
                w16(12, (uint16_t)nc);
                w16(14, flag[i]);
                w32(16, l_seq[i]);
                w32(20, -1);
                w32(24, -1);
                w32(28, 0);
                size_t o = 32;
                memcpy(d + o, nm.c_str(), nm.size() + 1);
                o += nm.size() + 1;
                memcpy(d + o, cigar_ops + cigar_off[i], 4 * (size_t)nc);
                o += 4 * (size_t)nc;
                const size_t sb = ((size_t)l_seq[i] + 1) / 2;
                memcpy(d + o, seq_packed + seq_off[i], sb);
                o += sb;
                memcpy(d + o, qual + qo[i], (size_t)l_seq[i]);
                o += (size_t)l_seq[i];
                if (has_sa[i]) memcpy(d + o, sa, sizeof sa);
            }
        }, CPU_COPY);
        const Writer::BlockOut nothing;
        b->w->write_block(blk, nothing);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sg_bam_write: %s\n", e.what());
        return 1;
    }
}

int sg_bam_close(void *h) {
    SgBam *b = (SgBam *)h;
    int rc = 0;
    try {
        b->w->close();
    } catch (...) {
        rc = 1;
    }
    delete b->w;
    delete b->pool;
    fclose(b->f);
    delete b;
    return rc;
}

}  // extern "C"
