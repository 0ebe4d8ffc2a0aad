// fade_main.cpp — the `fade` host driver: `annotate` over the fadehip C ABI, plus the host-only consumers
// `extract` (source/remap.d) and `out` (source/filter.d).
//
// Mirrors the CLI surface of the reference:
//   source/app.d:64-101   main, getopt(annotate): -t/--threads, --min-length, -w/--window-size, -b/--bam, -u/--ubam,
//                         -h/--help; `-b -u` together is an error (exit 1); < 3 args or -h prints help (exit 0)
//   source/anno.d:16-52   annotate(): warning line, open BAM/SAM + FASTA, @PG header line, per-record
//                         annotateTask, write to stdout as SAM / uBAM / BAM (source/util.d:65-76)
//   source/anno.d:94-107  tag order rs, am, as, ar, ab;  analysis.d:84-92,108-118 string contents
// The loop at anno.d:44-50 becomes read-chunk -> pack (pointers into the BAM bytes) -> fadehip_annotate_*
// -> format tags -> write: three threaded stages (reader / device+tags / writer) over two device slots.
// Additive flags: --gpus N, --batch N, --stats, --timing.
// There is no CPU alignment path in this program.
#include <memory>
#include "hts_lite.hpp"
#include "../../../include/fadehip.h"

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <deque>
#include <functional>
#include <future>
#include <sched.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char **environ;

#ifndef FADE_VERSION
#define FADE_VERSION "v0.5.0-mi355x"
#endif

using namespace htsl;

// the driver double-buffers: two of the ctx's slots per device (one batch computes while the previous one is collected)
static size_t kSlotsInUse = 1;  // batches in flight per device: one, a second when the device falls behind (annotate_main)
static const char *kHeader = "Fragmentase Artifact Detection and Elimination\nversion: " FADE_VERSION "\n";

static void print_full_help() {
    fprintf(stderr,
            "%s\nusage: fade [subcommand]\n"
            "    annotate: marks artifact reads in bam tags (must be done first)\n"
            "    out: eliminates artifact from reads(may require queryname sorted bam)\n"
            "    stats: reports extended information about artifact reads\n"
            "    stats-clip: reports extended information about all soft-clipped reads\n"
            "    extract: extracts artifacts into a mapped bam\n\n"
            "-h --help This help information.\n\n",
            kHeader);
}

static void print_anno_help() {
    fprintf(stderr,
            "%s\nannotate: performs re-alignment of soft-clips and annotates bam records with bitflag (rs) and "
            "realignment tags (am)\nusage: fade annotate [options] <input BAM/SAM> <Indexed fasta reference>\n\n"
            "-t     --threads extra threads for parsing the bam file\n"
            "    --min-length Minimum number of bases for a soft-clip to be considered for artifact detection\n"
            "-w --window-size Number of bases considered outside of read or mate region for re-alignment\n"
            "-b         --bam output bam\n"
            "-u        --ubam output uncompressed bam\n"
            "          --gpus number of MI355X devices, one process each on its own range of the input (default 1)\n"
            "    --out-shards with --gpus N: every device writes a complete file PREFIX.<k>.bam (.sam), nothing is merged\n"
            "         --batch records per device batch (default 262144)\n"
            "         --stats print the stats.d summary of this run to stderr\n"
            "-h        --help This help information.\n\n",
            kHeader);
}

struct Opts {
    int threads = 0, floor_len = 5, window = 300, gpus = 1, batch = 262144;
    bool bam = false, ubam = false, help = false, stats = false, timing = false, clip = false;
    std::vector<std::string> pos;
    std::string out_shards;  // --out-shards PREFIX (with --gpus N)
    std::string seen;  // one letter per option met: t m w b u c h g(pus) B(atch) s(tats) T(iming) S(hards)
};

// std.getopt with config.bundling: short flags bundle (-bu), values attach (-w100, -w 100, --window-size=100)
static bool parse_opts(int argc, char **argv, Opts &o, std::string &err) {
    auto need_int = [&](const std::string &name, const char *v, int &dst) {
        char *e = nullptr;
        long x = strtol(v, &e, 10);
        if (!v[0] || *e) { err = "Invalid value for option " + name + ": " + v; return false; }
        dst = (int)x;
        return true;
    };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--") { for (int k = i + 1; k < argc; k++) o.pos.push_back(argv[k]); break; }
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string name = a.substr(2), val;
            bool has_val = false;
            size_t eq = name.find('=');
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
            auto take = [&](int &dst) {
                if (!has_val) {
                    if (i + 1 >= argc) { err = "Missing value for argument --" + name; return false; }
                    val = argv[++i];
                }
                return need_int("--" + name, val.c_str(), dst);
            };
            if (name == "threads") { o.seen += 't'; if (!take(o.threads)) return false; }
            else if (name == "min-length") { o.seen += 'm'; if (!take(o.floor_len)) return false; }
            else if (name == "window-size") { o.seen += 'w'; if (!take(o.window)) return false; }
            else if (name == "gpus") { o.seen += 'g'; if (!take(o.gpus)) return false; }
            else if (name == "batch") { o.seen += 'B'; if (!take(o.batch)) return false; }
            else if (name == "out-shards") {
                o.seen += 'S';
                if (!has_val) {
                    if (i + 1 >= argc) { err = "Missing value for argument --" + name; return false; }
                    val = argv[++i];
                }
                if (val.empty()) { err = "Invalid value for option --out-shards: (empty)"; return false; }
                o.out_shards = val;
            }
            else if (name == "bam") { o.seen += 'b'; o.bam = true; }
            else if (name == "ubam") { o.seen += 'u'; o.ubam = true; }
            else if (name == "stats") { o.seen += 's'; o.stats = true; }
            else if (name == "timing") { o.seen += 'T'; o.timing = true; }
            else if (name == "clip") { o.seen += 'c'; o.clip = true; }
            else if (name == "help") o.help = true;
            else { err = "Unrecognized option --" + name; return false; }
        } else if (a.size() > 1 && a[0] == '-' && a != "-") {
            for (size_t k = 1; k < a.size(); k++) {
                const char c = a[k];
                if (c != 'h') o.seen += c;
                if (c == 'b') o.bam = true;
                else if (c == 'u') o.ubam = true;
                else if (c == 'h') o.help = true;
                else if (c == 'c') o.clip = true;
                else if (c == 't' || c == 'w') {
                    std::string val = a.substr(k + 1);
                    if (!val.empty() && val[0] == '=') val = val.substr(1);
                    if (val.empty()) {
                        if (i + 1 >= argc) { err = std::string("Missing value for argument -") + c; return false; }
                        val = argv[++i];
                    }
                    if (!need_int(std::string("-") + c, val.c_str(), c == 't' ? o.threads : o.window)) return false;
                    break;
                } else { err = std::string("Unrecognized option -") + c; return false; }
            }
        } else o.pos.push_back(a);
    }
    return true;
}


// ------------------------------------------------------------------ one batch in flight
// Pinned batch blocks (fadehip_batch_bytes / fadehip_batch_bind) are recycled: upload is one hipMemcpyAsync from them.
struct BlockPool {
    struct Block { void *p; size_t cap; };
    fadehip_ctx *ctx = nullptr;
    std::mutex m;
    std::vector<Block> free_;
    void *acquire(size_t bytes, size_t *cap) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t k = 0; k < free_.size(); k++)
                if (free_[k].cap >= bytes) {
                    Block b = free_[k];
                    free_.erase(free_.begin() + (long)k);
                    *cap = b.cap;
                    return b.p;
                }
        }
        void *p = nullptr;
        const size_t want = bytes + bytes / 4 + 4096;
        if (fadehip_host_alloc(ctx, want, &p)) throw std::runtime_error(std::string("pinned allocation: ") + fadehip_last_error(ctx));
        *cap = want;
        return p;
    }
    void release(void *p, size_t cap) {
        if (!p) return;
        std::lock_guard<std::mutex> l(m);
        free_.push_back({p, cap});
    }
    void drain() {
        for (auto &b : free_) fadehip_host_free(ctx, b.p);
        free_.clear();
    }
};

// the writer's BGZF blocks compressed on the device (fadehip_bgzf_deflate_*): two lanes, each with a pinned staging buffer
struct DeviceBgzf : BgzfDevice {
    fadehip_ctx *ctx;
    struct Lane { void *p = nullptr; size_t cap = 0; } lane_[FADEHIP_BGZF_LANES];
    explicit DeviceBgzf(fadehip_ctx *c) : ctx(c) {}
    ~DeviceBgzf() override {
        for (auto &l : lane_)
            if (l.p) fadehip_host_free(ctx, l.p);
    }
    int lanes() const override { return FADEHIP_BGZF_LANES; }
    uint8_t *stage(int lane, size_t bytes) override {
        Lane &l = lane_[lane];
        if (bytes > l.cap) {
            if (l.p) fadehip_host_free(ctx, l.p);
            l.p = nullptr;
            l.cap = bytes + bytes / 4 + (1u << 20);
            if (fadehip_host_alloc(ctx, l.cap, &l.p)) throw std::runtime_error(std::string("pinned allocation: ") + fadehip_last_error(ctx));
        }
        return (uint8_t *)l.p;
    }
    void submit(int lane, size_t bytes) override {
        if (fadehip_bgzf_deflate_submit(ctx, lane, lane_[lane].p, bytes)) throw std::runtime_error(std::string("device BGZF: ") + fadehip_last_error(ctx));
    }
    void wait(int lane, const uint8_t **out, size_t *n) override {
        if (fadehip_bgzf_deflate_wait(ctx, lane, out, n)) throw std::runtime_error(std::string("device BGZF: ") + fadehip_last_error(ctx));
    }
};

struct Chunk {
    RecordBlock blk;             // the records, framed in place (BAM: in the inflated bytes; SAM: parsed into the same layout)
    Writer::BlockOut bout;       // ... and what annotate adds to them
    size_t n_records() const { return blk.size(); }
    // anno.d:61-65: an unmapped record, or one without an S op, gets rs = 0 and nothing else: such records are not sent.
    std::vector<uint32_t> sent;  // indices (into recs) of the records that are
    fadehip_read_batch b;        // bound into `block`
    void *block = nullptr;
    size_t block_cap = 0;
    // what the writer stage needs of the results, copied out of the slot's pinned result block
    std::vector<uint8_t> rs_sent;
    std::vector<fadehip_aln> art;  // the artifact calls (art != 0); read_idx indexes `sent`
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_oversize = 0;
    int dev = 0, slot = 0;
};

template <class R>
static inline bool needs_device(const R &r) {
    if (r.flag() & 4) return false;
    for (int k = 0; k < r.n_cigar(); k++)
        if ((r.cigar_op(k) & 15u) == 4u) return true;
    return false;
}

// records -> the batch block, in parallel: count per range, prefix, fill
template <class Get>
static void pack_chunk_t(Chunk &c, Pool &pool, BlockPool &blocks, const Get &get) {
    const size_t n = c.n_records();
    const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)pool.size(), n / 4096 + 1));
    struct Range { size_t n_sent = 0, n_cig = 0, n_seq = 0; int64_t span = 0; };
    std::vector<Range> rg(nt + 1);
    std::vector<uint8_t> need(n);
    pool.parallel_for(nt, [&](size_t t) {
        Range r;
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
            const auto &rec = get(i);
            need[i] = needs_device(rec) ? 1 : 0;
            if (!need[i]) continue;
            r.n_sent++;
            r.n_cig += (size_t)rec.n_cigar();
            r.n_seq += ((size_t)rec.l_seq() + 1) / 2;
            r.span = std::max(r.span, cigar_ref_len(rec));
        }
        rg[t + 1] = r;
    }, CPU_PACK);
    int64_t span = 1;
    for (size_t t = 1; t <= nt; t++) {
        span = std::max(span, rg[t].span);
        rg[t].n_sent += rg[t - 1].n_sent;
        rg[t].n_cig += rg[t - 1].n_cig;
        rg[t].n_seq += rg[t - 1].n_seq;
    }
    const size_t ns = rg[nt].n_sent, ncig = rg[nt].n_cig, nseq = rg[nt].n_seq;
    if (nseq >= ((size_t)1 << 31)) throw std::runtime_error("batch too large: lower --batch");
    c.block = blocks.acquire(fadehip_batch_bytes((int32_t)ns, (int64_t)ncig, (int64_t)nseq), &c.block_cap);
    if (fadehip_batch_bind(c.block, (int32_t)ns, (int64_t)ncig, (int64_t)nseq, &c.b)) throw std::runtime_error(fadehip_last_error(nullptr));
    c.sent.resize(ns);
    int32_t *tid = const_cast<int32_t *>(c.b.tid), *pos = const_cast<int32_t *>(c.b.pos), *lseq = const_cast<int32_t *>(c.b.l_seq);
    uint16_t *flag = const_cast<uint16_t *>(c.b.flag);
    uint8_t *has_sa = const_cast<uint8_t *>(c.b.has_sa), *seq = const_cast<uint8_t *>(c.b.seq_packed);
    uint32_t *coff = const_cast<uint32_t *>(c.b.cigar_off), *soff = const_cast<uint32_t *>(c.b.seq_off), *cops = const_cast<uint32_t *>(c.b.cigar_ops);
    pool.parallel_for(nt, [&](size_t t) {
        size_t k = rg[t].n_sent, nc = rg[t].n_cig, nq = rg[t].n_seq;
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
            if (!need[i]) continue;
            const auto &r = get(i);
            c.sent[k] = (uint32_t)i;
            tid[k] = r.tid();
            pos[k] = r.pos();
            lseq[k] = r.l_seq();
            flag[k] = (uint16_t)r.flag();
            has_sa[k] = r.aux_exists("SA") ? 1 : 0;  // anno.d:73
            coff[k] = (uint32_t)nc;
            soff[k] = (uint32_t)nq;
            const size_t cb = 4 * (size_t)r.n_cigar(), sb = ((size_t)r.l_seq() + 1) / 2;
            if (cb) memcpy(cops + nc, r.cigar_bytes(), cb);
            if (sb) memcpy(seq + nq, r.seq(), sb);
            nc += (size_t)r.n_cigar();
            nq += sb;
            k++;
        }
    }, CPU_PACK);
    coff[ns] = (uint32_t)ncig;
    soff[ns] = (uint32_t)nseq;
    c.b.n_skipped = (int32_t)(n - ns);
    c.b.ref_span_bound = (int32_t)std::min<int64_t>(span, INT32_MAX);
}
static void pack_chunk(Chunk &c, Pool &pool, BlockPool &blocks) {
    pack_chunk_t(c, pool, blocks, [&](size_t i) { return c.blk.view(i); });
}

static std::string cigar_string(const uint32_t *ops, int n) {
    std::string s;
    for (int k = 0; k < n; k++) {
        append_int(s, ops[k] >> 4);
        s += CIGAR_STR[std::min<uint32_t>(ops[k] & 15, 9)];
    }
    return s;
}

// The four strings of an artifact call: analysis.d:84-92 / 108-118 + anno.d:98-107 (am, as, ar, ab; "left;right").
template <class R>
static void artifact_strings(const R &r, const fadehip_aln &a, const Header &h, std::string out[4]) {
    static const uint8_t comp[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};  // util.d:18-20
    const int lq = r.l_seq();
    std::string seq((size_t)lq, 'N'), qrc((size_t)lq, 'N'), bq((size_t)lq, '!');
    const uint8_t *sq = r.seq(), *ql = r.qual();
    for (int j = 0; j < lq; j++) {
        const int code = (sq[j >> 1] >> ((~j & 1) << 2)) & 15;
        seq[(size_t)j] = NT16_STR[code];
        qrc[(size_t)(lq - 1 - j)] = NT16_STR[comp[code]];  // util.d:23-34
        bq[(size_t)j] = (char)(ql[j] + 33);
    }
    const int nops = std::min(a.sw.n_ops, FADEHIP_MAX_OPS);
    const std::string cig = cigar_string(a.sw.ops, nops);
    const int64_t apos = a.win_start + a.sw.beg_ref;
    const int64_t pos = r.pos();
    std::string am = (r.tid() >= 0 && r.tid() < (int)h.names.size() ? h.names[(size_t)r.tid()] : std::string("*"));
    am += ',';
    append_int(am, apos);
    am += ',';
    am += cig;
    std::string l[4], rr[4];
    if (a.art & 1) {  // analysis.d:84-92
        const int64_t clip = a.clip_left;
        const int64_t overlap = apos >= pos - clip ? apos - (pos - clip) : 0;
        const int64_t lead = (a.sw.ops[0] & 15) == 4 ? (a.sw.ops[0] >> 4) : 0;
        const int64_t plen = std::min<int64_t>(lq, (lq - lead) + overlap);
        l[0] = am; l[1] = seq.substr(0, (size_t)plen); l[2] = qrc.substr((size_t)(lq - plen)); l[3] = bq.substr(0, (size_t)plen);
    }
    if (a.art & 2) {  // analysis.d:108-118
        const int64_t clip = a.clip_right;
        int64_t res_al = 0;
        for (int q = 0; q < nops; q++) {
            const uint32_t op = a.sw.ops[q] & 15;
            if (FADEHIP_OP_CONSUMES_REF(op)) res_al += a.sw.ops[q] >> 4;
        }
        const int64_t lhs = pos + a.aligned_len + clip, rhs = apos + res_al;
        const int64_t overlap = lhs >= rhs ? lhs - rhs : 0;
        const int64_t trail = (a.sw.ops[nops - 1] & 15) == 4 ? (a.sw.ops[nops - 1] >> 4) : 0;
        const int64_t plen = std::min<int64_t>(lq, (lq - trail) + overlap);
        rr[0] = am; rr[1] = seq.substr((size_t)(lq - plen)); rr[2] = qrc.substr(0, (size_t)plen); rr[3] = bq.substr((size_t)(lq - plen));
    }
    for (int k = 0; k < 4; k++) out[k] = l[k] + ";" + rr[k];  // anno.d:100-106
}

static void tag_owned_record(Rec &r, uint8_t rs, const fadehip_aln *a, const Header &h) {
    r.aux_update_uint("rs", rs);  // anno.d:63,94
    if (!a) return;
    std::string t[4];
    artifact_strings(r, *a, h, t);
    r.aux_update_str("am", t[0]);  // anno.d:100
    r.aux_update_str("as", t[1]);  // anno.d:102
    r.aux_update_str("ar", t[2]);  // anno.d:104
    r.aux_update_str("ab", t[3]);  // anno.d:106
}

// anno.d:94-107 over a chunk: rs for every record (0 for the ones anno.d:61-65 settles without the device), the artifact
// strings for the few that have them, keeping the reference's tag order rs, am, as, ar, ab
static void apply_tags(Chunk &c, const Header &h, Pool &pool) {
    const size_t n = c.n_records();
    std::vector<uint8_t> rs(n, 0);
    for (size_t k = 0; k < c.sent.size(); k++) rs[c.sent[k]] = c.rs_sent[k];
    std::vector<int> art_of(n, -1);
    for (size_t k = 0; k < c.art.size(); k++) {
        const int32_t si = c.art[k].read_idx;  // an index into the records that were sent
        if (si < 0 || (size_t)si >= c.sent.size()) throw std::runtime_error("device returned an alignment for a record that was not sent");
        art_of[c.sent[(size_t)si]] = (int)k;
    }
    const size_t nt = (size_t)pool.size();
    // Records framed in place: the new tags become a suffix behind the record's bytes (htslib appends an absent tag).
    // A record that already carries one of the five tags (annotating an annotated file) is rebuilt as a whole instead,
    // with htslib's update-in-place semantics.
    Writer::BlockOut &o = c.bout;
    o.sfx_off.assign(n + 1, 0);
    o.owned.clear();
    std::vector<std::string> art_str(c.art.size() * 4);
    std::vector<std::vector<std::pair<uint32_t, Rec>>> owned_t(nt);
    pool.parallel_for(nt, [&](size_t t) {
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
            const RecView v = c.blk.view(i);
            bool has_ours = false;
            for (size_t p = v.aux_off(); p + 3 <= v.nbytes();) {
                const size_t fs = v.aux_field_size(p + 2);
                if (!fs) break;
                const uint8_t a0 = v.bytes()[p], a1 = v.bytes()[p + 1];
                has_ours |= (a0 == 'r' && a1 == 's') || (a0 == 'a' && (a1 == 'm' || a1 == 's' || a1 == 'r' || a1 == 'b'));
                p += 2 + fs;
            }
            const fadehip_aln *a = art_of[i] >= 0 ? &c.art[(size_t)art_of[i]] : nullptr;
            if (has_ours) {
                Rec r;
                r.d.assign(v.bytes(), v.bytes() + v.nbytes());
                tag_owned_record(r, rs[i], a, h);
                owned_t[t].emplace_back((uint32_t)i, std::move(r));
                continue;
            }
            size_t len = 4;  // "rs" 'C' value
            if (a) {
                std::string *st = &art_str[(size_t)art_of[i] * 4];
                artifact_strings(v, *a, h, st);
                for (int k = 0; k < 4; k++) len += 3 + st[k].size() + 1;
            }
            o.sfx_off[i + 1] = (uint32_t)len;
        }
    }, CPU_TAGS);
    for (auto &v : owned_t)
        for (auto &e : v) o.owned.push_back(std::move(e));
    for (size_t i = 0; i < n; i++) o.sfx_off[i + 1] += o.sfx_off[i];
    o.sfx.resize(o.sfx_off[n]);
    pool.parallel_for(nt, [&](size_t t) {
        static const char names[4][2] = {{'a', 'm'}, {'a', 's'}, {'a', 'r'}, {'a', 'b'}};
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
            if (o.sfx_off[i + 1] == o.sfx_off[i]) continue;
            uint8_t *d = o.sfx.data() + o.sfx_off[i];
            d[0] = 'r'; d[1] = 's'; d[2] = 'C'; d[3] = rs[i];  // bam_aux_update_int of a ubyte: the smallest type
            d += 4;
            if (art_of[i] >= 0) {
                const std::string *st = &art_str[(size_t)art_of[i] * 4];
                for (int k = 0; k < 4; k++) {
                    d[0] = (uint8_t)names[k][0]; d[1] = (uint8_t)names[k][1]; d[2] = 'Z';
                    memcpy(d + 3, st[k].c_str(), st[k].size() + 1);
                    d += 3 + st[k].size() + 1;
                }
            }
        }
    }, CPU_TAGS);
}

// FADE_TRACE=1: every start / stop pair of a named stage clock as a line on stderr at the end (ms since the first one)
struct StageTrace {
    bool on = getenv("FADE_TRACE") != nullptr;
    std::mutex m;
    std::chrono::steady_clock::time_point origin = std::chrono::steady_clock::now();
    struct Ev { const char *name; double t0, t1; };
    std::vector<Ev> ev;
    void add(const char *name, std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        std::lock_guard<std::mutex> l(m);
        ev.push_back(Ev{name, std::chrono::duration<double>(a - origin).count(), std::chrono::duration<double>(b - origin).count()});
    }
    void dump() {
        if (!on) return;
        std::lock_guard<std::mutex> l(m);
        std::sort(ev.begin(), ev.end(), [](const Ev &a, const Ev &b) { return a.t0 < b.t0; });
        for (auto &e : ev) fprintf(stderr, "[trace] %-8s %9.3f ms  +%8.3f ms\n", e.name, e.t0 * 1e3, (e.t1 - e.t0) * 1e3);
    }
};
static StageTrace g_trace;

struct StageClock {
    double t = 0;
    const char *name = nullptr;
    std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    void stop() {
        const auto t1 = std::chrono::steady_clock::now();
        t += std::chrono::duration<double>(t1 - t0).count();
        if (g_trace.on && name) g_trace.add(name, t0, t1);
    }
};

// joins the stage threads on every way out of annotate_main (an exception past a joinable std::thread is std::terminate)
struct StageThreads {
    std::vector<std::thread> th;
    std::function<void()> unblock;  // closes the queues so that the stages can finish
    ~StageThreads() {
        if (unblock) unblock();
        for (auto &t : th)
            if (t.joinable()) t.join();
    }
};

// Threads when -t is not given: the CPUs this process may actually use — its affinity mask and, in a container, the
// cgroup's CPU quota (a 16-CPU share of a 256-thread host must not get 255 threads) — at most 64.
static int default_threads(int cap = 64) {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : CPU_COUNT(&set), CPU_COUNT(&set));
    auto quota = [](const char *path, const char *period_path) -> double {
        FILE *f = fopen(path, "r");
        if (!f) return 0;
        char a[64] = "", b[64] = "";
        const int got = fscanf(f, "%63s %63s", a, b);
        fclose(f);
        if (got < 1 || strcmp(a, "max") == 0 || atof(a) <= 0) return 0;
        double period = got == 2 ? atof(b) : 0;
        if (period_path) {
            period = 0;
            if (FILE *g = fopen(period_path, "r")) {
                if (fscanf(g, "%lf", &period) != 1) period = 0;
                fclose(g);
            }
        }
        return period > 0 ? atof(a) / period : 0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);  // cgroup v2: "<quota|max> <period>"
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");  // v1
    if (q > 0) n = std::min(n, (int)(q + 0.999));
    return std::max(1, std::min(n, cap));
}

// seconds since the kernel started this process (/proc/self/stat field 22 against /proc/uptime), for -T only
static double since_process_start() {
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return 0;
    char buf[2048];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');  // comm may contain spaces
    unsigned long long start = 0;
    if (!p) return 0;
    p += 2;
    for (int field = 3; field < 22 && p; field++) { p = strchr(p, ' '); if (p) p++; }
    if (!p || sscanf(p, "%llu", &start) != 1) return 0;
    double up = 0;
    f = fopen("/proc/uptime", "r");
    if (!f) return 0;
    if (fscanf(f, "%lf", &up) != 1) up = 0;
    fclose(f);
    return up - (double)start / (double)sysconf(_SC_CLK_TCK);
}

// ------------------------------------------------------------------ lanes: one process per GPU
// `fade annotate --gpus N` on a BAM file: N lanes, each a process of its own with its own reader, device and writer, on
// disjoint BGZF virtual-offset ranges of the input; no lane sees another's records, the only exchange
// is the final sum of the stats.d counters (RCCL when the lanes sit on distinct devices).  The parent never touches the
// GPU: it cuts the file, starts the lanes, and puts their outputs together — BGZF blocks concatenate.
struct LaneEnv {
    bool on = false;
    int k = 0, n = 1, device = 0, rccl = 0;
    bool shard = false;  // --out-shards: this lane writes a complete file (header and end-of-file block of its own)
    LaneRange range;
    std::string cl, status_path, id_path;
};
static LaneEnv lane_env() {
    LaneEnv e;
    const char *l = getenv("FADE_LANE");
    if (!l) return e;
    unsigned long long a = 0, b = 0, c = 0, d = 0;
    if (sscanf(l, "%d/%d:%d:%d:%llu:%llu:%llu:%llu", &e.k, &e.n, &e.device, &e.rccl, &a, &b, &c, &d) != 8) return e;
    e.on = true;
    e.range.on = true;
    e.range.coff_start = a;
    e.range.first_rec = b;
    e.range.coff_end = c;
    e.range.end_rec = d;
    if (const char *v = getenv("FADE_LANE_CL")) e.cl = v;
    if (const char *v = getenv("FADE_LANE_STATUS")) e.status_path = v;
    if (const char *v = getenv("FADE_LANE_NCCL_ID")) e.id_path = v;
    if (const char *v = getenv("FADE_LANE_SHARD")) e.shard = atoi(v) != 0;
    return e;
}

// a plausible BAM record at buf[o ..): the checks a splitter can make without the chain from the start of the file
static bool plausible_record(const uint8_t *buf, size_t n, size_t o, int n_ref, size_t *next) {
    if (o + 36 > n) return false;
    uint32_t bs;
    memcpy(&bs, buf + o, 4);
    if (bs < 32 || bs > (1u << 24)) return false;
    const RecView v(buf + o + 4, std::min<size_t>(bs, n - o - 4));
    if (v.tid() < -1 || v.tid() >= n_ref || v.mtid() < -1 || v.mtid() >= n_ref || v.pos() < -1 || v.mpos() < -1) return false;
    const size_t lq = (size_t)v.l_qname();
    if (lq < 1 || v.l_seq() < 0) return false;
    const size_t need = 32 + lq + 4 * (size_t)v.n_cigar() + ((size_t)v.l_seq() + 1) / 2 + (size_t)v.l_seq();
    if (need > bs) return false;
    if (o + 4 + 32 + lq <= n) {
        const uint8_t *q = buf + o + 4 + 32;
        if (q[lq - 1] != 0) return false;
        for (size_t k = 0; k + 1 < lq; k++)
            if (q[k] < 33 || q[k] > 126) return false;
    }
    *next = o + 4 + bs;
    return true;
}

static int annotate_lanes_main(const std::string &cl, const Opts &o, bool *fall_back) {
    *fall_back = true;  // until the lanes have been started, a single-lane run can still take over
    const int n_lanes = o.gpus;
    const std::string &path = o.pos[1];
    struct stat sb;
    if (path == "-" || stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return 1;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 1;
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    // the BGZF blocks of the file
    std::vector<uint64_t> boff;
    std::vector<uint32_t> bsz;
    {
        uint64_t at = 0;
        uint8_t h[18];
        while (pread(fileno(f), h, 18, (off_t)at) == 18) {
            if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') return 1;  // not BGZF: one lane reads it
            const uint32_t bs = (uint32_t)(h[16] | (h[17] << 8)) + 1u;
            boff.push_back(at);
            bsz.push_back(bs);
            at += bs;
        }
    }
    if (boff.size() < (size_t)(8 * n_lanes)) return 1;  // too small to be worth cutting
    auto isize_of = [&](size_t b) -> uint32_t {
        uint8_t t[4] = {0, 0, 0, 0};
        if (pread(fileno(f), t, 4, (off_t)(boff[b] + bsz[b] - 4)) != 4) return 0;
        return (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    };
    auto inflate_blocks = [&](size_t b0, size_t nb, std::vector<uint8_t> &out, std::vector<size_t> &start) -> bool {
        out.clear();
        start.clear();
        for (size_t b = b0; b < std::min(b0 + nb, boff.size()); b++) {
            std::vector<uint8_t> comp(bsz[b]);
            if (pread(fileno(f), comp.data(), comp.size(), (off_t)boff[b]) != (ssize_t)comp.size()) return false;
            const size_t xlen = (size_t)comp[10] | ((size_t)comp[11] << 8), hl = 12 + xlen;
            if (comp.size() < hl + 8) return false;
            const uint32_t isz = isize_of(b);
            if (isz > 65536) return false;
            start.push_back(out.size());
            const size_t base = out.size();
            out.resize(base + isz);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            zs.next_in = comp.data() + hl;
            zs.avail_in = (uInt)(comp.size() - hl - 8);
            zs.next_out = out.data() + base;
            zs.avail_out = isz;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != isz) return false;
        }
        start.push_back(out.size());
        return true;
    };
    // header, number of references
    int n_ref = 0;
    size_t hdr_bytes = 0;
    {
        Pool small(1);
        Reader rd(path, &small);
        if (!rd.is_bam()) return 1;
        n_ref = (int)rd.header().names.size();
        hdr_bytes = rd.bam_header_bytes();
    }
    // lane starts: lane 0 behind the header; lane k at the first offset of block k * nblk / N from which a chain of
    // plausible records runs through the next blocks.  The guess is checked for real by the lane before, whose own chain
    // must end exactly there.
    std::vector<LaneRange> lr((size_t)n_lanes);
    {
        size_t b = 0;
        uint64_t cum = 0;
        while (b < boff.size() && cum + isize_of(b) <= hdr_bytes) cum += isize_of(b++);
        if (b >= boff.size()) return 1;
        lr[0].on = true;
        lr[0].coff_start = boff[b];
        lr[0].first_rec = hdr_bytes - cum;
    }
    for (int k = 1; k < n_lanes; k++) {
        bool found = false;
        for (size_t b = boff.size() * (size_t)k / (size_t)n_lanes; b + 4 < boff.size() && !found; b++) {
            std::vector<uint8_t> buf;
            std::vector<size_t> st;
            if (!inflate_blocks(b, 4, buf, st)) return 1;
            for (size_t o0 = 0; o0 < st[1] && !found; o0++) {
                size_t o = o0, nx = 0;
                int cnt = 0;
                while (plausible_record(buf.data(), buf.size(), o, n_ref, &nx) && nx <= buf.size()) { o = nx; cnt++; }
                // the chain must run to the end of what was inflated (the last record may stick out) through at least 4 records
                if (cnt >= 4 && (o == buf.size() || (o + 4 <= buf.size() && plausible_record(buf.data(), buf.size(), o, n_ref, &nx)) || buf.size() - o < 36)) {
                    lr[(size_t)k].on = true;
                    lr[(size_t)k].coff_start = boff[b];
                    lr[(size_t)k].first_rec = o0;
                    found = true;
                }
            }
        }
        if (!found) return 1;
        lr[(size_t)k - 1].coff_end = lr[(size_t)k].coff_start;
        lr[(size_t)k - 1].end_rec = lr[(size_t)k].first_rec;
    }
    // devices: 0 .. N-1, or FADE_DEVICE_MAP (tests put two lanes on one device; RCCL wants distinct devices)
    std::vector<int> devmap((size_t)n_lanes);
    for (int d = 0; d < n_lanes; d++) devmap[(size_t)d] = d;
    bool distinct = true;
    if (const char *dm = getenv("FADE_DEVICE_MAP")) {
        int d = 0;
        for (const char *q = dm; *q && d < n_lanes; d++) {
            devmap[(size_t)d] = atoi(q);
            q = strchr(q, ',');
            if (!q) break;
            q++;
        }
        for (int a = 0; a < n_lanes; a++)
            for (int b2 = a + 1; b2 < n_lanes; b2++)
                if (devmap[(size_t)a] == devmap[(size_t)b2]) distinct = false;
    }
    // (a lane is a whole `fade annotate` of its own: its share of the cores, up to the 64 a single run takes — an 8-GPU node
    // has them)
    const int threads_each = std::max(1, std::min(64, (o.threads > 0 ? o.threads : default_threads(64 * n_lanes)) / n_lanes));
    const bool shards = !o.out_shards.empty();
    // Where the lanes write.  --out-shards: every lane a complete file of its own (header, its records, end-of-file block),
    // nothing is merged (BASELINE config 4: "sharded per GPU").  Otherwise ONE stream on stdout: lane 0 writes straight into
    // it (header first), lanes 1 .. N-1 into files of a private directory that this process forwards WHILE the lanes run —
    // into their final place at once when stdout is a file that can be written at an offset (a lane's place is known as soon
    // as the lanes before it have ended, and the copies of all lanes proceed side by side), one lane after the other when it
    // is a pipe.
    char dir[512] = "";
    {
        const char *tmpd = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
        snprintf(dir, sizeof dir, "%s/fade_lanes_XXXXXX", tmpd);
        if (!mkdtemp(dir)) { fprintf(stderr, "[E::fade annotate] cannot make a directory under %s: %s\n", tmpd, strerror(errno)); return 1; }  // (0700, a name nobody can guess)
    }
    std::vector<std::string> outp((size_t)n_lanes), stp((size_t)n_lanes);
    std::vector<int> ofd((size_t)n_lanes, -1);  // lanes 1 ..: this process's read side of the lane's file
    std::vector<pid_t> pids((size_t)n_lanes, -1);
    const std::string idp = std::string(dir) + "/ncclid";
    auto cleanup = [&] {
        for (int k = 0; k < n_lanes; k++) {
            if (ofd[(size_t)k] >= 0) close(ofd[(size_t)k]);
            if (!shards && !outp[(size_t)k].empty()) unlink(outp[(size_t)k].c_str());
            unlink(stp[(size_t)k].c_str());
        }
        unlink(idp.c_str());
        unlink((idp + ".tmp").c_str());
        rmdir(dir);
    };
    *fall_back = false;  // from here on the lanes own the run: a failure is reported, nothing is run twice
    if (o.timing)
        for (int k = 0; k < n_lanes; k++)
            fprintf(stderr, "[timing] lane %d of %d: device %d, blocks from file offset %llu (first record at +%llu) to %llu (+%llu)\n", k, n_lanes, devmap[(size_t)k],
                    (unsigned long long)lr[(size_t)k].coff_start, (unsigned long long)lr[(size_t)k].first_rec, (unsigned long long)lr[(size_t)k].coff_end,
                    (unsigned long long)lr[(size_t)k].end_rec);
    fprintf(stderr, "[W::fade annotate] Output SAM/BAM will not be sorted (regardless of prior sorting)\n");
    fflush(stdout);
    const auto t_spawn = std::chrono::steady_clock::now();
    auto secs = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spawn).count(); };
    bool ok = true;
    for (int k = 0; k < n_lanes; k++) stp[(size_t)k] = std::string(dir) + "/" + std::to_string(k) + ".status";
    // (FADE_LANES_LIVE=m: at most m lanes at a time, the next one starts when one ends — for boxes that allow few processes
    // on a device, i.e. tests that put eight lanes on one GPU)
    const int live_cap = std::max(1, std::min(n_lanes, getenv("FADE_LANES_LIVE") ? atoi(getenv("FADE_LANES_LIVE")) : n_lanes));
    int next_lane = 0;
    // the lanes sum their counters among themselves over RCCL when each has a device of its own and all of them run at once
    // (FADE_LANES_RCCL=1 asks for it regardless: a test of what RCCL says to two ranks on one device; =0 leaves the sum to
    // this process, which adds up the lanes' reports either way)
    const char *rccl_env = getenv("FADE_LANES_RCCL");
    const bool lanes_rccl = live_cap == n_lanes && (rccl_env ? atoi(rccl_env) != 0 : distinct);
    auto start_lane = [&](int k) -> bool {
        int wfd = -1;  // the lane's stdout (lane 0 of a merged run: this process's own)
        if (shards) {
            outp[(size_t)k] = o.out_shards + "." + std::to_string(k) + ((o.bam || o.ubam) ? ".bam" : ".sam");
            wfd = open(outp[(size_t)k].c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
        } else if (k > 0) {
            outp[(size_t)k] = std::string(dir) + "/" + std::to_string(k) + ".out";
            wfd = open(outp[(size_t)k].c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
            if (wfd >= 0) ofd[(size_t)k] = open(outp[(size_t)k].c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        }
        if ((shards || k > 0) && (wfd < 0 || (!shards && ofd[(size_t)k] < 0))) {
            fprintf(stderr, "[E::fade annotate] cannot create %s: %s\n", outp[(size_t)k].c_str(), strerror(errno));
            if (wfd >= 0) close(wfd);
            return false;
        }
        char lane[256];
        snprintf(lane, sizeof lane, "%d/%d:%d:%d:%llu:%llu:%llu:%llu", k, n_lanes, devmap[(size_t)k], lanes_rccl ? 1 : 0, (unsigned long long)lr[(size_t)k].coff_start,
                 (unsigned long long)lr[(size_t)k].first_rec, (unsigned long long)lr[(size_t)k].coff_end, (unsigned long long)lr[(size_t)k].end_rec);
        std::vector<std::string> envs;
        for (char **e = environ; *e; e++)
            if (strncmp(*e, "FADE_LANE", 9) != 0) envs.push_back(*e);
        envs.push_back(std::string("FADE_LANE=") + lane);
        envs.push_back("FADE_LANE_CL=" + cl);
        envs.push_back("FADE_LANE_STATUS=" + stp[(size_t)k]);
        envs.push_back("FADE_LANE_NCCL_ID=" + idp);
        if (shards) envs.push_back("FADE_LANE_SHARD=1");
        // (a lane whose file this process reads while it grows must write it front to back: side-by-side pwrites would leave
        // holes that read as zeros for a moment)
        if (!shards && k > 0) {
            for (auto &e : envs)
                if (e.rfind("FADE_BAM_WRITERS=", 0) == 0) e = "FADE_BAM_WRITERS=0";
            envs.push_back("FADE_BAM_WRITERS=0");
        }
        std::vector<char *> envp;
        for (auto &e : envs) envp.push_back(const_cast<char *>(e.c_str()));
        envp.push_back(nullptr);
        std::vector<std::string> args = {"/proc/self/exe", "annotate", "-t", std::to_string(threads_each), "--min-length", std::to_string(o.floor_len),
                                         "-w", std::to_string(o.window), "--batch", std::to_string(o.batch)};
        if (o.bam) args.push_back("-b");
        if (o.ubam) args.push_back("-u");
        if (o.timing) args.push_back("--timing");
        args.push_back(o.pos[1]);
        args.push_back(o.pos[2]);
        std::vector<char *> argv;
        for (auto &a : args) argv.push_back(const_cast<char *>(a.c_str()));
        argv.push_back(nullptr);
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        if (wfd >= 0) posix_spawn_file_actions_adddup2(&fa, wfd, 1);
        const int rc = posix_spawn(&pids[(size_t)k], "/proc/self/exe", &fa, nullptr, argv.data(), envp.data());
        posix_spawn_file_actions_destroy(&fa);
        if (wfd >= 0) close(wfd);
        if (rc != 0) {
            fprintf(stderr, "[E::fade annotate] cannot start lane %d: %s\n", k, strerror(rc));
            pids[(size_t)k] = -1;
            return false;
        }
        return true;
    };
    while (ok && next_lane < live_cap) ok = start_lane(next_lane++);
    // ---- the lanes run; their outputs are put together meanwhile
    struct Merge {
        std::mutex m;
        std::condition_variable cv;
        std::vector<char> ended;      // lane k has ended well: its file has its final size
        std::vector<uint64_t> size;   // ... which is this
        std::vector<char> copied;     // lane k's bytes are all in the output
        bool abort = false;
        std::string err;
    } mg;
    mg.ended.assign((size_t)n_lanes, 0);
    mg.size.assign((size_t)n_lanes, 0);
    mg.copied.assign((size_t)n_lanes, 0);
    struct stat so;
    const int oflags = fcntl(1, F_GETFL);
    const off_t base = lseek(1, 0, SEEK_CUR);
    // (FADE_LANES_MERGE=stream: one lane after the other even into a file — what a pipe gets; tests compare the two)
    const bool placed = !shards && fstat(1, &so) == 0 && S_ISREG(so.st_mode) && base >= 0 && oflags >= 0 && !(oflags & O_APPEND) &&
                        !(getenv("FADE_LANES_MERGE") && strcmp(getenv("FADE_LANES_MERGE"), "stream") == 0);
    std::vector<double> t_end((size_t)n_lanes, 0), t_copied((size_t)n_lanes, 0);
    std::vector<std::thread> copiers;
    if (!shards && ok)
        for (int k = 1; k < n_lanes; k++)
            copiers.emplace_back([&, k] {
                // lane k's bytes may go out once their place is known: behind the FINAL sizes of lanes 0 .. k-1 (placed), or
                // behind the last byte of lane k-1 in the stream
                uint64_t at = 0;
                {
                    std::unique_lock<std::mutex> l(mg.m);
                    mg.cv.wait(l, [&] {
                        if (mg.abort) return true;
                        for (int j = 0; j < k; j++)
                            if (!mg.ended[(size_t)j] || (!placed && !mg.copied[(size_t)j])) return false;
                        return true;
                    });
                    if (mg.abort) return;
                    for (int j = 0; j < k; j++) at += mg.size[(size_t)j];
                }
                std::vector<char> buf((size_t)4 << 20);
                uint64_t pos = 0;
                for (;;) {
                    const ssize_t got = pread(ofd[(size_t)k], buf.data(), buf.size(), (off_t)pos);
                    if (got < 0 && errno == EINTR) continue;
                    if (got < 0) { std::lock_guard<std::mutex> l(mg.m); mg.abort = true; mg.err = std::string("read error on a lane's output: ") + strerror(errno); mg.cv.notify_all(); return; }
                    if (got == 0) {
                        std::unique_lock<std::mutex> l(mg.m);
                        if (mg.abort) return;
                        if (mg.ended[(size_t)k] && pos >= mg.size[(size_t)k]) break;
                        mg.cv.wait_for(l, std::chrono::microseconds(300));  // (the lane is still writing)
                        continue;
                    }
                    size_t done = 0;
                    while (done < (size_t)got) {
                        const ssize_t w = placed ? pwrite(1, buf.data() + done, (size_t)got - done, (off_t)((uint64_t)base + at + pos + done))
                                                 : write(1, buf.data() + done, (size_t)got - done);
                        if (w < 0 && errno == EINTR) continue;
                        if (w <= 0) { std::lock_guard<std::mutex> l(mg.m); mg.abort = true; mg.err = "write error on the output stream"; mg.cv.notify_all(); return; }
                        done += (size_t)w;
                    }
                    pos += (uint64_t)got;
                }
                std::lock_guard<std::mutex> l(mg.m);
                mg.copied[(size_t)k] = 1;
                t_copied[(size_t)k] = secs();
                mg.cv.notify_all();
            });
    // reap whichever lane ends next; one that fails takes the others with it (a lane waiting in ncclCommInitRank for a dead
    // peer would wait for ever)
    int live = 0;
    for (int k = 0; k < n_lanes; k++) live += pids[(size_t)k] > 0;
    auto kill_all = [&] {
        for (int k = 0; k < n_lanes; k++)
            if (pids[(size_t)k] > 0) kill(pids[(size_t)k], SIGKILL);
    };
    if (!ok) kill_all();
    while (live > 0) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) { if (errno == EINTR) continue; break; }
        int k = 0;
        while (k < n_lanes && pids[(size_t)k] != p) k++;
        if (k == n_lanes) continue;
        pids[(size_t)k] = -1;
        live--;
        const bool good = WIFEXITED(st) && WEXITSTATUS(st) == 0;
        bool aborted;
        {
            std::lock_guard<std::mutex> l(mg.m);
            aborted = mg.abort;
        }
        if (!good || aborted) {
            if (ok && !good) fprintf(stderr, "[E::fade annotate] lane %d of %d failed\n", k, n_lanes);
            if (ok) {
                ok = false;
                kill_all();
                std::lock_guard<std::mutex> l(mg.m);
                mg.abort = true;
                mg.cv.notify_all();
            }
            continue;
        }
        uint64_t sz = 0;
        if (!shards) {
            if (k == 0) {
                const off_t e = lseek(1, 0, SEEK_CUR);  // (lane 0 wrote through this very file description)
                sz = placed && e >= base ? (uint64_t)(e - base) : 0;
            } else {
                struct stat sb2;
                if (fstat(ofd[(size_t)k], &sb2) == 0) sz = (uint64_t)sb2.st_size;
            }
        }
        {
            std::lock_guard<std::mutex> l(mg.m);
            mg.ended[(size_t)k] = 1;
            mg.size[(size_t)k] = sz;
            if (k == 0) { mg.copied[0] = 1; t_copied[0] = secs(); }
            t_end[(size_t)k] = secs();
            mg.cv.notify_all();
        }
        if (ok && next_lane < n_lanes) {
            if (start_lane(next_lane++)) live++;
            else {
                ok = false;
                kill_all();
                std::lock_guard<std::mutex> l(mg.m);
                mg.abort = true;
                mg.cv.notify_all();
            }
        }
    }
    for (auto &t : copiers) t.join();
    if (!mg.err.empty()) { fprintf(stderr, "[E::fade annotate] %s\n", mg.err.c_str()); ok = false; }
    int64_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t n_oversize = 0;
    for (int k = 0; k < n_lanes && ok; k++) {
        FILE *s = fopen(stp[(size_t)k].c_str(), "r");
        long long v[9], red[8];
        if (!s || fscanf(s, "%lld %lld %lld %lld %lld %lld %lld %lld %lld", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]) != 9) {
            fprintf(stderr, "[E::fade annotate] lane %d left no report\n", k);
            ok = false;
        } else {
            for (int q = 0; q < 8; q++) totals[q] += v[q];
            n_oversize += v[8];
            // lanes on distinct devices have summed the counters among themselves (RCCL): every lane then holds the totals
            if (lanes_rccl && fscanf(s, "%lld %lld %lld %lld %lld %lld %lld %lld", &red[0], &red[1], &red[2], &red[3], &red[4], &red[5], &red[6], &red[7]) == 8 && k == n_lanes - 1)
                for (int q = 0; q < 8; q++)
                    if (red[q] != totals[q]) { fprintf(stderr, "[E::fade annotate] the lanes' RCCL sum differs from the sum of their reports\n"); ok = false; break; }
        }
        if (s) fclose(s);
    }
    if (!ok) { cleanup(); return 1; }
    if (!shards) {
        if (placed) {
            uint64_t all = 0;
            for (int k = 0; k < n_lanes; k++) all += mg.size[(size_t)k];
            if (lseek(1, (off_t)((uint64_t)base + all), SEEK_SET) < 0) { fprintf(stderr, "[E::fade annotate] cannot seek the output: %s\n", strerror(errno)); cleanup(); return 1; }
        }
        if (o.bam || o.ubam) {
            size_t done = 0;
            while (done < sizeof BGZF_EOF) {
                const ssize_t w = write(1, BGZF_EOF + done, sizeof BGZF_EOF - done);
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) { fprintf(stderr, "[E::fade annotate] write error on the output stream\n"); cleanup(); return 1; }
                done += (size_t)w;
            }
        }
    }
    if (o.timing) {
        double last_end = 0, last_copy = 0;
        for (int k = 0; k < n_lanes; k++) { last_end = std::max(last_end, t_end[(size_t)k]); last_copy = std::max(last_copy, t_copied[(size_t)k]); }
        fprintf(stderr, "[timing] lanes: %s; the last lane ended after %.3f s, the last byte was in place after %.3f s (merge behind the lanes: %.3f s);",
                shards ? "a complete file per lane, nothing merged" : placed ? "lanes 1.. copied into their final place while the lanes ran" : "lanes forwarded in order while the lanes ran",
                last_end, shards ? last_end : last_copy, shards ? 0.0 : std::max(0.0, last_copy - last_end));
        for (int k = 0; k < n_lanes; k++) fprintf(stderr, " lane %d ended %.3f%s", k, t_end[(size_t)k], k + 1 < n_lanes ? "," : "\n");
    }
    cleanup();
    if (n_oversize)
        fprintf(stderr, "[W::fade annotate] %lld soft-clipped reads were not re-aligned: read or window beyond the kernels' limits\n", (long long)n_oversize);
    if (o.stats) {
        const double rc = (double)std::max<int64_t>(totals[0], 1);
        fprintf(stderr, "read count:\t%lld\nClipped %%:\t%g\n%% With Supplementary alns:\t%g\nArtifact rate:\t%g\n"
                        "%% With Supplementary alns and artifacts:\t%g\nArtifact rate left only:\t%g\nArtifact rate right only:\t%g\n",
                (long long)totals[0], totals[1] / rc, totals[2] / rc, totals[4] / rc, totals[3] / rc, totals[6] / rc, totals[7] / rc);
    }
    return 0;
}

// `fade annotate -b in.bam ref.fa` with the whole file path on the device (fadehip_bam_*): this process reads compressed
// bytes and writes compressed bytes; inflate, framing, annotateTask, the tags and deflate are kernels.  Three threads
// around the library's two halves: [reader: file -> pinned buffers, cut at BGZF member boundaries] -> [this thread:
// front] -> [back thread: back] -> [writer thread: fwrite].  FADE_BAM_DEVICE=0 (or any input that is not a BAM file, any
// output that is not BAM, --gpus N) takes the host pipeline of annotate_main instead.
static int annotate_stream_main(const std::string &cl, const Opts &o, bool *fall_back) {
    *fall_back = true;
    const LaneEnv lane = lane_env();  // set when this process is one lane of a `--gpus N` run (annotate_lanes_main): a range of the file
    const std::string &path = o.pos[1];
    struct stat sb;
    if (!(o.bam || o.ubam) || path == "-" || stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return 1;
    StageClock ck_total, ck_fasta, ck_upload, ck_front, ck_back, ck_fwrite, ck_fread, ck_inflate;
    ck_fasta.name = "fasta"; ck_upload.name = "upload"; ck_front.name = "front"; ck_back.name = "back"; ck_fwrite.name = "fwrite"; ck_fread.name = "fread"; ck_inflate.name = "inflate";
    ck_total.start();
    const int nthreads = o.threads > 0 ? o.threads : default_threads();
    // Who inflates: the device (FADE_BAM_INFLATE=device: only compressed bytes cross PCIe, the host cores stay free) or this
    // process's pool (host, the default with 8 threads or more: the cores have nothing else to do here, and the device then
    // spends its time on the compressor, which is its slowest kernel).
    const char *inf_env = getenv("FADE_BAM_INFLATE");
    const bool host_inflate = inf_env ? strcmp(inf_env, "host") == 0 : nthreads >= 8;
    Pool pool(host_inflate ? nthreads : std::min(nthreads, 4));
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 1;
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    // header; the member that holds the first record and the record's offset in its payload
    Header hdr;
    uint64_t coff = 0;
    uint32_t first_rec = 0;
    try {
        Reader rd(path, &pool);
        if (!rd.is_bam()) return 1;
        hdr = rd.header();
        const size_t hdr_bytes = rd.bam_header_bytes();
        uint64_t at = 0, cum = 0;
        if (lane.on) {
            coff = lane.range.coff_start;
            first_rec = (uint32_t)lane.range.first_rec;
        }
        while (!lane.on) {
            uint8_t h[18], t[4];
            if (pread(fileno(f), h, 18, (off_t)at) != 18) return 1;
            if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') return 1;  // (other gzip subfields first: the host path reads it)
            const uint32_t bs = (uint32_t)(h[16] | (h[17] << 8)) + 1u;
            if (pread(fileno(f), t, 4, (off_t)(at + bs - 4)) != 4) return 1;
            const uint32_t isz = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            if (cum + isz > hdr_bytes || (uint64_t)at + bs >= (uint64_t)sb.st_size) { coff = at; first_rec = (uint32_t)(hdr_bytes - cum); break; }
            cum += isz;
            at += bs;
        }
        if (first_rec > 65536) return 1;
    } catch (const std::exception &e) {
        return 1;  // the host path reports what is wrong with the file
    }
    *fall_back = false;
    if (o.timing) fprintf(stderr, "[timing] since process start %.3f s (annotate begins; file path on the device)\n", since_process_start());
    if (!lane.on) fprintf(stderr, "[W::fade annotate] Output SAM/BAM will not be sorted (regardless of prior sorting)\n");
    // a lane reads the members in front of coff_end whole and, of the member AT coff_end, the end_rec bytes in front of the next
    // lane's first record
    uint64_t read_end = (uint64_t)sb.st_size;
    uint32_t tail_trim = 0;
    const bool limited = lane.on && lane.range.coff_end != 0;
    if (limited) {
        read_end = lane.range.coff_end;
        if (lane.range.end_rec > 0) {
            uint8_t h[18];
            if (pread(fileno(f), h, 18, (off_t)lane.range.coff_end) != 18 || h[12] != 'B' || h[13] != 'C') { fprintf(stderr, "[E::fade annotate] lane range ends at no BGZF member\n"); return 1; }
            const uint64_t bsz = (uint64_t)(h[16] | (h[17] << 8)) + 1u;
            read_end += bsz;
            // a lane's last member is cut where the next lane's first record starts: by this process when it inflates, by the
            // library (tail_trim) when the device does
            uint8_t t[4];
            if (pread(fileno(f), t, 4, (off_t)(read_end - 4)) != 4) return 1;
            const uint32_t isz = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            if (lane.range.end_rec > isz) { fprintf(stderr, "[E::fade annotate] lane range ends beyond its last block\n"); return 1; }
            tail_trim = isz - (uint32_t)lane.range.end_rec;
        }
    }
    fadehip_ctx *ctx = nullptr;
    fadehip_bam_stream *st = nullptr;
    struct Guard {
        fadehip_ctx *&c;
        fadehip_bam_stream *&s;
        std::vector<void *> pinned, owned;  // registered with the ctx; allocated here
        ~Guard() {
            if (s) fadehip_bam_close(s);
            for (void *p : pinned) fadehip_host_free(c, p);
            fadehip_destroy(c);
            for (void *p : owned) free(p);
        }
    } guard{ctx, st, {}, {}};
    try {
        fadehip_params prm;
        fadehip_params_default(&prm);
        setenv("FADEHIP_TAIL_CUS", "0", 0);  // one batch at a time: no CU-masked stream, fewer queues
        setenv("FADEHIP_BLOCKING_SYNC", "1", 0);  // the threads that wait for the device sleep: the cores are the inflater's
        int device = 0;
        if (lane.on) device = lane.device;
        else if (const char *dm = getenv("FADE_DEVICE_MAP")) device = atoi(dm);
        if (hdr.names.empty()) { fprintf(stderr, "[E::fade annotate] input has no @SQ lines\n"); return 1; }
        // FADE_BAM_CHUNK_MB: bytes per front call — compressed when the device inflates (default 128), inflated when the
        // host does (default 32: the first call starts earlier and the last one drains sooner; 64 and 128 measured slower).
        const size_t chunk = (size_t)std::max(1, getenv("FADE_BAM_CHUNK_MB") ? atoi(getenv("FADE_BAM_CHUNK_MB")) : host_inflate ? 32 : 128) << 20;
        // The device is brought up on a thread of its own, beside everything the host can do without it (the FASTA, the input's
        // first members read and inflated): the ctx (the HIP runtime's own start is most of it), then the stream and what its
        // first calls would otherwise make one by one (fadehip_bam_prepare).  The genome goes up as soon as the ctx is there.
        std::string create_err;
        std::promise<int> ctx_made;
        std::future<int> ctx_ready = ctx_made.get_future();
        std::future<int> creating = std::async(std::launch::async, [&]() -> int {
            if (fadehip_create(&ctx, device, &prm)) { create_err = fadehip_last_error(nullptr); ctx_made.set_value(1); return 1; }
            ctx_made.set_value(0);
            std::vector<const char *> names;
            for (auto &n : hdr.names) names.push_back(n.c_str());
            fadehip_bam_config cfg;
            memset(&cfg, 0, sizeof cfg);
            cfg.floor_len = o.floor_len;
            cfg.window = o.window;
            cfg.n_ref = (int32_t)names.size();
            cfg.ref_names = names.data();
            cfg.first_record = first_rec;
            cfg.tail_trim = host_inflate ? 0 : tail_trim;
            cfg.flags = o.ubam ? FADEHIP_BAM_STORED : 0;  // -u: uncompressed BGZF (util.d:65-76, SAMWriterTypes.UBAM)
            if (fadehip_bam_open(ctx, &cfg, &st)) { create_err = fadehip_last_error(ctx); return 1; }
            const size_t call_bytes = host_inflate ? chunk : std::min<size_t>(chunk * 3, (size_t)1 << 30);
            if (!(getenv("FADE_BAM_PREPARE") && atoi(getenv("FADE_BAM_PREPARE")) == 0) && fadehip_bam_prepare(st, call_bytes)) { create_err = fadehip_last_error(ctx); return 1; }
            return 0;
        });
        // ---- the stages
        // Compressed bytes are read by a thread of their own, HEAD bytes into a buffer, so that a member cut by the end of one
        // read is completed by copying its beginning in front of the next.
        const size_t ccap = host_inflate ? std::max<size_t>(chunk / 2, 1 << 20) : chunk, HEAD = 65536 + 64;
        constexpr int NBUF = 3;
        struct CBuf { uint8_t *p = nullptr; size_t n = 0; uint64_t off = 0; bool eof = false; std::vector<uint8_t> own; };
        struct In { uint8_t *p = nullptr, *pin = nullptr; size_t n = 0; bool last = false, compressed = false; int cbuf = -1; };
        CBuf cbufs[NBUF];
        In bufs[NBUF];
        // (ordinary memory for now — the ctx is still being made; registered with it once it is there)
        std::vector<std::pair<void *, size_t>> raw_bufs;
        auto pinned = [&](size_t bytes) -> uint8_t * {
            const size_t al = (size_t)1 << 21, n = (bytes + al - 1) & ~(al - 1);
            void *q = aligned_alloc(al, n);
            if (!q) throw std::runtime_error("out of memory");
            guard.owned.push_back(q);
            raw_bufs.emplace_back(q, n);
            return (uint8_t *)q;
        };
        // FADE_BAM_DEVICE_SHARE=n: with the pool inflating, every n-th call's members go to the device as they are (inflated by
        // the kernel of FADE_BAM_INFLATE=device): the cores are the bottleneck of a file-to-file run once the device's half is
        // fast, and the device has time to spare for a share of the inflating.  0: none.
        const int dev_share = host_inflate ? std::max(0, getenv("FADE_BAM_DEVICE_SHARE") ? atoi(getenv("FADE_BAM_DEVICE_SHARE")) : 0) : 0;
        std::atomic<int> cbuf_refs[NBUF];
        for (auto &r : cbuf_refs) r = 0;
        // (Reading the members where the page cache has them — the file mapped, its pages entered by MADV_POPULATE_READ —
        // was measured and is slower than pread into a buffer: the pool inflates 20-30 % slower out of the mapping.)
        for (int k = 0; k < NBUF; k++) {
            if (host_inflate) {
                if (dev_share) cbufs[k].p = pinned(HEAD + ccap + 64);  // (some of its members cross PCIe as they are)
                else {
                    cbufs[k].own.resize(HEAD + ccap + 64);
                    cbufs[k].p = cbufs[k].own.data();
                }
                bufs[k].pin = bufs[k].p = pinned(chunk + 65536 + 64);
            } else {
                cbufs[k].p = pinned(HEAD + ccap + 64);  // (handed to front as they are: pinned)
            }
        }
        BoundedQueue<int> q_cfree(NBUF + 1), q_cfull(NBUF + 1), q_free(NBUF + 1), q_full(NBUF + 1), q_done(FADEHIP_BAM_CHUNKS + 1);
        struct OutRef { const uint8_t *p; size_t n; };
        // A call's bytes stay valid during the next back call only (include/fadehip.h: two staging buffers in turn), so the
        // back thread may run at most one call ahead of the call whose bytes the writer still holds: two credits, one taken
        // before each back call and given back when its bytes have been written.
        BoundedQueue<OutRef> q_write(2);
        BoundedQueue<int> q_credit(2);
        q_credit.push(0);
        q_credit.push(0);
        for (int k = 0; k < NBUF; k++) { q_cfree.push(k); q_free.push(k); }
        std::string stage_err;
        std::mutex err_m;
        auto set_err = [&](const std::string &e) {
            std::lock_guard<std::mutex> l(err_m);
            if (stage_err.empty()) stage_err = e;
        };
        std::atomic<bool> abort_all{false};
        StageThreads stages;
        stages.unblock = [&] {
            abort_all = true;
            q_cfree.close(); q_cfull.close(); q_free.close(); q_full.close(); q_done.close(); q_write.close(); q_credit.close();
        };
        // the members of buf[0, got): (offset, size, header length); stops in front of one that is not whole
        struct Mem { size_t off, size, hl; uint32_t isz; };
        auto scan_members = [&](const uint8_t *buf, size_t got, std::vector<Mem> &ms) -> size_t {
            size_t w = 0;
            ms.clear();
            while (w + 18 <= got) {
                const uint8_t *m = buf + w;
                if (m[0] != 0x1f || m[1] != 0x8b) throw std::runtime_error("not a BGZF member in " + path);
                const size_t xlen = (size_t)m[10] | ((size_t)m[11] << 8);
                if (w + 12 + xlen > got) break;
                size_t bs = 0;
                for (size_t x = 12; x + 4 <= 12 + xlen;) {
                    const size_t sl = (size_t)m[x + 2] | ((size_t)m[x + 3] << 8);
                    if (m[x] == 'B' && m[x + 1] == 'C' && sl == 2 && x + 6 <= 12 + xlen) bs = ((size_t)m[x + 4] | ((size_t)m[x + 5] << 8)) + 1;
                    x += 4 + sl;
                }
                if (bs < 12 + xlen + 10) throw std::runtime_error("BGZF member without a usable BC subfield in " + path);
                if (w + bs > got) break;
                uint32_t isz;
                memcpy(&isz, m + bs - 4, 4);
                if (isz > 65536) throw std::runtime_error("corrupt BGZF block (ISIZE beyond 64 KiB) in " + path);
                ms.push_back(Mem{w, bs, 12 + xlen, isz});
                w += bs;
            }
            return w;
        };
        stages.th.emplace_back([&] {  // file reader
            try {
                uint64_t at = coff;
                int k;
                bool eof = false;
                while (!eof && !abort_all && q_cfree.pop(k)) {
                    CBuf &c = cbufs[k];
                    ck_fread.start();
                    size_t got = 0;
                    c.off = at;
                    while (got < ccap) {
                        const size_t want = (size_t)std::min<uint64_t>(ccap - got, read_end - at);
                        const ssize_t r = want ? pread(fileno(f), c.p + HEAD + got, want, (off_t)at) : 0;
                        if (r < 0) throw std::runtime_error("read error on " + path);
                        if (r == 0) { eof = true; break; }
                        got += (size_t)r;
                        at += (uint64_t)r;
                    }
                    ck_fread.stop();
                    c.n = got;
                    c.eof = eof;
                    q_cfull.push(k);
                }
            } catch (const std::exception &e) {
                set_err(e.what());
            }
            q_cfull.close();
        });
        stages.th.emplace_back([&] {  // members: cut (and, in host mode, inflated on the pool) into the calls' buffers
            try {
                std::vector<uint8_t> tail;
                std::vector<Mem> ms;
                int ck;
                bool ended = false;
                unsigned batch_no = 0;
                while (!ended && !abort_all && q_cfull.pop(ck)) {
                    CBuf &c = cbufs[ck];
                    if (tail.size() > HEAD) throw std::runtime_error("BGZF member larger than 64 KiB in " + path);
                    uint8_t *cb = c.p + HEAD - tail.size();
                    if (!tail.empty()) memcpy(cb, tail.data(), tail.size());
                    const size_t got = tail.size() + c.n;
                    const uint64_t cb_off = c.off - tail.size();  // file offset of cb[0]
                    const size_t w = scan_members(cb, got, ms);
                    tail.assign(cb + w, cb + got);
                    if (c.eof && !tail.empty()) throw std::runtime_error("the file ends inside a BGZF member: " + path);
                    ended = c.eof;
                    if (!host_inflate) {
                        // the compressed bytes go to the device as they lie; the buffer comes back when front is done with it
                        In in;
                        in.p = cb; in.n = w; in.last = ended; in.cbuf = ck;
                        int slot;
                        if (!q_free.pop(slot)) break;
                        bufs[slot] = in;
                        q_full.push(slot);
                        continue;
                    }
                    // host mode: batches of members whose payloads fill a pinned buffer
                    size_t m0 = 0;
                    cbuf_refs[ck] = 1;  // (this thread's own hold on the compressed bytes)
                    do {
                        size_t total = 0, m1 = m0;
                        while (m1 < ms.size() && total + ms[m1].isz <= chunk) total += ms[m1++].isz;
                        int slot;
                        if (!q_free.pop(slot)) { ended = true; break; }
                        In &b = bufs[slot];
                        b.p = b.pin;
                        b.compressed = false;
                        // a call of the device's share: the members as they lie (never the run's last call, whose tail a lane trims here)
                        if (dev_share && m1 > m0 && ++batch_no % (unsigned)dev_share == 0 && !(ended && m1 == ms.size())) {
                            b.p = cb + ms[m0].off;
                            b.n = ms[m1 - 1].off + ms[m1 - 1].size - ms[m0].off;
                            b.last = false;
                            b.compressed = true;
                            b.cbuf = ck;
                            cbuf_refs[ck]++;
                            q_full.push(slot);
                            m0 = m1;
                            continue;
                        }
                        std::vector<size_t> ooff(m1 - m0 + 1, 0);
                        for (size_t j = m0; j < m1; j++) ooff[j - m0 + 1] = ooff[j - m0] + ms[j].isz;
                        std::atomic<bool> bad{false};
                        ck_inflate.start();
                        pool.parallel_for((m1 - m0 + 1) / 2, [&](size_t t) {
                            static thread_local std::unique_ptr<FastInflate> fi[2];
                            if (!fi[0]) { fi[0].reset(new FastInflate()); fi[1].reset(new FastInflate()); }
                            const uint8_t *src[2] = {nullptr, nullptr};
                            size_t slen[2] = {0, 0}, jj[2] = {0, 0};
                            int n = 0;
                            for (size_t j = m0 + 2 * t; j < std::min(m1, m0 + 2 * t + 2); j++) {
                                if (ms[j].isz == 0) {  // (an empty member must be one: hts_lite.hpp)
                                    if (ms[j].size < ms[j].hl + 8 || !bgzf_empty_member_ok(cb + ms[j].off + ms[j].hl, ms[j].size - ms[j].hl - 8, cb + ms[j].off + ms[j].size - 8)) bad = true;
                                    continue;
                                }
                                src[n] = cb + ms[j].off + ms[j].hl;
                                slen[n] = ms[j].size - ms[j].hl - 8;
                                jj[n++] = j;
                            }
                            if (n == 2) {
                                if (!FastInflate::inflate2(*fi[0], src[0], slen[0], b.p + ooff[jj[0] - m0], ms[jj[0]].isz, *fi[1], src[1], slen[1], b.p + ooff[jj[1] - m0], ms[jj[1]].isz)) bad = true;
                            } else if (n == 1) {
                                if (!fi[0]->inflate(src[0], slen[0], b.p + ooff[jj[0] - m0], ms[jj[0]].isz)) bad = true;
                            }
                            for (int q = 0; q < n; q++) {  // the members' CRC32 (RFC 1952 trailer), as htslib checks it
                                uint32_t crc;
                                memcpy(&crc, cb + ms[jj[q]].off + ms[jj[q]].size - 8, 4);
                                if (crc32_fast(0, b.p + ooff[jj[q] - m0], ms[jj[q]].isz) != crc) bad = true;
                            }
                        }, CPU_INFLATE);
                        ck_inflate.stop();
                        if (bad) throw std::runtime_error("BGZF block does not inflate to its ISIZE / CRC32 (corrupt input)");
                        // a lane's last member: only the bytes in front of the next lane's first record are this lane's
                        if (limited && m1 == ms.size() && m1 > m0 && cb_off + ms[m1 - 1].off == lane.range.coff_end) {
                            if (lane.range.end_rec > ms[m1 - 1].isz) throw std::runtime_error("lane range ends beyond its last block");
                            total -= ms[m1 - 1].isz - (size_t)lane.range.end_rec;
                        }
                        b.n = total;
                        b.last = ended && m1 == ms.size();
                        b.cbuf = -1;
                        q_full.push(slot);
                        m0 = m1;
                    } while (m0 < ms.size());
                    if (--cbuf_refs[ck] == 0) q_cfree.push(ck);  // (inflated: the compressed bytes are done with, unless a call of the device's share still reads them)
                }
            } catch (const std::exception &e) {
                set_err(e.what());
                abort_all = true;
                q_cfree.close();
            }
            q_full.close();
        });
        ck_fasta.start();
        Fasta fa = load_fasta(o.pos[2]);  // anno.d:23
        ck_fasta.stop();
        Header out_hdr = hdr;
        out_hdr.add_pg("fade-annotate", "fade", FADE_VERSION, lane.on ? lane.cl : cl);  // anno.d:25-32
        std::vector<int64_t> lens(hdr.names.size());
        std::vector<const uint8_t *> ptrs(hdr.names.size());
        for (size_t k = 0; k < hdr.names.size(); k++) {
            size_t q = 0;
            while (q < fa.names.size() && fa.names[q] != hdr.names[k]) q++;
            if (q == fa.names.size()) { fprintf(stderr, "[E::fade annotate] reference %s of the BAM header is not in %s\n", hdr.names[k].c_str(), o.pos[2].c_str()); return 1; }
            if ((int64_t)fa.seqs[q].size() < hdr.lens[k]) {
                fprintf(stderr, "[E::fade annotate] %s is shorter in the FASTA (%zu) than in the header (%lld)\n", hdr.names[k].c_str(), fa.seqs[q].size(), (long long)hdr.lens[k]);
                return 1;
            }
            lens[k] = hdr.lens[k];
            ptrs[k] = (const uint8_t *)fa.seqs[q].data();
        }
        ck_upload.start();
        if (ctx_ready.get()) { fprintf(stderr, "[E::fade annotate] cannot open the GPU path: %s\n", create_err.c_str()); return 1; }
        // the buffers the reader has been filling become staging memory now
        for (auto &rb : raw_bufs) {
            if (fadehip_host_register(ctx, rb.first, rb.second)) { fprintf(stderr, "[E::fade annotate] %s\n", fadehip_last_error(ctx)); return 1; }
            guard.pinned.push_back(rb.first);
        }
        if (fadehip_genome_upload(ctx, (int)lens.size(), lens.data(), ptrs.data())) { fprintf(stderr, "[E::fade annotate] genome upload: %s\n", fadehip_last_error(ctx)); return 1; }
        // the header goes out through the CPU writer (its members only: no end-of-file block yet; not a byte without a device)
        {
            Writer hw(stdout, o.ubam ? OutFmt::UBAM : OutFmt::BAM, out_hdr, &pool, nullptr, !lane.on || lane.k == 0 || lane.shard, false);
            hw.close();
        }
        if (creating.get()) { fprintf(stderr, "[E::fade annotate] cannot open the GPU path: %s\n", create_err.c_str()); return 1; }
        ck_upload.stop();
        // (the FASTA's text is on the device now; giving a genome's worth of pages back takes milliseconds, and the first call
        // is waiting: on a thread of its own)
        std::thread([seqs = std::move(fa.seqs)]() mutable { seqs.clear(); seqs.shrink_to_fit(); }).detach();
        stages.th.emplace_back([&] {  // back: compress, hand to the writer
            int tok;
            try {
                while (q_done.pop(tok)) {
                    const uint8_t *p = nullptr;
                    size_t n = 0;
                    int credit;
                    if (!q_credit.pop(credit)) break;  // (the run is being given up)
                    ck_back.start();
                    const int rc = fadehip_bam_back(st, &p, &n);
                    ck_back.stop();
                    if (rc) throw std::runtime_error(std::string("device: ") + fadehip_last_error(ctx));
                    q_write.push(OutRef{p, n});
                }
            } catch (const std::exception &e) {
                set_err(e.what());
                abort_all = true;
                while (q_done.pop(tok)) {}
            }
            q_write.close();
        });
        // The writer.  Into a file that can be written at an offset (a regular file, not opened for appending) a call's members
        // go out as a few pwrites side by side — one thread copies into the page cache at 8-10 GB/s, and 1.6 GB per 10 M reads
        // is then a stage as long as the device's; into a pipe they go out through stdout in order.  `placed_base` is where the
        // records start (the header has been flushed by then); the file offset is set behind the last byte at the end.
        std::atomic<long long> placed_base{-1};
        std::atomic<uint64_t> placed_bytes{0};
        // (FADE_BAM_WRITERS=n: off by default — measured on the GPU box, four pwrites side by side into one file took 0.21-0.23 s
        // per 1.6 GB against 0.18 s front to back: buffered writes to one inode serialise in the kernel)
        const int n_wr = std::max(1, std::min(8, getenv("FADE_BAM_WRITERS") ? atoi(getenv("FADE_BAM_WRITERS")) : 1));
        stages.th.emplace_back([&] {  // writer
            OutRef r;
            bool ok = true;
            uint64_t at = 0;
            std::vector<std::thread> helpers;
            while (q_write.pop(r)) {
                if (ok && r.n) {
                    ck_fwrite.start();
                    const long long base = placed_base.load();
                    if (base < 0) {
                        if (fwrite(r.p, 1, r.n, stdout) != r.n) { set_err("write error on the output stream"); ok = false; abort_all = true; }
                    } else {
                        std::atomic<bool> bad{false};
                        auto put = [&](size_t lo, size_t hi) {
                            while (lo < hi) {
                                const ssize_t w = pwrite(1, r.p + lo, hi - lo, (off_t)((uint64_t)base + at + lo));
                                if (w < 0 && errno == EINTR) continue;
                                if (w <= 0) { bad = true; return; }
                                lo += (size_t)w;
                            }
                        };
                        const size_t parts = r.n >= ((size_t)1 << 20) ? (size_t)n_wr : 1;
                        helpers.clear();
                        for (size_t q = 1; q < parts; q++) helpers.emplace_back(put, r.n * q / parts, r.n * (q + 1) / parts);
                        put(0, r.n / parts);
                        for (auto &h : helpers) h.join();
                        if (bad) { set_err("write error on the output stream"); ok = false; abort_all = true; }
                        at += r.n;
                        placed_bytes = at;
                    }
                    ck_fwrite.stop();
                }
                q_credit.push(0);
            }
        });
        {
            struct stat so;
            const int fl = fcntl(1, F_GETFL);
            if (fflush(stdout) == 0 && fstat(1, &so) == 0 && S_ISREG(so.st_mode) && fl >= 0 && !(fl & O_APPEND) &&
                getenv("FADE_BAM_WRITERS") && atoi(getenv("FADE_BAM_WRITERS")) > 1) {
                const off_t b = lseek(1, 0, SEEK_CUR);
                if (b >= 0) placed_base = (long long)b;
            }
        }
        int k;
        bool failed = false;
        while (!failed && !abort_all && q_full.pop(k)) {
            ck_front.start();
            const int rc = (host_inflate && !bufs[k].compressed) ? fadehip_bam_front_raw(st, bufs[k].p, bufs[k].n, bufs[k].last ? 1 : 0)
                                                                 : fadehip_bam_front(st, bufs[k].p, bufs[k].n, bufs[k].last ? 1 : 0);
            ck_front.stop();
            if (rc) { set_err(std::string("device: ") + fadehip_last_error(ctx)); failed = true; break; }
            if (bufs[k].cbuf >= 0) {  // (the compressed bytes have been copied up)
                if (!host_inflate) q_cfree.push(bufs[k].cbuf);
                else if (--cbuf_refs[bufs[k].cbuf] == 0) q_cfree.push(bufs[k].cbuf);
            }
            q_free.push(k);
            q_done.push(0);
        }
        q_done.close();
        // a stage gave up (a write error, a back call that failed) or front did: the stages in front of this loop may sit in
        // q_free / q_cfree, which nobody serves any more — closed here, BEFORE the join, so that they see the end
        if (failed || abort_all) { abort_all = true; q_free.close(); q_cfree.close(); while (q_full.pop(k)) {} }
        for (auto &t : stages.th) t.join();
        stages.unblock = nullptr;
        if (!stage_err.empty()) { fprintf(stderr, "[E::fade annotate] %s\n", stage_err.c_str()); return 1; }
        if (placed_base.load() >= 0 && lseek(1, (off_t)((uint64_t)placed_base.load() + placed_bytes.load()), SEEK_SET) < 0) {
            fprintf(stderr, "[E::fade annotate] cannot seek the output: %s\n", strerror(errno));
            return 1;
        }
        if ((!lane.on || lane.shard) && fwrite(BGZF_EOF, 1, sizeof BGZF_EOF, stdout) != sizeof BGZF_EOF) { fprintf(stderr, "[E::fade annotate] write error on the output stream\n"); return 1; }
        if (fflush(stdout) != 0 || ferror(stdout)) { fprintf(stderr, "[E::fade annotate] write error on the output stream\n"); return 1; }
        int64_t totals[8], n_rec = 0, n_over = 0;
        fadehip_bam_totals(st, totals, &n_rec, &n_over);
        if (lane.on) {
            // the lane's report for the parent; lanes on distinct devices first sum their counters among themselves (the one
            // collective of the path: ncclAllReduce over xGMI, one rank per process)
            long long red[8];
            bool have_red = false;
            if (lane.rccl && lane.n > 1) {
                int64_t t[8];
                std::copy(totals, totals + 8, t);
                if (fadehip_stats_allreduce_rank(ctx, lane.k, lane.n, lane.id_path.c_str(), t, 8)) { fprintf(stderr, "[E::fade annotate] stats all-reduce (lanes): %s\n", fadehip_last_error(ctx)); return 1; }
                for (int q = 0; q < 8; q++) red[q] = (long long)t[q];
                have_red = true;
            }
            FILE *sf = fopen(lane.status_path.c_str(), "w");
            if (!sf) { fprintf(stderr, "[E::fade annotate] cannot write %s\n", lane.status_path.c_str()); return 1; }
            for (int q = 0; q < 8; q++) fprintf(sf, "%lld ", (long long)totals[q]);
            fprintf(sf, "%lld\n", (long long)n_over);
            if (have_red) {
                for (int q = 0; q < 8; q++) fprintf(sf, "%lld ", red[q]);
                fprintf(sf, "\n");
            }
            fclose(sf);
        }
        if (n_over && !lane.on) fprintf(stderr, "[W::fade annotate] %lld soft-clipped reads were not re-aligned: read longer than %d bases or window longer than %d\n",
                            (long long)n_over, FADEHIP_MAX_LONG_QUERY, prm.max_ref_len);
        if (o.stats && !lane.on) {  // stats.d:56-72 layout
            const double rc = (double)std::max<int64_t>(totals[0], 1);
            fprintf(stderr, "read count:\t%lld\nClipped %%:\t%g\n%% With Supplementary alns:\t%g\nArtifact rate:\t%g\n"
                            "%% With Supplementary alns and artifacts:\t%g\nArtifact rate left only:\t%g\nArtifact rate right only:\t%g\n",
                    (long long)totals[0], totals[1] / rc, totals[2] / rc, totals[4] / rc, totals[3] / rc, totals[6] / rc, totals[7] / rc);
        }
        ck_total.stop();
        if (o.timing) {
            fprintf(stderr, "[timing] total %.3f s: fasta %.3f, create+genome upload %.3f | reader: file reads %.3f, inflate on %d host threads %.3f | front (%sframe, annotate, tags) %.3f | "
                            "back (deflate, copy out) %.3f | fwrite %.3f (stages overlap); %lld records\n",
                    ck_total.t, ck_fasta.t, ck_upload.t, ck_fread.t, host_inflate ? pool.size() : 0, ck_inflate.t, host_inflate ? "" : "inflate, ", ck_front.t, ck_back.t, ck_fwrite.t, (long long)n_rec);
        }
        g_trace.dump();
        if (getenv("FADEHIP_BAM_PROF")) { fadehip_bam_close(st); st = nullptr; }  // (prints the library's own clocks)
        if (!(getenv("FADE_FAST_EXIT") && atoi(getenv("FADE_FAST_EXIT")) == 0)) {
            if (o.timing) fprintf(stderr, "[timing] since process start %.3f s (leaving by _exit)\n", since_process_start());
            fflush(stdout);
            fflush(stderr);
            _exit(0);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "[E::fade annotate] %s\n", e.what());
        return 1;
    }
    return 0;
}

static int annotate_main(const std::string &cl, const Opts &o) {
    StageClock ck_total, ck_fasta, ck_upload, ck_read, ck_pack, ck_submit, ck_collect, ck_tags, ck_write;
    ck_total.start();
    if (o.timing) fprintf(stderr, "[timing] since process start %.3f s (annotate begins)\n", since_process_start());
    if (o.timing) fprintf(stderr, "[timing] %d threads%s\n", o.threads > 0 ? o.threads : default_threads(), o.threads > 0 ? "" : " (default: affinity and cgroup quota)");
    const LaneEnv lane = lane_env();  // set when this process is one lane of a `--gpus N` run (annotate_lanes_main)
    // anno.d:18-19 (htslib log format)
    if (!lane.on) fprintf(stderr, "[W::fade annotate] Output SAM/BAM will not be sorted (regardless of prior sorting)\n");
    const int nthreads = o.threads > 0 ? o.threads : default_threads();
    const int ngpu = lane.on ? 1 : std::max(1, o.gpus);
    std::vector<fadehip_ctx *> ctxs((size_t)ngpu, nullptr);
    std::vector<BlockPool> blocks((size_t)ngpu);
    struct CtxGuard {
        std::vector<fadehip_ctx *> &c;
        std::vector<BlockPool> &b;
        ~CtxGuard() {
            for (size_t k = 0; k < c.size(); k++) {
                if (c[k]) b[k].drain();
                fadehip_destroy(c[k]);
            }
        }
    } ctx_guard{ctxs, blocks};
    Pool pool(nthreads);
    try {
        Reader reader(o.pos[1], &pool);   // anno.d:22 (every stage's parallel work runs on the one pool)
        if (lane.on) reader.restrict_to(lane.range);  // this lane's records only
        // (the reader starts at once: the first batches inflate while the FASTA loads and the genome goes to HBM)
        // Stages: [reader: BGZF inflate / SAM parse] -> [this thread: pack into a pinned block, upload + run (both return
        // at once), fetch the oldest batch's results] -> [writer: tags, format, BGZF deflate].  Each stage has its own
        // pool; chunks keep their input order.  The device calls are asynchronous, so this one thread keeps every slot
        // of every device busy: batch k goes to device k % N, the device's slots in turn.
        // the reader may run ahead by about two million records (0.6 GB inflated) while the GPU path comes up
        const size_t ahead = (size_t)std::max(2, std::min(8, (2 << 20) / std::max(o.batch, 1)));
        BoundedQueue<std::unique_ptr<Chunk>> q_in(ahead), q_out(3);
        std::string stage_err;
        std::mutex err_m;
        auto set_stage_err = [&](const std::string &e) {
            std::lock_guard<std::mutex> l(err_m);
            if (stage_err.empty()) stage_err = e;
        };
        std::atomic<bool> abort_stages{false};
        StageThreads stages;
        stages.unblock = [&] {
            abort_stages = true;
            q_in.close();
            q_out.close();
        };
        stages.th.emplace_back([&] {
            try {
                while (!abort_stages) {
                    std::unique_ptr<Chunk> c(new Chunk());
                    ck_read.start();
                    const size_t got = reader.read_block(c->blk, (size_t)std::max(o.batch, 1));  // (SAM lines are parsed into the same layout)
                    ck_read.stop();
                    if (got == 0) break;
                    q_in.push(std::move(c));
                }
            } catch (const std::exception &e) {
                set_stage_err(e.what());
            }
            q_in.close();
        });
        // --gpus N uses devices 0..N-1; FADE_DEVICE_MAP="0,0" (tests on a one-GPU box) maps the N contexts elsewhere
        std::vector<int> devmap((size_t)ngpu);
        for (int d = 0; d < ngpu; d++) devmap[(size_t)d] = d;
        bool distinct_devices = true;
        if (lane.on) devmap[0] = lane.device;
        else if (const char *dm = getenv("FADE_DEVICE_MAP")) {
            int d = 0;
            for (const char *q = dm; *q && d < ngpu; d++) {
                devmap[(size_t)d] = atoi(q);
                q = strchr(q, ',');
                if (!q) break;
                q++;
            }
            for (int a = 0; a < ngpu; a++)
                for (int b2 = a + 1; b2 < ngpu; b2++)
                    if (devmap[(size_t)a] == devmap[(size_t)b2]) distinct_devices = false;
        }
        fadehip_params prm;
        fadehip_params_default(&prm);  // max_ref_len 2^20: any window the kernels can serve, whatever -w is
        prm.max_batch_reads = std::max(o.batch, 1);
        // Batches in flight per device.  A device annotates a batch in about a millisecond and the host needs thirty for
        // it, so one slot is enough here, and with one batch at a time nothing competes with the score pass for CUs: the
        // run then needs three HSA queues fewer (a slot's two streams and the CU-masked one; each is a 173 MB context-save
        // area to set up and to give back: ~80 ms per run in all).  FADE_SLOTS=2 restores the double-buffered form.
        // If the host ever waits for the device (more than 15 % of the time since the first batch went out), the second slot
        // is taken into use for the rest of the run: upload and run of a batch then overlap the one before.
        const bool slots_fixed = getenv("FADE_SLOTS") != nullptr;
        const double wait_frac = getenv("FADE_SLOT_WAIT") ? atof(getenv("FADE_SLOT_WAIT")) : 0.15;  // (tests: 0 forces the switch)
        if (const char *sl = getenv("FADE_SLOTS")) kSlotsInUse = (size_t)std::max(1, std::min(atoi(sl), FADEHIP_NUM_SLOTS));
        if (kSlotsInUse == 1) setenv("FADEHIP_TAIL_CUS", "0", 0);
        // the HIP runtime and the contexts come up on a helper thread while this one reads the FASTA
        std::string create_err;
        std::future<int> creating = std::async(std::launch::async, [&]() -> int {
            for (int d = 0; d < ngpu; d++)
                if (fadehip_create(&ctxs[(size_t)d], devmap[(size_t)d], &prm)) {
                    create_err = fadehip_last_error(nullptr);  // (the library keeps it per thread)
                    return 1;
                }
            return 0;
        });
        ck_fasta.start();
        Fasta fa = load_fasta(o.pos[2]);  // anno.d:23
        ck_fasta.stop();
        Header hdr = reader.header();     // anno.d:24
        hdr.add_pg("fade-annotate", "fade", FADE_VERSION, lane.on ? lane.cl : cl);  // anno.d:25-32

        // contigs of the BAM header, in tid order, must be present in the FASTA (fetchSequence by name, analysis.d:63)
        const Header &h = reader.header();
        std::vector<int64_t> lens(h.names.size());
        std::vector<const uint8_t *> ptrs(h.names.size());
        for (size_t k = 0; k < h.names.size(); k++) {
            size_t f = 0;
            while (f < fa.names.size() && fa.names[f] != h.names[k]) f++;
            if (f == fa.names.size()) {
                fprintf(stderr, "[E::fade annotate] reference %s of the BAM header is not in %s\n", h.names[k].c_str(), o.pos[2].c_str());
                return 1;
            }
            // analysis.d:55-59 clamps the window at the header's targetLength; the FASTA may be longer, never shorter
            if ((int64_t)fa.seqs[f].size() < h.lens[k]) {
                fprintf(stderr, "[E::fade annotate] %s is shorter in the FASTA (%zu) than in the header (%lld)\n", h.names[k].c_str(), fa.seqs[f].size(), (long long)h.lens[k]);
                return 1;
            }
            lens[k] = h.lens[k];
            ptrs[k] = (const uint8_t *)fa.seqs[f].data();
        }
        if (h.names.empty()) {
            fprintf(stderr, "[E::fade annotate] input has no @SQ lines\n");
            return 1;
        }
        auto die = [&](fadehip_ctx *c, const char *what) {
            fprintf(stderr, "[E::fade annotate] %s: %s\n", what, fadehip_last_error(c));
            return 1;
        };
        ck_upload.start();
        if (creating.get()) {
            fprintf(stderr, "[E::fade annotate] cannot open the GPU path: %s\n", create_err.c_str());
            return 1;
        }
        for (int d = 0; d < ngpu; d++) {
            blocks[(size_t)d].ctx = ctxs[(size_t)d];
            if (fadehip_genome_upload(ctxs[(size_t)d], (int)lens.size(), lens.data(), ptrs.data())) return die(ctxs[(size_t)d], "genome upload");
        }
        ck_upload.stop();
        fa.seqs.clear();
        fa.seqs.shrink_to_fit();
        // nothing is written to stdout before the GPU path is known to be usable
        const OutFmt fmt = o.bam ? OutFmt::BAM : o.ubam ? OutFmt::UBAM : OutFmt::SAM;  // util.d:65-76
        // BAM output: the BGZF blocks are compressed on the device (FADE_BGZF_DEVICE=0 keeps them on the host pool)
        std::unique_ptr<DeviceBgzf> dev_codec;
        if (fmt == OutFmt::BAM && !(getenv("FADE_BGZF_DEVICE") && atoi(getenv("FADE_BGZF_DEVICE")) == 0)) dev_codec.reset(new DeviceBgzf(ctxs[0]));
        Writer writer(stdout, fmt, hdr, &pool, dev_codec.get(), !lane.on || lane.k == 0 || lane.shard, !lane.on || lane.shard);

        StageThreads wstage;  // declared after the writer it uses: joined before the writer goes away
        wstage.unblock = [&] {
            abort_stages = true;
            q_in.close();
            q_out.close();
        };
        wstage.th.emplace_back([&] {
            std::unique_ptr<Chunk> c;
            try {
                while (q_out.pop(c)) {
                    ck_tags.start();
                    apply_tags(*c, hdr, pool);
                    ck_tags.stop();
                    ck_write.start();
                    writer.write_block(c->blk, c->bout);
                    ck_write.stop();
                }
            } catch (const std::exception &e) {
                set_stage_err(e.what());
                while (q_out.pop(c)) {}
            }
        });
        std::deque<std::unique_ptr<Chunk>> inflight;
        int64_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<std::vector<int64_t>> per_dev((size_t)ngpu, std::vector<int64_t>(8, 0));
        int64_t n_oversize = 0;
        bool failed = false;
        auto finish = [&](std::unique_ptr<Chunk> c) -> int {
            fadehip_anno_view v;
            ck_collect.start();
            const int crc = fadehip_annotate_results(ctxs[(size_t)c->dev], c->slot, &v);
            if (!crc) {
                // the slot's result block is reused by its next batch: keep what the writer stage needs
                c->rs_sent.assign(v.rs, v.rs + v.n_reads);
                c->art.clear();
                for (int k = 0; k < v.n_aln; k++)
                    if (v.aln[k].art) c->art.push_back(v.aln[k]);
                for (int k = 0; k < 8; k++) per_dev[(size_t)c->dev][(size_t)k] += v.stats[k];
                n_oversize += v.n_oversize;
            }
            ck_collect.stop();
            blocks[(size_t)c->dev].release(c->block, c->block_cap);
            c->block = nullptr;
            if (crc) {
                fprintf(stderr, "[E::fade annotate] results: %s\n", fadehip_last_error(ctxs[(size_t)c->dev]));
                return 1;
            }
            q_out.push(std::move(c));
            return 0;
        };
        size_t seq_no = 0;
        std::vector<int> next_slot((size_t)ngpu, 0);
        std::chrono::steady_clock::time_point t_first;
        std::unique_ptr<Chunk> c;
        // a failed stage (reader or writer) ends the run here: no further batch is packed or sent to a device
        auto stage_failed = [&] {
            std::lock_guard<std::mutex> l(err_m);
            return !stage_err.empty();
        };
        while (!failed && !stage_failed() && q_in.pop(c)) {
            c->dev = (int)(seq_no % (size_t)ngpu);
            c->slot = next_slot[(size_t)c->dev];
            next_slot[(size_t)c->dev] = (c->slot + 1) % (int)kSlotsInUse;
            if (seq_no == 0) t_first = std::chrono::steady_clock::now();
            seq_no++;
            ck_pack.start();
            pack_chunk(*c, pool, blocks[(size_t)c->dev]);  // (while the device works on the batch before)
            ck_pack.stop();
            // one batch per (device, slot): the batch that had this slot is fetched before the slot is handed the next
            while (!failed && inflight.size() >= (size_t)ngpu * kSlotsInUse) {
                if (finish(std::move(inflight.front()))) failed = true;
                inflight.pop_front();
            }
            if (failed) {
                blocks[(size_t)c->dev].release(c->block, c->block_cap);  // (the chunk in hand goes nowhere: its pinned block returns to the pool)
                c->block = nullptr;
                break;
            }
            ck_submit.start();
            const int src = fadehip_annotate_submit(ctxs[(size_t)c->dev], c->slot, &c->b, o.floor_len, o.window);
            ck_submit.stop();
            if (src) {
                fprintf(stderr, "[E::fade annotate] submit: %s\n", fadehip_last_error(ctxs[(size_t)c->dev]));
                blocks[(size_t)c->dev].release(c->block, c->block_cap);
                c->block = nullptr;
                failed = true;
                break;
            }
            inflight.push_back(std::move(c));
            if (!slots_fixed && kSlotsInUse == 1 && seq_no >= 4 * (size_t)ngpu &&
                ck_collect.t > wait_frac * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_first).count()) {
                kSlotsInUse = 2;  // the batches in flight keep slot 0; the next one of every device takes slot 1
                std::fill(next_slot.begin(), next_slot.end(), 1);
                if (o.timing) fprintf(stderr, "[timing] the host waited for the device: two batches in flight per device from batch %zu on\n", seq_no);
            }
        }
        while (!failed && !inflight.empty()) {
            if (finish(std::move(inflight.front()))) failed = true;
            inflight.pop_front();
        }
        if (stage_failed()) failed = true;
        if (failed) {  // the batches still in flight are abandoned (their pinned blocks go back), the reader runs out so that it can exit
            for (auto &f : inflight)
                if (f && f->block) {
                    (void)fadehip_sync(ctxs[(size_t)f->dev]);  // the DMA out of the block has ended before it is reused or freed
                    blocks[(size_t)f->dev].release(f->block, f->block_cap);
                    f->block = nullptr;
                }
            inflight.clear();
            abort_stages = true;
            while (q_in.pop(c)) {}
        }
        q_out.close();
        for (auto &t : wstage.th) t.join();
        for (auto &t : stages.th) t.join();
        wstage.unblock = nullptr;
        stages.unblock = nullptr;
        if (!stage_err.empty()) {
            fprintf(stderr, "[E::fade annotate] %s\n", stage_err.c_str());
            failed = true;
        }
        if (failed) return 1;
        writer.close();
        if (n_oversize && !lane.on)
            fprintf(stderr, "[W::fade annotate] %lld soft-clipped reads were not re-aligned: read longer than %d bases or window longer than %d\n",
                    (long long)n_oversize, FADEHIP_MAX_LONG_QUERY, prm.max_ref_len);
        // the one collective of the path: sum the stats.d counters over the devices (RCCL over xGMI)
        if (ngpu > 1 && distinct_devices) {
            std::vector<int64_t> flat((size_t)ngpu * 8);
            for (int d = 0; d < ngpu; d++) std::copy(per_dev[(size_t)d].begin(), per_dev[(size_t)d].end(), flat.begin() + d * 8);
            if (fadehip_stats_allreduce(ctxs.data(), ngpu, flat.data(), 8)) return die(ctxs[0], "stats all-reduce");
            std::copy(flat.begin(), flat.begin() + 8, totals);
        } else {
            // (several contexts mapped onto one device by FADE_DEVICE_MAP: RCCL takes one rank per device, sum here)
            for (int d = 0; d < ngpu; d++)
                for (int k = 0; k < 8; k++) totals[k] += per_dev[(size_t)d][(size_t)k];
        }
        if (lane.on) {
            // the lane's report for the parent; lanes on distinct devices first sum their counters among themselves, the one
            // collective of the path: ncclAllReduce over xGMI, one rank per process
            long long red[8];
            bool have_red = false;
            if (lane.rccl && lane.n > 1) {
                int64_t t[8];
                std::copy(totals, totals + 8, t);
                if (fadehip_stats_allreduce_rank(ctxs[0], lane.k, lane.n, lane.id_path.c_str(), t, 8)) return die(ctxs[0], "stats all-reduce (lanes)");
                for (int q = 0; q < 8; q++) red[q] = (long long)t[q];
                have_red = true;
            }
            FILE *sf = fopen(lane.status_path.c_str(), "w");
            if (!sf) { fprintf(stderr, "[E::fade annotate] cannot write %s\n", lane.status_path.c_str()); return 1; }
            for (int q = 0; q < 8; q++) fprintf(sf, "%lld ", (long long)totals[q]);
            fprintf(sf, "%lld\n", (long long)n_oversize);
            if (have_red) {
                for (int q = 0; q < 8; q++) fprintf(sf, "%lld ", red[q]);
                fprintf(sf, "\n");
            }
            fclose(sf);
        }
        if (o.stats && !lane.on) {  // stats.d:56-72 layout
            const double rc = (double)std::max<int64_t>(totals[0], 1);
            fprintf(stderr, "read count:\t%lld\nClipped %%:\t%g\n%% With Supplementary alns:\t%g\nArtifact rate:\t%g\n"
                            "%% With Supplementary alns and artifacts:\t%g\nArtifact rate left only:\t%g\nArtifact rate right only:\t%g\n",
                    (long long)totals[0], totals[1] / rc, totals[2] / rc, totals[4] / rc, totals[3] / rc, totals[6] / rc, totals[7] / rc);
        }
        ck_total.stop();
        if (o.timing)
            fprintf(stderr, "[timing] total %.3f s: fasta %.3f, create+genome upload %.3f | reader stage %.3f | pack %.3f, "
                            "submit %.3f, results %.3f | writer stage: tags %.3f, write %.3f (stages overlap)\n",
                    ck_total.t, ck_fasta.t, ck_upload.t, ck_read.t, ck_pack.t, ck_submit.t, ck_collect.t, ck_tags.t, ck_write.t);
        if (o.timing) {  // core-seconds of the pools' work by kind (thread CPU time)
            std::string line = "[timing] pool core-seconds:";
            double sum = 0;
            for (int k = 0; k < CPU_KINDS; k++) {
                const double sec = (double)cpu_meter()[k].load() * 1e-9;
                if (sec < 0.0005) continue;
                char buf[64];
                snprintf(buf, sizeof buf, " %s %.3f,", cpu_kind_name(k), sec);
                line += buf;
                sum += sec;
            }
            fprintf(stderr, "%s total %.3f\n", line.c_str(), sum);
            fprintf(stderr, "[timing] output thread: %.3f s inside fwrite\n", writer.io_seconds());
            const ReadProf &rp = read_prof();
            if (reader.is_bam())
                fprintf(stderr, "[timing] BAM reader thread: wait for file bytes %.3f, scan %.3f, inflate (parallel) %.3f, frame %.3f, carry %.3f, "
                                "layout check (parallel) %.3f\n", rp.wait_io, rp.scan, rp.inflate, rp.frame, rp.carry, rp.layout);
        }
        // Everything is written and flushed.  What is left would be giving back, piece by piece, what the process exit gives
        // back at once — the contexts' device buffers and streams, a gigabyte of pinned blocks, the pools' threads: 0.1 s of
        // a 0.8 s run.  FADE_FAST_EXIT=0 takes the long way (leak checkers, tests of the destructors).
        if (!(getenv("FADE_FAST_EXIT") && atoi(getenv("FADE_FAST_EXIT")) == 0)) {
            if (o.timing) fprintf(stderr, "[timing] since process start %.3f s (leaving by _exit)\n", since_process_start());
            fflush(stdout);
            fflush(stderr);
            _exit(0);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "[E::fade annotate] %s\n", e.what());
        return 1;
    }
    return 0;
}

// std.getopt rejects an option the subcommand did not declare (app.d:76-83,104-107,132-134)
static bool options_allowed(const Opts &o, const char *allowed) {
    for (char c : o.seen)
        if (!strchr(allowed, c)) {
            fprintf(stderr, "std.getopt.GetOptException: Unrecognized option %s%c\n", "-", c);
            return false;
        }
    return true;
}

static void print_extract_help() {
    fprintf(stderr,
            "%s\nextract: extracts artifacts into a mapped SAM/BAM (used after annotate)\n"
            "usage: fade extract [options] <annotated BAM/SAM> \n\n"
            "-t --threads extra threads for parsing the bam file\n"
            "-b     --bam output bam\n"
            "-u    --ubam output uncompressed bam\n"
            "-h    --help This help information.\n\n",
            kHeader);
}

static bool parse_cigar_string(const std::string &s, std::vector<uint32_t> &ops) {
    uint64_t num = 0;
    bool have = false;
    for (char c : s) {
        if (c >= '0' && c <= '9') { num = num * 10 + (uint64_t)(c - '0'); have = true; }
        else {
            const char *o = strchr(CIGAR_STR, c);
            if (!o || !have) return false;
            ops.push_back((uint32_t)(num << 4) | (uint32_t)(o - CIGAR_STR));
            num = 0;
            have = false;
        }
    }
    return !have;
}

// source/remap.d:11-87 — `fade extract`: one new mapped record per artifact side, built from the am tag
// (contig, 0-based pos, CIGAR), carrying the reverse-complemented read and its reversed qualities.  Pure host
// work: it consumes what the annotate path wrote and so validates the am grammar end to end.
static int extract_main(const std::string &cl, const Opts &o) {
    fprintf(stderr, "[W::fade extract] Output SAM/BAM will not be sorted\n");  // remap.d:13
    const int nthreads = o.threads > 0 ? o.threads : 2;
    Pool pool(nthreads);  // (reader and writer share it)
    try {
        Reader reader(o.pos[1], &pool);  // remap.d:17
        Header hdr = reader.header();
        hdr.add_pg("fade-extract", "fade", FADE_VERSION, cl);  // remap.d:18-26
        const OutFmt fmt = o.bam ? OutFmt::BAM : o.ubam ? OutFmt::UBAM : OutFmt::SAM;
        Writer writer(stdout, fmt, hdr, &pool);
        static const uint8_t comp[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
        std::vector<Rec> in, out;
        for (;;) {
            in.clear();
            out.clear();
            if (reader.read_chunk(in, 65536) == 0) break;
            for (const Rec &r : in) {
                const size_t prs = r.aux_find("rs");  // remap.d:31-33
                if (prs == std::string::npos) continue;
                uint32_t rsv = 0;
                const uint8_t ty = r.d[prs + 2];
                if (ty == 'C' || ty == 'c') rsv = r.d[prs + 3];
                else if (ty == 'S' || ty == 's') rsv = r.rd<uint16_t>(prs + 3);
                else if (ty == 'I' || ty == 'i') rsv = r.rd<uint32_t>(prs + 3);
                else continue;
                rsv &= 0xff;
                if (!(rsv & 6)) continue;  // remap.d:36-37
                const size_t pam = r.aux_find("am");  // remap.d:38-40
                if (pam == std::string::npos || r.d[pam + 2] != 'Z') continue;
                const std::string am((const char *)r.d.data() + pam + 3);
                const size_t semi = am.find(';');
                const std::string sides[2] = {am.substr(0, semi), semi == std::string::npos ? std::string() : am.substr(semi + 1)};
                for (int side = 0; side < 2; side++) {
                    if (!(rsv & (side == 0 ? 2u : 4u))) continue;  // remap.d:43,64
                    const std::string &f = sides[side];
                    const size_t c1 = f.find(','), c2 = c1 == std::string::npos ? c1 : f.find(',', c1 + 1);
                    if (c2 == std::string::npos) throw std::runtime_error("malformed am tag: " + am);
                    const int tid = reader.header().tid_of(f.substr(0, c1));
                    const int64_t pos = std::strtoll(f.c_str() + c1 + 1, nullptr, 10);
                    std::vector<uint32_t> cig;
                    if (!parse_cigar_string(f.substr(c2 + 1), cig)) throw std::runtime_error("malformed am CIGAR: " + am);
                    const int lq = r.l_seq();
                    const size_t lqn = (size_t)r.l_qname();
                    Rec n;
                    n.d.assign(32 + lqn + 4 * cig.size() + ((size_t)lq + 1) / 2 + (size_t)lq, 0);
                    int64_t reflen = 0;
                    for (uint32_t c : cig) {
                        const uint32_t op = c & 15;
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += c >> 4;
                    }
                    n.wr<int32_t>(0, tid);                       // remap.d:49
                    n.wr<int32_t>(4, (int32_t)pos);              // remap.d:50 (ZB: am holds a 0-based position)
                    n.d[8] = (uint8_t)lqn;
                    n.d[9] = 0;                                  // a fresh bam1_t: mapq 0
                    n.wr<uint16_t>(10, (uint16_t)reg2bin(pos < 0 ? 0 : pos, (pos < 0 ? 0 : pos) + (reflen > 0 ? reflen : 1)));
                    n.wr<uint16_t>(12, (uint16_t)cig.size());
                    n.wr<uint16_t>(14, (uint16_t)((r.flag() & 0x10) ? 0 : 0x10));  // remap.d:51-58
                    n.wr<int32_t>(16, lq);
                    n.wr<int32_t>(20, 0);                        // bam_init1 zero-fills: mtid 0, mpos 0, isize 0
                    n.wr<int32_t>(24, 0);
                    n.wr<int32_t>(28, 0);
                    memcpy(n.d.data() + 32, r.qname(), lqn);     // remap.d:48
                    size_t off = 32 + lqn;
                    if (!cig.empty()) memcpy(n.d.data() + off, cig.data(), 4 * cig.size());  // remap.d:61
                    off += 4 * cig.size();
                    const uint8_t *sq = r.seq(), *ql = r.qual();
                    for (int j = 0; j < lq; j++) {               // remap.d:59 reverse_complement_sam_record
                        const int code = (sq[j >> 1] >> ((~j & 1) << 2)) & 15;
                        const int k = lq - 1 - j;
                        n.d[off + (size_t)(k >> 1)] |= (uint8_t)(comp[code] << ((~k & 1) << 2));
                    }
                    off += ((size_t)lq + 1) / 2;
                    for (int j = 0; j < lq; j++) n.d[off + (size_t)j] = ql[lq - 1 - j];  // remap.d:60
                    out.push_back(std::move(n));
                }
            }
            writer.write(out);
        }
        writer.close();
    } catch (const std::exception &e) {
        fprintf(stderr, "[E::fade extract] %s\n", e.what());
        return 1;
    }
    return 0;
}

// ------------------------------------------------------------------ fade out (source/filter.d)
static void print_out_help() {
    fprintf(stderr,
            "%s\nout: removes all read and mates for reads contain the artifact (used after annotate)\n"
            "     or, with the -c flag, hard clips out artifact sequence from reads\n"
            "     it is reccomended that the input SAM/BAM be queryname sorted\n"
            "usage: fade out [options] <input BAM/SAM>\n\n"
            "-c    --clip clip reads instead of filtering them\n"
            "-t --threads extra threads for parsing the bam file\n"
            "-b     --bam output bam\n"
            "-u    --ubam output uncompressed bam\n"
            "-h    --help This help information.\n\n",
            kHeader);
}

// std.conv.parse!long on the front of s: optional sign, digits; consumes what it parsed.  false = ConvException.
static bool d_parse_long(std::string &s, long long &v) {
    size_t k = 0;
    bool neg = false;
    if (k < s.size() && (s[k] == '-' || s[k] == '+')) { neg = s[k] == '-'; k++; }
    if (k >= s.size() || s[k] < '0' || s[k] > '9') return false;
    long long x = 0;
    while (k < s.size() && s[k] >= '0' && s[k] <= '9') { x = x * 10 + (s[k] - '0'); k++; }
    v = neg ? -x : x;
    s.erase(0, k);
    return true;
}

// filter.d:127-167
static int numerically_aware_cmp(std::string a, std::string b) {
    while (!a.empty() && !b.empty()) {
        const bool nda = a[0] > '9' || a[0] < '0', ndb = b[0] > '9' || b[0] < '0';
        if (nda && ndb) {
            if (a[0] == b[0]) { a.erase(0, 1); b.erase(0, 1); continue; }
            return (unsigned char)a[0] < (unsigned char)b[0] ? -1 : 1;
        }
        long long ai = -1, bi = -1;
        const bool pa = d_parse_long(a, ai), pb = d_parse_long(b, bi);
        if (!pa && !pb) return a.size() == b.size() ? 0 : a.size() < b.size() ? -1 : 1;  // cannot happen: one side is a digit
        if (ai == bi) continue;
        return ai < bi ? -1 : 1;
    }
    return a.size() == b.size() ? 0 : a.size() < b.size() ? -1 : 1;
}

static bool rs_of(const Rec &r, uint32_t &rsv) {
    const size_t p = r.aux_find("rs");
    if (p == std::string::npos) return false;
    const uint8_t ty = r.d[p + 2];
    if (ty == 'C' || ty == 'c') rsv = r.d[p + 3];
    else if (ty == 'S' || ty == 's') rsv = r.rd<uint16_t>(p + 3);
    else if (ty == 'I' || ty == 'i') rsv = r.rd<uint32_t>(p + 3);
    else return false;
    rsv &= 0xff;
    return true;
}

struct OutStats {  // stats.d:16-72
    long long read_count = 0, clipped = 0, sup = 0, art_sup = 0, art = 0, art_mate = 0, aln_l = 0, aln_r = 0;
    void parse(uint32_t v) {
        const uint32_t sc = v & 1, al = (v >> 1) & 1, ar = (v >> 2) & 1, ml = (v >> 3) & 1, mr = (v >> 4) & 1, su = (v >> 5) & 1;
        clipped += sc;
        art += (al | ar);
        sup += su;
        art_sup += (al | ar) & su;
        art_mate += ((al & ml) | (ar & mr));
        aln_l += al;
        aln_r += ar;
    }
    static void ratio(const char *label, long long num, long long den) {
        if (den == 0) fprintf(stderr, "%s%s\n", label, num == 0 ? "nan" : "inf");
        else fprintf(stderr, "%s%g\n", label, (double)((float)num / (float)den));
    }
    void print() const {
        fprintf(stderr, "read count:\t%lld\n", read_count);
        ratio("Clipped %:\t", clipped, read_count);
        ratio("% With Supplementary alns:\t", sup, read_count);
        ratio("Artifact rate:\t", art, read_count);
        ratio("% With Supplementary alns and artifacts:\t", art_sup, read_count);
        ratio("Artifact rate left only:\t", aln_l, read_count);
        ratio("Artifact rate right only:\t", aln_r, read_count);
    }
};

static bool q_consuming(uint32_t op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
static bool r_consuming(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }

static Rec build_rec(const std::string &qname, int32_t tid, int32_t pos, int mapq, uint16_t flag, int32_t mtid, int32_t mpos,
                     int32_t tlen, const std::vector<uint32_t> &cig, const std::vector<uint8_t> &codes,
                     const std::vector<uint8_t> &qual, const uint8_t *aux, size_t aux_len) {
    Rec n;
    const size_t lqn = qname.size() + 1, lq = codes.size();
    n.d.assign(32 + lqn + 4 * cig.size() + (lq + 1) / 2 + lq + aux_len, 0);
    int64_t reflen = 0;
    for (uint32_t c : cig) if (r_consuming(c & 15)) reflen += c >> 4;
    n.wr<int32_t>(0, tid);
    n.wr<int32_t>(4, pos);
    n.d[8] = (uint8_t)lqn;
    n.d[9] = (uint8_t)mapq;
    n.wr<uint16_t>(10, (uint16_t)reg2bin(pos < 0 ? 0 : pos, (pos < 0 ? 0 : pos) + (reflen > 0 ? reflen : 1)));
    n.wr<uint16_t>(12, (uint16_t)cig.size());
    n.wr<uint16_t>(14, flag);
    n.wr<int32_t>(16, (int32_t)lq);
    n.wr<int32_t>(20, mtid);
    n.wr<int32_t>(24, mpos);
    n.wr<int32_t>(28, tlen);
    memcpy(n.d.data() + 32, qname.c_str(), lqn);
    size_t off = 32 + lqn;
    if (!cig.empty()) memcpy(n.d.data() + off, cig.data(), 4 * cig.size());
    off += 4 * cig.size();
    for (size_t k = 0; k < lq; k++) n.d[off + (k >> 1)] |= (uint8_t)(codes[k] << ((~k & 1) << 2));
    off += (lq + 1) / 2;
    if (lq) memcpy(n.d.data() + off, qual.data(), lq);
    off += lq;
    if (aux_len) memcpy(n.d.data() + off, aux, aux_len);
    return n;
}

// filter.d:15-91 clipRead
static void clip_read(Rec &rec, uint32_t rsv) {
    std::vector<uint32_t> cig((size_t)rec.n_cigar());
    for (int k = 0; k < rec.n_cigar(); k++) cig[(size_t)k] = rec.cigar_op(k);
    int64_t pos = rec.pos();
    const int lq = rec.l_seq();
    std::vector<uint8_t> codes((size_t)lq), qual((size_t)lq);
    for (int j = 0; j < lq; j++) {
        codes[(size_t)j] = (rec.seq()[j >> 1] >> ((~j & 1) << 2)) & 15;
        qual[(size_t)j] = rec.qual()[j];
    }
    const std::string name = rec.qname();
    size_t sb = 0, se = (size_t)lq;  // surviving [sb, se) of seq / qual
    const size_t pam = rec.aux_find("am");
    const std::string am = pam == std::string::npos ? std::string() : std::string((const char *)rec.d.data() + pam + 3);
    const size_t semi = am.find(';');
    const std::string sides[2] = {am.substr(0, semi), semi == std::string::npos ? std::string() : am.substr(semi + 1)};
    auto art_ref_len = [&](int side) -> int64_t {
        const std::string &f = sides[side];
        const size_t c1 = f.find(','), c2 = c1 == std::string::npos ? c1 : f.find(',', c1 + 1);
        std::vector<uint32_t> ac;
        if (c2 == std::string::npos || !parse_cigar_string(f.substr(c2 + 1), ac)) throw std::runtime_error("malformed am tag: " + am);
        int64_t n = 0;
        for (uint32_t c : ac) if (r_consuming(c & 15)) n += c >> 4;
        return n;
    };
    auto aligned_len = [&]() { int64_t n = 0; for (uint32_t c : cig) if (r_consuming(c & 15)) n += c >> 4; return n; };
    auto reset = [&]() {  // filter.d:47-51 / 79-83: a fresh (zero-filled) record keeping name, sequence, qualities
        std::vector<uint8_t> c2(codes.begin() + (long)sb, codes.begin() + (long)se), q2(qual.begin() + (long)sb, qual.begin() + (long)se);
        rec = build_rec(name, 0, 0, 0, 0, 0, 0, 0, {}, c2, q2, nullptr, 0);
    };
    if (rsv & 2) {  // filter.d:22-54
        int64_t to_trim = art_ref_len(0);
        uint32_t hard = 0;
        if (to_trim < aligned_len()) {
            while (to_trim) {
                const uint32_t op = cig[0] & 15;
                if (q_consuming(op)) { sb++; hard++; }
                if (r_consuming(op)) { pos++; to_trim--; }
                cig[0] -= 16;  // length - 1
                if ((cig[0] >> 4) == 0) cig.erase(cig.begin());
            }
        } else { reset(); return; }
        cig.insert(cig.begin(), (hard << 4) | 5u);
    }
    if (rsv & 4) {  // filter.d:55-86
        int64_t to_trim = art_ref_len(1);
        uint32_t hard = 0;
        if (to_trim < aligned_len()) {
            while (to_trim) {
                const uint32_t op = cig.back() & 15;
                if (q_consuming(op)) { se--; hard++; }
                if (r_consuming(op)) to_trim--;
                cig.back() -= 16;
                if ((cig.back() >> 4) == 0) cig.pop_back();
            }
        } else { reset(); return; }
        cig.push_back((hard << 4) | 5u);
    }
    // filter.d:87-90: cigar, sequence, qscores, pos; every other field and the aux tags stay
    std::vector<uint8_t> c2(codes.begin() + (long)sb, codes.begin() + (long)se), q2(qual.begin() + (long)sb, qual.begin() + (long)se);
    const size_t ao = rec.aux_off();
    std::vector<uint8_t> aux(rec.d.begin() + (long)ao, rec.d.end());
    rec = build_rec(name, rec.tid(), (int32_t)pos, rec.mapq(), (uint16_t)rec.flag(), rec.mtid(), rec.mpos(), rec.tlen(), cig, c2, q2,
                    aux.data(), aux.size());
}

// filter.d:169-268
static int out_main(const std::string &cl, const Opts &o) {
    const int nthreads = o.threads > 0 ? o.threads : 2;
    Pool pool(nthreads);  // (reader and writer share it)
    try {
        Reader reader(o.pos[1], &pool);
        Header hdr = reader.header();
        hdr.add_pg("fade-extract", "fade", FADE_VERSION, cl);  // filter.d:173-180 (the reference reuses this ID)
        const OutFmt fmt = o.bam ? OutFmt::BAM : o.ubam ? OutFmt::UBAM : OutFmt::SAM;
        Writer writer(stdout, fmt, hdr, &pool);
        OutStats stats;
        std::vector<Rec> in, out;
        auto next_chunk = [&]() { in.clear(); return reader.read_chunk(in, 65536) > 0; };
        if (o.clip) {  // filter.d:184-212
            fprintf(stderr, "[W::fade-out] Using the -c flag means the output SAM/BAM will not be sorted (regardless of prior sorting)\n");
            fprintf(stderr, "[W::fade-out] You also may need to fix mate information with a tool like Picard FixMateInformation\n");
            while (next_chunk()) {
                for (Rec &r : in) {
                    stats.read_count++;
                    uint32_t v;
                    if (rs_of(r, v)) {
                        stats.parse(v);
                        if (v & 6) clip_read(r, v);
                    }
                }
                writer.write(in);
            }
        } else {
            // filter.d:216-220: the first ten records decide whether the input looks name-sorted
            std::vector<Rec> all_first;
            next_chunk();
            bool sorted = true;
            for (size_t k = 1; k < std::min<size_t>(in.size(), 10); k++)
                if (numerically_aware_cmp(in[k].qname(), in[k - 1].qname()) < 0) sorted = false;
            if (sorted) {  // filter.d:221-246
                fprintf(stderr, "[W::fade-out] Output looks name-sorted, ejecting all reads with same readname if any have an artifact\n");
                std::vector<Rec> group;
                auto flush_group = [&]() {
                    bool art_found = false;
                    for (const Rec &r : group) {
                        stats.read_count++;
                        uint32_t v;
                        if (!rs_of(r, v)) continue;
                        stats.parse(v);
                        if (v & 6) art_found = true;
                    }
                    if (!art_found) for (Rec &r : group) out.push_back(std::move(r));
                    group.clear();
                };
                do {
                    out.clear();
                    for (Rec &r : in) {
                        if (!group.empty() && strcmp(group.back().qname(), r.qname()) != 0) flush_group();
                        group.push_back(std::move(r));
                    }
                    writer.write(out);
                } while (next_chunk());
                out.clear();
                if (!group.empty()) flush_group();
                writer.write(out);
            } else {  // filter.d:247-265
                fprintf(stderr, "[W::fade-out] Output doesn't look name-sorted, ejecting by only reads with an artifact\n");
                do {
                    out.clear();
                    for (Rec &r : in) {
                        stats.read_count++;
                        uint32_t v;
                        if (!rs_of(r, v)) continue;  // filter.d:254-257: records without rs are not written here
                        stats.parse(v);
                        if (!(v & 6)) out.push_back(std::move(r));
                    }
                    writer.write(out);
                } while (next_chunk());
            }
        }
        writer.close();
        stats.print();  // filter.d:267
    } catch (const std::exception &e) {
        fprintf(stderr, "[E::fade out] %s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char **argv) {
    std::string cl;  // app.d:66
    for (int i = 0; i < argc; i++) { if (i) cl += ' '; cl += argv[i]; }
    if (argc == 1) { print_full_help(); return 0; }  // app.d:67-73
    const std::string sub = argv[1];
    if (sub == "annotate") {
        Opts o;
        std::string err;
        if (!parse_opts(argc, argv, o, err)) {
            fprintf(stderr, "std.getopt.GetOptException: %s\n", err.c_str());
            return 1;
        }
        if (!options_allowed(o, "tmwbugBsTS")) return 1;
        // app.d:84-89: helpWanted | args.length < 3 (args = prog, "annotate", positionals...)
        if (o.help || o.pos.size() < 2) { print_anno_help(); return 0; }
        if (o.pos.size() < 3) {  // the reference indexes args[2] and dies; say why instead
            print_anno_help();
            fprintf(stderr, "[E::fade-annotate] an indexed fasta reference is required\n");
            return 1;
        }
        if (o.bam && o.ubam) {  // app.d:94-99
            fprintf(stderr, "[E::fade-annotate] Please use only one of the b or u flags\n");
            return 1;
        }
        if (!o.out_shards.empty() && o.gpus < 2) {
            fprintf(stderr, "[E::fade-annotate] --out-shards PREFIX writes one file per device: it goes with --gpus N (N >= 2)\n");
            return 1;
        }
        // --gpus N on a BAM file: one process per GPU, each on its own share of the input (annotate_lanes_main); input that
        // cannot be cut (a pipe, SAM text, a small file) is read by one process that deals batches to the N devices
        if (o.gpus > 1 && !lane_env().on && !(getenv("FADE_LANES") && atoi(getenv("FADE_LANES")) == 0)) {
            bool fall_back = true;
            const int lrc = annotate_lanes_main(cl, o, &fall_back);
            if (lrc == 0 || !fall_back) return lrc;
            if (!o.out_shards.empty()) {
                fprintf(stderr, "[E::fade-annotate] --out-shards needs an input that can be cut into ranges: a BAM file (not a pipe, not SAM text) of at least %d BGZF blocks\n", 8 * o.gpus);
                return 1;
            }
        }
        // BAM file in, BAM or uBAM out: the file path on the device (FADE_BAM_DEVICE=0: the host pipeline)
        if ((o.bam || o.ubam) && (o.gpus <= 1 || lane_env().on) && !(getenv("FADE_BAM_DEVICE") && atoi(getenv("FADE_BAM_DEVICE")) == 0)) {
            bool fall_back = true;
            const int src = annotate_stream_main(cl, o, &fall_back);
            if (src == 0 || !fall_back) return src;
        }
        const int rc = annotate_main(cl, o);
        if (o.timing) {
            long rss_kb = 0, hwm_kb = 0;
            if (FILE *f = fopen("/proc/self/status", "r")) {
                char line[256];
                while (fgets(line, sizeof line, f)) {
                    sscanf(line, "VmRSS: %ld", &rss_kb);
                    sscanf(line, "VmHWM: %ld", &hwm_kb);
                }
                fclose(f);
            }
            if (const char *dump = getenv("FADE_SMAPS_DUMP")) {  // tools/smaps_top.py
                FILE *fi = fopen("/proc/self/smaps", "r"), *fo = fopen(dump, "w");
                char buf[4096];
                size_t n;
                while (fi && fo && (n = fread(buf, 1, sizeof buf, fi)) > 0) fwrite(buf, 1, n, fo);
                if (fi) fclose(fi);
                if (fo) fclose(fo);
            }
            fprintf(stderr, "[timing] since process start %.3f s (annotate returned); resident %ld MB, peak %ld MB\n", since_process_start(),
                    rss_kb >> 10, hwm_kb >> 10);
        }
        return rc;
    }
    if (sub == "extract") {  // app.d:130-153
        Opts o;
        std::string err;
        if (!parse_opts(argc, argv, o, err)) {
            fprintf(stderr, "std.getopt.GetOptException: %s\n", err.c_str());
            return 1;
        }
        if (!options_allowed(o, "tbu")) return 1;
        if (o.help || o.pos.size() < 2) { print_extract_help(); return 0; }
        if (o.bam && o.ubam) {
            fprintf(stderr, "[E::fade-annotate] Please use only one of the b or u flags\n");  // app.d:149 (sic)
            return 1;
        }
        return extract_main(cl, o);
    }
    if (sub == "out") {  // app.d:102-129
        Opts o;
        std::string err;
        if (!parse_opts(argc, argv, o, err)) {
            fprintf(stderr, "std.getopt.GetOptException: %s\n", err.c_str());
            return 1;
        }
        if (!options_allowed(o, "ctbu")) return 1;
        if (o.help || o.pos.size() < 2) { print_out_help(); return 0; }
        if (o.bam && o.ubam) {
            fprintf(stderr, "[E::fade-annotate] Please use only one of the b or u flags\n");  // app.d:122 (sic)
            return 1;
        }
        return out_main(cl, o);
    }
    if (sub == "stats" || sub == "stats-clip") {
        fprintf(stderr, "[E::fade] %s is outside the MI355X annotate hot path; run the reference fade for it\n", sub.c_str());
        return 1;
    }
    if (sub == "-h" || sub == "--help") { print_full_help(); return 0; }
    fprintf(stderr, "[E::fade] %s is not a fade subcommand\n", sub.c_str());  // app.d:218-220
    print_full_help();
    return 1;
}
