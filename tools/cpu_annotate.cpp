// cpu_annotate.cpp — CPU end-to-end comparator for `fade annotate` (MEASUREMENT TOOL, not product code):
// the product's own reader / writer stages (fade_amd/csrc/host/hts_lite.hpp) around the oracle's striped AVX2
// restatement of annotateTask (oracle/: anno.d:55-110 + analysis.d:22-124, one SW call per qualifying clip as the
// reference makes them).  Same command line and the same output bytes as `fade annotate` (but for the @PG CL field),
// so that "GPU annotate vs CPU annotate" is measured on the same BAM with a stated thread count, and the two outputs
// can be diffed.  Lives under tools/ and links liboracle: nothing under fade_amd/ may do that.
//   g++ -O2 -std=c++17 -o tools/cpu_annotate tools/cpu_annotate.cpp -Ioracle -Loracle -lfadeoracle -Wl,-rpath,$PWD/oracle -lz -lpthread
#include "../fade_amd/csrc/host/hts_lite.hpp"
#include "fade_oracle.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

using namespace htsl;

int main(int argc, char **argv) {
    int threads = 16, floor_len = 5, window = 300, batch = 262144;
    bool bam = false, ubam = false, timing = false, records_only = false;
    std::vector<std::string> pos;
    std::string cl = "fade";
    for (int i = 1; i < argc; i++) cl += std::string(" ") + argv[i];
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-t" && i + 1 < argc) threads = atoi(argv[++i]);
        else if (a == "--min-length" && i + 1 < argc) floor_len = atoi(argv[++i]);
        else if (a == "-w" && i + 1 < argc) window = atoi(argv[++i]);
        else if (a == "--batch" && i + 1 < argc) batch = atoi(argv[++i]);
        else if (a == "-b") bam = true;
        else if (a == "-u") ubam = true;
        else if (a == "--timing") timing = true;
        else if (a == "--records-only") records_only = true;  // the path from bam1_t-shaped records alone: see below
        else pos.push_back(a);
    }
    if (pos.size() != 3 || pos[0] != "annotate") {
        fprintf(stderr, "usage: cpu_annotate annotate [-t N] [--min-length N] [-w N] [-b|-u] [--timing] <BAM/SAM> <fasta>\n");
        return 2;
    }
    const auto t_start = std::chrono::steady_clock::now();
    double t_sw = 0;
    fprintf(stderr, "[W::fade annotate] Output SAM/BAM will not be sorted (regardless of prior sorting)\n");
    // The same three overlapping stages as the GPU driver (reader -> annotateTask -> tags + writer) on one shared pool, with
    // the same zero-copy handling of BAM input (records framed in place, new tags as a suffix): the best CPU run this code
    // base can offer, so that the end-to-end comparison is about the re-alignment and not about the plumbing.
    struct Work {
        std::vector<Rec> recs;  // SAM input
        RecordBlock blk;        // BAM input
        bool is_block = false;
        Writer::BlockOut bout;
        size_t n() const { return is_block ? blk.size() : recs.size(); }
    };
    std::string stage_err;
    std::mutex err_m;
    auto set_err = [&](const std::string &e) {
        std::lock_guard<std::mutex> l(err_m);
        if (stage_err.empty()) stage_err = e;
    };
    try {
        Pool pool(threads);
        Reader reader(pos[1], &pool);
        Fasta fa = load_fasta(pos[2]);
        Header hdr = reader.header();
        hdr.add_pg("fade-annotate", "fade", "v0.5.0-mi355x", cl);
        const Header &h = reader.header();
        std::vector<const char *> names(h.names.size()), seqs(h.names.size());
        std::vector<int64_t> lens(h.names.size());
        for (size_t k = 0; k < h.names.size(); k++) {
            size_t f = 0;
            while (f < fa.names.size() && fa.names[f] != h.names[k]) f++;
            if (f == fa.names.size()) throw std::runtime_error("reference " + h.names[k] + " is not in the FASTA");
            names[k] = h.names[k].c_str();
            seqs[k] = fa.seqs[f].data();
            lens[k] = h.lens[k];
        }
        fo_genome g;
        g.n_contigs = (int)names.size();
        g.names = names.data();
        g.lengths = lens.data();
        g.seqs = seqs.data();
        fo_params prm;
        fo_params_default(&prm);
        prm.striped = 1;
        const OutFmt fmt = bam ? OutFmt::BAM : ubam ? OutFmt::UBAM : OutFmt::SAM;
        // --records-only (bench.py's cpu_baseline_from_records): the input is read, inflated and framed FIRST, outside the
        // clock; the clock then covers what starts from a record as htslib hands it over (bam1_t: anno.d:55) — the gate,
        // reverse complement, FASTA window, SW, gates, tag strings of every record on all threads — and nothing is written.
        std::unique_ptr<Writer> writer_p;
        if (!records_only) writer_p.reset(new Writer(stdout, fmt, hdr, &pool));
        BoundedQueue<std::unique_ptr<Work>> q_in(records_only ? (size_t)1 << 20 : 2), q_out(2);
        std::thread t_reader([&] {
            try {
                for (;;) {
                    std::unique_ptr<Work> w(new Work());
                    w->is_block = true;
                    const size_t got = w->is_block ? reader.read_block(w->blk, (size_t)batch) : reader.read_chunk(w->recs, (size_t)batch);
                    if (!got) break;
                    q_in.push(std::move(w));
                }
            } catch (const std::exception &e) {
                set_err(e.what());
            }
            q_in.close();
        });
        std::thread t_writer([&] {
            std::unique_ptr<Work> w;
            try {
                while (q_out.pop(w)) {
                    if (w->is_block) writer_p->write_block(w->blk, w->bout);
                    else writer_p->write(w->recs);
                }
            } catch (const std::exception &e) {
                set_err(e.what());
                while (q_out.pop(w)) {}
            }
        });
        struct Tags { std::string t[4]; };
        std::unique_ptr<Work> w;
        size_t n_records = 0;
        if (records_only) t_reader.join();  // (every record is in memory before the clock starts)
        try {
        while (q_in.pop(w)) {
            n_records += w->n();
            const auto t0 = std::chrono::steady_clock::now();
            const size_t n = w->n(), nt = (size_t)pool.size() * 8;
            std::vector<uint8_t> rs(n, 0);
            std::vector<std::vector<std::pair<uint32_t, Tags>>> tags_t(nt);
            std::vector<std::vector<std::pair<uint32_t, Rec>>> owned_t(nt);
            Writer::BlockOut &o = w->bout;
            if (w->is_block) o.sfx_off.assign(n + 1, 0);
            auto annotate = [&](const auto &r, fo_anno &a, std::vector<uint32_t> &cig) {
                cig.resize((size_t)r.n_cigar() + 1);
                if (r.n_cigar()) memcpy(cig.data(), r.cigar_bytes(), 4 * (size_t)r.n_cigar());
                fo_read rd;
                rd.qname = r.qname();
                rd.flag = (uint16_t)r.flag();
                rd.tid = r.tid();
                rd.pos = r.pos();
                rd.n_cigar = r.n_cigar();
                rd.cigar = cig.data();
                rd.l_seq = r.l_seq();
                rd.seq4 = r.seq();
                rd.qual = r.qual();
                rd.has_sa = r.aux_exists("SA") ? 1 : 0;
                memset(&a, 0, sizeof a);
                fo_annotate_task(&prm, &g, &rd, floor_len, window, &a);
            };
            auto tag_owned = [&](Rec &r, const fo_anno &a) {
                r.aux_update_uint("rs", a.rs);  // anno.d:63,94
                if (a.has_tags) {               // anno.d:98-107
                    r.aux_update_str("am", a.am);
                    r.aux_update_str("as", a.as_);
                    r.aux_update_str("ar", a.ar);
                    r.aux_update_str("ab", a.ab);
                }
            };
            pool.parallel_for(nt, [&](size_t t) {
                std::vector<uint32_t> cig;
                fo_anno a;
                for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                    if (!w->is_block) {
                        annotate(w->recs[i], a, cig);
                        tag_owned(w->recs[i], a);
                        fo_anno_free(&a);
                        continue;
                    }
                    const RecView v = w->blk.view(i);
                    annotate(v, a, cig);
                    bool has_ours = false;  // a record that already carries one of the five tags is rebuilt (update in place)
                    for (size_t p = v.aux_off(); p + 3 <= v.nbytes();) {
                        const size_t fs = v.aux_field_size(p + 2);
                        if (!fs) break;
                        const uint8_t a0 = v.bytes()[p], a1 = v.bytes()[p + 1];
                        has_ours |= (a0 == 'r' && a1 == 's') || (a0 == 'a' && (a1 == 'm' || a1 == 's' || a1 == 'r' || a1 == 'b'));
                        p += 2 + fs;
                    }
                    if (has_ours) {
                        Rec r;
                        r.d.assign(v.bytes(), v.bytes() + v.nbytes());
                        tag_owned(r, a);
                        owned_t[t].emplace_back((uint32_t)i, std::move(r));
                    } else {
                        rs[i] = a.rs;
                        size_t len = 4;  // "rs" 'C' value
                        if (a.has_tags) {
                            Tags tg;
                            tg.t[0] = a.am; tg.t[1] = a.as_; tg.t[2] = a.ar; tg.t[3] = a.ab;
                            for (int k = 0; k < 4; k++) len += 3 + tg.t[k].size() + 1;
                            tags_t[t].emplace_back((uint32_t)i, std::move(tg));
                        }
                        o.sfx_off[i + 1] = (uint32_t)len;
                    }
                    fo_anno_free(&a);
                }
            });
            if (w->is_block) {
                for (auto &v : owned_t)
                    for (auto &e : v) o.owned.push_back(std::move(e));
                for (size_t i = 0; i < n; i++) o.sfx_off[i + 1] += o.sfx_off[i];
                o.sfx.resize(o.sfx_off[n]);
                pool.parallel_for(nt, [&](size_t t) {
                    static const char tn[4][2] = {{'a', 'm'}, {'a', 's'}, {'a', 'r'}, {'a', 'b'}};
                    size_t next_tag = 0;
                    for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                        if (o.sfx_off[i + 1] == o.sfx_off[i]) continue;
                        uint8_t *d = o.sfx.data() + o.sfx_off[i];
                        d[0] = 'r'; d[1] = 's'; d[2] = 'C'; d[3] = rs[i];
                        d += 4;
                        if (next_tag < tags_t[t].size() && tags_t[t][next_tag].first == i) {
                            const Tags &tg = tags_t[t][next_tag++].second;
                            for (int k = 0; k < 4; k++) {
                                d[0] = (uint8_t)tn[k][0]; d[1] = (uint8_t)tn[k][1]; d[2] = 'Z';
                                memcpy(d + 3, tg.t[k].c_str(), tg.t[k].size() + 1);
                                d += 3 + tg.t[k].size() + 1;
                            }
                        }
                    }
                });
            }
            t_sw += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (!records_only) q_out.push(std::move(w));
        }
        } catch (const std::exception &e) {  // let the reader run out so that it can be joined
            set_err(e.what());
            while (q_in.pop(w)) {}
        }
        q_out.close();
        t_writer.join();
        if (t_reader.joinable()) t_reader.join();
        if (!stage_err.empty()) throw std::runtime_error(stage_err);
        if (writer_p) writer_p->close();
        if (records_only) printf("{\"records\": %zu, \"seconds\": %.6f, \"threads\": %d}\n", n_records, t_sw, threads);
    } catch (const std::exception &e) {
        fprintf(stderr, "[E::cpu_annotate] %s\n", e.what());
        return 1;
    }
    if (timing)
        fprintf(stderr, "[timing] total %.3f s, annotateTask + tags stage %.3f s (%d threads; reader, this stage and the writer overlap on one pool)\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), t_sw, threads);
    return 0;
}
