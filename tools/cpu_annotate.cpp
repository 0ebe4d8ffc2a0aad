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

using namespace htsl;

int main(int argc, char **argv) {
    int threads = 16, floor_len = 5, window = 300, batch = 262144;
    bool bam = false, ubam = false, timing = false;
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
        else pos.push_back(a);
    }
    if (pos.size() != 3 || pos[0] != "annotate") {
        fprintf(stderr, "usage: cpu_annotate annotate [-t N] [--min-length N] [-w N] [-b|-u] [--timing] <BAM/SAM> <fasta>\n");
        return 2;
    }
    const auto t_start = std::chrono::steady_clock::now();
    double t_sw = 0;
    fprintf(stderr, "[W::fade annotate] Output SAM/BAM will not be sorted (regardless of prior sorting)\n");
    try {
        Pool rpool(threads), pool(threads), wpool(threads);
        Reader reader(pos[1], &rpool);
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
        Writer writer(stdout, fmt, hdr, &wpool);
        std::vector<Rec> recs;
        for (;;) {
            recs.clear();
            if (!reader.read_chunk(recs, (size_t)batch)) break;
            const auto t0 = std::chrono::steady_clock::now();
            const size_t n = recs.size(), nt = (size_t)pool.size() * 8;
            pool.parallel_for(nt, [&](size_t t) {
                std::vector<uint32_t> cig;
                for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                    Rec &r = recs[i];
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
                    fo_anno a;
                    memset(&a, 0, sizeof a);
                    fo_annotate_task(&prm, &g, &rd, floor_len, window, &a);
                    r.aux_update_uint("rs", a.rs);          // anno.d:63,94
                    if (a.has_tags) {                       // anno.d:98-107
                        r.aux_update_str("am", a.am);
                        r.aux_update_str("as", a.as_);
                        r.aux_update_str("ar", a.ar);
                        r.aux_update_str("ab", a.ab);
                    }
                    fo_anno_free(&a);
                }
            });
            t_sw += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            writer.write(recs);
        }
        writer.close();
    } catch (const std::exception &e) {
        fprintf(stderr, "[E::cpu_annotate] %s\n", e.what());
        return 1;
    }
    if (timing)
        fprintf(stderr, "[timing] total %.3f s, annotateTask stage %.3f s (%d threads; stages run one after another per chunk)\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), t_sw, threads);
    return 0;
}
