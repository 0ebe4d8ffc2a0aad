// sam2bam.cpp — SAM/BAM -> BAM with the product's own reader / writer (measurement helper: makes the BAM inputs of
// tools/e2e_cli.py without annotating anything).
//   g++ -O2 -std=c++17 -o tools/sam2bam tools/sam2bam.cpp -lz -lpthread
#include "../fade_amd/csrc/host/hts_lite.hpp"
using namespace htsl;
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    try {
        Pool rp(16), wp(16);
        Reader rd(argv[1], &rp);
        Writer wr(stdout, OutFmt::BAM, rd.header(), &wp);
        std::vector<Rec> recs;
        for (;;) {
            recs.clear();
            if (!rd.read_chunk(recs, 262144)) break;
            wr.write(recs);
        }
        wr.close();
    } catch (const std::exception &e) {
        fprintf(stderr, "sam2bam: %s\n", e.what());
        return 1;
    }
    return 0;
}
