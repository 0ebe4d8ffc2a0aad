// sam2bam.cpp — SAM/BAM -> BAM with the product's own reader / writer, through the block path `fade annotate` uses
// (records framed in place, the writer's layout hints to the compressor).  Measurement helper: makes the BAM inputs of
// tools/e2e_cli.py without annotating anything; tests/test_bgzf_codec.py uses it to check the hinted BGZF output.
//   g++ -O2 -std=c++17 -o tools/sam2bam tools/sam2bam.cpp -lz -lpthread
#include "../fade_amd/csrc/host/hts_lite.hpp"
using namespace htsl;
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    try {
        Pool pool(16);
        Reader rd(argv[1], &pool);
        Writer wr(stdout, OutFmt::BAM, rd.header(), &pool);
        RecordBlock blk;
        const Writer::BlockOut nothing_added;
        while (rd.read_block(blk, 262144)) wr.write_block(blk, nothing_added);
        wr.close();
    } catch (const std::exception &e) {
        fprintf(stderr, "sam2bam: %s\n", e.what());
        return 1;
    }
    return 0;
}
