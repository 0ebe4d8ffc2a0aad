// Where the BAM reader's time goes (run on the box whose cores you care about):
//   g++ -O2 -std=c++17 -o /tmp/reader_stages tools/reader_stages.cpp -lz -lpthread && /tmp/reader_stages in.bam 16
#include "../fade_amd/csrc/host/hts_lite.hpp"
#include <chrono>
using namespace htsl;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int nt = atoi(argv[2]);
    {
        Pool pool(nt);
        FILE *f = fopen(argv[1], "rb");
        ByteSource src(f);
        BgzfIn bg(&src, &pool);
        const double t0 = now();
        size_t tot = 0;
        while (bg.more()) {
            tot += bg.avail();
            bg.consume(bg.avail());
        }
        printf("BGZF layer (read + inflate + CRC): %.3f s, %.0f MB/s inflated\n", now() - t0, tot / (now() - t0) / 1e6);
        fclose(f);
    }
    {
        Pool pool(nt);
        const double t0 = now();
        Reader rd(argv[1], &pool);
        size_t n = 0;
        std::vector<Rec> recs;
        double t_free = 0;
        for (;;) {
            const double a = now();
            recs.clear();
            t_free += now() - a;
            const size_t k = rd.read_chunk(recs, 262144);
            if (!k) break;
            n += k;
        }
        printf("Reader (BGZF + framing + per-record copies): %.3f s, %.2f M records/s; freeing the records %.3f s\n", now() - t0,
               n / (now() - t0) / 1e6, t_free);
    }
    return 0;
}
