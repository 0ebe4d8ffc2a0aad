// gpu_deflate_model.cpp — CPU model of the gfx950 BGZF compressor (bgzf_deflate.hpp), phase by phase with the same
// data structures (per-segment hash tables of 4-way buckets, pieces of 64 positions, token / match bitmaps, the match
// list, seams; 256 position ranges that each emit their bits at a scanned offset, OR-ing the words they share) and the
// SAME serial helpers (bgzf_huff.hpp).  It exists so that the format logic is checked against zlib's inflate where there is no GPU:
//   every stream inflates to its input, on BAM-like, text, random, constant, tiny and empty-ish inputs;
//   sizes are printed next to zlib -6 and -1.
// Build + run: make -C fade_amd/csrc build/gpu_deflate_model && fade_amd/csrc/build/gpu_deflate_model
#include "../../bgzf_huff.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace fadehip::bgzf;

namespace {

constexpr int WG = 256, MIN_MATCH = 4, MAX_MATCH = 258, MAXW = 16;
int WAYS = 4;
// the two geometries of bgzf_deflate_body.hpp, and the knobs of phase A (model only: the device's are constants)
int BLOCK = 0xff00, N_SEG = 8, N_BUCKETS = 512, SEG_CAP = (8192 + 2560) / 8, SEED_PIECES = 16;
bool SKIP_RUNS = false, SEAM = true;
int SHORT4 = 32768, SHORT5 = 32768, SHORT6 = 32768, NEAR = 2;

inline uint32_t load32(const uint8_t *d, int p) {
    uint32_t v;
    memcpy(&v, d + p, 4);
    return v;
}
inline uint32_t hash4(uint32_t v) { return (((v * 0x9E3779B1u) >> 16) * (uint32_t)N_BUCKETS) >> 16; }

// returns the raw DEFLATE stream of src[0..n), n <= BLOCK
std::vector<uint8_t> model_deflate(const uint8_t *src, int n, bool lazy) {
    std::vector<uint8_t> data((size_t)BLOCK + 600, 0);
    memcpy(data.data(), src, (size_t)n);
    std::vector<uint32_t> match_rec;
    const int n_words = (BLOCK + 31) / 32;
    std::vector<uint32_t> tok((size_t)n_words + 2, 0), mat((size_t)n_words + 2, 0);
    // ---- phase A (bgzf_deflate_body.hpp phase_a_segment): the block's pieces of 64 positions in N_SEG segments, each with a
    // hash table of its own (N_BUCKETS x 4 ways, newest first), seeded with the SEED_PIECES pieces in front of it; a
    // position's candidates: the nearest of the distances 1..8 whose four bytes agree, and its bucket's four; greedy parse on
    // 64-position masks with one step of laziness; a match ends with its segment, and the seam gives it back what the next
    // segment's parse allows
    const int n_pieces = (n + 63) / 64, seg_pieces = (n_pieces + N_SEG - 1) / N_SEG;
    struct Seam { int rec, over, seg_end, next_end; };
    std::vector<Seam> seams;
    for (int w = 0; w < N_SEG; w++) {
        const int first = std::min(w * seg_pieces, n_pieces), end = std::min((w + 1) * seg_pieces, n_pieces);
        const int seg_end = std::min(n, end * 64);
        std::vector<uint16_t> head((size_t)N_BUCKETS * WAYS, 0);
        auto insert_piece = [&](int piece, uint16_t cand[64][MAXW], uint32_t val[64]) {
            bool ins[64];
            for (int l = 0; l < 64; l++) {  // every lane reads its bucket, then every lane writes (lockstep); of two lanes on one bucket the later stays
                const int p = piece * 64 + l;
                ins[l] = false;
                val[l] = 0;
                for (int k = 0; k < WAYS; k++) cand[l][k] = 0;
                if (p + MIN_MATCH <= n) {
                    val[l] = load32(data.data(), p);
                    for (int k = 0; k < WAYS; k++) cand[l][k] = head[(size_t)hash4(val[l]) * WAYS + k];
                    ins[l] = !SKIP_RUNS || p == 0 || load32(data.data(), p - 1) != val[l];
                }
            }
            for (int l = 0; l < 64; l++)
                if (ins[l]) {
                    uint16_t *bk = &head[(size_t)hash4(val[l]) * WAYS];
                    bk[0] = (uint16_t)(piece * 64 + l + 1);
                    for (int k = 1; k < WAYS; k++) bk[k] = cand[l][k - 1];
                }
        };
        uint16_t cand[64][MAXW];
        uint32_t val[64];
        for (int piece = std::max(0, first - SEED_PIECES); piece < first; piece++) insert_piece(piece, cand, val);
        int carry = first * 64, mcount = 0, over = 0, over_rec = -1;
        bool full = false;
        for (int piece = first; piece < end; piece++) {
            insert_piece(piece, cand, val);
            uint32_t len[64], ulen[64], dist[64];
            for (int l = 0; l < 64; l++) {
                const int p = piece * 64 + l;
                len[l] = ulen[l] = dist[l] = 0;
                if (p + MIN_MATCH > n || p < carry) continue;
                const int maxlen = std::min(MAX_MATCH, n - p);
                int cp[MAXW + 1], nc = 0;
                for (int d = 1; d <= NEAR && d <= p; d++)
                    if (load32(data.data(), p - d) == val[l]) { cp[nc++] = p - d; break; }
                for (int k = 0; k < WAYS; k++)
                    if (cand[l][k]) {
                        const int c = (int)cand[l][k] - 1;
                        if (p - c <= 32768 && c < p && load32(data.data(), c) == val[l]) cp[nc++] = c;
                    }
                for (int k = 0; k < nc; k++) {
                    uint32_t ln = 4;
                    while ((int)ln < maxlen && data[(size_t)cp[k] + ln] == data[(size_t)p + ln]) ln++;
                    if (ln > len[l]) { len[l] = ln; dist[l] = (uint32_t)(p - cp[k]); }
                }
                ulen[l] = len[l];
                len[l] = std::min<uint32_t>(len[l], (uint32_t)std::max(seg_end - p, 0));
                if (len[l] < (uint32_t)MIN_MATCH) len[l] = 0;
                // short matches at a distance cost more bits than the literals they replace (MODEL_SHORT4 / 5 / 6: the
                // greatest distance at which a match of 4 / 5 / 6 bytes is still taken)
                if (len[l] == 4 && (int)dist[l] > SHORT4) len[l] = 0;
                if (len[l] == 5 && (int)dist[l] > SHORT5) len[l] = 0;
                if (len[l] == 6 && (int)dist[l] > SHORT6) len[l] = 0;
            }
            const int cb = piece * 64, valid = std::min(64, n - cb);
            uint64_t has = 0;
            for (int l = 0; l < valid; l++) {
                const bool yield = lazy && len[l] && l < 63 && len[l + 1] > len[l];
                if (len[l] && !yield) has |= 1ull << l;
            }
            if (full || mcount + __builtin_popcountll(has) > SEG_CAP) { full = true; has = 0; }
            int cur = std::max(carry - cb, 0);
            uint64_t tokmask = 0, matmask = 0;
            const uint64_t vmask = valid == 64 ? ~0ull : ((1ull << valid) - 1);
            int j_last = -1;
            while (cur < valid) {
                const uint64_t rem = has & ~((cur >= 64) ? ~0ull : ((1ull << cur) - 1));
                if (!rem) {
                    tokmask |= vmask & ~((1ull << cur) - 1);
                    cur = valid;
                    break;
                }
                const int j = __builtin_ctzll(rem);
                tokmask |= (((j == 63) ? ~0ull : ((1ull << (j + 1)) - 1)) & ~((1ull << cur) - 1));
                matmask |= 1ull << j;
                cur = j + (int)len[j];
                j_last = j;
            }
            if (j_last >= 0 && cb + cur == seg_end) {
                over = (int)(ulen[j_last] - len[j_last]);
                over_rec = (int)match_rec.size() + __builtin_popcountll(matmask & ((1ull << j_last) - 1));
            }
            carry = std::max(carry, cb + cur);
            for (int l = 0; l < 64; l++)
                if ((matmask >> l) & 1) match_rec.push_back(dist[l] | ((len[l] - 3) << 16));
            mcount += __builtin_popcountll(matmask);
            tok[(size_t)(cb >> 5)] |= (uint32_t)tokmask;
            tok[(size_t)(cb >> 5) + 1] |= (uint32_t)(tokmask >> 32);
            mat[(size_t)(cb >> 5)] |= (uint32_t)matmask;
            mat[(size_t)(cb >> 5) + 1] |= (uint32_t)(matmask >> 32);
        }
        if (SEAM && over > 0 && w + 1 < N_SEG && end < n_pieces) seams.push_back({over_rec, over, end * 64, std::min(n, (end + seg_pieces) * 64)});
    }
    for (const Seam &sm : seams) {  // phase_a_seam
        int give = std::min(sm.over, sm.next_end - sm.seg_end);
        for (int q = sm.seg_end; q < sm.seg_end + give; q++)
            if ((mat[(size_t)(q >> 5)] >> (q & 31)) & 1u) { give = q - sm.seg_end; break; }
        if (give <= 0) continue;
        for (int q = sm.seg_end; q < sm.seg_end + give; q++) tok[(size_t)(q >> 5)] &= ~(1u << (q & 31));
        match_rec[(size_t)sm.rec] += (uint32_t)give << 16;
    }
    // ---- phase B: histograms (match index = matches before the position)
    std::vector<uint32_t> mpre((size_t)n_words + 1, 0);
    for (int w = 0; w < n_words; w++) mpre[(size_t)w + 1] = mpre[(size_t)w] + (uint32_t)__builtin_popcount(mat[(size_t)w]);
    uint32_t lfreq[NUM_LITLEN + 2] = {0}, dfreq[NUM_DIST + 2] = {0};
    for (int w = 0; w < n_words; w++) {
        uint32_t tw = tok[(size_t)w];
        while (tw) {
            const int b = __builtin_ctz(tw);
            tw &= tw - 1;
            const int p = 32 * w + b;
            if ((mat[(size_t)w] >> b) & 1) {
                const uint32_t idx = mpre[(size_t)w] + (uint32_t)__builtin_popcount(mat[(size_t)w] & ((1u << b) - 1));
                const uint32_t rec = match_rec[idx];
                lfreq[length_symbol((rec >> 16) + 3).sym]++;
                dfreq[dist_symbol(rec & 0xffffu).sym]++;
            } else lfreq[data[(size_t)p]]++;
        }
    }
    lfreq[256] = 1;
    // ---- code lengths
    auto build = [&](const uint32_t *freq, int nsym, int max_bits, uint8_t *len) {
        uint32_t f[NUM_LITLEN + 2];
        for (int s = 0; s < nsym; s++) f[s] = freq[s];
        int used = 0;
        for (int s = 0; s < nsym; s++) used += f[s] != 0;
        for (int s = 0; used < 2 && s < nsym; s++)  // at least two codes (zlib does the same for old inflaters)
            if (!f[s]) { f[s] = 1; used++; }
        std::vector<std::pair<uint32_t, int>> v;
        for (int s = 0; s < nsym; s++)
            if (f[s]) v.push_back({f[s], s});
        std::sort(v.begin(), v.end());
        std::vector<uint32_t> A(v.size()), bl(32);
        for (size_t k = 0; k < v.size(); k++) A[k] = v[k].first;
        mr_code_lengths(A.data(), (int)A.size());
        limit_code_lengths(A.data(), (int)A.size(), max_bits, bl.data());
        for (int s = 0; s < nsym; s++) len[s] = 0;
        for (size_t k = 0; k < v.size(); k++) len[v[k].second] = (uint8_t)A[k];
    };
    uint8_t ll[NUM_LITLEN + 2], dl[NUM_DIST + 2];
    uint16_t lc[NUM_LITLEN + 2], dc[NUM_DIST + 2];
    build(lfreq, NUM_LITLEN, MAX_LITLEN_BITS, ll);
    build(dfreq, NUM_DIST, MAX_LITLEN_BITS, dl);
    canonical_codes(ll, NUM_LITLEN, MAX_LITLEN_BITS, lc);
    canonical_codes(dl, NUM_DIST, MAX_LITLEN_BITS, dc);
    // ---- header
    std::vector<uint32_t> words((size_t)BLOCK / 2 + 4096, 0);
    uint32_t hdr_bits;
    {
        std::vector<uint32_t> hw(256, 0);
        WordSink sink;
        sink.w = hw.data();
        uint32_t f19[32], sf[32], ss[32], cll[32], clc[32], bl[32];
        MemArr a_f{f19}, a_sf{sf}, a_ss{ss}, a_cll{cll}, a_clc{clc}, a_bl{bl};
        MemArrT<uint8_t> a_ll{ll}, a_dl{dl};
        write_dynamic_header_t(sink, a_ll, a_dl, a_f, a_sf, a_ss, a_cll, a_clc, a_bl);
        hdr_bits = sink.finish();
        for (size_t k = 0; k < hw.size(); k++) words[k] = hw[k];
    }
    // ---- phase D: 256 ranges of 8 bitmap words (256 positions) each
    auto token_bits = [&](int w, int b, uint64_t *bits, int *nb) {
        const int p = 32 * w + b;
        if ((mat[(size_t)w] >> b) & 1) {
            const uint32_t idx = mpre[(size_t)w] + (uint32_t)__builtin_popcount(mat[(size_t)w] & ((1u << b) - 1));
            const uint32_t rec = match_rec[idx];
            const Sym ls = length_symbol((rec >> 16) + 3), ds = dist_symbol(rec & 0xffffu);
            uint64_t v = lc[ls.sym];
            int k = ll[ls.sym];
            v |= (uint64_t)ls.eval << k; k += (int)ls.ebits;
            v |= (uint64_t)dc[ds.sym] << k; k += dl[ds.sym];
            v |= (uint64_t)ds.eval << k; k += (int)ds.ebits;
            *bits = v; *nb = k;
        } else {
            *bits = lc[data[(size_t)p]];
            *nb = ll[data[(size_t)p]];
        }
    };
    uint32_t range_bits[WG + 1];
    for (int t = 0; t < WG; t++) {
        uint32_t sum = 0;
        for (int w = 8 * t; w < 8 * t + 8 && w < n_words; w++) {
            uint32_t tw = tok[(size_t)w];
            while (tw) {
                const int b = __builtin_ctz(tw);
                tw &= tw - 1;
                uint64_t v; int k;
                token_bits(w, b, &v, &k);
                sum += (uint32_t)k;
            }
        }
        if (t == WG - 1) sum += ll[256];
        range_bits[t] = sum;
    }
    uint32_t off = hdr_bits;
    for (int t = 0; t < WG; t++) { const uint32_t s = range_bits[t]; range_bits[t] = off; off += s; }
    const uint32_t total_bits = off;
    for (int t = 0; t < WG; t++) {
        uint64_t acc = 0;
        int cnt = (int)(range_bits[t] & 31u);
        uint32_t wi = range_bits[t] >> 5;
        auto put = [&](uint64_t v, int k) {
            acc |= v << cnt;
            cnt += k;
            while (cnt >= 32) { words[wi++] |= (uint32_t)acc; acc >>= 32; cnt -= 32; }  // (device: first / last word by atomicOr)
        };
        for (int w = 8 * t; w < 8 * t + 8 && w < n_words; w++) {
            uint32_t tw = tok[(size_t)w];
            while (tw) {
                const int b = __builtin_ctz(tw);
                tw &= tw - 1;
                uint64_t v; int k;
                token_bits(w, b, &v, &k);
                // a token has up to 48 bits: in two pieces so that the 64-bit accumulator (< 32 pending bits) never overflows
                if (k > 24) { put(v & 0xffffffu, 24); put(v >> 24, k - 24); }
                else put(v, k);
            }
        }
        if (t == WG - 1) put(lc[256], ll[256]);
        if (cnt) words[wi] |= (uint32_t)acc;
    }
    const size_t nbytes = (total_bits + 7) / 8;
    std::vector<uint8_t> out(nbytes);
    memcpy(out.data(), words.data(), nbytes);
    if (nbytes > (size_t)n + 5) {  // stored block instead
        out.assign((size_t)n + 5, 0);
        out[0] = 1;
        out[1] = (uint8_t)(n & 255); out[2] = (uint8_t)(n >> 8);
        out[3] = (uint8_t)~out[1]; out[4] = (uint8_t)~out[2];
        memcpy(out.data() + 5, src, (size_t)n);
    }
    return out;
}

bool inflate_ok(const std::vector<uint8_t> &z, const uint8_t *src, int n) {
    std::vector<uint8_t> back((size_t)n + 64);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(z.data());
    zs.avail_in = (uInt)z.size();
    zs.next_out = back.data();
    zs.avail_out = (uInt)back.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == (uLong)n && zs.avail_in == 0 && memcmp(back.data(), src, (size_t)n) == 0;
    inflateEnd(&zs);
    return ok;
}

size_t zlib_size(const uint8_t *src, int n, int level) {
    std::vector<uint8_t> out((size_t)n + 1024);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<uint8_t *>(src);
    zs.avail_in = (uInt)n;
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    const size_t r = zs.total_out;
    deflateEnd(&zs);
    return r;
}

// BAM-like payload: records of 150-base reads (fixed fields, qname r<k>, CIGAR, packed bases, qualities, an rs tag)
std::vector<uint8_t> bam_like(size_t bytes, uint32_t seed, bool run_quals) {
    std::mt19937 rng(seed);
    std::vector<uint8_t> v;
    uint32_t k = seed * 1000;
    while (v.size() < bytes) {
        std::string nm = "r" + std::to_string(k++ / 2);
        const uint32_t bs = 32 + (uint32_t)nm.size() + 1 + 4 + 75 + 150 + 4;
        auto p32 = [&](uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); };
        p32(bs); p32(rng() % 4); p32(rng() % 25000000);
        v.push_back((uint8_t)(nm.size() + 1)); v.push_back(60);
        v.push_back(0x49); v.push_back(0x12);
        v.push_back(1); v.push_back(0);
        v.push_back((k & 1) ? 99 : 147); v.push_back(0);
        p32(150); p32(0xffffffffu); p32(0xffffffffu); p32(0);
        for (char c : nm) v.push_back((uint8_t)c);
        v.push_back(0);
        p32((150u << 4) | 0);
        static const uint8_t nt[4] = {1, 2, 4, 8};
        for (int i = 0; i < 75; i++) v.push_back((uint8_t)((nt[rng() & 3] << 4) | nt[(rng() >> 2) & 3]));
        if (run_quals) {  // the law of tests/test_bgzf_codec.py::test_block_writer_with_layout_hints
            uint8_t q = 37;
            for (int i = 0; i < 150; i++) {
                if (rng() % 100 < 8) { const uint32_t u = rng() % 100; q = u < 5 ? 2 : u < 15 ? 11 : u < 30 ? 25 : 37; }
                v.push_back(q);
            }
        } else
            for (int i = 0; i < 150; i++) v.push_back((uint8_t)(20 + rng() % 21));
        v.push_back('r'); v.push_back('s'); v.push_back('C'); v.push_back((uint8_t)(rng() % 10 == 0 ? 3 : 0));
    }
    v.resize(bytes);
    return v;
}

}  // namespace

int main(int argc, char **argv) {
    // the shared helpers first
    {
        uint32_t x2n[32];
        crc_x2n_table(x2n);
        std::mt19937 rng(5);
        std::vector<uint8_t> a(1000), b(777);
        for (auto &c : a) c = (uint8_t)rng();
        for (auto &c : b) c = (uint8_t)rng();
        std::vector<uint8_t> ab(a);
        ab.insert(ab.end(), b.begin(), b.end());
        const uint32_t ca = (uint32_t)crc32(0, a.data(), (uInt)a.size()), cb = (uint32_t)crc32(0, b.data(), (uInt)b.size());
        const uint32_t want = (uint32_t)crc32(0, ab.data(), (uInt)ab.size());
        const uint32_t got = crc_mulmod(crc_x8n((uint32_t)b.size(), x2n), ca) ^ cb;
        if (got != want) { printf("FAIL crc combine %08x != %08x\n", got, want); return 1; }
        uint32_t t = 0xffffffffu;
        for (uint8_t c : a) t = crc_table_entry((t ^ c) & 255u) ^ (t >> 8);
        if ((t ^ 0xffffffffu) != ca) { printf("FAIL crc table\n"); return 1; }
        for (uint32_t len = 3; len <= 258; len++) {
            const Sym s = length_symbol(len);
            static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
            static const int ext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
            if (s.sym < 257 || s.sym > 285 || (int)s.ebits != ext[s.sym - 257] || base[s.sym - 257] + (int)s.eval != (int)len) { printf("FAIL length symbol %u\n", len); return 1; }
        }
        for (uint32_t d = 1; d <= 32768; d++) {
            const Sym s = dist_symbol(d);
            static const int base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
            static const int ext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
            if (s.sym > 29 || (int)s.ebits != ext[s.sym] || base[s.sym] + (int)s.eval != (int)d) { printf("FAIL dist symbol %u\n", d); return 1; }
        }
        // Fibonacci frequencies force the length limit
        std::vector<uint32_t> A(40), bl(32);
        uint32_t f0 = 1, f1 = 1;
        for (auto &x : A) { x = f0; const uint32_t f2 = f0 + f1; f0 = f1; f1 = f2; }
        mr_code_lengths(A.data(), (int)A.size());
        if (A[0] <= 15) { printf("FAIL expected an over-long code\n"); return 1; }
        limit_code_lengths(A.data(), (int)A.size(), 15, bl.data());
        double kraft = 0;
        for (uint32_t l : A) { if (l < 1 || l > 15) { printf("FAIL limit\n"); return 1; } kraft += 1.0 / (double)(1u << l); }
        if (kraft > 1.0 + 1e-12) { printf("FAIL kraft %f\n", kraft); return 1; }
    }
    struct Case { std::string name; std::vector<uint8_t> data; };
    std::vector<Case> cases;
    cases.push_back({"bam-like, uniform qualities", bam_like((size_t)BLOCK * 6 + 1234, 1, false)});
    cases.push_back({"bam-like, run-heavy qualities", bam_like((size_t)BLOCK * 6 + 99, 2, true)});
    {
        std::mt19937 rng(3);
        std::vector<uint8_t> r((size_t)BLOCK * 2 + 5);
        for (auto &c : r) c = (uint8_t)rng();
        cases.push_back({"random bytes", r});
        std::vector<uint8_t> z((size_t)BLOCK + 77, 0);
        cases.push_back({"zeros", z});
        std::string txt;
        while (txt.size() < (size_t)BLOCK * 2) txt += "the quick brown fox jumps over the lazy dog " + std::to_string(rng() % 1000) + "\n";
        cases.push_back({"text", std::vector<uint8_t>(txt.begin(), txt.end())});
        std::vector<uint8_t> rep((size_t)BLOCK);
        for (size_t i = 0; i < rep.size(); i++) rep[i] = (uint8_t)((i % 7) * 31 + (i / 5000));
        cases.push_back({"short period", rep});
        std::vector<uint8_t> many((size_t)BLOCK);  // 4-byte repeats everywhere: floods the match list
        for (size_t i = 0; i < many.size(); i += 8) { uint32_t w = rng() % 512; memcpy(&many[i], &w, 4); memcpy(&many[i + 4], &w, std::min<size_t>(4, many.size() - i - 4)); }
        cases.push_back({"match flood", many});
        for (int n : {1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 258, 259, 1000})
            cases.push_back({"tiny " + std::to_string(n), std::vector<uint8_t>(r.begin(), r.begin() + n)});
        cases.push_back({"tiny zeros 300", std::vector<uint8_t>(300, 0)});
    }
    int fails = 0;
    struct Geom { const char *name; int block, buckets, cap; };
    const Geom geoms[2] = {{"0xff00-byte blocks", 0xff00, 512, (8192 + 2560) / 8}, {"0x7f00-byte blocks", 0x7f00, 384, 2560 / 8}};
    // extra payloads from files (experiments: python dumps of tests/test_gpu_bgzf.py's payloads)
    for (int k = 1; k < argc; k++) {
        FILE *f = fopen(argv[k], "rb");
        if (!f) continue;
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + got);
        fclose(f);
        cases.push_back({std::string("file ") + argv[k], d});
    }
    for (const Geom &g : geoms)
    for (const Case &c : cases) {
        BLOCK = g.block;
        N_BUCKETS = g.buckets;
        SEG_CAP = g.cap;
        N_SEG = 8;
        // (experiments: MODEL_SEG, MODEL_BUCKETS, MODEL_SEED, MODEL_SKIP_RUNS, MODEL_SEAM, MODEL_CAP override the device's constants)
        if (getenv("MODEL_SEG")) N_SEG = atoi(getenv("MODEL_SEG"));
        if (getenv("MODEL_BUCKETS")) N_BUCKETS = atoi(getenv("MODEL_BUCKETS"));
        if (getenv("MODEL_SEED")) SEED_PIECES = atoi(getenv("MODEL_SEED"));
        if (getenv("MODEL_SKIP_RUNS")) SKIP_RUNS = atoi(getenv("MODEL_SKIP_RUNS")) != 0;
        if (getenv("MODEL_SEAM")) SEAM = atoi(getenv("MODEL_SEAM")) != 0;
        if (getenv("MODEL_CAP")) SEG_CAP = atoi(getenv("MODEL_CAP"));
        if (getenv("MODEL_WAYS")) WAYS = atoi(getenv("MODEL_WAYS"));
        if (getenv("MODEL_NEAR")) NEAR = atoi(getenv("MODEL_NEAR"));
        if (getenv("MODEL_SHORT4")) SHORT4 = atoi(getenv("MODEL_SHORT4"));
        if (getenv("MODEL_SHORT5")) SHORT5 = atoi(getenv("MODEL_SHORT5"));
        if (getenv("MODEL_SHORT6")) SHORT6 = atoi(getenv("MODEL_SHORT6"));
        for (int lazy = 1; lazy < 2; lazy++) {
            size_t total = 0, z6 = 0, z1 = 0;
            bool ok = true;
            for (size_t o = 0; o < c.data.size(); o += (size_t)BLOCK) {
                const int n = (int)std::min<size_t>((size_t)BLOCK, c.data.size() - o);
                const std::vector<uint8_t> z = model_deflate(c.data.data() + o, n, lazy != 0);
                // MODEL_DUMP=path: the 0x7f00 geometry's raw DEFLATE streams of the payload files, each behind its length
                // (u32): what tests/test_gpu_bgzf.py holds the device's members to, byte for byte
                if (getenv("MODEL_DUMP") && g.block == 0x7f00 && c.name.rfind("file ", 0) == 0) {
                    FILE *df = fopen(getenv("MODEL_DUMP"), "ab");
                    if (df) {
                        const uint32_t zl = (uint32_t)z.size();
                        fwrite(&zl, 4, 1, df);
                        fwrite(z.data(), 1, z.size(), df);
                        fclose(df);
                    }
                }
                if (!inflate_ok(z, c.data.data() + o, n)) { ok = false; printf("FAIL %s block at %zu (n = %d)\n", c.name.c_str(), o, n); }
                if (z.size() > 65510) { ok = false; printf("FAIL %s: %zu bytes do not fit a BGZF block\n", c.name.c_str(), z.size()); }
                total += z.size() + 26;
            }
            for (size_t o = 0; o < c.data.size(); o += 0xff00) {  // the comparator: zlib over htslib's blocks
                const int n = (int)std::min<size_t>(0xff00, c.data.size() - o);
                z6 += zlib_size(c.data.data() + o, n, 6) + 26;
                z1 += zlib_size(c.data.data() + o, n, 1) + 26;
            }
            if (!ok) fails++;
            if (c.data.size() > 2000)
                printf("%s %-36s %9zu -> %9zu (%.4f)   zlib -6 %.4f  -1 %.4f %s\n", g.name, c.name.c_str(), c.data.size(), total,
                       (double)total / (double)c.data.size(), (double)z6 / (double)c.data.size(), (double)z1 / (double)c.data.size(), ok ? "" : "  <-- FAIL");
        }
    }
    if (fails) { printf("%d case(s) failed\n", fails); return 1; }
    printf("gpu deflate model: all streams inflate to their input\n");
    return 0;
}
