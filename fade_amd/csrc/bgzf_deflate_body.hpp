// bgzf_deflate_body.hpp — the compressor's kernels for one block geometry (FADEHIP_BGZF_GEOM = 64 or 32, namespace
// FADEHIP_BGZF_NS): included twice by bgzf_deflate.hpp.  No include guard on purpose.

namespace fadehip {
namespace FADEHIP_BGZF_NS {
using namespace ::fadehip::bgzf;  // bgzf_huff.hpp's serial helpers

#if FADEHIP_BGZF_GEOM == 64
// htslib's block size; the block takes the CU's whole LDS: one workgroup of eight waves per CU
constexpr int BLOCK = 0xff00, DATA_BYTES = 65536, WG = 512, MIN_WAVES_PER_SIMD = 2, N_BUCKETS = 512, MAX_MATCHES = 8192 + 2560;
#else
// half of it: 80 KB of LDS, TWO workgroups of eight waves per CU at 128 VGPRs: all of a SIMD's four wave slots.  (Six
// waves per workgroup — a slot per SIMD left to the file path's other kernels, as round 3 had it — were measured too: the
// compressor alone 18.1 instead of 19.7 GB/s, and the kernels that then run beside it crawl — pack_write 0.97 ms instead
// of 0.03, rewrite 1.35 instead of 0.23 — so that a call's device time is the same either way; tools/r04/stream_timeline.py.)
constexpr int BLOCK = 0x7f00, DATA_BYTES = 32768, WG = 512, MIN_WAVES_PER_SIMD = 4, N_BUCKETS = 384, MAX_MATCHES = 2560;
#endif
constexpr int DIST_T0 = 320;  // threads DIST_T0 .. + 29 serve the distance alphabet where the first 286 serve literals / lengths
constexpr int N_WAVES = WG / 64;
constexpr int WAYS = 4;
// phase A runs on segments, a wave each with a hash table of its own: N_BUCKETS buckets of WAYS 16-bit positions (what a
// wave can reach back to: the last N_BUCKETS * WAYS positions on average)
constexpr int HEAD_WAVE_BYTES = N_BUCKETS * WAYS * 2;
constexpr int HEAD_BYTES = HEAD_WAVE_BYTES * N_WAVES;
constexpr int SEG_CAP = MAX_MATCHES / N_WAVES;  // match records a segment may leave (more: literals from there on)
constexpr int NEAR = 2;                         // distances tried directly (the table does not hold a piece's own positions yet)
constexpr int SEED_PIECES = 16;                 // pieces in front of a segment whose positions are hashed for it (1 KB of history)
constexpr int MIN_MATCH = 4, MAX_MATCH = 258;
constexpr int N_WORDS = (BLOCK + 31) / 32;   // words of a per-position bitmap
constexpr int WPT = (N_WORDS + WG - 1) / WG;  // ... per thread in the per-range phases (a range = 64 positions)
constexpr int SLOT = 65536;                  // bytes of a block's output slot (payload <= 65510: BSIZE is 16 bits)
constexpr int MAX_PAYLOAD = 65536 - 26;

// LDS layout (bytes)
constexpr int BITMAP_BYTES = ((N_WORDS * 4 + 255) / 256) * 256;
constexpr int L_DATA = 0, L_HEAD = DATA_BYTES, L_MATCH = L_HEAD + HEAD_BYTES, L_TOK = L_MATCH + MAX_MATCHES * 4, L_MAT = L_TOK + BITMAP_BYTES,
              L_MISC = L_MAT + BITMAP_BYTES, LDS_BYTES = L_MISC + 6144;
static_assert(LDS_BYTES <= (FADEHIP_BGZF_GEOM == 64 ? 160 : 80) * 1024, "the workgroups of a CU must fit its LDS");
static_assert(BLOCK % 16 == 0 && BLOCK + 256 <= DATA_BYTES && BLOCK < 65535, "block size");
// ... of the head region once the matches are found; the CRC tables go where the match records were once they are emitted
constexpr int H_H8 = 0, H_AL = H_H8 + 8 * 320 * 4, H_SL = H_AL + 320 * 4, H_AD = H_SL + 320 * 4, H_SD = H_AD + 64 * 4, H_END = H_SD + 64 * 4;
static_assert(H_END <= HEAD_BYTES, "phase B temporaries must fit the hash region");

struct Misc {  // the small arrays of a block
    uint32_t abort, dbg[3], carry, mcount, full, blk, m_l, m_d, hdr_bits, total_bits, stored, pad[3];
    uint32_t seg_mcount[N_WAVES], seg_prefix[N_WAVES + 1];  // match records per segment of phase A; ... of the segments in front
    alignas(8) uint32_t freq_l[320];  // (read two at a time where the symbols are ranked)
    uint32_t freq_d[64];
    uint8_t ll[320], dl[64];
    uint16_t lc[320], dc[64];
    uint32_t hdr[160];
    uint32_t bl_l[16], bl_d[16], nc_l[16], nc_d[16];
    uint32_t x2n[32];
    uint32_t sortbuf[64];
    uint32_t wtmp[N_WAVES];
    uint32_t crc_part[N_WAVES];
    // the header's run-length tokens (symbol | extra << 8) and the code-length alphabet
    uint16_t cltok[320];
    uint32_t cl_freq[NUM_CL + 1], cl_len[NUM_CL + 1], cl_code[NUM_CL + 1], cl_n, cl_hlit, cl_hdist, cl_hclen;
};
static_assert(sizeof(Misc) <= 6144, "Misc outgrew its slice");

struct DeflateArgs {
    const uint8_t *src;   // the byte stream (device)
    uint64_t n_bytes;
    uint32_t n_blocks;
    uint8_t *slots;       // [n_blocks][SLOT]
    uint32_t *out_size;   // [n_blocks] payload bytes
    uint32_t *out_crc;    // [n_blocks]
    uint32_t *ticket;     // blocks are drawn from here
    unsigned long long *prof;  // optional [8]: shader clocks per phase, summed over blocks by lane 0 (FADEHIP_BGZF_PROF)
};

__device__ __forceinline__ uint32_t lds_load32u(const uint8_t *base, uint32_t p) {  // 4 bytes at any offset
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base) + (p >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], p & 3u);
}
__device__ __forceinline__ uint64_t lds_load64u(const uint8_t *base, uint32_t p) {  // 8 bytes at any offset
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base) + (p >> 2);
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, p & 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, p & 3u) << 32);
}
__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (((v * 0x9E3779B1u) >> 16) * (uint32_t)N_BUCKETS) >> 16; }

// the next ticket of a wave-shared counter, as a wave-uniform value (kept out of line: inlined into a loop the claim was
// once hoisted around the loop's exec-mask bookkeeping and a back edge re-used a stale ticket)
__device__ __noinline__ int claim_ticket(uint32_t *counter) {
    uint32_t tk = 0;
    if ((threadIdx.x & 63) == 0) tk = atomicAdd(counter, 1u);
    return __builtin_amdgcn_readlane((int)tk, 0);
}
// exclusive scan of one value per thread over the workgroup (tmp: N_WAVES words of LDS); *total = the sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *tmp, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();  // tmp may still be read from an earlier scan
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < N_WAVES; w++) {
        const uint32_t t = tmp[w];
        if (w < wave) base += t;
        sum += t;
    }
    *total = sum;
    return base + inc - v;
}

// An array of up to 64 NR entries spread over the lanes of a wavefront (entry i in lane i % 64 of register i / 64), read
// and written with v_readlane / v_writelane by code the whole wave runs in lockstep on wave-uniform indices: the accessor
// the serial Huffman routines of bgzf_huff.hpp take on the device (a dependent LDS round trip costs ~130 clocks, a lane
// access ~10, and those routines are chains of dependent accesses).
template <int NR>
struct WaveArr {
    uint32_t r[NR];
    __device__ __forceinline__ uint32_t get(int i) const {  // i is wave-uniform
        const int k = i >> 6, l = i & 63;
        uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)r[0], l);
#pragma unroll
        for (int j = 1; j < NR; j++)
            if (k == j) v = (uint32_t)__builtin_amdgcn_readlane((int)r[j], l);
        return v;
    }
    __device__ __forceinline__ void set(int i, uint32_t v) {  // i and v are wave-uniform
        const int k = i >> 6, l = i & 63;
#pragma unroll
        for (int j = 0; j < NR; j++)
            if (k == j) asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(r[j]) : "s"(v), "s"(l) : "m0");  // (one SGPR per VALU instruction)
    }
};
// bit sink of the header: whole words to LDS by lane 0, the accumulator wave-uniform
struct LdsSink {
    uint32_t *w;
    uint64_t acc = 0;
    int cnt = 0;
    uint32_t wi = 0;
    __device__ __forceinline__ void put(uint32_t v, int n) {
        acc |= (uint64_t)v << cnt;
        cnt += n;
        if (cnt >= 32) {
            if ((threadIdx.x & 63) == 0) w[wi] = (uint32_t)acc;
            wi++;
            acc >>= 32;
            cnt -= 32;
        }
    }
    __device__ __forceinline__ uint32_t finish() {
        if (cnt && (threadIdx.x & 63) == 0) w[wi] = (uint32_t)acc;
        return 32u * wi + (uint32_t)cnt;
    }
};

// bits of the token that starts at bit b of bitmap word w (a literal, or the match whose record the match bitmap counts to)
__device__ __forceinline__ void token_bits(const uint8_t *data, uint32_t mw, uint32_t mbase, const uint32_t *match, const Misc *ms, int w, int b,
                                           uint64_t &bits, int &nb) {
    if ((mw >> b) & 1u) {
        const uint32_t rec = match[mbase + (uint32_t)__builtin_popcount(mw & ((1u << b) - 1u))];
        const Sym ls = length_symbol((rec >> 16) + 3u), ds = dist_symbol(rec & 0xffffu);
        uint64_t v = ms->lc[ls.sym];
        int k = ms->ll[ls.sym];
        v |= (uint64_t)ls.eval << k;
        k += (int)ls.ebits;
        v |= (uint64_t)ms->dc[ds.sym] << k;
        k += ms->dl[ds.sym];
        v |= (uint64_t)ds.eval << k;
        k += (int)ds.ebits;
        bits = v;
        nb = k;
    } else {
        const uint32_t c = data[32 * w + b];
        bits = ms->lc[c];
        nb = ms->ll[c];
    }
}

// CRC-32 (reflected, P = 0xEDB88320) advanced by one 32-bit word WITHOUT a table: bit k of (state ^ word) alone becomes the
// constant K[k] after 32 shifts, so the new state is the XOR of the K[k] of the set bits — 32 x (bit-field extract, and,
// xor) in registers.  The tables of slicing-by-4 cost four LDS look-ups a word, and the CRC is summed while one lane makes
// the code lengths out of the same LDS: its chain of dependent accesses waited behind them.
struct CrcBitK {
    uint32_t k[32];
    constexpr CrcBitK() : k() {
        for (int b = 0; b < 32; b++) {
            uint32_t c = 1u << b;
            for (int i = 0; i < 32; i++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            k[b] = c;
        }
    }
};
__device__ __forceinline__ uint32_t crc_word(uint32_t c) {
    constexpr CrcBitK K;
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) r ^= K.k[b] & (uint32_t)__builtin_amdgcn_sbfe((int)c, (uint32_t)b, 1u);
    return r;
}
__device__ __forceinline__ uint32_t crc_byte(uint32_t c, uint32_t byte) {
    c ^= byte;
#pragma unroll
    for (int i = 0; i < 8; i++) c = (c >> 1) ^ (0xEDB88320u & (uint32_t)__builtin_amdgcn_sbfe((int)c, 0u, 1u));
    return c;
}

// An array of up to 64 words held by a wave in ONE VGPR: element k is lane k's, read and written with a uniform index
// (v_readlane; a write is a compare of the lane number and a select — this compiler has no builtin for v_writelane: a few
// clocks, no LDS).  For bgzf_huff.hpp's templates over small alphabets; every lane of the wave must run the code (uniform
// control flow).
struct LaneArr {
    uint32_t v;
    int lane;
    __device__ __forceinline__ uint32_t get(int i) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, i); }
    __device__ __forceinline__ void set(int i, uint32_t x) { v = lane == i ? x : v; }
};

// Minimum-redundancy code lengths (Moffat & Katajainen) by a WHOLE WAVE: the same lengths as bgzf_huff.hpp's
// mr_code_lengths, which is a chain of ~3 m dependent LDS accesses for one lane (the longest serial stretch of a block once
// phase A stopped being one).  Only its first pass is a chain by nature — the merge of the sorted leaves with the internal
// nodes it makes — and stays with lane 0; it leaves each internal node's parent in A[0 .. m-2).  Then: the internal nodes'
// depths by pointer jumping (ceil(log2 m) rounds of dep[i] += dep[par[i]], par[i] = par[par[i]] over all nodes at once, instead of
// m dependent double look-ups), a histogram of those depths, the number of leaves at each depth from it (a level holds twice the
// internal nodes of the level above; what is not an internal node is a leaf), and every leaf reads its depth off the running
// sums by its position.  A[0 .. m) = frequencies ascending on entry, lengths (descending) on return; scratch: 4 arrays of m + 2.
__device__ __forceinline__ void mr_code_lengths_wave(uint32_t *A, const int m, uint32_t *dep, uint32_t *par, uint32_t *cnt, uint32_t *cum, const int lane,
                                                     unsigned long long *prof = nullptr) {
    unsigned long long tp = prof ? __builtin_readcyclecounter() : 0ull;
    auto sub = [&](int k) {
        if (prof && lane == 0) {
            const unsigned long long t = __builtin_readcyclecounter();
            atomicAdd(&prof[k], t - tp);
            tp = t;
        }
    };
    if (m == 1) {
        if (lane == 0) A[0] = 1;
        return;
    }
    {
        // (One lane with its values in vector registers took 390 clocks a node; holding the next four nodes and leaves in
        // registers, loaded three picks ahead, 485: the chain is bound by the instructions it issues as much as by the LDS.)
        // the merge, run by the whole wave on UNIFORM values (read through v_readfirstlane): its compares, selects and counters
        // are then the scalar unit's, whose dependent operations follow each other faster than one lane's vector operations
        // The fronts of the two queues (ar, al) are scalars; the places behind them (A[root + 1], A[leaf + 1]) are loads in flight,
        // in vector registers, turned into scalars only when their turn comes — a pick then does not wait for the LDS.  A
        // node is written after the place behind root may have been read ahead: its value is put into that register too.
        auto rfl = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
        constexpr uint32_t INF = 0xffffffffu;
        const uint32_t a01 = rfl(A[0]) + rfl(A[1]);
        A[0] = a01;
        int root = 0, leaf = 2;
        uint32_t ar = a01, al = leaf < m ? rfl(A[2]) : INF;
        uint32_t pr = A[1], pl = A[3];  // (in flight)
        for (int next = 1; next < m - 1; next++) {
            uint32_t v;
            if (ar < al) {  // (leaf >= m: al is the maximum)
                v = ar;
                A[root++] = (uint32_t)next;
                ar = rfl(pr);
                pr = A[root + 1];
            } else {
                v = al;
                leaf++;
                al = leaf < m ? rfl(pl) : INF;
                pl = A[leaf + 1];
            }
            if (leaf >= m || (root < next && ar < al)) {
                v += ar;
                A[root++] = (uint32_t)next;
                ar = rfl(pr);
                pr = A[root + 1];
            } else {
                v += al;
                leaf++;
                al = leaf < m ? rfl(pl) : INF;
                pl = A[leaf + 1];
            }
            A[next] = v;
            if (root == next) ar = v;           // the node just made is the internal queue's front,
            else if (root + 1 == next) pr = v;  // or the place behind it
        }
    }
    __builtin_amdgcn_wave_barrier();
    sub(41);
    const int ni = m - 1;  // internal nodes 0 .. m-2, the root last
    constexpr int PER = (NUM_LITLEN + 2 + 63) / 64;
    for (int i = lane; i < ni; i += 64) {
        par[i] = i == ni - 1 ? (uint32_t)i : A[i];
        dep[i] = i == ni - 1 ? 0u : 1u;
    }
    for (int i = lane; i < m + 2; i += 64) cnt[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; (1 << r) < ni; r++) {
        uint32_t nd[PER], np[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = lane + 64 * k;
            nd[k] = np[k] = 0;
            if (i < ni) {
                const uint32_t q = par[i];
                nd[k] = dep[i] + dep[q];
                np[k] = par[q];
            }
        }
        __builtin_amdgcn_wave_barrier();  // (a wave's LDS accesses complete in order: every read of the round is ahead of its writes)
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = lane + 64 * k;
            if (i < ni) { dep[i] = nd[k]; par[i] = np[k]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    sub(42);
    for (int i = lane; i < ni; i += 64) atomicAdd(&cnt[dep[i]], 1u);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {  // leaves per depth, as running sums: cum[d] = leaves at depths 0 .. d
        uint32_t avbl = 1, run = 0;
        int d = 0;
        while (avbl > 0 && d <= m) {
            const uint32_t used = cnt[d];
            run += avbl - used;
            cum[d] = run;
            avbl = 2u * used;
            d++;
        }
        cum[m + 1] = (uint32_t)d;  // levels
    }
    __builtin_amdgcn_wave_barrier();
    sub(43);
    const int levels = (int)cum[m + 1];
    // the q-th most frequent leaf sits at the first depth whose running sum exceeds q: the sums do not decrease, so that depth is
    // the number of sums <= q among the first levels - 1 — counted against the sums held in a register (a lane each, read with a
    // uniform index) instead of walked in LDS by every lane for itself (12.7 k -> 3 k clocks a block)
    if (levels <= 64) {
        const uint32_t cumreg = lane < levels ? cum[lane] : 0u;
        uint32_t dq[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) dq[k] = 0;
        for (int d = 0; d < levels - 1; d++) {
            const uint32_t cd = (uint32_t)__builtin_amdgcn_readlane((int)cumreg, d);
#pragma unroll
            for (int k = 0; k < PER; k++) dq[k] += cd <= (uint32_t)(lane + 64 * k);
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int q = lane + 64 * k;
            if (q < m) A[m - 1 - q] = dq[k];
        }
    } else {
        for (int q = lane; q < m; q += 64) {
            int d = 0;
            while (d < levels - 1 && cum[d] <= (uint32_t)q) d++;
            A[m - 1 - q] = (uint32_t)d;
        }
    }
    __builtin_amdgcn_wave_barrier();
    sub(44);
}

// ---- phase A: matches and the parse of ONE SEGMENT of the block, by one wave that needs nobody else.
// Round 3 ran phase A as a pipeline of wave roles (a hasher, extenders, ONE parser, rings with sequence numbers between
// them): a block's time was the parser's, the other waves waited a third to two thirds of theirs, and a quarter of the
// kernel's instructions were polls.  Now the block's pieces (64 positions each) are cut into N_WAVES contiguous segments
// and every wave does all three steps for its own: it hashes a piece's positions into a table OF ITS OWN (the 512 bytes
// in front of the segment first, so that the segment's first record finds the one before it), extends the candidates of
// the positions no earlier match covers — it knows where its parse stands, the extenders of the pipeline could not —,
// parses the piece on lane masks, and goes on.  No wait, no ring, no poll; the waves meet at the barrier behind phase A.
// What it costs: a match cannot cross a segment's end (it is cut there, so that the next segment's parse starts at its
// first position), and a segment's table sees only its own positions — in BAM payload matches reach one or two records
// back, a few hundred bytes.
struct SegArgs {
    uint8_t *data;
    uint16_t *head;    // this wave's table
    uint32_t *match;   // this wave's match records [SEG_CAP]
    uint32_t *tokw, *matw;
    int n, first_piece, end_piece, lane;
    unsigned long long *prof;  // FADEHIP_BGZF_PROF: the first wave's clocks by part of a piece (nullptr otherwise)
};
// what a segment's wave leaves for the seam behind it: the record of its last match if that match was cut at the segment's
// end, and by how many bytes it would have gone on (phase_a_seam lengthens it again as far as the next segment's parse allows)
struct SegOut {
    uint32_t mcount, over_rec, over;
};
__device__ __forceinline__ SegOut phase_a_segment(const SegArgs r) {
    uint8_t *const data = r.data;
    const int n = r.n, lane = r.lane;
    const int seg_end = min(n, r.end_piece * 64);  // first position behind the segment
    // history: the positions of the pieces just in front of the segment (another wave's) go into this wave's table
    for (int piece = max(0, r.first_piece - SEED_PIECES); piece < r.first_piece; piece++) {
        const uint32_t p = (uint32_t)piece * 64u + (uint32_t)lane;
        if ((int)p + MIN_MATCH <= n) {
            uint2 *bucket = reinterpret_cast<uint2 *>(r.head) + hash4(lds_load32u(data, p));
            const uint2 bk = *bucket;
            *bucket = make_uint2((p + 1u) | (bk.x << 16), (bk.x >> 16) | (bk.y << 16));
        }
    }
    int carry = r.first_piece * 64;  // positions below it are covered by a match already taken
    uint32_t mcount = 0, full = 0, over_rec = 0, over = 0;
    // (the parts of a piece, clocked: compiled in with -DFADEHIP_BGZF_PROF_A=1 only — seven more branches a piece are scalar
    // instructions, and the CU's one scalar unit is what sixteen waves of this phase queue for)
#ifndef FADEHIP_BGZF_PROF_A
#define FADEHIP_BGZF_PROF_A 0
#endif
    unsigned long long tq = (FADEHIP_BGZF_PROF_A && r.prof) ? __builtin_readcyclecounter() : 0ull, tpart[7] = {0, 0, 0, 0, 0, 0, 0};
    auto part = [&](int k) {
        if (FADEHIP_BGZF_PROF_A && r.prof) {
            const unsigned long long t = __builtin_readcyclecounter();
            tpart[k - 56] += t - tq;
            tq = t;
        }
    };
    for (int piece = r.first_piece; piece < r.end_piece; piece++) {
        const uint32_t p = (uint32_t)piece * 64u + (uint32_t)lane;
        const bool valid = (int)p + MIN_MATCH <= n;
        const uint32_t v = valid ? lds_load32u(data, p) : 0u;
        uint2 bk = make_uint2(0, 0);
        if (valid) {
            uint2 *bucket = reinterpret_cast<uint2 *>(r.head) + hash4(v);
            bk = *bucket;
            // (leaving the positions inside runs of one byte out of the table was modelled — host/selftest/gpu_deflate_model.cpp,
            // MODEL_SKIP_RUNS — and made the output 0.3 % larger on run-heavy qualities, not smaller)
            *bucket = make_uint2((p + 1u) | (bk.x << 16), (bk.x >> 16) | (bk.y << 16));  // newest first; the oldest of the four leaves
        }
        uint32_t len = 0, dist = 0;
        part(56);
        if (valid && (int)p >= carry) {
            const uint32_t maxlen = (uint32_t)min(MAX_MATCH, n - (int)p);
            // up to five candidates: the nearer of the distances 1 and 2 whose four bytes agree (what a piece's own positions,
            // not yet in the table, would offer: runs and 16-bit patterns; the distances 3 .. 8 of the role pipeline bring
            // nothing on the payloads this geometry is taken for — gpu_deflate_model MODEL_NEAR — and cost twelve LDS reads a
            // position), and the bucket's four
            // positions; their loads are issued together and they are extended side by side (one LDS round trip per
            // eight bytes of the LONGEST match, not per candidate)
            uint32_t cp[5];
            uint32_t alive = 0;
            {
                uint32_t vd[NEAR];
#pragma unroll
                for (uint32_t d = 1; d <= (uint32_t)NEAR; d++) vd[d - 1] = lds_load32u(data, p >= d ? p - d : p);
                uint32_t dsmall = 0;
#pragma unroll
                for (uint32_t d = NEAR; d >= 1u; d--)
                    if (p >= d && vd[d - 1] == v) dsmall = d;
                cp[0] = p - dsmall;
                if (dsmall) alive |= 1u;
            }
            const uint32_t c4[4] = {bk.x & 0xffffu, bk.x >> 16, bk.y & 0xffffu, bk.y >> 16};
            uint32_t cv[4];
#pragma unroll
            for (int w = 0; w < WAYS; w++) {
                cp[1 + w] = c4[w] ? c4[w] - 1u : 0u;
                cv[w] = lds_load32u(data, cp[1 + w]);
            }
#pragma unroll
            for (int w = 0; w < WAYS; w++)
                if (c4[w] && p - cp[1 + w] <= 32768u && cv[w] == v) alive |= 2u << w;
            uint32_t cl[5] = {0, 0, 0, 0, 0};
            uint32_t off = 4;
            part(57);
            while (alive && off < maxlen) {  // eight bytes per round trip
                const uint64_t pw = lds_load64u(data, p + off);
                uint64_t x[5];
#pragma unroll
                for (int k = 0; k < 5; k++) x[k] = lds_load64u(data, cp[k] + off) ^ pw;
#pragma unroll
                for (int k = 0; k < 5; k++)
                    if (((alive >> k) & 1u) && x[k]) {
                        cl[k] = off + ((uint32_t)__builtin_ctzll(x[k]) >> 3);
                        alive &= ~(1u << k);
                    }
                off += 8;
            }
            part(58);
            uint32_t best_cp = p;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                // (still equal where the limit was reached: the longest there is; a compare may have read past the limit)
                const uint32_t l = min(((alive >> k) & 1u) ? maxlen : cl[k], maxlen);
                if (l > len) { len = l; best_cp = cp[k]; }
            }
            dist = p - best_cp;
        }
        // a match ends with its segment (the next segment's parse starts at that segment's first position); what it would have
        // had beyond is remembered for the seam
        const uint32_t ulen = len;
        len = min(len, (uint32_t)max(seg_end - (int)p, 0));
        if (len < (uint32_t)MIN_MATCH) len = 0;
        // ---- the parse of this piece, on 64-bit lane masks: a match yields to a longer one at the next position
        const uint32_t len_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)len, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
        const bool yield = len && lane < 63 && len_next > len;
        const int cb = piece * 64, nv = min(64, n - cb);
        uint64_t has = __ballot(len != 0 && !yield);
        // the segment's match list is full: literals from here on (a piece adds 64 at most: counted only near the end)
        if (full || (mcount > (uint32_t)(SEG_CAP - 64) && mcount + (uint32_t)__popcll(has) > (uint32_t)SEG_CAP)) { full = 1; has = 0; }
        const uint64_t vmask = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
        // greedy: from `cur`, the next match start at or after it is taken and covers its length; what no match covers is a literal
        part(60);
        // (every operation here is the scalar unit's, which the CU's sixteen waves share: as few as the greedy walk needs)
        uint64_t matmask = 0, covered = 0;
        int cur = max(carry - cb, 0);
        const int cur0 = cur;
        int j_last = -1;
        const uint64_t from0 = cur0 >= 64 ? 0ull : (~0ull << cur0);  // positions at or behind cur0
        uint64_t rem = has & from0;  // match starts still ahead of the walk (has holds none beyond the block's end)
        while (rem) {
            const int j = (int)__builtin_ctzll(rem);
            const int e = j + (int)__builtin_amdgcn_readlane((int)len, j);  // first position after the match
            const uint64_t above_j = 2ull << j;                            // (0 for j = 63: no position inside the match)
            const uint64_t at_e = e >= 64 ? 0ull : (1ull << e);
            matmask |= 1ull << j;
            covered |= at_e - above_j;  // positions j + 1 .. e - 1 (e >= 64: everything above j)
            rem &= ~(at_e - 1ull);      // the starts the match covers are gone (e >= 64: all of them)
            cur = e;
            j_last = j;
        }
        part(61);
        if (j_last >= 0 && cb + cur == seg_end) {  // the piece's last match reaches the segment's end: was it cut there?
            const uint32_t cut = (uint32_t)__builtin_amdgcn_readlane((int)(ulen - len), j_last);
            over = cut;
            over_rec = mcount + (uint32_t)__popcll(matmask & ((1ull << j_last) - 1ull));
        }
        const uint64_t tokmask = vmask & ~covered & from0;
        if (cur < nv) cur = nv;
        if (cb + cur > carry) carry = cb + cur;
        if ((matmask >> lane) & 1ull) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(matmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)matmask, 0u));
            r.match[mcount + before] = dist | ((len - 3u) << 16);
        }
        mcount += (uint32_t)__popcll(matmask);
        if (lane == 0) {
            *reinterpret_cast<uint2 *>(r.tokw + 2 * piece) = make_uint2((uint32_t)tokmask, (uint32_t)(tokmask >> 32));
            *reinterpret_cast<uint2 *>(r.matw + 2 * piece) = make_uint2((uint32_t)matmask, (uint32_t)(matmask >> 32));
        }
        part(62);
    }
    if (FADEHIP_BGZF_PROF_A && r.prof && lane == 0)
        for (int k = 0; k < 7; k++) atomicAdd(&r.prof[56 + k], tpart[k]);
    SegOut o;
    o.mcount = mcount;
    o.over_rec = over_rec;
    o.over = over;
    return o;
}
// The seam behind a segment whose last match was cut at its end by `over` bytes: the match is given back as many of them
// as the next segment's parse allows — up to its first match start inside that stretch; the literals it then covers
// leave the token bitmap.  One lane; runs between the barrier behind phase A and the histograms.
__device__ __forceinline__ void phase_a_seam(uint32_t *match_rec, uint32_t over, int seg_end, int next_end, uint32_t *tokw, const uint32_t *matw) {
    int give = min((int)over, next_end - seg_end);  // (within the next segment: its bitmap words are this seam's alone)
    for (int q = seg_end; q < seg_end + give; q++)
        if ((matw[q >> 5] >> (q & 31)) & 1u) { give = q - seg_end; break; }
    if (give <= 0) return;
    for (int q = seg_end; q < seg_end + give; q++) tokw[q >> 5] &= ~(1u << (q & 31));
    *match_rec += (uint32_t)give << 16;
}

__global__ __launch_bounds__(WG, MIN_WAVES_PER_SIMD) void bgzf_deflate_kernel(DeflateArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint8_t *const data = lds + L_DATA;
    uint16_t *const head = reinterpret_cast<uint16_t *>(lds + L_HEAD);
    uint32_t *const match = reinterpret_cast<uint32_t *>(lds + L_MATCH);
    uint32_t *const tokw = reinterpret_cast<uint32_t *>(lds + L_TOK);
    uint32_t *const matw = reinterpret_cast<uint32_t *>(lds + L_MAT);
    Misc *const ms = reinterpret_cast<Misc *>(lds + L_MISC);
    uint32_t *const h8 = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_H8);
    uint32_t *const A_l = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_AL), *const S_l = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_SL);
    uint32_t *const A_d = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_AD), *const S_d = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_SD);
    // (the CRC is summed without tables — crc_word — by the waves that would otherwise idle while the code lengths are made)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) crc_x2n_table(ms->x2n);
    for (;;) {
        __syncthreads();  // the previous block's LDS is no longer read
        if (tid == 0) ms->blk = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const uint32_t blk = ms->blk;
        if (blk >= a.n_blocks) break;  // (uniform: every thread reads the same word)
        unsigned long long t_prev = a.prof ? __builtin_readcyclecounter() : 0ull;
        auto stamp = [&](int k) {
            if (a.prof && tid == 0) {
                const unsigned long long t = __builtin_readcyclecounter();
                atomicAdd(&a.prof[k], t - t_prev);
                t_prev = t;
            }
        };
        const uint64_t off = (uint64_t)blk * BLOCK;
        const int n = (int)(a.n_bytes - off < (uint64_t)BLOCK ? a.n_bytes - off : (uint64_t)BLOCK);
        const uint8_t *src = a.src + off;
        uint8_t *const out = a.slots + (uint64_t)blk * SLOT;
        uint32_t *const out32 = reinterpret_cast<uint32_t *>(out);

        // ---- load the block (the stream starts 16-byte aligned and BLOCK is a multiple of 16), clear the tables
        {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
            uint4 *d4 = reinterpret_cast<uint4 *>(data);
            const int n16 = n >> 4;
            for (int k = tid; k < n16; k += WG) d4[k] = s4[k];
            for (int k = (n16 << 4) + tid; k < n; k += WG) data[k] = src[k];
            for (int k = n + tid; k < ((n + 15) & ~15) + 272 && k < DATA_BYTES; k += WG) data[k] = 0;  // what a compare may read past the end
            uint4 *z = reinterpret_cast<uint4 *>(lds + L_HEAD);
            for (int k = tid; k < HEAD_BYTES / 16; k += WG) z[k] = make_uint4(0, 0, 0, 0);
            uint4 *zb = reinterpret_cast<uint4 *>(lds + L_TOK);
            for (int k = tid; k < 2 * BITMAP_BYTES / 16; k += WG) zb[k] = make_uint4(0, 0, 0, 0);
            if (tid == 0) { ms->carry = 0; ms->mcount = 0; ms->stored = 0; ms->abort = 0; }
        }
        __syncthreads();
        stamp(0);

        // ---- A: matches and the parse: the block's pieces of 64 positions in N_WAVES contiguous segments, a wave each
        // (phase_a_segment: no wave waits for another)
        const int n_pieces = (n + 63) >> 6;
        const int seg_pieces = (n_pieces + N_WAVES - 1) / N_WAVES;
        {
            SegArgs sa;
            sa.data = data;
            sa.head = head + (size_t)wave * (HEAD_WAVE_BYTES / 2);
            sa.match = match + (size_t)wave * SEG_CAP;
            sa.tokw = tokw;
            sa.matw = matw;
            sa.n = n;
            sa.first_piece = min(wave * seg_pieces, n_pieces);
            sa.end_piece = min((wave + 1) * seg_pieces, n_pieces);
            sa.lane = lane;
            sa.prof = (a.prof && wave == 0) ? a.prof : nullptr;
            const SegOut so = phase_a_segment(sa);
            if (lane == 0) ms->seg_mcount[wave] = so.mcount;
            __syncthreads();
            // the seams: a match cut at its segment's end goes on into the next segment as far as that one's parse lets it
            if (lane == 0 && so.over && wave + 1 < N_WAVES && sa.end_piece < n_pieces)
                phase_a_seam(sa.match + so.over_rec, so.over, sa.end_piece * 64, min(n, (sa.end_piece + seg_pieces) * 64), tokw, matw);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int w = 0; w < N_WAVES; w++) { ms->seg_prefix[w] = run; run += ms->seg_mcount[w]; }
            ms->seg_prefix[N_WAVES] = run;
            ms->mcount = run;
        }
        stamp(1);

        // ---- B: match-index prefix, histograms
        {
            uint32_t *z = reinterpret_cast<uint32_t *>(lds + L_HEAD);
            for (int k = tid; k < H_END / 4; k += WG) z[k] = 0;
            if (tid < 16) { ms->bl_l[tid] = 0; ms->bl_d[tid] = 0; }
        }
        __syncthreads();
        const int w0 = WPT * tid, w1 = min(w0 + WPT, N_WORDS);
        uint32_t tw_r[WPT], mw_r[WPT], mb_r[WPT];  // this range's bitmap words and the match index each word starts at
        {
            uint32_t cnt = 0;
#pragma unroll
            for (int k = 0; k < WPT; k++) {
                const bool in = w0 + k < w1;
                tw_r[k] = in ? tokw[w0 + k] : 0u;
                mw_r[k] = in ? matw[w0 + k] : 0u;
                cnt += (uint32_t)__builtin_popcount(mw_r[k]);
            }
            uint32_t all;
            uint32_t at = block_excl_scan(cnt, ms->wtmp, &all);  // (its barriers also publish seg_prefix)
            // a word's matches lie in its segment's region of the match records: at the word's rank among the segment's matches
#pragma unroll
            for (int k = 0; k < WPT; k++) {
                const int sg = min(((w0 + k) >> 1) / seg_pieces, N_WAVES - 1);
                mb_r[k] = (uint32_t)sg * (uint32_t)SEG_CAP + (at - ms->seg_prefix[sg]);
                at += (uint32_t)__builtin_popcount(mw_r[k]);
            }
        }
        {
            // The literals of a word without a loop over its tokens: the word's 32 bytes come as two 16-byte loads and every
            // position adds its bit of the literal bitmap (0 or 1) to its byte's counter — a token at a time, the wave took the
            // match's path (its symbols are sixty instructions) in nearly every turn because SOME lane had a match.  The few
            // matches keep a loop of their own.
            uint32_t *hl = h8 + (lane & 7) * 320;
#pragma unroll
            for (int k = 0; k < WPT; k++) {
                const uint32_t mw = mw_r[k], lit = tw_r[k] & ~mw;
                if (lit) {
                    const uint4 *d4 = reinterpret_cast<const uint4 *>(data + 32 * (w0 + k));
                    const uint4 lo = d4[0], hi = d4[1];
                    const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int b = 0; b < 32; b++) atomicAdd(&hl[(dw[b >> 2] >> (8 * (b & 3))) & 255u], (lit >> b) & 1u);
                }
                uint32_t tw = tw_r[k] & mw;
                while (tw) {
                    const int b = __builtin_ctz(tw);
                    tw &= tw - 1u;
                    const uint32_t rec = match[mb_r[k] + (uint32_t)__builtin_popcount(mw & ((1u << b) - 1u))];
                    atomicAdd(&hl[length_symbol((rec >> 16) + 3u).sym], 1u);
                    atomicAdd(&hl[288 + dist_symbol(rec & 0xffffu).sym], 1u);
                }
            }
        }
        __syncthreads();
        for (int s = tid; s < 320; s += WG) {
            uint32_t f = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) f += h8[k * 320 + s];
            if (s < 288) {
                const uint32_t fl = s == 256 ? 1u : (s < NUM_LITLEN ? f : 0u);
                ms->freq_l[s] = fl;
                // the symbol's rank key (below), in what was the token bitmap: every thread has its words of it in registers by now
                tokw[s] = ((fl - 1u) << 9) | (uint32_t)s;
            } else ms->freq_d[s - 288] = (s - 288 < NUM_DIST) ? f : 0u;
        }
        __syncthreads();
        stamp(2);
        if (tid == 0) {  // at least two distance codes (as zlib makes sure of, for old inflaters)
            int used = 0;
            for (int s = 0; s < NUM_DIST; s++) used += ms->freq_d[s] != 0;
            for (int s = 0; used < 2 && s < NUM_DIST; s++)
                if (!ms->freq_d[s]) { ms->freq_d[s] = 1; used++; }
            ms->m_l = 0;
            ms->m_d = 0;
        }
        __syncthreads();
        // rank the used symbols by (frequency, symbol): the two in one key, (frequency - 1) << 9 | symbol (a block's tokens
        // are far fewer than 2^23; an unused symbol's key wraps to the top and is smaller than nobody's), so that a
        // symbol's rank is the number of smaller keys: the keys are made once (where the frequencies are summed), four come
        // with a load, and a symbol is a compare and an add
        if (tid < NUM_LITLEN) {
            const uint32_t f = ms->freq_l[tid];
            if (f) {
                const uint32_t key = ((f - 1u) << 9) | (uint32_t)tid;
                uint32_t less = 0;
                const uint4 *k4 = reinterpret_cast<const uint4 *>(tokw);  // the 288 keys (the two behind the alphabet: unused, at the top)
#pragma unroll 4
                for (int j = 0; j < 288 / 4; j++) {
                    const uint4 g = k4[j];
                    less += (g.x < key) + (g.y < key) + (g.z < key) + (g.w < key);
                }
                const uint32_t r = less;
                A_l[r] = f;
                S_l[r] = (uint32_t)tid;
                atomicAdd(&ms->m_l, 1u);
            }
        } else if (tid >= DIST_T0 && tid < DIST_T0 + NUM_DIST) {
            const int sd = tid - DIST_T0;
            const uint32_t f = ms->freq_d[sd];
            if (f) {
                uint32_t r = 0;
                for (int j = 0; j < NUM_DIST; j++) {
                    const uint32_t g = ms->freq_d[j];
                    r += g && (g < f || (g == f && j < sd));
                }
                A_d[r] = f;
                S_d[r] = (uint32_t)sd;
                atomicAdd(&ms->m_d, 1u);
            }
        }
        if (tid < 320) ms->ll[tid] = 0;
        if (tid < 64) ms->dl[tid] = 0;
        __syncthreads();
        unsigned long long t_sub = a.prof ? __builtin_readcyclecounter() : 0ull;
        auto sub = [&](int k) {
            if (a.prof && tid == 0) {
                const unsigned long long t = __builtin_readcyclecounter();
                atomicAdd(&a.prof[k], t - t_sub);
                t_sub = t;
            }
        };
        if (a.prof && tid == 0) atomicAdd(&a.prof[40], t_sub - t_prev);  // (ranks)
        // minimum-redundancy lengths: wave 0 for the literal / length alphabet (mr_code_lengths_wave), a lane of wave 1 for the
        // thirty distance codes, the other waves sum the CRC meanwhile
        if (wave == 0) {
            // (h8's histograms have been summed: its space serves the wave as scratch)
            const int m = (int)ms->m_l;
            __builtin_amdgcn_s_setprio(3);  // the block's longest serial stretch: ahead of the CRC's waves at issue
            mr_code_lengths_wave(A_l, m, h8, h8 + 320, h8 + 640, h8 + 960, lane, a.prof);
            __builtin_amdgcn_s_setprio(0);
            t_sub = a.prof ? __builtin_readcyclecounter() : 0ull;
            if (lane == 0) limit_code_lengths(A_l, m, MAX_LITLEN_BITS, ms->sortbuf);
            sub(45);
        } else if (tid == 64) {
            const int m = (int)ms->m_d;
            mr_code_lengths(A_d, m);
            limit_code_lengths(A_d, m, MAX_LITLEN_BITS, ms->sortbuf + 32);
        } else if (wave >= 2) {
            // ---- CRC-32 of the input, by the waves that have nothing to do meanwhile: slicing-by-4 over a piece per thread,
            // combined by x^(8n) mod P
            constexpr int CT = WG - 128;                                // threads of waves 2 ..
            constexpr int PIECE = ((DATA_BYTES / CT + 3) / 4 + 1) * 4;  // bytes per thread, a multiple of 4, CT * PIECE >= BLOCK
            static_assert(PIECE * CT >= BLOCK, "CRC pieces must cover the block");
            const int lo = PIECE * (tid - 128), hi = min(lo + PIECE, n);
            uint32_t part = 0;
            if (lo < hi) {
                uint32_t c = 0xffffffffu;
                int k = lo;
                const uint32_t *dw = reinterpret_cast<const uint32_t *>(data);
                for (; k + 4 <= hi; k += 4) c = crc_word(c ^ dw[k >> 2]);
                for (; k < hi; k++) c = crc_byte(c, data[k]);
                c = ~c;
                part = crc_mulmod(crc_x8n((uint32_t)(n - hi), ms->x2n), c);  // crc(A || B) = x^(8 |B|) crc(A) ^ crc(B)
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) part ^= (uint32_t)__shfl_xor((int)part, m, 64);
            if (lane == 0) ms->crc_part[wave] = part;
        }
        __syncthreads();
        sub(46);
        if ((uint32_t)tid < ms->m_l) {
            ms->ll[S_l[tid]] = (uint8_t)A_l[tid];
            atomicAdd(&ms->bl_l[A_l[tid]], 1u);
        } else if (tid >= DIST_T0 && (uint32_t)(tid - DIST_T0) < ms->m_d) {
            ms->dl[S_d[tid - DIST_T0]] = (uint8_t)A_d[tid - DIST_T0];
            atomicAdd(&ms->bl_d[A_d[tid - DIST_T0]], 1u);
        }
        __syncthreads();
        // Canonical codes (RFC 1951 §3.2.2), a symbol a lane: waves 0..4 hold the literal / length alphabet, wave 5 the
        // distances.  A symbol's code is the first code of its length plus the symbols of the same length in front of it:
        // those of its own wave from a ballot, those of the waves in front from the waves' counts (cw) — instead of every
        // lane walking the lengths in front of it in LDS.  The first codes are made by two lanes meanwhile.
        uint32_t *const cw = h8 + 1600;  // [6 waves][16 lengths] (h8 is free again: the code lengths are done)
        uint32_t my_len = 0, my_within = 0;
        const bool is_d = wave == 5;
        const int sym = is_d ? lane : tid;
        if (wave < 6) {
            if (is_d ? sym < NUM_DIST : sym < NUM_LITLEN) my_len = is_d ? ms->dl[sym] : ms->ll[sym];
#pragma unroll
            for (int bb = 1; bb <= MAX_LITLEN_BITS; bb++) {
                const unsigned long long mk = __ballot(my_len == (uint32_t)bb);
                if (my_len == (uint32_t)bb) my_within = (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
                if (lane == bb) cw[wave * 16 + bb] = (uint32_t)__builtin_popcountll(mk);
            }
        }
        if (tid == 6 * 64 || tid == 7 * 64) {  // first code of each length
            uint32_t *bl = tid == 7 * 64 ? ms->bl_d : ms->bl_l, *nc = tid == 7 * 64 ? ms->nc_d : ms->nc_l;
            uint32_t c = 0;
            bl[0] = 0;
            nc[0] = 0;
            for (int b = 1; b <= MAX_LITLEN_BITS; b++) {
                c = (c + bl[b - 1]) << 1;
                nc[b] = c;
            }
        }
        __syncthreads();
        if (my_len) {
            uint32_t before = my_within;
            if (!is_d)
                for (int w = 0; w < wave; w++) before += cw[w * 16 + (int)my_len];
            const uint32_t code = (is_d ? ms->nc_d : ms->nc_l)[my_len] + before;
            (is_d ? ms->dc : ms->lc)[sym] = (uint16_t)(__builtin_bitreverse32(code) >> (32u - my_len));
        }
        __syncthreads();
        sub(47);
        stamp(3);

        // ---- C: the dynamic-block header (RFC 1951 §3.2.7), in parallel: the hlit + hdist code lengths are cut into runs,
        // every run start expands its run into code-length symbols (16 / 17 / 18 and plain lengths, the same sequence the
        // serial cl_rle of bgzf_huff.hpp emits), a scan places them, the 19-symbol code is made by one lane, and every
        // symbol's bits are OR-ed into the header words at scanned offsets.
        if (tid == 0) { ms->cl_hlit = 257; ms->cl_hdist = 1; ms->cl_n = 0; }
        if (tid < NUM_CL + 1) { ms->cl_freq[tid] = 0; ms->cl_len[tid] = 0; ms->cl_code[tid] = 0; }
        for (int k = tid; k < 160; k += WG) ms->hdr[k] = 0;
        __syncthreads();
        t_sub = a.prof ? __builtin_readcyclecounter() : 0ull;
        if (tid < NUM_LITLEN && ms->ll[tid]) atomicMax(&ms->cl_hlit, (uint32_t)tid + 1u);
        if (tid >= DIST_T0 && tid < DIST_T0 + NUM_DIST && ms->dl[tid - DIST_T0]) atomicMax(&ms->cl_hdist, (uint32_t)(tid - DIST_T0) + 1u);
        __syncthreads();
        {
            const int hlit = (int)ms->cl_hlit, hdist = (int)ms->cl_hdist, nseq = hlit + hdist;
            auto at = [&](int k) -> int { return k < hlit ? (int)ms->ll[k] : (int)ms->dl[k - hlit]; };
            uint32_t ntok = 0;
            int v = 0, run = 0;
            if (tid < nseq) {
                v = at(tid);
                if (tid == 0 || at(tid - 1) != v) {  // a run starts here
                    run = 1;
                    while (tid + run < nseq && at(tid + run) == v) run++;
                    if (v == 0) {
                        const int full18 = run / 138, rem = run % 138;
                        ntok = (uint32_t)full18 + (rem >= 3 ? 1u : (uint32_t)rem);
                    } else {
                        const int left = run - 1, full16 = left / 6, rem = left % 6;
                        ntok = 1u + (uint32_t)full16 + (rem >= 3 ? 1u : (uint32_t)rem);
                    }
                }
            }
            uint32_t nt_all;
            uint32_t at_tok = block_excl_scan(ntok, ms->wtmp, &nt_all);
            if (run) {
                auto emit = [&](int sym, int extra) {
                    ms->cltok[at_tok++] = (uint16_t)(sym | (extra << 8));
                    atomicAdd(&ms->cl_freq[sym], 1u);
                };
                int left = run;
                if (v == 0) {
                    while (left >= 11) { const int r = left > 138 ? 138 : left; emit(18, r - 11); left -= r; }
                    if (left >= 3) { emit(17, left - 3); left = 0; }
                    while (left-- > 0) emit(0, 0);
                } else {
                    emit(v, 0);
                    left--;
                    while (left >= 3) { const int r = left > 6 ? 6 : left; emit(16, r - 3); left -= r; }
                    while (left-- > 0) emit(v, 0);
                }
            }
            if (tid == 0) ms->cl_n = nt_all;
        }
        __syncthreads();
        sub(48);
        // The code-length code — 19 symbols, 7 bits at most — by the LAST WAVE with its arrays in registers: element k of an
        // array is lane k of a VGPR, read by v_readlane with a uniform index and written by a select on the lane number, so that Moffat–
        // Katajainen's chain of a few hundred dependent accesses costs a few clocks each instead of an LDS round trip (one
        // lane with its arrays in LDS: 88 k clocks a block, most of this phase).  The other waves count their ranges' bits
        // meanwhile; this wave counts its own afterwards.
        if (wave == N_WAVES - 1) {
            const uint32_t f = lane < NUM_CL ? ms->cl_freq[lane] : 0u;
            // rank among the used symbols by (frequency, symbol), as the insertion sort of build_header orders them
            const uint32_t key = ((f - 1u) << 5) | (uint32_t)lane;  // (unused: wraps to the top)
            uint32_t less = 0;
#pragma unroll
            for (int j = 0; j < NUM_CL; j++) less += (uint32_t)__builtin_amdgcn_readlane((int)key, j) < key;
            const int m = __builtin_popcountll(__ballot(f != 0u));
            uint32_t *sf = ms->sortbuf, *ss = ms->sortbuf + 20;
            if (f) { sf[less] = f; ss[less] = (uint32_t)lane; }
            __builtin_amdgcn_wave_barrier();
            LaneArr A{lane < m ? sf[lane] : 0u, lane}, bl{0u, lane};
            const uint32_t sym_r = lane < m ? ss[lane] : 0u;  // the symbol whose length lane r will hold
            mr_code_lengths_t(A, m);
            limit_code_lengths_t(A, m, MAX_CL_BITS, bl);
            if (lane < m) ms->cl_len[sym_r] = A.v;  // (cl_len was cleared with the phase's other arrays)
            __builtin_amdgcn_wave_barrier();
            // canonical codes, a symbol a lane: the first code of each length from the counts (ballots), plus the symbols of
            // the same length in front
            const uint32_t len = lane < NUM_CL ? ms->cl_len[lane] : 0u;
            uint32_t c = 0, prev = 0, code = 0;
#pragma unroll
            for (int bb = 1; bb <= MAX_CL_BITS; bb++) {
                c = (c + prev) << 1;
                const unsigned long long mk = __ballot(len == (uint32_t)bb);
                prev = (uint32_t)__builtin_popcountll(mk);
                if (len == (uint32_t)bb) code = c + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
            }
            if (len) ms->cl_code[lane] = __builtin_bitreverse32(code) >> (32u - len);
            // BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN, then 3 bits per code-length code length in cl_order: lane k holds the
            // k-th of them (the length of symbol cl_order(k), fetched from that symbol's lane), the last one used is the highest
            // set bit of a ballot, and every lane ORs its three bits into the (cleared) header words
            // (cl_order as two packed constants, five bits an entry: a table indexed by the lane would be a load from memory)
            constexpr uint64_t ORD0 = 0x22caa324e804a30ull, ORD1 = 0x3c2e1346cull;
            const int ord = lane < 12 ? (int)((ORD0 >> (5 * lane)) & 31u) : (int)((ORD1 >> (5 * ((lane < NUM_CL ? lane : 12) - 12))) & 31u);
            const uint32_t in_order = lane < NUM_CL ? (uint32_t)__shfl((int)len, ord, 64) : 0u;
            const unsigned long long used = __ballot(in_order != 0u);
            const int hclen = max(4, used ? 64 - (int)__builtin_clzll(used) : 0);
            if (lane < hclen) {
                const uint32_t pos = 17u + 3u * (uint32_t)lane;
                atomicOr(&ms->hdr[pos >> 5], in_order << (pos & 31u));
                if ((pos & 31u) > 29u) atomicOr(&ms->hdr[(pos >> 5) + 1u], in_order >> (32u - (pos & 31u)));
            }
            if (lane == 0) {
                ms->cl_hclen = (uint32_t)hclen;
                atomicOr(&ms->hdr[0], 1u | (2u << 1) | ((ms->cl_hlit - 257u) << 3) | ((ms->cl_hdist - 1u) << 8) | ((uint32_t)(hclen - 4) << 13));
            }
        }
        // ---- D (first part): the bit counts of the threads' position ranges (WPT bitmap words each).  The literals of a word
        // without a chain: its 32 bytes come as two 16-byte loads, the 32 code lengths are looked up side by side (a token at a
        // time was two dependent LDS round trips each: 33 k clocks a block) and summed where the bitmap says a literal starts;
        // the few matches keep their own loop.
        uint32_t my_bits = 0;
#pragma unroll
        for (int k = 0; k < WPT; k++) {
            const uint32_t lit = tw_r[k] & ~mw_r[k];
            if (lit) {
                const uint4 *d4 = reinterpret_cast<const uint4 *>(data + 32 * (w0 + k));
                const uint4 lo = d4[0], hi = d4[1];
                const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int b = 0; b < 32; b++) {
                    const uint32_t c = (dw[b >> 2] >> (8 * (b & 3))) & 255u;
                    const uint32_t l = ms->ll[c];
                    my_bits += ((lit >> b) & 1u) ? l : 0u;
                }
            }
            uint32_t tw = tw_r[k] & mw_r[k];
            while (tw) {
                const int b = __builtin_ctz(tw);
                tw &= tw - 1u;
                uint64_t v;
                int nb;
                token_bits(data, mw_r[k], mb_r[k], match, ms, w0 + k, b, v, nb);
                my_bits += (uint32_t)nb;
            }
        }
        if (tid == WG - 1) my_bits += ms->ll[256];  // end of block
        sub(49);
        __syncthreads();
        sub(50);
        {
            const uint32_t nt_all = ms->cl_n, fixed_bits = 17u + 3u * ms->cl_hclen;
            uint32_t bits = 0, val = 0;
            if ((uint32_t)tid < nt_all) {
                const uint32_t t = ms->cltok[tid], sy = t & 255u, ex = t >> 8;
                const uint32_t l = ms->cl_len[sy], eb = sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u;
                val = ms->cl_code[sy] | (ex << l);
                bits = l + eb;
            }
            uint32_t cl_bits_all;
            const uint32_t o = fixed_bits + block_excl_scan(bits, ms->wtmp, &cl_bits_all);
            if (bits) {
                atomicOr(&ms->hdr[o >> 5], val << (o & 31u));
                if ((o & 31u) + bits > 32u) atomicOr(&ms->hdr[(o >> 5) + 1u], val >> (32u - (o & 31u)));
            }
            if (tid == 0) ms->hdr_bits = fixed_bits + cl_bits_all;
        }
        __syncthreads();  // hdr_bits is there
        sub(51);
        stamp(4);
        t_sub = a.prof ? __builtin_readcyclecounter() : 0ull;
        uint32_t tok_bits_all;
        const uint32_t b0 = ms->hdr_bits + block_excl_scan(my_bits, ms->wtmp, &tok_bits_all);
        if (tid == 0) {
            const uint32_t run = ms->hdr_bits + tok_bits_all;
            ms->total_bits = run;
            const uint32_t bytes = (run + 7u) >> 3;
            ms->stored = bytes > (uint32_t)n + 5u || bytes > (uint32_t)MAX_PAYLOAD;
        }
        __syncthreads();
        const uint32_t total_bits = ms->total_bits;
        if (ms->stored) {
            if (tid == 0) {
                out[0] = 1;  // BFINAL = 1, BTYPE = 00; LEN, NLEN
                out[1] = (uint8_t)(n & 255);
                out[2] = (uint8_t)(n >> 8);
                out[3] = (uint8_t)~(n & 255);
                out[4] = (uint8_t)~(n >> 8);
                a.out_size[blk] = (uint32_t)n + 5u;
            }
            for (int k = tid; k < n; k += WG) out[5 + k] = data[k];
        } else {
            const uint32_t b1 = b0 + my_bits;
            const uint32_t hdr_words = (ms->hdr_bits + 31u) >> 5;
            // The bit stream is put together in LDS — in what were the hash tables, free by now — and leaves as ONE coalesced
            // copy: ranges of two threads meet inside a word, and a thread's words lie hundreds of bytes from its neighbour's, so
            // that straight to the slot every word was a partial-line write and every seam a global atomic (PMC, round 4: 2.8
            // bytes written per payload byte for 0.58 of output).  A stream that does not fit there (hardly compressible bytes)
            // goes the old way.
            const uint32_t out_words = (total_bits + 31u) >> 5;
            const bool in_lds = out_words * 4u <= (uint32_t)HEAD_BYTES;
            uint32_t *const sink = in_lds ? reinterpret_cast<uint32_t *>(lds + L_HEAD) : out32;
            if (in_lds) {
                for (uint32_t k = tid; k < out_words; k += WG) sink[k] = 0;
            } else {
                // words that more than one writer touches are cleared first and OR-ed atomically; the others are stored whole
                if (my_bits) {
                    sink[b0 >> 5] = 0;
                    sink[(b1 - 1u) >> 5] = 0;
                }
                for (uint32_t k = tid; k < hdr_words; k += WG) sink[k] = 0;
            }
            __syncthreads();
            sub(52);
            for (uint32_t k = tid; k < hdr_words; k += WG) atomicOr(&sink[k], ms->hdr[k]);
            if (my_bits) {
                uint64_t acc = 0;
                int cnt = (int)(b0 & 31u);
                uint32_t wi = b0 >> 5;
                const uint32_t w_first = wi, w_last = (b1 - 1u) >> 5;
                auto put = [&](uint64_t v, int k) {
                    acc |= v << cnt;
                    cnt += k;
                    if (cnt >= 32) {
                        if (wi == w_first || wi == w_last) atomicOr(&sink[wi], (uint32_t)acc);
                        else sink[wi] = (uint32_t)acc;
                        wi++;
                        acc >>= 32;
                        cnt -= 32;
                    }
                };
                // The chain that places a thread's tokens took 58 k clocks for 64 tokens: a literal is two look-ups, a match a
                // chain of five and sixty instructions — and some lane of the wave has a match in nearly every turn, so every
                // turn paid for one.  The first MPRE matches of a word are therefore resolved AHEAD of the chain (a turn per
                // match of the busiest lane, not per token), bits | length << 58 in a register pair; the chain picks them by
                // rank.  A word's further matches take the long way.
                constexpr int MPRE = 3;
                uint64_t pre[WPT][MPRE];
#pragma unroll
                for (int k = 0; k < WPT; k++) {
                    uint32_t mm = tw_r[k] & mw_r[k];
#pragma unroll
                    for (int q = 0; q < MPRE; q++) {
                        pre[k][q] = 0;
                        if (mm) {
                            const int b = __builtin_ctz(mm);
                            mm &= mm - 1u;
                            uint64_t v;
                            int nb;
                            token_bits(data, mw_r[k], mb_r[k], match, ms, w0 + k, b, v, nb);
                            pre[k][q] = v | ((uint64_t)nb << 58);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < WPT; k++) {
                    uint32_t tw = tw_r[k];
                    const uint32_t mw = mw_r[k];
                    while (tw) {
                        const int b = __builtin_ctz(tw);
                        tw &= tw - 1u;
                        uint64_t v;
                        int nb;
                        if ((mw >> b) & 1u) {
                            const int r = __builtin_popcount(mw & tw_r[k] & ((1u << b) - 1u));
                            if (r < MPRE) {
                                const uint64_t pv = r == 0 ? pre[k][0] : (r == 1 ? pre[k][1] : pre[k][2]);
                                v = pv & ((1ull << 58) - 1ull);
                                nb = (int)(pv >> 58);
                            } else token_bits(data, mw, mb_r[k], match, ms, w0 + k, b, v, nb);
                        } else {
                            const uint32_t c = data[32 * (w0 + k) + b];
                            v = ms->lc[c];
                            nb = ms->ll[c];
                        }
                        if (nb > 24) {  // (a token has up to 48 bits and the accumulator up to 31 pending)
                            put(v & 0xffffffull, 24);
                            put(v >> 24, nb - 24);
                        } else put(v, nb);
                    }
                }
                if (tid == WG - 1) put(ms->lc[256], ms->ll[256]);
                if (cnt) atomicOr(&sink[wi], (uint32_t)acc);
            }
            sub(53);
            if (in_lds) {
                __syncthreads();
                sub(54);
                for (uint32_t k = tid; k < out_words; k += WG) out32[k] = sink[k];
            }
            sub(55);
            if (tid == 0) a.out_size[blk] = (total_bits + 7u) >> 3;
        }
        __syncthreads();
        stamp(5);

        if (tid == 0) {  // (the CRC was summed beside the code lengths)
            uint32_t c = 0;
            for (int w = 2; w < N_WAVES; w++) c ^= ms->crc_part[w];
            a.out_crc[blk] = c;
        }
        stamp(6);
    }
}

// exclusive scan of the members' sizes (payload + 26 bytes of BGZF header and trailer): one workgroup
__global__ __launch_bounds__(1024) void bgzf_scan_kernel(const uint32_t *out_size, uint32_t n_blocks, uint64_t *member_off, uint64_t *total) {
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u, lo = (uint32_t)tid * per, hi = min(lo + per, n_blocks);
    __shared__ int failed;
    if (tid == 0) failed = 0;
    __syncthreads();
    uint64_t s = 0;
    for (uint32_t k = lo; k < hi; k++) {
        if (out_size[k] > (uint32_t)MAX_PAYLOAD) failed = 1;  // a block the compressor gave up on (see spin_until)
        s += (uint64_t)out_size[k] + 26u;
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 1024; k++) { const uint64_t c = part[k]; part[k] = run; run += c; }
        *total = failed ? 0ull : run;  // 0: the host reports the failure instead of copying anything
    }
    __syncthreads();
    uint64_t at = part[tid];
    for (uint32_t k = lo; k < hi; k++) { member_off[k] = at; at += (uint64_t)out_size[k] + 26u; }
}

// member k = 18 bytes of header (BSIZE in the BC subfield), the payload, CRC32, ISIZE — packed one after the other
__global__ __launch_bounds__(256) void bgzf_pack_kernel(const uint8_t *slots, const uint32_t *out_size, const uint32_t *out_crc,
                                                        const uint64_t *member_off, uint64_t n_bytes, uint32_t n_blocks, uint8_t *dst) {
    const uint32_t blk = blockIdx.x;
    if (blk >= n_blocks) return;
    const uint32_t sz = out_size[blk];
    if (sz > (uint32_t)MAX_PAYLOAD) return;  // (a failed block: nothing is packed, the scan has zeroed the total)
    uint8_t *d = dst + member_off[blk];
    const uint8_t *s = slots + (uint64_t)blk * SLOT;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const uint32_t bsize = sz + 25u;  // total member size - 1
        const uint8_t h[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255u), (uint8_t)(bsize >> 8)};
        for (int k = 0; k < 18; k++) d[k] = h[k];
        const uint64_t off = (uint64_t)blk * BLOCK;
        const uint32_t isize = (uint32_t)(n_bytes - off < (uint64_t)BLOCK ? n_bytes - off : (uint64_t)BLOCK), crc = out_crc[blk];
        uint8_t *t = d + 18 + sz;
        for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(isize >> (8 * k)); }
    }
    // payload: destination-aligned dwords assembled from the (aligned) slot, the ragged ends byte by byte
    uint8_t *p = d + 18;
    const uint32_t mis = (uint32_t)((4u - ((uintptr_t)p & 3u)) & 3u), headn = mis < sz ? mis : sz;
    if ((uint32_t)tid < headn) p[tid] = s[tid];
    const uint32_t body = (sz - headn) >> 2;
    uint32_t *p32 = reinterpret_cast<uint32_t *>(p + headn);
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s);
    for (uint32_t k = tid; k < body; k += 256) {
        const uint32_t so = headn + 4u * k;  // source byte offset of this destination word
        p32[k] = __builtin_amdgcn_alignbyte(s32[(so >> 2) + 1], s32[so >> 2], so & 3u);
    }
    const uint32_t done = headn + 4u * body;
    if ((uint32_t)tid < sz - done) p[done + tid] = s[done + tid];
}

}  // namespace FADEHIP_BGZF_NS
}  // namespace fadehip
