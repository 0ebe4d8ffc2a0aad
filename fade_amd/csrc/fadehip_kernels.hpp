// fadehip_kernels.hpp — gfx950 device code of the `fade annotate` hot path.
//
// Replaces, on the device:
//   source/anno.d:61-74        gate, parse_clips (util.d:37-62), SA -> rs base bits      gate_kernel
//   source/analysis.d:34-64    floor, reverse complement (util.d:23-34), window, fetch   gate_kernel + loaders
//   source/analysis.d:67       p.sw_striped(q_seq, ref_seq)  (libparasail, un-vendored)  sw_pk_kernel<R,1|2|3> + traceback_path
//   source/analysis.d:69-83,98-107  artifact gates, ReadStatus bits (readstatus.d:5-26) select_one, traceback_path
//
// Execution model: integer DP, no MFMA.  A 16-lane DPP row owns two alignments packed as int16 halves
// (8 per wave; the int32 A/B kernel: one per row, 4 per wave); a lane owns R consecutive query rows (16*R >= Lq) and the
// 16 lanes sweep the reference window as an anti-diagonal wave.  Neighbour exchange is DPP row_shr:1, whose zero fill at
// each 16-lane row boundary IS the DP boundary condition.  The reference window is staged once into LDS as class-pair
// table offsets; the 4-bit/cell trace of the traced pass leaves the wave as fully coalesced 256-byte stores.
//
// Default pipeline (two-pass): three launches per batch and row class, nothing read back in between —
//   gate_kernel -> sw_pk_kernel<R,1>: score + end cell + wave snapshots; the wave then finishes each of its alignments
//   that needs no traceback (select_one: non-candidates, forced gap-free diagonals) and buckets the rest
//   -> sw_pk_kernel<R,2> (<R,3> with non-default A.4 rules), ONE persistent launch: traced re-computation of the
//   candidates' last steps, their tracebacks (traceback_path: CIGAR, FADE's gates, rs bits), re-tracing of the paths that
//   left their steps.  The stats.d:45-54 counters are summed on the way.  sw_pk_kernel<R,0> + traceback_kernel and
//   sw_forward_kernel<R> are the single-pass packed / int32 variants kept for A/B measurements (FADEHIP_KERNEL=pk|int32);
//   sw_long_kernel + traceback_kernel serve queries beyond 512 bases and windows beyond 8000.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fadehip.h"

namespace fadehip {

// ---------------------------------------------------------------- encodings
// nt16 codes as htslib's seq_nt16_str "=ACMGRSVTWYHKDBN" (util.d:31).
// Scoring class: A0 C1 G2 T3 N4, everything else 5 = parasail's '*' wildcard (score 0); 6 = pad column.
__host__ __device__ constexpr uint64_t make_class_lut() {
    uint64_t v = 0;
    for (int k = 0; k < 16; k++) {
        uint64_t c = 5;
        if (k == 1) c = 0;
        if (k == 2) c = 1;
        if (k == 4) c = 2;
        if (k == 8) c = 3;
        if (k == 15) c = 4;
        v |= c << (4 * k);
    }
    return v;
}
// util.d:18-20 seq_comp_table
__host__ __device__ constexpr uint64_t make_comp_lut() {
    const int t[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    uint64_t v = 0;
    for (int k = 0; k < 16; k++) v |= (uint64_t)t[k] << (4 * k);
    return v;
}
constexpr uint64_t CLASS_LUT = make_class_lut();
constexpr uint64_t COMP_LUT = make_comp_lut();
constexpr int PAD_CLASS = 6;
constexpr int C64_STRIDE = 16;  // the three 64-bit batch counters sit 128 bytes apart (one L2 line each)
constexpr int STAT_PARTS = 8;  // the stats.d counters are kept as this many partial sums (stats[8 * part + k])

__device__ __forceinline__ uint32_t lut4(uint64_t lut, uint32_t code) { return (uint32_t)(lut >> (4 * code)) & 15u; }
// base b of a packed array: byte b>>1, even base in the high nibble (BAM convention).
__device__ __forceinline__ uint32_t nib_at(const uint8_t *p, uint64_t b) {
    const uint32_t v = p[b >> 1];
    return (b & 1) ? (v & 15u) : (v >> 4);
}

// ---------------------------------------------------------------- work descriptors
struct Work {           // 32 bytes
    uint64_t r_base;    // first base of the reference window in the packed reference
    uint32_t q_base;    // first base of the query in the packed query array
    uint32_t lq, lr;
    uint32_t idx;       // level 1: pair index; level 2: read index
    uint32_t flags;     // bit0: reverse-complement the query while loading (analysis.d:40)
    uint32_t out;       // entry of the run's result array this alignment reports into (one array for all lists)
};
struct Meta {           // level-2 side data per work item, 32 bytes
    int64_t win_start;
    int32_t clip_left, clip_right, aligned_len;
    int32_t pad[3];
};
struct Fwd {            // forward pass result per work item
    int32_t score, end_q, end_r, pad;
};

constexpr int NUM_CLASSES = 10;
__host__ __device__ constexpr int class_rows(int c) {
    constexpr int t[NUM_CLASSES] = {4, 6, 8, 10, 12, 14, 16, 20, 24, 32};
    return t[c];
}
__host__ __device__ inline int class_of_len(int lq) {
    for (int c = 0; c < NUM_CLASSES; c++)
        if (16 * class_rows(c) >= lq) return c;
    return -1;
}
// Work lists: one per row class of the wave kernels plus one for queries longer than 16 * 32 = 512 bases, which
// take sw_long_kernel (a thread per alignment; rare in short-read libraries, e.g. merged pairs).
constexpr int LONG_LIST = NUM_CLASSES, NUM_LISTS = NUM_CLASSES + 1;
constexpr int MAX_LONG_QUERY = 1 << 15;
// The wave kernels stage a group's window in LDS 2,048 columns at a time (CH_COLS; 2 bytes per column): a window of any
// length streams through.  What bounds it is the end-cell key: the score pass keeps the sweep step of a row pair's best
// cell as 2 t + 1 in 16 bits, so a sweep has at most 32,767 steps.  Windows beyond WAVE_MAX_WINDOW take sw_long_kernel.
constexpr int WAVE_MAX_WINDOW = 32000;
constexpr int CH_BLOCKS = 512, CH_COLS = 4 * CH_BLOCKS;  // blocks (of 4 sweep steps = 4 columns at lane 0) per staged chunk
__host__ __device__ inline int list_of_len(int lq, int64_t lr = 0) {
    const int c = class_of_len(lq);
    if (c >= 0) return lr > WAVE_MAX_WINDOW ? LONG_LIST : c;
    return lq <= MAX_LONG_QUERY ? LONG_LIST : -1;
}

struct ScoreTab {
    uint32_t prof[8];  // prof[q class] : 8 x 4-bit entries (W + open) indexed by ref class*4
    int32_t open, ext, match, mismatch;
    uint32_t rules;    // FADEHIP_RULE_* (include/fadehip.h): the assumptions about libparasail that could not be checked
};
__host__ __device__ inline bool rule(uint32_t rules, uint32_t bit) { return (rules & bit) != 0; }

// ---------------------------------------------------------------- ASCII -> packed 4-bit
__constant__ uint8_t c_ascii_code[256];

// Two residues per thread -> one output byte.  `bad` is set when a byte is '=' (FADEHIP_E_RESIDUE).
__global__ void pack_ascii_kernel(const uint8_t *__restrict__ in, uint64_t n_bases, uint64_t out_base,
                                  uint8_t *__restrict__ out, int reject_eq, int *bad) {
    // out_base must be even; thread k writes out[(out_base>>1) + k]
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b = 2 * k;
    if (b >= n_bases) return;
    const uint8_t a0 = in[b];
    const uint8_t a1 = (b + 1 < n_bases) ? in[b + 1] : 0;
    if (reject_eq && (a0 == '=' || a1 == '=')) *bad = 1;
    out[(out_base >> 1) + k] = (uint8_t)((c_ascii_code[a0] << 4) | c_ascii_code[a1]);
}

struct Cand {           // pass-2 work item: which alignment, and the sweep step its traced re-computation resumes at
    uint32_t src;       // index into the class work list
    uint32_t c0;        // first step T0 (multiple of CK_COLS; 0 = from the start of the sweep)
};
// Pass 1 snapshots the whole wave state (H, E-hat of every row, the three values in flight between lanes and
// the reference-class shift register) after every 32 steps of the anti-diagonal sweep, so pass 2 can resume
// the sweep at any multiple of 32 exactly as if it had never stopped.
#ifndef FADEHIP_CK_SHIFT
#define FADEHIP_CK_SHIFT 7
#endif
constexpr int CK_SHIFT = FADEHIP_CK_SHIFT, CK_COLS = 1 << CK_SHIFT;  // snapshot stride (a multiple of the 16- or 32-step key window)
static_assert(CK_SHIFT >= 5 && CK_SHIFT <= 9, "snapshots fall on key-window ends");
__host__ __device__ constexpr int ck_dwords(int R) { return 2 * R + 4; }  // per lane per snapshot
constexpr int NUM_BUCKETS = 10;                       // pass-2 lists by number of sweep steps to re-compute
__host__ __device__ constexpr int bucket_cols(int b) {
    constexpr int t[NUM_BUCKETS] = {32, 48, 64, 96, 128, 192, 256, 384, 512, 1 << 30};
    return t[b];
}

// ---------------------------------------------------------------- gate (anno.d:61-74, analysis.d:34-59)
struct GateArgs {
    int32_t n_reads;
    const int32_t *tid, *pos, *l_seq;
    const uint16_t *flag;
    const uint8_t *has_sa;
    const uint32_t *cigar_off, *cigar_ops, *seq_off;
    int32_t floor_len, window;
    int32_t n_contigs;
    const int64_t *contig_len;
    const uint64_t *contig_base;  // first base of each contig in the packed genome
    int32_t max_ref_len;
    int32_t wave_lr_bound;       // longest window the wave kernels' launches were sized for (LDS, snapshots): the host's bound
    int32_t long_lr_bound, long_lq_bound;  // the same for the long list's row and trace buffers
    uint32_t list_cap[NUM_LISTS];  // entries reserved per work list (from the host's count of the batch's read lengths)
    uint32_t out_cap;            // entries of the run's result array (all lists report into one array)
    uint32_t n_cigar_ops, n_seq_bytes;  // lengths of cigar_ops / seq_packed: the offsets are checked against them HERE
    uint8_t *rs;
    Work *work[NUM_LISTS];
    Meta *meta[NUM_LISTS];
    uint32_t *counters;  // NL = NUM_LISTS: [0..NL) item counts, [NL..2NL) max lr, [2NL] error bits, [2NL+1] max lq of the long list, [2NL+2] oversize reads, [2NL+3] result entries handed out
    unsigned long long *counters64;  // [k * C64_STRIDE]: k = 0 DP cells, 1 packed sequence bytes read (query + window), 2 checkpoint bytes
    unsigned long long *stats;       // stats.d:45-54: [0] read_count, [1] clipped, [2] sup (the artifact counters come from traceback_kernel)
};

// A block orders its items by window length (an octet sweeps its longest window).  With every record of a batch sent,
// one in ten is an item: 1024 records give a block ~100 items to order (256 gave it ~25: 2 % more columns per octet)
constexpr int GATE_BLOCK = 1024;
__global__ __launch_bounds__(GATE_BLOCK) void gate_kernel(GateArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // per-thread contribution to the batch counters; reduced per wave before touching memory
    unsigned long long cells = 0, seq_bytes = 0, ck_bytes = 0;
    uint32_t lr_for_max = 0, errbits = 0, st_bits = 0, oversize = 0;
    int cls = -1;
    Work w;
    Meta m;
    if (i < a.n_reads) {
        uint32_t c0 = a.cigar_off[i], c1 = a.cigar_off[i + 1];
        const uint32_t so0 = a.seq_off[i], so1 = a.seq_off[i + 1];
        const int32_t lq = a.l_seq[i];
        // the caller's offsets are checked before anything is read through them (upload no longer walks the records
        // when the caller passes its bounds): a record that fails is skipped and the batch is failed at results
        bool bad_rec = c0 > c1 || c1 > a.n_cigar_ops || so0 > so1 || so1 > a.n_seq_bytes || lq < 0;
        if (bad_rec) { errbits |= 64u; c1 = c0; }
        // anno.d:61: count S ops; util.d:37-62 parse_clips; dhtslib alignedLength (M,D,N,=,X)
        int n_soft = 0;
        uint32_t clipL = 0, clipR = 0;
        int64_t aligned = 0;
        bool first = true;
        for (uint32_t k = c0; k < c1; k++) {
            const uint32_t op = a.cigar_ops[k] & 15u, len = a.cigar_ops[k] >> 4;
            if (op == 4) n_soft++;
            if (FADEHIP_OP_CONSUMES_REF(op)) aligned += len;
            if (op == 5) continue;  // util.d:44-45 skips hard clips
            const bool is_sc = (op == 4);
            if (first && !is_sc) first = false;
            else if (first && is_sc) clipL = len;
            else if (is_sc) clipR = len;
        }
        const uint32_t flag = a.flag[i];
        uint8_t rs = 0;
        bool want = false;
        if (!((flag & 4u) || n_soft == 0) && !bad_rec) {          // anno.d:61-65
            if (clipL != 0 || clipR != 0) rs |= 1;    // anno.d:69-70
            if (a.has_sa[i]) rs |= 32;                // anno.d:73-74
            // analysis.d:34: only clips strictly longer than the floor are re-aligned
            want = (clipL != 0 && (int64_t)clipL > a.floor_len) || (clipR != 0 && (int64_t)clipR > a.floor_len);
        }
        a.rs[i] = rs;
        st_bits = 4u | (rs & 1u) | ((rs >> 4) & 2u);  // bit2 read, bit0 clipped, bit1 sup
        const int32_t tid = a.tid[i];
        if (want && (tid < 0 || tid >= a.n_contigs)) {
            errbits |= 1u;  // mapped record without a valid contig
            want = false;
        }
        if (want && lq > 0 && so1 - so0 < (uint32_t)(lq + 1) / 2u) {
            errbits |= 8u;  // a record that must be re-aligned came without its bases
            want = false;
        }
        if (want && lq > 0) {
            // analysis.d:45-59
            const int64_t pos = a.pos[i];
            int64_t start = pos - a.window;
            if (start < 0) start = 0;
            int64_t end = pos + aligned + a.window;
            if (end > a.contig_len[tid]) end = a.contig_len[tid];
            const int64_t lr = end - start;
            if (lr > 0) {
                cls = list_of_len(lq, lr);
                // a read or window beyond the kernels' limits is left un-re-aligned (its rs keeps the sc / sup bits) and counted
                if (cls < 0) oversize = 1u;  // read longer than MAX_LONG_QUERY
                else if (lr > a.max_ref_len) { oversize = 1u; cls = -1; }  // window longer than max_ref_len
                else if (cls != LONG_LIST ? lr > a.wave_lr_bound : (lr > a.long_lr_bound || lq > a.long_lq_bound)) {
                    errbits |= 16u;  // the caller's ref_span_bound was too small
                    cls = -1;
                }
                else {
                    cells = (unsigned long long)lq * (unsigned long long)lr;
                    seq_bytes = (unsigned long long)((lq + 1) / 2 + (lr + 1) / 2);
                    ck_bytes = cls == LONG_LIST ? 0ull : (unsigned long long)lq * (unsigned long long)(lr >> CK_SHIFT) * 4ull;  // H + E-hat, int16 each
                    lr_for_max = (uint32_t)lr;
                    w.r_base = a.contig_base[tid] + (uint64_t)start;
                    w.q_base = so0 * 2u;
                    w.lq = (uint32_t)lq;
                    w.lr = (uint32_t)lr;
                    w.idx = (uint32_t)i;
                    w.flags = 1u;
                    w.out = 0;
                    m.win_start = start;
                    m.clip_left = (int32_t)clipL;
                    m.clip_right = (int32_t)clipR;
                    m.aligned_len = (int32_t)aligned;
                    m.pad[0] = m.pad[1] = m.pad[2] = 0;
                }
            }
        }
    }
    // block-aggregated append: slots are first reserved in LDS, then one global atomic per block and
    // class (per-lane or even per-wave atomics on one address serialise at ~12 ns each and dominated
    // this kernel)
    __shared__ uint32_t s_cnt[NUM_LISTS], s_base[NUM_LISTS], s_maxlr[NUM_LISTS], s_err, s_items, s_maxlq, s_out_base;
    __shared__ unsigned long long s_cells, s_bytes, s_ck;
    __shared__ uint32_t s_key[GATE_BLOCK], s_rank[GATE_BLOCK], s_stat[3];
    if (threadIdx.x < NUM_LISTS) { s_cnt[threadIdx.x] = 0; s_maxlr[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { s_err = 0; s_cells = 0; s_bytes = 0; s_ck = 0; s_items = 0; s_maxlq = 0; s_stat[0] = s_stat[1] = s_stat[2] = 0; }
    __syncthreads();
    uint32_t compact = 0;
    if (cls >= 0) {
        atomicAdd(&s_cnt[cls], 1u);
        atomicMax(&s_maxlr[cls], lr_for_max);
        if (cls == LONG_LIST) atomicMax(&s_maxlq, w.lq);
        compact = atomicAdd(&s_items, 1u);
        s_key[compact] = ((uint32_t)cls << 24) | lr_for_max;  // lr <= 16000
    }
    if (__ballot(cells != 0)) {
        for (int sh = 32; sh >= 1; sh >>= 1) {
            cells += __shfl_xor(cells, sh, 64);
            seq_bytes += __shfl_xor(seq_bytes, sh, 64);
            ck_bytes += __shfl_xor(ck_bytes, sh, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_cells, cells);
            atomicAdd(&s_bytes, seq_bytes);
            atomicAdd(&s_ck, ck_bytes);
        }
    }
    if (errbits) atomicOr(&s_err, errbits);
    if (oversize) atomicAdd(&a.counters[2 * NUM_LISTS + 2], 1u);  // rare by construction
    {
        // stats.d:45-54, the part known here: reads, clipped, supplementary — one LDS add per wave and counter
        const unsigned long long m_read = __ballot(st_bits & 4u), m_sc = __ballot(st_bits & 1u), m_sup = __ballot(st_bits & 2u);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_stat[0], (uint32_t)__popcll(m_read));
            if (m_sc) atomicAdd(&s_stat[1], (uint32_t)__popcll(m_sc));
            if (m_sup) atomicAdd(&s_stat[2], (uint32_t)__popcll(m_sup));
        }
    }
    __syncthreads();
    if (threadIdx.x < NUM_LISTS && s_cnt[threadIdx.x]) {
        s_base[threadIdx.x] = atomicAdd(&a.counters[threadIdx.x], s_cnt[threadIdx.x]);
        atomicMax(&a.counters[NUM_LISTS + threadIdx.x], s_maxlr[threadIdx.x]);
    }
    if (threadIdx.x == 0) {
        if (s_items) s_out_base = atomicAdd(&a.counters[2 * NUM_LISTS + 3], s_items);  // the block's entries of the result array
        if (s_cells) {
            atomicAdd(&a.counters64[0 * C64_STRIDE], s_cells);
            atomicAdd(&a.counters64[1 * C64_STRIDE], s_bytes);
            atomicAdd(&a.counters64[2 * C64_STRIDE], s_ck);
        }
        if (s_err) atomicOr(&a.counters[2 * NUM_LISTS], s_err);
        if (s_maxlq) atomicMax(&a.counters[2 * NUM_LISTS + 1], s_maxlq);
    }
    if (threadIdx.x >= 64 && threadIdx.x < 67 && a.stats && s_stat[threadIdx.x - 64])
        atomicAdd(&a.stats[8 * (blockIdx.x % STAT_PARTS) + threadIdx.x - 64], (unsigned long long)s_stat[threadIdx.x - 64]);
    // The block's items of one class are appended in order of decreasing window length: the eight alignments of
    // a wave sweep max(lr) + 15 steps, so neighbours of equal length waste none (C2: 339 -> 323 columns per
    // octet).  Rank = items of the same class with a longer window (ties by list position); a block holds
    // ~100 items for 10 % clipped reads, and only threads below the item count loop.
    if (threadIdx.x < s_items) {
        const uint32_t me = s_key[threadIdx.x];
        const uint32_t n = s_items;
        uint32_t r = 0;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t o = s_key[k];
            r += ((o ^ me) >> 24) == 0 && (o > me || (o == me && k < threadIdx.x));
        }
        s_rank[threadIdx.x] = r;
    }
    __syncthreads();
    if (cls >= 0) {
        const uint32_t slot = s_base[cls] + s_rank[compact];
        w.out = s_out_base + compact;
        if (slot < a.list_cap[cls] && w.out < a.out_cap) {
            a.work[cls][slot] = w;
            a.meta[cls][slot] = m;
        } else atomicOr(&a.counters[2 * NUM_LISTS], 32u);  // more items than the host's bound: never written, reported at collect
    }
}

// ---------------------------------------------------------------- forward SW with trace
// Pass 2 runs all buckets in one persistent launch, longest first.  Each wave turns the selection's bucket counts into
// the table itself: octet `oct` (a ticket) belongs to slot k with oct_first[k] <= oct < oct_first[k+1], bucket
// NUM_BUCKETS - 1 - k; its candidates are cand[bucket * cap + 8 * (oct - oct_first[k]) ...].  The trace of an octet goes
// to the scratch region of the WAVE that drew it (blockIdx.x * quad_stride dwords): the scratch is sized by the launch,
// not by the number of candidates.
// what the score pass's waves need to finish their alignments on the spot (select_one, below)
struct SelArgs {
    const Meta *meta;       // nullptr: level 1, every alignment is traced
    int32_t floor_len;
    int32_t trace_all;
    int32_t span_slack;     // columns added to the expected path span (24; tests shrink it to force re-traced paths)
    Cand *cand;             // [NUM_BUCKETS][cap]
    uint32_t cap;
    uint32_t *bucket_n;     // [NUM_BUCKETS] candidates per bucket (atomically appended)
    fadehip_aln *out;
    uint8_t *rs;
    unsigned long long *stats;
    int32_t gate;
    int32_t match;          // score of a matching pair; 0 switches the forced-diagonal shortcut off
    int32_t mismatch;
    int32_t enabled;        // 0: the launch only scores (single-pass A/B kernels never get here)
    int32_t no_ckpt;        // 1: this launch leaves no snapshots: every candidate is traced from step 0
};

struct SwArgs {
    const Work *work;
    int32_t n_items;        // items in this launch (quads = ceil(n/4))
    const uint8_t *q_nib;   // packed queries
    const uint8_t *r_nib;   // packed reference
    uint32_t *trace;        // quad-major trace storage
    uint64_t quad_stride;   // dwords per quad
    int32_t ref_stride;     // LDS bytes per group (multiple of 16)
    Fwd *fwd;
    ScoreTab sc;
    // two-pass path (sw_pk_kernel MODE 1 / 2)
    const Cand *cand;       // MODE 2: n_items candidates; item k is alignment cand[k].src
    uint32_t *ckpt;         // [pass-1 octet][checkpoint][H rows | E rows][64 lanes]
    uint64_t ck_stride;     // dwords per pass-1 octet
    int32_t n_ck;           // checkpoints per octet
    // Launches are sized on the host from upper bounds; what the gate and the selection actually produced stays on
    // the device (fadehip_annotate_run never reads a counter back):
    const uint32_t *count_dev;  // MODE 0 / 1: items in the class list (this launch covers [item_base, item_base + n_items) of them); nullptr = n_items is exact
    uint32_t item_base;
    SelArgs sel;                // MODE 1: the selection, done by the wave that scored the alignment
    const uint32_t *bucket_n;   // MODE 2: [NUM_BUCKETS] candidates per bucket, as the selection left them
    uint32_t cand_cap;          // MODE 2: entries per bucket of `cand`
    uint32_t *ticket;           // MODE 2 (MODE 1 when set: the persistent variant): the waves of the launch draw octets from this counter until the table / list is exhausted
    unsigned long long *cand_total;  // MODE 2: + candidates of this launch (by the wave that draws ticket 0)
    // MODE 2 walks the traceback of an octet's candidates right after tracing them (same wave, trace still in L2)
    const Meta *meta;           // nullptr for level 1
    fadehip_aln *out;
    uint8_t *rs;
    unsigned long long *stats;
    int32_t floor_len, gate, early_out;
    unsigned long long *rerun_total;  // candidates whose path left the traced steps and were traced again from further back
};

#define DPP_ROW_SHR1 0x111

// One DP cell in the "hat" domain: Eh = E + open, Fh = F + open, Dp = D + open,
// T = max(Dp, Eh, Fh), H = max(T - open, 0) as one saturating subtract.
// Trace flags pushed MSB-first: nd (H!=D), nf (H!=F), eo (E opened), fo (F opened).
template <int R>
__global__ __launch_bounds__(64) void sw_forward_kernel(SwArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int lane = threadIdx.x;
    const int g = lane >> 4, lig = lane & 15;
    const int quad = blockIdx.x;
    const int item = quad * 4 + g;
    const bool have = item < a.n_items;
    Work w;
    if (have) w = a.work[item];
    else { w.r_base = 0; w.q_base = 0; w.lq = 0; w.lr = 0; w.idx = 0; w.flags = 0; w.out = 0; }
    const int lq = (int)w.lq, lr = (int)w.lr;

    // steps this wave needs: longest window of its 4 groups + 15 lanes of skew, in blocks of 4
    int maxlr = __builtin_amdgcn_readlane(lr, 0);
    maxlr = max(maxlr, __builtin_amdgcn_readlane(lr, 16));
    maxlr = max(maxlr, __builtin_amdgcn_readlane(lr, 32));
    maxlr = max(maxlr, __builtin_amdgcn_readlane(lr, 48));
    const int n_blocks = (maxlr + 15 + 3) >> 2;

    // ---- stage the reference window into LDS as class*4 bytes; pad columns get PAD_CLASS*4
    uint8_t *lref = lds + g * a.ref_stride;
    const int n_cols = n_blocks * 4;
    for (int k = lig; k < n_cols; k += 16) {
        uint32_t c4 = PAD_CLASS * 4;
        if (k < lr) c4 = lut4(CLASS_LUT, nib_at(a.r_nib, w.r_base + (uint64_t)k)) * 4u;
        lref[k] = (uint8_t)c4;
    }

    // ---- per-row score profiles from the (reverse-complemented) query
    uint32_t prof[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int row = lig * R + r;
        uint32_t p = a.sc.prof[PAD_CLASS];
        if (row < lq) {
            uint32_t code;
            if (w.flags & 1u) code = lut4(COMP_LUT, nib_at(a.q_nib, (uint64_t)w.q_base + (uint32_t)(lq - 1 - row)));
            else code = nib_at(a.q_nib, (uint64_t)w.q_base + (uint32_t)row);
            p = a.sc.prof[lut4(CLASS_LUT, code)];
        }
        prof[r] = p;
    }
    __syncthreads();

    int32_t Hl[R], Eh[R];
    uint32_t best[R];
#pragma unroll
    for (int r = 0; r < R; r++) { Hl[r] = 0; Eh[r] = 0; best[r] = 0; }
    int32_t hu_out = 0, fu_out = 0, hu_prev = 0;
    uint32_t rc = PAD_CLASS * 4;
    const int32_t open = a.sc.open, ext = a.sc.ext;
    uint32_t *tq = a.trace + (uint64_t)quad * a.quad_stride + lane;
    constexpr int ND = R / 2;  // trace dwords per lane per 4-step block

    for (int blk = 0; blk < n_blocks; blk++) {
        const uint32_t rw = *reinterpret_cast<const uint32_t *>(lref + blk * 4);
        uint32_t acc[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) acc[k] = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int t = blk * 4 + s;
            const uint32_t fresh = (rw >> (8 * s)) & 0xffu;
            // shift the column state one lane to the right; lane 0 of each 16-lane row takes the
            // next reference column / the DP boundary zeros.
            rc = (uint32_t)__builtin_amdgcn_update_dpp((int)fresh, (int)rc, DPP_ROW_SHR1, 0xf, 0xf, false);
            int32_t hu = __builtin_amdgcn_update_dpp(0, hu_out, DPP_ROW_SHR1, 0xf, 0xf, true);
            int32_t fu = __builtin_amdgcn_update_dpp(0, fu_out, DPP_ROW_SHR1, 0xf, 0xf, true);
            const bool valid = (uint32_t)(t - lig) < (uint32_t)lr;
            const uint32_t vmul = valid ? 65536u : 0u;
            const uint32_t ctv = valid ? (uint32_t)(0xffff - t) : 0u;
            int32_t hd = hu_prev;
            hu_prev = hu;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int32_t wq = (int32_t)__builtin_amdgcn_ubfe(prof[r], rc, 4);
                const int32_t Dp = hd + wq;
                const int32_t hl = Hl[r];
                const int32_t Ee = Eh[r] - ext;
                const int32_t En = max(hl, Ee);
                const int32_t Fe = fu - ext;
                const int32_t Fn = max(hu, Fe);
                const int32_t T = max(max(Dp, En), Fn);
                const int32_t H = (int32_t)__builtin_elementwise_sub_sat((uint32_t)T, (uint32_t)open);
                // each flag is the sign bit of a difference, shifted into the accumulator by one
                // v_alignbit: nd (D < H), nf (F < H), eo (E opened), fo (F opened)
                const int k0 = (s * R + r) * 4;  // flag index of `nd`; nf, eo, fo follow
                uint32_t &ac = acc[k0 >> 5];
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Dp - T), 31);
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Fn - T), 31);
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Ee - hl), 31);
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Fe - hu), 31);
                const uint32_t key = __umul24((uint32_t)H, vmul) + ctv;
                best[r] = max(best[r], key);
                hd = hl;
                Hl[r] = H;
                Eh[r] = En;
                hu = H;
                fu = Fn;
            }
            hu_out = hu;
            fu_out = fu;
        }
        uint32_t *tp = tq + (uint64_t)blk * (ND * 64);
#pragma unroll
        for (int k = 0; k < ND; k++) tp[k * 64] = acc[k];
    }

    // ---- end cell: max H, then smallest ref index, then smallest query index (Appendix A.3)
    uint32_t bk = 0;
    int brow = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int row = lig * R + r;
        if (row < lq && best[r] > bk) { bk = best[r]; brow = row; }
    }
    // low half holds 0xffff - t; column j = t - lig
    uint64_t comp = bk ? ((((uint64_t)(bk + (uint32_t)lig)) << 16) | (uint64_t)(0xffff - brow)) : 0ull;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const uint64_t o = __shfl_xor(comp, m, 64);
        comp = o > comp ? o : comp;
    }
    if (have && lig == 0) {
        Fwd f;
        f.score = (int32_t)(comp >> 32);
        f.end_r = comp ? (int32_t)(0xffff - ((comp >> 16) & 0xffff)) : 0;
        f.end_q = comp ? (int32_t)(0xffff - (comp & 0xffff)) : 0;
        f.pad = 0;
        a.fwd[item] = f;
    }
}

// The same sweep with ONE alignment per wavefront: the 64 lanes own 64 * R consecutive query rows (R = 16: up to 1,024 bases,
// R = 32: up to 2,048, R = 64: up to 4,096), the hand-off between lanes is a DPP wave_shr:1 instead of row_shr:1 (lane 0 takes the boundary
// zeros), the skew is 63 steps.  For the reads the 16-lane kernels cannot hold (longer than 512 bases) and the windows
// they cannot stage (beyond WAVE_MAX_WINDOW): analysis.d:45-59 sets no limit on either.  int32 arithmetic (scores reach
// 2 * 4,096), four trace flags per cell in the layout of sw_forward_kernel with g = 0 and 64 lanes of rows — the
// traceback reads it with the same function.
#define DPP_WAVE_SHR1 0x138
template <int R>
__global__ __launch_bounds__(64) void sw_forward64_kernel(SwArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int lane = threadIdx.x;
    const int lig = lane;
    const int quad = blockIdx.x;
    const int item = quad;
    int n_live = a.n_items;
    if (a.count_dev) n_live = min(n_live, max(0, (int)*a.count_dev - (int)a.item_base));
    const bool have = item < n_live;
    Work w;
    if (have) w = a.work[item];
    else { w.r_base = 0; w.q_base = 0; w.lq = 0; w.lr = 0; w.idx = 0; w.flags = 0; w.out = 0; }
    const int lq = (int)w.lq, lr = (int)w.lr;

    // steps this wave needs: the window + 63 lanes of skew, in blocks of 4
    const int n_blocks = have ? (lr + 63 + 3) >> 2 : 0;

    // ---- stage the reference window into LDS as class*4 bytes; pad columns get PAD_CLASS*4
    uint8_t *lref = lds;
    const int n_cols = n_blocks * 4;
    for (int k = lig; k < n_cols; k += 64) {
        uint32_t c4 = PAD_CLASS * 4;
        if (k < lr) c4 = lut4(CLASS_LUT, nib_at(a.r_nib, w.r_base + (uint64_t)k)) * 4u;
        lref[k] = (uint8_t)c4;
    }

    // ---- per-row score profiles from the (reverse-complemented) query
    uint32_t prof[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int row = lig * R + r;
        uint32_t p = a.sc.prof[PAD_CLASS];
        if (row < lq) {
            uint32_t code;
            if (w.flags & 1u) code = lut4(COMP_LUT, nib_at(a.q_nib, (uint64_t)w.q_base + (uint32_t)(lq - 1 - row)));
            else code = nib_at(a.q_nib, (uint64_t)w.q_base + (uint32_t)row);
            p = a.sc.prof[lut4(CLASS_LUT, code)];
        }
        prof[r] = p;
    }
    __syncthreads();

    int32_t Hl[R], Eh[R];
    uint32_t best[R];
#pragma unroll
    for (int r = 0; r < R; r++) { Hl[r] = 0; Eh[r] = 0; best[r] = 0; }
    int32_t hu_out = 0, fu_out = 0, hu_prev = 0;
    uint32_t rc = PAD_CLASS * 4;
    const int32_t open = a.sc.open, ext = a.sc.ext;
    uint32_t *tq = a.trace + (uint64_t)quad * a.quad_stride + lane;
    constexpr int ND = R / 2;  // trace dwords per lane per 4-step block

    for (int blk = 0; blk < n_blocks; blk++) {
        const uint32_t rw = *reinterpret_cast<const uint32_t *>(lref + blk * 4);
        uint32_t acc[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) acc[k] = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int t = blk * 4 + s;
            const uint32_t fresh = (rw >> (8 * s)) & 0xffu;
            // shift the column state one lane to the right; lane 0 of each 16-lane row takes the
            // next reference column / the DP boundary zeros.
            rc = (uint32_t)__builtin_amdgcn_update_dpp((int)fresh, (int)rc, DPP_WAVE_SHR1, 0xf, 0xf, false);
            int32_t hu = __builtin_amdgcn_update_dpp(0, hu_out, DPP_WAVE_SHR1, 0xf, 0xf, true);
            int32_t fu = __builtin_amdgcn_update_dpp(0, fu_out, DPP_WAVE_SHR1, 0xf, 0xf, true);
            const bool valid = (uint32_t)(t - lig) < (uint32_t)lr;
            const uint32_t vmul = valid ? 65536u : 0u;
            const uint32_t ctv = valid ? (uint32_t)(0xffff - t) : 0u;
            int32_t hd = hu_prev;
            hu_prev = hu;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int32_t wq = (int32_t)__builtin_amdgcn_ubfe(prof[r], rc, 4);
                const int32_t Dp = hd + wq;
                const int32_t hl = Hl[r];
                const int32_t Ee = Eh[r] - ext;
                const int32_t En = max(hl, Ee);
                const int32_t Fe = fu - ext;
                const int32_t Fn = max(hu, Fe);
                const int32_t T = max(max(Dp, En), Fn);
                const int32_t H = (int32_t)__builtin_elementwise_sub_sat((uint32_t)T, (uint32_t)open);
                // each flag is the sign bit of a difference, shifted into the accumulator by one
                // v_alignbit: nd (D < H), nf (F < H), eo (E opened), fo (F opened)
                const int k0 = (s * R + r) * 4;  // flag index of `nd`; nf, eo, fo follow
                uint32_t &ac = acc[k0 >> 5];
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Dp - T), 31);
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Fn - T), 31);
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Ee - hl), 31);
                ac = __builtin_amdgcn_alignbit(ac, (uint32_t)(Fe - hu), 31);
                const uint32_t key = __umul24((uint32_t)H, vmul) + ctv;
                best[r] = max(best[r], key);
                hd = hl;
                Hl[r] = H;
                Eh[r] = En;
                hu = H;
                fu = Fn;
            }
            hu_out = hu;
            fu_out = fu;
        }
        uint32_t *tp = tq + (uint64_t)blk * (ND * 64);
#pragma unroll
        for (int k = 0; k < ND; k++) tp[k * 64] = acc[k];
    }

    // ---- end cell: max H, then smallest ref index, then smallest query index (Appendix A.3)
    uint32_t bk = 0;
    int brow = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int row = lig * R + r;
        if (row < lq && best[r] > bk) { bk = best[r]; brow = row; }
    }
    // low half holds 0xffff - t; column j = t - lig
    uint64_t comp = bk ? ((((uint64_t)(bk + (uint32_t)lig)) << 16) | (uint64_t)(0xffff - brow)) : 0ull;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint64_t o = __shfl_xor(comp, m, 64);
        comp = o > comp ? o : comp;
    }
    if (have && lig == 0) {
        Fwd f;
        f.score = (int32_t)(comp >> 32);
        f.end_r = comp ? (int32_t)(0xffff - ((comp >> 16) & 0xffff)) : 0;
        f.end_q = comp ? (int32_t)(0xffff - (comp & 0xffff)) : 0;
        f.pad = 0;
        a.fwd[item] = f;
    }
}

// ---------------------------------------------------------------- traceback + artifact gates
struct TbArgs {
    const Work *work;
    const Meta *meta;       // nullptr for level 1
    const Fwd *fwd;
    int32_t n_items;
    int32_t R;
    const uint8_t *q_nib, *r_nib;
    const uint32_t *trace;
    uint64_t quad_stride;
    ScoreTab sc;
    fadehip_aln *out;       // the run's result array; an alignment reports into entry Work::out
    uint8_t *rs;            // level 2: per-read status, OR-ed with the artifact bits
    unsigned long long *stats;  // level 2: stats.d:45-54 [3] art_sup, [4] art, [6] aln_l, [7] aln_r are added here
    int32_t floor_len;
    int32_t gate;           // 1: apply analysis.d:69-83,98-107
    int32_t early_out;      // 1: a path that leaves the traced steps with > 10 ops already is not re-run (see below)
    int32_t packed;         // trace layout: 0 sw_forward_kernel (quads), 1 packed kernels (octets), 2 sw_long_kernel (per thread), 3 sw_forward64_kernel (a wave each)
    const uint8_t *ltrace;  // packed == 2: cell (i, j) of item k is nibble j & 1 of ltrace[(i * lhalf + j / 2) * n_items + k]
    int32_t lhalf;          // packed == 2: bytes per trace row
    const uint32_t *count_dev;  // items on the list (nullptr = n_items is exact), as in SwArgs
    uint32_t item_base;
};

__device__ __forceinline__ uint32_t trace_nibble(const uint32_t *tq, int R, int g, int i, int j) {
    const int lig = i / R, r = i - lig * R;
    const int t = j + lig;
    const int k0 = ((t & 3) * R + r) * 4;
    const uint32_t wv = tq[((uint64_t)(t >> 2) * (R >> 1) + (k0 >> 5)) * 64 + (g * 16 + lig)];
    return (wv >> (28 - (k0 & 31))) & 15u;
}

// Packed trace: [block of 4 steps][R dwords][64 lanes]; dword k of a block holds, for s = 0..3, the cell of step
// 4*blk + s in row (k + s) % R — four cells of one DP diagonal — at nibble 3 - s of the low / high half (A / B).
// A traceback that walks a diagonal therefore stays in one dword for up to four steps.
__device__ __forceinline__ uint32_t trace_nibble_pk(const uint32_t *to, int R, int g, int half, int i, int j) {
    const int lig = i / R, r = i - lig * R;
    const int t = j + lig;
    const int s = t & 3;
    int k = r - s;
    if (k < 0) k += R;
    const uint32_t wv = to[((uint64_t)(t >> 2) * R + k) * 64 + (g * 16 + lig)];
    return (wv >> (16 * half + 4 * (3 - s))) & 15u;
}

// 8 consecutive nibbles starting at nibble index n0 of a packed sequence: the two aligned dwords that hold them
struct Nib8 {
    uint64_t word;
    uint64_t byte0;  // byte index of the word's first byte
};
__device__ __forceinline__ Nib8 load_nib8(const uint8_t *p, uint64_t n0) {
    Nib8 r;
    r.byte0 = (n0 >> 1) & ~3ull;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p + r.byte0);
    const uint32_t lo = q[0];
    // 8 nibbles starting in byte (n0 >> 1) end at most 4 bytes later: the second dword is needed unless they fit
    const uint32_t hi = (((n0 + 7) >> 1) - r.byte0 >= 4) ? q[1] : 0u;
    r.word = (uint64_t)lo | ((uint64_t)hi << 32);
    return r;
}
__device__ __forceinline__ uint32_t nib8_at(const Nib8 &w, uint64_t n) {
    const uint32_t byte = (uint32_t)((w.word >> (8 * ((n >> 1) - w.byte0))) & 0xffu);
    return (n & 1) ? (byte & 15u) : (byte >> 4);
}

// One alignment: list entry `src`, traced from sweep step c0 on (0 = from the start) into `tq`, where it occupies group g
// (16-lane row), half `half`; `item` is its index in the launch (thread-per-alignment trace layout only).
// Returns the artifact bits it set (bit0 left, bit1 right) | 4 if the read is supplementary | 8 if the path left the
// traced steps and the alignment must be traced again from further back (nothing is written then).
__device__ __forceinline__ uint32_t traceback_path(const TbArgs &a, const int src, const int c0, const uint32_t *tq, const int g,
                                                   const int half, const int item) {
    const Work w = a.work[src];
    const Fwd f = a.fwd[src];
    const int R = a.R;
    const int lq = (int)w.lq;
    const bool rcq = w.flags & 1u;
    const int32_t open = a.sc.open, ext = a.sc.ext;
    const bool hdir_f_first = rule(a.sc.rules, FADEHIP_RULE_HDIR_DIAG_F_E), eq_by_char = rule(a.sc.rules, FADEHIP_RULE_EQ_BY_CHAR);
    const bool pad = rule(a.sc.rules, FADEHIP_RULE_PAD_SOFTCLIP), n_eq_n = rule(a.sc.rules, FADEHIP_RULE_N_MATCHES_N);
    const int op_ref_only = rule(a.sc.rules, FADEHIP_RULE_SAM_GAP_LETTERS) ? 2 : 1, op_query_only = 3 - op_ref_only;  // A.5: 'D' / 'I'

    // Appendix A.4 traceback.  The value of the current cell is carried along (h), so the
    // ZERO stop needs no stored flag.
    uint32_t ring[FADEHIP_MAX_OPS];
    uint32_t first_gen = 0;  // first generated run == last run of the CIGAR
    int n_runs = 0;
    int cur_op = -1;
    uint32_t cur_len = 0;
    int i = f.end_q, j = f.end_r, state = 0;
    int32_t h = f.score;
    bool done = false, left_range = false;  // path ended (H reached 0) / path left the traced steps
    auto emit = [&](int op) {
        if (op == cur_op) cur_len++;
        else {
            if (cur_op >= 0) {
                const uint32_t v = (cur_len << 4) | (uint32_t)cur_op;
                if (n_runs == 0) first_gen = v;
                ring[n_runs & (FADEHIP_MAX_OPS - 1)] = v;
                n_runs++;
            }
            cur_op = op; cur_len = 1;
        }
    };
    auto in_range = [&](int ii, int jj) { return ii >= 0 && jj >= 0 && !(c0 > 0 && jj + ii / R < c0); };
    auto nibble = [&](int ii, int jj) -> uint32_t {
        if (a.packed == 2) {
            const uint32_t v = a.ltrace[((uint64_t)ii * (uint32_t)a.lhalf + (uint32_t)(jj >> 1)) * (uint32_t)a.n_items + (uint32_t)item];
            return (jj & 1) ? (v >> 4) : (v & 15u);
        }
        return (a.packed == 1) ? trace_nibble_pk(tq, R, g, half, ii, jj - c0) : trace_nibble(tq, R, g, ii, jj);  // (0 and 3 share a layout)
    };
    constexpr int DIAG_BATCH = 8;
    while (i >= 0 && j >= 0) {
        if (c0 > 0 && j + i / R < c0) { left_range = true; break; }  // cell (i, j) was computed at step j + i/R
        if (state == 0) {
            // A path is mostly diagonal runs, and every step of this walk is a chain of dependent global loads
            // (trace nibble, query base, window base).  Fetch the next DIAG_BATCH cells of the diagonal at once and
            // consume them while the flags say "diagonal": one memory round trip per 8 steps instead of per step.
            uint32_t nbk[DIAG_BATCH], qck[DIAG_BATCH], rck[DIAG_BATCH];
            bool vk[DIAG_BATCH];
            // the batch's 8 query and 8 window bases are 8 consecutive nibbles each: two aligned dwords per
            // sequence instead of 8 byte loads (this kernel is bound by its scattered memory transactions)
            const int nv = min(DIAG_BATCH, min(i, j) + 1);  // cells of the batch inside the matrix
            const uint64_t qn_lo = rcq ? (uint64_t)w.q_base + (uint32_t)(lq - 1 - i) : (uint64_t)w.q_base + (uint32_t)(i - (nv - 1));
            const uint64_t rn_lo = w.r_base + (uint64_t)(j - (nv - 1));
            const Nib8 qw = load_nib8(a.q_nib, qn_lo), rw = load_nib8(a.r_nib, rn_lo);
#pragma unroll
            for (int k = 0; k < DIAG_BATCH; k++) {
                const int ii = i - k, jj = j - k;
                vk[k] = in_range(ii, jj);
                nbk[k] = 0; qck[k] = 0; rck[k] = 0;
                if (vk[k]) {
                    nbk[k] = nibble(ii, jj);
                    const uint32_t qraw = nib8_at(qw, rcq ? qn_lo + (uint32_t)k : qn_lo + (uint32_t)(nv - 1 - k));
                    qck[k] = rcq ? lut4(COMP_LUT, qraw) : qraw;
                    rck[k] = nib8_at(rw, rn_lo + (uint32_t)(nv - 1 - k));
                }
            }
            bool stop = false;
#pragma unroll
            for (int k = 0; k < DIAG_BATCH; k++) {
                if (stop) continue;
                if (!vk[k]) { stop = true; continue; }          // the outer loop re-checks bounds and range
                if (h == 0) { done = true; stop = true; continue; }
                if (nbk[k] & 8u) {                               // H != D: H == F (query-only) has priority over E (A.4; bit 4 = "H != E" the other way)
                    state = ((nbk[k] & 4u) != 0) == hdir_f_first ? 1 : 2;
                    stop = true;
                    continue;
                }
                const uint32_t cq = lut4(CLASS_LUT, qck[k]), cr = lut4(CLASS_LUT, rck[k]);
                const int32_t wsc = (cq == 5 || cr == 5) ? 0 : ((cq == cr && (cq != 4 || n_eq_n)) ? a.sc.match : a.sc.mismatch);
                h -= wsc;
                emit((eq_by_char ? (qck[k] == rck[k] && qck[k] != 0) : wsc > 0) ? 7 : 8);  // '=' : 'X' by residue equality (A.4)
                i--; j--;
            }
            if (done) break;
            continue;
        }
        const uint32_t nb = nibble(i, j);
        if (state == 1) {  // E: consumes reference only -> 'D'
            if (nb & 2u) { h += open; state = 0; } else h += ext;
            j--;
            emit(op_ref_only);
        } else {           // F: consumes query only -> 'I'
            if (nb & 1u) { h += open; state = 0; } else h += ext;
            i--;
            emit(op_query_only);
        }
    }
    if (cur_op >= 0) {
        const uint32_t v = (cur_len << 4) | (uint32_t)cur_op;
        if (n_runs == 0) first_gen = v;
        ring[n_runs & (FADEHIP_MAX_OPS - 1)] = v;
        n_runs++;
    }

    if (left_range && !(state == 0 && h == 0) && a.early_out && n_runs + ((pad && lq - 1 - f.end_q > 0) ? 1 : 0) > 10) {
        // The path continues before step T0, but it already has more than 10 ops and can only gain more:
        // analysis.d:69-70 rejects it whatever the rest looks like.  Report it like an untraced alignment.
        fadehip_aln o;
        o.read_idx = (int32_t)w.idx;
        o.art = 0;
        o.sw.score = f.score;
        o.sw.end_query = f.end_q;
        o.sw.end_ref = f.end_r;
        o.sw.beg_query = o.sw.beg_ref = -1;
        o.sw.n_ops = 0;
#pragma unroll
        for (int k = 0; k < FADEHIP_MAX_OPS; k++) o.sw.ops[k] = 0;
        const Meta m = a.meta[src];
        o.win_start = m.win_start;
        o.win_len = (int32_t)w.lr;
        o.clip_left = m.clip_left;
        o.clip_right = m.clip_right;
        o.aligned_len = m.aligned_len;
        a.out[w.out] = o;
        return 0u;
    }
    if (left_range && !(state == 0 && h == 0)) return 8u;  // the path continues before step T0: trace again from further back
    uint32_t art_ret = 0;
    fadehip_aln o;
    o.read_idx = (int32_t)w.idx;
    o.art = 0;
    o.sw.score = f.score;
    o.sw.end_query = f.end_q;
    o.sw.end_ref = f.end_r;
    o.sw.beg_query = i + 1;
    o.sw.beg_ref = j + 1;
    // Appendix A.6: [beg_query S] + runs (front = last generated) + [(lq-1-end_query) S]
    int n = 0;
    uint32_t first_op = 0, last_op = 0;
#pragma unroll
    for (int k = 0; k < FADEHIP_MAX_OPS; k++) o.sw.ops[k] = 0;
    if (pad && o.sw.beg_query > 0) {
        first_op = ((uint32_t)o.sw.beg_query << 4) | 4u;
        o.sw.ops[0] = first_op;
        n = 1;
    }
    for (int k = 0; k < n_runs; k++) {  // k-th run of the CIGAR = generated run n_runs-1-k
        if (k < FADEHIP_MAX_OPS) {
            const uint32_t v = ring[(n_runs - 1 - k) & (FADEHIP_MAX_OPS - 1)];
            if (n < FADEHIP_MAX_OPS) o.sw.ops[n] = v;
            if (n == 0) first_op = v;
        }
        n++;
    }
    if (n_runs > 0) last_op = first_gen;
    const int tail = lq - 1 - f.end_q;
    if (pad && tail > 0) {
        last_op = ((uint32_t)tail << 4) | 4u;
        if (n < FADEHIP_MAX_OPS) o.sw.ops[n] = last_op;
        n++;
    }
    o.sw.n_ops = n;
    if (a.early_out && n > 10) {
        // more than 10 ops: analysis.d:69-70 rejects the result whatever the ops are.  Reported like an alignment that was
        // not traced, exactly as the score pass's forced-diagonal shortcut reports such a path (select_one)
        o.sw.beg_query = o.sw.beg_ref = -1;
        o.sw.n_ops = 0;
#pragma unroll
        for (int k = 0; k < FADEHIP_MAX_OPS; k++) o.sw.ops[k] = 0;
    }

    if (a.meta) {
        const Meta m = a.meta[src];
        o.win_start = m.win_start;
        o.win_len = (int32_t)w.lr;
        o.clip_left = m.clip_left;
        o.clip_right = m.clip_right;
        o.aligned_len = m.aligned_len;
        if (a.gate && n >= 1 && n <= 10) {  // analysis.d:69-70
            const uint32_t lead_s = (first_op & 15u) == 4u ? (first_op >> 4) : 0u;
            const uint32_t trail_s = (n > 1 && (last_op & 15u) == 4u) ? (last_op >> 4) : 0u;
            // analysis.d:34,74-80: left clip longer than the floor, last op '=', score > 1.8*clip
            // (float cutoff == integer 5*score > 9*clip), result has leading S only
            if (m.clip_left != 0 && m.clip_left > a.floor_len && (last_op & 15u) == 7u &&
                5 * (int64_t)f.score > 9 * (int64_t)m.clip_left && trail_s == 0 && lead_s != 0)
                o.art |= 1;
            // analysis.d:34,98-104
            if (m.clip_right != 0 && m.clip_right > a.floor_len && (first_op & 15u) == 7u &&
                5 * (int64_t)f.score > 9 * (int64_t)m.clip_right && lead_s == 0 && trail_s != 0)
                o.art |= 2;
            if (o.art) {  // readstatus.d: bit1 art_left, bit2 art_right
                const uint8_t before = a.rs[w.idx];
                a.rs[w.idx] = before | (uint8_t)(o.art << 1);
                art_ret = (uint32_t)o.art | ((before & 32u) ? 4u : 0u);
            }
        }
    } else {
        o.win_start = 0;
        o.win_len = (int32_t)w.lr;
        o.clip_left = o.clip_right = o.aligned_len = 0;
    }
    a.out[w.out] = o;
    return art_ret;
}

// stats.d:45-54, the artifact part, from the per-alignment codes of a wave: per-wave popcounts, one atomic per wave and
// non-zero counter
__device__ __forceinline__ void add_artifact_stats(unsigned long long *stats, uint32_t r, uint32_t part) {
    const unsigned long long m_art = __ballot(r & 3u), m_sup = __ballot((r & 3u) && (r & 4u)), m_l = __ballot(r & 1u), m_r = __ballot(r & 2u);
    if ((threadIdx.x & 63) == 0 && m_art) {
        unsigned long long *st = stats + 8 * (part % STAT_PARTS);
        atomicAdd(&st[4], (unsigned long long)__popcll(m_art));
        if (m_sup) atomicAdd(&st[3], (unsigned long long)__popcll(m_sup));
        if (m_l) atomicAdd(&st[6], (unsigned long long)__popcll(m_l));
        if (m_r) atomicAdd(&st[7], (unsigned long long)__popcll(m_r));
    }
}

// ---------------------------------------------------------------- pass-2 selection
// After pass 1 every alignment has score and end cell.  Level 2: only alignments that can still become an
// artifact call need a CIGAR — left: clip > floor, 5*score > 9*clip and the end cell in the last query row
// (analysis.d:74-80 needs the last op to be '=' with no trailing S); right: clip > floor, 5*score > 9*clip and
// end cell above the last row (analysis.d:102-104 needs a trailing S).  Everything else gets its record
// here with n_ops = 0 ("not traced").  Level 1 (meta == nullptr) traces everything.

// One alignment of the score pass, right after its wave has its score and end cell (a lane per alignment: the dependent
// loads of the diagonal walk hide behind the sweeps of the SIMD's other waves; as a kernel of its own the selection
// waited for wave slots next to the other slots' score passes).  Returns the artifact bits it set (bit0 left, bit1
// right) | 4 if the read is supplementary.
__device__ __forceinline__ uint32_t select_one(const SelArgs &a, const int item, const Work &w, const Fwd &f, const int R,
                                               const uint8_t *q_nib, const uint8_t *r_nib, const uint32_t rules) {
    int b = -1;
    uint32_t art_ret = 0;
    Cand c;
    c.src = 0;
    c.c0 = 0;
    {
        bool cand = true;
        if (a.meta && !a.trace_all) {
            const Meta m = a.meta[item];
            const bool left = m.clip_left > a.floor_len && 5 * (int64_t)f.score > 9 * (int64_t)m.clip_left &&
                              f.end_q == (int32_t)w.lq - 1;
            const bool right = m.clip_right > a.floor_len && 5 * (int64_t)f.score > 9 * (int64_t)m.clip_right &&
                               f.end_q < (int32_t)w.lq - 1;
            // (without A.6's soft-clip padding no result CIGAR has an S op and analysis.d:78-80 / 102-104 never hold)
            cand = (left || right) && rule(rules, FADEHIP_RULE_PAD_SOFTCLIP);
            if (!cand) {
                fadehip_aln o;
                o.read_idx = (int32_t)w.idx;
                o.art = 0;
                o.win_start = m.win_start;
                o.win_len = (int32_t)w.lr;
                o.clip_left = m.clip_left;
                o.clip_right = m.clip_right;
                o.aligned_len = m.aligned_len;
                o.sw.score = f.score;
                o.sw.end_query = f.end_q;
                o.sw.end_ref = f.end_r;
                o.sw.beg_query = -1;
                o.sw.beg_ref = -1;
                o.sw.n_ops = 0;
                for (int k = 0; k < FADEHIP_MAX_OPS; k++) o.sw.ops[k] = 0;
                a.out[w.out] = o;
            }
        }
        // Forced-diagonal shortcut.  Walk back along the end cell's diagonal d_0 = (end_q, end_r), d_1, ... with
        // P_0 = score, P_{k+1} = P_k - W(d_k).  If the partial sums stay positive and hit exactly 0 at some d_L (a cell,
        // or the matrix border, where H is 0 by definition), the traceback is forced: H(d_k) >= P_k because the diagonal
        // chains up from H(d_L) >= 0 (H >= D everywhere), and H(d_k) <= P_k because any excess would chain up to more than
        // the given H(d_0) = score; so every cell equals its diagonal predecessor + W — the direction the traceback tries
        // first under either A.4 priority — and it stops at d_L.  CIGAR = [beg_query S] runs of = / X [tail S] with no
        // DP re-computation.  This is the planted artifact (an exact or near-exact reverse-complement copy), by far the
        // most common candidate; gapped paths, and the rare ones with more runs than an op array holds, go to pass 2.
        if (cand && a.match > 0 && f.score > 0) {
            const bool rcq = w.flags & 1u;
            const int lq = (int)w.lq;
            const bool n_eq_n = rule(rules, FADEHIP_RULE_N_MATCHES_N), eq_by_char = rule(rules, FADEHIP_RULE_EQ_BY_CHAR);
            const bool pad = rule(rules, FADEHIP_RULE_PAD_SOFTCLIP);
            // Up to nine runs of = / X are kept, in registers (r0 = the run nearest the start of the alignment): with the
            // two S pads that is every CIGAR analysis.d:69 can still accept (<= 10 ops), and a level-2 result with more ops
            // than that is rejected whatever they are, so it is reported as "not traced" right here.  Only gapped paths (and,
            // at level 1 / trace_all, forced diagonals of more than nine runs) go to pass 2.  Nothing here is indexed
            // dynamically, so the score pass needs no scratch memory.
            constexpr int MAX_RUNS = 9;
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, r8 = 0;
            int n_runs = 0, cur_op = -1;
            uint32_t cur_len = 0;
            int P = f.score, L = 0;
            bool ok = false, give_up = false;
            const int kmax = min(f.end_q, f.end_r) + 1;  // cells of the diagonal inside the matrix
            auto push_run = [&](uint32_t v) {
                r8 = r7; r7 = r6; r6 = r5; r5 = r4; r4 = r3; r3 = r2; r2 = r1; r1 = r0; r0 = v;
                n_runs++;
            };
            // 32 cells per round trip: the loads of four 8-cell pieces are issued together
            for (int k0 = 0; k0 < kmax && !ok && !give_up; k0 += 32) {
                Nib8 qw[4], rw[4];
                uint64_t qlo[4], rlo[4];
                int nv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = k0 + 8 * u;
                    nv[u] = max(0, min(8, kmax - kk));
                    qlo[u] = rlo[u] = 0;
                    qw[u].word = rw[u].word = 0;
                    qw[u].byte0 = rw[u].byte0 = 0;
                    if (nv[u] > 0) {
                        qlo[u] = rcq ? (uint64_t)w.q_base + (uint32_t)(lq - 1 - f.end_q + kk)
                                     : (uint64_t)w.q_base + (uint32_t)(f.end_q - kk - (nv[u] - 1));
                        rlo[u] = w.r_base + (uint64_t)(f.end_r - kk - (nv[u] - 1));
                        qw[u] = load_nib8(q_nib, qlo[u]);
                        rw[u] = load_nib8(r_nib, rlo[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (k < nv[u] && !ok && !give_up) {
                            const uint32_t qraw = nib8_at(qw[u], rcq ? qlo[u] + (uint32_t)k : qlo[u] + (uint32_t)(nv[u] - 1 - k));
                            const uint32_t qc = rcq ? lut4(COMP_LUT, qraw) : qraw, rc = nib8_at(rw[u], rlo[u] + (uint32_t)(nv[u] - 1 - k));
                            const uint32_t cq = lut4(CLASS_LUT, qc), cr = lut4(CLASS_LUT, rc);
                            const int wsc = (cq == 5 || cr == 5) ? 0 : ((cq == cr && (cq != 4 || n_eq_n)) ? a.match : a.mismatch);
                            const int op = (eq_by_char ? (qc == rc && qc != 0) : wsc > 0) ? 7 : 8;
                            if (op == cur_op) cur_len++;
                            else {
                                if (cur_op >= 0) push_run((cur_len << 4) | (uint32_t)cur_op);
                                cur_op = op;
                                cur_len = 1;
                            }
                            P -= wsc;
                            L = k0 + 8 * u + k + 1;
                            if (P == 0) ok = true;
                            else if (P < 0) give_up = true;
                        }
                    }
                }
            }
            bool too_many = false;  // a forced diagonal whose CIGAR has more than 10 ops: analysis.d:69-70 rejects it
            if (ok) {
                push_run((cur_len << 4) | (uint32_t)cur_op);
                const int lead0 = f.end_q - L + 1, tail0 = lq - 1 - f.end_q;
                const int n_all = n_runs + ((pad && lead0 > 0) ? 1 : 0) + ((pad && tail0 > 0) ? 1 : 0);
                too_many = n_all > 10 && a.meta && a.gate && !a.trace_all;
                if (n_runs > MAX_RUNS && !too_many) ok = false;
            }
            if (ok) {
                cand = false;
                fadehip_aln o;
                o.read_idx = (int32_t)w.idx;
                o.art = 0;
                o.sw.score = f.score;
                o.sw.end_query = f.end_q;
                o.sw.end_ref = f.end_r;
#pragma unroll
                for (int k = 0; k < FADEHIP_MAX_OPS; k++) o.sw.ops[k] = 0;
                if (too_many) {  // reported like an alignment that was not traced (as traceback_path does for such a path)
                    o.sw.beg_query = o.sw.beg_ref = -1;
                    o.sw.n_ops = 0;
                    const Meta m = a.meta[item];
                    o.win_start = m.win_start;
                    o.win_len = (int32_t)w.lr;
                    o.clip_left = m.clip_left;
                    o.clip_right = m.clip_right;
                    o.aligned_len = m.aligned_len;
                    a.out[w.out] = o;
                } else {
                o.sw.beg_query = f.end_q - L + 1;
                o.sw.beg_ref = f.end_r - L + 1;
                const int lead = o.sw.beg_query, tail = lq - 1 - f.end_q;
                // [lead S] r0 .. r(n_runs-1) [tail S]: the ops go straight to the result entry in memory (a dynamic index
                // into memory costs an address, into registers it would cost scratch); the rest of the record follows below
                uint32_t *const gops = a.out[w.out].sw.ops;
                int n = 0;
                const uint32_t lead_op = ((uint32_t)lead << 4) | 4u, tail_op = ((uint32_t)tail << 4) | 4u;
                if (pad && lead > 0) gops[n++] = lead_op;
                if (n_runs > 0) gops[n + 0] = r0;
                if (n_runs > 1) gops[n + 1] = r1;
                if (n_runs > 2) gops[n + 2] = r2;
                if (n_runs > 3) gops[n + 3] = r3;
                if (n_runs > 4) gops[n + 4] = r4;
                if (n_runs > 5) gops[n + 5] = r5;
                if (n_runs > 6) gops[n + 6] = r6;
                if (n_runs > 7) gops[n + 7] = r7;
                if (n_runs > 8) gops[n + 8] = r8;
                n += n_runs;
                if (pad && tail > 0) gops[n++] = tail_op;
#pragma unroll
                for (int k = 2; k < FADEHIP_MAX_OPS; k++)
                    if (k >= n) gops[k] = 0;
                if (n < 2) gops[1] = 0;
                o.sw.n_ops = n;
                const uint32_t last_run = n_runs == 1 ? r0 : n_runs == 2 ? r1 : n_runs == 3 ? r2 : n_runs == 4 ? r3 : n_runs == 5 ? r4 :
                                          n_runs == 6 ? r5 : n_runs == 7 ? r6 : n_runs == 8 ? r7 : r8;
                const uint32_t first_op = (pad && lead > 0) ? lead_op : r0;
                const uint32_t last_op = (pad && tail > 0) ? tail_op : last_run;
                if (a.meta) {
                    const Meta m = a.meta[item];
                    o.win_start = m.win_start;
                    o.win_len = (int32_t)w.lr;
                    o.clip_left = m.clip_left;
                    o.clip_right = m.clip_right;
                    o.aligned_len = m.aligned_len;
                    if (a.gate && n <= 10) {  // analysis.d:69-70
                        const uint32_t lead_s = (first_op & 15u) == 4u ? (first_op >> 4) : 0u;
                        const uint32_t trail_s = (n > 1 && (last_op & 15u) == 4u) ? (last_op >> 4) : 0u;
                        // analysis.d:74-80 (last op '=', leading S only) / 98-104 (first op '=', trailing S only)
                        if (m.clip_left != 0 && m.clip_left > a.floor_len && (last_op & 15u) == 7u && trail_s == 0 && lead_s != 0 &&
                            5 * (int64_t)f.score > 9 * (int64_t)m.clip_left)
                            o.art |= 1;
                        if (m.clip_right != 0 && m.clip_right > a.floor_len && (first_op & 15u) == 7u && lead_s == 0 && trail_s != 0 &&
                            5 * (int64_t)f.score > 9 * (int64_t)m.clip_right)
                            o.art |= 2;
                        if (o.art) {
                            const uint8_t before = a.rs[w.idx];
                            a.rs[w.idx] = before | (uint8_t)(o.art << 1);
                            art_ret = (uint32_t)o.art | ((before & 32u) ? 4u : 0u);
                        }
                    }
                } else {
                    o.win_start = 0;
                    o.win_len = (int32_t)w.lr;
                    o.clip_left = o.clip_right = o.aligned_len = 0;
                }
                // everything but the ops (already in place)
                fadehip_aln *const dst = &a.out[w.out];
                dst->read_idx = o.read_idx;
                dst->art = o.art;
                dst->win_start = o.win_start;
                dst->win_len = o.win_len;
                dst->clip_left = o.clip_left;
                dst->clip_right = o.clip_right;
                dst->aligned_len = o.aligned_len;
                dst->sw.score = o.sw.score;
                dst->sw.end_query = o.sw.end_query;
                dst->sw.end_ref = o.sw.end_ref;
                dst->sw.beg_query = o.sw.beg_query;
                dst->sw.beg_ref = o.sw.beg_ref;
                dst->sw.n_ops = o.sw.n_ops;
                }
            }
        }
        if (cand) {
            // sweep steps the path is expected to span: a cell (i, j) is computed at step j + i / R.  Columns:
            // score/2 for clean matches, a quarter more for mismatches, + slack; rows ~ columns.  Any value is
            // correct (a path that leaves the traced steps is re-run from step 0), this one is cheap.
            const int span_cols = a.span_slack >= 0 ? f.score / 2 + f.score / 8 + a.span_slack : 1;
            const int span = span_cols + span_cols / R + 2;
            const int t_end = f.end_r + f.end_q / R;
            int c0 = t_end + 1 - span;
            c0 = (c0 < CK_COLS || a.no_ckpt) ? 0 : (c0 & ~(CK_COLS - 1));
            const int steps = t_end + 1 - c0;
            b = 0;
            while (steps > bucket_cols(b)) b++;
            c.src = (uint32_t)item;
            c.c0 = (uint32_t)c0;
            const uint32_t slot = atomicAdd(&a.bucket_n[b], 1u);
            if (slot < a.cap) a.cand[(uint64_t)b * a.cap + slot] = c;
        }
    }
    (void)b;
    return art_ret;
}

// ---------------------------------------------------------------- forward SW, packed int16 (two alignments per 16-lane row)
// Same recurrence as sw_forward_kernel with every DP value scaled by 8 and held as a pair of int16:
// low half = alignment A, high half = alignment B of the same 16-lane row, so one v_pk_* instruction
// advances two cells.  The scale makes every positive difference >= 8, which turns the four trace
// flags into v_pk_min_u16(diff, weight) with weights 8,4,2,1 — a nibble per cell by plain addition.
// One wave = 8 alignments ("octet").  Trace: [block of 4 steps][R dwords][64 lanes], each dword =
// 4 cell-pairs of one diagonal (see trace_nibble_pk), low / high half for alignment A / B.
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v as_s2(uint32_t x) { return __builtin_bit_cast(s2v, x); }
__device__ __forceinline__ u2v as_u2(uint32_t x) { return __builtin_bit_cast(u2v, x); }
__device__ __forceinline__ uint32_t as_u32(s2v x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_u32(u2v x) { return __builtin_bit_cast(uint32_t, x); }
// flag arithmetic goes through asm so that hipcc keeps it packed (it otherwise rewrites min(x-y,1)
// into per-half compares + selects)
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <int K>
__device__ __forceinline__ uint32_t pk_min_k(uint32_t a) {
    uint32_t d;
    asm("v_pk_min_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "n"(K));
    return d;
}
__device__ __forceinline__ uint32_t pk_shl4_add(uint32_t acc, uint32_t m) {
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, 16, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(acc), "v"(m));
    return d;
}

template <int SH>
__device__ __forceinline__ uint32_t lshl_add(uint32_t a, uint32_t b) {  // (a << SH) + b in one VALU op
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(SH), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {  // (a & mask) | c
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(mask), "v"(c));
    return d;
}

// max(a, b, c) per 16-bit half for NON-NEGATIVE int16 inputs below 0x7c00, in one instruction: gfx950's packed
// three-input f16 maximum.  A non-negative int16 read as f16 (sign 0, denormals preserved: the kernels run with
// float_denorm_mode_16_64 = 3) orders exactly like the integer, and 0x7c00 (Inf) is never reached: DP values are
// 8 * score <= 8 * 2 * 512 + 8 * open.  There is no packed integer max3.
__device__ __forceinline__ uint32_t pk_max3_nonneg(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_mad4(uint32_t h, uint32_t c) {  // 4*h + c per 16-bit half
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, 4, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_mad8(uint32_t h, uint32_t c) {  // 8*h + c per 16-bit half, c wave-uniform (SGPR)
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, 8, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "s"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_mul_ffff(uint32_t a) {  // 0/1 per half -> 0x0000/0xffff
    uint32_t d;
    asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "s"(0xffffu));
    return d;
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {  // (a & mask) | (b & ~mask)
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}

constexpr int PK_SCALE = 8;  // must stay 8: the shifts below are log2(8) and log2(8) + 16

// MODE 0: single pass — full window, trace to memory, end cell
// MODE 1: pass 1 — score and end cell only, plus a snapshot of the wave state every CK_COLS steps
// MODE 2: pass 2 — traced re-computation of sweep steps [T0, end_ref + lane(end_query)] of a candidate,
//         resumed from the snapshot taken after step T0-1; no end-cell tracking
// MODE 3: MODE 2 with the A.4 rule switches read at run time (FADEHIP_RULE_HDIR_DIAG_F_E, _GAP_TIE_EXTENDS off):
//         a few more instructions per cell, used only when a rule differs from the default
// waves per SIMD asked of the register allocator for the score pass: 4 up to R = 10 (128 VGPRs, no spills), 3 up to R = 16
// (168; R = 16 spills ~100 registers outside its sweep and is still 6 % faster than at 2 waves), 2 beyond (R = 20 / 24:
// +27 % / +30 % over the unconstrained allocation, which took 256 VGPRs and one wave).  The traced pass is latency-bound:
// 3 or 4 waves per SIMD (40 / 85 spilled registers) changed nothing measurable, it is left alone.
__host__ __device__ constexpr int pk_min_waves(int R, int MODE, int LG = 16, int WV = 0) { return WV ? WV : MODE != 1 ? 1 : (LG == 8 ? (R <= 13 ? 4 : 3) : (R <= 10 ? 4 : (R <= 16 ? 3 : 2))); }

// LONGW: the launch may hold windows longer than one staged chunk (CH_COLS columns); the sweep then re-stages at chunk
// boundaries.  A kernel of its own, so that the ordinary launch keeps its registers (the chunk loop cost the 150-base
// score pass 29 spilled registers when it was a run-time branch).
// LG: lanes per alignment pair.  16 (everything but one A/B variant): four pairs, eight alignments per wavefront.  8
// (FADEHIP_SCORE_G8, MODE 1 only): eight pairs, sixteen alignments per wavefront, a lane owning R rows of 8 R
// (19 x 8 = 152 rows for 150-base reads instead of 10 x 16 = 160; the skew is 7 steps instead of 15).  DPP row_shr:1
// still shifts within 16-lane rows, so lane 8 of a row is given the DP boundary by hand.
// PERSIST (FADEHIP_SCORE_PERSIST, MODE 1 only, the other A/B variant): the score pass as a persistent launch whose waves draw
// octets from a ticket.  Both variants are instantiations of their own: the ordinary score pass keeps its registers.
template <int R, int MODE, bool LONGW = false, int LG = 16, bool PERSIST = false, int WV = 0>
__global__ __launch_bounds__(64, pk_min_waves(R, MODE, LG, WV)) void sw_pk_kernel(SwArgs a) {
    static_assert(LG == 16 || (LG == 8 && MODE == 1 && !LONGW), "eight-lane groups exist for the score pass only");
    static_assert(!PERSIST || MODE == 1, "the persistent variant is a score pass");
    extern __shared__ __align__(16) uint8_t lds[];
    const int lane = threadIdx.x;
    constexpr int NG = 64 / LG, NPW = 2 * NG;  // groups, alignments per wavefront
    const int g = lane / LG, lig = lane & (LG - 1);
    const uint32_t lig0m = (LG == 8 && lig == 0) ? 0xffffffffu : 0u;  // LG = 8: lanes that take the DP boundary instead of their neighbour's values
    (void)lig0m;
    // ---- score table in LDS, indexed by the PAIR of column classes (cA, cB) of a lane's two alignments: entry
    // (7 cA + cB) = 16 bytes {rowA.lo, rowB.lo, rowA.hi, rowB.hi}, row c = 8 bytes, byte q = 8 * W'(query class q,
    // column class c).  The class word that travels along the lanes IS the entry's byte offset, so a step costs one
    // LDS read and no address arithmetic; every row then picks its bytes with v_perm_b32 (selector = query class).
    __shared__ uint4 wtab[49];
    if (lane < 49) {
        const int cA = lane / 7, cB = lane - 7 * cA;
        uint32_t alo = 0, ahi = 0, blo = 0, bhi = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            alo |= (((a.sc.prof[q] >> (4 * cA)) & 15u) * PK_SCALE) << (8 * q);
            blo |= (((a.sc.prof[q] >> (4 * cB)) & 15u) * PK_SCALE) << (8 * q);
        }
#pragma unroll
        for (int q = 4; q < 7; q++) {
            ahi |= (((a.sc.prof[q] >> (4 * cA)) & 15u) * PK_SCALE) << (8 * (q - 4));
            bhi |= (((a.sc.prof[q] >> (4 * cB)) & 15u) * PK_SCALE) << (8 * (q - 4));
        }
        wtab[lane] = make_uint4(alo, blo, ahi, bhi);
    }
    // items of this launch: the host's bound, clamped by what the gate put on the list
    int n_items = a.n_items;
    constexpr bool P2 = (MODE >= 2);
    if constexpr (!P2) {
        if (a.count_dev) n_items = min(n_items, max(0, (int)*a.count_dev - (int)a.item_base));
    }
    // MODE 2 is a persistent launch: its size comes from an upper bound, the number of octets from the device-side
    // table, so every wave draws octets from a ticket counter until the table is exhausted (an exit every wave reaches).
    // MODE 0 / 1: one octet per wave, blockIdx.x.
    for (bool first_round = true;; first_round = false) {
    int oct = blockIdx.x;
    uint32_t srcA = 0, srcB = 0, c0A = 0, c0B = 0;
    bool haveA = false, haveB = false;  // MODE 2: this half holds a candidate that still has to be traced
    if constexpr (P2) {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(a.ticket, 1u);
        oct = (int)__builtin_amdgcn_readfirstlane(tk);
        // the table, from the bucket counts (wave-uniform: scalar loads)
        int first = 0, b = -1, local = 0, n_b = 0;
        uint32_t total = 0;
#pragma unroll
        for (int k = 0; k < NUM_BUCKETS; k++) {
            const uint32_t cnt = min(a.bucket_n[NUM_BUCKETS - 1 - k], a.cand_cap);  // longest bucket first (a bucket never holds more than its capacity)
            const int octs_k = (int)((cnt + 7) / 8);
            if (b < 0 && oct < first + octs_k) { b = NUM_BUCKETS - 1 - k; local = oct - first; n_b = (int)cnt; }
            first += octs_k;
            total += cnt;
        }
        if (oct == 0 && lane == 0 && total && a.cand_total) atomicAdd(a.cand_total, (unsigned long long)total);
        if (b < 0) break;  // oct >= number of octets: nothing left
        if (!first_round) __syncthreads();  // the previous octet's window is no longer read
        const Cand *cl = a.cand + (uint64_t)b * a.cand_cap;
        const int kA = local * 8 + g * 2, kB = kA + 1;
        if (kA < n_b) { const Cand c = cl[kA]; srcA = c.src; c0A = c.c0; haveA = true; }
        if (kB < n_b) { const Cand c = cl[kB]; srcB = c.src; c0B = c.c0; haveB = true; }
    } else {
        if constexpr (PERSIST) {
            uint32_t tk = 0;
            if (lane == 0) tk = atomicAdd(a.ticket, 1u);
            oct = (int)__builtin_amdgcn_readfirstlane(tk);
            if (oct * NPW >= n_items) break;
            if (!first_round) __syncthreads();  // the previous octet's window is no longer read
        } else {
            if (!first_round || oct * NPW >= n_items) break;
        }
    }
    // MODE 2: an octet is traced, its tracebacks walked, and the members whose path left the traced steps are traced
    // again from further back (4 snapshots, then step 0, where no path can leave) — all by the wave that drew it
    for (int attempt = 0;; attempt++) {
    const int itemA = oct * NPW + g * 2, itemB = itemA + 1;
    Work wa, wb;
    wa.r_base = wb.r_base = 0; wa.q_base = wb.q_base = 0; wa.lq = wb.lq = 0; wa.lr = wb.lr = 0;
    wa.idx = wb.idx = 0; wa.flags = wb.flags = 0; wa.out = wb.out = 0;
    int stepsA = 0, stepsB = 0;  // MODE 2: sweep steps to run for each half
    const uint64_t trace_off = (uint64_t)(P2 ? (int)blockIdx.x : oct) * a.quad_stride;
    if constexpr (P2) {
        // lane 0 feeds column T0 + tau at relative step tau: the staged "window" is columns [T0, end_ref]
        if (haveA) {
            wa = a.work[srcA];
            const Fwd f = a.fwd[srcA];
            stepsA = f.end_r + f.end_q / R - (int)c0A + 1;
            wa.lr = (uint32_t)max(0, f.end_r + 1 - (int)c0A);
            wa.r_base += c0A;
        }
        if (haveB) {
            wb = a.work[srcB];
            const Fwd f = a.fwd[srcB];
            stepsB = f.end_r + f.end_q / R - (int)c0B + 1;
            wb.lr = (uint32_t)max(0, f.end_r + 1 - (int)c0B);
            wb.r_base += c0B;
        }
    } else {
        if (itemA < n_items) wa = a.work[itemA];
        if (itemB < n_items) wb = a.work[itemB];
    }
    const int lqA = (int)wa.lq, lrA = (int)wa.lr, lqB = (int)wb.lq, lrB = (int)wb.lr;
    int mx = P2 ? max(stepsA, stepsB) : max(lrA, lrB) + LG - 1;
    int maxst = __builtin_amdgcn_readlane(mx, 0);
#pragma unroll
    for (int k = 1; k < NG; k++) maxst = max(maxst, __builtin_amdgcn_readlane(mx, k * LG));
    const int n_blocks = (maxst + 3) >> 2;

    // ---- stage both windows: 16 bits per column = (7 classA + classB) * 16, the byte offset into wtab
    // Eight columns per lane and trip: the two aligned dwords that hold them (load_nib8) instead of eight byte loads,
    // and one 16-byte LDS store (ref_stride is a multiple of 16 and n_cols of 4; the last store may run up to 4
    // columns into the slack the host adds to ref_stride).
    uint16_t *lref = reinterpret_cast<uint16_t *>(lds + g * a.ref_stride);
    const int n_cols = n_blocks * 4;
    // columns [c0, c0 + CH_COLS + 8) of the window go to lref[0 ..): a chunk and the first columns of the next (the sweep
    // fetches a block's class words one block ahead)
    auto stage = [&](const int c0) __attribute__((always_inline)) {
        const int c1 = min(n_cols, c0 + CH_COLS + 8);
        for (int k0 = c0 + lig * 8; k0 < c1; k0 += 8 * LG) {
            Nib8 na, nb;
            na.word = nb.word = 0; na.byte0 = nb.byte0 = 0;
            const uint64_t ra0 = wa.r_base + (uint64_t)k0, rb0 = wb.r_base + (uint64_t)k0;
            if (k0 < lrA) na = load_nib8(a.r_nib, ra0);
            if (k0 < lrB) nb = load_nib8(a.r_nib, rb0);
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t ca = PAD_CLASS, cb = PAD_CLASS;
                if (k0 + j < lrA) ca = lut4(CLASS_LUT, nib8_at(na, ra0 + (uint32_t)j));
                if (k0 + j < lrB) cb = lut4(CLASS_LUT, nib8_at(nb, rb0 + (uint32_t)j));
                const uint32_t e = (7u * ca + cb) * 16u;
                if (j & 1) o[j >> 1] |= e << 16;
                else o[j >> 1] = e;
            }
            *reinterpret_cast<uint4 *>(lref + (k0 - c0)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    stage(0);
    // query classes per row (A | B << 16); `special` = some real row is N or a wildcard (class >= 4).  A lane's R rows
    // are R consecutive bases of the (reverse-complemented) query: eight per pair of aligned dwords.
    uint32_t qcls[R];
    bool special = false;
    {
        constexpr int NCH = (R + 7) / 8;
        const int row0 = lig * R;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            // rows row0 + 8 ch .. + 7 are query bases (rc: lq - 1 - row, descending) of A and of B
            const int rlo = row0 + 8 * ch;
            const int nA = max(0, min(8, min(R - 8 * ch, lqA - rlo))), nB = max(0, min(8, min(R - 8 * ch, lqB - rlo)));
            const bool rcA = wa.flags & 1u, rcB = wb.flags & 1u;
            // first (lowest) nibble index of the piece
            const uint64_t qa0 = rcA ? (uint64_t)wa.q_base + (uint32_t)(lqA - rlo - nA) : (uint64_t)wa.q_base + (uint32_t)rlo;
            const uint64_t qb0 = rcB ? (uint64_t)wb.q_base + (uint32_t)(lqB - rlo - nB) : (uint64_t)wb.q_base + (uint32_t)rlo;
            Nib8 na, nb;
            na.word = nb.word = 0; na.byte0 = nb.byte0 = 0;
            if (nA > 0) na = load_nib8(a.q_nib, qa0);
            if (nB > 0) nb = load_nib8(a.q_nib, qb0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 8 * ch + j;
                if (r < R) {
                    uint32_t qa = PAD_CLASS, qb = PAD_CLASS;
                    if (j < nA) {
                        const uint32_t raw = nib8_at(na, rcA ? qa0 + (uint32_t)(nA - 1 - j) : qa0 + (uint32_t)j);
                        qa = lut4(CLASS_LUT, rcA ? lut4(COMP_LUT, raw) : raw);
                    }
                    if (j < nB) {
                        const uint32_t raw = nib8_at(nb, rcB ? qb0 + (uint32_t)(nB - 1 - j) : qb0 + (uint32_t)j);
                        qb = lut4(CLASS_LUT, rcB ? lut4(COMP_LUT, raw) : raw);
                    }
                    special |= qa == 4u || qa == 5u || qb == 4u || qb == 5u;
                    qcls[r] = qa | (qb << 16);
                }
            }
        }
    }

    // End-cell tracking.  MODE 0: a 32-bit key (H, 0xffff - t) per row and alignment.  MODE 1: one packed 16-bit
    // key per row PAIR, 4*H8 + (31 - t % 32) = (32*H, position inside the current 32-step window), folded into
    // (GH, GT) = (best 32*H so far, its step) at every window end: 2 + 0.3 instructions per cell pair instead of 3.
    // MODE 1, row classes up to 24: ONE key per pair of rows, kept with v_pk_maximum3_f16 (1.5 instead of 2 instructions
    // per cell pair), which orders the 16-bit patterns only below the f16 infinity 0x7c00.  Row classes up to 14 (scores
    // <= 448): 8*H8 + 2*(31 - t % 32) + (row even) = 64 * score + 6 low bits, 32-step windows and half the fold work.
    // Row classes 16 .. 24 (scores <= 768): 4*H8 + 2*(15 - t % 16) + (row even) = 32 * score + 5 low bits, 16-step
    // windows (the fold work of the unpaired key).  R = 32 (scores up to 1024) keeps one key per row.
    constexpr int ROWS = R * LG;                                   // query rows of the kernel: what bounds the score
    constexpr bool PAIRKEY = (MODE == 1) && (ROWS <= 384);
    constexpr int KW = (PAIRKEY && ROWS > 224) ? 16 : 32;          // steps per key window
    constexpr int KLOW = PAIRKEY ? (KW == 32 ? 6 : 5) : 5;          // key bits below the score
    constexpr uint32_t KLOWM = ((1u << KLOW) - 1u) * 0x10001u;      // ... as a mask over both halves
    uint32_t Hl[R], Eh[R], bestA[R], bestB[R];  // MODE 1 reuses bestA as the window key and bestB as GH
    uint32_t GT[R];
#pragma unroll
    for (int r = 0; r < R; r++) { Hl[r] = 0; Eh[r] = 0; bestA[r] = 0; bestB[r] = 0; GT[r] = 0; }
    uint32_t hu_out = 0, fu_out = 0, hu_prev = 0;
    uint32_t rc = (7 * PAD_CLASS + PAD_CLASS) * 16;
    if constexpr (P2) {
        // resume: the wave state pass 1 snapshotted after step T0-1, per half from that half's own octet
        constexpr int CKD = ck_dwords(R);
        const uint32_t laneA = ((srcA >> 1) & 3u) * 16u + (uint32_t)lig, laneB = ((srcB >> 1) & 3u) * 16u + (uint32_t)lig;
        const uint32_t snA = c0A ? (c0A >> CK_SHIFT) - 1 : 0, snB = c0B ? (c0B >> CK_SHIFT) - 1 : 0;
        const uint32_t *ckA = a.ckpt + (uint64_t)(srcA >> 3) * a.ck_stride + (uint64_t)snA * (CKD * 64) + laneA;
        const uint32_t *ckB = a.ckpt + (uint64_t)(srcB >> 3) * a.ck_stride + (uint64_t)snB * (CKD * 64) + laneB;
        const int shA = (srcA & 1u) ? 16 : 0, shB = (srcB & 1u) ? 16 : 0;
        auto pick = [&](int k) -> uint32_t {
            uint32_t va = 0, vb = 0;
            if (haveA && c0A) va = (ckA[k * 64] >> shA) & 0xffffu;
            if (haveB && c0B) vb = (ckB[k * 64] >> shB) & 0xffffu;
            return va | (vb << 16);
        };
#pragma unroll
        for (int r = 0; r < R; r++) {
            Hl[r] = pick(r);
            Eh[r] = pick(R + r);
        }
        hu_out = pick(2 * R);
        fu_out = pick(2 * R + 1);
        hu_prev = pick(2 * R + 2);
        // the class register of a pass-1 octet holds (7 classA + classB) * 16 of ITS two alignments: take the one
        // this half resumes and pair it with the other half's
        uint32_t ra = PAD_CLASS, rb = PAD_CLASS;
        if (haveA && c0A) {
            const uint32_t pr = (ckA[(2 * R + 3) * 64] & 0xffffu) >> 4;
            ra = (srcA & 1u) ? pr % 7u : pr / 7u;
        }
        if (haveB && c0B) {
            const uint32_t pr = (ckB[(2 * R + 3) * 64] & 0xffffu) >> 4;
            rb = (srcB & 1u) ? pr % 7u : pr / 7u;
        }
        rc = (7u * ra + rb) * 16u;
    }
    __syncthreads();
    const uint32_t ext8 = (uint32_t)(a.sc.ext * PK_SCALE) * 0x10001u;
    const uint32_t open8 = (uint32_t)(a.sc.open * PK_SCALE) * 0x10001u;
    uint32_t *tq = a.trace + trace_off + lane;
    uint32_t *ckw = a.ckpt + (uint64_t)oct * a.ck_stride + lane;
    const uint32_t himask = __builtin_amdgcn_readfirstlane(0xffff0000u);
    // rule switches (wave-uniform; the default path never reads them inside the sweep)
    const bool hdir_f_first = rule(a.sc.rules, FADEHIP_RULE_HDIR_DIAG_F_E), tie_extends = rule(a.sc.rules, FADEHIP_RULE_GAP_TIE_EXTENDS);
    const uint32_t tie_flip2 = tie_extends ? 0u : 0x00020002u, tie_flip1 = tie_extends ? 0u : 0x00010001u;
    // A.3 with the rule off: ties go to the smallest query index, then the smallest ref index.  The per-row keys already
    // hold each row's first column; the paired key ranks (H, even row, position) instead of (H, position, even row)
    const bool end_min_ref = rule(a.sc.rules, FADEHIP_RULE_END_MIN_REF_THEN_QUERY);
    (void)hdir_f_first; (void)tie_flip2; (void)tie_flip1; (void)end_min_ref;

    // The sweep exists twice.  FAST (no N / wildcard in either query of the octet, i.e. nearly always): the four
    // A,C,G,T bytes of the two column rows sit in one register pair, so ONE v_perm_b32 yields both halves of the
    // pair's 8*W' (pad rows select the constant 0) and the diagonal term costs 2 instructions.  General: one
    // v_perm_b32 per alignment over the full 7-class rows plus v_add3_u32 (3 instructions).
    // MODE 1 runs the blocks in groups of 8 (= CK_COLS steps) and snapshots between groups, so that the hot
    // loop has the same shape in every mode
    auto sweep = [&](auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    uint32_t selA[R], selB[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t qa = qcls[r] & 0xffu, qb = qcls[r] >> 16;
        if constexpr (FAST) {
            selA[r] = 0x0c000c00u | (qa == PAD_CLASS ? 0x0cu : qa) | ((qb == PAD_CLASS ? 0x0cu : qb + 4u) << 16);
            selB[r] = 0;
        } else {
            selA[r] = 0x0c0c0c00u | qa;
            selB[r] = 0x0c000c0cu | (qb << 16);
        }
    }
    const uint8_t *wt = reinterpret_cast<const uint8_t *>(wtab);
    constexpr int GROUP = (MODE == 1) ? KW / 4 : (1 << 30);  // blocks (of 4 steps) per key window
    // (score pass: a block's four class words are fetched a block ahead, so that their LDS latency runs beside the block
    // before: +0.5-1 % in tools/stream_probe.py, within noise in bench.py, no register more.  The traced passes have no
    // registers to spare and are not issue-bound.)
    uint64_t rw_next = *reinterpret_cast<const uint64_t *>(lref);
    // a window longer than a chunk streams through LDS: at every chunk boundary the next CH_COLS (+ 8) columns replace the
    // last (a wave is a workgroup: the barrier costs nothing, and windows of up to 2,044 columns never get here)
    for (int cb = 0; cb < (LONGW ? n_blocks : 1); cb += CH_BLOCKS) {
    if constexpr (LONGW) {
        if (cb) {
            __syncthreads();
            stage(cb * 4);
            __syncthreads();
        }
    }
    const int cb_end = LONGW ? min(n_blocks, cb + CH_BLOCKS) : n_blocks;
    for (int blk0 = cb; blk0 < cb_end; blk0 += GROUP) {
    const int blk_end = (MODE == 1) ? min(cb_end, blk0 + GROUP) : cb_end;
    for (int blk = blk0; blk < blk_end; blk++) {
        uint64_t rw;
        if constexpr (MODE == 1) {
            rw = rw_next;
            rw_next = *reinterpret_cast<const uint64_t *>(lref + (min(blk + 1, n_blocks - 1) - cb) * 4);  // (up to one block into the next chunk's columns)
        } else {
            rw = *reinterpret_cast<const uint64_t *>(lref + (blk - cb) * 4);
        }
        uint32_t acc[R];
        if constexpr (MODE != 1) {
#pragma unroll
            for (int k = 0; k < R; k++) acc[k] = 0;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int t = blk * 4 + s;
            const uint32_t fresh = (uint32_t)(rw >> (16 * s)) & 0xffffu;
            rc = (uint32_t)__builtin_amdgcn_update_dpp((int)fresh, (int)rc, DPP_ROW_SHR1, 0xf, 0xf, false);
            uint32_t hu = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hu_out, DPP_ROW_SHR1, 0xf, 0xf, true);
            uint32_t fu = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)fu_out, DPP_ROW_SHR1, 0xf, 0xf, true);
            if constexpr (LG == 8) {  // lane 8 of a DPP row starts a group of its own
                rc = bfi(lig0m, fresh, rc);
                hu &= ~lig0m;
                fu &= ~lig0m;
            }
            uint2 tA, tB;
            if constexpr (FAST) {
                const uint2 e = *reinterpret_cast<const uint2 *>(wt + rc);
                tA.x = e.x;
                tB.x = e.y;
                tA.y = tB.y = 0;
            } else {
                const uint4 e = *reinterpret_cast<const uint4 *>(wt + rc);
                tA = make_uint2(e.x, e.z);
                tB = make_uint2(e.y, e.w);
            }
            const uint32_t ct = (uint32_t)(0xffff - t);
            const uint32_t tk = (uint32_t)(KW - 1 - (t & (KW - 1))) * 0x10001u;  // MODE 1: position inside the key window
            // PAIRKEY: ... and which row of the pair: 2 * position + (row even), or KW * (row even) + position
            const uint32_t tk_even = end_min_ref ? tk * 2u + 0x10001u : tk + (uint32_t)KW * 0x10001u, tk_odd = end_min_ref ? tk * 2u : tk;
            uint32_t kprev = 0;
            uint32_t hd = hu_prev;
            hu_prev = hu;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t Dp;
                if constexpr (FAST) {
                    Dp = hd + __builtin_amdgcn_perm(tB.x, tA.x, selA[r]);
                } else {
                    const uint32_t wA = __builtin_amdgcn_perm(tA.y, tA.x, selA[r]);
                    const uint32_t wB = __builtin_amdgcn_perm(tB.y, tB.x, selB[r]);
                    Dp = hd + wA + wB;
                }
                const uint32_t hl = Hl[r];
                const uint32_t Ee = as_u32(as_s2(Eh[r]) - as_s2(ext8));
                const uint32_t En = as_u32(__builtin_elementwise_max(as_s2(hl), as_s2(Ee)));
                const uint32_t Fe = as_u32(as_s2(fu) - as_s2(ext8));
                const uint32_t Fn = as_u32(__builtin_elementwise_max(as_s2(hu), as_s2(Fe)));
                const uint32_t T = pk_max3_nonneg(Dp, En, Fn);  // Dp = H + W' >= 0, En >= H_left >= 0, Fn >= H_up >= 0
                const uint32_t H = as_u32(__builtin_elementwise_sub_sat(as_u2(T), as_u2(open8)));
                if constexpr (MODE != 1) {
                    // trace nibble: 8*(H!=D) + 4*(H!=F) + 2*(E opened) + 1*(F opened)
                    const uint32_t m1 = pk_min_k<8>(pk_sub(T, Dp));
                    uint32_t m2, m3, m4;
                    if constexpr (MODE == 3) {
                        // A.4 switches: bit 4 says H != E when E has priority over F; with "ties open" a gap opens iff
                        // H - open >= G - ext, i.e. unless the extension is strictly larger
                        m2 = pk_min_k<4>(pk_sub(T, hdir_f_first ? Fn : En));
                        m3 = pk_min_k<2>(pk_sub(En, tie_extends ? Ee : hl)) ^ tie_flip2;
                        m4 = pk_min_k<1>(pk_sub(Fn, tie_extends ? Fe : hu)) ^ tie_flip1;
                    } else {
                        m2 = pk_min_k<4>(pk_sub(T, Fn));
                        m3 = pk_min_k<2>(pk_sub(En, Ee));
                        m4 = pk_min_k<1>(pk_sub(Fn, Fe));
                    }
                    uint32_t &ac = acc[(r - s + R) % R];  // dword k of the block = the 4 cells (s, row (k + s) % R): a diagonal
                    ac = pk_shl4_add(ac, m1) + m2 + m3 + m4;
                }
                if constexpr (MODE == 0) {
                    // end-cell keys: (H, 0xffff - t) per alignment
                    bestA[r] = max(bestA[r], (H << 16) | ct);
                    bestB[r] = max(bestB[r], and_or(H, himask, ct));
                }
                if constexpr (MODE == 1 && !PAIRKEY) {
                    bestA[r] = as_u32(__builtin_elementwise_max(as_u2(bestA[r]), as_u2(pk_mad4(H, tk))));
                }
                if constexpr (PAIRKEY) {
                    const uint32_t k = KW == 32 ? pk_mad8(H, (r & 1) ? tk_odd : tk_even) : pk_mad4(H, (r & 1) ? tk_odd : tk_even);
                    if (r & 1) bestA[r >> 1] = pk_max3_nonneg(bestA[r >> 1], kprev, k);
                    else if ((R & 1) && r == R - 1) bestA[r >> 1] = pk_max3_nonneg(bestA[r >> 1], k, k);  // (an odd R: the last row has no partner)
                    else kprev = k;
                }
                hd = hl;
                Hl[r] = H;
                Eh[r] = En;
                hu = H;
                fu = Fn;
            }
            hu_out = hu;
            fu_out = fu;
        }
        if constexpr (MODE != 1) {
            uint32_t *tp = tq + (uint64_t)blk * (R * 64);
#pragma unroll
            for (int k = 0; k < R; k++) tp[k * 64] = acc[k];
        }
    }
    if constexpr (MODE == 1) {
        // fold the window's keys into (GH, GT); a later window wins only with a strictly larger H (Appendix A.3)
        // PAIRKEY: GT holds 2 * t + (row odd) = (2 * (KW * window + KW - 1) + 1) - (2 * (KW - 1 - t % KW) + (row even))
        // (a pair key under the other A.3 rule: the row bit ranks with H, GT = 2 KW * window + KW * (row odd) + t % KW)
        const uint32_t win = (uint32_t)(blk0 / GROUP);
        const uint32_t wbase = PAIRKEY ? (win * (2 * KW) + (2 * KW - 1)) * 0x10001u : (win * 32 + 31) * 0x10001u;
        const uint32_t HMASK = (PAIRKEY && !end_min_ref) ? ~(((1u << (KLOW - 1)) - 1u) * 0x10001u) : ~KLOWM, PMASK = KLOWM;
#pragma unroll
        for (int r = 0; r < (PAIRKEY ? (R + 1) / 2 : R); r++) {
            const uint32_t wk = bestA[r];
            const uint32_t wH = wk & HMASK;
            const uint32_t imp = as_u32(__builtin_elementwise_sub_sat(as_u2(wH), as_u2(bestB[r])));
            const uint32_t tnew = pk_sub(wbase, wk & PMASK);
            const uint32_t mask = pk_mul_ffff(pk_min_k<1>(imp));
            bestB[r] = as_u32(__builtin_elementwise_max(as_u2(bestB[r]), as_u2(wH)));
            GT[r] = bfi(mask, tnew, GT[r]);
            bestA[r] = 0;
        }
        // snapshot of the wave state after step 32(k+1)-1
        constexpr int WIN_PER_CK = CK_COLS / KW;
        const int sn = ((int)win + 1) / WIN_PER_CK - 1;
        if (blk_end == blk0 + GROUP && ((int)win + 1) % WIN_PER_CK == 0 && sn < a.n_ck) {
            constexpr int CKD = ck_dwords(R);
            uint32_t *cp = ckw + (uint64_t)sn * (CKD * 64);
#pragma unroll
            for (int r = 0; r < R; r++) {
                cp[r * 64] = Hl[r];
                cp[(R + r) * 64] = Eh[r];
            }
            cp[(2 * R) * 64] = hu_out;
            cp[(2 * R + 1) * 64] = fu_out;
            cp[(2 * R + 2) * 64] = hu_prev;
            cp[(2 * R + 3) * 64] = rc;
        }
    }
    }
    }  // chunks
    };  // sweep
    if (__any(special)) sweep(std::false_type{});
    else sweep(std::true_type{});

    // per-lane candidates: NK 32-bit keys (H8 << 16) | (0xffff - t) per alignment and the lane-local row each belongs to
    constexpr int NK = PAIRKEY ? (R + 1) / 2 : R;
    uint32_t rowA[R], rowB[R];
#pragma unroll
    for (int r = 0; r < R; r++) { rowA[r] = (uint32_t)r; rowB[r] = (uint32_t)r; }
    if constexpr (MODE == 1 && !PAIRKEY) {
        // (GH, GT) -> 32-bit keys, 0 when H == 0
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t gh = bestB[r], gt = GT[r];
            const uint32_t ha = (gh & 0xffffu) >> 2, hb = gh >> 18;  // 32*H = 4*H8
            bestA[r] = ha ? ((ha << 16) | (0xffffu - (gt & 0xffffu))) : 0u;
            bestB[r] = hb ? ((hb << 16) | (0xffffu - (gt >> 16))) : 0u;
        }
    }
    if constexpr (PAIRKEY) {
#pragma unroll
        for (int p2 = 0; p2 < NK; p2++) {
            const uint32_t gh = bestB[p2], gt = GT[p2];
            // the key's score field is 64 * score = 8 * H8 (6 low bits) or 32 * score = 4 * H8 (5 low bits)
            constexpr int KS = KLOW - 3, KB = KLOW - 1;  // shift back to H8; bits of the position inside a window
            const uint32_t ha = ((gh & 0xffffu) >> KLOW) << 3, hb = (gh >> (16 + KLOW)) << 3;
            static_assert(KS == 2 || KS == 3, "key scale");
            const uint32_t gta = gt & 0xffffu, gtb = gt >> 16;
            // GT = 2 t + (row odd), or 2 KW * (t / KW) + KW * (row odd) + t % KW under the other A.3 rule
            const uint32_t ta = end_min_ref ? gta >> 1 : ((gta >> KLOW) << KB) | (gta & (uint32_t)(KW - 1));
            const uint32_t tb = end_min_ref ? gtb >> 1 : ((gtb >> KLOW) << KB) | (gtb & (uint32_t)(KW - 1));
            rowA[p2] = 2u * (uint32_t)p2 + (end_min_ref ? (gta & 1u) : ((gta >> KB) & 1u));
            rowB[p2] = 2u * (uint32_t)p2 + (end_min_ref ? (gtb & 1u) : ((gtb >> KB) & 1u));
            bestA[p2] = ha ? ((ha << 16) | (0xffffu - ta)) : 0u;
            bestB[p2] = hb ? ((hb << 16) | (0xffffu - tb)) : 0u;
        }
    }
    if constexpr (!P2) {
        // ---- end cells (Appendix A.3) for A and B: equal keys keep the earlier (smaller) row
        uint32_t bka = 0, bkb = 0;
        int browa = 0, browb = 0;
#pragma unroll
        for (int r = 0; r < NK; r++) {
            const int ra = lig * R + (int)rowA[r], rb = lig * R + (int)rowB[r];
            // a lane's rows share the column of a step, so equal keys mean equal columns; under the other A.3 rule the
            // earlier row also wins against a later row's smaller column
            const uint32_t ka = end_min_ref ? bestA[r] : (bestA[r] & 0xffff0000u), kb = end_min_ref ? bestB[r] : (bestB[r] & 0xffff0000u);
            if (ra < lqA && ka > (end_min_ref ? bka : (bka & 0xffff0000u))) { bka = bestA[r]; browa = ra; }
            if (rb < lqB && kb > (end_min_ref ? bkb : (bkb & 0xffff0000u))) { bkb = bestB[r]; browb = rb; }
        }
        // (H, -column, -row), or (H, -row, -column)
        uint64_t ca = 0, cb = 0;
        if (bka >> 16) {
            const uint64_t h = bka >> 16, nj = (bka + (uint32_t)lig) & 0xffffu, ni = (uint64_t)(0xffff - browa);
            ca = end_min_ref ? (h << 32) | (nj << 16) | ni : (h << 32) | (ni << 16) | nj;
        }
        if (bkb >> 16) {
            const uint64_t h = bkb >> 16, nj = (bkb + (uint32_t)lig) & 0xffffu, ni = (uint64_t)(0xffff - browb);
            cb = end_min_ref ? (h << 32) | (nj << 16) | ni : (h << 32) | (ni << 16) | nj;
        }
#pragma unroll
        for (int m = 1; m < LG; m <<= 1) {
            const uint64_t oa = __shfl_xor(ca, m, 64), ob = __shfl_xor(cb, m, 64);
            ca = oa > ca ? oa : ca;
            cb = ob > cb ? ob : cb;
        }
        if (!end_min_ref) {  // back to (H, -column, -row) for the unpacking below
            ca = (ca & 0xffffffff00000000ull) | ((ca & 0xffffull) << 16) | ((ca >> 16) & 0xffffull);
            cb = (cb & 0xffffffff00000000ull) | ((cb & 0xffffull) << 16) | ((cb >> 16) & 0xffffull);
        }
        if (lig == 0) {
            if (itemA < n_items) {
                Fwd f;
                f.score = (int32_t)(ca >> 32) / PK_SCALE;
                f.end_r = ca ? (int32_t)(0xffff - ((ca >> 16) & 0xffff)) : 0;
                f.end_q = ca ? (int32_t)(0xffff - (ca & 0xffff)) : 0;
                f.pad = 0;
                a.fwd[itemA] = f;
            }
            if (itemB < n_items) {
                Fwd f;
                f.score = (int32_t)(cb >> 32) / PK_SCALE;
                f.end_r = cb ? (int32_t)(0xffff - ((cb >> 16) & 0xffff)) : 0;
                f.end_q = cb ? (int32_t)(0xffff - (cb & 0xffff)) : 0;
                f.pad = 0;
                a.fwd[itemB] = f;
            }
        }
        if constexpr (MODE == 1) {
            if (a.sel.enabled) {
                // every lane of a group holds both results after the butterfly: lane 0 finishes alignment A, lane 1 B
                uint32_t art = 0;
                const int item = lig ? itemB : itemA;
                if (lig < 2 && item < n_items) {
                    const uint64_t cc = lig ? cb : ca;
                    Fwd f;
                    f.score = (int32_t)(cc >> 32) / PK_SCALE;
                    f.end_r = cc ? (int32_t)(0xffff - ((cc >> 16) & 0xffff)) : 0;
                    f.end_q = cc ? (int32_t)(0xffff - (cc & 0xffff)) : 0;
                    f.pad = 0;
                    // (R here is pass 2's rows per lane: it sizes the steps a candidate's re-computation will take)
                    art = select_one(a.sel, item, lig ? wb : wa, f, LG == 8 ? (R + 1) / 2 : R, a.q_nib, a.r_nib, a.sc.rules);
                }
                if (a.sel.stats) add_artifact_stats(a.sel.stats, art, blockIdx.x);
            }
        }
        break;
    } else {
        // ---- tracebacks of the octet's members, by lanes 0..7 (member e = group e / 2, half e & 1) of this wave
        __threadfence();  // the trace this wave has just stored is what its lanes read back
        const int gg = (lane & 7) >> 1, hh = lane & 1;
        const uint32_t sA = (uint32_t)__shfl((int)srcA, gg * 16, 64), sB = (uint32_t)__shfl((int)srcB, gg * 16, 64);
        const uint32_t tA = (uint32_t)__shfl((int)c0A, gg * 16, 64), tB = (uint32_t)__shfl((int)c0B, gg * 16, 64);
        const int hA = __shfl((int)haveA, gg * 16, 64), hB = __shfl((int)haveB, gg * 16, 64);
        uint32_t code = 0;
        if (lane < 8 && (hh ? hB : hA)) {
            TbArgs tb;
            tb.work = a.work;
            tb.meta = a.meta;
            tb.fwd = a.fwd;
            tb.n_items = 0;
            tb.R = R;
            tb.q_nib = a.q_nib;
            tb.r_nib = a.r_nib;
            tb.trace = a.trace;
            tb.quad_stride = a.quad_stride;
            tb.sc = a.sc;
            tb.out = a.out;
            tb.rs = a.rs;
            tb.stats = a.stats;
            tb.floor_len = a.floor_len;
            tb.gate = a.gate;
            tb.early_out = a.early_out;
            tb.packed = 1;
            tb.ltrace = nullptr;
            tb.lhalf = 0;
            tb.count_dev = nullptr;
            tb.item_base = 0;
            code = traceback_path(tb, (int)(hh ? sB : sA), (int)(hh ? tB : tA), a.trace + trace_off, gg, hh, 0);
        }
        if (a.stats) add_artifact_stats(a.stats, code, blockIdx.x);
        const unsigned long long inc = __ballot(code & 8u);
        if (inc == 0ull) break;
        // the members whose path left the traced steps stay, with T0 further back; the others are done
        haveA = haveA && ((inc >> (2 * g)) & 1ull);
        haveB = haveB && ((inc >> (2 * g + 1)) & 1ull);
        constexpr uint32_t BACK = CK_COLS >= 128 ? (uint32_t)CK_COLS : 128u;  // a multiple of the snapshot stride
        c0A = (attempt == 0 && c0A > BACK) ? c0A - BACK : 0u;
        c0B = (attempt == 0 && c0B > BACK) ? c0B - BACK : 0u;
        if (lane == 0 && a.rerun_total) atomicAdd(a.rerun_total, (unsigned long long)__popcll(inc));
        __syncthreads();  // the octet's window is staged again
    }
    }  // attempts at this octet
    }  // octets of this wave
}

// ---------------------------------------------------------------- forward SW for queries longer than 512 bases
// A thread per alignment, row by row over the window, same recurrence, flags and tie rules as the wave kernels
// (hat domain: E-hat = E + open, F-hat = F + open, borders 0).  The previous row's H and F-hat and the 4-bit trace
// live in global memory, interleaved over the launch's items so that the lanes of a wave touch consecutive
// addresses.  Not a fast path: short-read libraries put a handful of merged pairs here, if any.
struct LongArgs {
    const Work *work;
    int32_t n_items;
    const uint8_t *q_nib, *r_nib;
    int32_t *hrow, *frow;   // [max_lr][n_items]
    uint8_t *trace;         // [max_lq * lhalf][n_items], two cells per byte (even column in the low nibble)
    int32_t lhalf;          // (max_lr + 1) / 2
    int32_t max_lq, max_lr;
    Fwd *fwd;
    ScoreTab sc;
    const uint32_t *count_dev;  // items on the long list (nullptr = n_items is exact); n_items stays the interleave stride
    uint32_t item_base;
};

__global__ __launch_bounds__(64) void sw_long_kernel(LongArgs a) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    int n_live = a.n_items;
    if (a.count_dev) n_live = min(n_live, max(0, (int)*a.count_dev - (int)a.item_base));
    const bool have = item < n_live;
    Work w;
    w.r_base = 0; w.q_base = 0; w.lq = 0; w.lr = 0; w.idx = 0; w.flags = 0; w.out = 0;
    if (have) w = a.work[item];
    const int lq = (int)w.lq, lr = (int)w.lr;
    const uint32_t n = (uint32_t)a.n_items;
    const int32_t open = a.sc.open, ext = a.sc.ext;
    const bool hdir_f_first = rule(a.sc.rules, FADEHIP_RULE_HDIR_DIAG_F_E), tie_extends = rule(a.sc.rules, FADEHIP_RULE_GAP_TIE_EXTENDS);
    const bool end_min_ref = rule(a.sc.rules, FADEHIP_RULE_END_MIN_REF_THEN_QUERY);
    for (int j = 0; j < lr; j++) {
        a.hrow[(uint64_t)j * n + item] = 0;
        a.frow[(uint64_t)j * n + item] = 0;
    }
    int32_t best = 0;
    int bi = 0, bj = 0;
    for (int i = 0; i < lq; i++) {
        uint32_t code;
        if (w.flags & 1u) code = lut4(COMP_LUT, nib_at(a.q_nib, (uint64_t)w.q_base + (uint32_t)(lq - 1 - i)));
        else code = nib_at(a.q_nib, (uint64_t)w.q_base + (uint32_t)i);
        const uint32_t qc = lut4(CLASS_LUT, code);
        uint32_t prof = a.sc.prof[0];
#pragma unroll
        for (int c = 1; c < 7; c++) prof = (qc == (uint32_t)c) ? a.sc.prof[c] : prof;
        int32_t hl = 0, El = 0, hd = 0;
        uint8_t *trow = a.trace + ((uint64_t)i * (uint32_t)a.lhalf) * n + item;
        // eight columns at a time: their loads are independent of the recurrence, so one memory round trip
        // serves eight cells (a thread is otherwise bound by ~0.4 us of load latency per cell)
        constexpr int CB = 8;
        for (int j0 = 0; j0 < lr; j0 += CB) {
            int32_t hu8[CB], fu8[CB];
            uint32_t cc8[CB];
#pragma unroll
            for (int k = 0; k < CB; k++) {
                const int j = j0 + k;
                hu8[k] = 0; fu8[k] = 0; cc8[k] = 0;
                if (j < lr) {
                    const uint64_t at = (uint64_t)j * n + item;
                    hu8[k] = a.hrow[at];
                    fu8[k] = a.frow[at];
                    cc8[k] = lut4(CLASS_LUT, nib_at(a.r_nib, w.r_base + (uint64_t)j));
                }
            }
            uint32_t pair = 0;
#pragma unroll
            for (int k = 0; k < CB; k++) {
                const int j = j0 + k;
                if (j < lr) {
                    const int32_t hu = hu8[k], fu = fu8[k];
                    const int32_t Dp = hd + (int32_t)((prof >> (4 * cc8[k])) & 15u);
                    const int32_t Ee = El - ext;
                    const int32_t En = max(hl, Ee);
                    const int32_t Fe = fu - ext;
                    const int32_t Fn = max(hu, Fe);
                    const int32_t T = max(max(Dp, En), Fn);
                    const int32_t H = max(T - open, 0);
                    // trace nibble as in the wave kernels: 8 (D < T), 4 (F < T), 2 (E opened), 1 (F opened)
                    // (A.4 switches: bit 4 is "H != E" when E has priority; with "ties open" a gap opens unless its extension is strictly larger)
                    const uint32_t nb = (Dp < T ? 8u : 0u) | ((hdir_f_first ? Fn : En) < T ? 4u : 0u) |
                                        ((tie_extends ? Ee < hl : Ee <= hl) ? 2u : 0u) | ((tie_extends ? Fe < hu : Fe <= hu) ? 1u : 0u);
                    if (k & 1) trow[(uint64_t)(j >> 1) * n] = (uint8_t)(pair | (nb << 4));
                    else {
                        pair = nb;
                        if (j == lr - 1) trow[(uint64_t)(j >> 1) * n] = (uint8_t)pair;  // odd window length: last cell alone
                    }
                    // end cell: max H, then smallest ref index, then smallest query index (rows are visited in order)
                    if (H > best || (end_min_ref && H == best && H > 0 && j < bj)) { best = H; bi = i; bj = j; }
                    const uint64_t at = (uint64_t)j * n + item;
                    a.hrow[at] = H;
                    a.frow[at] = Fn;
                    hd = hu;
                    hl = H;
                    El = En;
                }
            }
        }
    }
    if (have) {
        Fwd f;
        f.score = best;
        f.end_q = best ? bi : 0;
        f.end_r = best ? bj : 0;
        f.pad = 0;
        a.fwd[item] = f;
    }
}

// the single-pass kernels' and sw_long_kernel's traceback: a thread per alignment of the launch, traced from step 0
__device__ __forceinline__ uint32_t traceback_one(const TbArgs &a, const int item) {
    int n_items = a.n_items;
    if (a.count_dev) n_items = min(n_items, max(0, (int)*a.count_dev - (int)a.item_base));
    if (item >= n_items) return 0u;
    if (a.packed == 3)  // sw_forward64_kernel: one alignment per wave, its 64 lanes in the place of a 16-lane group
        return traceback_path(a, item, 0, a.trace + (uint64_t)item * a.quad_stride, 0, 0, item);
    const uint64_t trace_off = (uint64_t)(a.packed ? (item >> 3) : (item >> 2)) * a.quad_stride;
    return traceback_path(a, item, 0, a.trace + trace_off, a.packed ? ((item >> 1) & 3) : (item & 3), item & 1, item);
}

__global__ void traceback_kernel(TbArgs a) {
    const uint32_t r = traceback_one(a, (int)(blockIdx.x * blockDim.x + threadIdx.x));
    if (a.stats) add_artifact_stats(a.stats, r, blockIdx.x);
}

// ---------------------------------------------------------------- run totals kept on the device
struct PlanOut {            // in the slot's counter block, read by the host after the run
    unsigned long long cand_total;   // candidates traced by pass 2 (all classes)
    unsigned long long rerun_total;  // candidates traced again because their path left the traced steps
};

// level 1 builds its class lists on the host: their counts go where the gate would have left them
__global__ void set_counts_kernel(uint32_t *dst, uint32_t v) { *dst = v; }

}  // namespace fadehip
