// bam_device.hpp — the BAM record stream handled on the device (gfx950), between the BGZF inflater and the BGZF
// compressor: what dhtslib's SAMReader / SAMRecord / SAMWriter do around annotateTask (anno.d:44-50, 61-107), so that a
// file passes through the device as bytes and the host only moves compressed blocks.
//
//   inflated bytes U  --frame-->  record offsets  --pack-->  the batch arrays of fadehip_read_batch (the records
//   anno.d:61-65 does not settle)  --[gate, score pass, pass 2: fadehip_kernels.hpp]-->  rs, alignments
//   --tag sizes, scan-->  output offsets  --rewrite-->  output bytes O (records + rs / am / as / ar / ab)  --> compressor
//
// Framing.  BAM records are chained by block_size; following the chain is serial, so it is done speculatively per 64 KB
// segment: a wave looks for the first position in its segment that looks like a record (block_size, refID, pos,
// l_read_name, n_cigar_op, l_seq, next_refID, next_pos consistent with each other and with the header, name
// NUL-terminated) and walks the chain from there; a single wave then checks, segment by segment, that the chain really
// enters each segment where its wave assumed — and walks any segment again, serially, where it did not.  The plausibility
// test only decides how often that happens, never what the result is.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fadehip.h"

namespace fadehip {
namespace bam {

typedef uint32_t u32u __attribute__((aligned(1)));
typedef int32_t i32u __attribute__((aligned(1)));
typedef uint16_t u16u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return *reinterpret_cast<const u32u *>(p); }
__device__ __forceinline__ int32_t ld32s(const uint8_t *p) { return *reinterpret_cast<const i32u *>(p); }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return *reinterpret_cast<const u16u *>(p); }

// framing segment: a walk is a chain of one dependent load per record, so the segment sets the framing's latency (64 KB: ~200
// records, 185 us per call; 16 KB: a quarter of that).  The resolving wave no longer pays per segment (bam_frame_resolve_kernel
// takes the linked segments of a group at once), which is what kept the segments large.
constexpr uint32_t SEG = 16384;
constexpr uint32_t SEG_SLOTS = SEG / 36 + 2;    // record starts a segment can hold (a record takes at least 36 bytes)
constexpr uint32_t EXIT_INCOMPLETE = 0x80000000u, EXIT_BAD = 0x40000000u, EXIT_MASK = 0x3fffffffu;
constexpr uint32_t MAX_U = 0x3fffff00u;         // inflated bytes per chunk (offsets are 30 bits + two flags)

// ---- counters of a chunk, read back by the host between the stages
struct ChunkCounts {
    uint32_t n_records;     // complete records framed
    uint32_t consumed;      // bytes of U covered by them (the rest is carried over to the next chunk)
    uint32_t frame_err;     // != 0: a record on the chain is impossible (block_size < 32), at offset frame_err_at
    uint32_t frame_err_at;
    uint32_t n_redone;      // segments whose speculative entry was wrong (diagnostics)
    uint32_t n_bad_layout;  // records whose fields do not fit their block_size / whose aux area is not whole fields
    uint32_t n_sent;        // records that go to the gate kernel (mapped, with an S op)
    uint32_t n_cig;         // their CIGAR ops
    uint32_t n_seq;         // their packed sequence bytes
    uint32_t l_seq_min, l_seq_max, span_max, n_long_q;
    uint32_t n_ours;        // records that already carry one of rs / am / as / ar / ab
    uint32_t pad0, pad1;
    uint64_t out_bytes;     // bytes of the rewritten record stream
    uint64_t pad2;
};

// ============================================================================================ framing
__device__ __forceinline__ bool plausible(const uint8_t *u, uint32_t at, uint32_t u_len, int32_t n_ref) {
    if (at + 36u > u_len) return false;
    const uint8_t *p = u + at;
    const uint32_t bs = ld32(p);
    const int32_t tid = ld32s(p + 4), pos = ld32s(p + 8), lseq = ld32s(p + 20), ntid = ld32s(p + 24), npos = ld32s(p + 28);
    const uint32_t lname = p[12], ncig = ld16(p + 16);
    if (bs < 33u || bs > (1u << 29)) return false;
    if (tid < -1 || tid >= n_ref || ntid < -1 || ntid >= n_ref || pos < -1 || npos < -1 || lseq < 0 || lname == 0) return false;
    const uint64_t need = 32ull + lname + 4ull * ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq;
    if (need > bs) return false;
    const uint32_t nul = at + 36u + lname - 1u;
    if (nul < u_len && u[nul] != 0) return false;
    return true;
}

// Walk the chain from `c` while records start in front of seg_end; record starts go to slots[0 .. n).  Returns the exit:
// the start of the first record at or behind seg_end, or (| EXIT_INCOMPLETE) the start of a record that is not whole in
// U, or (| EXIT_BAD) the start of an impossible one.
__device__ __forceinline__ uint32_t walk_segment(const uint8_t *u, uint32_t u_len, uint32_t c, uint32_t seg_end, uint32_t *slots, uint32_t *n_out) {
    uint32_t n = 0;
    uint32_t ex;
    for (;;) {
        if (c >= seg_end) { ex = c; break; }
        if (c + 4u > u_len) { ex = c | EXIT_INCOMPLETE; break; }
        const uint32_t bs = ld32(u + c);
        if (bs < 32u || bs > (1u << 29)) { ex = c | EXIT_BAD; break; }
        if ((uint64_t)c + 4ull + bs > (uint64_t)u_len) { ex = c | EXIT_INCOMPLETE; break; }
        if (n < SEG_SLOTS) slots[n] = c;
        n++;
        c += 4u + bs;
    }
    *n_out = n;
    return ex;
}

struct FrameArgs {
    const uint8_t *u;
    uint32_t u_len;             // bytes of U (carried-over bytes + this chunk's inflated bytes)
    uint32_t first;             // where the chain starts in U
    int32_t n_ref;
    uint32_t n_seg_cap;         // segments the arrays hold
    uint32_t *cand, *exit_, *cnt, *base;  // per segment
    uint32_t *slots;            // [n_seg][SEG_SLOTS]
    uint32_t *rec_off;          // [n_records + 1] out
    uint32_t rec_cap;
    ChunkCounts *counts;
};

// one wave per WALK_SEGS segments: the lanes look for each segment's first plausible record together, then WALK_SEGS lanes walk
// one segment each (a walk is a chain of ~200 dependent loads: the more waves share the segments, the sooner it is over)
constexpr uint32_t WALK_SEGS = 8;
__global__ __launch_bounds__(64) void bam_frame_walk_kernel(FrameArgs a) {
    const uint32_t u_len = a.u_len;
    const uint32_t n_seg = (u_len + SEG - 1) / SEG;
    const int lane = threadIdx.x;
    const uint32_t s0 = blockIdx.x * WALK_SEGS;
    uint32_t my_cand = 0xffffffffu;
    for (uint32_t j = 0; j < WALK_SEGS; j++) {
        const uint32_t s = s0 + j;
        if (s >= n_seg) break;  // (uniform)
        uint32_t found = 0xffffffffu;
        if (s == 0) {
            found = a.first;
        } else {
            // a record start in this segment, or — when one record covers it all — none: the search stops at the segment's end
            const uint32_t lo = s * SEG, hi = min(lo + SEG, u_len);
            for (uint32_t at = lo; at < hi && found == 0xffffffffu; at += 64u) {
                const bool ok = at + (uint32_t)lane < hi && plausible(a.u, at + (uint32_t)lane, u_len, a.n_ref);
                const unsigned long long m = __ballot(ok);
                if (m) found = at + (uint32_t)__ffsll((long long)m) - 1u;
            }
        }
        if ((uint32_t)lane == j) my_cand = found;
    }
    const uint32_t s = s0 + (uint32_t)lane;
    if ((uint32_t)lane < WALK_SEGS && s < n_seg && s < a.n_seg_cap) {
        uint32_t n = 0, ex = 0xffffffffu;
        if (my_cand != 0xffffffffu) ex = walk_segment(a.u, u_len, my_cand, min((s + 1u) * SEG, 0xffffffffu - SEG), a.slots + (size_t)s * SEG_SLOTS, &n);
        a.cand[s] = my_cand;
        a.exit_[s] = ex;
        a.cnt[s] = n;
    }
}

// one wave: the true chain through the segments
__global__ __launch_bounds__(64) void bam_frame_resolve_kernel(FrameArgs a) {
    const uint32_t u_len = a.u_len;
    const uint32_t n_seg = min((u_len + SEG - 1) / SEG, a.n_seg_cap);
    const int lane = threadIdx.x;
    uint32_t entry = a.first, total = 0, redone = 0, err = 0, err_at = 0;
    bool stop = false;
    if (entry > u_len) { stop = true; err = 2; err_at = entry; }
    for (uint32_t s0 = 0; s0 < n_seg; s0 += 64u) {
        const uint32_t s = s0 + (uint32_t)lane;
        uint32_t c = 0xffffffffu, e = 0, n = 0;
        if (s < n_seg) { c = a.cand[s]; e = a.exit_[s]; n = a.cnt[s]; }
        uint32_t my_base = 0, my_n = 0;
        // The group's head without a chain: a segment's walk is the true one when it started where the chain enters it — its
        // candidate is the exit of the segment in front (for the group's first: the entry carried here), inside the segment.
        // That holds for all but a handful of a call's segments (where a look-alike precedes the first record, or one record
        // covers a whole segment); for the lanes up to the first that is different, or the first whose walk ended early, the
        // record counts are a prefix sum and the entry is the last one's exit.  The rest of the group takes the chain below.
        uint32_t j_from = 0;
        if (!stop) {
            uint32_t prev_e = (uint32_t)__shfl_up((int)e, 1, 64);
            if (lane == 0) prev_e = entry;
            const bool clean = !(e & (EXIT_BAD | EXIT_INCOMPLETE));
            const bool link = s < n_seg && c != 0xffffffffu && prev_e == c && c < (s + 1u) * SEG;  // (a flagged exit in front never equals a candidate: the flags are high bits)
            const unsigned long long links = __ballot(link), cleans = __ballot(clean || s >= n_seg);
            uint32_t k = ~links ? (uint32_t)__builtin_ctzll(~links) : 64u;          // lanes [0, k) are linked
            if (~cleans) k = min(k, (uint32_t)__builtin_ctzll(~cleans) + 1u);         // ... up to and with the first walk that ended early
            if (k) {
                // exclusive prefix sum of the counts over lanes [0, k)
                uint32_t inc = (uint32_t)lane < k ? n : 0u;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
                    if (lane >= d) inc += up;
                }
                if ((uint32_t)lane < k) { my_base = total + inc - n; my_n = n; }
                total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                const uint32_t ex = (uint32_t)__shfl((int)e, (int)k - 1, 64);
                if (ex & EXIT_BAD) { stop = true; err = 1; err_at = ex & EXIT_MASK; entry = ex & EXIT_MASK; }
                else if (ex & EXIT_INCOMPLETE) { stop = true; entry = ex & EXIT_MASK; }
                else entry = ex;
                j_from = k;
            }
        }
        for (uint32_t j = j_from; j < 64u && s0 + j < n_seg; j++) {
            const uint32_t sj = s0 + j;
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)j), ej = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)j),
                           nj = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)j);
            uint32_t use_n = 0;
            const uint32_t base_j = total;
            if (!stop && entry < (sj + 1u) * SEG) {
                uint32_t ex = ej;
                use_n = nj;
                if (entry != cj) {
                    // the wave of this segment assumed another entry: walk it again from the true one (one lane; rare)
                    uint32_t n2 = 0, ex2 = 0;
                    if (lane == 0) ex2 = walk_segment(a.u, u_len, entry, (sj + 1u) * SEG, a.slots + (size_t)sj * SEG_SLOTS, &n2);
                    ex = (uint32_t)__builtin_amdgcn_readlane((int)ex2, 0);
                    use_n = (uint32_t)__builtin_amdgcn_readlane((int)n2, 0);
                    redone++;
                }
                total += use_n;
                if (ex & EXIT_BAD) { stop = true; err = 1; err_at = ex & EXIT_MASK; entry = ex & EXIT_MASK; }
                else if (ex & EXIT_INCOMPLETE) { stop = true; entry = ex & EXIT_MASK; }
                else entry = ex;
            }
            if ((uint32_t)lane == j) { my_base = base_j; my_n = use_n; }
        }
        if (s < n_seg) { a.base[s] = my_base; a.cnt[s] = my_n; }
    }
    if (lane == 0) {
        ChunkCounts *cc = a.counts;
        cc->n_records = total;
        cc->consumed = min(entry, u_len);
        cc->frame_err = err;
        cc->frame_err_at = err_at;
        cc->n_redone = redone;
    }
}

// rec_off[base[s] + k] = slots[s][k]; rec_off[n_records] = consumed
__global__ __launch_bounds__(256) void bam_frame_compact_kernel(FrameArgs a) {
    const uint32_t u_len = a.u_len;
    const uint32_t n_seg = min((u_len + SEG - 1) / SEG, a.n_seg_cap);
    const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.counts->n_records < a.rec_cap) a.rec_off[a.counts->n_records] = a.counts->consumed;
    if (s >= n_seg) return;
    const uint32_t n = min(a.cnt[s], SEG_SLOTS), b = a.base[s];
    for (uint32_t k = (uint32_t)lane; k < n; k += 64u)
        if (b + k < a.rec_cap) a.rec_off[b + k] = a.slots[(size_t)s * SEG_SLOTS + k];
}

// ============================================================================================ records
struct RecHdr {
    const uint8_t *p;   // at block_size
    uint32_t bs;        // block_size
    int32_t tid, pos, lseq;
    uint32_t lname, ncig, flag;
    uint32_t cig_off, seq_off, qual_off, aux_off;  // relative to p
    uint32_t end;                                  // 4 + bs
};
__device__ __forceinline__ RecHdr rec_header(const uint8_t *p) {
    RecHdr r;
    r.p = p;
    r.bs = ld32(p);
    r.tid = ld32s(p + 4);
    r.pos = ld32s(p + 8);
    r.lname = p[12];
    r.ncig = ld16(p + 16);
    r.flag = ld16(p + 18);
    r.lseq = ld32s(p + 20);
    r.cig_off = 36u + r.lname;
    r.seq_off = r.cig_off + 4u * r.ncig;
    const uint32_t lq = r.lseq > 0 ? (uint32_t)r.lseq : 0u;
    r.qual_off = r.seq_off + (lq + 1u) / 2u;
    r.aux_off = r.qual_off + lq;
    r.end = 4u + r.bs;
    return r;
}
__device__ __forceinline__ int aux_type_size(uint8_t t) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        default: return 0;
    }
}
// size of the aux field whose type byte is at offset q of the record (type byte included), 0 if malformed / not whole
__device__ __forceinline__ uint32_t aux_field_size(const uint8_t *p, uint32_t q, uint32_t end) {
    if (q >= end) return 0;
    const uint8_t t = p[q];
    const int s = aux_type_size(t);
    if (s) return q + 1u + (uint32_t)s <= end ? 1u + (uint32_t)s : 0u;
    if (t == 'Z' || t == 'H') {
        uint32_t k = q + 1u;
        while (k + 4u <= end) {  // four bytes per load: stop at the dword that holds a zero byte
            const uint32_t v = ld32(p + k);
            if ((v - 0x01010101u) & ~v & 0x80808080u) break;
            k += 4u;
        }
        while (k < end && p[k]) k++;
        return k < end ? k - q + 1u : 0u;
    }
    if (t == 'B') {
        if (q + 6u > end) return 0;
        const int es = aux_type_size(p[q + 1]);
        const uint64_t cnt = ld32(p + q + 2);
        if (!es) return 0;
        const uint64_t fs = 6ull + (uint64_t)es * cnt;
        return fs <= (uint64_t)(end - q) ? (uint32_t)fs : 0u;
    }
    return 0;
}
__device__ __forceinline__ bool is_ours(uint8_t a0, uint8_t a1) {
    return (a0 == 'r' && a1 == 's') || (a0 == 'a' && (a1 == 'm' || a1 == 's' || a1 == 'r' || a1 == 'b'));
}

enum : uint32_t { INFO_NEED = 1, INFO_SA = 2, INFO_OURS = 4, INFO_BAD = 8 };

struct PackArgs {
    const uint8_t *u;
    const uint32_t *rec_off;
    const ChunkCounts *counts_in;  // n_records
    uint32_t r0, r1_cap;           // records [r0, min(r1_cap, n_records)) of the chunk are this batch
    uint32_t *info;                // per record of the batch
    uint32_t *blk_sums;            // [n_blocks][3]: sent, cig, seq
    uint32_t *blk_base;            // [n_blocks][3]
    ChunkCounts *counts;           // the batch's totals
    // outputs of the write pass: the batch block (fadehip_read_batch's arrays) + the maps
    int32_t *tid, *pos, *lseq;
    uint32_t *cigar_off, *seq_off;
    uint16_t *flag;
    uint8_t *has_sa;
    uint32_t *cigar_ops;
    uint8_t *seq;
    int32_t *sent_of;              // [records of the batch]: index among the sent records or -1
};
constexpr int PACK_BLOCK = 1024;

__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *tmp16, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) tmp16[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < 16; w++) {
        const uint32_t t = tmp16[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// thread per record: layout check, anno.d:61-65 (does the record go to the device?), SA, our own tags; block sums
// (The launch is sized from an ESTIMATE of the records the bytes hold — a bound from the bytes alone would be eight times
// too many workgroups, each waiting for a wave slot beside the compressor; the grid strides over the blocks of PACK_BLOCK
// records there really are, whatever the estimate was.)
__global__ __launch_bounds__(PACK_BLOCK) void bam_pack_count_kernel(PackArgs a) {
    __shared__ uint32_t red[16][8];
    const uint32_t n_all = a.counts_in->n_records;
    const uint32_t r1 = min(a.r1_cap, n_all);
    const uint32_t n_blk = r1 > a.r0 ? (r1 - a.r0 + PACK_BLOCK - 1) / PACK_BLOCK : 0u;
    for (uint32_t blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
    const uint32_t i = a.r0 + blk * PACK_BLOCK + threadIdx.x;
    uint32_t info = 0, ncig = 0, nseq = 0, lq = 0, span = 0;
    if (i < r1) {
        const RecHdr r = rec_header(a.u + a.rec_off[i]);
        bool ok = r.bs >= 32u && r.lseq >= 0 && r.lname >= 1u && r.aux_off <= r.end;
        bool sa = false, ours = false;
        if (ok) {
            uint32_t q = r.aux_off;
            while (q < r.end) {
                if (q + 3u > r.end) { ok = false; break; }
                const uint32_t fs = aux_field_size(r.p, q + 2u, r.end);
                if (!fs) { ok = false; break; }
                const uint8_t a0 = r.p[q], a1 = r.p[q + 1];
                sa |= (a0 == 'S' && a1 == 'A');
                ours |= is_ours(a0, a1);
                q += 2u + fs;
            }
        }
        if (!ok) info = INFO_BAD;
        else {
            bool soft = false;
            uint64_t sp = 0;
            for (uint32_t k = 0; k < r.ncig; k++) {
                const uint32_t op = ld32(r.p + r.cig_off + 4u * k);
                soft |= (op & 15u) == 4u;
                if (FADEHIP_OP_CONSUMES_REF(op & 15u)) sp += op >> 4;
            }
            const bool need = !(r.flag & 4u) && soft;
            info = (need ? INFO_NEED : 0u) | (sa ? INFO_SA : 0u) | (ours ? INFO_OURS : 0u);
            if (need) {
                ncig = r.ncig;
                lq = (uint32_t)r.lseq;
                nseq = (lq + 1u) / 2u;
                span = (uint32_t)min(sp, (uint64_t)0x7fffffffu);
            }
        }
        a.info[i - a.r0] = info;
    }
    // block reductions: sums of sent / cig / seq, min / max of l_seq, max span, counts of long queries, bad, ours
    const bool need = (info & INFO_NEED) != 0;
    uint32_t v[8] = {need ? 1u : 0u, ncig, nseq, need ? lq : 0xffffffffu, need ? lq : 0u, span, (need && lq > 512u) ? 1u : 0u,
                     ((info & INFO_BAD) ? 1u : 0u) | ((info & INFO_OURS) ? 0x10000u : 0u)};
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v[0] += (uint32_t)__shfl_xor((int)v[0], m, 64);
        v[1] += (uint32_t)__shfl_xor((int)v[1], m, 64);
        v[2] += (uint32_t)__shfl_xor((int)v[2], m, 64);
        v[3] = min(v[3], (uint32_t)__shfl_xor((int)v[3], m, 64));
        v[4] = max(v[4], (uint32_t)__shfl_xor((int)v[4], m, 64));
        v[5] = max(v[5], (uint32_t)__shfl_xor((int)v[5], m, 64));
        v[6] += (uint32_t)__shfl_xor((int)v[6], m, 64);
        v[7] += (uint32_t)__shfl_xor((int)v[7], m, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 8; k++) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t[8] = {0, 0, 0, 0xffffffffu, 0, 0, 0, 0};
        for (int w = 0; w < 16; w++) {
            t[0] += red[w][0]; t[1] += red[w][1]; t[2] += red[w][2];
            t[3] = min(t[3], red[w][3]); t[4] = max(t[4], red[w][4]); t[5] = max(t[5], red[w][5]);
            t[6] += red[w][6]; t[7] += red[w][7];
        }
        a.blk_sums[3 * blk + 0] = t[0];
        a.blk_sums[3 * blk + 1] = t[1];
        a.blk_sums[3 * blk + 2] = t[2];
        ChunkCounts *c = a.counts;
        if (t[0]) {
            atomicMin(&c->l_seq_min, t[3]);
            atomicMax(&c->l_seq_max, t[4]);
            atomicMax(&c->span_max, t[5]);
        }
        if (t[6]) atomicAdd(&c->n_long_q, t[6]);
        if (t[7] & 0xffffu) atomicAdd(&c->n_bad_layout, t[7] & 0xffffu);
        if (t[7] >> 16) atomicAdd(&c->n_ours, t[7] >> 16);
    }
    __syncthreads();  // `red` is written again by the block's next round
    }
}

// one block: exclusive scan of the block sums (three columns) -> blk_base, totals -> counts
__global__ __launch_bounds__(1024) void bam_pack_scan_kernel(PackArgs a, uint32_t n_blocks) {
    __shared__ uint32_t part[1024][3];
    const int tid = threadIdx.x;
    {   // (the count kernel wrote sums for the blocks of records there are; beyond them the array holds nothing)
        const uint32_t r1 = min(a.r1_cap, a.counts_in->n_records);
        n_blocks = min(n_blocks, r1 > a.r0 ? (r1 - a.r0 + PACK_BLOCK - 1) / PACK_BLOCK : 0u);
    }
    const uint32_t per = (n_blocks + 1023u) / 1024u, lo = (uint32_t)tid * per, hi = min(lo + per, n_blocks);
    uint32_t s[3] = {0, 0, 0};
    for (uint32_t k = lo; k < hi; k++)
        for (int c = 0; c < 3; c++) s[c] += a.blk_sums[3 * k + c];
    for (int c = 0; c < 3; c++) part[tid][c] = s[c];
    __syncthreads();
    if (tid < 3) {
        uint32_t run = 0;
        for (int k = 0; k < 1024; k++) { const uint32_t t = part[k][tid]; part[k][tid] = run; run += t; }
        if (tid == 0) a.counts->n_sent = run;
        if (tid == 1) a.counts->n_cig = run;
        if (tid == 2) a.counts->n_seq = run;
    }
    __syncthreads();
    uint32_t at[3] = {part[tid][0], part[tid][1], part[tid][2]};
    for (uint32_t k = lo; k < hi; k++)
        for (int c = 0; c < 3; c++) { a.blk_base[3 * k + c] = at[c]; at[c] += a.blk_sums[3 * k + c]; }
}

// thread per record: the sent records' fields, CIGARs and bases into the batch arrays
__global__ __launch_bounds__(PACK_BLOCK) void bam_pack_write_kernel(PackArgs a) {
    __shared__ uint32_t tmp[16];
    const uint32_t n_all = a.counts_in->n_records;
    const uint32_t r1 = min(a.r1_cap, n_all);
    const uint32_t i = a.r0 + blockIdx.x * PACK_BLOCK + threadIdx.x;
    const bool live = i < r1;
    const uint32_t info = live ? a.info[i - a.r0] : 0u;
    const bool need = (info & INFO_NEED) != 0;
    RecHdr r;
    uint32_t ncig = 0, nseq = 0;
    if (need) {
        r = rec_header(a.u + a.rec_off[i]);
        ncig = r.ncig;
        nseq = ((uint32_t)r.lseq + 1u) / 2u;
    }
    uint32_t tot;
    const uint32_t k = a.blk_base[3 * blockIdx.x + 0] + block_scan_1024(need ? 1u : 0u, tmp, &tot);
    const uint32_t c0 = a.blk_base[3 * blockIdx.x + 1] + block_scan_1024(ncig, tmp, &tot);
    const uint32_t q0 = a.blk_base[3 * blockIdx.x + 2] + block_scan_1024(nseq, tmp, &tot);
    if (live) a.sent_of[i - a.r0] = need ? (int32_t)k : -1;
    if (need) {
        a.tid[k] = r.tid;
        a.pos[k] = r.pos;
        a.lseq[k] = r.lseq;
        a.flag[k] = (uint16_t)r.flag;
        a.has_sa[k] = (info & INFO_SA) ? 1 : 0;
        a.cigar_off[k] = c0;
        a.seq_off[k] = q0;
        for (uint32_t j = 0; j < ncig; j++) a.cigar_ops[c0 + j] = ld32(r.p + r.cig_off + 4u * j);
        const uint8_t *s = r.p + r.seq_off;
        for (uint32_t j = 0; j < nseq; j++) a.seq[q0 + j] = s[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.cigar_off[a.counts->n_sent] = a.counts->n_cig;
        a.seq_off[a.counts->n_sent] = a.counts->n_seq;
    }
}

// ============================================================================================ tags
// art_of[sent index] = index of the record's alignment entry when it is an artifact call (anno.d:94-107)
__global__ void bam_art_index_kernel(const fadehip_aln *aln, const uint32_t *n_aln_dev, uint32_t aln_cap, int32_t *art_of, uint32_t n_sent) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = min(*n_aln_dev, aln_cap);
    if (k >= n) return;
    const fadehip_aln &a = aln[k];
    if (a.art && a.read_idx >= 0 && (uint32_t)a.read_idx < n_sent) art_of[a.read_idx] = (int32_t)k;
}

struct Names {
    const char *text;         // contig names back to back
    const uint32_t *off;      // [n + 1]
    int32_t n;
};

__device__ __forceinline__ uint32_t dec_len(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10ull) { v /= 10ull; n++; }
    return n;
}
// a byte sink that either counts or writes
struct Sink {
    uint8_t *d;  // nullptr: count only
    uint32_t n;
    __device__ __forceinline__ void put(uint8_t c) {
        if (d) d[n] = c;
        n++;
    }
    __device__ __forceinline__ void dec(int64_t v) {
        if (v < 0) { put('-'); v = -v; }
        const uint32_t len = dec_len((uint64_t)v);
        if (d) {
            uint64_t x = (uint64_t)v;
            for (uint32_t k = len; k-- > 0;) { d[n + k] = (uint8_t)('0' + x % 10ull); x /= 10ull; }
        }
        n += len;
    }
};

// What anno.d:94-107 adds to one record, given the device's results (one thread; artifact calls are a few per cent of
// the records and their strings a few hundred bytes).  The strings: analysis.d:84-92 / 108-118 + anno.d:98-107, as
// fade_main.cpp:artifact_strings builds them on the host.
struct ArtStrings {
    const RecHdr *r;
    const fadehip_aln *a;
    const Names *nm;
    int lq, nops;
    int64_t apos;
    int64_t plen_l, plen_r;  // -1: that side is absent
    __device__ __forceinline__ void init(const RecHdr *r_, const fadehip_aln *a_, const Names *nm_) {
        r = r_; a = a_; nm = nm_;
        lq = r->lseq;
        nops = min(a->sw.n_ops, FADEHIP_MAX_OPS);
        apos = a->win_start + a->sw.beg_ref;
        const int64_t pos = r->pos;
        plen_l = plen_r = -1;
        if (a->art & 1) {  // analysis.d:84-92
            const int64_t clip = a->clip_left;
            const int64_t overlap = apos >= pos - clip ? apos - (pos - clip) : 0;
            const int64_t lead = (nops > 0 && (a->sw.ops[0] & 15u) == 4u) ? (int64_t)(a->sw.ops[0] >> 4) : 0;
            plen_l = min((int64_t)lq, ((int64_t)lq - lead) + overlap);
            plen_l = max(plen_l, (int64_t)0);
        }
        if (a->art & 2) {  // analysis.d:108-118
            const int64_t clip = a->clip_right;
            int64_t res_al = 0;
            for (int q = 0; q < nops; q++) {
                const uint32_t op = a->sw.ops[q] & 15u;
                if (FADEHIP_OP_CONSUMES_REF(op)) res_al += a->sw.ops[q] >> 4;
            }
            const int64_t lhs = pos + a->aligned_len + clip, rhs = apos + res_al;
            const int64_t overlap = lhs >= rhs ? lhs - rhs : 0;
            const int64_t trail = (nops > 0 && (a->sw.ops[nops - 1] & 15u) == 4u) ? (int64_t)(a->sw.ops[nops - 1] >> 4) : 0;
            plen_r = min((int64_t)lq, ((int64_t)lq - trail) + overlap);
            plen_r = max(plen_r, (int64_t)0);
        }
    }
    __device__ __forceinline__ void am_side(Sink &s) const {
        if (r->tid >= 0 && r->tid < nm->n) {
            for (uint32_t k = nm->off[r->tid]; k < nm->off[r->tid + 1]; k++) s.put((uint8_t)nm->text[k]);
        } else s.put('*');
        s.put(',');
        s.dec(apos);
        s.put(',');
        for (int k = 0; k < nops; k++) {
            s.dec((int64_t)(a->sw.ops[k] >> 4));
            s.put((uint8_t)"MIDNSHP=XB"[min(a->sw.ops[k] & 15u, 9u)]);
        }
    }
    __device__ __forceinline__ uint8_t base_code(int j) const { return (r->p[r->seq_off + ((uint32_t)j >> 1)] >> ((~j & 1) << 2)) & 15; }
    // which: 0 am, 1 as, 2 ar, 3 ab — the string "left;right" without its NUL
    __device__ __forceinline__ void string(int which, Sink &s) const {
        const char *nt16 = "=ACMGRSVTWYHKDBN";
        const uint8_t comp[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};  // util.d:18-20
        const uint8_t *ql = r->p + r->qual_off;
        if (which == 0) {
            if (plen_l >= 0) am_side(s);
            s.put(';');
            if (plen_r >= 0) am_side(s);
            return;
        }
        if (plen_l >= 0) {
            const int pl = (int)plen_l;
            if (which == 1) for (int j = 0; j < pl; j++) s.put((uint8_t)nt16[base_code(j)]);                        // seq[0 : plen]
            if (which == 2) for (int m = lq - pl; m < lq; m++) s.put((uint8_t)nt16[comp[base_code(lq - 1 - m)]]);   // qrc[lq - plen :]
            if (which == 3) for (int j = 0; j < pl; j++) s.put((uint8_t)(ql[j] + 33));                              // bq[0 : plen]
        }
        s.put(';');
        if (plen_r >= 0) {
            const int pr = (int)plen_r;
            if (which == 1) for (int j = lq - pr; j < lq; j++) s.put((uint8_t)nt16[base_code(j)]);                   // seq[lq - plen :]
            if (which == 2) for (int m = 0; m < pr; m++) s.put((uint8_t)nt16[comp[base_code(lq - 1 - m)]]);         // qrc[0 : plen]
            if (which == 3) for (int j = lq - pr; j < lq; j++) s.put((uint8_t)(ql[j] + 33));                        // bq[lq - plen :]
        }
    }
};

// The record as it leaves: block_size, the record's bytes, and the tags of anno.d:63,94-107 — appended when absent,
// updated the way htslib's bam_aux_update_int / bam_aux_update_str do when the record already carries them (first
// occurrence, in place or replaced at the same position).  One thread; `out` = nullptr counts.  Returns the bytes.
__device__ __forceinline__ uint32_t emit_record(const RecHdr &r, uint32_t info, uint8_t rs, const fadehip_aln *a, const Names *nm, uint8_t *out) {
    Sink s{out, 4u};  // (block_size is written last)
    ArtStrings st;
    if (a) st.init(&r, a, nm);
    const char tags[5][2] = {{'r', 's'}, {'a', 'm'}, {'a', 's'}, {'a', 'r'}, {'a', 'b'}};
    bool done[5] = {false, a == nullptr, a == nullptr, a == nullptr, a == nullptr};  // (no artifact: the strings are not touched)
    if (!(info & INFO_OURS)) {
        // the body is copied by the caller's wave (rewrite kernel) or counted here
        s.n += r.bs;
    } else {
        // fixed part up to the aux area, then field by field
        for (uint32_t k = 4; k < r.aux_off; k++) s.put(r.p[k]);
        uint32_t q = r.aux_off;
        while (q + 3u <= r.end) {
            const uint32_t fs = aux_field_size(r.p, q + 2u, r.end);
            if (!fs) break;
            int mine = -1;
            for (int t = 0; t < 5; t++)
                if (!done[t] && r.p[q] == (uint8_t)tags[t][0] && r.p[q + 1] == (uint8_t)tags[t][1]) mine = t;
            if (mine < 0) {
                for (uint32_t k = q; k < q + 2u + fs; k++) s.put(r.p[k]);
            } else if (mine == 0) {
                // bam_aux_update_int (htslib sam.c) of a value 0 .. 63: every integer slot is wide enough, it is reused and
                // its type letter becomes the unsigned one of its size ("\0CS\0I"[old_sz]); a non-integer rs: EINVAL, the
                // field stays as it is and nothing is appended
                done[0] = true;
                const uint8_t ot = r.p[q + 2];
                const uint32_t os = (uint32_t)aux_type_size(ot);
                const bool is_int = ot == 'c' || ot == 'C' || ot == 's' || ot == 'S' || ot == 'i' || ot == 'I';
                if (is_int) {
                    s.put('r'); s.put('s');
                    s.put(os == 1u ? 'C' : os == 2u ? 'S' : 'I');
                    s.put(rs);
                    for (uint32_t k = 1; k < os; k++) s.put(0);
                } else {
                    for (uint32_t k = q; k < q + 2u + fs; k++) s.put(r.p[k]);
                }
            } else {
                done[mine] = true;
                if (r.p[q + 2] == 'Z') {
                    s.put((uint8_t)tags[mine][0]); s.put((uint8_t)tags[mine][1]); s.put('Z');
                    st.string(mine - 1, s);
                    s.put(0);
                } else {  // bam_aux_update_str on a tag of another type: EINVAL, unchanged
                    for (uint32_t k = q; k < q + 2u + fs; k++) s.put(r.p[k]);
                }
            }
            q += 2u + fs;
        }
    }
    if (!done[0]) { s.put('r'); s.put('s'); s.put('C'); s.put(rs); }  // bam_aux_update_int of a ubyte: the smallest type
    for (int t = 1; t < 5; t++)
        if (!done[t]) {
            s.put((uint8_t)tags[t][0]); s.put((uint8_t)tags[t][1]); s.put('Z');
            st.string(t - 1, s);
            s.put(0);
        }
    if (out) {
        const uint32_t bs = s.n - 4u;
        out[0] = (uint8_t)bs; out[1] = (uint8_t)(bs >> 8); out[2] = (uint8_t)(bs >> 16); out[3] = (uint8_t)(bs >> 24);
    }
    return s.n;
}

struct TagArgs {
    const uint8_t *u;
    const uint32_t *rec_off;
    const ChunkCounts *counts_in;
    uint32_t r0, r1_cap;
    const uint32_t *info;
    const int32_t *sent_of;
    const uint8_t *rs;         // per sent record
    const fadehip_aln *aln;
    const int32_t *art_of;     // per sent record: alignment entry or -1
    Names names;
    uint32_t *out_size;        // per record of the batch
    uint64_t *blk_sums, *blk_base;
    ChunkCounts *counts;       // out_bytes
    uint64_t out_base;         // bytes of O in front of this batch
    uint8_t *o;
};
constexpr int TAG_BLOCK = 256;

__device__ __forceinline__ const fadehip_aln *aln_of(const TagArgs &a, int32_t sent, uint8_t *rs) {
    *rs = 0;
    if (sent < 0) return nullptr;
    *rs = a.rs[sent];
    const int32_t k = a.art_of[sent];
    return k >= 0 ? a.aln + k : nullptr;
}

__global__ __launch_bounds__(TAG_BLOCK) void bam_tag_size_kernel(TagArgs a) {
    __shared__ uint64_t red[TAG_BLOCK / 64];
    const uint32_t r1 = min(a.r1_cap, a.counts_in->n_records);
    const uint32_t i = a.r0 + blockIdx.x * TAG_BLOCK + threadIdx.x;
    uint64_t sz = 0;
    if (i < r1) {
        const RecHdr r = rec_header(a.u + a.rec_off[i]);
        uint8_t rs;
        const fadehip_aln *al = aln_of(a, a.sent_of[i - a.r0], &rs);
        const uint32_t info = a.info[i - a.r0];
        sz = (info & INFO_BAD) ? 0u : emit_record(r, info, rs, al, &a.names, nullptr);
        a.out_size[i - a.r0] = (uint32_t)sz;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sz += (uint64_t)__shfl_xor((long long)sz, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sz;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < TAG_BLOCK / 64; w++) t += red[w];
        a.blk_sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void bam_tag_scan_kernel(TagArgs a, uint32_t n_blocks) {
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u, lo = (uint32_t)tid * per, hi = min(lo + per, n_blocks);
    uint64_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += a.blk_sums[k];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 1024; k++) { const uint64_t t = part[k]; part[k] = run; run += t; }
        a.counts->out_bytes = run;
    }
    __syncthreads();
    uint64_t at = part[tid];
    for (uint32_t k = lo; k < hi; k++) { a.blk_base[k] = at; at += a.blk_sums[k]; }
}

// Sixteen lanes per record, four records per wavefront at a time: the body moved as (unaligned) dwords by the sixteen, the
// tags written by the first of them.  A record costs a chain of dependent loads (info, offset, header, then the body) —
// a wave that took its records one after the other spent 22 us on each; four chains side by side, and a third of the copy
// instructions.  A block serves the TAG_BLOCK records of one tag-size block (whose base applies) with REWRITE_WAVES waves.
constexpr int REWRITE_WAVES = 16;
__global__ __launch_bounds__(REWRITE_WAVES * 64) void bam_rewrite_kernel(TagArgs a) {
    __shared__ uint64_t off[TAG_BLOCK];
    __shared__ uint64_t wave_sum[TAG_BLOCK / 64 + 1];
    const uint32_t r1 = min(a.r1_cap, a.counts_in->n_records);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = a.r0 + blockIdx.x * TAG_BLOCK;
    // offsets of the block's records: a scan of out_size by the first TAG_BLOCK threads
    uint64_t sz = 0, inc = 0;
    if (threadIdx.x < TAG_BLOCK) {
        const uint32_t i = i0 + threadIdx.x;
        sz = i < r1 ? (uint64_t)a.out_size[i - a.r0] : 0ull;
        inc = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = (uint64_t)__shfl_up((long long)inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wave_sum[wave + 1] = inc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        wave_sum[0] = 0;
        for (int w = 1; w <= TAG_BLOCK / 64; w++) wave_sum[w] += wave_sum[w - 1];
    }
    __syncthreads();
    if (threadIdx.x < TAG_BLOCK) off[threadIdx.x] = a.out_base + a.blk_base[blockIdx.x] + wave_sum[wave] + inc - sz;
    __syncthreads();
    constexpr int PER_WAVE = TAG_BLOCK / REWRITE_WAVES;
    static_assert(PER_WAVE % 4 == 0, "four records per wavefront at a time");
    const uint32_t sub = (uint32_t)lane >> 4, sl = (uint32_t)lane & 15u;
    for (int j = 0; j < PER_WAVE / 4; j++) {
        const uint32_t k = (uint32_t)wave * PER_WAVE + 4u * (uint32_t)j + sub, ij = i0 + k;
        if (ij >= r1) continue;
        const uint32_t info = a.info[ij - a.r0];
        if (info & INFO_BAD) continue;
        const RecHdr r = rec_header(a.u + a.rec_off[ij]);
        uint8_t *dst = a.o + off[k];
        if (!(info & INFO_OURS)) {
            const uint32_t nbody = r.end - 4u, whole = nbody & ~3u;
            const uint8_t *from = r.p + 4;
            uint8_t *to = dst + 4;
            for (uint32_t q = 4u * sl; q < whole; q += 64u) *reinterpret_cast<u32u *>(to + q) = ld32(from + q);
            if (sl < nbody - whole) to[whole + sl] = from[whole + sl];
        }
        if (sl == 0) {
            uint8_t rs;
            const fadehip_aln *al = aln_of(a, a.sent_of[ij - a.r0], &rs);
            emit_record(r, info, rs, al, &a.names, dst);  // (a record without our tags: its body is counted, not written, here)
        }
    }
}

}  // namespace bam
}  // namespace fadehip
