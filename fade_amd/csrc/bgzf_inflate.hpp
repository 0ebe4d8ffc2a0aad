// bgzf_inflate.hpp — BGZF members inflated on the device (gfx950): the reader side of the file path, what htslib's
// bgzf_read -> inflate does under dhtslib's SAMReader (anno.d:44, bam.allRecords).
//
// A DEFLATE stream decodes serially (every code's position depends on the one before), so the unit of parallelism is the
// BGZF block: ONE WAVEFRONT PER BLOCK, thousands of blocks in flight (a 1.6 GB BAM has 50,000 of them; the device holds
// 8,192 waves).  Everything a wave decides is wave-uniform — the bit buffer, the symbol, the branch — and is kept in
// scalar registers (values that come back from LDS go through v_readfirstlane); the 64 lanes are used as
//   * the input window: 64 dwords of the stream in one VGPR, the next 64 in another, read with v_readlane (no memory
//     latency on the dependent chain but the table look-up itself),
//   * the literal buffer: up to 64 literals collected in one VGPR with v_writelane and stored as one 64-byte line,
//   * the copier of a match (lane i moves byte i; an overlapping match reads i mod distance),
//   * the builders of the decoding tables (every table entry decodes its own index canonically, 16 entries per lane) and
//     of the CRC-32 (a segment per lane, combined by x^(8n) mod P).
// Tables (per wave, 3.9 KB of LDS): 10-bit literal/length and 8-bit distance tables of u16 entries (symbol << 4 | code
// length), longer codes by canonical decoding (count[] per length + symbols sorted by code), as zlib's puff.c does for
// every code.  Every loop consumes input bits or produces output bytes and stops at the member's ends, so a corrupt
// stream ends in a status code, never in a hang.  The output of a block is read back by its own wave only (matches,
// CRC): vector memory operations of one wavefront are performed in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgzf_deflate.hpp"
#include "bgzf_huff.hpp"

namespace fadehip {
namespace bgzf {

struct InflateBlock {  // one per BGZF member, from the host's scan of the member headers and trailers
    uint64_t src_off;  // first byte of the member's DEFLATE stream in `comp`
    uint64_t dst_off;  // where its bytes go in `out` (running sum of ISIZE)
    uint32_t src_len;  // bytes of DEFLATE stream
    uint32_t isize;    // ISIZE of the trailer
    uint32_t crc;      // CRC32 of the trailer
    uint32_t pad;
};
static_assert(sizeof(InflateBlock) == 32, "InflateBlock layout");

struct InflateArgs {
    const uint8_t *comp;         // the members, with at least 1 KB of readable bytes behind the last one
    const InflateBlock *blocks;
    uint32_t n_blocks;
    uint8_t *out;
    const uint64_t *out_shift;   // device-side: bytes added to every dst_off (bytes carried over in front), or nullptr
    uint32_t *status;            // [n_blocks] 0 = fine, INF_E_* otherwise
    uint32_t *ticket;            // [0] ticket, [1] number of failed blocks
    int check_crc;
};

enum : uint32_t {
    INF_E_BTYPE = 1, INF_E_STORED = 2, INF_E_HEADER = 3, INF_E_CODE = 4, INF_E_DIST = 5, INF_E_OVERRUN_OUT = 6,
    INF_E_OVERRUN_IN = 7, INF_E_SIZE = 8, INF_E_CRC = 9,
};

constexpr int INF_WG = 256, INF_WAVES = INF_WG / 64;
constexpr int LIT_BITS = 10, DIST_BITS = 8, CL_BITS = 7;

struct InfWave {
    uint16_t lit_tab[1 << LIT_BITS];
    uint16_t dist_tab[1 << DIST_BITS];
    uint16_t cl_tab[1 << CL_BITS];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t cl_sorted[20];
    uint32_t count[3][16];  // codes per length: literal/length, distance, code-length alphabet
    uint8_t lens[320];
    uint8_t cl_lens[24];
};
struct InfShared {
    uint32_t crct[1024];
    uint32_t x2n[32];
    InfWave w[INF_WAVES];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// (LDS written by some lanes, read by others of the same wave: the ds operations of a wave complete in order; the fence
// keeps the compiler from moving them and waits for the writes)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Canonical decoding of one code from the low bits of `bits` (first code bit = bit 0), puff.c's loop: lengths 1..15.
// Returns symbol << 4 | length, 0 if no code matches.
__device__ __forceinline__ uint32_t canon_decode(uint32_t bits, const uint32_t *count, const uint16_t *sorted, int max_len) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= max_len; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int cnt = (int)count[len];
        if (code - cnt < first) return ((uint32_t)sorted[index + (code - first)] << 4) | (uint32_t)len;
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}

// lens[0..n) -> count[], sorted[], tab[0 .. 1 << T).  The whole wave; false when the lengths over-subscribe the code space.
__device__ __forceinline__ bool build_table(const uint8_t *lens, int n, uint32_t *count, uint16_t *sorted, uint16_t *tab, int T, int lane) {
    if (lane < 16) count[lane] = 0;
    wave_lds_sync();
    for (int s = lane; s < n; s += 64) {
        const uint32_t l = lens[s];
        if (l) atomicAdd(&count[l], 1u);
    }
    wave_lds_sync();
    // lane L keeps the position of the next symbol of length L in sorted[]
    uint32_t offs = 0;
    int left = 1;  // (every lane runs the same check)
    for (int l = 1; l < 16; l++) {
        const uint32_t c = count[l];
        if (l < lane) offs += c;
        left = (left << 1) - (int)c;
        if (left < 0) left = -(1 << 20);
    }
    if ((int)uni((uint32_t)left) < 0) return false;  // (uniform: the caller's loops must not look divergent to the compiler)
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t l = s < n ? (uint32_t)lens[s] : 0u;
        for (int L = 1; L < 16; L++) {
            const unsigned long long m = __ballot(l == (uint32_t)L);
            if (m == 0ull) continue;
            const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)offs, L);
            if (l == (uint32_t)L) sorted[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)s;
            if (lane == L) offs += (uint32_t)__popcll(m);
        }
    }
    wave_lds_sync();
    for (int e = lane; e < (1 << T); e += 64) tab[e] = (uint16_t)canon_decode((uint32_t)e, count, sorted, T);
    wave_lds_sync();
    return true;
}

// The member's bits, least significant first: 64 dwords of the stream per VGPR, two VGPRs (the second is fetched while
// the first is consumed).  Nothing beyond the member is read: a stream that runs past its end decodes zeros until the
// output bound stops it, and is then reported.
struct BitReader {
    const uint32_t *base;  // 4-byte aligned, at or before the first byte of the stream
    uint32_t win, winn;    // per lane: dword ws + lane, dword ws + 64 + lane
    uint32_t ws, wi;       // uniform: first dword of `win`, next dword to take from it
    uint32_t last_dw;      // last dword that holds bytes of the member: nothing beyond it is read (zeros instead)
    uint64_t bb;           // uniform
    int bc;                // uniform: valid bits in bb
    int lane;
    __device__ __forceinline__ void start(const uint8_t *p, uint32_t n_bytes, int lane_) {
        lane = lane_;
        const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
        base = reinterpret_cast<const uint32_t *>(p - mis);
        last_dw = (mis + n_bytes + 3u) >> 2;  // (one dword of the member's trailer may be read too)
        seek(mis);
    }
    __device__ __forceinline__ uint32_t fetch(uint32_t dw) const { return dw <= last_dw ? base[dw] : 0u; }
    // continue at byte `at` (relative to base)
    __device__ __forceinline__ void seek(uint32_t at) {
        const uint32_t d = at >> 2;
        ws = d & ~63u;
        wi = d - ws;
        win = fetch(ws + (uint32_t)lane);
        winn = fetch(ws + 64u + (uint32_t)lane);
        const uint32_t skip = (at & 3u) * 8u;
        bb = (uint64_t)(take32() >> skip);
        bc = 32 - (int)skip;
    }
    __device__ __forceinline__ uint32_t take32() {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)wi);
        wi++;
        if (wi == 64u) {
            win = winn;
            ws += 64u;
            winn = fetch(ws + 64u + (uint32_t)lane);
            wi = 0;
        }
        return v;
    }
    __device__ __forceinline__ void refill() {  // at least 32 valid bits afterwards
        if (bc < 32) {
            bb |= (uint64_t)take32() << bc;
            bc += 32;
        }
    }
    __device__ __forceinline__ uint32_t bits(int n) {  // n <= 16, after refill()
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        bc -= n;
        return v;
    }
    // bytes of the stream consumed, counting partially used bytes as consumed (relative to base)
    __device__ __forceinline__ uint32_t byte_pos() const { return (ws + wi) * 4u - ((uint32_t)bc >> 3); }
};

__device__ __forceinline__ void lane_write(uint32_t &reg, uint32_t value, uint32_t lane_sel) {
    // v_writelane takes one scalar operand from the constant bus: the lane select goes through m0
    const uint32_t v = uni(value), l = uni(lane_sel);
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(v), "s"(l) : "m0");
}

// One member.  Returns 0 or INF_E_*.
__device__ __forceinline__ uint32_t inflate_member(InfWave *w, const uint8_t *src, uint32_t src_len, uint8_t *out, uint32_t isize, int lane) {
    BitReader br;
    br.start(src, src_len, lane);
    const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t in_limit = mis + src_len + 8u;  // byte_pos() beyond this: the stream ran past its member
    uint32_t pos = 0;    // bytes produced, pending literals included
    uint32_t pend = 0;   // lane k: the k-th pending literal
    uint32_t pn = 0;     // pending literals
    uint32_t err = 0;
    bool last = false;
    // a DEFLATE block needs at least 3 bits of header: the member's bits bound the number of blocks
    for (uint32_t guard = 0; !last && !err && guard <= 8u * src_len + 8u; guard++) {
        br.refill();
        last = br.bits(1) != 0;
        const uint32_t btype = br.bits(2);
        if (btype == 3u) { err = INF_E_BTYPE; break; }
        if (btype == 0u) {
            // stored: to the byte boundary, LEN, NLEN, LEN bytes
            br.bits(br.bc & 7);
            br.refill();
            const uint32_t len = br.bits(16);
            br.refill();
            const uint32_t nlen = br.bits(16);
            if ((len ^ nlen) != 0xffffu) { err = INF_E_STORED; break; }
            const uint32_t at = br.byte_pos();
            if (at + len > in_limit) { err = INF_E_OVERRUN_IN; break; }
            if (pos + len > isize) { err = INF_E_OVERRUN_OUT; break; }
            if (pn) {
                if ((uint32_t)lane < pn) out[pos - pn + (uint32_t)lane] = (uint8_t)pend;
                pn = 0;
            }
            const uint8_t *s = reinterpret_cast<const uint8_t *>(br.base) + at;
            for (uint32_t k = (uint32_t)lane; k < len; k += 64u) out[pos + k] = s[k];
            pos += len;
            br.seek(at + len);
            continue;
        }
        const uint16_t *lit_tab = w->lit_tab, *dist_tab = w->dist_tab;
        if (btype == 1u) {
            for (int s = lane; s < 288; s += 64) w->lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            if (lane < 32) w->lens[288 + lane] = 5;
            wave_lds_sync();
            if (!build_table(w->lens, 288, w->count[0], w->lit_sorted, w->lit_tab, LIT_BITS, lane) ||
                !build_table(w->lens + 288, 32, w->count[1], w->dist_sorted, w->dist_tab, DIST_BITS, lane)) { err = INF_E_HEADER; break; }
        } else {
            br.refill();
            const uint32_t hlit = br.bits(5) + 257u, hdist = br.bits(5) + 1u, hclen = br.bits(4) + 4u;
            if (hlit > 286u || hdist > 30u) { err = INF_E_HEADER; break; }
            if (lane < 19) w->cl_lens[lane] = 0;
            wave_lds_sync();
            for (uint32_t k = 0; k < hclen; k++) {
                br.refill();
                const uint32_t v = br.bits(3);
                if (lane == 0) w->cl_lens[cl_order((int)k)] = (uint8_t)v;
            }
            wave_lds_sync();
            if (!build_table(w->cl_lens, 19, w->count[2], w->cl_sorted, w->cl_tab, CL_BITS, lane)) { err = INF_E_HEADER; break; }
            const uint32_t total = hlit + hdist;
            uint32_t k = 0, prev = 0;
            while (k < total && !err) {
                br.refill();
                const uint32_t e = uni((uint32_t)w->cl_tab[(uint32_t)br.bb & ((1u << CL_BITS) - 1u)]);
                const uint32_t cl = e & 15u, sym = e >> 4;
                if (cl == 0u) { err = INF_E_HEADER; break; }
                br.bits((int)cl);
                uint32_t rep = 1, val = sym;
                if (sym == 16u) {
                    if (k == 0u) { err = INF_E_HEADER; break; }
                    rep = 3u + br.bits(2);
                    val = prev;
                } else if (sym == 17u) {
                    rep = 3u + br.bits(3);
                    val = 0;
                } else if (sym == 18u) {
                    rep = 11u + br.bits(7);
                    val = 0;
                }
                if (k + rep > total) { err = INF_E_HEADER; break; }
                if ((uint32_t)lane < rep) w->lens[k + (uint32_t)lane] = (uint8_t)val;
                if (rep > 64u && (uint32_t)lane + 64u < rep) w->lens[k + 64u + (uint32_t)lane] = (uint8_t)val;
                if (rep > 128u && (uint32_t)lane + 128u < rep) w->lens[k + 128u + (uint32_t)lane] = (uint8_t)val;
                k += rep;
                prev = val;
            }
            if (err) break;
            wave_lds_sync();
            if (uni((uint32_t)w->lens[256]) == 0u) { err = INF_E_HEADER; break; }
            if (!build_table(w->lens, (int)hlit, w->count[0], w->lit_sorted, w->lit_tab, LIT_BITS, lane) ||
                !build_table(w->lens + hlit, (int)hdist, w->count[1], w->dist_sorted, w->dist_tab, DIST_BITS, lane)) { err = INF_E_HEADER; break; }
        }
        // ---- the block's symbols
        bool eob = false;
        while (!eob && !err) {
            // Literal runs first, in a loop of their own with one exit: a table entry e is a literal with a short code iff
            // 1 <= e < 4096 (symbol << 4 | length, length >= 1).  The run ends when the literal line is full, so the line's
            // room in the output is checked once, not per literal; 32 valid bits serve three codes of at most 10 bits.
            if (pos - pn + 64u <= isize) {
                bool lit = true;
#define FADEHIP_INF_LITERAL()                                                                       \
    {                                                                                               \
        const uint32_t e_ = uni((uint32_t)lit_tab[(uint32_t)br.bb & ((1u << LIT_BITS) - 1u)]);      \
        if (e_ - 1u >= 4095u) { lit = false; break; }                                               \
        br.bits((int)(e_ & 15u));                                                                   \
        lane_write(pend, e_ >> 4, pn);                                                              \
        pn++;                                                                                       \
        pos++;                                                                                      \
        if (pn == 64u) break;                                                                       \
    }
                for (;;) {
                    br.refill();
                    FADEHIP_INF_LITERAL()
                    FADEHIP_INF_LITERAL()
                    FADEHIP_INF_LITERAL()
                }
#undef FADEHIP_INF_LITERAL
                if (pn == 64u) {
                    out[pos - 64u + (uint32_t)lane] = (uint8_t)pend;
                    pn = 0;
                    if (lit) continue;
                }
            }
            br.refill();
            uint32_t e = uni((uint32_t)lit_tab[(uint32_t)br.bb & ((1u << LIT_BITS) - 1u)]);
            if ((e & 15u) == 0u) {
                e = canon_decode((uint32_t)br.bb, w->count[0], w->lit_sorted, 15);
                e = uni(e);
                if (e == 0u) { err = INF_E_CODE; break; }
            }
            br.bits((int)(e & 15u));
            uint32_t sym = e >> 4;
            if (sym < 256u) {
                if (pos >= isize) { err = INF_E_OVERRUN_OUT; break; }
                lane_write(pend, sym, pn);
                pn++;
                pos++;
                if (pn == 64u) {
                    out[pos - 64u + (uint32_t)lane] = (uint8_t)pend;
                    pn = 0;
                }
                continue;
            }
            if (sym == 256u) { eob = true; break; }
            sym -= 257u;
            if (sym >= 29u) { err = INF_E_CODE; break; }
            uint32_t length;
            if (sym < 4u) length = 3u + sym;
            else if (sym == 28u) length = 258u;
            else {
                const uint32_t eb = (sym >> 2) - 1u;
                length = 3u + ((4u + (sym & 3u)) << eb) + br.bits((int)eb);
            }
            br.refill();
            uint32_t d = uni((uint32_t)dist_tab[(uint32_t)br.bb & ((1u << DIST_BITS) - 1u)]);
            if ((d & 15u) == 0u) {
                d = uni(canon_decode((uint32_t)br.bb, w->count[1], w->dist_sorted, 15));
                if (d == 0u) { err = INF_E_CODE; break; }
            }
            br.bits((int)(d & 15u));
            const uint32_t dsym = d >> 4;
            if (dsym >= 30u) { err = INF_E_CODE; break; }
            uint32_t dist;
            if (dsym < 2u) dist = 1u + dsym;
            else {
                const uint32_t eb = (dsym >> 1) - 1u;
                dist = 1u + ((2u + (dsym & 1u)) << eb) + br.bits((int)eb);
            }
            if (dist > pos) { err = INF_E_DIST; break; }
            if (pos + length > isize) { err = INF_E_OVERRUN_OUT; break; }
            if (pn) {
                if ((uint32_t)lane < pn) out[pos - pn + (uint32_t)lane] = (uint8_t)pend;
                pn = 0;
            }
            const uint8_t *from = out + (pos - dist);
            uint8_t *to = out + pos;
            if (dist >= length) {
                for (uint32_t k = (uint32_t)lane; k < length; k += 64u) to[k] = from[k];
            } else {
                for (uint32_t k = (uint32_t)lane; k < length; k += 64u) to[k] = from[k % dist];
            }
            pos += length;
            if (br.byte_pos() > in_limit) err = INF_E_OVERRUN_IN;
        }
        if (!err && br.byte_pos() > in_limit) err = INF_E_OVERRUN_IN;
    }
    if (!err && !last) err = INF_E_OVERRUN_IN;
    if (pn && !err) {
        if ((uint32_t)lane < pn) out[pos - pn + (uint32_t)lane] = (uint8_t)pend;
    }
    if (!err && pos != isize) err = INF_E_SIZE;
    return err;
}

// CRC-32 of out[0..n): a segment per lane (slicing by 4 from the workgroup's LDS tables), combined
__device__ __forceinline__ uint32_t wave_crc32(const InfShared *sh, const uint8_t *p, uint32_t n, int lane) {
    const uint32_t seg = (((n + 63u) / 64u) + 3u) & ~3u;
    const uint32_t lo = min(seg * (uint32_t)lane, n), hi = min(lo + seg, n);
    uint32_t part = 0;
    if (lo < hi) {
        const uint32_t *crct = sh->crct;
        uint32_t c = 0xffffffffu, k = lo;
        for (; k < hi && ((uintptr_t)(p + k) & 3u); k++) c = crct[(c ^ p[k]) & 255u] ^ (c >> 8);
        for (; k + 4u <= hi; k += 4u) {
            c ^= *reinterpret_cast<const uint32_t *>(p + k);
            c = crct[768u + (c & 255u)] ^ crct[512u + ((c >> 8) & 255u)] ^ crct[256u + ((c >> 16) & 255u)] ^ crct[c >> 24];
        }
        for (; k < hi; k++) c = crct[(c ^ p[k]) & 255u] ^ (c >> 8);
        part = crc_mulmod(crc_x8n(n - hi, sh->x2n), ~c);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part ^= (uint32_t)__shfl_xor((int)part, m, 64);
    return part;
}

__global__ __launch_bounds__(INF_WG) void bgzf_inflate_kernel(InflateArgs a) {
    __shared__ InfShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sh.crct[tid] = crc_table_entry((uint32_t)tid);
    if (tid == 0) crc_x2n_table(sh.x2n);
    __syncthreads();
    for (int t = 1; t < 4; t++) {
        sh.crct[256 * t + tid] = (sh.crct[256 * (t - 1) + tid] >> 8) ^ sh.crct[sh.crct[256 * (t - 1) + tid] & 255u];
        __syncthreads();
    }
    InfWave *w = &sh.w[wave];
    const uint64_t shift = a.out_shift ? *a.out_shift : 0ull;
    bool live = true;
    while (live) {
        const uint32_t b = uni((uint32_t)claim_ticket(a.ticket));
        if (b >= a.n_blocks) {
            live = false;
        } else {
            const InflateBlock blk = a.blocks[b];
            uint8_t *dst = a.out + blk.dst_off + shift;
            // (a member that claims ISIZE 0 goes through the decoder as well: any byte of output is an error there, and its
            // CRC32 must be that of no bytes — a zeroed trailer must not make a member's records vanish without a word)
            uint32_t err = inflate_member(w, a.comp + blk.src_off, blk.src_len, dst, blk.isize, lane);
            if (!err && a.check_crc) {
                const uint32_t c = blk.isize ? wave_crc32(&sh, dst, blk.isize, lane) : 0u;
                if (c != blk.crc) err = INF_E_CRC;
            }
            if (lane == 0) {
                a.status[b] = err;
                if (err) atomicAdd(a.ticket + 1, 1u);
            }
        }
    }
}

}  // namespace bgzf
}  // namespace fadehip
