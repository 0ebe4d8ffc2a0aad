// bgzf_deflate.hpp — BGZF (SAM spec §4.1) compression on gfx950: one 256-thread workgroup per BGZF block of up to 0xff00
// input bytes, the whole block resident in LDS (153 KB of the CU's 160 KB: one workgroup per CU, 256 blocks in flight).
//
// What it replaces: htslib's bgzf_write -> zlib deflate behind `SAMWriter(..., SAMWriterTypes.BAM)` (source/util.d:65-76),
// i.e. the serialised write at source/anno.d:47-49 — 7 of the 12.7 core-seconds `fade annotate` spent per 10 M reads.
//
// Per block (host/selftest/gpu_deflate_model.cpp is the same algorithm on the CPU, checked with zlib's inflate):
//   A  matches.  Pieces of 64 positions, a wave each, take turns at the hash heads (4-way buckets of 16-bit positions, 4 Ki
//      buckets): a piece's lookups see every earlier piece's inserts.  Nearer than that, distances 1..8 are tried directly
//      (runs, short periods).  Candidates are extended in LDS; the wave then waits for its turn to parse its piece greedily
//      (a match yields to a longer one at the next position), on 64-bit lane masks in scalar registers: token bitmap, match
//      bitmap, match records.
//   B  symbol histograms (8 sub-histograms against same-address LDS atomics), minimum-redundancy code lengths (Moffat &
//      Katajainen in place, one lane per alphabet), 15-bit limit, canonical codes.
//   C  the dynamic-block header, one lane, while the others count their tokens' bits.
//   D  256 position ranges emit their tokens at scanned bit offsets straight into the block's output slot; a word that
//      two ranges share is OR-ed atomically.  A block that would not shrink is stored.
//   CRC-32 of the input by slicing-by-4 over 256 pieces, combined with the x^(8n) mod P arithmetic of bgzf_huff.hpp.
// A second kernel scans the block sizes and a third assembles the BGZF members (header, payload, CRC32, ISIZE) into one
// contiguous byte stream: what goes to the file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bgzf_huff.hpp"

namespace fadehip {
namespace bgzf {

constexpr int BLOCK = 0xff00;  // input bytes per BGZF block (htslib's BGZF_BLOCK_SIZE)
constexpr int WG = 256;
constexpr int HASH_BITS = 12, WAYS = 4;
constexpr int MAX_MATCHES = 8192;
constexpr int MIN_MATCH = 4, MAX_MATCH = 258;
constexpr int N_WORDS = (BLOCK + 31) / 32;  // words of a per-position bitmap
constexpr int SLOT = 65536;                 // bytes of a block's output slot (payload <= 65510: BSIZE is 16 bits)
constexpr int MAX_PAYLOAD = 65536 - 26;

// LDS layout (bytes)
constexpr int L_DATA = 0, L_HEAD = 65536, L_MATCH = L_HEAD + 32768, L_TOK = L_MATCH + 32768, L_MAT = L_TOK + 8192,
              L_MISC = L_MAT + 8192, LDS_BYTES = L_MISC + 6144;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup must fit the CU's LDS");
// ... of the head region once the matches are found
constexpr int H_MPRE = 0, H_H8 = 8192, H_AL = H_H8 + 8 * 320 * 4, H_SL = H_AL + 288 * 4, H_AD = H_SL + 288 * 4, H_SD = H_AD + 32 * 4,
              H_CRCT = H_SD + 32 * 4, H_END = H_CRCT + 4096;
static_assert(H_END <= 32768, "phase B temporaries must fit the hash region");

struct Misc {  // the small arrays of a block
    uint32_t turn1, turn2, carry, mcount, full, blk, m_l, m_d, hdr_bits, total_bits, stored, pad[5];
    uint32_t freq_l[288], freq_d[32];
    uint8_t ll[288], dl[32];
    uint16_t lc[288], dc[32];
    uint32_t hdr[160];
    uint32_t scan[264];
    uint32_t bl_l[16], bl_d[16], nc_l[16], nc_d[16];
    uint32_t x2n[32];
    uint8_t cl_sym[320], cl_ext[320];
    uint32_t sortbuf[64];
    uint32_t crc_part[4];
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
};

__device__ __forceinline__ uint32_t lds_load32u(const uint8_t *base, uint32_t p) {  // 4 bytes at any offset
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base) + (p >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], p & 3u);
}
__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - HASH_BITS); }

// length of the match between positions c < p (their first four bytes are known to be equal), at most maxlen
__device__ __forceinline__ uint32_t match_len(const uint8_t *data, uint32_t c, uint32_t p, uint32_t maxlen) {
    uint32_t len = 4;
    while (len < maxlen) {
        const uint32_t x = lds_load32u(data, c + len) ^ lds_load32u(data, p + len);
        if (x) {
            len += (uint32_t)__builtin_ctz(x) >> 3;
            break;
        }
        len += 4;
    }
    return len < maxlen ? len : maxlen;
}

__device__ __forceinline__ void spin_until(uint32_t *turn, uint32_t v) {
    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != v) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void publish(uint32_t *turn, uint32_t v) {
    __hip_atomic_store(turn, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// exclusive scan of one value per thread over the workgroup (tmp: 4 words of LDS); *total = the sum
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
    for (int w = 0; w < WG / 64; w++) {
        const uint32_t t = tmp[w];
        if (w < wave) base += t;
        sum += t;
    }
    *total = sum;
    return base + inc - v;
}

// bits of the token that starts at bit b of bitmap word w (a literal, or the match whose record the match bitmap counts to)
__device__ __forceinline__ void token_bits(const uint8_t *data, const uint32_t *matw, const uint32_t *mpre, const uint32_t *match,
                                           const Misc *ms, int w, int b, uint64_t &bits, int &nb) {
    const uint32_t mw = matw[w];
    if ((mw >> b) & 1u) {
        const uint32_t rec = match[mpre[w] + (uint32_t)__builtin_popcount(mw & ((1u << b) - 1u))];
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

__global__ __launch_bounds__(WG) void bgzf_deflate_kernel(DeflateArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint8_t *const data = lds + L_DATA;
    uint16_t *const head = reinterpret_cast<uint16_t *>(lds + L_HEAD);
    uint32_t *const match = reinterpret_cast<uint32_t *>(lds + L_MATCH);
    uint32_t *const tokw = reinterpret_cast<uint32_t *>(lds + L_TOK);
    uint32_t *const matw = reinterpret_cast<uint32_t *>(lds + L_MAT);
    Misc *const ms = reinterpret_cast<Misc *>(lds + L_MISC);
    uint32_t *const mpre = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_MPRE);
    uint32_t *const h8 = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_H8);
    uint32_t *const A_l = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_AL), *const S_l = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_SL);
    uint32_t *const A_d = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_AD), *const S_d = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_SD);
    uint32_t *const crct = reinterpret_cast<uint32_t *>(lds + L_HEAD + H_CRCT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) crc_x2n_table(ms->x2n);
    for (;;) {
        __syncthreads();  // the previous block's LDS is no longer read
        if (tid == 0) ms->blk = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const uint32_t blk = ms->blk;
        if (blk >= a.n_blocks) break;  // (uniform: every thread reads the same word)
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
            for (int k = n + tid; k < ((n + 15) & ~15) + 272 && k < 65536; k += WG) data[k] = 0;  // what a compare may read past the end
            uint4 *z = reinterpret_cast<uint4 *>(lds + L_HEAD);
            for (int k = tid; k < 32768 / 16; k += WG) z[k] = make_uint4(0, 0, 0, 0);
            uint4 *zb = reinterpret_cast<uint4 *>(lds + L_TOK);
            for (int k = tid; k < 16384 / 16; k += WG) zb[k] = make_uint4(0, 0, 0, 0);
            if (tid == 0) { ms->turn1 = 0; ms->turn2 = 0; ms->carry = 0; ms->mcount = 0; ms->full = 0; ms->stored = 0; }
        }
        __syncthreads();

        // ---- A: matches and the parse, 64 positions per wave and turn
        const int n_pieces = (n + 63) >> 6;
        for (int piece = wave; piece < n_pieces; piece += WG / 64) {
            const uint32_t p = (uint32_t)piece * 64u + (uint32_t)lane;
            const bool valid = (int)p + MIN_MATCH <= n;
            const uint32_t v = valid ? lds_load32u(data, p) : 0u;
            const uint32_t h = hash4(v);
            uint2 *bucket = reinterpret_cast<uint2 *>(head) + h;
            spin_until(&ms->turn1, (uint32_t)piece);
            uint2 bk = make_uint2(0, 0);
            if (valid) {
                bk = *bucket;
                *bucket = make_uint2((p + 1u) | (bk.x << 16), (bk.x >> 16) | (bk.y << 16));  // newest first; the oldest of the four leaves
            }
            if (lane == 0) publish(&ms->turn1, (uint32_t)piece + 1u);
            uint32_t len = 0, dist = 0;
            if (valid) {
                const uint32_t maxlen = (uint32_t)min(MAX_MATCH, n - (int)p);
                for (uint32_t d = 1; d <= 8u && d <= p; d++)
                    if (lds_load32u(data, p - d) == v) {
                        len = match_len(data, p - d, p, maxlen);
                        dist = d;
                        break;
                    }
                const uint32_t c4[4] = {bk.x & 0xffffu, bk.x >> 16, bk.y & 0xffffu, bk.y >> 16};
#pragma unroll
                for (int w = 0; w < WAYS; w++) {
                    if (c4[w]) {
                        const uint32_t c = c4[w] - 1u, d2 = p - c;
                        if (d2 <= 32768u && lds_load32u(data, c) == v) {
                            const uint32_t ln = match_len(data, c, p, maxlen);
                            if (ln > len) { len = ln; dist = d2; }
                        }
                    }
                }
            }
            if (len < (uint32_t)MIN_MATCH) len = 0;
            // a match yields to a longer one at the next position
            const uint32_t len_next = (uint32_t)__shfl_down((int)len, 1, 64);
            const bool yield = len && lane < 63 && len_next > len;
            // -- the parse of this piece, in turn
            spin_until(&ms->turn2, (uint32_t)piece);
            const int cb = piece * 64, nv = min(64, n - cb);
            int carry = (int)ms->carry;
            uint32_t mcount = ms->mcount, full = ms->full;
            uint64_t has = __ballot(len != 0 && !yield);
            if (full || mcount + (uint32_t)__popcll(has) > (uint32_t)MAX_MATCHES) { full = 1; has = 0; }  // the match list is full: literals from here on
            const uint64_t vmask = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
            uint64_t tokmask = 0, matmask = 0;
            int cur = max(carry - cb, 0);
            while (cur < nv) {
                const uint64_t rem = has & (~0ull << cur);
                if (!rem) {
                    tokmask |= vmask & (~0ull << cur);
                    cur = nv;
                    break;
                }
                const int j = (int)__builtin_ctzll(rem);
                tokmask |= (j == 63 ? ~0ull : ((1ull << (j + 1)) - 1ull)) & (~0ull << cur);
                matmask |= 1ull << j;
                cur = j + (int)__builtin_amdgcn_readlane((int)len, j);
            }
            if (cb + cur > carry) carry = cb + cur;
            if ((matmask >> lane) & 1ull)
                match[mcount + (uint32_t)__popcll(matmask & ((1ull << lane) - 1ull))] = dist | ((len - 3u) << 16);
            if (lane == 0) {
                tokw[2 * piece] = (uint32_t)tokmask;
                tokw[2 * piece + 1] = (uint32_t)(tokmask >> 32);
                matw[2 * piece] = (uint32_t)matmask;
                matw[2 * piece + 1] = (uint32_t)(matmask >> 32);
                ms->carry = (uint32_t)carry;
                ms->mcount = mcount + (uint32_t)__popcll(matmask);
                ms->full = full;
                publish(&ms->turn2, (uint32_t)piece + 1u);
            }
        }
        __syncthreads();

        // ---- B: match-index prefix, histograms
        {
            uint32_t *z = reinterpret_cast<uint32_t *>(lds + L_HEAD);
            for (int k = tid; k < H_END / 4; k += WG) z[k] = 0;
            if (tid < 16) { ms->bl_l[tid] = 0; ms->bl_d[tid] = 0; }
        }
        __syncthreads();
        const int w0 = 8 * tid, w1 = min(w0 + 8, N_WORDS);
        {
            uint32_t cnt = 0;
            for (int w = w0; w < w1; w++) cnt += (uint32_t)__builtin_popcount(matw[w]);
            uint32_t all;
            uint32_t at = block_excl_scan(cnt, ms->crc_part, &all);
            for (int w = w0; w < w1; w++) { mpre[w] = at; at += (uint32_t)__builtin_popcount(matw[w]); }
        }
        {
            uint32_t *hl = h8 + (lane & 7) * 320;
            for (int w = w0; w < w1; w++) {
                uint32_t tw = tokw[w];
                const uint32_t mw = matw[w];
                while (tw) {
                    const int b = __builtin_ctz(tw);
                    tw &= tw - 1u;
                    if ((mw >> b) & 1u) {
                        const uint32_t rec = match[mpre[w] + (uint32_t)__builtin_popcount(mw & ((1u << b) - 1u))];
                        atomicAdd(&hl[length_symbol((rec >> 16) + 3u).sym], 1u);
                        atomicAdd(&hl[288 + dist_symbol(rec & 0xffffu).sym], 1u);
                    } else atomicAdd(&hl[data[32 * w + b]], 1u);
                }
            }
        }
        __syncthreads();
        for (int s = tid; s < 320; s += WG) {
            uint32_t f = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) f += h8[k * 320 + s];
            if (s < 288) ms->freq_l[s] = s == 256 ? 1u : (s < NUM_LITLEN ? f : 0u);
            else ms->freq_d[s - 288] = (s - 288 < NUM_DIST) ? f : 0u;
        }
        __syncthreads();
        if (tid == 0) {  // at least two distance codes (as zlib makes sure of, for old inflaters)
            int used = 0;
            for (int s = 0; s < NUM_DIST; s++) used += ms->freq_d[s] != 0;
            for (int s = 0; used < 2 && s < NUM_DIST; s++)
                if (!ms->freq_d[s]) { ms->freq_d[s] = 1; used++; }
            ms->m_l = 0;
            ms->m_d = 0;
        }
        __syncthreads();
        // rank the used symbols by (frequency, symbol)
        for (int s = tid; s < NUM_LITLEN; s += WG) {
            const uint32_t f = ms->freq_l[s];
            if (f) {
                uint32_t r = 0;
                for (int j = 0; j < NUM_LITLEN; j++) {
                    const uint32_t g = ms->freq_l[j];
                    r += g && (g < f || (g == f && j < s));
                }
                A_l[r] = f;
                S_l[r] = (uint32_t)s;
                atomicAdd(&ms->m_l, 1u);
            }
        }
        if (tid < NUM_DIST) {
            const uint32_t f = ms->freq_d[tid];
            if (f) {
                uint32_t r = 0;
                for (int j = 0; j < NUM_DIST; j++) {
                    const uint32_t g = ms->freq_d[j];
                    r += g && (g < f || (g == f && j < tid));
                }
                A_d[r] = f;
                S_d[r] = (uint32_t)tid;
                atomicAdd(&ms->m_d, 1u);
            }
        }
        for (int s = tid; s < 288; s += WG) ms->ll[s] = 0;
        if (tid < 32) ms->dl[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            mr_code_lengths(A_l, (int)ms->m_l);
            limit_code_lengths(A_l, (int)ms->m_l, MAX_LITLEN_BITS, ms->sortbuf);
        }
        if (tid == 64) {
            mr_code_lengths(A_d, (int)ms->m_d);
            limit_code_lengths(A_d, (int)ms->m_d, MAX_LITLEN_BITS, ms->sortbuf + 32);
        }
        __syncthreads();
        for (uint32_t k = tid; k < ms->m_l; k += WG) {
            ms->ll[S_l[k]] = (uint8_t)A_l[k];
            atomicAdd(&ms->bl_l[A_l[k]], 1u);
        }
        if ((uint32_t)tid < ms->m_d) {
            ms->dl[S_d[tid]] = (uint8_t)A_d[tid];
            atomicAdd(&ms->bl_d[A_d[tid]], 1u);
        }
        __syncthreads();
        if (tid == 0 || tid == 64) {  // first code of each length (RFC 1951 §3.2.2)
            uint32_t *bl = tid ? ms->bl_d : ms->bl_l, *nc = tid ? ms->nc_d : ms->nc_l;
            uint32_t c = 0;
            bl[0] = 0;
            nc[0] = 0;
            for (int b = 1; b <= MAX_LITLEN_BITS; b++) {
                c = (c + bl[b - 1]) << 1;
                nc[b] = c;
            }
        }
        __syncthreads();
        for (int s = tid; s < NUM_LITLEN + NUM_DIST; s += WG) {
            const bool is_d = s >= NUM_LITLEN;
            const int sym = is_d ? s - NUM_LITLEN : s, nsym = is_d ? NUM_DIST : NUM_LITLEN;
            const uint8_t *lens = is_d ? ms->dl : ms->ll;
            const uint32_t l = lens[sym];
            (void)nsym;
            if (l) {
                uint32_t before = 0;
                for (int j = 0; j < sym; j++) before += lens[j] == l;
                const uint32_t code = (is_d ? ms->nc_d : ms->nc_l)[l] + before;
                (is_d ? ms->dc : ms->lc)[sym] = (uint16_t)(__builtin_bitreverse32(code) >> (32u - l));
            }
        }
        for (int k = tid; k < 160; k += WG) ms->hdr[k] = 0;
        __syncthreads();

        // ---- C + D: the header (one lane) beside the bit counts of the 256 position ranges
        if (tid == 0) {
            BitW bw{ms->hdr, 0};
            ms->hdr_bits = write_dynamic_header(bw, ms->ll, ms->dl, ms->cl_sym, ms->cl_ext, ms->sortbuf);
        }
        uint32_t my_bits = 0;
        for (int w = w0; w < w1; w++) {
            uint32_t tw = tokw[w];
            while (tw) {
                const int b = __builtin_ctz(tw);
                tw &= tw - 1u;
                uint64_t v;
                int k;
                token_bits(data, matw, mpre, match, ms, w, b, v, k);
                my_bits += (uint32_t)k;
            }
        }
        if (tid == WG - 1) my_bits += ms->ll[256];  // end of block
        __syncthreads();  // hdr_bits is there
        uint32_t tok_bits_all;
        const uint32_t b0 = ms->hdr_bits + block_excl_scan(my_bits, ms->crc_part, &tok_bits_all);
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
            // words that more than one writer touches are cleared first and OR-ed atomically; the others are stored whole
            if (my_bits) {
                out32[b0 >> 5] = 0;
                out32[(b1 - 1u) >> 5] = 0;
            }
            for (uint32_t k = tid; k < hdr_words; k += WG) out32[k] = 0;
            __syncthreads();
            for (uint32_t k = tid; k < hdr_words; k += WG) atomicOr(&out32[k], ms->hdr[k]);
            if (my_bits) {
                uint64_t acc = 0;
                int cnt = (int)(b0 & 31u);
                uint32_t wi = b0 >> 5;
                const uint32_t w_first = wi, w_last = (b1 - 1u) >> 5;
                auto put = [&](uint64_t v, int k) {
                    acc |= v << cnt;
                    cnt += k;
                    if (cnt >= 32) {
                        if (wi == w_first || wi == w_last) atomicOr(&out32[wi], (uint32_t)acc);
                        else out32[wi] = (uint32_t)acc;
                        wi++;
                        acc >>= 32;
                        cnt -= 32;
                    }
                };
                for (int w = w0; w < w1; w++) {
                    uint32_t tw = tokw[w];
                    while (tw) {
                        const int b = __builtin_ctz(tw);
                        tw &= tw - 1u;
                        uint64_t v;
                        int k;
                        token_bits(data, matw, mpre, match, ms, w, b, v, k);
                        if (k > 24) {  // (a token has up to 48 bits and the accumulator up to 31 pending)
                            put(v & 0xffffffull, 24);
                            put(v >> 24, k - 24);
                        } else put(v, k);
                    }
                }
                if (tid == WG - 1) put(ms->lc[256], ms->ll[256]);
                if (cnt) atomicOr(&out32[wi], (uint32_t)acc);
            }
            if (tid == 0) a.out_size[blk] = (total_bits + 7u) >> 3;
        }

        // ---- CRC-32 of the input: slicing-by-4 over 256 pieces of 256 bytes, combined
        for (int k = tid; k < 256; k += WG) crct[k] = crc_table_entry((uint32_t)k);
        __syncthreads();
        for (int t = 1; t < 4; t++) {
            crct[256 * t + tid] = (crct[256 * (t - 1) + tid] >> 8) ^ crct[crct[256 * (t - 1) + tid] & 255u];
            __syncthreads();
        }
        uint32_t part = 0;
        {
            const int lo = 256 * tid, hi = min(lo + 256, n);
            if (lo < hi) {
                uint32_t c = 0xffffffffu;
                int k = lo;
                const uint32_t *dw = reinterpret_cast<const uint32_t *>(data);
                for (; k + 4 <= hi; k += 4) {
                    c ^= dw[k >> 2];
                    c = crct[768 + (c & 255u)] ^ crct[512 + ((c >> 8) & 255u)] ^ crct[256 + ((c >> 16) & 255u)] ^ crct[c >> 24];
                }
                for (; k < hi; k++) c = crct[(c ^ data[k]) & 255u] ^ (c >> 8);
                c = ~c;
                part = crc_mulmod(crc_x8n((uint32_t)(n - hi), ms->x2n), c);  // crc(A || B) = x^(8 |B|) crc(A) ^ crc(B)
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) part ^= (uint32_t)__shfl_xor((int)part, m, 64);
        if (lane == 0) ms->crc_part[wave] = part;
        __syncthreads();
        if (tid == 0) a.out_crc[blk] = ms->crc_part[0] ^ ms->crc_part[1] ^ ms->crc_part[2] ^ ms->crc_part[3];
    }
}

// exclusive scan of the members' sizes (payload + 26 bytes of BGZF header and trailer): one workgroup
__global__ __launch_bounds__(1024) void bgzf_scan_kernel(const uint32_t *out_size, uint32_t n_blocks, uint64_t *member_off, uint64_t *total) {
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u, lo = (uint32_t)tid * per, hi = min(lo + per, n_blocks);
    uint64_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += (uint64_t)out_size[k] + 26u;
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 1024; k++) { const uint64_t c = part[k]; part[k] = run; run += c; }
        *total = run;
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

}  // namespace bgzf
}  // namespace fadehip
