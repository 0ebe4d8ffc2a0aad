// bgzf_deflate.hpp — BGZF (SAM spec §4.1) compression on gfx950: one 1024-thread workgroup per BGZF block of up to 0xff00
// input bytes, the whole block resident in LDS (153 KB of the CU's 160 KB: one workgroup per CU, 256 blocks in flight;
// the LDS footprint leaves room for one workgroup only, so the sixteen waves that hide its latencies come from that one).
//
// What it replaces: htslib's bgzf_write -> zlib deflate behind `SAMWriter(..., SAMWriterTypes.BAM)` (source/util.d:65-76),
// i.e. the serialised write at source/anno.d:47-49 — 7 of the 12.7 core-seconds `fade annotate` spent per 10 M reads.
//
// Per block (host/selftest/gpu_deflate_model.cpp is the same algorithm on the CPU, checked with zlib's inflate):
//   A  matches.  Pieces of 64 positions, a wave each, take turns at the hash heads (4-way buckets of 16-bit positions, 4 Ki
//      buckets): a piece's lookups see every earlier piece's inserts.  Nearer than that, distances 1..8 are tried directly
//      (runs, short periods).  The up to five candidates of a position are extended side by side in LDS; the wave then
//      waits for its turn to parse its piece greedily (a match yields to a longer one at the next position), on 64-bit
//      lane masks in scalar registers: token bitmap, match bitmap, match records.
//   B  symbol histograms (8 sub-histograms against same-address LDS atomics), minimum-redundancy code lengths (Moffat &
//      Katajainen in place, the array spread over the lanes of a wave), 15-bit limit, canonical codes.
//   C  the dynamic-block header, one wave (lane arrays again), while the others count their tokens' bits.
//   D  1024 position ranges emit their tokens at scanned bit offsets straight into the block's output slot; a word that
//      two ranges share is OR-ed atomically.  A block that would not shrink is stored.
//   CRC-32 of the input by slicing-by-4 over 1024 pieces, combined with the x^(8n) mod P arithmetic of bgzf_huff.hpp.
// A second kernel scans the block sizes and a third assembles the BGZF members (header, payload, CRC32, ISIZE) into one
// contiguous byte stream: what goes to the file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bgzf_huff.hpp"

namespace fadehip {
namespace bgzf {

constexpr int BLOCK = 0xff00;  // input bytes per BGZF block (htslib's BGZF_BLOCK_SIZE)
constexpr int WG = 1024;
constexpr int N_WAVES = WG / 64;
constexpr int HASH_BITS = 12, WAYS = 4;
constexpr int MAX_MATCHES = 8192;
constexpr int MIN_MATCH = 4, MAX_MATCH = 258;
constexpr int N_WORDS = (BLOCK + 31) / 32;   // words of a per-position bitmap
constexpr int WPT = (N_WORDS + WG - 1) / WG;  // ... per thread in the per-range phases (a range = 64 positions)
constexpr int SLOT = 65536;                  // bytes of a block's output slot (payload <= 65510: BSIZE is 16 bits)
constexpr int MAX_PAYLOAD = 65536 - 26;

// LDS layout (bytes)
constexpr int L_DATA = 0, L_HEAD = 65536, L_MATCH = L_HEAD + 32768, L_TOK = L_MATCH + 32768, L_MAT = L_TOK + 8192,
              L_MISC = L_MAT + 8192, LDS_BYTES = L_MISC + 6144;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup must fit the CU's LDS");
// ... of the head region once the matches are found
constexpr int H_MPRE = 0, H_H8 = 8192, H_AL = H_H8 + 8 * 320 * 4, H_SL = H_AL + 320 * 4, H_AD = H_SL + 320 * 4, H_SD = H_AD + 64 * 4,
              H_CRCT = H_SD + 64 * 4, H_END = H_CRCT + 4096;
static_assert(H_END <= 32768, "phase B temporaries must fit the hash region");

struct Misc {  // the small arrays of a block
    uint32_t turn1, turn2, carry, mcount, full, blk, m_l, m_d, hdr_bits, total_bits, stored, pad[5];
    uint32_t freq_l[320], freq_d[64];
    uint8_t ll[320], dl[64];
    uint16_t lc[320], dc[64];
    uint32_t hdr[160];
    uint32_t bl_l[16], bl_d[16], nc_l[16], nc_d[16];
    uint32_t x2n[32];
    uint32_t sortbuf[64];
    uint32_t wtmp[N_WAVES];
    uint32_t crc_part[N_WAVES];
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
__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - HASH_BITS); }

__device__ __forceinline__ void spin_until(uint32_t *turn, uint32_t v) {
    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != v) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void publish(uint32_t *turn, uint32_t v) {
    __hip_atomic_store(turn, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
            for (int k = n + tid; k < ((n + 15) & ~15) + 272 && k < 65536; k += WG) data[k] = 0;  // what a compare may read past the end
            uint4 *z = reinterpret_cast<uint4 *>(lds + L_HEAD);
            for (int k = tid; k < 32768 / 16; k += WG) z[k] = make_uint4(0, 0, 0, 0);
            uint4 *zb = reinterpret_cast<uint4 *>(lds + L_TOK);
            for (int k = tid; k < 16384 / 16; k += WG) zb[k] = make_uint4(0, 0, 0, 0);
            if (tid == 0) { ms->turn1 = 0; ms->turn2 = 0; ms->carry = 0; ms->mcount = 0; ms->full = 0; ms->stored = 0; }
        }
        __syncthreads();
        stamp(0);

        // ---- A: matches and the parse, 64 positions per wave and turn
        const int n_pieces = (n + 63) >> 6;
        for (int piece = wave; piece < n_pieces; piece += N_WAVES) {
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
            // a position that an earlier match already covers can start no token: its matches are never looked at (the
            // parse below starts at `carry`, which only grows), so they need not be found either
            const uint32_t covered_to = __hip_atomic_load(&ms->carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (valid && p >= covered_to) {
                const uint32_t maxlen = (uint32_t)min(MAX_MATCH, n - (int)p);
                // up to five candidates: the nearest of the distances 1 .. 8 whose four bytes agree, and the bucket's four
                // positions; their loads are issued together and they are extended side by side (one LDS round trip per
                // four bytes of the LONGEST match, not per candidate)
                uint32_t cp[5];
                uint32_t alive = 0;
                {
                    uint32_t vd[8];
#pragma unroll
                    for (uint32_t d = 1; d <= 8u; d++) vd[d - 1] = lds_load32u(data, p >= d ? p - d : p);
                    uint32_t dsmall = 0;
#pragma unroll
                    for (uint32_t d = 8; d >= 1u; d--)
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
                while (alive && off < maxlen) {
                    const uint32_t pw = lds_load32u(data, p + off);
                    uint32_t x[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) x[k] = lds_load32u(data, cp[k] + off) ^ pw;
#pragma unroll
                    for (int k = 0; k < 5; k++)
                        if (((alive >> k) & 1u) && x[k]) {
                            cl[k] = off + ((uint32_t)__builtin_ctz(x[k]) >> 3);
                            alive &= ~(1u << k);
                        }
                    off += 4;
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    uint32_t l = ((alive >> k) & 1u) ? maxlen : cl[k];  // still equal where the limit was reached
                    if (l > maxlen) l = maxlen;
                    if (l > len) { len = l; dist = p - cp[k]; }
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
                __hip_atomic_store(&ms->carry, (uint32_t)carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ms->mcount = mcount + (uint32_t)__popcll(matmask);
                ms->full = full;
                publish(&ms->turn2, (uint32_t)piece + 1u);
            }
        }
        __syncthreads();
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
            uint32_t at = block_excl_scan(cnt, ms->wtmp, &all);
#pragma unroll
            for (int k = 0; k < WPT; k++) {
                mb_r[k] = at;
                at += (uint32_t)__builtin_popcount(mw_r[k]);
            }
            (void)mpre;
        }
        {
            uint32_t *hl = h8 + (lane & 7) * 320;
#pragma unroll
            for (int k = 0; k < WPT; k++) {
                uint32_t tw = tw_r[k];
                const uint32_t mw = mw_r[k];
                while (tw) {
                    const int b = __builtin_ctz(tw);
                    tw &= tw - 1u;
                    if ((mw >> b) & 1u) {
                        const uint32_t rec = match[mb_r[k] + (uint32_t)__builtin_popcount(mw & ((1u << b) - 1u))];
                        atomicAdd(&hl[length_symbol((rec >> 16) + 3u).sym], 1u);
                        atomicAdd(&hl[288 + dist_symbol(rec & 0xffffu).sym], 1u);
                    } else atomicAdd(&hl[data[32 * (w0 + k) + b]], 1u);
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
        // rank the used symbols by (frequency, symbol)
        if (tid < NUM_LITLEN) {
            const uint32_t f = ms->freq_l[tid];
            if (f) {
                uint32_t r = 0;
                for (int j = 0; j < NUM_LITLEN; j++) {
                    const uint32_t g = ms->freq_l[j];
                    r += g && (g < f || (g == f && j < tid));
                }
                A_l[r] = f;
                S_l[r] = (uint32_t)tid;
                atomicAdd(&ms->m_l, 1u);
            }
        } else if (tid >= 512 && tid < 512 + NUM_DIST) {
            const int sd = tid - 512;
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
        // minimum-redundancy lengths: wave 0 the literal / length alphabet, wave 1 the distances, arrays in lane registers
        if (wave == 0) {
            WaveArr<5> A;
#pragma unroll
            for (int k = 0; k < 5; k++) A.r[k] = A_l[64 * k + lane];
            const int m = (int)ms->m_l;
            mr_code_lengths_t(A, m);
#pragma unroll
            for (int k = 0; k < 5; k++) A_l[64 * k + lane] = A.r[k];
            if (lane == 0 && A_l[0] > (uint32_t)MAX_LITLEN_BITS) limit_code_lengths(A_l, m, MAX_LITLEN_BITS, ms->sortbuf);  // (rare)
        } else if (wave == 1) {
            WaveArr<1> A;
            A.r[0] = A_d[lane];
            const int m = (int)ms->m_d;
            mr_code_lengths_t(A, m);
            A_d[lane] = A.r[0];
            if (lane == 0 && A_d[0] > (uint32_t)MAX_LITLEN_BITS) limit_code_lengths(A_d, m, MAX_LITLEN_BITS, ms->sortbuf + 32);
        }
        __syncthreads();
        if ((uint32_t)tid < ms->m_l) {
            ms->ll[S_l[tid]] = (uint8_t)A_l[tid];
            atomicAdd(&ms->bl_l[A_l[tid]], 1u);
        } else if (tid >= 512 && (uint32_t)(tid - 512) < ms->m_d) {
            ms->dl[S_d[tid - 512]] = (uint8_t)A_d[tid - 512];
            atomicAdd(&ms->bl_d[A_d[tid - 512]], 1u);
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
        if (tid < NUM_LITLEN + NUM_DIST) {
            const bool is_d = tid >= NUM_LITLEN;
            const int sym = is_d ? tid - NUM_LITLEN : tid;
            const uint8_t *lens = is_d ? ms->dl : ms->ll;
            const uint32_t l = lens[sym];
            if (l) {
                uint32_t before = 0;
                for (int j = 0; j < sym; j++) before += lens[j] == l;
                const uint32_t code = (is_d ? ms->nc_d : ms->nc_l)[l] + before;
                (is_d ? ms->dc : ms->lc)[sym] = (uint16_t)(__builtin_bitreverse32(code) >> (32u - l));
            }
        }
        __syncthreads();
        stamp(3);

        // ---- C + D: the header (wave 0, lane arrays) beside the bit counts of the 1024 position ranges
        if (wave == 0) {
            WaveArr<5> LL;
            WaveArr<1> DL, fr, sf, ss, cll, clc, bl;
#pragma unroll
            for (int k = 0; k < 5; k++) LL.r[k] = ms->ll[64 * k + lane];
            DL.r[0] = ms->dl[lane];
            fr.r[0] = sf.r[0] = ss.r[0] = cll.r[0] = clc.r[0] = bl.r[0] = 0;
            LdsSink sink;
            sink.w = ms->hdr;
            write_dynamic_header_t(sink, LL, DL, fr, sf, ss, cll, clc, bl);
            const uint32_t hb = sink.finish();
            if (lane == 0) ms->hdr_bits = hb;
        }
        uint32_t my_bits = 0;
#pragma unroll
        for (int k = 0; k < WPT; k++) {
            uint32_t tw = tw_r[k];
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
        __syncthreads();  // hdr_bits is there
        stamp(4);
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
#pragma unroll
                for (int k = 0; k < WPT; k++) {
                    uint32_t tw = tw_r[k];
                    while (tw) {
                        const int b = __builtin_ctz(tw);
                        tw &= tw - 1u;
                        uint64_t v;
                        int nb;
                        token_bits(data, mw_r[k], mb_r[k], match, ms, w0 + k, b, v, nb);
                        if (nb > 24) {  // (a token has up to 48 bits and the accumulator up to 31 pending)
                            put(v & 0xffffffull, 24);
                            put(v >> 24, nb - 24);
                        } else put(v, nb);
                    }
                }
                if (tid == WG - 1) put(ms->lc[256], ms->ll[256]);
                if (cnt) atomicOr(&out32[wi], (uint32_t)acc);
            }
            if (tid == 0) a.out_size[blk] = (total_bits + 7u) >> 3;
        }
        __syncthreads();
        stamp(5);

        // ---- CRC-32 of the input: slicing-by-4 over 1024 pieces of 64 bytes, combined
        if (tid < 256) crct[tid] = crc_table_entry((uint32_t)tid);
        __syncthreads();
        for (int t = 1; t < 4; t++) {
            if (tid < 256) crct[256 * t + tid] = (crct[256 * (t - 1) + tid] >> 8) ^ crct[crct[256 * (t - 1) + tid] & 255u];
            __syncthreads();
        }
        uint32_t part = 0;
        {
            constexpr int PIECE = 65536 / WG;
            const int lo = PIECE * tid, hi = min(lo + PIECE, n);
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
        if (tid == 0) {
            uint32_t c = 0;
            for (int w = 0; w < N_WAVES; w++) c ^= ms->crc_part[w];
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
