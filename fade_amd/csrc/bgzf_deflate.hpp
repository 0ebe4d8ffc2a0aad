// bgzf_deflate.hpp — BGZF (SAM spec §4.1) compression on gfx950: one workgroup per BGZF block, the whole block resident in
// LDS, in two geometries with a phase A of their own each (below).
//
// What it replaces: htslib's bgzf_write -> zlib deflate behind `SAMWriter(..., SAMWriterTypes.BAM)` (source/util.d:65-76),
// i.e. the serialised write at source/anno.d:47-49 — 7 of the 12.7 core-seconds `fade annotate` spent per 10 M reads.
//
// Per block (host/selftest/gpu_deflate_model.cpp is the algorithm of the smaller geometry on the CPU, checked with zlib's
// inflate; it reproduces the device's output byte for byte):
//   A  matches and the parse over pieces of 64 positions: token bitmap, match bitmap, match records.
//        0x7f00-byte blocks (bgzf_deflate_body.hpp): the pieces in eight contiguous segments, a wave each — own hash table
//        (384 buckets of four 16-bit positions, the KB in front of the segment entered first), candidates = the bucket's four
//        and the nearest of the distances 1..8, extended side by side in LDS, greedy parse with one step of laziness on
//        64-bit lane masks; a match cut at its segment's end is lengthened again at the seam.  No wave waits for another.
//        0xff00-byte blocks (bgzf_deflate_roles.hpp): round 3's pipeline of wave roles — one hasher over ONE table for the
//        whole block, six extenders, one parser, rings between them: every match of the block in reach, half the rate.
//   B  symbol histograms (8 sub-histograms against same-address LDS atomics), minimum-redundancy code lengths (Moffat &
//      Katajainen in place), 15-bit limit, canonical codes; the waves that have no part in the code lengths sum the CRC-32
//      meanwhile (slicing-by-4 over a piece per thread, combined with the x^(8n) mod P arithmetic of bgzf_huff.hpp).
//   C  the dynamic-block header, in parallel, while the others count their tokens' bits.
//   D  the threads' position ranges emit their tokens at scanned bit offsets straight into the block's output slot; a word that
//      two ranges share is OR-ed atomically.  A block that would not shrink is stored.
// A second kernel scans the block sizes and a third assembles the BGZF members (header, payload, CRC32, ISIZE) into one
// contiguous byte stream — in pinned host memory: what goes to the file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bgzf_huff.hpp"

// Two geometries.  64: htslib's 0xff00-byte blocks, the block and its tables fill the CU's LDS, one workgroup per CU, the
// role pipeline — the smallest output (9.8 GB/s of payload).  32: 0x7f00-byte blocks, 80 KB of LDS, two workgroups of eight
// waves per CU, phase A on per-wave segments (19 GB/s); the output is 0.6 % larger on uniform-quality BAM payload than the
// other geometry's and 1-2 % larger where qualities run.  fadehip.hip picks per call: 32 while the stream is mostly
// incompressible bases (ratio above 0.45: there zlib -6 is 2 % behind either), 64 otherwise, so that the output stays
// below zlib -6's in both regimes (tests/test_gpu_bgzf.py).
#define FADEHIP_BGZF_GEOM 64
#define FADEHIP_BGZF_NS bgzf64
#include "bgzf_deflate_roles.hpp"
#undef FADEHIP_BGZF_GEOM
#undef FADEHIP_BGZF_NS
#define FADEHIP_BGZF_GEOM 32
#define FADEHIP_BGZF_NS bgzf32
#include "bgzf_deflate_body.hpp"
#undef FADEHIP_BGZF_GEOM
#undef FADEHIP_BGZF_NS

namespace fadehip {
namespace bgzf {
using bgzf64::claim_ticket;  // (the inflater draws its tickets the same way)

// Uncompressed BGZF (`fade annotate -u`, what htslib writes at level 0): a member = 18 bytes of header, one stored DEFLATE
// block (5 bytes + the payload), CRC32, ISIZE.  Sizes are known in advance, so member k of a stream cut into STORE_BLOCK
// bytes lies at k * STORE_MEMBER; a workgroup per member sums the CRC (a segment per thread, slicing by 4, combined by
// x^(8n) mod P) and moves the bytes.
constexpr int STORE_BLOCK = 0xff00, STORE_MEMBER = 18 + 5 + STORE_BLOCK + 8, STORE_WG = 256;
__global__ __launch_bounds__(STORE_WG) void bgzf_store_kernel(const uint8_t *src, uint64_t n_bytes, uint32_t n_blocks, uint8_t *dst) {
    __shared__ uint32_t crct[1024];
    __shared__ uint32_t x2n[32];
    __shared__ uint32_t part[STORE_WG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t blk = blockIdx.x;
    if (blk >= n_blocks) return;
    crct[tid] = crc_table_entry((uint32_t)tid);
    if (tid == 0) crc_x2n_table(x2n);
    __syncthreads();
    for (int t = 1; t < 4; t++) {
        crct[256 * t + tid] = (crct[256 * (t - 1) + tid] >> 8) ^ crct[crct[256 * (t - 1) + tid] & 255u];
        __syncthreads();
    }
    const uint64_t off = (uint64_t)blk * STORE_BLOCK;
    const uint32_t n = (uint32_t)(n_bytes - off < (uint64_t)STORE_BLOCK ? n_bytes - off : (uint64_t)STORE_BLOCK);
    const uint8_t *p = src + off;  // (16-byte aligned: the stream starts aligned and STORE_BLOCK is a multiple of 16)
    uint8_t *m = dst + (uint64_t)blk * STORE_MEMBER;
    const uint32_t seg = (((n + STORE_WG - 1) / STORE_WG) + 3u) & ~3u;
    const uint32_t lo = min(seg * (uint32_t)tid, n), hi = min(lo + seg, n);
    uint32_t c_part = 0;
    if (lo < hi) {
        uint32_t c = 0xffffffffu, k = lo;
        for (; k + 4u <= hi; k += 4u) {
            c ^= *reinterpret_cast<const uint32_t *>(p + k);
            c = crct[768u + (c & 255u)] ^ crct[512u + ((c >> 8) & 255u)] ^ crct[256u + ((c >> 16) & 255u)] ^ crct[c >> 24];
        }
        for (; k < hi; k++) c = crct[(c ^ p[k]) & 255u] ^ (c >> 8);
        c_part = crc_mulmod(crc_x8n(n - hi, x2n), ~c);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) c_part ^= (uint32_t)__shfl_xor((int)c_part, s, 64);
    if (lane == 0) part[wave] = c_part;
    for (uint32_t k = (uint32_t)tid; k < n; k += STORE_WG) m[23 + k] = p[k];
    __syncthreads();
    if (tid == 0) {
        uint32_t crc = 0;
        for (int w = 0; w < STORE_WG / 64; w++) crc ^= part[w];
        const uint32_t bsize = 18u + 5u + n + 8u - 1u;
        const uint8_t h[23] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255u), (uint8_t)(bsize >> 8),
                               1, (uint8_t)(n & 255u), (uint8_t)(n >> 8), (uint8_t)(~n & 255u), (uint8_t)((~n >> 8) & 255u)};
        for (int k = 0; k < 23; k++) m[k] = h[k];
        uint8_t *t = m + 23 + n;
        for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(n >> (8 * k)); }
    }
}
}  // namespace bgzf
}  // namespace fadehip
